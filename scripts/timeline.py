"""Prints the tail of a rocprofv3 kernel trace as a timeline: usage timeline.py <kernel_trace.csv> [last N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r".*(k_\w+).*", r"\1", r["Kernel_Name"])[:20]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-20s q%-2s start %9.1f end %9.1f dur %8.1f  grid %s" % (name, r.get("Queue_Id"), a, b, b - a, r.get("Grid_Size_X", "")))
