#!/bin/bash
# Kernel trace of config 4 (one 16384 x 16384 noisy page): per-kernel durations of the serial schedule.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export IPA_PIPELINE=0 IPA_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -- python3 bench.py --size 16k --steps 10 --warmup 2 --no-cpu --check 0 > gpurun_out/prof_c4.log 2>&1 || { tail -20 gpurun_out/prof_c4.log; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_c4/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"])/1e3), "%6.2f%%" % float(r["Percentage"]))
PY
timeout -k 10 120 python3 scripts/probe_hyst16k.py 50 150
