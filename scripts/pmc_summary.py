"""Summarise rocprofv3 --pmc csv output: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections, re
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not k.startswith("k_"):
        continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):.5g}")
