#!/bin/bash
# Kernel durations of the Otsu threshold on the config-4 page
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_otsu
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_otsu -- python3 scripts/bench_config4_ops.py --only OTSU > gpurun_out/prof_otsu.log 2>&1 || { tail gpurun_out/prof_otsu.log; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_otsu/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"])>=10: print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%9.1f us" % (float(r["AverageNs"])/1e3))
PY
f=$(find gpurun_out/prof_otsu -name '*kernel_trace.csv' | head -1)
python3 scripts/timeline.py "$f" 14
rm -rf gpurun_out/prof_otsu
