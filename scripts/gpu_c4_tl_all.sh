#!/bin/bash
# Full kernel timeline of one config-4 pass of the DEFAULT (overlapped) schedule.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c4 -- python3 bench.py --size 16k --steps 3 --warmup 1 --no-cpu --check 0 > gpurun_out/prof_c4.log 2>&1 || { tail -20 gpurun_out/prof_c4.log; exit 1; }
f=$(find gpurun_out/prof_c4 -name '*kernel_trace.csv' | head -1)
python3 scripts/timeline.py "$f" 200 > gpurun_out/c4_timeline.txt
rm -rf gpurun_out/prof_c4
