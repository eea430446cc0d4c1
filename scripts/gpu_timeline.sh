#!/bin/bash
# Kernel trace of the default (overlapped) schedule and the timeline of its last step.  usage: gpu_timeline.sh [bench args]
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 3 --warmup 1 --no-cpu --check 0 "$@" > gpurun_out/prof_tl.log 2>&1 || { tail -20 gpurun_out/prof_tl.log; exit 1; }
f=$(find gpurun_out/prof_tl -name '*kernel_trace.csv' | head -1)
python3 scripts/timeline.py "$f" 140 > gpurun_out/timeline.txt
tail -1 gpurun_out/prof_tl.log | cut -c1-200
rm -rf gpurun_out/prof_tl
