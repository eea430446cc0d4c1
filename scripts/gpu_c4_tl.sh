#!/bin/bash
# Timeline of the hysteresis kernels of one config-4 pass (serial schedule).  Environment is passed through.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export IPA_PIPELINE=0 IPA_OVERLAP=0
rm -rf gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c4 -- python3 bench.py --size 16k --steps 3 --warmup 1 --no-cpu --check 0 > gpurun_out/prof_c4.log 2>&1 || { tail -20 gpurun_out/prof_c4.log; exit 1; }
f=$(find gpurun_out/prof_c4 -name '*kernel_trace.csv' | head -1)
python3 scripts/timeline.py "$f" 60 | grep -E "hyst|canny|fill|compact" | tail -24
rm -rf gpurun_out/prof_c4
