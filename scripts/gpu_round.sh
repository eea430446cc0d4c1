#!/bin/bash
# One GPU session: parity tests, kernel-trace stats, bench.  usage: scripts/gpu_round.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests_$tag.log 2>&1 || { tail -30 gpurun_out/gpu_tests_$tag.log; exit 1; }
tail -2 gpurun_out/gpu_tests_$tag.log
# kernel durations of the serial schedule (each stage once per step over the whole batch), the one
# bench.py's per-stage roofline figures refer to
export IPA_PIPELINE=0 IPA_OVERLAP=0
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 5 --warmup 2 --no-cpu --check 1 > gpurun_out/prof_$tag.log 2>&1 || { tail -20 gpurun_out/prof_$tag.log; exit 1; }
tail -1 gpurun_out/prof_$tag.log
unset IPA_PIPELINE IPA_OVERLAP
timeout -k 10 900 python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_$tag.log 2> gpurun_out/bench_$tag.err || { tail -20 gpurun_out/bench_$tag.err; exit 1; }
tail -1 gpurun_out/bench_$tag.log
