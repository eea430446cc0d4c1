#!/bin/bash
# The benchmark lines of the configurations beside the default one (1024 x 4K): a single 1080p diagram, the 4K batch
# with GaussianBlur before Canny, and the 16384 x 16384 noisy page with 101-px line kernels.
# usage: scripts/gpu_configs.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
one() {
  name=$1; shift
  timeout -k 10 500 python3 bench.py "$@" > gpurun_out/bench_${tag}_$name.log 2> gpurun_out/bench_${tag}_$name.err || { tail -20 gpurun_out/bench_${tag}_$name.err; exit 1; }
  grep '^{' gpurun_out/bench_${tag}_$name.log | tail -1 > gpurun_out/bench_${tag}_$name.json
  python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['unit'], d['ms_per_step'], 'ms', d['parity_check'])" gpurun_out/bench_${tag}_$name.json
}
one 1080p_x1 --size 1080p --steps 200 --warmup 20 && one 16k_x1 --size 16k --steps 20 --warmup 3 --check 1 && one 4k_x1024_blur --blur --steps 10 --warmup 3
