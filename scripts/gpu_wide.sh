#!/bin/bash
# Parity tests with the wide Canny kernel forced (8 and 4 pixel pairs per lane), then stage timings.
# usage: scripts/gpu_wide.sh <tag> [library paths to time ...]
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for np in 8 4; do
  IPA_CANNY_WIDE=$np timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_images.py tests/test_gpu_config4.py tests/test_gpu_config2.py -x -q --timeout 120 --timeout-method thread -k "canny or Canny or front_end or reference_diagram or config4 or config2 or tiny or randomized" > gpurun_out/wide_tests_${tag}_np$np.log 2>&1 || { tail -40 gpurun_out/wide_tests_${tag}_np$np.log; exit 1; }
  tail -2 gpurun_out/wide_tests_${tag}_np$np.log
done
for lib in "$@"; do
for np in 4 8; do
  IPA_LIB_PATH=$lib IPA_PIPELINE=0 IPA_CANNY_WIDE=$np timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --check 1 > gpurun_out/wide_bench_${tag}_np$np.log 2>&1 || { tail -20 gpurun_out/wide_bench_${tag}_np$np.log; exit 1; }
  python3 - <<PY
import json
l=[x for x in open("gpurun_out/wide_bench_${tag}_np$np.log") if x.startswith("{")][-1]
d=json.loads(l)
print("$lib np=$np", d["ms_per_step"], {k[:12]:round(v["ms"],2) for k,v in d["stages"].items()}, d["parity_check"])
PY
done
done
