#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 120 ./scripts/micro/lds_sorted > gpurun_out/lds_sorted.log 2>&1 || { tail gpurun_out/lds_sorted.log; exit 1; }
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r2a_noalloc.log 2>&1 || { tail -40 gpurun_out/r2a_noalloc.log; exit 1; }
tail -3 gpurun_out/r2a_noalloc.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_config4.py -x -v -s --timeout 600 --timeout-method thread > gpurun_out/r2a_cfg4.log 2>&1 || { tail -40 gpurun_out/r2a_cfg4.log; exit 1; }
tail -3 gpurun_out/r2a_cfg4.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_config2.py -x -v --timeout 600 --timeout-method thread > gpurun_out/r2a_cfg2.log 2>&1 || { tail -40 gpurun_out/r2a_cfg2.log; exit 1; }
tail -3 gpurun_out/r2a_cfg2.log
