#!/bin/bash
# Config 4 under different builds of the library, alternating on one box.  usage: gpu_c4_libs.sh lib1.so lib2.so ...
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do
  for lib in "$@"; do
    IPA_LIB_PATH=$lib timeout -k 10 200 python3 bench.py --size 16k --steps 20 --warmup 3 --no-cpu --check 1 > gpurun_out/c4lib.log 2>&1 || { tail -20 gpurun_out/c4lib.log; exit 1; }
    python3 - "$lib" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/c4lib.log") if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[1].split("/")[-1], "step", d["ms_per_step"], "hyst", d["stages"]["canny_hysteresis"]["ms"], d["parity_check"])
PY
  done
done
