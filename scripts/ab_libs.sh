#!/bin/bash
# A/B of builds of the library on one box: every round runs every build once (alternating), so that clock and
# thermal drift hits all of them alike.  usage: scripts/ab_libs.sh <rounds> <stage> "<env settings>" lib1.so lib2.so ...
set -u
rounds=$1; stage=$2; envs=$3; shift 3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    env $envs IPA_LIB_PATH=$lib IPA_PIPELINE=0 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu --check 0 --batch ${IPA_AB_BATCH:-512} > gpurun_out/ab.log 2>&1 || { tail -20 gpurun_out/ab.log; exit 1; }
    python3 - "$lib" "$stage" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/ab.log") if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[1].split("/")[-1], sys.argv[2], d["stages"].get(sys.argv[2],{}).get("ms"), "step", d["ms_per_step"])
PY
  done
done
