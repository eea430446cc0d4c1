#!/bin/bash
# Config 4 (one 16384 x 16384 noisy page) under different settings of one environment variable, alternating on one box.
# usage: scripts/gpu_c4_env.sh <rounds> <VAR> <value> [<value> ...]
set -u
rounds=$1; var=$2; shift 2
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    env $var=$v timeout -k 10 200 python3 bench.py --size 16k --steps 20 --warmup 3 --no-cpu --check 0 > gpurun_out/c4env.log 2>&1 || { tail -20 gpurun_out/c4env.log; exit 1; }
    python3 - "$var=$v" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/c4env.log") if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[1], "step", d["ms_per_step"], " ".join(f"{k}={v['ms']}" for k,v in d["stages"].items()))
PY
  done
done
