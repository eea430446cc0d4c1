#!/bin/bash
# Like gpu_c4_env.sh with a fixed extra environment and bench arguments: usage: gpu_c4_env2.sh "<env>" "<bench args>" VAR v1 v2 ...
set -u
envs=$1; bargs=$2; var=$3; shift 3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do
  for v in "$@"; do
    env $envs $var=$v timeout -k 10 200 python3 bench.py $bargs --no-cpu --check 0 > gpurun_out/c4env.log 2>&1 || { tail -20 gpurun_out/c4env.log; exit 1; }
    python3 - "$var=$v" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/c4env.log") if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[1], "step", d["ms_per_step"], "vote", d["stages"]["hough_vote"]["ms"])
PY
  done
done
