#!/bin/bash
# step time of the front-end under schedule settings given as "VAR=val VAR=val" strings, one run each
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  cfg="$1"; shift
  env $cfg timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu --check 0 > gpurun_out/sched.log 2>&1 || { tail -20 gpurun_out/sched.log; exit 1; }
  python3 - "$cfg" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/sched.log") if x.startswith("{")][-1]
d=json.loads(l)
print(sys.argv[1], "->", d["ms_per_step"], "serial sum", round(sum(v["ms"] for v in d["stages"].values()),2), {k[:12]: round(v["ms"],2) for k,v in d["stages"].items()})
PY
done
