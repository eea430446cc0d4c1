#!/bin/bash
# Step time by chunk count: usage gpu_chunks.sh <batch> <k> [<k> ...]   (two rounds, alternating)
set -u
batch=$1; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 1 2; do
  for k in "$@"; do
    IPA_PIPELINE=$k timeout -k 10 400 python3 bench.py --batch $batch --steps 6 --warmup 2 --no-cpu --check 0 > gpurun_out/chunks.log 2>&1 || { tail -20 gpurun_out/chunks.log; exit 1; }
    python3 - "$batch" "$k" <<'PY'
import json,sys
l=[x for x in open("gpurun_out/chunks.log") if x.startswith("{")][-1]
d=json.loads(l)
print("batch", sys.argv[1], "chunks", sys.argv[2], "step", d["ms_per_step"], "us/img", round(d["ms_per_step"]*1e3/int(sys.argv[1]),2))
PY
  done
done
