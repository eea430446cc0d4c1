#!/bin/bash
# Stage durations (serial schedule) of kernel variants whose outputs are knowingly incomplete (no parity check).
# usage: scripts/ab_stage_nocheck.sh "VAR=a VAR=b ..."
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export IPA_PIPELINE=0 IPA_OVERLAP=0
i=0
for kv in $1; do
  i=$((i+1))
  env $(echo "$kv" | tr ',' ' ') timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu --check 0 > gpurun_out/ab_nc_$i.json 2> gpurun_out/ab_nc_$i.err || { echo "$kv failed"; tail -3 gpurun_out/ab_nc_$i.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_nc_$i.json').read().strip().splitlines()[-1])
print('$kv', d['ms_per_step'], ' '.join('%s=%.3f' % (k, v['ms']) for k, v in d['stages'].items() if 'canny_nms' in k or 'gray' in k))"
done
