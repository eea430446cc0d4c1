#!/bin/bash
# A/B of environment settings on the stage durations of the headline batch (serial schedule).
# usage: scripts/ab_stage.sh "VAR=a VAR=b ..." (one bench run per word; prints the stage table of each)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export IPA_PIPELINE=0 IPA_OVERLAP=0
i=0
for kv in $1; do
  i=$((i+1))
  env "$kv" timeout -k 10 200 python3 bench.py --steps 4 --warmup 2 --no-cpu --check 2 > gpurun_out/ab_stage_$i.json 2> gpurun_out/ab_stage_$i.err || { echo "$kv failed"; tail -3 gpurun_out/ab_stage_$i.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_stage_$i.json').read().strip().splitlines()[-1])
print('$kv', d['ms_per_step'], d['parity_ok'], ' '.join('%s=%.3f' % (k, v['ms']) for k, v in d['stages'].items()))"
done
