#!/bin/bash
# A/B of groups of environment settings on the headline batch (overlapped schedule).
# usage: scripts/ab_env2.sh "A=1,B=2 A=3 -" : one bench run per word, commas separate the variables of a run, "-" = none
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
i=0
for kv in $1; do
  i=$((i+1))
  args=$(echo "$kv" | tr ',' ' '); [ "$kv" = "-" ] && args="IPA_NONE=0"
  env $args timeout -k 10 200 python3 bench.py --steps 6 --warmup 2 --no-cpu --check 2 > gpurun_out/ab_env_$i.json 2> gpurun_out/ab_env_$i.err || { echo "$kv failed"; tail -3 gpurun_out/ab_env_$i.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/ab_env_$i.json').read().strip().splitlines()[-1]); print('$kv', d['ms_per_step'], d['parity_ok'], 'hyst=%.3f' % d['stages']['canny_hysteresis']['ms'])"
done
