#!/bin/bash
# A/B of environment settings over batch sizes (full pass, overlapped default schedule).
# usage: scripts/ab_batch.sh "A=1,B=2 -" "1 16 64 256"
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for n in $2; do
  for kv in $1; do
    args=$(echo "$kv" | tr ',' ' '); [ "$kv" = "-" ] && args="IPA_NONE=0"
    env $args timeout -k 10 200 python3 bench.py --batch $n --steps 20 --warmup 5 --no-cpu --check 1 2> gpurun_out/ab_batch.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $n', '$kv', d['ms_per_step'], d['parity_ok'], 'vote=%.4f compact=%.4f' % (d['stages']['hough_vote']['ms'], d['stages']['hough_compact']['ms']))" || { echo "$kv $n failed"; tail -3 gpurun_out/ab_batch.err; exit 1; }
  done
done
