#!/bin/bash
# usage: scripts/ab_full.sh "ENV=a" "ENV=b" ...   -- the default bench (1024 x 4K) once per environment string
for envs in "$@"; do
  r=$(env $envs timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu --check 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms'] for k,v in d['stages'].items()}, d['parity_check'][:8])")
  echo "[$envs] $r"
done
