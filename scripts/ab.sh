#!/bin/bash
# usage: scripts/ab.sh "ENV1=a ENV2=b" "ENV1=c" ...   -- runs the bench (batch 256) once per environment string
for envs in "$@"; do
  r=$(env $envs timeout -k 10 300 python3 bench.py --batch 256 --steps 3 --warmup 1 --no-cpu --check 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, d['parity_check'][:8])")
  echo "[$envs] $r"
done
