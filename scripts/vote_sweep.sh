#!/bin/bash
# A/B of the Hough vote kernel configurations (bin width x LDS budget) on the bench workload.
for mode in 1 0; do for kb in 40 53 64 78; do
  r=$(IPA_VOTE_MODE=$mode IPA_VOTE_LDS_KB=$kb timeout -k 10 300 python3 bench.py --batch 256 --steps 3 --warmup 1 --no-cpu --check 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stages']['hough_vote']['ms'], d['parity_check'])")
  echo "mode=$mode lds_kb=$kb vote_ms=$r"
done; done
