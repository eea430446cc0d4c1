#!/bin/bash
# A/B of the Hough vote kernel configurations on the bench workload.
for sw in 0 1; do for kb in 40 78; do
  r=$(IPA_VOTE_SEGWALK=$sw IPA_VOTE_LDS_KB=$kb timeout -k 10 300 python3 bench.py --batch 256 --steps 3 --warmup 1 --no-cpu --check 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['stages']['hough_vote']['ms'], d['stages']['canny_nms']['ms'], d['parity_check'])")
  echo "segwalk=$sw lds_kb=$kb vote_ms/nms_ms=$r"
done; done
