#!/bin/bash
set -e
run() { echo "== $*"; env "$@" timeout -k 10 120 python -u scripts/overlap_probe.py --vote-only --reps 3 --pairs hough+thr6,hough+gray3,hough+open6,hough+canny3 2>/dev/null | grep -E "alone|pair"; }
run IPA_PAD_VOTE=12 IPA_VOTE_RESIDENT=256
