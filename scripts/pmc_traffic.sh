#!/bin/bash
# HBM traffic counters for the default bench workload (two separate rocprofv3 --pmc passes).
# usage: scripts/pmc_traffic.sh <outdir> <bench args...>
set -u
out=$1; shift
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for c in FETCH_SIZE WRITE_SIZE; do
  name=$(echo $c | tr 'A-Z' 'a-z' | sed 's/_size//')
  timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d "$out/$name" -- python3 bench.py "$@" > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
done
