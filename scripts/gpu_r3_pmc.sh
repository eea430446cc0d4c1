#!/bin/bash
# Counter passes of the serial schedule (each stage once per step over the whole batch).
# usage: scripts/gpu_r3_pmc.sh <tag> [bench args...]   (environment such as IPA_CANNY_WIDE is passed through)
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export IPA_PIPELINE=0 IPA_OVERLAP=0
bash scripts/pmc_passes.sh gpurun_out/pmc_$tag --steps 2 --warmup 1 --no-cpu --check 0 "$@" || exit 1
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_${tag}_summary.txt || exit 1
python3 scripts/traffic_from_pmc.py gpurun_out/pmc_$tag 4kx1024 gpurun_out/traffic_$tag.json > gpurun_out/traffic_$tag.txt || exit 1
rm -rf gpurun_out/pmc_$tag
tail -20 gpurun_out/traffic_$tag.txt
