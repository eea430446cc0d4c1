#!/bin/bash
# Counter passes and kernel durations of config 4 (one 16384 x 16384 noisy page), serial schedule.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export IPA_PIPELINE=0 IPA_OVERLAP=0
bash scripts/pmc_passes.sh gpurun_out/pmc_c4 --size 16k --steps 2 --warmup 1 --no-cpu --check 0 || exit 1
python3 scripts/pmc_summary.py gpurun_out/pmc_c4 > gpurun_out/pmc_c4_summary.txt || exit 1
python3 scripts/traffic_from_pmc.py gpurun_out/pmc_c4 16kx1 gpurun_out/traffic_c4.json > gpurun_out/traffic_c4.txt || exit 1
rm -rf gpurun_out/pmc_c4
cat gpurun_out/traffic_c4.txt
rm -rf gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4 -- python3 bench.py --size 16k --steps 10 --warmup 2 --no-cpu --check 0 > gpurun_out/prof_c4.log 2>&1 || { tail -20 gpurun_out/prof_c4.log; exit 1; }
cp $(find gpurun_out/prof_c4 -name '*kernel_stats.csv' | head -1) gpurun_out/c4_kernel_stats.csv
rm -rf gpurun_out/prof_c4
head -12 gpurun_out/c4_kernel_stats.csv | cut -c1-120
