#!/bin/bash
# Kernel-trace stats and two SQ counter passes of a python script.
# usage: scripts/pmc_script.sh <outdir> <script.py> [args...]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$@" > "$out/trace.log" 2>&1 || { tail -5 "$out/trace.log"; exit 1; }
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc $PMC --output-format csv -d "$out/$name" -- python3 "$@" > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT" run sq1 "$@" || exit 1
PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" run sq2 "$@" || exit 1
python3 scripts/pmc_summary.py "$out" > "$out/summary.txt"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
cut -c1-160 "$out/kernel_stats.csv" | head -12
cat "$out/summary.txt"
