#!/bin/bash
# Collects rocprofv3 PMC counters for the bench (one pass per run, as the hardware needs).
# usage: scripts/pmc_passes.sh <outdir> <bench args...>
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py $BENCH_ARGS > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
BENCH_ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_WAVES || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
