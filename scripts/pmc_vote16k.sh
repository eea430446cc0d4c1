#!/bin/bash
# LDS / issue counters of the vote kernel on the 16k x 16k configuration (two rocprofv3 --pmc passes)
set -u
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc16
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc16/$name -- python3 bench.py --size 16k --steps 1 --warmup 1 --no-cpu --check 0 > gpurun_out/pmc16/$name.log 2>&1 || exit 1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES
python3 - <<PY
import csv, glob, collections
for name in ("sq1", "sq2"):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc16/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vote" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(name, {c: round(sum(v) / len(v)) for c, v in vals.items()})
PY
