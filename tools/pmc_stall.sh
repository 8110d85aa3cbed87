#!/bin/bash
# usage (GPU box): bash tools/pmc_stall.sh [outdir]  -- stall / latency counters of the bench launch,
# one rocprofv3 --pmc pass per counter group (counters only with --kernel-trace).
export TMPDIR=/tmp
out=${1:-$PWD/gpurun_out/pmc_stall}
mkdir -p "$out"
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_IFETCH_LEVEL"
 "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64"
)
i=0
for g in "${groups[@]}"; do
  rm -rf "$out/g$i"
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/g$i" -- python3 bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-extras > "$out/g$i.log" 2>&1 || echo "group $i failed"
  echo "group $i done"
  i=$((i+1))
done
python3 - "$out" <<'PYEOF'
import csv, glob, sys
out = sys.argv[1]
agg = {}
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cmpc_solve" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
with open(f"{out}/summary.txt", "w") as fh:
    for k in sorted(agg):
        line = f"{k:32s} {agg[k]:.6e}"
        print(line); fh.write(line + "\n")
PYEOF
