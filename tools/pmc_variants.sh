#!/bin/bash
# usage: pmc_variants.sh  (run on the GPU box) -- per-variant instruction counts
export TMPDIR=/tmp
for v in BASE NO_GTPG NO_FACTOR NO_VEC NO_STEP; do
  rm -rf /tmp/pmcv
  CMPC_VARIANT=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d /tmp/pmcv -- python3 tools/run_variant.py > /tmp/pmcv.log 2>&1
  python3 - <<PYEOF
import csv,glob
for f in glob.glob("/tmp/pmcv/*/*counter_collection.csv"):
    agg={}
    for r in csv.DictReader(open(f)):
        if "cmpc_solve" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    its=float(open("/tmp/pmcv.log").read().split("iterations")[-1].split()[0])
    print("$v", "sweeps", its, {k: round(v/its/21) for k,v in agg.items()}, "(per stage-sweep)")
PYEOF
done
