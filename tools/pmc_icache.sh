#!/bin/bash
# usage (GPU box): bash tools/pmc_icache.sh -- instruction-cache and scalar-cache counters of one bench launch
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_icache
rm -rf "$out"; mkdir -p "$out"
i=0
for g in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$out/g$i" -- python3 bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-extras > "$out/g$i.log" 2>&1 || echo "group $i failed"
  echo "group $i done"; i=$((i+1))
done
python3 - "$out" <<'PYEOF'
import csv, glob, sys
agg = {}
for f in glob.glob(f"{sys.argv[1]}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cmpc_solve" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k in sorted(agg): print(f"{k:32s} {agg[k]:.6e}")
PYEOF
