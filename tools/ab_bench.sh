#!/bin/bash
# usage (GPU box): bash tools/ab_bench.sh libA.so libB.so [rounds]   -- alternating serial bench runs of two builds
# of the HIP library in one session (run-to-run and box-to-box noise is ~3 %, so A/B in one call)
a=$1; b=$2; n=${3:-3}
for i in $(seq 1 $n); do
  for lib in $a $b; do
    CMPC_LIB_PATH=$PWD/$lib python3 bench.py --no-cpu-baseline --no-extras --steps ${AB_STEPS:-6} ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['outcome']['all_instances_per_s']), round(d['ms_per_step'],1), round(d['roofline']['kernel_ms'],1))"
  done
done
