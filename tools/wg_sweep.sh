#!/bin/bash
# usage (GPU box): bash tools/wg_sweep.sh [lib.so] [from] [to] -- instances/s against resident workgroups per CU (CMPC_WG_PER_CU),
# B = 20480 so that the queue's rounding does not blur the slope.  The knob exists in the developer build only
# (build.py: build_hip_dev -> tools/libcmpc_amd_dev.so, -DCMPC_DEV_KNOBS); the shipped library reads no environment.
lib=${1:-tools/libcmpc_amd_dev.so}; a=${2:-1}; b=${3:-8}
for n in $(seq $a $b); do
  CMPC_WG_PER_CU=$n CMPC_LIB_PATH=${lib:+$PWD/$lib} python3 bench.py --no-cpu-baseline --no-extras --batch 20480 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-shipped} wg_per_cu $n', round(d['outcome']['all_instances_per_s']), round(d['roofline']['kernel_ms'],1))"
done
