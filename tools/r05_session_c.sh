#!/bin/bash
# round 5, session c8 (GPU box): final form of the forward sweep with its loads a stage ahead
out=$PWD/gpurun_out/r05r; mkdir -p "$out"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > "$out/gpu_tests.log" 2>&1; rc=$?; echo "gpu tests rc $rc"; tail -3 "$out/gpu_tests.log"
[ $rc -eq 0 ] || exit 1
bash tools/ab_bench.sh tools/ab/libI_base.so tools/ab/libM_ahead.so 3 > "$out/ab_forward_loads_ahead.txt" 2>&1 && cat "$out/ab_forward_loads_ahead.txt"
AB_ARGS="--batch 65536" AB_STEPS=3 bash tools/ab_bench.sh tools/ab/libI_base.so tools/ab/libM_ahead.so 1 > "$out/ab_forward_loads_ahead_65536.txt" 2>&1 && cat "$out/ab_forward_loads_ahead_65536.txt"
python tools/phase_profile.py randomized 8192 2>&1 | grep -v amdgpu.ids > "$out/phase_randomized.txt"; head -12 "$out/phase_randomized.txt"
CMPC_WG_PER_CU=1 CMPC_LIB_PATH=$PWD/tools/libcmpc_amd_dev.so python3 bench.py --no-cpu-baseline --no-extras --steps 3 --batch 2048 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one workgroup per CU, B=2048:', round(d['outcome']['all_instances_per_s']), round(d['ms_per_step'],1))"
