#!/bin/bash
# round-5 GPU session A: first run of the GPU suite on the new kernel + A/B against the round-4 kernel (same ABI)
out=gpurun_out/r05a; mkdir -p $out
NEW="online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd/libcmpc_amd.so"
timeout -k 10 600 python -m pytest tests -m gpu -q > $out/gpu_tests.log 2>&1; echo "gpu tests rc $?"; tail -15 $out/gpu_tests.log
AB_STEPS=6 timeout -k 10 300 bash tools/ab_bench.sh tools/ab/libA_r04abi.so $NEW 2 > $out/ab_8192.txt 2>&1; cat $out/ab_8192.txt
AB_STEPS=3 AB_ARGS="--batch 65536" timeout -k 10 300 bash tools/ab_bench.sh tools/ab/libA_r04abi.so $NEW 1 > $out/ab_65536.txt 2>&1; cat $out/ab_65536.txt
AB_STEPS=6 AB_ARGS="--workload payload --batch 4096" timeout -k 10 200 bash tools/ab_bench.sh tools/ab/libA_r04abi.so $NEW 1 > $out/ab_payload.txt 2>&1; cat $out/ab_payload.txt
AB_STEPS=6 AB_ARGS="--workload perturbed --batch 256" timeout -k 10 200 bash tools/ab_bench.sh tools/ab/libA_r04abi.so $NEW 1 > $out/ab_perturbed.txt 2>&1; cat $out/ab_perturbed.txt
