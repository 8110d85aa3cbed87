#!/bin/bash
# round-5 GPU session B: A/B against the round-4 kernel, residency sweep, profile + stall counters of the new kernel
out=gpurun_out/r05d; mkdir -p $out
NEW="online-non-linear-centroidal-mpc-with-stability-guarantees-for-robust-locomotion-of-legged-robots-_amd/libcmpc_amd.so"
AB_STEPS=6 timeout -k 10 300 bash tools/ab_bench.sh tools/ab/libA_r04abi.so $NEW 3 > $out/ab_8192.txt 2>&1; cat $out/ab_8192.txt
AB_STEPS=3 AB_ARGS="--batch 65536" timeout -k 10 300 bash tools/ab_bench.sh tools/ab/libA_r04abi.so $NEW 1 > $out/ab_65536.txt 2>&1; cat $out/ab_65536.txt
timeout -k 10 300 bash tools/wg_sweep.sh tools/libcmpc_amd_dev.so 4 7 > $out/wg_sweep.txt 2>&1; cat $out/wg_sweep.txt
timeout -k 10 500 bash tools/profile_round.sh r05d > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log
timeout -k 10 400 bash tools/pmc_stall.sh $PWD/$out/pmc_stall > $out/pmc_stall.log 2>&1; tail -40 $out/pmc_stall.log
