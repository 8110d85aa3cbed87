#!/bin/bash
# usage (GPU box, repo root): bash tools/round_report.sh r03b   -- everything profiles/ holds for a round, in one call
tag=${1:-rXX}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
python -m pytest tests -m gpu -q > "$out/gpu_tests.log" 2>&1; echo "gpu tests rc $?"; tail -2 "$out/gpu_tests.log"
bash tools/profile_round.sh $tag > "$out/profile_round.log" 2>&1; tail -3 "$out/profile_round.log"
bash tools/pmc_stall.sh "$out/pmc_stall" > "$out/pmc_stall.log" 2>&1; echo "pmc done"
python bench.py --workload perturbed --batch 256 --steps 6 --warmup 1 --no-extras --no-cpu-baseline > "$out/bench_line_perturbed.json" 2>/dev/null
python bench.py --workload payload --batch 4096 --steps 6 --warmup 1 --no-extras --no-cpu-baseline > "$out/bench_line_payload.json" 2>/dev/null
python bench.py --workload long_horizon --steps 4 --warmup 1 --no-extras --no-cpu-baseline > "$out/bench_line_long_horizon.json" 2>/dev/null
echo "config lines done"
# the whole-body QP kernel: kernel statistics and HBM counters of three B = 65536 launches / one launch
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/wbc_stats" -- python3 tools/wbc_profile.py > "$out/wbc_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/wbc_fetch" -- python3 tools/wbc_profile.py 1 > "$out/wbc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/wbc_write" -- python3 tools/wbc_profile.py 1 > "$out/wbc_write.log" 2>&1
echo "wbc profile done"
python tools/tail_study.py randomized 8192 > "$out/tail_randomized.txt" 2>&1
python tools/tail_study.py long_horizon 2048 > "$out/tail_long_horizon.txt" 2>&1
python bench.py --seed 777 --steps 6 --warmup 1 --no-extras --no-cpu-baseline > "$out/bench_line_seed777.json" 2>/dev/null
python bench.py --kernel single --workload perturbed --batch 256 --steps 6 --warmup 1 --no-extras --no-cpu-baseline > "$out/bench_line_perturbed_single_wave.json" 2>/dev/null
(python tools/small_batch_latency.py randomized 1 16 256 && KERNEL=single python tools/small_batch_latency.py randomized 1 16 256) 2>&1 | grep -v amdgpu.ids > "$out/small_batch_latency.txt"
python tools/walk_demo.py > "$out/walk_demo.txt" 2>&1; tail -2 "$out/walk_demo.txt"
python tools/parity_report.py > "$out/parity_report.txt" 2>&1; echo "parity report done"
for w in "randomized 8192" "payload 4096" "perturbed 4096"; do python tools/full_parity.py $w; done > "$out/full_parity.txt" 2>&1; tail -4 "$out/full_parity.txt"
CMPC_BENCH_REHEARSAL=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-extras --no-cpu-baseline > "$out/selflaunch_2rank.json" 2> "$out/selflaunch_2rank.err"; echo "self-launch rc $?"
# round 5: phase profile, residency sweep (developer build: CMPC_WG_PER_CU), kernel crossover, the deal over eight ranks replayed
python tools/phase_profile.py randomized 8192 2>&1 | grep -v amdgpu.ids > "$out/phase_randomized.txt"
bash tools/wg_sweep.sh tools/libcmpc_amd_dev.so 1 7 > "$out/wg_sweep.txt" 2>&1
python tools/kernel_crossover.py randomized 2048 3072 4096 5120 8192 2>&1 | grep -v amdgpu.ids > "$out/kernel_crossover.txt"
python tools/kernel_crossover.py payload 2048 3072 4096 5120 2>&1 | grep -v amdgpu.ids >> "$out/kernel_crossover.txt"
python tools/deal_replay.py randomized 8 2>&1 | grep -v amdgpu.ids > "$out/deal_replay.txt"
python tools/deal_replay.py --sorted randomized 8 2>&1 | grep -v amdgpu.ids >> "$out/deal_replay.txt"
python tools/deal_replay.py long_horizon 8 2>&1 | grep -v amdgpu.ids >> "$out/deal_replay.txt"
echo "round-5 extras done"
