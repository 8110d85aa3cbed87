#!/bin/bash
# round 5 (GPU box): parity on EVERY instance at BASELINE.json's full sizes (config 4: 65 536, config 5: 16 384), both 4-vertex
# kernels on the shard size, and the headline over eight seeds
out=$PWD/gpurun_out/r05v; mkdir -p "$out"
(python tools/full_parity.py randomized 65536 && python tools/full_parity.py long_horizon 16384) 2>&1 | grep -v amdgpu.ids > "$out/full_parity_full_size.txt"; tail -12 "$out/full_parity_full_size.txt"
for seed in 1 2 3 4 5 6 7 8; do
  python bench.py --seed $seed --steps 4 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed $seed', round(d['value']), round(d['outcome']['all_instances_per_s']), round(d['ms_per_step'],1), d['iterations']['max'], round(d['outcome']['converged'],4))"
done > "$out/seed_band.txt" 2>&1; cat "$out/seed_band.txt"
