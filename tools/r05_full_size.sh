#!/bin/bash
# round 5 (GPU box): parity on EVERY instance at BASELINE.json's full sizes (config 4: 65 536, config 5: 16 384), and the
# headline over eight seeds.  (No pipes around the long steps: held-back output reads as a hung run.)
out=$PWD/gpurun_out/r05v; mkdir -p "$out"
python tools/full_parity.py randomized 65536 > "$out/full_parity_65536.txt" 2>&1; echo "config 4 full size done"; tail -5 "$out/full_parity_65536.txt"
python tools/full_parity.py long_horizon 16384 > "$out/full_parity_long_16384.txt" 2>&1; echo "config 5 full size done"; tail -5 "$out/full_parity_long_16384.txt"
for seed in 1 2 3 4 5 6 7 8; do
  python bench.py --seed $seed --steps 4 --warmup 1 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('seed $seed', round(d['value']), round(d['outcome']['all_instances_per_s']), round(d['ms_per_step'],1), d['iterations']['max'], round(d['outcome']['converged'],4))" | tee -a "$out/seed_band.txt"
done
