#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r01d
# Collects what profiles/ holds per round: bench line, rocprofv3 kernel stats of the bench command,
# and the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, counters only with --kernel-trace).
set -e
tag=${1:-rXX}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err"
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/stats.log" 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$out/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$out/write.log" 2>&1
echo "write done"
python3 - "$out" <<'EOF'
import csv, glob, json, sys
out = sys.argv[1]
def counter(d, name):
    tot = 0.0
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cmpc_solve" in r["Kernel_Name"] and r["Counter_Name"] == name:
                tot += float(r["Counter_Value"])
    return tot
fk, wk = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
json.dump({"fetch_size_kb": fk, "write_size_kb": wk, "hbm_bytes_per_launch": (fk + wk) * 1024.0}, open(f"{out}/traffic_raw.json", "w"))
for f in glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cmpc_solve" in r["Name"]:
            print("kernel stats:", r)
print("traffic:", fk, wk)
EOF
