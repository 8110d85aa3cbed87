#!/bin/bash
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh r02a
# Collects what profiles/ holds per round:
#   bench line, rocprofv3 kernel stats of the bench command (strictly serial launches),
#   FETCH_SIZE / WRITE_SIZE PMC passes of one bench launch (separate runs, counters only with --kernel-trace),
#   the same two passes over tools/ubench/hbm_calib (known byte count at the solver's 8 B/lane access width:
#   MI355X_MICROARCH.md says counters at widths other than 16 B/lane must be calibrated before use).
set -e
tag=${1:-rXX}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err"
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$out/stats.log" 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > "$out/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > "$out/write.log" 2>&1
echo "write done"
calib=tools/ubench/hbm_calib
if [ -x $calib ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/calib_fetch" -- $calib 4 > "$out/calib_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/calib_write" -- $calib 4 > "$out/calib_write.log" 2>&1
  echo "calibration done"
fi
python3 - "$out" <<'EOF2'
import csv, glob, json, sys
out = sys.argv[1]
def counter(d, name, kernel):
    tot = 0.0
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                tot += float(r["Counter_Value"])
    return tot
fk, wk = counter("fetch", "FETCH_SIZE", "cmpc_solve"), counter("write", "WRITE_SIZE", "cmpc_solve")
res = {"fetch_size_kb": fk, "write_size_kb": wk}
cf, cw = counter("calib_fetch", "FETCH_SIZE", "calib_read8"), counter("calib_write", "WRITE_SIZE", "calib_write8")
nbytes = 0
for line in open(f"{out}/calib_fetch.log") if glob.glob(f"{out}/calib_fetch.log") else []:
    if line.startswith('{"bytes"'):
        nbytes = json.loads(line)["bytes"]
if cf > 0 and cw > 0 and nbytes:
    res.update(calib_bytes=nbytes, calib_fetch_size_kb=cf, calib_write_size_kb=cw,
               fetch_factor=nbytes / (cf * 1024.0), write_factor=nbytes / (cw * 1024.0))
    res["hbm_bytes_per_launch"] = fk * 1024.0 * res["fetch_factor"] + wk * 1024.0 * res["write_factor"]
else:
    res["hbm_bytes_per_launch"] = (fk + wk) * 1024.0
json.dump(res, open(f"{out}/traffic_raw.json", "w"), indent=1)
for f in glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "cmpc_solve" in r["Name"]:
            print("kernel stats:", r)
print("traffic:", res)
EOF2
