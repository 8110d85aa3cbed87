"""Developer script: copy what tools/round_report.sh <tag> left under gpurun_out/<tag>/ into profiles/<tag>_* (the
summaries the design documents cite; the raw rocprofv3 output stays in gpurun_out/, which is scratch).
usage: python tools/collect_profiles.py r03c"""
import csv, glob, json, os, shutil, sys, time
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern), recursive=True))
    return f[0] if f else None


def rows(path, keep):
    with open(path) as fh:
        r = list(csv.reader(fh))
    return [r[0]] + [x for x in r[1:] if keep(x, r[0])]


def write(name, table):
    with open(os.path.join(dst, f"{tag}_{name}"), "w", newline="") as fh:
        csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC).writerows(table)


for name in ("bench_line.json", "bench_line_perturbed.json", "bench_line_payload.json", "bench_line_long_horizon.json",
             "walk_demo.txt", "parity_report.txt", "full_parity.txt", "selflaunch_2rank.json", "tail_randomized.txt",
             "tail_long_horizon.txt", "bench_line_seed777.json", "bench_line_perturbed_single_wave.json", "small_batch_latency.txt",
             "phase_randomized.txt", "wg_sweep.txt", "kernel_crossover.txt", "deal_replay.txt", "gpu_tests.log"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
ks = one("stats/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
kt = one("stats/**/*kernel_trace.csv")
if kt:
    write("bench_kernel_trace_cmpc.csv", rows(kt, lambda x, h: "cmpc_solve" in x[h.index("Kernel_Name")]))
for d, out, kern in (("fetch", "pmc_fetch_size.csv", "cmpc_solve"), ("write", "pmc_write_size.csv", "cmpc_solve"),
                     ("calib_fetch", "pmc_calib_fetch_size.csv", "calib_"), ("calib_write", "pmc_calib_write_size.csv", "calib_")):
    cc = one(f"{d}/**/*counter_collection.csv")
    if cc:
        write(out, rows(cc, lambda x, h: kern in x[h.index("Kernel_Name")]))
tr = os.path.join(src, "traffic_raw.json")
if os.path.exists(tr):
    t = json.load(open(tr))
    line = json.loads(open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1])
    res = {"workload": "randomized", "batch": line["config"]["global_batch"], "N": line["config"]["horizon"],
           "kernel": line["roofline"]["kernel"], "fetch_size_kb": t["fetch_size_kb"], "write_size_kb": t["write_size_kb"]}
    if "fetch_factor" in t:
        res["calibration"] = {"bytes": t["calib_bytes"], "fetch_size_kb": t["calib_fetch_size_kb"], "write_size_kb": t["calib_write_size_kb"],
                              "fetch_factor": t["fetch_factor"], "write_factor": t["write_factor"],
                              "note": "tools/ubench/hbm_calib streams 4 GiB at 8 B per lane in the same session"}
        res["read_gb"] = t["fetch_size_kb"] * 1024 * t["fetch_factor"] / 1e9
        res["written_gb"] = t["write_size_kb"] * 1024 * t["write_factor"] / 1e9
    res["hbm_bytes_per_launch"] = t["hbm_bytes_per_launch"]
    res["algorithmic_bytes_per_launch"] = line["roofline"]["algorithmic_bytes_per_solve"] * line["config"]["global_batch"]
    res["command"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                      f"--steps 1 --warmup 0 --no-cpu-baseline --no-extras   (tools/profile_round.sh {tag})")
    res["collected"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
    json.dump(res, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
# whole-body QP kernel: statistics of three launches, HBM counters of one (16 B / lane accesses are not used there: the
# FETCH_SIZE correction of the 8 B / lane calibration above applies to its row loads as well)
wk = one("wbc_stats/**/*kernel_stats.csv")
if wk:
    write("wbc_kernel_stats.csv", rows(wk, lambda x, h: "wbc_qp" in x[h.index("Name")]))
    wb = {}
    for d, key in (("wbc_fetch", "FETCH_SIZE"), ("wbc_write", "WRITE_SIZE")):
        cc = one(f"{d}/**/*counter_collection.csv")
        if cc:
            wb[key + "_kb"] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(cc))
                                  if "wbc_qp" in r["Kernel_Name"] and r["Counter_Name"] == key)
    if wb:
        tr_ = os.path.join(src, "traffic_raw.json")
        ff = json.load(open(tr_)).get("fetch_factor", 2.0) if os.path.exists(tr_) else 2.0
        wb["fetch_factor_applied"] = ff
        wb["hbm_bytes_per_launch"] = wb.get("FETCH_SIZE_kb", 0.0) * 1024 * ff + wb.get("WRITE_SIZE_kb", 0.0) * 1024
        wb["batch"] = 65536
        wb["algorithmic_bytes_per_launch"] = 65536 * (8 * (2 * 900 + 2 * 30 + 360 + 30 + 30 + 12) + 8)
        json.dump(wb, open(os.path.join(dst, f"{tag}_wbc_traffic.json"), "w"), indent=1)
# stall summary: sum every counter of the solve kernel over the passes of tools/pmc_stall.sh
tot = {}
for cc in glob.glob(os.path.join(src, "pmc_stall", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(cc)):
        if "cmpc_solve" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
if tot:
    with open(os.path.join(dst, f"{tag}_pmc_stall_summary.txt"), "w") as fh:
        for k in sorted(tot):
            fh.write(f"{k:32s} {tot[k]:.6e}\n")
print(sorted(os.path.basename(f) for f in glob.glob(os.path.join(dst, f"{tag}_*"))))
