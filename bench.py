#!/usr/bin/env python3
"""Benchmark of the hot path: batched centroidal-MPC solves per second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the batched solver over one batch of synthetic parameter records that are
already resident in HBM.  Workload: the per-GPU shard of BASELINE config 4 (65536 domain-randomised
instances over 8 GPUs = 8192 per GPU, horizon N = 20, 4 vertices per foot, cold start), so that
--gpus 8 is exactly config 4 (weak scaling).  Every rank solves its shard with no communication and
one RCCL all-gather returns the first-stage feedback (x_1, u_0) + status of every instance.

Consecutive steps are independent batches, so they are enqueued on --streams (default 2) alternating
HIP streams, each with its own solver handle, scratch and output buffers: a launch is a queue of
instances of very different length (20 iterations typical, max_iter = 100 for the ~1 % that never
settle), its last stragglers keep a few workgroups busy long after the queue is empty, and the next
step's workgroups fill the idle CUs in the meantime.  --streams 1 gives the strictly serial number.

The JSON line also carries
  roofline      HBM classification of SURVEY.md 8d: algorithmic bytes B_io = 11 728 B per cold N=20
                solve x solves per launch / kernel time (HIP events on the launch stream) vs 8 TB/s
  cpu_baseline  the C oracle (a port, not CasADi/IPOPT) on the host cores, bounded sample, rank 0, N=1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PER_GPU_BATCH = 8192
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VECTOR_PEAK_TFLOPS = 78.6


def algorithmic_bytes(N, nu, warm):
    """B_io of SURVEY.md 8d: record (+mass, mu) in, solution out (+ warm start in), status/iters/kkt."""
    return 8 * ((19 * N + 22 + 2) + (nu * N + 20 * (N + 1)) * (1 + int(warm))) + 16


def measured_traffic(workload, batch, N):
    """HBM bytes per launch from the committed PMC runs (profiles/*traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate rocprofv3 --pmc passes of this same command), or None."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("batch") == batch and d.get("N") == N:
            best = (d["hbm_bytes_per_launch"], os.path.basename(path))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (solver handles) the steps alternate over")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="instances per GPU")
    ap.add_argument("--workload", default="randomized", choices=["perturbed", "payload", "randomized"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP solver has no CPU fallback")
    # CMPC_BENCH_REHEARSAL=1: several ranks share the visible GPU(s) over gloo (to rehearse the N > 1
    # code path on a one-GPU box); the real runs use one GPU per rank over RCCL.
    rehearsal = os.environ.get("CMPC_BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import cmpc_amd  # noqa: F401
    from cmpc_amd import workloads as wl, dist as cdist
    from cmpc_amd.solver import BatchedCentroidalMPC

    B_total = args.batch * world
    spec, rec_all = wl.make_workload(args.workload, B=B_total)
    lo, hi = cdist.shard_bounds(B_total, world, rank)
    rec = torch.from_numpy(rec_all[lo:hi].copy()).to(device)       # resident in HBM before timing
    S = max(1, args.streams)
    solvers = [BatchedCentroidalMPC(spec, device=device) for _ in range(S)]
    streams = [torch.cuda.Stream(device=device) for _ in range(S)]
    outs = [torch.empty((hi - lo, spec.nsol), dtype=torch.float64, device=device) for _ in range(S)]

    def step(i):
        j = i % S
        with torch.cuda.stream(streams[j]):
            XU, status, iters, kkt = solvers[j].solve(rec, out=outs[j])
            fb = cdist.first_stage_feedback(XU, spec.N, spec.nu)
            packed = torch.cat((fb, status.to(fb.dtype)[:, None], iters.to(fb.dtype)[:, None]), dim=1)
            full = cdist.gather_shards(packed, B_total)            # the ONE collective (no-op at N=1)
        return full

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for i in range(args.warmup):
        step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        full = step(i)
    sync()
    elapsed = time.perf_counter() - t0
    # HIP events around each handle's last launch (on its own launch stream), read after the timed
    # region so that the query does not serialise the streams
    kernel_ms = [solvers[j].last_kernel_ms() for j in range(min(S, args.steps))]
    last_ms = float(np.mean(kernel_ms))                           # average launch duration, timed region
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())

    st = full[:, -2].to(torch.int32)
    it = full[:, -1]
    conv_frac = float((st == 0).double().mean().item())
    mean_iters = float(it.mean().item())
    value = B_total * args.steps / elapsed
    b_io = algorithmic_bytes(spec.N, spec.nu, warm=False)
    achieved = b_io * (hi - lo) / (last_ms * 1e-3) / 1e9             # GB/s of algorithmic bytes, this rank
    flops = 3.0e6 * mean_iters * (hi - lo) / (last_ms * 1e-3) / 1e12  # ~3 Mflop per Newton/Riccati step

    traffic = measured_traffic(args.workload, args.batch, spec.N)
    result = {
        "metric": "centroidal-MPC solves/sec, N=20 horizon",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE config 4 shard: {args.workload}, batch {args.batch}/GPU "
                               f"({B_total} total), N={spec.N}, 2 feet x {spec.nv} vertices, cold start",
                   "global_batch": B_total, "horizon": spec.N, "tol": spec.tol, "max_iter": spec.max_iter,
                   "converged_fraction": conv_frac, "mean_iterations": mean_iters,
                   "parallelism": f"batch-sharded x{world}, final all-gather of (x1,u0,status); "
                                  f"steps alternate over {S} HIP stream(s)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic[0] if traffic else None,
                     "traffic_source": traffic[1] if traffic else None,
                     "kernel": "cmpc_solve_kernel<4>", "kernel_ms": last_ms, "concurrent_launches": S,
                     "algorithmic_bytes_per_solve": b_io,
                     "note": "latency/FP64-issue bound in practice (SURVEY 8d): see fp64_tflops",
                     "fp64_tflops_model": flops, "fp64_frac_of_vector_peak": flops / FP64_VECTOR_PEAK_TFLOPS},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_lib as ol
        threads = min(16, os.cpu_count() or 1)
        nsample = 64 * threads
        sample = rec_all[:nsample]
        cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter, k1=spec.k1, k2=spec.k2,
                             prox=spec.prox)
        ol.solve_batch(cs, sample[:threads], nthreads=threads)      # warm the library
        t0 = time.perf_counter()
        _, st_c, it_c, _ = ol.solve_batch(cs, sample, nthreads=threads)
        dt = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": nsample / dt, "unit": "solves/s", "cores": threads, "kind": "port",
                                  "sample": f"first {nsample} instances of the same workload, C oracle "
                                            f"(same algorithm, -O3, OpenMP over the batch), {dt:.1f} s; "
                                            f"CasADi/IPOPT cannot run here (SURVEY 8c)"}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
