"""bench.py started bare with --gpus N launches its own ranks (the driver's command shape).  The CPU tier
rehearses the launcher itself -- child processes, rendezvous on 127.0.0.1 over gloo, one gather through the
product's dist layer, exit code propagation -- with CMPC_BENCH_DRYRUN=1 (no solver, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=(), **env):
    e = dict(os.environ, CMPC_BENCH_DRYRUN="1", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                           *extra], env=e, capture_output=True, text=True, timeout=300)


def test_bare_command_spawns_its_ranks_and_rank0_prints_one_line():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["gathered_in_order"]
    assert d["rank_stats"] == [[0.0, 0.0], [1.0, 2.0]]          # per-rank scalars reach rank 0 in rank order


def test_failed_rank_makes_the_launcher_exit_nonzero():
    r = _run(2, CMPC_BENCH_DRYRUN_FAIL="1")
    assert r.returncode != 0


def test_launcher_process_never_imports_torch():
    # the parent must not touch the GPU: no torch import at module level or inside the launcher, and no exec
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def main():")[0]
    code = [l.strip() for l in head.splitlines()]
    assert not any(l.startswith(("import torch", "from torch")) for l in code)
    assert not any("os.exec" in l or "execv" in l for l in code)


def test_hung_rank_is_stopped_by_the_launch_timeout():
    # one rank never reaches the collective: the launcher ends its own children after --launch-timeout and says so
    import time
    t0 = time.time()
    r = _run(2, extra=("--launch-timeout", "20"), CMPC_BENCH_DRYRUN_HANG="1")
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert "launch-timeout" in r.stderr
    assert time.time() - t0 < 120


def test_cpu_baseline_counts_the_instances_it_solved(monkeypatch):
    """Rounds 2-3 divided 256 x threads = 65 536 'instances' by the time of the 8192 that exist.  The figure must come
    from the number of instances the oracle returned: a fake 256-thread host, a batch smaller than 256 x 256."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    import cmpc_amd  # noqa: F401
    from cmpc_amd import workloads as wl
    from oracle import oracle_lib as ol
    spec, rec = wl.make_workload("randomized", B=48, N=4)
    monkeypatch.setattr(bench, "effective_cores", lambda: (256, {"sched_getaffinity": 256, "cgroup_cpu_quota": None,
                                                                  "os_cpu_count": 256}))
    calls = []

    def counting(cs, recs, nthreads=0):
        calls.append((recs.shape[0], nthreads))
        return ol.solve_batch(cs, recs, nthreads=min(nthreads, 4))
    d = bench.cpu_baseline(spec, rec, budget_s=1e9, single_budget_s=0.0, solve_batch=counting)
    assert d["cores"] == 256 and d["instances_solved"] == 48              # clamped to what the batch holds
    big = [n for n, t in calls if t == 256]
    assert big[-1] == d["instances_solved"]
    assert abs(d["all_instances_per_s"] - d["instances_solved"] / d["seconds"]) < 1e-9 * d["all_instances_per_s"]
    assert f"first {d['instances_solved']} instances" in d["sample"]
    assert d["single_thread_instances_solved"] >= 48 and d["single_thread_solves_per_s"] > 0


def test_effective_cores_respects_a_cgroup_quota(monkeypatch, tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    import builtins
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("1600000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    cores, detail = bench.effective_cores()
    assert cores == 16 and detail["sched_getaffinity"] == 256 and detail["cgroup_cpu_quota"] == 16.0
