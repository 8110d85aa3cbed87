"""bench.py started bare with --gpus N launches its own ranks (the driver's command shape).  The CPU tier
rehearses the launcher itself -- child processes, rendezvous on 127.0.0.1 over gloo, one gather through the
product's dist layer, exit code propagation -- with CMPC_BENCH_DRYRUN=1 (no solver, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, **env):
    e = dict(os.environ, CMPC_BENCH_DRYRUN="1", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                          env=e, capture_output=True, text=True, timeout=300)


def test_bare_command_spawns_its_ranks_and_rank0_prints_one_line():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["gathered_in_order"]


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
