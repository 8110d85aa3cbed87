"""Multi-rank path on CPU: world_size-2 gloo, contiguous shards, one gather.  The per-shard solver is
the C oracle here (test stand-in for the GPU kernel; the sharding / gather code is the product's)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmpc_amd import dist as cdist


def test_shard_bounds_cover_batch_exactly():
    for B in (0, 1, 7, 64, 65, 1000):
        for world in (1, 2, 3, 8):
            spans = [cdist.shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, B, N, tmp, deal=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import cmpc_amd  # noqa: F401
    from cmpc_amd import workloads as wl
    from oracle import oracle_lib as ol
    spec, rec = wl.make_workload("perturbed", B=B, N=N, scale=0.5)
    cs = ol.default_spec(N=spec.N, nv=spec.nv, tol=spec.tol, max_iter=spec.max_iter)

    def solve_fn(shard):
        out, st, it, kkt = ol.solve_batch(cs, shard.numpy(), nthreads=1)
        return torch.from_numpy(out), torch.from_numpy(st), torch.from_numpy(it), torch.from_numpy(kkt)

    fb, st, it = cdist.solve_sharded(solve_fn, torch.from_numpy(rec), spec.N, spec.nu, deal_spec=spec if deal else None)
    np.save(os.path.join(tmp, f"fb{rank}.npy"), fb.numpy())
    np.save(os.path.join(tmp, f"st{rank}.npy"), st.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 8])
def test_two_rank_gloo_shard_and_gather(tmp_path, oracle, B):
    N = 4
    port = _free_port()
    mp.spawn(_worker, args=(2, port, B, N, str(tmp_path)), nprocs=2, join=True)
    from cmpc_amd import workloads as wl
    spec, rec = wl.make_workload("perturbed", B=B, N=N, scale=0.5)
    cs = oracle.default_spec(N=N, nv=4, tol=spec.tol, max_iter=spec.max_iter)
    full, st, _, _ = oracle.solve_batch(cs, rec)
    want = np.concatenate([full[:, 20:40], full[:, 20 * (N + 1):20 * (N + 1) + spec.nu]], axis=1)
    for r in range(2):
        got = np.load(tmp_path / f"fb{r}.npy")
        assert got.shape == (B, 20 + spec.nu)
        assert np.array_equal(got, want)                     # every rank holds the full, ordered result
        assert np.array_equal(np.load(tmp_path / f"st{r}.npy"), st)


def test_shard_order_deals_every_instance_once_and_by_predicted_cost():
    from cmpc_amd import workloads as wl, queue_order as qo
    spec, rec = wl.make_workload("randomized", B=1001, N=20)
    for world in (1, 2, 3, 8):
        order = cdist.shard_order(rec, spec, world)
        assert sorted(order.tolist()) == list(range(1001))
        sizes = [len(cdist.dealt_rows(order, world, r)) for r in range(world)]
        assert sizes == [hi - lo for lo, hi in (cdist.shard_bounds(1001, world, r) for r in range(world))]
        if world > 1:
            key = qo.bucket_of(qo.predicted_iterations(rec, spec))
            assert (np.diff(key[order]) <= 0).all()                     # by decreasing predicted cost, ...
            pred = qo.predicted_iterations(rec, spec)
            loads = [pred[cdist.dealt_rows(order, world, r)].sum() for r in range(world)]
            cont = [pred[lo:hi].sum() for lo, hi in (cdist.shard_bounds(1001, world, r) for r in range(world))]
            assert max(loads) - min(loads) <= max(cont) - min(cont) + 1e-9   # ... which balances the predicted load
    assert cdist.shard_order(rec[:0], spec, 4).shape == (0,)


@pytest.mark.parametrize("world,B", [(2, 7), (3, 10), (3, 3)])
def test_dealt_shards_gather_back_to_input_order(tmp_path, oracle, world, B):
    """The deal by predicted cost over world-size 2 and 3 with ragged batches: the gathered tensor is bit for
    bit what the contiguous path returns, on every rank."""
    N = 4
    for deal in (False, True):
        d = tmp_path / ("deal" if deal else "cont")
        d.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), B, N, str(d), deal), nprocs=world, join=True)
    for r in range(world):
        for name in ("fb", "st"):
            a, b = np.load(tmp_path / "cont" / f"{name}{r}.npy"), np.load(tmp_path / "deal" / f"{name}{r}.npy")
            assert np.array_equal(a, b), (name, r)
    assert np.array_equal(np.load(tmp_path / "deal" / "fb0.npy"), np.load(tmp_path / "deal" / f"fb{world - 1}.npy"))


def test_gather_is_identity_without_process_group():
    x = torch.arange(12.0).reshape(4, 3)
    assert cdist.gather_shards(x, 4) is x
