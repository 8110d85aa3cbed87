"""Multi-GPU layer: the batch shards contiguously over ranks, every rank solves its shard with no
communication, and ONE collective gathers the results (RCCL all-gather over xGMI on GPUs; the same
code runs over gloo on CPU tensors in the tests).  The reference has no counterpart (single
process, one instance per tick): SURVEY.md 8e.

The gathered payload is small (first-stage feedback x_1, u_0: 20 + nu doubles per instance, or the
full trajectory on request), so a single direct all-gather is used; there is no reduction in this
workload and therefore no ring all-reduce.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world_size, rank):
    """Contiguous shard [lo, hi) of a batch of B over world_size ranks (first ranks take the remainder)."""
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def first_stage_feedback(XU, N, nu):
    """(x_1, u_0) of every instance: what the whole-body controller consumes each tick
    (reference: self.x, self.u at code/centroidal_mpc_vertices.py:614-616)."""
    x1 = XU[:, 20:40]
    u0 = XU[:, 20 * (N + 1):20 * (N + 1) + nu]
    return torch.cat((x1, u0), dim=1).contiguous()


def gather_shards(local, B, group=None):
    """All-gather row shards of unequal length into the full (B, cols) tensor on every rank.

    `local` is this rank's (hi-lo, cols) block under ``shard_bounds``.  One collective: shards are
    padded to the longest shard so a single ``all_gather_into_tensor`` suffices.
    """
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    cols = local.shape[1]
    longest = (B + world - 1) // world
    padded = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0], cols), dtype=local.dtype, device=local.device)
        padded = torch.cat((local, pad), dim=0)
    padded = padded.contiguous()
    if padded.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal path (several ranks sharing one GPU, gloo): stage through the host
        host = torch.empty((world * longest, cols), dtype=local.dtype)
        dist.all_gather_into_tensor(host, padded.cpu(), group=group)
        full = host.to(local.device)
    else:
        full = torch.empty((world * longest, cols), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, padded, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        parts.append(full[r * longest:r * longest + (hi - lo)])
    return torch.cat(parts, dim=0)


def solve_sharded(solve_fn, records, N, nu, group=None, gather="feedback"):
    """Solve this rank's shard of `records` (the FULL (B, nrec) batch, identical on every rank) and
    gather.  `solve_fn(shard) -> (XU, status, iters, kkt)`.  Returns (gathered, status_all, iters_all).
    """
    B = records.shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(B, world, rank)
    XU, status, iters, _ = solve_fn(records[lo:hi].contiguous())
    payload = first_stage_feedback(XU, N, nu) if gather == "feedback" else XU
    # status / iteration counts ride in the same collective as two extra fp64 columns
    packed = torch.cat((payload, status.to(payload.dtype)[:, None], iters.to(payload.dtype)[:, None]), dim=1)
    full = gather_shards(packed, B, group)
    return full[:, :-2], full[:, -2].to(torch.int32), full[:, -1].to(torch.int32)


def gather_rank_stats(values, group=None, device=None):
    """(world, len(values)) fp64 on the host: a few scalars of every rank (step time, gather time) on every rank, so
    that the one JSON line of a multi-GPU run shows skew between ranks next to the cost of the collective."""
    v = torch.tensor([float(x) for x in values], dtype=torch.float64)
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return v[None, :]
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "gloo":
        v = v.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    out = torch.empty((world, v.numel()), dtype=torch.float64, device=v.device)
    dist.all_gather_into_tensor(out, v[None, :].contiguous(), group=group)
    return out.cpu()
