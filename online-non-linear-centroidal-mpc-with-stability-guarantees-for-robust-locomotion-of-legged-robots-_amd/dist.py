"""Multi-GPU layer: the batch is dealt to the ranks by predicted cost (or cut into contiguous shards), every rank
solves its shard with no communication, and ONE collective gathers the results (RCCL all-gather over xGMI on GPUs; the
same code runs over gloo on CPU tensors in the tests).  The reference has no counterpart (single
process, one instance per tick): SURVEY.md 8e.

Why a deal.  A step ends when the slowest rank does, and a rank's launch ends when its longest instance does: with
contiguous shards the step time is the WORST of `world` shard makespans, and a shard's makespan moves by +-10 % with the
instances it happens to hold.  `shard_order` sorts the batch by the launch's own cost predictor
(queue_order: the same eighteen record features the kernel queues by) and deals it round-robin, so every rank gets the
same share of the expected stragglers ("assign shards round-robin by expected difficulty").

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


def shard_order(records, spec, world_size):
    """Deal of a batch over `world_size` ranks by predicted cost: a permutation `order` of range(B) such that rank r
    solves the instances order[r::world_size].  Stable sort by decreasing bucket of the predicted iteration count
    (cmpc_amd.queue_order, the predictor of the launch's own queue), dealt round-robin; every rank computes the same
    permutation from the same records.  Shard sizes are those of `shard_bounds`."""
    import numpy as np
    from . import queue_order as qo
    rec = records.detach().cpu().numpy() if isinstance(records, torch.Tensor) else np.asarray(records)
    if rec.shape[0] == 0:
        return np.zeros(0, dtype=np.int64)
    if world_size <= 1:
        return np.arange(rec.shape[0], dtype=np.int64)
    key = qo.bucket_of(qo.predicted_iterations(rec, spec))
    return np.argsort(-key, kind="stable").astype(np.int64)


def dealt_rows(order, world_size, rank):
    """Input-order indices of the instances rank `rank` solves under the deal `order`."""
    return order[rank::world_size]


def gather_dealt(local, order, group=None):
    """All-gather the shards of a deal and scatter the rows back to INPUT order: row i of the result belongs to
    instance i on every rank.  `local`: this rank's (len(order[rank::world]), cols) block, rows in the order of
    `dealt_rows`.  One collective, as `gather_shards`."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    B, cols = int(order.shape[0]), local.shape[1]
    longest = (B + world - 1) // world
    padded = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0], cols), dtype=local.dtype, device=local.device)
        padded = torch.cat((local, pad), dim=0)
    padded = padded.contiguous()
    if padded.is_cuda and dist.get_backend(group) == "gloo":      # rehearsal path (ranks sharing one GPU): through the host
        host = torch.empty((world * longest, cols), dtype=local.dtype)
        dist.all_gather_into_tensor(host, padded.cpu(), group=group)
        full = host.to(local.device)
    else:
        full = torch.empty((world * longest, cols), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, padded, group=group)
    # row r * longest + j of `full` is instance order[r + j * world]
    pos = torch.arange(B, dtype=torch.int64)
    src = (pos % world) * longest + pos // world                  # where the instance at deal position `pos` sits in `full`
    out = torch.empty((B, cols), dtype=local.dtype, device=local.device)
    out[torch.as_tensor(order, dtype=torch.int64).to(local.device)] = full[src.to(local.device)]
    return out


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


def solve_sharded(solve_fn, records, N, nu, group=None, gather="feedback", deal_spec=None):
    """Solve this rank's shard of `records` (the FULL (B, nrec) batch, identical on every rank) and
    gather.  `solve_fn(shard) -> (XU, status, iters, kkt)`.  Returns (gathered, status_all, iters_all), rows in input
    order.  deal_spec: a ProblemSpec -> the batch is dealt by predicted cost (`shard_order`); None -> contiguous shards.
    """
    B = records.shape[0]
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    if deal_spec is not None and world > 1:
        order = shard_order(records, deal_spec, world)
        rows = torch.as_tensor(dealt_rows(order, world, rank), dtype=torch.int64).to(records.device)
        shard = records[rows].contiguous()
    else:
        order = None
        lo, hi = shard_bounds(B, world, rank)
        shard = records[lo:hi].contiguous()
    XU, status, iters, _ = solve_fn(shard)
    payload = first_stage_feedback(XU, N, nu) if gather == "feedback" else XU
    # status / iteration counts ride in the same collective as two extra fp64 columns
    packed = torch.cat((payload, status.to(payload.dtype)[:, None], iters.to(payload.dtype)[:, None]), dim=1)
    full = gather_shards(packed, B, group) if order is None else gather_dealt(packed, order, group)
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
