"""Multi-GPU sharding of a plan batch: one process per GPU, contiguous shards, one all-gather.

Planning problems are independent (the reference already runs 32 at once as separate processes,
QTOS/generateHeightField.py:344-352), so the only exchange is re-assembling the plan batch:
ONE ``all_gather`` of the solution nodes (n_vars doubles per plan) with the status word packed as an
extra column.  With ``torch.distributed`` backend "nccl" this is one RCCL all-gather over xGMI; "gloo"
is used by the CPU tests.  The 1 kHz CSV rows (1.48 MB per plan) are sampled after the gather, never shipped.
"""
import numpy as np


def shard_bounds(n_items, world_size, rank):
    """Contiguous [begin, end) of rank's shard; the first (n_items % world_size) ranks get one more."""
    base, rem = divmod(n_items, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_buffers(n_total, n_vars, world, dtype, device):
    """Send / receive buffers of gather_plans, to be reused by a caller that gathers batch after batch (bench.py):
    no allocation and no fill kernels between the solve and the collective."""
    import torch
    per = -(-n_total // world)
    packed = torch.zeros((per, n_vars + 1), dtype=dtype, device=device)
    packed[:, n_vars] = -1.0
    return packed, torch.empty((world * per, n_vars + 1), dtype=dtype, device=device)


def gather_plans(nodes_local, status_local, n_total, group=None, work=None):
    """All-gather the per-rank shards (torch tensors, same device) into full (n_total, n_vars) /
    (n_total,) tensors on every rank with ONE collective: the status word of a plan travels as an
    extra column of its node row (small integers are exact in float64), shards are padded to equal
    length (pad rows carry status -1).  work = gather_buffers(...): reused buffers -- the returned tensors are then
    VIEWS of them and the next call overwrites them (clone what must outlive it)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    n_vars = nodes_local.shape[1]
    n_loc = nodes_local.shape[0]
    if n_loc > per:
        raise ValueError("gather_plans: local shard of %d plans exceeds ceil(%d / %d) = %d" % (n_loc, n_total, world, per))
    packed, gathered = work if work is not None else gather_buffers(n_total, n_vars, world, nodes_local.dtype, nodes_local.device)
    if work is not None:
        # reused buffers: they must fit THIS batch (a handle reused with another batch size would ship stale rows)
        if (tuple(packed.shape) != (per, n_vars + 1) or tuple(gathered.shape) != (world * per, n_vars + 1)
                or packed.dtype != nodes_local.dtype or gathered.dtype != nodes_local.dtype
                or packed.device != nodes_local.device or gathered.device != nodes_local.device):
            raise ValueError("gather_plans: work buffers do not match the batch (want %s / %s %s on %s)" %
                             ((per, n_vars + 1), (world * per, n_vars + 1), nodes_local.dtype, nodes_local.device))
        if n_loc < per:
            packed[n_loc:, n_vars] = -1.0   # pad rows of a ragged shard: never a stale status 0
    packed[:n_loc, :n_vars] = nodes_local
    packed[:n_loc, n_vars] = status_local.to(nodes_local.dtype)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    if world * per != n_total:   # ragged batch: drop the pad rows
        keep = []
        for r in range(world):
            b, e = shard_bounds(n_total, world, r)
            keep.append(torch.arange(r * per, r * per + (e - b), device=nodes_local.device))
        gathered = gathered[torch.cat(keep)]
    return gathered[:, :n_vars], gathered[:, n_vars].round().to(status_local.dtype)


def plan_sharded(solve_fn, start, goal, group=None, device="cpu"):
    """solve_fn(start_shard, goal_shard) -> (nodes, status) numpy; returns the full batch on every
    rank.  start/goal: full-batch numpy arrays, identical on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b, e = shard_bounds(len(start), world, rank)
    nodes, status = solve_fn(start[b:e], goal[b:e])
    tn = torch.as_tensor(np.ascontiguousarray(nodes), dtype=torch.float64, device=device)
    ts = torch.as_tensor(np.ascontiguousarray(status).astype(np.int32), device=device)
    all_nodes, all_status = gather_plans(tn, ts, len(start), group)
    return all_nodes.cpu().numpy(), all_status.cpu().numpy()
