"""Many-scene mode across the GPUs of one node: contiguous sharding + one gather of the results.

Scenes are independent (the reference is stateless per planning call, test_9.py:92-221), so each
rank plans its own contiguous block with the single-GPU pipeline and the only exchange is the
collection of fixed-stride result records - ``all_gather`` over RCCL/xGMI (backend "nccl" on
ROCm).  No other collective exists on this path.  Everything here is backend-agnostic
(``gloo`` on CPU in tests).
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of scene indices of ``rank``: (start, count); earlier ranks take the remainder."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def path_capacity(max_pts: int, decimate: int = 2, midpoint: bool = True) -> int:
    """Most stations a cycle's path can have: the DP path holds at most ``max_pts`` points, every ``decimate``-th goes
    to the QP, the midpoint re-interleave adds one (test_9.py:187, :204-210); the trajectory has one point more."""
    return (int(max_pts) + decimate - 1) // decimate + (1 if midpoint else 0)


def record_width(col: int, max_pts: int, path_cap: int | None = None) -> int:
    """float64 slots per scene: status, traj_len, path_len, dp rows, path (s, l), trajectory (x, y, theta, kappa).
    ``path_cap`` (see ``path_capacity``) trims the path and trajectory arrays to the entries a cycle can fill."""
    cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
    return 3 + col + 2 * cap + 4 * (cap + 1)


def pack_records(res, col: int, max_pts: int, path_cap: int | None = None, planner=None):
    """CycleResult (torch tensors or numpy arrays) -> one (B, record_width) float64 matrix.  With ``planner`` and device
    tensors the packing is one kernel of the library (``Planner.pack_records``) instead of a handful of torch ops."""
    import torch
    if planner is not None and (isinstance(res.status, np.ndarray) or (torch.is_tensor(res.status) and res.status.is_cuda)):
        return planner.pack_records(res, col, max_pts, path_cap)
    as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
    st, tl, pl_ = as_t(res.status), as_t(res.traj_len), as_t(res.path_len)
    B = st.shape[0]
    parts = [st.to(torch.float64).reshape(B, 1), tl.to(torch.float64).reshape(B, 1),
             pl_.to(torch.float64).reshape(B, 1), as_t(res.dp_rows).reshape(B, col),
             as_t(res.path_s).reshape(B, max_pts)[:, :cap], as_t(res.path_l).reshape(B, max_pts)[:, :cap],
             as_t(res.traj).reshape(B, max_pts + 1, 4)[:, :cap + 1].reshape(B, 4 * (cap + 1))]
    return torch.cat(parts, dim=1).contiguous()


def unpack_records(rec, col: int, max_pts: int, path_cap: int | None = None):
    """Inverse of pack_records: dict of arrays keyed like CycleResult (path / trajectory arrays ``path_cap`` long)."""
    cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
    B = rec.shape[0]
    o = 3
    out = {"status": rec[:, 0].round().long(), "traj_len": rec[:, 1].round().long(),
           "path_len": rec[:, 2].round().long(), "dp_rows": rec[:, o:o + col]}
    o += col
    out["path_s"] = rec[:, o:o + cap]
    o += cap
    out["path_l"] = rec[:, o:o + cap]
    o += cap
    out["traj"] = rec[:, o:].reshape(B, cap + 1, 4)
    return out


def gather_records(local, total: int, group=None):
    """All ranks receive the (total, width) matrix of every rank's records, in scene order.

    ``local`` is this rank's (count, width) matrix for the block ``shard_range(total, rank, world)``.
    Equal shards use one ``all_gather_into_tensor``; ragged shards are padded to the largest block.
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [shard_range(total, r, world)[1] for r in range(world)]
    assert local.shape[0] == counts[rank], "local block does not match shard_range"
    width = local.shape[1]
    biggest = max(counts)
    if min(counts) == biggest:
        out = torch.empty((total, width), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((biggest, width), dtype=local.dtype, device=local.device)
    padded[:counts[rank]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
