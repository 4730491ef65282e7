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


def record_width(col: int, max_pts: int, path_cap: int | None = None, fields: str = "full") -> int:
    """float64 slots per scene.  ``fields`` = "full": status, traj_len, path_len, dp rows, path (s, l), trajectory (x, y,
    theta, kappa); "trajectory": status, traj_len, trajectory - all the reference's controller consumes
    (controller/controller.py:66-71).  ``path_cap`` (see ``path_capacity``) trims the path and trajectory arrays to the
    entries a cycle can fill."""
    cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
    if fields == "trajectory":
        return 2 + 4 * (cap + 1)
    if fields != "full":
        raise ValueError("fields must be 'full' or 'trajectory'")
    return 3 + col + 2 * cap + 4 * (cap + 1)


def pack_records(res, col: int, max_pts: int, path_cap: int | None = None, planner=None, fields: str = "full"):
    """CycleResult (torch tensors or numpy arrays) -> one (B, record_width) float64 matrix.  With ``planner`` and device
    tensors the packing is one kernel of the library (``Planner.pack_records``) instead of a handful of torch ops."""
    import torch
    if planner is not None and (isinstance(res.status, np.ndarray) or (torch.is_tensor(res.status) and res.status.is_cuda)):
        return planner.pack_records(res, col, max_pts, path_cap, fields=fields)
    as_t = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
    st, tl = as_t(res.status), as_t(res.traj_len)
    B = st.shape[0]
    traj = as_t(res.traj).reshape(B, max_pts + 1, 4)[:, :cap + 1].reshape(B, 4 * (cap + 1))
    if fields == "trajectory":
        parts = [st.to(torch.float64).reshape(B, 1), tl.to(torch.float64).reshape(B, 1), traj]
    elif fields == "full":
        parts = [st.to(torch.float64).reshape(B, 1), tl.to(torch.float64).reshape(B, 1),
                 as_t(res.path_len).to(torch.float64).reshape(B, 1), as_t(res.dp_rows).reshape(B, col),
                 as_t(res.path_s).reshape(B, max_pts)[:, :cap], as_t(res.path_l).reshape(B, max_pts)[:, :cap], traj]
    else:
        raise ValueError("fields must be 'full' or 'trajectory'")
    return torch.cat(parts, dim=1).contiguous()


def unpack_records(rec, col: int, max_pts: int, path_cap: int | None = None, fields: str = "full"):
    """Inverse of pack_records: dict of arrays keyed like CycleResult (path / trajectory arrays ``path_cap`` long)."""
    cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
    B = rec.shape[0]
    out = {"status": rec[:, 0].round().long(), "traj_len": rec[:, 1].round().long()}
    if fields == "trajectory":
        out["traj"] = rec[:, 2:].reshape(B, cap + 1, 4)
        return out
    o = 3
    out["path_len"] = rec[:, 2].round().long()
    out["dp_rows"] = rec[:, o:o + col]
    o += col
    out["path_s"] = rec[:, o:o + cap]
    o += cap
    out["path_l"] = rec[:, o:o + cap]
    o += cap
    out["traj"] = rec[:, o:].reshape(B, cap + 1, 4)
    return out


def gather_records(local, total: int, group=None, dst: int | None = None, alone_too: bool = False):
    """Collect every rank's records in scene order.  ``local`` is this rank's (count, width) matrix for the block
    ``shard_range(total, rank, world)``.

    ``dst`` = None: all ranks receive the (total, width) matrix (``all_gather``; equal shards use one
    ``all_gather_into_tensor``, ragged shards are padded to the largest block).  ``dst`` = r: a gather - only rank r
    receives the matrix, the others send their block and get None (what BASELINE's "RCCL gather" asks for: on N GPUs
    rank r takes in (N - 1) blocks per step and nobody else takes in anything).  ``alone_too``: issue the collective even in a
    process group of ONE rank (the identity, but through RCCL: what a one-GPU box can exercise of the N > 1 path)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not alone_too):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [shard_range(total, r, world)[1] for r in range(world)]
    assert local.shape[0] == counts[rank], "local block does not match shard_range"
    width = local.shape[1]
    biggest = max(counts)
    even = min(counts) == biggest
    if dst is None:
        if even:
            out = torch.empty((total, width), dtype=local.dtype, device=local.device)
            dist.all_gather_into_tensor(out, local.contiguous(), group=group)
            return out
        padded = torch.zeros((biggest, width), dtype=local.dtype, device=local.device)
        padded[:counts[rank]] = local
        bufs = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(bufs, padded, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    send = local.contiguous()
    if not even:
        send = torch.zeros((biggest, width), dtype=local.dtype, device=local.device)
        send[:counts[rank]] = local
    if rank != dst:
        dist.gather(send, None, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        return None
    if even:                                             # the blocks land side by side in the result: no copy afterwards
        out = torch.empty((total, width), dtype=local.dtype, device=local.device)
        dist.gather(send, list(out.split(biggest, dim=0)), dst=dist.get_global_rank(group, dst) if group is not None else dst,
                    group=group)
        return out
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


class StepGather:
    """The per-step result exchange of the many-scene mode (``bench.py --gpus N``; one instance per rank).

    ``submit(res)`` packs a cycle's outputs into records on the stream on which they become complete, gathers them on
    a stream of its own - so that the exchange of step k overlaps both stages of step k + 1 - and keeps the tensors of
    the last ``depth`` steps referenced until their gather has certainly finished (the caching allocator would otherwise
    hand their memory to the next step while RCCL still reads it).  With CPU tensors (``gloo`` in tests) there are no
    streams and ``planner`` may be any object with ``pack_records`` or None (torch packing)."""

    def __init__(self, col: int, max_pts: int, total: int, planner=None, fields: str = "full", dst: int | None = 0,
                 group=None, device=None, depth: int = 3, timing: bool = False, alone_too: bool = False):
        self.col, self.max_pts, self.total = int(col), int(max_pts), int(total)
        self.cap = path_capacity(max_pts)
        self.planner, self.fields, self.dst, self.group, self.depth = planner, fields, dst, group, int(depth)
        self.width = record_width(col, max_pts, self.cap, fields)
        self.in_flight = []
        self.stream = None
        self.alone_too = bool(alone_too)  # the collective even in a one-rank process group (gather_records)
        self.timing = bool(timing)       # event pairs around every gather on its stream (``gather_ms``)
        self.timed = []
        if device is not None and getattr(device, "type", "cpu") == "cuda":
            import torch
            self.stream = torch.cuda.Stream(device=device)

    def submit(self, res):
        import torch
        if self.stream is None:                          # CPU tensors: everything is synchronous
            rec = pack_records(res, self.col, self.max_pts, path_cap=self.cap, planner=self.planner, fields=self.fields)
            out = gather_records(rec, self.total, group=self.group, dst=self.dst, alone_too=self.alone_too)
            self.in_flight.append((rec, out, None))
            if len(self.in_flight) > self.depth:
                self.in_flight.pop(0)
            return out
        rs = self.planner.torch_result_stream()
        with torch.cuda.stream(rs):
            rec = pack_records(res, self.col, self.max_pts, path_cap=self.cap, planner=self.planner, fields=self.fields)
        self.stream.wait_stream(rs)
        with torch.cuda.stream(self.stream):
            began = None
            if self.timing:
                began = torch.cuda.Event(enable_timing=True)
                began.record(self.stream)
            out = gather_records(rec, self.total, group=self.group, dst=self.dst, alone_too=self.alone_too)
            done = torch.cuda.Event(enable_timing=self.timing)
            done.record(self.stream)
            if self.timing:
                self.timed.append((began, done))
        self.in_flight.append((rec, out, done))
        if len(self.in_flight) > self.depth:
            self.in_flight.pop(0)[2].synchronize()       # `depth` steps old: long done, costs nothing
        return out

    def unpack(self, out):
        return unpack_records(out, self.col, self.max_pts, path_cap=self.cap, fields=self.fields)

    def gather_ms(self, reset: bool = True):
        """(mean, min, max, count) of the gathers' own durations on their stream in ms, from the event pairs recorded since
        the last reset (``timing=True``; call after ``drain``).  None without timing or without a stream."""
        if not self.timed:
            return None
        ms = [a.elapsed_time(b) for a, b in self.timed]
        if reset:
            self.timed = []
        return sum(ms) / len(ms), min(ms), max(ms), len(ms)

    def drain(self):
        for _, _, done in self.in_flight:
            if done is not None:
                done.synchronize()
        self.in_flight.clear()

    def bytes_per_rank_and_step(self, scenes_per_rank: int) -> int:
        return int(scenes_per_rank) * self.width * 8
