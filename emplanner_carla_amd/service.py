"""The planning process body on the GPU: requests in the reference's wire format in, replies out.

The reference runs ``motion_planning(conn)`` (test_9.py:92-220) in a child process: it receives the tuple the driver
sends (test_9.py:390-392)

    (static_obs [(x, y, dis)], dynamic_obs [(x, y, dis, speed)], vehicle_loc, pred_loc, vehicle_v, vehicle_a,
     global_frenet_path [(x, y, theta, kappa)], match_point_list)

and answers with (trajectory [(x, y, theta, kappa)], match_point_list, path_s, path_l) (test_9.py:220).
``plan_requests`` does that for a BATCH of requests with two device calls - ``emp_reference_line`` (match on the
global path, 51-node window, smoothing) and ``emp_plan_cycle`` (projection, virtual obstacles of the first dynamic
obstacle, S-L DP, path QP, Cartesian tail) - and ``motion_planning(conn)`` is the one-request loop with the
reference's Pipe protocol.  Host logic here is only what the reference does with Python lists before the numerics:
static obstacles count only if the nearest is within 30 m (test_9.py:117), only the first dynamic obstacle is used
(:141-142).
"""
from __future__ import annotations

import logging

import numpy as np

from .api import Planner, dp_params, max_path_points, qp_params, smooth_params

log = logging.getLogger("emplanner_carla_amd.service")

#: how many requests in a row may be answered with the previous trajectory before the loop gives up (ValueError): at the
#: reference's planning period (test_9.py:356: one request per 0.5 s of simulated time at most) ten repeats are five seconds
#: of driving on a trajectory that was planned for an older scene
MAX_CONSECUTIVE_REPEATS = 10

INFEASIBLE_BANNER = "********************     can't find a feasible path      ********************"


def _path_array(path):
    """[(x, y, theta, kappa), ...] -> (n, 4) float64.  One NumPy conversion where the nodes are plain 4-sequences of numbers (what the
    reference's driver sends: 40 us for 240 nodes instead of 250 for the per-element form, which stays as the fall-back for anything
    else - longer tuples, objects with __float__)."""
    try:
        a = np.asarray(path, dtype=np.float64)
        if a.ndim == 2 and a.shape[1] == 4:
            return a
    except (TypeError, ValueError):
        pass
    return np.asarray([[float(p[0]), float(p[1]), float(p[2]), float(p[3])] for p in path], dtype=np.float64)


def pack_requests(requests):
    """The request tuples of a batch as the padded arrays the two device calls take (host logic of test_9.py:
    95-131, 141-142 only: which obstacles count, which dynamic obstacle is used)."""
    B = len(requests)
    G = max(len(r[6]) for r in requests)
    K = max(1, max((len(r[0]) for r in requests), default=1))
    a = dict(global_path=np.zeros((B, G, 4)), n_global=np.zeros(B, np.int32), pred=np.zeros((B, 2)),
             veh=np.zeros((B, 2)), v=np.zeros((B, 2)), a=np.zeros((B, 2)), pre_match=np.zeros(B, np.int32),
             obs_xy=np.zeros((B, K, 2)), n_obs=np.zeros(B, np.int32), dyn=np.full((B, 2), np.nan))
    for b, (static, dynamic, vehicle_loc, pred_loc, vehicle_v, vehicle_a, path, match_list) in enumerate(requests):
        a["n_global"][b] = len(path)
        a["global_path"][b, :len(path)] = _path_array(path)
        a["pred"][b], a["veh"][b], a["v"][b], a["a"][b] = pred_loc, vehicle_loc, vehicle_v, vehicle_a
        a["pre_match"][b] = int(match_list[0])
        if len(static) != 0 and static[0][-1] <= 30:                       # test_9.py:117
            a["n_obs"][b] = len(static)
            a["obs_xy"][b, :len(static)] = [(float(o[0]), float(o[1])) for o in static]
        if len(dynamic) != 0:                                               # test_9.py:141-142
            a["dyn"][b] = (float(dynamic[0][2]), float(dynamic[0][3]))
    return a


def plan_arrays(planner: Planner, a, dp=None, qp=None, sp=None, stages=None):
    """The two device calls on packed request arrays (the dict of ``pack_requests``; also what the wire server decodes
    its fixed-stride records into).  Returns (reference-line status (B,), match index (B,), CycleResult, max_pts)."""
    dp = dp or dp_params()
    qp = qp or qp_params()
    sp = sp or smooth_params()
    M = max_path_points(dp)
    if stages is None and len(a["n_global"]):
        # ONE device call (ABI 11): the front end runs inside emp_plan_cycle and its 51-point lines never leave the device
        res = planner.plan_cycle(dp, qp, sp, None, None, max_pts=M, origin_xy=a["veh"], start_xy=a["pred"], start_v=a["v"],
                                 start_a=a["a"], obs_xy=a["obs_xy"], n_obs=a["n_obs"], dyn_dis_speed=a["dyn"],
                                 global_path=a["global_path"], n_global=a["n_global"], pre_match_index=a["pre_match"])
        return res.ref_status, res.match_index, res, M
    # the two-call form, for callers that want the stages (the stage-by-stage parity tests): same kernels, same results
    ref, n_ref, match, _, st_ref = planner.reference_line(sp, a["global_path"], a["n_global"], a["pred"], a["pre_match"])
    n_ref_used = np.where(st_ref == 0, n_ref, 2).astype(np.int32)
    res = planner.plan_cycle(dp, qp, sp, max_pts=M, ref_line=ref, n_ref=n_ref_used, origin_xy=a["veh"], start_xy=a["pred"],
                             start_v=a["v"], start_a=a["a"], obs_xy=a["obs_xy"], n_obs=a["n_obs"], dyn_dis_speed=a["dyn"])
    if stages is not None:
        stages.update(inputs=a, ref_line=ref, n_ref=n_ref_used, match=match, ref_status=st_ref, cycle=res)
    return st_ref, match, res, M


class RequestPlanner:
    """One request at a time, as fast as the host side allows (the reference's call shape: test_9.py:92-96, 220, 390-395).  The
    request tuple is written straight into ONE page-locked block (``api.HostRing`` slot made for requests), ``emp_plan_cycle`` runs
    the front end and the cycle in one call - one PCIe copy in, six + one kernels, one copy out - and the reply tuple is built from
    the page-locked outputs.  Same kernels and results as ``plan_requests`` (0.54 -> 0.3 ms per request in-process, round 6)."""

    def __init__(self, planner: Planner, dp=None, qp=None, sp=None, max_static: int = 8):
        self.planner, self.dp, self.qp, self.sp = planner, dp or dp_params(), qp or qp_params(), sp or smooth_params()
        self.max_static = int(max_static)
        self.ring = None
        self.M = max_path_points(self.dp)

    def _slot(self, n_path, n_static):
        r = self.ring
        if r is None or r.max_global < n_path or r.max_obs < n_static:
            if r is not None:
                r.close()
            G = max(64, -(-int(n_path) // 64) * 64)
            self.max_static = max(self.max_static, int(n_static))
            r = self.ring = self.planner.host_ring(self.dp, 1, 51, self.max_static, self.M, depth=1, max_global=G)
        return r.slots[0]

    def plan(self, request):
        """(reply tuple or None, status, match index): None where the reference would have raised or a QP is infeasible, like
        ``plan_requests``."""
        static, dynamic, vehicle_loc, pred_loc, vehicle_v, vehicle_a, path, match_list = request
        use_static = len(static) != 0 and static[0][-1] <= 30                   # test_9.py:117
        slot = self._slot(len(path), len(static) if use_static else 0)
        i = slot.inputs
        n = len(path)
        i["global_path"][0, :n] = _path_array(path)
        i["n_global"][0] = n
        i["pre_match_index"][0] = int(match_list[0])
        i["origin_xy"][0], i["start_xy"][0], i["start_v"][0], i["start_a"][0] = vehicle_loc, pred_loc, vehicle_v, vehicle_a
        if use_static:
            i["n_obs"][0] = len(static)
            i["obs_xy"][0, :len(static)] = [(float(o[0]), float(o[1])) for o in static]
        else:
            i["n_obs"][0] = 0
        slot.use_dyn = len(dynamic) != 0                                          # test_9.py:141-142: the first one only
        if slot.use_dyn:
            i["dyn_dis_speed"][0] = (float(dynamic[0][2]), float(dynamic[0][3]))
        self.planner.plan_cycle(self.dp, self.qp, self.sp, None, None, None, None, None, None, None, None, slot=slot)
        slot.wait()
        o = slot.outputs
        st_ref, st = int(o["ref_status"][0]), int(o["status"][0])
        match = int(o["match_index"][0])
        status = st_ref | st
        if st_ref != 0 or (st & ~1) != 0:
            return None, status, match
        m, k = int(o["traj_len"][0]), int(o["path_len"][0])
        traj = [tuple(row) for row in o["traj"][0, :m].tolist()]
        return (traj, [match], o["path_s"][0, :k].tolist(), o["path_l"][0, :k].tolist()), status, match

    def close(self):
        if self.ring is not None:
            self.ring.close()
            self.ring = None


class CycleStream:
    """``plan_arrays`` for callers that have more than one batch in the air (the wire server's sessions, a driver that plans for
    several vehicles): the overlapped host path of the planning cycle (``api.HostRing``, EMP_HOST_PINNED).

    ``submit`` copies the batch - global paths included - into the next page-locked ring slot and queues front end and cycle, one
    call, on the staged pipeline - it returns while the inputs are still crossing PCIe; ``result`` waits for that batch's outputs and hands them
    back as ordinary arrays.  Calls from different threads interleave: while one thread waits in ``result``, another's
    ``submit`` already has the GPU working on the next batch (``submit`` itself is serialised by a lock: one context, one
    call at a time).  Results are bit for bit those of ``plan_arrays`` - same kernels, same order."""

    def __init__(self, planner: Planner, capacity: int = 256, max_static: int = 8):
        import threading
        self.planner, self.capacity, self.max_static = planner, int(capacity), int(max_static)
        self._lock = threading.Lock()            # one context: one call at a time
        self._free = threading.Condition()       # slots are owned from submit() until result() has copied their outputs out
        self._rings = {}
        planner.set_pipeline(1)                                             # staged: two batches in flight
        planner.set_fence(False)                                            # the front end reads nothing of the cycles in flight

    def _ring(self, dp, B, G, mo):
        cap = max(self.capacity, 1 << max(B - 1, 0).bit_length())
        G = max(64, -(-int(G) // 64) * 64)                                  # global-path capacity in steps of 64 nodes
        key = (int(dp.row), int(dp.col), float(dp.sample_s), float(dp.sampling_res), cap, G, mo)
        if key not in self._rings:
            ring = self.planner.host_ring(dp, cap, 51, mo, max_path_points(dp), max_global=G)     # slots for REQUESTS
            ring.free = list(ring.slots)
            self._rings[key] = ring
        return self._rings[key]

    def _take_slot(self, ring):
        """A slot nobody owns.  Sessions consume their results in any order, so the ring is NOT walked round-robin: a slot goes
        back to the free list only when its batch's outputs have been copied out (``result``)."""
        with self._free:
            while not ring.free:
                self._free.wait()
            return ring.free.pop(0)

    def submit(self, a, dp=None, qp=None, sp=None):
        dp = dp or dp_params()
        qp = qp or qp_params()
        sp = sp or smooth_params()
        B = len(a["n_global"])
        if B == 0:
            z = np.zeros(0, np.int32)
            return dict(B=0, st_ref=z, match=z, M=max_path_points(dp), slot=None, ring=None)
        with self._lock:
            ring = self._ring(dp, B, int(a["global_path"].shape[1]), max(int(a["obs_xy"].shape[1]), self.max_static))
        slot = self._take_slot(ring)                                        # may wait for another session's result() - outside the lock
        issued = False
        try:
            with self._lock:
                pl = self.planner
                G = int(a["global_path"].shape[1])
                slot.inputs["global_path"][:B, :G] = a["global_path"]       # (nodes past n_global are never read)
                slot.inputs["obs_xy"][:B] = 0.0
                slot.inputs["obs_xy"][:B, :a["obs_xy"].shape[1]] = a["obs_xy"]
                slot.load(n_global=a["n_global"], pre_match_index=a["pre_match"], origin_xy=a["veh"], start_xy=a["pred"],
                          start_v=a["v"], start_a=a["a"], n_obs=a["n_obs"], dyn_dis_speed=a["dyn"])
                cap, slot.B = slot.B, B
                try:
                    issued = True                    # from here on the slot may carry a ticket: _release waits for it
                    pl.plan_cycle(dp, qp, sp, None, None, None, None, None, None, None, None, slot=slot)
                finally:
                    slot.B = cap
                return dict(B=B, M=slot.max_pts, slot=slot, ring=ring)
        except BaseException:
            # a failing request (a refused argument, a HIP error, a shape that does not fit) must not shrink the ring: after
            # `depth` such requests every session would wait in _take_slot for ever (the advisor's round-5 finding)
            self._release(ring, slot, wait=issued)
            raise

    def _release(self, ring, slot, wait=True):
        """Hand a slot back to its ring.  ``wait``: a call may have been issued on it - its outputs must have landed (or the call
        must have failed) before another batch writes the slot's inputs."""
        try:
            if wait:
                slot.wait()
        except Exception:                        # the wait itself failed: the slot still goes back, the error is the caller's
            slot._ticket = None
        finally:
            with self._free:
                ring.free.append(slot)
                self._free.notify_all()

    def result(self, h):
        """(reference-line status, match index, CycleResult, max_pts) of a submitted batch, as ``plan_arrays`` returns them."""
        from .api import CycleResult
        if h["slot"] is None:
            z = np.zeros((0,))
            return h["st_ref"], h["match"], CycleResult(*([z] * 10)), h["M"]
        slot, h["slot"] = h["slot"], None        # a handle is consumed once, whatever happens below
        try:
            slot.wait()                          # emp_wait_ticket: safe beside another thread's submit - the lock is NOT held
            out = {k: np.array(v[:h["B"]]) for k, v in slot.outputs.items()}
        finally:
            self._release(h["ring"], slot, wait=False)     # only now may another batch take the slot
        return out["ref_status"], out["match_index"], CycleResult(**out), h["M"]

    def plan_arrays(self, a, dp=None, qp=None, sp=None):
        return self.result(self.submit(a, dp, qp, sp))

    def close(self):
        with self._lock:
            for r in self._rings.values():
                r.close()
            self._rings = {}
            self.planner.set_fence(True)
            self.planner.set_pipeline(0)


def plan_requests(planner: Planner, requests, dp=None, qp=None, sp=None, stages=None):
    """requests: list of request tuples.  Returns a list of (reply tuple or None, status): None where the reference
    would have raised (IndexError paths) or where a QP is infeasible.  ``stages``: a dict that receives the packed
    inputs and the raw outputs of the two device calls (for the stage-by-stage parity tests)."""
    B = len(requests)
    if B == 0:
        return []
    st_ref, match, res, _ = plan_arrays(planner, pack_requests(requests), dp, qp, sp, stages)
    out = []
    for b in range(B):
        status = int(st_ref[b]) | int(res.status[b])
        if int(st_ref[b]) != 0 or (int(res.status[b]) & ~1) != 0:
            out.append((None, status))
            continue
        m, k = int(res.traj_len[b]), int(res.path_len[b])
        traj = [tuple(float(x) for x in row) for row in res.traj[b, :m]]
        out.append(((traj, [int(match[b])], [float(x) for x in res.path_s[b, :k]], [float(x) for x in res.path_l[b, :k]]),
                    status))
    return out


def answer_refused(reply, status, match_index, previous, on_infeasible):
    """What the planning loop sends for a request that was refused (reply None).  IndexError cases raise, as the
    reference does (path_planning.py:63, :267).  An infeasible path / smoothing QP has no faithful answer - the reference
    ignores cvxopt's status (path_planning.py:211-218) and sends whatever its last iterate was - so the policy is the
    caller's:
      "previous" (default)  the last valid trajectory and path again, with the NEW match index: an unmodified reference
                            driver unpacks the reply and hands element 0 straight to its controller (test_9.py:395-399), so
                            it must be a trajectory; ValueError if there has been no valid reply yet;
      "raise"               ValueError (the planning process ends, like the reference's on an exception);
      "sentinel"            (None, match_point_list, [], []) for a driver patched to keep its previous path."""
    if reply is not None:
        return reply
    if status & (2 | 4):
        raise IndexError("list index out of range")
    if on_infeasible == "sentinel":
        return (None, [int(match_index)], [], [])
    if on_infeasible == "previous" and previous is not None:
        return (previous[0], [int(match_index)], previous[2], previous[3])     # the new match index is still valid (test_9.py:99)
    if on_infeasible not in ("previous", "raise"):
        raise ValueError(f"on_infeasible must be 'previous', 'raise' or 'sentinel', not {on_infeasible!r}")
    raise ValueError("path or smoothing QP infeasible" + ("" if on_infeasible == "raise" else " and no previous trajectory to repeat"))


class RefusalPolicy:
    """The state of ``answer_refused`` across the requests of one planning loop: the last valid reply, and how many
    requests in a row it has been repeated for.  Every repeat is logged (WARNING, logger emplanner_carla_amd.service) with
    its status bits and its count; after ``max_repeats`` consecutive repeats the next refused request raises ValueError -
    a driver is never left following an arbitrarily stale trajectory without anyone being told."""

    def __init__(self, on_infeasible: str = "previous", max_repeats: int = MAX_CONSECUTIVE_REPEATS):
        self.on_infeasible, self.max_repeats = on_infeasible, int(max_repeats)
        self.previous, self.repeats = None, 0

    def answer(self, reply, status, match_index):
        if reply is not None:
            self.previous, self.repeats = reply, 0
            return reply
        out = answer_refused(reply, status, match_index, self.previous, self.on_infeasible)     # raises for the other policies
        if self.on_infeasible == "previous":
            self.repeats += 1
            if self.repeats > self.max_repeats:
                raise ValueError(f"path or smoothing QP infeasible for {self.repeats} requests in a row (status {status}): "
                                 f"not repeating a trajectory that old (max_repeats = {self.max_repeats})")
            log.warning("request refused (status %d: path or smoothing QP infeasible): previous trajectory sent again, "
                        "repeat %d of at most %d", status, self.repeats, self.max_repeats)
        else:
            log.warning("request refused (status %d): sentinel reply sent", status)
        return out


def motion_planning(conn, device_id: int = 0, dp=None, on_infeasible: str = "previous", strict: bool = False,
                    max_repeats: int = MAX_CONSECUTIVE_REPEATS):
    """Drop-in for the reference's planning process (test_9.py:92-220): ``multiprocessing.Process(target=
    motion_planning, args=(conn,))``.  Blocks on ``conn.recv()`` like the reference and creates its device context here,
    in the child.  A request on which the reference raises IndexError raises IndexError here too - the child ends, as the
    reference's does.  A request whose path or smoothing QP is infeasible: see ``answer_refused`` (default: the previous
    valid trajectory again, so that an UNMODIFIED driver keeps running - logged every time and at most ``max_repeats``
    times in a row, ``RefusalPolicy``; ``strict=True`` is ``on_infeasible="raise"``).
    ``dp`` overrides the lattice (default: the reference's keyword defaults, path_planning.py:277-279)."""
    planner = Planner(device_id)                                            # created in the child process
    if strict:
        on_infeasible = "raise"
    policy = RefusalPolicy(on_infeasible, max_repeats)
    one = RequestPlanner(planner, dp=dp)
    while 1:
        request = conn.recv()
        reply, status, match = one.plan(request)
        if status & 1:
            print(INFEASIBLE_BANNER)                                        # path_planning.py:351
        conn.send(policy.answer(reply, status, match))
