"""Batched Python API over the C-ABI: one ``Planner`` object = one ``emp_ctx`` (device + stream).

Every method takes either NumPy arrays (host memory: the library stages them) or torch CUDA
tensors (device memory: used in place, nothing is copied) - never a mix - and returns arrays of
the same kind.  Shapes follow include/emplanner.h: batch-major, padded, with length arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from ._lib import DpParams, QpParams, SmoothParams, SpeedDpParams, SpeedQpParams, MpcParams, EmpError


def dp_params(row=12, col=6, sample_s=15, sample_l=1.5, sampling_res=2, w_collision_cost=1e12,
              w_smooth_cost=(300, 1000, 5000), w_reference_cost=20) -> DpParams:
    """Keyword defaults of reference DP_algorithm (planner/path_planning.py:276-279)."""
    p = DpParams()
    p.row, p.col = int(row), int(col)
    p.sample_s, p.sample_l, p.sampling_res = float(sample_s), float(sample_l), float(sampling_res)
    p.w_collision = float(w_collision_cost)
    for i in range(3):
        p.w_smooth[i] = float(w_smooth_cost[i])
    p.w_ref = float(w_reference_cost)
    return p


def dp_params_from_cfg(cfg) -> DpParams:
    return dp_params(cfg.row, cfg.col, cfg.sample_s, cfg.sample_l, cfg.sampling_res, cfg.w_collision_cost,
                     cfg.w_smooth_cost, cfg.w_reference_cost)


def qp_params(**kw) -> QpParams:
    """Keyword defaults of reference Quadratic_planning (path_planning.py:78-81) + driver switches."""
    q = QpParams()
    L.load().emp_qp_params_default(C.byref(q))
    names = {"dp_sampling_res": "ds", "w_cost_l": "w_l", "w_cost_dl": "w_dl", "w_cost_ddl": "w_ddl",
             "w_cost_dddl": "w_dddl", "w_cost_centre": "w_centre", "w_cost_end_l": "w_end_l",
             "w_cost_end_dl": "w_end_dl", "w_cost_end_ddl": "w_end_ddl"}
    for k, v in kw.items():
        setattr(q, names.get(k, k), v)
    return q


def smooth_params(w_cost_smooth=0.4, w_cost_length=0.3, w_cost_ref=0.3, x_thre=0.2, y_thre=0.2) -> SmoothParams:
    """Keyword defaults of reference smooth_reference_line (planning_utils.py:262-264)."""
    s = SmoothParams()
    s.w_smooth, s.w_length, s.w_ref = float(w_cost_smooth), float(w_cost_length), float(w_cost_ref)
    s.x_thre, s.y_thre = float(x_thre), float(y_thre)
    return s


def mpc_params(vehicle_para=(1.015, 2.910 - 1.015, 1412, -148970, -82204, 1537), q_diag=(250.0, 1.0, 50.0, 1.0),
               f_diag=(1.0, 1.0, 1.0, 1.0), r=1.0) -> MpcParams:
    """vehicle_para is unpacked exactly as the reference does, ``(a, b, Cf, Cr, m, Iz) = vehicle_para``
    (controller.py:132); the default is the tuple the reference's drivers pass (test_9.py:316).  q_diag / f_diag / r
    are the Q, F, R of Lateral_MPC_controller._control (controller.py:321-328)."""
    p = MpcParams()
    p.a, p.b, p.Cf, p.Cr, p.m, p.Iz = (float(v) for v in vehicle_para)
    for i in range(4):
        p.q_diag[i], p.f_diag[i] = float(q_diag[i]), float(f_diag[i])
    p.r = float(r)
    return p


def lqr_params(vehicle_para=(1.015, 2.910 - 1.015, 1412, -148970, -82204, 1537), q_diag=(200.0, 1.0, 50.0, 1.0),
               r=1.0) -> MpcParams:
    """Parameters of Lateral_LQR_controller._control (controller.py:592-598); vehicle_para unpacked as in mpc_params."""
    return mpc_params(vehicle_para=vehicle_para, q_diag=q_diag, r=r)


@dataclass
class LqrResult:
    steer: object        # (B,) raw steering command -K e_rr + delta_f (not clipped, as in the reference)
    K: object            # (B, 4)
    e_rr: object         # (B, 4)
    k_r: object          # (B,)
    min_index: object    # (B,) int32
    pre_pro: object      # (B, 4)
    sweeps: object       # (B,) int32 Riccati sweeps performed
    status: object


@dataclass
class MpcResult:
    steer: object        # (B,) first control of the horizon (the reference's res['x'][0])
    u: object            # (B, 12)
    e_rr: object         # (B, 4) e_d, e_d_dot, e_fi, e_fi_dot
    k_r: object          # (B,)
    min_index: object    # (B,) int32 match index (next call's min_index)
    pre_pro: object      # (B, 4) predicted x, y and projected x, y
    H: object            # (B, 12, 12) or None
    f: object            # (B, 12) or None
    iters: object
    status: object


SPEED_DP_COLS, SPEED_QP_POINTS, SPEED_DENSE_POINTS = 16, 17, 401
STB_RANGE, STB_INDEX, STB_QP_FAILED, STB_NO_PROFILE = 2, 4, 8, 64     # EMP_STB_* status bits


def speed_qp_params(w_cost_s_dot2=10, w_cost_v_ref=50, w_cost_jerk=500, reference_speed=50) -> SpeedQpParams:
    """Keyword defaults of reference speed_QP (speed_planning_test.py:410-411)."""
    return SpeedQpParams(float(w_cost_s_dot2), float(w_cost_v_ref), float(w_cost_jerk), float(reference_speed))


def speed_dp_params(reference_speed=50, w_cost_ref_speed=4000, w_cost_accel=100, w_cost_obs=10000000) -> SpeedDpParams:
    """Keyword defaults of reference speed_DP (speed_planning_test.py:101-102)."""
    p = SpeedDpParams()
    p.reference_speed, p.w_cost_ref_speed = float(reference_speed), float(w_cost_ref_speed)
    p.w_cost_accel, p.w_cost_obs = float(w_cost_accel), float(w_cost_obs)
    return p


#: nodes of the local reference line (reference planning_utils.py:244-246: 10 back + 40 forward + the match)
REF_LINE_POINTS = 51

#: shape of the reference's S-T tables (speed_planning_test.py:114-122)
ST_ROWS, ST_COLS = 40, 16


@dataclass
class SpeedDpResult:
    """Outputs of ``Planner.speed_dp`` (arrays or torch tensors, batch-major)."""
    cost: object        # (B, 40, 16) dp_st_cost, or None
    s_dot: object       # (B, 40, 16) dp_st_s_dot, or None
    node: object        # (B, 40, 16) int32 dp_st_node, or None
    end_node: object    # (B, 2) int32 terminal (row, col)
    speed_s: object     # (B, 16) s of the chosen node per t column, NaN after the terminal column
    speed_t: object     # (B, 16)


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "device")


class _Args:
    """Collects array arguments of one call, checks they live in one memory space, makes outputs."""

    def __init__(self, *inputs, planner=None):
        self.torch = any(_is_torch(x) for x in inputs if x is not None)
        self.keep = []
        self.outs = []               # torch output tensors of this call
        self.cycle = False           # set by Planner.plan_cycle: the call that runs on a lane of its own when pipelined
        self.planner = planner
        self.same_stream = False
        if self.torch:
            import torch
            self.t = torch
            self.device = next(x.device for x in inputs if _is_torch(x))
            if self.device.type != "cuda":
                raise ValueError("torch tensors passed to the planner must live on the GPU")
            # inputs may still be in flight on torch's stream; our kernels run on the context's own stream:
            # order the two streams on the device (no host block)
            cur = torch.cuda.current_stream(self.device)
            self.same_stream = planner is not None and int(cur.cuda_stream) == int(planner.stream or 0)
            if planner is None:
                cur.synchronize()
            elif not self.same_stream:               # nothing to order when the caller already works on our stream
                planner.torch_stream().wait_stream(cur)
        self.where = L.EMP_DEVICE if self.torch else L.EMP_HOST

    def done(self):
        """After the library call: torch's current stream waits for the planner's stream, so that reading an output
        tensor from torch code (``.cpu()``, another kernel) sees the finished result.  In pipelined mode the results
        of a cycle are produced on the lane that ran it: a caller on any OTHER stream waits for it; a caller
        that works on the planner's own streams orders itself (``Planner.torch_result_stream``)."""
        if not (self.torch and self.planner is not None):
            return
        piped = self.planner.pipelined and self.cycle
        if piped:
            # (No Tensor.record_stream on the outputs: it would tie their memory to a stream that dies with the planner
            # while the caching allocator still wants to record events on it; Planner.plan_cycle keeps the outputs of
            # the calls in flight referenced instead.)
            if not self.same_stream:
                self.t.cuda.current_stream(self.device).wait_stream(self.planner.torch_result_stream())
        elif not self.same_stream:
            self.t.cuda.current_stream(self.device).wait_stream(self.planner.torch_stream())

    def inp(self, x, dtype, shape=None):
        if x is None:
            return None
        if self.torch:
            if not _is_torch(x):
                raise ValueError("mixing torch device tensors and host arrays in one call is not supported")
            td = {np.float64: self.t.float64, np.int32: self.t.int32}[dtype]
            if x.dtype != td or not x.is_contiguous():
                # the conversion kernel runs on torch's current stream AFTER the wait issued in __init__: order the
                # planner's stream behind it as well, or its kernels could read the copy before it is written
                x = x.to(td).contiguous()
                cur = self.t.cuda.current_stream(self.device)
                if self.planner is None:
                    cur.synchronize()
                elif not self.same_stream:
                    self.planner.torch_stream().wait_stream(cur)
            if shape is not None and tuple(x.shape) != tuple(shape):
                raise ValueError(f"expected shape {tuple(shape)}, got {tuple(x.shape)}")
            self.keep.append(x)
            return C.c_void_p(x.data_ptr())
        a = np.ascontiguousarray(x, dtype=dtype)
        if shape is not None and a.shape != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
        self.keep.append(a)
        return C.c_void_p(a.ctypes.data)

    def out(self, shape, dtype, into=None):
        if into is not None:             # the caller's own output array of an earlier call (plan_cycle(out=...)): same memory again
            if tuple(into.shape) != tuple(shape):
                raise ValueError(f"out: expected shape {tuple(shape)}, got {tuple(into.shape)}")
            if self.torch:
                self.outs.append(into)
                return into, C.c_void_p(into.data_ptr())
            return into, C.c_void_p(into.ctypes.data)
        if self.torch:
            td = {np.float64: self.t.float64, np.int32: self.t.int32}[dtype]
            # empty, not zeros: a fill kernel on torch's stream would race with ours; the library zero-fills
            # its outputs on the context's stream
            a = self.t.empty(tuple(shape), dtype=td, device=self.device)
            self.outs.append(a)
            return a, C.c_void_p(a.data_ptr())
        a = np.zeros(tuple(shape), dtype=dtype)
        return a, C.c_void_p(a.ctypes.data)


class HostRing:
    """The overlapped host path of the planning cycle (reference: one planning request per Pipe message, test_9.py:92-96,
    220, 390-395 - here a batch of them per call).  ``depth`` slots, each a set of page-locked NumPy arrays for the inputs
    and outputs of ``B`` scenes.  Per call: take the next slot (``next()`` waits, on the host, until the call that used it
    ``depth`` calls ago has delivered), fill its ``inputs`` in place or ``load(**arrays)`` them, ``plan_cycle(..., slot=slot)``;
    the call returns while the inputs are still crossing PCIe on the copy stream, the previous call computes, and the call
    before that is sending its outputs home.  ``slot.wait()`` - or taking the slot again - makes ``slot.outputs`` valid."""

    class Slot:
        def __init__(self, ring, index):
            self.ring, self.index = ring, index
            pl, B, P, mo, M, col = ring.planner, ring.B, ring.max_ref, ring.max_obs, ring.max_pts, ring.col
            self.B, self.max_ref, self.max_obs, self.max_pts = B, P, mo, M
            G = self.max_global = ring.max_global       # > 0: a slot for REQUESTS - the cycle starts from the global path
            self.use_dyn = False
            self._ticket = None
            f, i = np.float64, np.int32
            # ONE page-locked block for the inputs and one for the outputs (arrays at 256-byte offsets): the library then moves
            # each as a single PCIe copy (emp_context.h Stage::place)
            line = ((("global_path", (B, G, 4), f), ("n_global", (B,), i), ("pre_match_index", (B,), i)) if G > 0 else
                    (("ref_line", (B, P, 4), f), ("n_ref", (B,), i)))
            self.inputs, self._in_block = self._carve(pl, line + (
                ("origin_xy", (B, 2), f), ("start_xy", (B, 2), f),
                ("start_v", (B, 2), f), ("start_a", (B, 2), f), ("obs_xy", (B, max(mo, 1), 2), f), ("n_obs", (B,), i),
                ("dyn_dis_speed", (B, 2), f)))
            self.outputs, self._out_block = self._carve(pl, (
                ("dp_rows", (B, col), f), ("dp_s", (B, M), f), ("dp_l", (B, M), f), ("dp_len", (B,), i), ("path_s", (B, M), f),
                ("path_l", (B, M), f), ("path_len", (B,), i), ("traj", (B, M + 1, 4), f), ("traj_len", (B,), i),
                ("status", (B,), i)) + ((("match_index", (B,), i), ("ref_status", (B,), i)) if G > 0 else ()))

        @staticmethod
        def _carve(pl, spec):
            offs, total = [], 0
            for _, shp, dt in spec:
                offs.append(total)
                total += -(-int(np.prod(shp)) * np.dtype(dt).itemsize // 256) * 256
            block = pl.pinned_empty((total,), np.uint8)
            views = {}
            for (name, shp, dt), o in zip(spec, offs):
                n = int(np.prod(shp)) * np.dtype(dt).itemsize
                views[name] = block[o:o + n].view(dt).reshape(shp)
            return views, block

        def load(self, pool=None, **arrays):
            """Copy ordinary arrays into the slot's page-locked inputs (``dyn_dis_speed`` switches the virtual obstacles of
            test_9.py:137-169 on for this call).  Arrays with fewer scenes than the slot holds fill its head: set ``slot.B``
            to that count for the call.  ``pool``: a ``concurrent.futures`` executor - the big array (the
            reference lines) is then copied in chunks by its workers (NumPy's copy releases the GIL)."""
            self.use_dyn = arrays.get("dyn_dis_speed") is not None
            jobs = []
            for name, src in arrays.items():
                if src is None:
                    continue
                dst = self.inputs[name]
                src = np.asarray(src)
                if len(src) < len(dst):                    # a smaller batch in this slot: its scenes are the head of the arrays
                    dst = dst[:len(src)]
                if pool is not None and dst.nbytes >= (1 << 20):
                    n = len(dst)
                    parts = max(2, getattr(pool, "_max_workers", 4))
                    for k in range(parts):
                        a, b = n * k // parts, n * (k + 1) // parts
                        jobs.append(pool.submit(np.copyto, dst[a:b], src[a:b]))
                else:
                    np.copyto(dst, src)
            for j in jobs:
                j.result()
            return self

        def wait(self):
            """Block until this slot's latest call has delivered its outputs."""
            if self._ticket is None:
                return self
            pl = self.ring.planner
            rc = pl._lib.emp_wait_ticket(pl._h, int(self._ticket))      # thread-safe beside a call in progress (emplanner.h)
            if rc != 0:
                raise EmpError(f"emp_wait_ticket failed ({rc})")
            self._ticket = None
            return self

    def __init__(self, planner, p, B, max_ref, max_obs, max_pts, depth, max_global=0):
        self.planner, self.B, self.max_ref, self.max_obs, self.max_pts, self.col = planner, B, max_ref, max_obs, max_pts, int(p.col)
        self.max_global = int(max_global)
        if self.max_global > 0 and max_ref != REF_LINE_POINTS:
            raise ValueError("a request ring plans on the front end's 51-point reference lines")
        self.slots = [HostRing.Slot(self, k) for k in range(depth)]
        self._next = 0

    def depth_seen(self):
        return max(int(self.planner._lib.emp_pipeline_depth(self.planner._h)), 1)

    def next(self) -> "HostRing.Slot":
        s = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        return s.wait()

    def wait_all(self):
        for s in self.slots:
            s.wait()

    def close(self):
        self.planner.synchronize()
        for s in self.slots:
            s.inputs, s.outputs = {}, {}
            for block in (s._in_block, s._out_block):
                self.planner.pinned_free(block)
        self.slots = []


@dataclass
class CycleResult:
    dp_rows: object      # (B, col) float64
    dp_s: object         # (B, max_pts)
    dp_l: object
    dp_len: object       # (B,) int32
    path_s: object
    path_l: object
    path_len: object
    traj: object         # (B, max_pts + 1, 4)
    traj_len: object
    status: object       # (B,) int32 bit mask
    match_index: object = None   # (B,) int32 - only from a cycle that started at the global path (front end fused in)
    ref_status: object = None    # (B,) int32 - the front end's own status (OR it into ``status``)


class Planner:
    """One device context.  Create it AFTER forking (the reference plans in a child process)."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load()
        h = C.c_void_p()
        rc = self._lib.emp_create(int(device_id), C.byref(h))
        if rc != 0:
            msg = self._lib.emp_last_error(None)
            raise EmpError(f"emp_create({device_id}) failed ({rc}): {msg.decode() if msg else ''}")
        self._h = h
        self.device_id = int(device_id)
        self._torch_stream = None
        self._torch_lane_streams = {}
        self.pipelined = False
        self.pipe_mode = 0
        self.in_flight = 1
        self._retain = 1
        self._inflight = []
        self._pinned = {}                # data address -> emp_host_alloc pointer of the pinned_empty arrays still alive
        self._cur = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.emp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        cur, self._cur = self._cur, None
        if rc != 0:
            msg = self._lib.emp_last_error(self._h)
            raise EmpError(f"libemplanner call failed ({rc}): {msg.decode() if msg else ''}")
        if cur is not None:
            cur.done()

    # ---- housekeeping ------------------------------------------------------------------
    def synchronize(self):
        self._check(self._lib.emp_synchronize(self._h))

    def set_timing(self, enabled: bool, only: str | None = None):
        """Bracket kernel launches with HIP events on the planner's stream; ``only`` restricts the events to one
        named kernel (an event pair costs a few microseconds of stream time per launch)."""
        self._check(self._lib.emp_set_timing_filter(self._h, only.encode() if only else None))
        self._check(self._lib.emp_set_timing(self._h, int(bool(enabled))))

    def kernel_ms(self, name: str) -> float:
        """Mean duration (ms) of the launches of kernel ``name`` since set_timing(True); < 0 if none."""
        return float(self._lib.emp_kernel_ms(self._h, name.encode()))

    def kernel_launches(self, name: str) -> int:
        return int(self._lib.emp_kernel_launches(self._h, name.encode()))

    def kernel_samples(self, name: str) -> np.ndarray:
        """The individual durations (ms) behind ``kernel_ms``, in launch order."""
        n = self.kernel_launches(name)
        out = np.zeros(max(n, 1), np.float64)
        got = int(self._lib.emp_kernel_samples(self._h, name.encode(), out.ctypes.data, n))
        if got < 0:
            raise RuntimeError("emp_kernel_samples failed")
        return out[:min(n, got)]

    @property
    def stream(self):
        """Raw hipStream_t of the context."""
        return self._lib.emp_stream(self._h)

    def set_pipeline(self, mode=True):
        """Several batches in flight for consecutive ``plan_cycle`` calls on device tensors (include/emplanner.h,
        emp_set_pipeline).  ``mode``: False / 0 = off; True / "staged" / 1 = two batches, the back stage (path QP,
        Cartesian tail) of one overlapping the front stage (projection, DP) of the next; an int n >= 2 = n batches on n
        lanes (a stream and a pool of temporaries each), whole cycles overlapping freely - the highest throughput; "auto" =
        the library picks three lanes when every stream of the process gets a hardware queue of its own (GPU_MAX_HW_QUEUES,
        option "foreign_streams") and the staged form otherwise (``pipeline_form()`` tells).  The
        outputs of a cycle are then complete on ``torch_result_stream()`` (lane mode: the lane of the LATEST call) and
        ``synchronize()`` waits for everything."""
        auto = isinstance(mode, str) and mode == "auto"
        m = L.EMP_PIPELINE_AUTO if auto else L.EMP_PIPELINE_STAGED if (mode is True or mode == "staged") else max(int(mode), 0)
        if m >= 2 and m + 1 > L.hw_queues():
            import warnings
            warnings.warn(f"{m} lanes + the main stream on GPU_MAX_HW_QUEUES={L.hw_queues()} hardware queues: lanes will share "
                          "queues and serialise (set_pipeline('auto') picks the form the process can sustain; "
                          "emplanner_carla_amd._lib.configure_hw_queues() before HIP initialises gives it the queues)",
                          RuntimeWarning, stacklevel=2)
        self._check(self._lib.emp_set_pipeline(self._h, m))
        if auto:                                 # the library chose: three lanes if every stream gets a hardware queue, else staged
            m = int(self._lib.emp_pipeline_form(self._h, None, None))
        self._torch_lane_streams = {}            # the library destroys the streams the new mode does not use
        self.pipe_mode = m
        self.in_flight = 2 if m == L.EMP_PIPELINE_STAGED else max(m, 1)       # batches that overlap on the GPU
        self._retain = max(int(self._lib.emp_pipeline_depth(self._h)), self.in_flight)     # calls whose outputs stay referenced
        self.pipelined = m != 0
        self._inflight = []                      # emp_set_pipeline has drained every stream

    def pipeline_form(self):
        """(form, hardware queues seen, other streams counted): form 0 = off, 1 = staged, n = lanes (emp_pipeline_form); the two
        counts are what the latest ``set_pipeline("auto")`` based its choice on (0, 0 after an explicit mode)."""
        q, o = C.c_int32(0), C.c_int32(0)
        return int(self._lib.emp_pipeline_form(self._h, C.byref(q), C.byref(o))), int(q.value), int(o.value)

    def set_fence(self, enabled: bool):
        """emp_set_fence: whether calls other than a pipelined ``plan_cycle`` wait for the cycles in flight (default) or
        overlap them (``False``: only for work that does not read a cycle's outputs)."""
        self._check(self._lib.emp_set_fence(self._h, 1 if enabled else 0))

    def set_option(self, name, value: int):
        """emp_set_option (include/emplanner.h, emp_option): ``name`` is a key of ``_lib.OPTIONS`` (the list is appended to this
        docstring at import) or the option's number.  Takes effect at the next call ("foreign_streams": at the next
        ``set_pipeline("auto")``).  The library reads no environment variable."""
        key = L.OPTIONS[name] if isinstance(name, str) else int(name)
        self._check(self._lib.emp_set_option(self._h, key, int(value)))

    def get_option(self, name) -> int:
        key = L.OPTIONS[name] if isinstance(name, str) else int(name)
        v = C.c_int32(0)
        self._check(self._lib.emp_get_option(self._h, key, C.byref(v)))
        return int(v.value)

    def sweep_clock(self):
        """With option "sweep_clock_probe" on: (shader clock in MHz, mean and longest wavefront residence in us) over the sweep
        launches recorded since the probe was switched on (the 32 most recent of them), or None when nothing was recorded."""
        mean_us, max_us = C.c_double(0.0), C.c_double(0.0)
        mhz = float(self._lib.emp_sweep_clock_mhz(self._h, C.byref(mean_us), C.byref(max_us)))
        return None if mhz < 0 else (mhz, float(mean_us.value), float(max_us.value))

    def cycle_graph_replays(self) -> int:
        """Option "cycle_graph": how many plan_cycle calls of this planner were one hipGraphLaunch (emp_cycle_graph_replays)."""
        return int(self._lib.emp_cycle_graph_replays(self._h))

    def edge_probe(self):
        """With option "edge_clock_probe" on: (mean wavefront residence us, first start to last end us, mean wavefronts resident
        at once, wavefronts) of the latest edge-cost launch, or None when nothing was recorded."""
        a, b, c, n = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
        if self._lib.emp_edge_probe(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(n)) != 0:
            return None
        return float(a.value), float(b.value), float(c.value), int(n.value)

    def edge_clock_mhz(self):
        """With option "edge_clock_probe" on: the shader clock (MHz) of the latest edge-cost launch, or None (emp_edge_clock_mhz)."""
        mhz = float(self._lib.emp_edge_clock_mhz(self._h))
        return None if mhz < 0 else mhz

    def sweep_probe_spans(self):
        """With option "sweep_clock_probe" on: (us between the first and the last wavefront START of a sweep launch, us from
        the first wavefront's start to the last wavefront's end), averaged over the recorded launches; None when nothing
        was recorded."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        rc = int(self._lib.emp_sweep_probe_spans(self._h, C.byref(a), C.byref(b)))
        return None if rc != 0 else (float(a.value), float(b.value))

    def torch_result_stream(self):
        """The stream on which the latest cycle's outputs become complete (its lane in pipelined mode)."""
        if not self.pipelined:
            return self.torch_stream()
        import torch
        h = int(self._lib.emp_result_stream(self._h))
        st = self._torch_lane_streams.get(h)
        if st is None:
            st = self._torch_lane_streams[h] = torch.cuda.ExternalStream(h, device=torch.device("cuda", self.device_id))
        return st

    def torch_stream(self):
        """The context's stream as a torch stream: ``with torch.cuda.stream(pl.torch_stream()):`` orders torch
        work (e.g. an RCCL gather of the results) after the planner's kernels without a host sync."""
        if self._torch_stream is None:
            import torch
            self._torch_stream = torch.cuda.ExternalStream(int(self.stream), device=torch.device("cuda", self.device_id))
        return self._torch_stream

    def _args(self, *inputs):
        self._cur = _Args(*inputs, planner=self)
        return self._cur

    # ---- DP ------------------------------------------------------------------------------
    def edge_tensor_elems(self, p: DpParams, B: int, layout=L.EMP_EDGE_CANONICAL) -> int:
        return int(self._lib.emp_edge_tensor_elems(C.byref(p), int(B), int(layout)))

    def dp_edge_costs(self, p: DpParams, obs_s, obs_l, n_obs, start, layout=L.EMP_EDGE_CANONICAL):
        """ref cal_start_cost / cal_neighbor_cost for all lattice edges.
        returns start_cost (B,row) and edge: canonical (B, col-1, row_i, row_k) or tiled (flat)."""
        a = self._args(obs_s, obs_l, n_obs, start)
        B = int(start.shape[0])
        mo = int(obs_s.shape[1]) if obs_s is not None and len(obs_s.shape) == 2 else 0
        c0, c0p = a.out((B, p.row), np.float64)
        n = self.edge_tensor_elems(p, B, layout)
        e, ep = a.out((n,), np.float64)
        self._check(self._lib.emp_dp_edge_costs(
            self._h, C.byref(p), B, mo, a.inp(obs_s, np.float64, (B, mo)), a.inp(obs_l, np.float64, (B, mo)),
            a.inp(n_obs, np.int32, (B,)), a.inp(start, np.float64, (B, 4)), c0p, ep, int(layout), a.where))
        if layout == L.EMP_EDGE_CANONICAL or p.row > 32:      # beyond 32 rows the library's tensor is the canonical one either way
            e = e.reshape(B, p.col - 1, p.row, p.row)
        return c0, e

    def dp_plan(self, p: DpParams, obs_s, obs_l, n_obs, start, mode=L.EMP_DP_TWO_KERNEL):
        """ref DP_algorithm up to the backtrack: returns rows (B,col) f64, min_cost (B,), status (B,)."""
        a = self._args(obs_s, obs_l, n_obs, start)
        B = int(start.shape[0])
        mo = int(obs_s.shape[1]) if obs_s is not None and len(obs_s.shape) == 2 else 0
        rows, rp = a.out((B, p.col), np.float64)
        mc, mp = a.out((B,), np.float64)
        st, sp = a.out((B,), np.int32)
        self._check(self._lib.emp_dp_plan(
            self._h, C.byref(p), B, mo, a.inp(obs_s, np.float64, (B, mo)), a.inp(obs_l, np.float64, (B, mo)),
            a.inp(n_obs, np.int32, (B,)), a.inp(start, np.float64, (B, 4)), int(mode), rp, mp, sp, a.where))
        return rows, mc, st

    def dp_sweep(self, p: DpParams, start_cost, edge_tiled):
        """Min-plus sweep + backtrack on a caller-provided tiled edge tensor."""
        a = self._args(start_cost, edge_tiled)
        B = int(start_cost.shape[0])
        rows, rp = a.out((B, p.col), np.float64)
        mc, mp = a.out((B,), np.float64)
        st, sp = a.out((B,), np.int32)
        self._check(self._lib.emp_dp_sweep(
            self._h, C.byref(p), B, a.inp(start_cost, np.float64, (B, p.row)),
            a.inp(edge_tiled, np.float64, (self.edge_tensor_elems(p, B, L.EMP_EDGE_TILED),)), rp, mp, sp, a.where))
        return rows, mc, st

    def dp_enrich(self, p: DpParams, rows, start, max_pts: int):
        """ref enrich_DP_s_l: rows -> (path_s, path_l) padded to max_pts, path_len, status."""
        a = self._args(rows, start)
        B = int(start.shape[0])
        ps, psp = a.out((B, max_pts), np.float64)
        pl, plp = a.out((B, max_pts), np.float64)
        ln, lnp = a.out((B,), np.int32)
        st, sp = a.out((B,), np.int32)
        self._check(self._lib.emp_dp_enrich(
            self._h, C.byref(p), B, a.inp(rows, np.float64, (B, p.col)), a.inp(start, np.float64, (B, 4)),
            int(max_pts), psp, plp, lnp, sp, a.where))
        return ps, pl, ln, st

    def enrich_nodes(self, node_s, node_l, n_nodes, start, resolution, max_pts: int):
        """ref enrich_DP_s_l on explicit node lists: returns path_s, path_l (B,max_pts), path_len, status."""
        a = self._args(node_s, node_l, start)
        B, K = int(node_s.shape[0]), int(node_s.shape[1])
        ps, psp = a.out((B, max_pts), np.float64)
        pl, plp = a.out((B, max_pts), np.float64)
        ln, lnp = a.out((B,), np.int32)
        st, sp = a.out((B,), np.int32)
        self._check(self._lib.emp_enrich_nodes(
            self._h, B, K, float(resolution), a.inp(node_s, np.float64, (B, K)), a.inp(node_l, np.float64, (B, K)),
            a.inp(n_nodes, np.int32, (B,)), a.inp(start, np.float64, (B, 4)), int(max_pts), psp, plp, lnp, sp, a.where))
        return ps, pl, ln, st

    # ---- Cartesian <-> Frenet -------------------------------------------------------------
    def frenet_project(self, ref_line, n_ref, origin_xy, start_xy, start_v, start_a, obs_xy, n_obs):
        """ref cal_s_map_fun + cal_s_l_fun (obstacles, start) + cal_s_l_deri_fun (start): test_9.py:113-177.
        returns s_map (B,P), obs_s (B,mo), obs_l (B,mo), begin_sl (B,2), start (B,4)."""
        a = self._args(ref_line, origin_xy)
        B, P = int(ref_line.shape[0]), int(ref_line.shape[1])
        mo = int(obs_xy.shape[1]) if obs_xy is not None else 0
        sm, smp = a.out((B, P), np.float64)
        os_, osp = a.out((B, mo), np.float64)
        ol_, olp = a.out((B, mo), np.float64)
        bsl, bslp = a.out((B, 2), np.float64)
        st, stp = a.out((B, 4), np.float64)
        self._check(self._lib.emp_frenet_project(
            self._h, B, P, mo, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(n_ref, np.int32, (B,)),
            a.inp(origin_xy, np.float64, (B, 2)), a.inp(start_xy, np.float64, (B, 2)),
            a.inp(start_v, np.float64, (B, 2)), a.inp(start_a, np.float64, (B, 2)),
            a.inp(obs_xy, np.float64, (B, mo, 2)) if mo else None, a.inp(n_obs, np.int32, (B,)) if mo else None,
            smp, osp if mo else None, olp if mo else None, bslp, stp, a.where))
        return sm, os_, ol_, bsl, st

    def match_projection(self, ref_line, n_ref, xy, n_pts):
        """ref match_projection_points: returns match_index (B,K) int32, proj (B,K,4)."""
        a = self._args(ref_line, xy)
        B, P, K = int(ref_line.shape[0]), int(ref_line.shape[1]), int(xy.shape[1])
        mi, mip = a.out((B, K), np.int32)
        pr, prp = a.out((B, K, 4), np.float64)
        self._check(self._lib.emp_match_projection(
            self._h, B, P, K, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(n_ref, np.int32, (B,)),
            a.inp(xy, np.float64, (B, K, 2)), a.inp(n_pts, np.int32, (B,)), mip, prp, a.where))
        return mi, pr

    def find_match_points(self, ref_line, n_ref, xy, n_pts, is_first_run, pre_match_index):
        """ref find_match_points: returns match_index (B,K) int32, proj (B,K,4)."""
        a = self._args(ref_line, xy)
        B, P, K = int(ref_line.shape[0]), int(ref_line.shape[1]), int(xy.shape[1])
        mi, mip = a.out((B, K), np.int32)
        pr, prp = a.out((B, K, 4), np.float64)
        self._check(self._lib.emp_find_match_points(
            self._h, B, P, K, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(n_ref, np.int32, (B,)),
            a.inp(xy, np.float64, (B, K, 2)), a.inp(n_pts, np.int32, (B,)), a.inp(is_first_run, np.int32, (B,)),
            a.inp(pre_match_index, np.int32, (B,)), mip, prp, a.where))
        return mi, pr

    def heading_kappa(self, xy, n_pts):
        """ref cal_heading_kappa: xy (B,M,2) -> theta, kappa (B,M)."""
        a = self._args(xy)
        B, M = int(xy.shape[0]), int(xy.shape[1])
        th, thp = a.out((B, M), np.float64)
        kp, kpp = a.out((B, M), np.float64)
        self._check(self._lib.emp_heading_kappa(self._h, B, M, a.inp(xy, np.float64, (B, M, 2)),
                                                a.inp(n_pts, np.int32, (B,)), thp, kpp, a.where))
        return th, kp

    def s_map(self, ref_line, n_ref, origin_xy):
        """ref cal_s_map_fun: returns s_map (B,P)."""
        a = self._args(ref_line, origin_xy)
        B, P = int(ref_line.shape[0]), int(ref_line.shape[1])
        sm, smp = a.out((B, P), np.float64)
        self._check(self._lib.emp_s_map(self._h, B, P, a.inp(ref_line, np.float64, (B, P, 4)),
                                        a.inp(n_ref, np.int32, (B,)), a.inp(origin_xy, np.float64, (B, 2)), smp, a.where))
        return sm

    def s_l(self, ref_line, s_map, n_ref, xy, n_pts, match_index=None, want_l=True):
        """ref cal_s_l_fun (or cal_projection_s_fun when match_index is given): returns s, l (B,K)."""
        a = self._args(ref_line, xy)
        B, P, K = int(ref_line.shape[0]), int(ref_line.shape[1]), int(xy.shape[1])
        s_, sp_ = a.out((B, K), np.float64)
        l_, lp_ = a.out((B, K), np.float64) if want_l else (None, None)
        self._check(self._lib.emp_s_l(
            self._h, B, P, K, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(s_map, np.float64, (B, P)),
            a.inp(n_ref, np.int32, (B,)), a.inp(xy, np.float64, (B, K, 2)), a.inp(n_pts, np.int32, (B,)),
            a.inp(match_index, np.int32, (B, K)) if match_index is not None else None, sp_, lp_, a.where))
        return s_, l_

    def s_l_deri(self, ref_line, n_ref, xy, v_xy, a_xy, n_pts, origin_xy):
        """ref cal_s_l_deri_fun: returns (B,K,7) = l, dl/dt, ds/dt, d2l/dt2, dl/ds, d2s/dt2, d2l/ds2."""
        a = self._args(ref_line, xy)
        B, P, K = int(ref_line.shape[0]), int(ref_line.shape[1]), int(xy.shape[1])
        o, op_ = a.out((B, K, 7), np.float64)
        self._check(self._lib.emp_s_l_deri(
            self._h, B, P, K, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(n_ref, np.int32, (B,)),
            a.inp(xy, np.float64, (B, K, 2)), a.inp(v_xy, np.float64, (B, K, 2)), a.inp(a_xy, np.float64, (B, K, 2)),
            a.inp(n_pts, np.int32, (B,)), a.inp(origin_xy, np.float64, (B, 2)), op_, a.where))
        return o

    def proj_point(self, ref_line, s_map, n_ref, s, pre_match_index):
        """ref cal_proj_point for n independent queries: returns out (n,4), index (n,), status (n,)."""
        a = self._args(ref_line, s)
        n, P = int(ref_line.shape[0]), int(ref_line.shape[1])
        o, op_ = a.out((n, 4), np.float64)
        ix, ixp = a.out((n,), np.int32)
        st, stp = a.out((n,), np.int32)
        self._check(self._lib.emp_proj_point(
            self._h, n, P, a.inp(ref_line, np.float64, (n, P, 4)), a.inp(s_map, np.float64, (n, P)),
            a.inp(n_ref, np.int32, (n,)), a.inp(s, np.float64, (n,)), a.inp(pre_match_index, np.int32, (n,)), op_, ixp,
            stp, a.where))
        return o, ix, st

    def trajectory_index2s(self, x, y, n_pts):
        a = self._args(x, y)
        B, M = int(x.shape[0]), int(x.shape[1])
        o, op_ = a.out((B, M), np.float64)
        self._check(self._lib.emp_trajectory_index2s(self._h, B, M, a.inp(x, np.float64, (B, M)),
                                                     a.inp(y, np.float64, (B, M)), a.inp(n_pts, np.int32, (B,)), op_,
                                                     a.where))
        return o

    def frenet2cartesian(self, ref_line, index2s, n_ref, sl, n_pts, proj_only=False):
        """ref Frenet2Cartesian / CalcProjPoint: sl (B,K,4) = s,l,dl,ddl -> out (B,K,4), status (B,)."""
        a = self._args(ref_line, sl)
        B, P, K = int(ref_line.shape[0]), int(ref_line.shape[1]), int(sl.shape[1])
        o, op_ = a.out((B, K, 4), np.float64)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_frenet2cartesian(
            self._h, B, P, K, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(index2s, np.float64, (B, P)),
            a.inp(n_ref, np.int32, (B,)), a.inp(sl, np.float64, (B, K, 4)), a.inp(n_pts, np.int32, (B,)), op_, stp,
            int(bool(proj_only)), a.where))
        return o, st

    def dy_obs_deri(self, rows):
        """ref cal_dy_obs_deri: rows (n,5) = l, vx, vy, heading, kappa -> (n,3) = s_dot, l_dot, dl."""
        a = self._args(rows)
        n = int(rows.shape[0])
        o, op_ = a.out((n, 3), np.float64)
        self._check(self._lib.emp_dy_obs_deri(self._h, n, a.inp(rows, np.float64, (n, 5)), op_, a.where))
        return o

    # ---- front end of the cycle (reference test_9.py:99-110) --------------------------------------
    def reference_line(self, sp: SmoothParams, global_path, n_global, pred_xy, pre_match_index, is_first_run=None):
        """find_match_points (one point, windowed) -> sampling -> smooth_reference_line.  global_path (B,G,4),
        pred_xy (B,2), pre_match_index (B,) -> ref_line (B,51,4), n_ref (B,), match_index (B,), iters, status."""
        a = self._args(global_path, pred_xy)
        B, G = int(global_path.shape[0]), int(global_path.shape[1])
        ref, refp = a.out((B, REF_LINE_POINTS, 4), np.float64)
        nr, nrp = a.out((B,), np.int32)
        mi, mip = a.out((B,), np.int32)
        it, itp = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        first = a.inp(is_first_run, np.int32, (B,)) if is_first_run is not None else None
        self._check(self._lib.emp_reference_line(
            self._h, C.byref(sp), B, G, a.inp(global_path, np.float64, (B, G, 4)), a.inp(n_global, np.int32, (B,)),
            a.inp(pred_xy, np.float64, (B, 2)), first, a.inp(pre_match_index, np.int32, (B,)), refp, nrp, mip, itp, stp,
            a.where))
        return ref, nr, mi, it, st

    # ---- lateral MPC controller (reference controller/controller.py:65-337) -----------------------
    def mpc_lateral(self, p: MpcParams, target_path, n_path, state, vx, min_index, qp_matrices=False) -> MpcResult:
        """ref Lateral_MPC_controller._control for B vehicles: target_path (B,M,4), state (B,5) = x, y, fi, Vy, fi_dot,
        vx (B,), min_index (B,)."""
        a = self._args(target_path, state)
        B, M = int(target_path.shape[0]), int(target_path.shape[1])
        steer, sp_ = a.out((B,), np.float64)
        u, up = a.out((B, 12), np.float64)
        e, ep = a.out((B, 4), np.float64)
        k, kp = a.out((B,), np.float64)
        mi, mip = a.out((B,), np.int32)
        pp, ppp = a.out((B, 4), np.float64)
        H, Hp = a.out((B, 12, 12), np.float64) if qp_matrices else (None, None)
        f, fp = a.out((B, 12), np.float64) if qp_matrices else (None, None)
        it, itp = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_mpc_lateral(
            self._h, C.byref(p), B, M, a.inp(target_path, np.float64, (B, M, 4)), a.inp(n_path, np.int32, (B,)),
            a.inp(state, np.float64, (B, 5)), a.inp(vx, np.float64, (B,)), a.inp(min_index, np.int32, (B,)), sp_, up, ep, kp,
            mip, ppp, Hp, fp, itp, stp, a.where))
        return MpcResult(steer, u, e, k, mi, pp, H, f, it, st)

    def lqr_lateral(self, p: MpcParams, target_path, n_path, state, vx, min_index) -> LqrResult:
        """ref Lateral_LQR_controller._control for B vehicles (same inputs as mpc_lateral)."""
        a = self._args(target_path, state)
        B, M = int(target_path.shape[0]), int(target_path.shape[1])
        steer, sp_ = a.out((B,), np.float64)
        K, Kp = a.out((B, 4), np.float64)
        e, ep = a.out((B, 4), np.float64)
        k, kp = a.out((B,), np.float64)
        mi, mip = a.out((B,), np.int32)
        pp, ppp = a.out((B, 4), np.float64)
        sw, swp = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_lqr_lateral(
            self._h, C.byref(p), B, M, a.inp(target_path, np.float64, (B, M, 4)), a.inp(n_path, np.int32, (B,)),
            a.inp(state, np.float64, (B, 5)), a.inp(vx, np.float64, (B,)), a.inp(min_index, np.int32, (B,)), sp_, Kp, ep, kp,
            mip, ppp, swp, stp, a.where))
        return LqrResult(steer, K, e, k, mi, pp, sw, st)

    # ---- S-T speed DP (reference planner/speed_planning_test.py) ------------------------------
    def st_graph(self, obs_s, obs_l, obs_s_dot, obs_l_dot):
        """ref generate_st_graph: four (B, K) arrays (NaN = empty slot) -> s_in, s_out, t_in, t_out (B, K)."""
        a = self._args(obs_s, obs_l, obs_s_dot, obs_l_dot)
        B, K = int(obs_s.shape[0]), int(obs_s.shape[1])
        outs = [a.out((B, K), np.float64) for _ in range(4)]
        self._check(self._lib.emp_st_graph(
            self._h, B, K, a.inp(obs_s, np.float64, (B, K)), a.inp(obs_l, np.float64, (B, K)),
            a.inp(obs_s_dot, np.float64, (B, K)), a.inp(obs_l_dot, np.float64, (B, K)),
            outs[0][1], outs[1][1], outs[2][1], outs[3][1], a.where))
        return tuple(o[0] for o in outs)

    def speed_dp(self, p: SpeedDpParams, s_in, s_out, t_in, t_out, plan_start_s_dot, tables=True) -> SpeedDpResult:
        """ref speed_DP: S-T segments (B, K) + start speed (B,) -> tables, terminal node, chosen (s, t) per column."""
        a = self._args(s_in, s_out, t_in, t_out, plan_start_s_dot)
        B, K = int(s_in.shape[0]), int(s_in.shape[1])
        shape = (B, ST_ROWS, ST_COLS)
        cost, cp = a.out(shape, np.float64) if tables else (None, None)
        sd, sp = a.out(shape, np.float64) if tables else (None, None)
        nd, np_ = a.out(shape, np.int32) if tables else (None, None)
        en, ep = a.out((B, 2), np.int32)
        ss, ssp = a.out((B, ST_COLS), np.float64)
        tt, ttp = a.out((B, ST_COLS), np.float64)
        self._check(self._lib.emp_speed_dp(
            self._h, C.byref(p), B, K, a.inp(s_in, np.float64, (B, K)), a.inp(s_out, np.float64, (B, K)),
            a.inp(t_in, np.float64, (B, K)), a.inp(t_out, np.float64, (B, K)),
            a.inp(plan_start_s_dot, np.float64, (B,)), cp, sp, np_, ep, ssp, ttp, a.where))
        return SpeedDpResult(cost, sd, nd, en, ss, tt)

    def st_edge_costs(self, p: SpeedDpParams, edges, s_in, s_out, t_in, t_out):
        """ref CalcDpCost / CalcObsCost: edges (B, E, 5) = s0, t0, s_dot0, s1, t1 -> total (B, E), obstacle term (B, E)."""
        a = self._args(edges, s_in)
        B, E, K = int(edges.shape[0]), int(edges.shape[1]), int(s_in.shape[1])
        tot, tp = a.out((B, E), np.float64)
        obs, op_ = a.out((B, E), np.float64)
        self._check(self._lib.emp_st_edge_costs(
            self._h, C.byref(p), B, E, K, a.inp(edges, np.float64, (B, E, 5)), a.inp(s_in, np.float64, (B, K)),
            a.inp(s_out, np.float64, (B, K)), a.inp(t_in, np.float64, (B, K)), a.inp(t_out, np.float64, (B, K)),
            tp, op_, a.where))
        return tot, obs

    def st_collision_cost(self, w_cost_obs, min_dis):
        """ref CalcCollisionCost: distances (n,) -> costs (n,)."""
        a = self._args(min_dis)
        n = int(min_dis.shape[0])
        c, cp = a.out((n,), np.float64)
        self._check(self._lib.emp_st_collision_cost(self._h, n, float(w_cost_obs), a.inp(min_dis, np.float64, (n,)), cp,
                                                    a.where))
        return c

    def speed_start_condition(self, vx, vy, ax, ay, heading):
        """ref calc_speed_planning_start_condition (:23-35): Cartesian velocity / acceleration and heading of the planning
        start, (n,) each -> plan_start_s_dot, plan_start_s_dot2 (n,)."""
        a = self._args(vx, vy, ax, ay, heading)
        n = int(vx.shape[0])
        s1, p1 = a.out((n,), np.float64)
        s2, p2 = a.out((n,), np.float64)
        self._check(self._lib.emp_speed_start_condition(
            self._h, n, a.inp(vx, np.float64, (n,)), a.inp(vy, np.float64, (n,)), a.inp(ax, np.float64, (n,)),
            a.inp(ay, np.float64, (n,)), a.inp(heading, np.float64, (n,)), p1, p2, a.where))
        return s1, s2

    # ---- S-T speed planning back end (reference speed_planning_test.py:308-620) ---------------------
    def speed_convex_space(self, dp_speed_s, dp_speed_t, path_index2s, path_kappa, path_len, s_in, s_out, t_in, t_out,
                           max_lateral_accel=0.2 * 9.8):
        """ref generate_convex_space: DP profile (B,16) + path (B,P) + S-T segments (B,K) -> s_lb, s_ub, s_dot_lb,
        s_dot_ub (B,16), status (B,)."""
        a = self._args(dp_speed_s, path_index2s, s_in)
        B, P, K = int(dp_speed_s.shape[0]), int(path_index2s.shape[1]), int(s_in.shape[1])
        outs = [a.out((B, SPEED_DP_COLS), np.float64) for _ in range(4)]
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_speed_convex_space(
            self._h, B, K, P, float(max_lateral_accel), a.inp(dp_speed_s, np.float64, (B, SPEED_DP_COLS)),
            a.inp(dp_speed_t, np.float64, (B, SPEED_DP_COLS)), a.inp(path_index2s, np.float64, (B, P)),
            a.inp(path_kappa, np.float64, (B, P)), a.inp(path_len, np.int32, (B,)), a.inp(s_in, np.float64, (B, K)),
            a.inp(s_out, np.float64, (B, K)), a.inp(t_in, np.float64, (B, K)), a.inp(t_out, np.float64, (B, K)),
            outs[0][1], outs[1][1], outs[2][1], outs[3][1], stp, a.where))
        return outs[0][0], outs[1][0], outs[2][0], outs[3][0], st

    def speed_qp(self, p: SpeedQpParams, plan_start_s_dot, plan_start_s_dot2, dp_speed_s, dp_speed_t, s_lb, s_ub, s_dot_lb,
                 s_dot_ub):
        """ref speed_QP (the problem it states, see include/emplanner.h): -> qp_s, qp_s_dot, qp_s_dot2, relative_time
        (B,17), iters (B,), status (B,)."""
        a = self._args(dp_speed_s, s_lb)
        B = int(dp_speed_s.shape[0])
        outs = [a.out((B, SPEED_QP_POINTS), np.float64) for _ in range(4)]
        it, itp = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        c16 = lambda x: a.inp(x, np.float64, (B, SPEED_DP_COLS))
        self._check(self._lib.emp_speed_qp(
            self._h, C.byref(p), B, a.inp(plan_start_s_dot, np.float64, (B,)), a.inp(plan_start_s_dot2, np.float64, (B,)),
            c16(dp_speed_s), c16(dp_speed_t), c16(s_lb), c16(s_ub), c16(s_dot_lb), c16(s_dot_ub), outs[0][1], outs[1][1],
            outs[2][1], outs[3][1], itp, stp, a.where))
        return outs[0][0], outs[1][0], outs[2][0], outs[3][0], it, st

    def speed_increase_points(self, s_init, s_dot_init, s_dot2_init, relative_time_init):
        """ref increase_points: (B,17) profiles -> s, s_dot, s_dot2, relative_time (B,401), status (B,)."""
        a = self._args(s_init)
        B = int(s_init.shape[0])
        outs = [a.out((B, SPEED_DENSE_POINTS), np.float64) for _ in range(4)]
        st, stp = a.out((B,), np.int32)
        c17 = lambda x: a.inp(x, np.float64, (B, SPEED_QP_POINTS))
        self._check(self._lib.emp_speed_increase_points(self._h, B, c17(s_init), c17(s_dot_init), c17(s_dot2_init),
                                                        c17(relative_time_init), outs[0][1], outs[1][1], outs[2][1],
                                                        outs[3][1], stp, a.where))
        return outs[0][0], outs[1][0], outs[2][0], outs[3][0], st

    def path_speed_merge(self, s, s_dot, s_dot2, relative_time, current_time, path_s, x_init, y_init, heading_init,
                         kappa_init, n_init):
        """ref path_speed_merge: speed samples (B,401) x path arrays (B,P) -> trajectory (B,7,401) = x, y, heading,
        kappa, speed, accel, time; status (B,)."""
        a = self._args(s, path_s)
        B, P = int(s.shape[0]), int(path_s.shape[1])
        out, outp = a.out((B, 7, SPEED_DENSE_POINTS), np.float64)
        st, stp = a.out((B,), np.int32)
        d = lambda x: a.inp(x, np.float64, (B, SPEED_DENSE_POINTS))
        q = lambda x: a.inp(x, np.float64, (B, P))
        self._check(self._lib.emp_path_speed_merge(
            self._h, B, P, d(s), d(s_dot), d(s_dot2), d(relative_time), a.inp(current_time, np.float64, (B,)), q(path_s),
            q(x_init), q(y_init), q(heading_init), q(kappa_init), a.inp(n_init, np.int32, (B,)), outp, stp, a.where))
        return out, st

    def speed_plan(self, dp: SpeedDpParams, qp: SpeedQpParams, obs_s, obs_l, obs_s_dot, obs_l_dot, plan_start_s_dot,
                   plan_start_s_dot2, path_index2s, path_kappa, path_len, current_time, path_s, x_init, y_init,
                   heading_init, kappa_init, n_init, max_lateral_accel=0.2 * 9.8):
        """The whole S-T speed planner for B scenes, stage by stage on the device (the order of the reference's
        functions in speed_planning_test.py: generate_st_graph -> speed_DP -> generate_convex_space -> speed_QP ->
        increase_points -> path_speed_merge).  With torch inputs nothing leaves the GPU between the stages.  Returns a
        dict: trajectory (B,7,401), status (B,) = OR of the stages' status bits, and every intermediate result."""
        sets = self.st_graph(obs_s, obs_l, obs_s_dot, obs_l_dot)
        res = self.speed_dp(dp, *sets, plan_start_s_dot, tables=False)
        cs = self.speed_convex_space(res.speed_s, res.speed_t, path_index2s, path_kappa, path_len, *sets,
                                     max_lateral_accel=max_lateral_accel)
        q = self.speed_qp(qp, plan_start_s_dot, plan_start_s_dot2, res.speed_s, res.speed_t, *cs[:4])
        d = self.speed_increase_points(*q[:4])
        traj, st_merge = self.path_speed_merge(*d[:4], current_time, path_s, x_init, y_init, heading_init, kappa_init, n_init)
        status = cs[4] | q[5] | d[4] | st_merge
        return dict(trajectory=traj, status=status, st_sets=sets, dp=res, convex_space=cs[:4], qp=q[:4], qp_iters=q[4],
                    dense=d[:4], stage_status=(cs[4], q[5], d[4], st_merge))

    # ---- QP stages ------------------------------------------------------------------------
    def lmin_lmax(self, dp_s, dp_l, n_pts, obs_s, obs_l, n_obs, obs_length, obs_width):
        """ref cal_lmin_lmax: returns l_min, l_max (B,M), status (B,)."""
        a = self._args(dp_s, obs_s)
        B, M = int(dp_s.shape[0]), int(dp_s.shape[1])
        mo = int(obs_s.shape[1])
        lo, lop = a.out((B, M), np.float64)
        hi, hip = a.out((B, M), np.float64)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_lmin_lmax(
            self._h, B, M, mo, a.inp(dp_s, np.float64, (B, M)), a.inp(dp_l, np.float64, (B, M)),
            a.inp(n_pts, np.int32, (B,)), a.inp(obs_s, np.float64, (B, mo)), a.inp(obs_l, np.float64, (B, mo)),
            a.inp(n_obs, np.int32, (B,)), float(obs_length), float(obs_width), lop, hip, stp, a.where))
        return lo, hi, st

    def path_qp(self, q: QpParams, l_min, l_max, n_pts, start_l3):
        """ref Quadratic_planning: returns qp_l, qp_dl, qp_ddl (B,M), iters (B,), status (B,)."""
        a = self._args(l_min, l_max, start_l3)
        B, M = int(l_min.shape[0]), int(l_min.shape[1])
        outs = [a.out((B, M), np.float64) for _ in range(3)]
        it, itp = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_path_qp(
            self._h, C.byref(q), B, M, a.inp(l_min, np.float64, (B, M)), a.inp(l_max, np.float64, (B, M)),
            a.inp(n_pts, np.int32, (B,)), a.inp(start_l3, np.float64, (B, 3)), outs[0][1], outs[1][1], outs[2][1],
            itp, stp, a.where))
        return outs[0][0], outs[1][0], outs[2][0], it, st

    def smooth_line(self, sp: SmoothParams, xy, n_pts):
        """ref smooth_reference_line: xy (B,M,2) -> out (B,M,4) x,y,theta,kappa; iters, status."""
        a = self._args(xy)
        B, M = int(xy.shape[0]), int(xy.shape[1])
        out, outp = a.out((B, M, 4), np.float64)
        it, itp = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_smooth_line(self._h, C.byref(sp), B, M, a.inp(xy, np.float64, (B, M, 2)),
                                              a.inp(n_pts, np.int32, (B,)), outp, itp, stp, a.where))
        return out, it, st

    def frenet_path_to_xy(self, ref_line, s_map, n_ref, begin_sl, path_s, path_l, n_pts):
        """ref frenet_2_x_y_theta_kappa before its smoothing call: returns target_xy (B,M+1,2), n_out, status."""
        a = self._args(ref_line, path_s)
        B, P, M = int(ref_line.shape[0]), int(ref_line.shape[1]), int(path_s.shape[1])
        t, tp = a.out((B, M + 1, 2), np.float64)
        no, nop = a.out((B,), np.int32)
        st, stp = a.out((B,), np.int32)
        self._check(self._lib.emp_frenet_path_to_xy(
            self._h, B, P, M, a.inp(ref_line, np.float64, (B, P, 4)), a.inp(s_map, np.float64, (B, P)),
            a.inp(n_ref, np.int32, (B,)), a.inp(begin_sl, np.float64, (B, 2)), a.inp(path_s, np.float64, (B, M)),
            a.inp(path_l, np.float64, (B, M)), a.inp(n_pts, np.int32, (B,)), tp, nop, stp, a.where))
        return t, no, st

    # ---- whole cycle ------------------------------------------------------------------------
    def pinned_empty(self, shape, dtype=np.float64) -> np.ndarray:
        """A NumPy array on page-locked host memory of this context (emp_host_alloc): what ``plan_cycle_pinned`` moves over
        PCIe without a staging copy.  Lives until ``close()`` (or ``pinned_free``)."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) if len(tuple(shape)) else 1
        ptr = C.c_void_p()
        self._check(self._lib.emp_host_alloc(self._h, max(n * dt.itemsize, 8), C.byref(ptr)))
        buf = (C.c_char * max(n * dt.itemsize, 8)).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dt, count=n).reshape(tuple(shape))
        self._pinned[arr.__array_interface__["data"][0]] = ptr.value
        return arr

    def pinned_free(self, arr):
        """Give a ``pinned_empty`` array back (the array must not be used afterwards)."""
        key = arr.__array_interface__["data"][0]
        ptr = self._pinned.pop(key)
        self._check(self._lib.emp_host_free(self._h, C.c_void_p(ptr)))

    def host_ring(self, p: DpParams, B: int, max_ref: int, max_obs: int, max_pts=None, depth=None, max_global: int = 0) -> "HostRing":
        """``depth`` (default: the pipeline depth) slots of page-locked input and output arrays for ``B`` scenes each: the
        overlapped host path of the planning cycle (``HostRing``).  ``max_global`` > 0: slots for REQUESTS - their inputs hold
        the global path (B, max_global, 4), n_global and pre_match_index instead of a reference line, the cycle runs the front
        end itself and the outputs carry match_index and ref_status."""
        return HostRing(self, p, int(B), int(max_ref), int(max_obs), int(max_pts) if max_pts else max_path_points(p),
                        int(depth) if depth else max(int(self._lib.emp_pipeline_depth(self._h)), 1), int(max_global))

    def wait_cycle(self, calls_back: int = 0):
        """Block until the host outputs of the pinned cycle issued ``calls_back`` calls ago are in place (emp_wait_cycle)."""
        self._check(self._lib.emp_wait_cycle(self._h, int(calls_back)))

    def plan_cycle(self, p: DpParams, q: QpParams, sp: SmoothParams, ref_line, n_ref, origin_xy, start_xy, start_v,
                   start_a, obs_xy, n_obs, max_pts=None, mode=L.EMP_DP_TWO_KERNEL, dyn_dis_speed=None, slot=None,
                   out: "CycleResult" = None, global_path=None, n_global=None, pre_match_index=None) -> CycleResult:
        """ref motion_planning body, test_9.py:113-218, for a batch of scenes.  dyn_dis_speed (B,2): distance and speed
        of each scene's first dynamic obstacle (NaN = none) for the virtual obstacles of test_9.py:137-169.
        ``slot``: a ``HostRing`` slot whose page-locked arrays ARE the inputs (the array arguments are then ignored) and
        receive the outputs - with a pipeline set the call returns before they are there (``slot.wait()``).
        ``out``: the ``CycleResult`` of an earlier call with the same sizes - its arrays are written again instead of new ones
        being allocated.  With the same inputs' memory too, consecutive calls have one signature, which is what option
        "cycle_graph" (EMP_OPT_CYCLE_GRAPH: the call's launches replayed as one hipGraph) needs.
        ``global_path`` (B,G,4), ``n_global`` (B,), ``pre_match_index`` (B,): the cycle starts from the GLOBAL path - the
        reference's find_match_points / sampling / smooth_reference_line (test_9.py:99-110, ``reference_line``) run in front of it
        in the same call, the 51-point line stays on the device, ``start_xy`` is the predicted location; ``ref_line`` and ``n_ref``
        are then ignored (pass None) and the result carries ``match_index`` and ``ref_status``."""
        if slot is not None:
            return self._plan_cycle_pinned(p, q, sp, slot, mode)
        front = global_path is not None
        a = self._args(global_path if front else ref_line, origin_xy)
        a.cycle = True               # pipelined mode: the outputs become complete on the result stream (see _Args.done)
        B, P = (int(global_path.shape[0]), REF_LINE_POINTS) if front else (int(ref_line.shape[0]), int(ref_line.shape[1]))
        mo = int(obs_xy.shape[1]) if obs_xy is not None else 0
        M = int(max_pts) if max_pts else max_path_points(p)
        io = L.CycleIO()
        if front:
            G = int(global_path.shape[1])
            io.global_path = a.inp(global_path, np.float64, (B, G, 4))
            io.n_global = a.inp(n_global, np.int32, (B,))
            io.pre_match_index = a.inp(pre_match_index, np.int32, (B,))
            io.max_global = G
        else:
            io.ref_line = a.inp(ref_line, np.float64, (B, P, 4))
            io.n_ref = a.inp(n_ref, np.int32, (B,))
        io.origin_xy = a.inp(origin_xy, np.float64, (B, 2))
        io.start_xy = a.inp(start_xy, np.float64, (B, 2))
        io.start_v = a.inp(start_v, np.float64, (B, 2))
        io.start_a = a.inp(start_a, np.float64, (B, 2))
        io.obs_xy = a.inp(obs_xy, np.float64, (B, mo, 2)) if mo else None
        io.n_obs = a.inp(n_obs, np.int32, (B,)) if mo else None
        io.dyn_dis_speed = a.inp(dyn_dis_speed, np.float64, (B, 2)) if dyn_dis_speed is not None else None
        res = {}
        for name, shape, dt in (("dp_rows", (B, p.col), np.float64), ("dp_s", (B, M), np.float64),
                                ("dp_l", (B, M), np.float64), ("dp_len", (B,), np.int32),
                                ("path_s", (B, M), np.float64), ("path_l", (B, M), np.float64),
                                ("path_len", (B,), np.int32), ("traj", (B, M + 1, 4), np.float64),
                                ("traj_len", (B,), np.int32), ("status", (B,), np.int32)) + \
                ((("match_index", (B,), np.int32), ("ref_status", (B,), np.int32)) if front else ()):
            arr, ptr = a.out(shape, dt, getattr(out, name) if out is not None else None)
            res[name] = arr
            setattr(io, name, ptr)
        self._check(self._lib.emp_plan_cycle(self._h, C.byref(p), C.byref(q), C.byref(sp), B, P, mo, M, int(mode),
                                             C.byref(io), a.where))
        if self.pipelined and a.torch:
            # The outputs of the calls in flight stay referenced here even if the caller drops them at once: their
            # memory must not come back from torch's allocator into a later call's outputs while this call, or a
            # consumer queued behind it on its result stream, still uses it.  Entry k is released when call k + n
            # has been issued; the memory may then go to call k + n + 1, which runs on ANOTHER lane - so call k + n
            # (emp_plan_cycle, lane mode) first orders the main stream behind the tail of its lane, i.e. behind call k
            # and its consumers, and every later call is ordered behind the main stream.  (Staged mode: call
            # k + emp_pipeline_depth() is not issued before the back stage of call k is done.)
            self._inflight.append((list(res.values()), a.keep))
            if len(self._inflight) > self._retain:
                self._inflight.pop(0)
        return CycleResult(**res)

    def _plan_cycle_pinned(self, p, q, sp, slot, mode):
        io = L.CycleIO()
        front = getattr(slot, "max_global", 0) > 0           # a slot made for requests: the cycle starts from the global path
        names = ("global_path", "n_global", "pre_match_index") if front else ("ref_line", "n_ref")
        if front:
            io.max_global = slot.max_global
        for name in names + ("origin_xy", "start_xy", "start_v", "start_a"):
            setattr(io, name, C.c_void_p(slot.inputs[name].ctypes.data))
        mo = slot.max_obs
        io.obs_xy = C.c_void_p(slot.inputs["obs_xy"].ctypes.data) if mo else None
        io.n_obs = C.c_void_p(slot.inputs["n_obs"].ctypes.data) if mo else None
        io.dyn_dis_speed = C.c_void_p(slot.inputs["dyn_dis_speed"].ctypes.data) if slot.use_dyn else None
        for name, arr in slot.outputs.items():
            setattr(io, name, C.c_void_p(arr.ctypes.data))
        self._check(self._lib.emp_plan_cycle(self._h, C.byref(p), C.byref(q), C.byref(sp), slot.B, slot.max_ref, mo, slot.max_pts,
                                             int(mode), C.byref(io), L.EMP_HOST_PINNED))
        slot._ticket = int(self._lib.emp_cycle_ticket(self._h)) if self.pipelined else None
        return CycleResult(**slot.outputs)

    def pack_records(self, res: "CycleResult", col: int, max_pts: int, path_cap=None, fields: str = "full"):
        """One fixed-stride float64 record per scene from a cycle's outputs on the device, in ONE launch
        (emp_pack_records / emp_pack_trajectory_records; layouts of ``emplanner_carla_amd.dist.record_width``).  The
        launch goes to the stream on which the cycle's outputs become complete (``torch_result_stream()``); a caller on
        another stream is ordered with it.  ``fields``: "full" or "trajectory" (status, traj_len, trajectory only)."""
        if fields not in ("full", "trajectory"):
            raise ValueError("fields must be 'full' or 'trajectory'")
        cap = int(max_pts) if path_cap is None else min(int(path_cap), int(max_pts))
        B = int(res.status.shape[0])
        full = fields == "full"
        width = (3 + int(col) + 2 * cap if full else 2) + 4 * (cap + 1)
        shapes = [(res.status, np.int32, (B,)), (res.traj_len, np.int32, (B,))]
        if full:
            shapes += [(res.path_len, np.int32, (B,)), (res.dp_rows, np.float64, (B, int(col))),
                       (res.path_s, np.float64, (B, int(max_pts))), (res.path_l, np.float64, (B, int(max_pts)))]
        shapes.append((res.traj, np.float64, (B, int(max_pts) + 1, 4)))

        def call(ptrs, rp, on_rs, where):
            if full:
                return self._lib.emp_pack_records(self._h, B, int(col), int(max_pts), cap, *ptrs, rp, on_rs, where)
            return self._lib.emp_pack_trajectory_records(self._h, B, int(max_pts), cap, *ptrs, rp, on_rs, where)

        if not _is_torch(res.status):                       # host arrays: staged through the library like any other call
            a = self._args(res.status)
            ptrs = [a.inp(x, dt, shp) for x, dt, shp in shapes]
            rec, rp = a.out((B, width), np.float64)
            self._check(call(ptrs, rp, 0, a.where))
            return rec
        import torch
        dev = res.status.device
        cur = torch.cuda.current_stream(dev)
        target = self.torch_result_stream()
        foreign = int(cur.cuda_stream) != int(target.cuda_stream)
        if foreign:
            target.wait_stream(cur)
        rec = torch.empty((B, width), dtype=torch.float64, device=dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        arrs = [x for x, _, _ in shapes]
        for t_ in arrs:
            if not (t_.is_cuda and t_.is_contiguous()):
                raise ValueError("pack_records takes the contiguous device tensors plan_cycle returned")
        self._cur = None
        self._check(call([ptr(t_) for t_ in arrs], ptr(rec), 1 if self.pipelined else 0, L.EMP_DEVICE))
        if foreign:
            cur.wait_stream(target)
        return rec

    # ---- scalar utilities -------------------------------------------------------------------
    def quintic_coefficients(self, bc):
        """ref cal_quintic_coefficient: bc (n,8) -> coeff (n,6) in the absolute-s basis."""
        a = self._args(bc)
        n = int(bc.shape[0])
        c, cp = a.out((n, 6), np.float64)
        self._check(self._lib.emp_quintic_coefficients(self._h, n, a.inp(bc, np.float64, (n, 8)), cp, a.where))
        return c

    def obs_cost(self, square_d, w_collision, danger_dis=4, safe_dis=6):
        """ref cal_obs_cost: square_d (n, samples) -> cost (n,); 10 samples per lattice edge in the DP, any count here."""
        a = self._args(square_d)
        n, m = int(square_d.shape[0]), int(square_d.shape[1])
        c, cp = a.out((n,), np.float64)
        self._check(self._lib.emp_obs_cost_n(self._h, n, m, float(w_collision), float(danger_dis), float(safe_dis),
                                             a.inp(square_d, np.float64, (n, m)), cp, a.where))
        return c

    def free_edge_costs(self, edges, obs_s, obs_l, n_obs, w_collision=1e12, w_smooth=(300.0, 1000.0, 5000.0), w_ref=20.0):
        """ref cal_start_cost / cal_neighbor_cost for free edges: edges (n, 8) = start s, l, dl, ddl, span, end l, sample_s, 0;
        obstacles per edge obs_s, obs_l (n, max_obs), n_obs (n,) -> cost (n,)."""
        a = self._args(edges)
        n = int(edges.shape[0])
        mo = int(obs_s.shape[1]) if obs_s is not None else 0
        c, cp = a.out((n,), np.float64)
        w3 = (C.c_double * 3)(*[float(v) for v in w_smooth])
        self._check(self._lib.emp_free_edge_costs(
            self._h, n, mo, a.inp(edges, np.float64, (n, 8)), a.inp(obs_s, np.float64, (n, mo)) if mo else None,
            a.inp(obs_l, np.float64, (n, mo)) if mo else None, a.inp(n_obs, np.int32, (n,)) if mo else None,
            float(w_collision), w3, float(w_ref), cp, a.where))
        return c


Planner.set_option.__doc__ += "\n        Options: " + ", ".join(f'"{k}"' for k in L.OPTIONS) + "."


def st_grid():
    """The reference's hard-coded S-T samples (speed_planning_test.py:114,116): s_list (40,), t_list (16,)."""
    s_list = np.concatenate((np.arange(0, 5, 0.5), np.arange(5.5, 15, 1), np.arange(16, 30, 1.5), np.arange(32, 55, 2.5)))
    return s_list, np.arange(0.5, 8.5, 0.5)


def max_path_points(p: DpParams) -> int:
    """Upper bound of len(enrich_DP_s_l output).  Per lattice segment the reference takes
    len(arange(0, int(end_s - start_s), res)) samples (path_planning.py:405/:423); the float difference is
    sample_s up to an ulp, so int() of it is at most floor(sample_s) (+1 ulp guard)."""
    per = int(np.ceil(int(p.sample_s + 1e-9) / p.sampling_res))
    return p.col * max(per, 1) + 1


def tile_edges(edge_canonical: np.ndarray, row: int) -> np.ndarray:
    """Host helper: canonical (B, col-1, row_i, row_k) -> the tiled layout of include/emplanner.h."""
    B, ncol = edge_canonical.shape[0], edge_canonical.shape[1]
    S = 64 // row
    tiles = (B + S - 1) // S
    out = np.zeros((tiles, ncol, row, 64))
    for b in range(B):
        t, s = divmod(b, S)
        # out[t, j, k, s*row + i] = e[b, j, i, k]
        out[t, :, :, s * row:(s + 1) * row] = np.transpose(edge_canonical[b], (0, 2, 1))
    return out.reshape(-1)
