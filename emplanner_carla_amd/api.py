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
from ._lib import DpParams, QpParams, SmoothParams, EmpError


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


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "device")


class _Args:
    """Collects array arguments of one call, checks they live in one memory space, makes outputs."""

    def __init__(self, *inputs):
        self.torch = any(_is_torch(x) for x in inputs if x is not None)
        self.keep = []
        if self.torch:
            import torch
            self.t = torch
            self.device = next(x.device for x in inputs if _is_torch(x))
            if self.device.type != "cuda":
                raise ValueError("torch tensors passed to the planner must live on the GPU")
        self.where = L.EMP_DEVICE if self.torch else L.EMP_HOST

    def inp(self, x, dtype, shape=None):
        if x is None:
            return None
        if self.torch:
            if not _is_torch(x):
                raise ValueError("mixing torch device tensors and host arrays in one call is not supported")
            td = {np.float64: self.t.float64, np.int32: self.t.int32}[dtype]
            if x.dtype != td or not x.is_contiguous():
                x = x.to(td).contiguous()
            if shape is not None and tuple(x.shape) != tuple(shape):
                raise ValueError(f"expected shape {tuple(shape)}, got {tuple(x.shape)}")
            self.keep.append(x)
            return C.c_void_p(x.data_ptr())
        a = np.ascontiguousarray(x, dtype=dtype)
        if shape is not None and a.shape != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
        self.keep.append(a)
        return C.c_void_p(a.ctypes.data)

    def out(self, shape, dtype):
        if self.torch:
            td = {np.float64: self.t.float64, np.int32: self.t.int32}[dtype]
            a = self.t.zeros(tuple(shape), dtype=td, device=self.device)
            return a, C.c_void_p(a.data_ptr())
        a = np.zeros(tuple(shape), dtype=dtype)
        return a, C.c_void_p(a.ctypes.data)


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
        if rc != 0:
            msg = self._lib.emp_last_error(self._h)
            raise EmpError(f"libemplanner call failed ({rc}): {msg.decode() if msg else ''}")

    # ---- housekeeping ------------------------------------------------------------------
    def synchronize(self):
        self._check(self._lib.emp_synchronize(self._h))

    def set_timing(self, enabled: bool):
        self._check(self._lib.emp_set_timing(self._h, int(bool(enabled))))

    def kernel_ms(self, name: str) -> float:
        return float(self._lib.emp_kernel_ms(self._h, name.encode()))

    @property
    def stream(self):
        return self._lib.emp_stream(self._h)

    # ---- DP ------------------------------------------------------------------------------
    def edge_tensor_elems(self, p: DpParams, B: int, layout=L.EMP_EDGE_CANONICAL) -> int:
        return int(self._lib.emp_edge_tensor_elems(C.byref(p), int(B), int(layout)))

    def dp_edge_costs(self, p: DpParams, obs_s, obs_l, n_obs, start, layout=L.EMP_EDGE_CANONICAL):
        """ref cal_start_cost / cal_neighbor_cost for all lattice edges.
        returns start_cost (B,row) and edge: canonical (B, col-1, row_i, row_k) or tiled (flat)."""
        a = _Args(obs_s, obs_l, n_obs, start)
        B = int(start.shape[0])
        mo = int(obs_s.shape[1]) if obs_s is not None and len(obs_s.shape) == 2 else 0
        c0, c0p = a.out((B, p.row), np.float64)
        n = self.edge_tensor_elems(p, B, layout)
        e, ep = a.out((n,), np.float64)
        self._check(self._lib.emp_dp_edge_costs(
            self._h, C.byref(p), B, mo, a.inp(obs_s, np.float64, (B, mo)), a.inp(obs_l, np.float64, (B, mo)),
            a.inp(n_obs, np.int32, (B,)), a.inp(start, np.float64, (B, 4)), c0p, ep, int(layout), a.where))
        if layout == L.EMP_EDGE_CANONICAL:
            e = e.reshape(B, p.col - 1, p.row, p.row)
        return c0, e

    def dp_plan(self, p: DpParams, obs_s, obs_l, n_obs, start, mode=L.EMP_DP_FUSED):
        """ref DP_algorithm up to the backtrack: returns rows (B,col) f64, min_cost (B,), status (B,)."""
        a = _Args(obs_s, obs_l, n_obs, start)
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
        a = _Args(start_cost, edge_tiled)
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
        a = _Args(rows, start)
        B = int(start.shape[0])
        ps, psp = a.out((B, max_pts), np.float64)
        pl, plp = a.out((B, max_pts), np.float64)
        ln, lnp = a.out((B,), np.int32)
        st, sp = a.out((B,), np.int32)
        self._check(self._lib.emp_dp_enrich(
            self._h, C.byref(p), B, a.inp(rows, np.float64, (B, p.col)), a.inp(start, np.float64, (B, 4)),
            int(max_pts), psp, plp, lnp, sp, a.where))
        return ps, pl, ln, st


def max_path_points(p: DpParams) -> int:
    """Upper bound of len(enrich_DP_s_l output): col * ceil(int(sample_s + 1) / res) + 1."""
    per = int(np.ceil((int(p.sample_s) + 1) / p.sampling_res))
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
