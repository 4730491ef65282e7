"""Scene / vehicle independence of every batched entry point beside the planning cycle (which
tests/test_gpu_fullsize.py covers): a permuted batch must give bit-identical per-item results.

Several kernels put more than one item into a wavefront (5 vehicles per wavefront in the MPC kernel, one vehicle
per lane in the LQR kernel, 7 scenes per wavefront in the DP sweep, x and y of one line in the two halves of a
wavefront in the smoother) and exchange data with lane shifts and cross-lane reductions; nothing of one item may
reach another, whatever its neighbours are - including neighbours whose own problem fails.
"""
from __future__ import annotations

import numpy as np
import pytest

from emplanner_carla_amd import scenes as S
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pl():
    from emplanner_carla_amd.api import Planner
    p = Planner(0)
    yield p
    p.close()


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    if a.dtype.kind == "f":
        assert np.array_equal(a, b, equal_nan=True), f"{what}: {np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))[:4].tolist()}"
    else:
        assert np.array_equal(a, b), what


def _paths(rng, B, M, amp):
    path = np.zeros((B, M, 4))
    n = rng.integers(6, M + 1, B).astype(np.int32)
    state = np.zeros((B, 5))
    for b in range(B):
        t = np.arange(n[b]) * 2.4
        xy = np.stack([t, amp * np.sin(t / 40.0 + rng.uniform(0, 3))], axis=1)
        th = np.arctan2(np.gradient(xy[:, 1]), np.gradient(xy[:, 0]))
        ka = np.gradient(th) / np.hypot(np.gradient(xy[:, 0]), np.gradient(xy[:, 1]))
        path[b, :n[b]] = np.column_stack([xy, th, ka])
        at = int(rng.integers(0, n[b] - 2))
        # every fifth vehicle is far off the path (saturated controls, large errors)
        off = 6.0 if b % 5 == 0 else 0.5
        state[b] = [xy[at, 0] + rng.normal(0, off), xy[at, 1] + rng.normal(0, off), th[at] + rng.normal(0, 0.1 * off),
                    rng.normal(0, 0.3), rng.normal(0, 0.1)]
    return path, n, state


def test_mpc_vehicles_are_independent(pl):
    from emplanner_carla_amd.api import mpc_params
    g = load_golden("mpc.npz")
    rng = np.random.default_rng(21)
    B = 1003                                                # not a multiple of 5: a partly filled wavefront
    path, n, state = _paths(rng, B, 48, 10.0)
    n[::17] = 0                                             # empty paths (refused) next to healthy vehicles
    vx = rng.choice([0.005, 3.0, 9.0, 18.0], B)
    mi = np.zeros(B, np.int32)
    p = mpc_params(vehicle_para=tuple(g["vehicle_para"]))
    r = pl.mpc_lateral(p, path, n, state, vx, mi)
    assert (r.status[n > 0] == 0).all() and (r.status[n == 0] != 0).all()
    perm = rng.permutation(B)
    rp = pl.mpc_lateral(p, path[perm], n[perm], state[perm], vx[perm], mi[perm])
    ok = (n > 0)[perm]
    _same(r.status[perm], rp.status, "status")
    for name in ("steer", "u", "e_rr", "k_r", "min_index", "pre_pro"):
        _same(getattr(r, name)[perm][ok], getattr(rp, name)[ok], name)
    # and one vehicle alone
    for b in (1, 5, 998):
        r1 = pl.mpc_lateral(p, path[b:b + 1], n[b:b + 1], state[b:b + 1], vx[b:b + 1], mi[b:b + 1])
        _same(r1.u[0], r.u[b], f"vehicle {b} alone")


def test_lqr_vehicles_are_independent(pl):
    from emplanner_carla_amd.api import lqr_params
    g = load_golden("mpc.npz")
    rng = np.random.default_rng(22)
    B = 517
    path, n, state = _paths(rng, B, 40, 8.0)
    vx = rng.choice([0.02, 1.0, 4.0, 11.0, 22.0], B)       # 0.02 m/s: the Riccati iteration runs to its cap
    mi = np.zeros(B, np.int32)
    p = lqr_params(vehicle_para=tuple(g["vehicle_para"]))
    r = pl.lqr_lateral(p, path, n, state, vx, mi)
    perm = rng.permutation(B)
    rp = pl.lqr_lateral(p, path[perm], n[perm], state[perm], vx[perm], mi[perm])
    for name in ("steer", "min_index", "sweeps", "status"):
        _same(getattr(r, name)[perm], getattr(rp, name), name)


def test_speed_dp_scenes_are_independent(pl):
    from emplanner_carla_amd.api import speed_dp_params
    o = S.make_dynamic_batch(range(700, 764))
    B = len(o[4])
    sets = pl.st_graph(*o[:4])
    res = pl.speed_dp(speed_dp_params(), *sets, o[4])
    perm = np.random.default_rng(23).permutation(B)
    setp = pl.st_graph(*[a[perm] for a in o[:4]])
    for i in range(4):
        _same(sets[i][perm], setp[i], f"st_graph output {i}")
    resp = pl.speed_dp(speed_dp_params(), *setp, o[4][perm])
    for name in ("cost", "node", "s_dot", "end_node", "speed_s", "speed_t"):
        _same(getattr(res, name)[perm], getattr(resp, name), name)


def test_reference_lines_and_smoothing_are_independent(pl):
    from emplanner_carla_amd.api import smooth_params
    rng = np.random.default_rng(24)
    B, G = 257, 160
    gp = np.zeros((B, G, 4))
    n_global = np.full(B, G, np.int32)
    n_global[::13] = 40                                    # too short for the 51-node window: refused
    at = rng.integers(2, 100, B)
    for b in range(B):
        t = np.arange(G) * 2.0
        y = 12 * np.sin(t / 45.0 + rng.uniform(0, 3)) + rng.normal(0, 0.08, G)
        th = np.arctan2(np.gradient(y), np.gradient(t))
        gp[b] = np.column_stack([t, y, th, np.gradient(th) / 2.0])
    at = np.minimum(at, n_global - 3)
    pred = gp[np.arange(B), at, :2] + rng.normal(0, 0.4, (B, 2))
    pre = np.maximum(at - 2, 0).astype(np.int32)
    sp = smooth_params()
    out = pl.reference_line(sp, gp, n_global, pred, pre)
    assert (out[4][n_global >= 51] == 0).all() and (out[4][n_global < 51] != 0).all()
    perm = rng.permutation(B)
    outp = pl.reference_line(sp, gp[perm], n_global[perm], pred[perm], pre[perm])
    ok = (out[4] == 0)[perm]
    _same(out[4][perm], outp[4], "status")
    _same(out[2][perm], outp[2], "match index")
    _same(out[0][perm][ok], outp[0][ok], "reference line")
    # the stand-alone smoother with ragged lengths (register-only solver for <= 32 points, pair solver up to 64,
    # LDS solver beyond): every line beside lines of the other kinds
    M = 90
    n_pts = rng.choice([3, 8, 23, 32, 33, 51, 64, 65, 90], B).astype(np.int32)
    xy = np.zeros((B, M, 2))
    for b in range(B):
        t = np.arange(n_pts[b]) * 2.0
        xy[b, :n_pts[b]] = np.column_stack([t + rng.normal(0, 0.1, n_pts[b]),
                                            9 * np.sin(t / 30.0) + rng.normal(0, 0.1, n_pts[b])])
    sm, it, st = pl.smooth_line(sp, xy, n_pts)
    assert (st == 0).all()
    smp, itp, stp = pl.smooth_line(sp, xy[perm], n_pts[perm])
    _same(it[perm], itp, "smoothing iterations")
    for b in range(B):
        k = n_pts[perm][b]
        _same(sm[perm][b, :k], smp[b, :k], f"smoothed line {b}")


def test_dp_scenes_are_independent_on_the_wide_lattice(pl):
    """config 5's 120x21 lattice: three scenes per wavefront in the sweep."""
    from emplanner_carla_amd.api import dp_params_from_cfg
    cfg = S.CFG5
    b = S.make_batch(range(40, 62), cfg)                  # 22 scenes: a partly filled last tile
    p = dp_params_from_cfg(cfg)
    rows, mc, st = pl.dp_plan(p, b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)
    perm = np.random.default_rng(25).permutation(len(rows))
    rows2, mc2, st2 = pl.dp_plan(p, b.sl_obs_s[perm], b.sl_obs_l[perm], b.n_obs[perm], b.sl_start[perm])
    _same(rows[perm], rows2, "rows")
    _same(mc[perm], mc2, "min cost")
    _same(st[perm], st2, "status")


def test_speed_back_end_scenes_are_independent(pl):
    """Convex space (one scene per lane), speed QP (two scenes per wavefront, healthy next to infeasible ones),
    densification and merge: a permuted batch is bit-identical per scene."""
    from emplanner_carla_amd.api import speed_qp_params
    g = load_golden("speed_backend.npz")
    B = len(g["v0"])
    rep = 5                                               # 480 scenes: several blocks of the one-lane-per-scene kernel
    tile = lambda a: np.concatenate([a] * rep)
    ins = {k: tile(g[k]) for k in ("dp_s", "dp_t", "path_index2s", "path_kappa", "s_in", "s_out", "t_in", "t_out", "v0",
                                   "qp_a0", "merge_now", "merge_path_s", "merge_x", "merge_y", "merge_heading",
                                   "merge_kappa")}
    ins["path_len"] = tile(g["path_len"]).astype(np.int32)
    ins["v0"] = ins["v0"] + np.repeat(np.arange(rep), B) * 0.37          # not the same problem five times
    n_init = np.full(B * rep, g["merge_x"].shape[1], np.int32)

    def run(idx):
        x = {k: v[idx] for k, v in ins.items()}
        cs = pl.speed_convex_space(x["dp_s"], x["dp_t"], x["path_index2s"], x["path_kappa"], x["path_len"], x["s_in"],
                                   x["s_out"], x["t_in"], x["t_out"])
        q = pl.speed_qp(speed_qp_params(), x["v0"], x["qp_a0"], x["dp_s"], x["dp_t"], *cs[:4])
        d = pl.speed_increase_points(*q[:4])
        m = pl.path_speed_merge(*d[:4], x["merge_now"], x["merge_path_s"], x["merge_x"], x["merge_y"], x["merge_heading"],
                                x["merge_kappa"], n_init[idx])
        return list(cs) + list(q) + list(d) + list(m)

    base = run(np.arange(B * rep))
    st_qp = base[10]
    assert (st_qp == 0).sum() > 100 and (st_qp == 8).sum() > 5 and (st_qp == 4).sum() > 50
    perm = np.random.default_rng(31).permutation(B * rep)
    other = run(perm)
    for k, (a, b) in enumerate(zip(base, other)):
        _same(a[perm], b, f"output {k}")


def test_garbage_scenes_neither_hang_nor_touch_their_neighbours(pl):
    """NaN / Inf coordinates, absurd counts and degenerate reference lines in a few scenes of a batch: the call
    returns (every loop of every kernel is bounded), those scenes come back refused or with whatever the arithmetic
    yields, and every other scene is bit-identical to the clean batch."""
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    cfg = S.CFG2
    b = S.make_batch(range(300, 364), cfg)
    B, P = b.ref.shape[:2]
    clean = dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v,
                 start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
    p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
    want = pl.plan_cycle(p, q, sp, **clean)
    bad = {k: v.copy() for k, v in clean.items()}
    bad["ref_line"][3, 10, 0] = np.nan                       # a NaN in the reference line
    bad["ref_line"][5, :, :2] = 0.0                          # every point the same: zero-length line
    bad["obs_xy"][7, 2] = [np.inf, -np.inf]                  # an obstacle at infinity
    bad["start_v"][9] = [np.nan, np.nan]
    bad["n_ref"][11] = 1                                     # one-point line
    bad["n_ref"][13] = 100000                                # a count beyond the row
    bad["n_obs"][15] = 1 << 30                               # a count beyond the row
    bad["n_obs"][17] = -5
    bad["start_xy"][19] = [1e300, -1e300]
    bad["n_ref"][21] = 0
    touched = [3, 5, 7, 9, 11, 13, 15, 17, 19, 21]
    got = pl.plan_cycle(p, q, sp, **bad)
    pl.synchronize()
    keep = np.setdiff1d(np.arange(B), touched)
    for name in ("dp_rows", "traj", "traj_len", "status", "path_l"):
        _same(getattr(want, name)[keep], getattr(got, name)[keep], name)
    assert (want.status[keep] == 0).sum() > 30
