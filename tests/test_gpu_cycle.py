"""GPU parity: Cartesian<->Frenet, QP stages and the whole planning cycle through the C-ABI.

Reference side = golden vectors of the imported reference (tests/golden), tolerance 1e-6 relative with
the magnitude floors written at each check (the north-star tolerance).  QP outputs are additionally
certified against the reference's own dense formulation (KKT certificate of oracle/qp_dense.py),
because the reference's solver (cvxopt) is absent and unpinned.
"""
import numpy as np
import pytest

from emplanner_carla_amd import scenes as S
from oracle import qp_dense
from oracle import ref_port as op
from tests.conftest import assert_dp_l_vs_reference, assert_rel, load_golden, rel_close

pytestmark = pytest.mark.gpu
RTOL = 1e-6

GOLD = {"cfg1": (S.CFG1, "cycle_cfg1_20x5_0obs.npz", {}),
        "default": (S.CFG_DEFAULT, "cycle_default_6x12_3obs.npz", {}),
        "cfg2": (S.CFG2, "cycle_cfg2_40x9_8obs.npz", {}),
        # SURVEY 8(d)'s geometry (arc radii 150-1000 m, survey layout on odd seeds, starts off the nodes on every other pair)
        "cfg2_tight": (S.CFG2, "cycle_cfg2_40x9_8obs_tight.npz", {}),
        # the first 32 scenes of the benchmark batch (scenes.BENCH_START_AHEAD)
        "cfg2_bench": (S.CFG2, "cycle_cfg2_40x9_8obs_bench.npz", {}),
        "cfg2_worst": (S.CFG2, "cycle_cfg2_40x9_8obs_worst.npz", {}),
        "default_t7": (S.CFG_DEFAULT, "cycle_default_6x12_3obs_t7.npz", dict(decimate=1, midpoint=0)),
        "default_t6": (S.CFG_DEFAULT, "cycle_default_6x12_3obs_t6.npz", dict(decimate=1, midpoint=0, use_qp=0))}


@pytest.fixture(scope="module")
def planner():
    from tests.conftest import make_planner
    p = make_planner(0)
    yield p
    p.close()


def _inputs(g):
    B, P = g["in_ref"].shape[:2]
    return dict(ref_line=g["in_ref"], n_ref=np.full(B, P, np.int32), origin_xy=g["in_origin_xy"],
                start_xy=g["in_start_xy"], start_v=g["in_start_v"], start_a=g["in_start_a"],
                obs_xy=g["in_obs_xy"], n_obs=g["in_n_obs"].astype(np.int32))


@pytest.mark.parametrize("key", ["cfg1", "default", "cfg2", "cfg2_tight", "cfg2_bench", "cfg2_worst"])
def test_frenet_project_vs_reference(planner, key):
    cfg, fname, _ = GOLD[key]
    g = load_golden(fname)
    i = _inputs(g)
    sm, os_, ol_, bsl, start = planner.frenet_project(**i)
    assert_rel(sm, g["s_map"], RTOL, "s_map")
    for b in range(len(sm)):
        k = int(i["n_obs"][b])
        assert_rel(os_[b, :k], g["obs_s"][b, :k], RTOL, "obs_s")
        assert_rel(ol_[b, :k], g["obs_l"][b, :k], RTOL, "obs_l")
    assert_rel(bsl, g["begin"], RTOL, "begin s,l")
    assert_rel(start[:, :2], g["start"][:, :2], RTOL, "start s,l")
    assert_rel(start[:, 2], g["start"][:, 2], RTOL, "start dl/ds")        # |dl| <= 0.05: floor 0.1
    assert_rel(start[:, 3], g["start"][:, 3], RTOL, "start d2l/ds2")


def _near_tie_line(first):
    """A 51-node line with two nodes 5 m from the point (0, 0) - A = (-4, 3) at exactly 25 m^2, B = (4, 3 + a few ulp) a last bit
    above 25 m^2 whose ROOT rounds to 5.0 too - each the closest node of a straight stretch tangent to that circle, every other
    node at 29 m^2 or more.  `first` = which of the two comes first along the line (node 10; the other is node 35)."""
    ulp = np.spacing(3.0)
    for k in range(1, 16):
        yb = 3.0 + k * ulp
        d2 = 4.0 * 4.0 + yb * yb
        if d2 > 25.0 and np.sqrt(d2) == 5.0:
            break
    else:
        raise AssertionError("no near tie found")
    A, B = np.array([-4.0, 3.0]), np.array([4.0, yb])
    tA, tB = np.array([0.6, 0.8]), np.array([-0.6, 0.8])          # unit tangents of the circle of radius 5 at A and at B
    (p0, t0), (p1, t1) = ((A, tA), (B, tB)) if first == "A" else ((B, tB), (A, tA))
    nodes = [p0 + 2.0 * (i - 10) * t0 for i in range(21)]
    nodes[10] = p0.copy()
    far = [np.array(v, dtype=np.float64) for v in ((40.0, 40.0), (60.0, 10.0), (60.0, -40.0), (0.0, -60.0))]
    nodes += far
    tail = [p1 + 2.0 * (i - 35) * t1 for i in range(25, 51)]
    tail[10] = p1.copy()
    nodes += tail
    xy = np.array(nodes)
    assert xy.shape == (51, 2)
    d2_all = xy[:, 0] ** 2 + xy[:, 1] ** 2
    assert np.sum(d2_all < 28.0) == 2 and d2_all[10] != d2_all[35] and np.sqrt(d2_all[10]) == np.sqrt(d2_all[35]) == 5.0
    th = np.arctan2(np.gradient(xy[:, 1]), np.gradient(xy[:, 0]))
    return np.concatenate([xy, th[:, None], np.zeros((51, 1))], axis=1)


@pytest.mark.parametrize("first", ["A", "B"])
def test_projection_scan_where_two_squared_distances_share_a_root(planner, first):
    """The projection kernel's nearest-node scans compare SQUARED distances and are redone on the distances when two squares are
    within their last bits of each other (emp_tail_kernels.h, match_scan_wave64): here two nodes of the line are 5.0 m from the
    point after rounding while their squares differ by one unit in the last place - the reference's strict `<` on the distances
    (planning_utils.py:390-396) keeps the FIRST of them, the squares alone would pick the smaller.  Origin, planning start and
    obstacles all sit on such points; s differs by tens of metres between the two nodes."""
    line = _near_tie_line(first)
    pt = (0.0, 0.0)
    obs = [pt, (0.5, 9.0), pt]
    sm, os_, ol_, bsl, start = planner.frenet_project(ref_line=line[None], n_ref=np.array([51], np.int32), origin_xy=np.array([pt]),
                                                      start_xy=np.array([pt]), start_v=np.array([[5.0, 1.0]]),
                                                      start_a=np.array([[0.1, 0.0]]), obs_xy=np.array([obs]), n_obs=np.array([3], np.int32))
    nodes = [tuple(r) for r in line]
    want_map = op.cal_s_map_fun(nodes, pt)
    assert abs(want_map[10]) < 1.0 and abs(want_map[35]) > 30.0          # node 10 is the first of the pair: the s axis starts there
    assert_rel(sm[0], np.asarray(want_map), RTOL, "s_map")
    want_s, want_l = op.cal_s_l_fun(obs, nodes, want_map)
    assert_rel(os_[0, :3], np.asarray(want_s), RTOL, "obs_s")
    assert_rel(ol_[0, :3], np.asarray(want_l), RTOL, "obs_l")
    bs, bl = op.cal_s_l_fun([pt], nodes, want_map)
    assert_rel(bsl[0], np.array([bs[0], bl[0]]), RTOL, "begin s, l")


def test_match_and_heading_functions(planner):
    g = load_golden("functions.npz")
    path = g["mp_path"][None]
    n_ref = np.array([path.shape[1]], np.int32)
    pts = g["mp_pts"][None]
    mi, pr = planner.match_projection(path, n_ref, pts, np.array([pts.shape[1]], np.int32))
    assert np.array_equal(mi[0], g["mp_index"])
    assert_rel(pr[0], g["mp_proj"], RTOL, "projection")
    for mode, out in zip(g["fm_modes"], g["fm_out"]):
        mi, pr = planner.find_match_points(path, n_ref, pts[:, :3], np.array([3], np.int32),
                                           np.array([int(mode[0])], np.int32), np.array([int(mode[1])], np.int32))
        assert np.array_equal(mi[0].astype(np.float64), out[:3])
        assert_rel(pr[0].reshape(-1), out[3:], RTOL, "find_match_points projection")
    th, kp = planner.heading_kappa(g["hk_xy"][None], np.array([len(g["hk_xy"])], np.int32))
    assert_rel(th[0], g["hk_theta"], RTOL, "theta")
    assert_rel(kp[0], g["hk_kappa"], RTOL, "kappa")
    # ragged batch: two polylines of different length in one call
    xy = np.zeros((2, 40, 2))
    xy[0, :37] = g["hk_xy"]
    xy[1, :20] = g["hk_xy"][:20]
    th, kp = planner.heading_kappa(xy, np.array([37, 20], np.int32))
    t20, k20 = op.cal_heading_kappa([tuple(p) for p in g["hk_xy"][:20]])
    assert_rel(th[0, :37], g["hk_theta"], RTOL)
    assert_rel(th[1, :20], t20, RTOL)
    assert_rel(kp[1, :20], k20, RTOL)


def test_scalar_utilities(planner):
    g = load_golden("functions.npz")
    c = planner.quintic_coefficients(g["quintic_bc"])
    for b, coef, v in zip(g["quintic_bc"], c, g["quintic_vals"]):
        ts = np.linspace(b[6], b[7], 11)
        # Compare the polynomial on its segment, not raw coefficients.  In the absolute-s basis the
        # terms c_k s^k reach ~1e9 and cancel to O(1): the golden values (reference coefficients evaluated
        # in float64) carry a rounding error of a few eps * sum|c_k s^k| by construction.  We evaluate OUR
        # coefficients in extended precision and allow exactly that error bound on top of 1e-6 relative.
        tl_ = ts.astype(np.longdouble)
        got = sum(np.longdouble(coef[k]) * tl_ ** k for k in range(6)).astype(np.float64)
        mag = sum(np.abs(coef[k]) * np.abs(ts) ** k for k in range(6))
        err = np.abs(got - v)
        assert (err <= RTOL * np.maximum(np.abs(v), 1.0) + 16 * np.finfo(float).eps * mag).all(), \
            f"quintic on segment: {err.max():.3e}"
    assert np.array_equal(planner.obs_cost(g["obs_sq"], 1e12), g["obs_cost"])
    assert np.array_equal(planner.obs_cost(g["obs_sq"], 7.5, danger_dis=3, safe_dis=5), g["obs_cost_w3"])


@pytest.mark.parametrize("key", ["cfg2", "cfg2_tight", "cfg2_bench", "cfg2_worst"])
def test_lmin_lmax_and_index_error(planner, key):
    cfg, fname, _ = GOLD[key]
    g = load_golden(fname)
    B = len(g["seeds"])
    nq = g["n_qp"].astype(np.int32)
    M = 24
    dps = np.zeros((B, M))
    dpl = np.zeros((B, M))
    for b in range(B):
        n = int(g["dp_len"][b])
        dps[b, :nq[b]] = g["dp_s"][b, :n][::2]
        dpl[b, :nq[b]] = g["dp_l"][b, :n][::2]
    lo, hi, st = planner.lmin_lmax(dps, dpl, nq, np.nan_to_num(g["obs_s"]), np.nan_to_num(g["obs_l"]),
                                   g["in_n_obs"].astype(np.int32), 5, 5)
    for b in range(B):
        if g["status"][b] == 3:       # the reference raised IndexError (path_planning.py:267 / :272) on this scene
            assert st[b] == 4
            continue
        assert st[b] == 0
        assert np.array_equal(lo[b, :nq[b]], g["l_min"][b, :nq[b]])
        assert np.array_equal(hi[b, :nq[b]], g["l_max"][b, :nq[b]])
    # an obstacle mapped within two stations of the path end: the reference raises IndexError (:267/:272)
    obs_s = np.array([[dps[0, nq[0] - 1] - 1.0]])
    with pytest.raises(IndexError):
        op.cal_lmin_lmax(list(dps[0, :nq[0]]), list(dpl[0, :nq[0]]), [obs_s[0, 0]], [3.0], 5, 5)
    lo, hi, st = planner.lmin_lmax(dps[:1], dpl[:1], nq[:1], obs_s, np.array([[3.0]]), np.array([1], np.int32), 5, 5)
    assert st[0] == 4


@pytest.mark.parametrize("key", ["cfg2", "cfg2_tight", "cfg2_bench", "cfg2_worst", "default", "default_t7"])
def test_path_qp_vs_reference_formulation(planner, key):
    from emplanner_carla_amd.api import qp_params
    cfg, fname, mode = GOLD[key]
    g = load_golden(fname)
    B = len(g["seeds"])
    ok = ~np.isnan(g["l_min"][:, 0])
    nq = g["n_qp"].astype(np.int32)
    M = int(nq.max())
    lo = np.nan_to_num(g["l_min"][:, :M])
    hi = np.nan_to_num(g["l_max"][:, :M])
    l, dl, ddl, iters, st = planner.path_qp(qp_params(), lo, hi, nq, g["start"][:, 1:].copy())
    n_checked = 0
    for b in np.nonzero(ok)[0]:
        n = nq[b]
        if g["status"][b] == 4:
            assert st[b] == 8, "reference formulation infeasible -> EMP_ST_QP_FAILED"
            continue
        assert st[b] == 0 and iters[b] <= 40
        assert_rel(l[b, :n], g["qp_l"][b, :n], RTOL, "qp_l")
        assert_rel(dl[b, :n], g["qp_dl"][b, :n], RTOL, "qp_dl")
        assert_rel(ddl[b, :n], g["qp_ddl"][b, :n], RTOL, "qp_ddl")
        if n_checked < 4:      # solver-independent certificate against the reference's dense matrices
            H, f, G, h, A, bb = op.path_qp_matrices(lo[b, :n], hi[b, :n], *g["start"][b, 1:])
            x = np.stack([l[b, :n], dl[b, :n], ddl[b, :n]], axis=1).reshape(-1)
            cert = qp_dense.kkt_certificate(H, f, G, h, A, bb, x)
            assert cert["stationarity"] < 1e-7 and cert["ineq_violation"] < 1e-9 and cert["eq_violation"] < 1e-9
        n_checked += 1
    assert n_checked >= 5


def test_smooth_line_reference_line_size(planner):
    """smooth_reference_line on 51-point lines (the size motion_planning smooths, test_9.py:110)."""
    from emplanner_carla_amd.api import smooth_params
    rng = np.random.default_rng(21)
    B, m = 6, 51
    xy = np.zeros((B, 64, 2))
    n_pts = np.array([51, 51, 30, 2, 51, 17], np.int32)
    want = []
    for b in range(B):
        t = np.arange(n_pts[b]) * 2.0
        pts = np.stack([t * np.cos(0.3 * b) + rng.normal(0, 0.15, n_pts[b]),
                        t * np.sin(0.3 * b) + 15 * np.sin(t / 40.0) + rng.normal(0, 0.15, n_pts[b])], axis=1)
        xy[b, :n_pts[b]] = pts
        want.append(np.asarray(op.smooth_reference_line([tuple(p) for p in pts]), dtype=np.float64))
    out, iters, st = planner.smooth_line(smooth_params(), xy, n_pts)
    assert (st == 0).all()
    for b in range(B):
        n = n_pts[b]
        assert_rel(out[b, :n, :2], want[b][:, :2], RTOL, "smoothed xy")
        assert_rel(out[b, :n, 2], want[b][:, 2], RTOL, "theta")
        assert_rel(out[b, :n, 3], want[b][:, 3], RTOL, "kappa")


@pytest.mark.parametrize("cap,sizes", [(24, [24, 23, 9, 2]), (100, [100, 70, 65, 33])])
def test_smooth_line_all_kernel_paths(planner, cap, sizes):
    """The three smoothing code paths: <= 32 points (x | y on the half-waves), 33..64 (one point per lane,
    covered above) and > 64 (LDS-resident solver), each against the faithful port."""
    from emplanner_carla_amd.api import smooth_params
    rng = np.random.default_rng(cap)
    B = len(sizes)
    xy = np.zeros((B, cap, 2))
    n_pts = np.array(sizes, np.int32)
    want = []
    for b in range(B):
        t = np.arange(n_pts[b]) * 2.0
        pts = np.stack([t * np.cos(0.2 * b) + rng.normal(0, 0.12, n_pts[b]),
                        t * np.sin(0.2 * b) + 10 * np.sin(t / 35.0) + rng.normal(0, 0.12, n_pts[b])], axis=1)
        xy[b, :n_pts[b]] = pts
        want.append(np.asarray(op.smooth_reference_line([tuple(p) for p in pts]), dtype=np.float64))
    out, iters, st = planner.smooth_line(smooth_params(), xy, n_pts)
    assert (st == 0).all() and (iters > 0).all()
    for b in range(B):
        n = n_pts[b]
        assert_rel(out[b, :n, :2], want[b][:, :2], RTOL, "smoothed xy")
        if n >= 3:
            assert_rel(out[b, :n, 2], want[b][:, 2], RTOL, "theta")
            assert_rel(out[b, :n, 3], want[b][:, 3], RTOL, "kappa")


@pytest.mark.parametrize("key", list(GOLD))
def test_full_cycle_vs_reference(planner, key):
    """emp_plan_cycle == reference motion_planning body (test_9 / test_7 / test_6 driver forms)."""
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    cfg, fname, mode = GOLD[key]
    g = load_golden(fname)
    q = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width, **mode)
    r = planner.plan_cycle(dp_params_from_cfg(cfg), q, smooth_params(), **_inputs(g))
    B = len(g["seeds"])
    n_traj = 0
    for b in range(B):
        n = int(g["dp_len"][b])
        assert r.dp_len[b] == n
        assert np.array_equal(r.dp_s[b, :n], g["dp_s"][b, :n]) or np.allclose(r.dp_s[b, :n], g["dp_s"][b, :n], rtol=RTOL)
        assert_dp_l_vs_reference(r.dp_l[b, :n], g["dp_l"][b, :n])
        assert bool(r.status[b] & 1) == bool(g["dp_infeasible_banner"][b])
        if g["status"][b] == 3:       # the reference raised IndexError in cal_lmin_lmax (path_planning.py:267 / :272)
            assert r.status[b] & 4 and r.traj_len[b] == 0
            continue
        if g["status"][b] == 4:
            assert r.status[b] & 8 and r.traj_len[b] == 0
            continue
        assert g["status"][b] == 0 and (r.status[b] & ~1) == 0, f"scene {b}: status {r.status[b]}"
        m = int(g["traj_len"][b])
        assert r.traj_len[b] == m and r.path_len[b] == m - 1
        assert_rel(r.path_s[b, :m - 1], g["path_s"][b, :m - 1], RTOL, "path_s")
        assert_rel(r.path_l[b, :m - 1], g["path_l"][b, :m - 1], RTOL, "path_l")
        assert_rel(r.traj[b, :m, :3], g["traj"][b, :m, :3], RTOL, "x, y, theta")
        assert_rel(r.traj[b, :m, 3], g["traj"][b, :m, 3], RTOL, "kappa")   # |kappa| ~ 1e-3..1e-1 1/m
        n_traj += 1
    assert n_traj >= 5


def test_cycle_batch_properties_and_device_tensors(planner):
    """BASELINE configs[2] shape on 1024 scenes, inputs resident on the GPU (torch tensors):
    size-independent properties + agreement with the host-pointer path."""
    import torch
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    cfg = S.CFG2
    b = S.make_batch(range(5000, 6024), cfg)
    B, P = b.ref.shape[:2]
    host = dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy,
                start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
    p, q, sp = dp_params_from_cfg(cfg), qp_params(), smooth_params()
    r = planner.plan_cycle(p, q, sp, **host)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in host.items()}
    rd = planner.plan_cycle(p, q, sp, **dev)
    planner.synchronize()
    for name in ("dp_rows", "dp_l", "path_l", "traj", "traj_len", "status"):
        assert np.array_equal(getattr(rd, name).cpu().numpy(), getattr(r, name)), name
    ok = (r.status & ~1) == 0
    assert ok.mean() > 0.6 and (r.status & 8).any(), "mostly drivable scenes plus some infeasible QPs"
    # un-smoothed targets of every scene, to check the smoothing QP's +-0.2 m box
    sm, _, _, bsl, _ = planner.frenet_project(**host)
    tgt, cnt, st = planner.frenet_path_to_xy(b.ref, sm, host["n_ref"], bsl, r.path_s, r.path_l, r.path_len)
    for i in np.nonzero(ok)[0]:
        m = r.traj_len[i]
        n = r.path_len[i]
        assert m == n + 1 == 23 and cnt[i] == m
        # the path QP honours its pinned end state: the last station sits on the reference line
        assert abs(r.path_l[i, n - 1]) < 1e-9
        x = r.traj[i, :m]
        assert np.isfinite(x).all()
        assert (np.abs(x[:, :2] - tgt[i, :m]) <= 0.2 + 1e-9).all(), "smoothed point left its box"
        seg = np.hypot(np.diff(x[:, 0]), np.diff(x[:, 1]))
        assert (seg[1:] > 0.5).all() and (seg < 6.0).all(), "consecutive trajectory points 2.5-5 m apart"
        # (points 0 and 1 nearly coincide - the reference prepends the planning start, path_planning.py:31-34 -
        # so heading / curvature there are ill-defined in the reference too)
        assert (np.abs(x[2:, 3]) < 0.5).all()


def test_cycle_edge_cases(planner):
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    cfg = S.CFG_DEFAULT
    p, q, sp = dp_params_from_cfg(cfg), qp_params(), smooth_params()
    b = S.make_batch(range(3), cfg)
    B, P = b.ref.shape[:2]
    base = dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy,
                start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
    # empty batch
    e = {k: v[:0] for k, v in base.items()}
    r = planner.plan_cycle(p, q, sp, **e)
    assert r.traj.shape[0] == 0
    # a reference line too short for the DP horizon: the reference truncates at s_map[-1] (path_planning.py:40)
    short = dict(base)
    short["n_ref"] = np.full(B, 30, np.int32)
    r = planner.plan_cycle(p, q, sp, **short)
    want = op.plan_cycle([tuple(x) for x in b.ref[0, :30]], b.origin_xy[0], b.start_xy[0], b.start_v[0], b.start_a[0],
                         b.obs_xy[0, :b.n_obs[0]], dp_kwargs=dict(sampling_res=cfg.sampling_res, row=cfg.row,
                                                                  col=cfg.col, sample_s=cfg.sample_s,
                                                                  sample_l=cfg.sample_l), verbose=False)
    m = len(want["trajectory"])
    assert r.traj_len[0] == m and m < 26
    assert_rel(r.traj[0, :m, :3], np.asarray(want["trajectory"], dtype=np.float64)[:, :3], RTOL, "truncated traj")
    # capacity too small -> flagged, nothing written out of bounds
    r = planner.plan_cycle(p, q, sp, max_pts=20, **base)
    assert (r.status & 32).all() and (r.traj_len == 0).all()


def _global_path(rng, n, ds=2.0):
    """A long noisy S-curve with heading / curvature from the port's cal_heading_kappa (what global_planning feeds)."""
    t = np.arange(n) * ds
    xy = np.stack([t + rng.normal(0, 0.05, n), 25.0 * np.sin(t / 60.0) + rng.normal(0, 0.05, n)], axis=1)
    th, ka = op.cal_heading_kappa([tuple(p) for p in xy])
    return np.column_stack([xy, th, ka])


def test_reference_line_front_end_vs_port(planner):
    """emp_reference_line == find_match_points -> sampling -> smooth_reference_line (test_9.py:99-110), including
    matches near both ends of the global path, a first run, a backward search and a path that is too short."""
    from emplanner_carla_amd.api import smooth_params
    rng = np.random.default_rng(5)
    G = 220
    cases = [(220, 60, 57, 0), (220, 3, 5, 0), (220, 214, 210, 0), (220, 100, 0, 1), (220, 80, 90, 0), (51, 20, 18, 0),
             (40, 20, 18, 0)]
    B = len(cases)
    gp = np.zeros((B, G, 4))
    n_global = np.zeros(B, np.int32)
    pred = np.zeros((B, 2))
    pre = np.zeros(B, np.int32)
    first = np.zeros(B, np.int32)
    want = []
    for b, (n, at, pre_idx, is_first) in enumerate(cases):
        path = _global_path(rng, n)
        gp[b, :n] = path
        n_global[b], pre[b], first[b] = n, pre_idx, is_first
        pred[b] = path[at, :2] + np.array([0.4, -0.7])
        nodes = [tuple(r) for r in path]
        match, _ = op.find_match_points([tuple(pred[b])], nodes, bool(is_first), int(pre_idx))
        if n >= 51:
            local = op.sampling(int(match[0]), nodes)
            assert len(local) == 51
            want.append((int(match[0]), np.asarray(op.smooth_reference_line(local), dtype=np.float64)))
        else:
            want.append((int(match[0]), None))
    ref, n_ref, mi, it, st = planner.reference_line(smooth_params(), gp, n_global, pred, pre, first)
    for b in range(B):
        assert mi[b] == want[b][0], f"match index of case {b}"
        if want[b][1] is None:
            assert st[b] != 0 and n_ref[b] == 0
            continue
        assert st[b] == 0 and n_ref[b] == 51 and it[b] > 0
        assert_rel(ref[b, :, :2], want[b][1][:, :2], RTOL, "reference line xy")
        assert_rel(ref[b, :, 2], want[b][1][:, 2], RTOL, "theta")
        assert_rel(ref[b, :, 3], want[b][1][:, 3], RTOL, "kappa")


def test_reference_line_feeds_the_cycle(planner):
    """Front end and cycle chained on the device: the cycle run on emp_reference_line's output equals the cycle run on
    the port's smoothed reference line."""
    import torch
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params, max_path_points
    rng = np.random.default_rng(9)
    # 60 m horizon: the 51-node local line reaches 80 m ahead of the match (40 nodes, 2 m apart)
    cfg = S.LatticeConfig("front_end_24x9", row=9, col=24, sample_s=2.5, sample_l=1.5, sampling_res=2, n_obs=8)
    B, G = 4, 200
    gp = np.stack([_global_path(rng, G) for _ in range(B)])
    at = np.array([30, 60, 90, 120])
    pred = gp[np.arange(B), at + 1, :2] + rng.normal(0, 0.2, (B, 2))
    origin = gp[np.arange(B), at, :2] + rng.normal(0, 0.2, (B, 2))
    pre = (at - 2).astype(np.int32)
    layout = [(8, 4.5), (15, -4.5), (21, 5.0)]                   # (nodes ahead of the match, lateral offset)
    obs = np.zeros((B, 8, 2))
    for b in range(B):
        for k, (di, off) in enumerate(layout):
            th = gp[b, at[b] + di, 2]
            obs[b, k] = gp[b, at[b] + di, :2] + off * np.array([-np.sin(th), np.cos(th)])
    n_obs = np.full(B, len(layout), np.int32)
    v = np.tile([8.0, 0.5], (B, 1))
    a0 = np.zeros((B, 2))
    dev = torch.device("cuda:0")
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    sp = smooth_params()
    ref, n_ref, mi, it, st = planner.reference_line(sp, t(gp), t(np.full(B, G, np.int32)), t(pred), t(pre))
    planner.synchronize()                      # outputs are produced on the planner's stream
    assert (st.cpu().numpy() == 0).all()
    p, q = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width)
    res = planner.plan_cycle(p, q, sp, max_pts=max_path_points(p), ref_line=ref, n_ref=n_ref, origin_xy=t(origin), start_xy=t(pred),
                             start_v=t(v), start_a=t(a0), obs_xy=t(obs), n_obs=t(n_obs))
    planner.synchronize()
    status = res.status.cpu().numpy()
    traj = res.traj.cpu().numpy()
    tl = res.traj_len.cpu().numpy()
    checked = 0
    for b in range(B):
        nodes = [tuple(r) for r in gp[b]]
        match, _ = op.find_match_points([tuple(pred[b])], nodes, False, int(pre[b]))
        line = op.smooth_reference_line(op.sampling(int(match[0]), nodes))
        try:
            out = op.plan_cycle(line, tuple(origin[b]), tuple(pred[b]), tuple(v[b]), tuple(a0[b]), [tuple(o) for o in obs[b, :len(layout)]],
                                dp_kwargs=dict(row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l,
                                               sampling_res=cfg.sampling_res), obs_length=cfg.obs_length,
                                obs_width=cfg.obs_width, verbose=False)
            port_ok = bool(out["dp_feasible"]) and out.get("qp_status", "optimal") == "optimal" and out["smooth_status"] == "optimal"
        except IndexError:
            port_ok = False
        assert port_ok == (status[b] == 0), f"scene {b}: port ok {port_ok}, device status {status[b]}"
        if not port_ok:
            continue
        want = np.asarray(out["trajectory"], dtype=np.float64)
        assert tl[b] == len(want)
        assert_rel(traj[b, :tl[b], :2], want[:, :2], RTOL, "trajectory xy")
        assert_rel(traj[b, :tl[b], 2], want[:, 2], RTOL, "trajectory theta")
        checked += 1
    assert checked >= 2


def _driver_request(g, c):
    ns, nd = int(g["n_static"][c]), int(g["n_dynamic"][c])
    return ([tuple(r) for r in g["static"][c, :ns]], [tuple(r) for r in g["dynamic"][c, :nd]], tuple(g["veh"][c]),
            tuple(g["pred"][c]), tuple(g["v"][c]), tuple(g["a"][c]), [tuple(r) for r in g["path"][c]], [int(g["pre_match"][c])])


def test_planning_process_body_vs_reference_driver(planner):
    """emplanner_carla_amd.service.plan_requests == the reference's motion_planning(conn) (test_9.py:92-220), run for real
    through a fake Pipe when the fixture was made: front end, the nearest-static-obstacle rule, the virtual obstacles of
    the first dynamic obstacle, DP, QPs.  All 18 requests in ONE batch of two device calls.  This fixture runs the DP
    with sample_s = 14.7 (keyword default overridden from outside, see tests/golden/make_golden_driver.py)."""
    from emplanner_carla_amd import service
    from emplanner_carla_amd.api import dp_params
    g = load_golden("driver_s147.npz")
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    replies = service.plan_requests(planner, reqs, dp=dp_params(sample_s=14.7))
    compared = 0
    for c, (reply, status) in enumerate(replies):
        if not g["qp_ok"][c]:
            # an infeasible QP: the reference ignores cvxopt's status and sends its last iterate (path_planning.py:
            # 211-218); here the request is refused
            assert reply is None and status & (8 | 16), f"request {c}: status {status}"
            continue
        assert bool(g["ok"][c]) and reply is not None, f"request {c}: status {status}"
        compared += 1
        traj, match, ps, pl = reply
        n, m = int(g["n_traj"][c]), int(g["n_path"][c])
        assert match[0] == g["match"][c]
        assert len(traj) == n and len(ps) == m, f"request {c} (kind {g['case'][c]}): {len(traj)} trajectory points, reference {n}"
        assert_rel(np.asarray(ps), g["path_s"][c, :m], RTOL, "path_s")
        assert_rel(np.asarray(pl), g["path_l"][c, :m], RTOL, "path_l")
        t = np.asarray(traj)
        assert_rel(t[:, :3], g["traj"][c, :n, :3], RTOL, "x, y, theta")
        assert_rel(t[:, 3], g["traj"][c, :n, 3], RTOL, "kappa")
    assert compared >= 12


def test_planning_process_body_default_lattice_stage_by_stage(planner):
    """The driver exactly as it is runs the DP with sample_s = 15 (path_planning.py:277-279): the reference then sizes
    each densified segment with int(end_s - start_s) (:405, :423), i.e. int(15 -+ 1 ulp), and which side of 15 the
    difference falls on follows the last bits of the planning start's s - the output of the reference-line smoothing QP,
    which no two solvers reproduce to the bit (the reference's own cvxopt included).  So every stage of the GPU chain is
    checked against oracle/ref_port fed with the GPU's OWN upstream output, at 1e-6 / index-exact / equal point counts:
    front end -> projection -> DP (rows, densified path) -> bounds + path QP + midpoints -> Cartesian tail, on all 18
    requests of the reference driver run (test_9.py:99-218).  Requests whose stations equal the reference run's
    (no int() tie flipped) are ALSO compared with the reference's recorded reply directly."""
    from emplanner_carla_amd import service
    g = load_golden("driver.npz")
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    st = {}
    replies = service.plan_requests(planner, reqs, stages=st)
    a, ref, n_ref, res = st["inputs"], st["ref_line"], st["n_ref"], st["cycle"]
    B = len(reqs)
    assert (st["ref_status"] == 0).all()
    sm, os_, ol_, bsl, start = planner.frenet_project(ref, n_ref, a["veh"], a["pred"], a["v"], a["a"], a["obs_xy"], a["n_obs"])
    flipped, direct, staged = [], 0, 0
    for c in range(B):
        static, dynamic, veh, pred, v, acc, path, match_list = reqs[c]
        # ---- stage 1, front end (test_9.py:99-110): windowed match, 51-node window, smoothing QP
        want_match, _ = op.find_match_points([tuple(pred)], path, False, match_list[0])
        assert int(st["match"][c]) == want_match[0] == int(g["match"][c])
        want_line = np.asarray(op.smooth_reference_line(op.sampling(want_match[0], path)), dtype=np.float64)
        P = int(n_ref[c])
        assert P == len(want_line)
        assert_rel(ref[c, :P, :3], want_line[:, :3], RTOL, f"request {c}: reference line x, y, theta")
        assert_rel(ref[c, :P, 3], want_line[:, 3], RTOL, f"request {c}: reference line kappa")
        # ---- stage 2, projection (test_9.py:113-177) on the GPU's reference line
        line = [tuple(r) for r in ref[c, :P]]
        s_map = op.cal_s_map_fun(line, origin_xy=tuple(veh))
        assert_rel(sm[c, :P], np.asarray(s_map), RTOL, f"request {c}: s_map")
        k = int(a["n_obs"][c])
        if k:
            ws, wl = op.cal_s_l_fun([tuple(x) for x in a["obs_xy"][c, :k]], line, s_map)
            assert_rel(os_[c, :k], np.asarray(ws), RTOL, f"request {c}: obstacle s")
            assert_rel(ol_[c, :k], np.asarray(wl), RTOL, f"request {c}: obstacle l")
        bs, bl = op.cal_s_l_fun([tuple(pred)], line, s_map)
        assert_rel(bsl[c], np.asarray([bs[0], bl[0]]), RTOL, f"request {c}: begin s, l")
        l0, _, _, _, dl0, _, ddl0 = op.cal_s_l_deri_fun([tuple(pred)], [tuple(v)], [tuple(acc)], line, tuple(pred))
        assert_rel(start[c, 1:2], np.asarray([l0[0]]), RTOL, f"request {c}: start l")
        assert_rel(start[c, 2:], np.asarray([dl0[0], ddl0[0]]), RTOL, f"request {c}: start dl, ddl")
        # ---- stage 3, DP (test_9.py:180) fed with the GPU's projection; virtual obstacles from the GPU's begin_s (:137-169)
        obs_s, obs_l = list(os_[c, :k]), list(ol_[c, :k])
        dyn = None if np.isnan(a["dyn"][c, 0]) else tuple(a["dyn"][c])
        for vs, vl in op.virtual_obstacles(float(start[c, 0]), tuple(v), dyn):
            obs_s.append(vs)
            obs_l.append(vl)
        dp_s, dp_l, rows, _ = op.DP_algorithm(obs_s, obs_l, float(start[c, 0]), float(start[c, 1]), float(start[c, 2]),
                                              float(start[c, 3]), _return_rows=True, _verbose=False)
        assert np.array_equal(res.dp_rows[c], np.asarray(rows, dtype=np.float64)), f"request {c}: DP rows"
        n = int(res.dp_len[c])
        assert n == len(dp_s), f"request {c}: {n} densified points, the port fed with the same start s yields {len(dp_s)}"
        assert_rel(res.dp_s[c, :n], np.asarray(dp_s), RTOL, f"request {c}: dp_s")
        assert_rel(res.dp_l[c, :n], np.asarray(dp_l), RTOL, f"request {c}: dp_l")
        # ---- stage 4, bounds + path QP + midpoints (test_9.py:187-210) fed with the GPU's densified path
        ds, dl = list(res.dp_s[c, :n:2]), list(res.dp_l[c, :n:2])
        try:
            l_min, l_max = op.cal_lmin_lmax(ds, dl, obs_s, obs_l, 5, 5)
        except IndexError:
            assert res.status[c] & 4, f"request {c}: the reference raises IndexError in cal_lmin_lmax"
            continue
        ql, _, _, status = op.Quadratic_planning(l_min, l_max, float(start[c, 1]), float(start[c, 2]), float(start[c, 3]),
                                                 _return_status=True)
        if status != "optimal":
            assert res.status[c] & 8 and replies[c][0] is None, f"request {c}: infeasible path QP must be refused"
            continue
        assert (res.status[c] & ~1) == 0 and replies[c][0] is not None, f"request {c}: status {res.status[c]}"
        path_s = [ds[0]] + [(ds[j] + ds[j - 1]) / 2 for j in range(1, len(ql))] + [ds[-1]]
        path_l = [ql[0]] + [(ql[j] + ql[j - 1]) / 2 for j in range(1, len(ql))] + [ql[-1]]
        m = len(path_s)
        assert res.path_len[c] == m
        assert_rel(res.path_s[c, :m], np.asarray(path_s), RTOL, f"request {c}: path s")
        assert_rel(res.path_l[c, :m], np.asarray(path_l), RTOL, f"request {c}: path l")
        # ---- stage 5, Cartesian tail (test_9.py:212-218) fed with the GPU's path
        want = np.asarray(op.frenet_2_x_y_theta_kappa(float(bsl[c, 0]), float(bsl[c, 1]), list(res.path_s[c, :m]),
                                                      list(res.path_l[c, :m]), line, list(sm[c, :P])), dtype=np.float64)
        t = len(want)
        assert res.traj_len[c] == t, f"request {c}: {res.traj_len[c]} trajectory points, port {t}"
        assert_rel(res.traj[c, :t, :3], want[:, :3], RTOL, f"request {c}: trajectory x, y, theta")
        assert_rel(res.traj[c, :t, 3], want[:, 3], RTOL, f"request {c}: trajectory kappa")
        staged += 1
        # ---- the reference's own reply, wherever the point counts agree (no int() tie flipped by the last bits of s)
        if not g["qp_ok"][c]:
            continue
        traj, match, ps, pl = replies[c][0]
        # an int() tie that fell the other way moves every station behind it by one sample (two opposite flips leave
        # the point count unchanged): the reference's stations are the criterion, not the count
        if m != int(g["n_path"][c]) or not rel_close(np.asarray(ps), g["path_s"][c, :m], RTOL).all():
            flipped.append(c)
            continue
        nt = int(g["n_traj"][c])
        assert len(traj) == nt
        assert_rel(np.asarray(pl), g["path_l"][c, :m], RTOL, f"request {c}: path_l vs the reference run")
        tj = np.asarray(traj)
        assert_rel(tj[:, :3], g["traj"][c, :nt, :3], RTOL, f"request {c}: trajectory vs the reference run")
        assert_rel(tj[:, 3], g["traj"][c, :nt, 3], RTOL, f"request {c}: kappa vs the reference run")
        direct += 1
    print(f"default lattice: {staged} requests staged at 1e-6, {direct} also equal to the reference run, "
          f"int() tie flipped on requests {flipped}")
    assert staged >= 14
    assert direct + len(flipped) >= 12


def test_motion_planning_process_loop(planner):
    """The Pipe-protocol loop (drop-in for the reference's child process) answers a request and blocks for the next."""
    from emplanner_carla_amd import service
    from emplanner_carla_amd.api import dp_params
    g = load_golden("driver_s147.npz")        # non-integer sample_s: no int() truncation flips (HISTORY.md 7)

    class Done(Exception):
        pass

    class FakeConn:
        def __init__(self, reqs):
            self.reqs, self.sent = list(reqs), []

        def recv(self):
            if not self.reqs:
                raise Done()
            return self.reqs.pop(0)

        def send(self, obj):
            self.sent.append(obj)

    conn = FakeConn([_driver_request(g, 4), _driver_request(g, 9)])
    with pytest.raises(Done):
        service.motion_planning(conn, dp=dp_params(sample_s=14.7))
    assert len(conn.sent) == 2
    for reply, c in zip(conn.sent, (4, 9)):
        traj, match, ps, pl = reply
        assert len(traj) == g["n_traj"][c] and match == [int(g["match"][c])]
        assert isinstance(traj[0], tuple) and len(traj[0]) == 4 and isinstance(ps, list)
        assert_rel(np.array(traj)[:, :3], g["traj"][c, :len(traj), :3], 1e-6, "trajectory")


def _child_motion_planning(conn, sample_s):
    from emplanner_carla_amd import service
    from emplanner_carla_amd.api import dp_params
    service.motion_planning(conn, dp=dp_params(sample_s=sample_s))


def test_motion_planning_in_a_real_child_process():
    """The reference's process model (test_9.py:225-227, 390-395, 445): the planner runs in a child started with
    multiprocessing.Process, creates its device context there, is fed request tuples over a Pipe and is terminated by
    the parent.  Spawned, not forked: this pytest process has used HIP already, and a HIP runtime does not survive a
    fork (the reference's author ran on Windows, where spawn is the only start method).  Three requests, the second
    with an infeasible path QP: the loop must stay alive and answer it with the PREVIOUS valid trajectory and the new
    match index - what an unmodified driver, which hands element 0 of the reply to its controller, can digest."""
    import multiprocessing as mp
    g = load_golden("driver_s147.npz")
    bad = int(np.flatnonzero(~g["qp_ok"])[0])
    order = [4, bad, 9]
    ctx = mp.get_context("spawn")
    parent, child = ctx.Pipe()
    p = ctx.Process(target=_child_motion_planning, args=(child, 14.7), daemon=True)
    p.start()
    try:
        got = []
        for c in order:
            parent.send(_driver_request(g, c))
            assert parent.poll(180.0), f"no reply to request {c} (child alive: {p.is_alive()})"
            got.append(parent.recv())
        assert p.is_alive()
    finally:
        p.terminate()                                                       # test_9.py:445
        p.join(30)
    for c, reply in zip(order, got):
        traj, match, ps, pl = reply
        if c == bad:
            assert traj == got[0][0] and ps == got[0][2] and pl == got[0][3] and match == [int(g["match"][c])]
            continue
        n = int(g["n_traj"][c])
        assert len(traj) == n and match == [int(g["match"][c])]
        assert isinstance(traj[0], tuple) and len(traj[0]) == 4 and isinstance(ps, list)
        assert_rel(np.array(traj)[:, :3], g["traj"][c, :n, :3], RTOL, f"request {c}: trajectory from the child process")
        assert_rel(np.array(ps), g["path_s"][c, :len(ps)], RTOL, "path_s")


def test_cycle_with_dynamic_obstacle_on_the_fine_lattice_vs_port(planner):
    """The virtual obstacles of the first dynamic obstacle (test_9.py:137-169) on the 40x9 lattice, next to eight
    static obstacles: every stage against oracle/ref_port.plan_cycle (itself pinned on the reference driver run)."""
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    cfg = S.CFG2
    seeds = list(range(40, 96))
    b = S.make_batch(seeds, cfg)
    B, P = b.ref.shape[:2]
    rng = np.random.default_rng(8)
    dyn = np.full((B, 2), np.nan)
    has = rng.random(B) < 0.75
    dyn[has, 0] = rng.uniform(8.0, 45.0, has.sum())                 # distance ahead
    dyn[has, 1] = rng.uniform(0.0, 6.0, has.sum())                  # its speed; the ego's is ~8 m/s
    v = np.tile([8.0, 0.0], (B, 1)) + rng.normal(0, 0.2, (B, 2))
    r = planner.plan_cycle(dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width),
                           smooth_params(), ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy,
                           start_xy=b.start_xy, start_v=v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs,
                           dyn_dis_speed=dyn)
    kw = dict(sampling_res=cfg.sampling_res, row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l)
    compared = with_virtual = 0
    for i in range(B):
        d = None if np.isnan(dyn[i, 0]) else tuple(dyn[i])
        try:
            want = op.plan_cycle([tuple(x) for x in b.ref[i]], b.origin_xy[i], b.start_xy[i], v[i], b.start_a[i],
                                 b.obs_xy[i, :b.n_obs[i]], dp_kwargs=kw, obs_length=cfg.obs_length,
                                 obs_width=cfg.obs_width, verbose=False, dyn_dis_speed=d)
        except IndexError:
            assert r.status[i] & (2 | 4), f"scene {i}: the reference raises IndexError"
            continue
        with_virtual += len(want["obs_s"]) > b.n_obs[i]
        assert np.array_equal(r.dp_rows[i], np.asarray(want["dp_rows"], dtype=np.float64)), f"scene {i}: DP rows"
        if want.get("qp_status") not in (None, "optimal"):
            assert r.status[i] & 8
            continue
        assert (r.status[i] & ~1) == 0, f"scene {i}: status {r.status[i]}"
        m = len(want["trajectory"])
        assert r.traj_len[i] == m
        assert_rel(r.traj[i, :m, :3], np.asarray(want["trajectory"], dtype=np.float64)[:, :3], RTOL, f"scene {i} trajectory")
        compared += 1
    assert compared >= 10 and with_virtual >= 8


def test_smoothing_active_set_on_hard_polylines(planner):
    """The active-set smoother (emp_qp_wave.h: box_qp_active_set_lanes) on polylines that are much rougher than its
    0.2 m boxes - most coordinates end on a bound, zig-zags flip between the two bounds, some points repeat - against
    the faithful port's certified minimiser, for the half-wave (<= 32 points) and the full-wave (33..64) paths."""
    from emplanner_carla_amd.api import smooth_params
    rng = np.random.default_rng(77)
    cases = []
    for m in (5, 23, 32, 33, 51, 64):
        for kind in range(6):
            t = np.arange(m) * 2.0
            base = np.column_stack([t, 8 * np.sin(t / 25.0)])
            if kind == 0:
                pts = base + rng.normal(0, 0.6, (m, 2))                   # noise three times the box
            elif kind == 1:
                pts = base + 0.5 * np.column_stack([(-1.0) ** np.arange(m), (-1.0) ** (np.arange(m) // 2)])   # zig-zag
            elif kind == 2:
                pts = base.copy()
                pts[m // 2:] += [0.0, 1.5]                                # a step in the line
            elif kind == 3:
                pts = np.repeat(base[:(m + 1) // 2], 2, axis=0)[:m]       # every point twice
            elif kind == 4:
                pts = base + rng.normal(0, 0.02, (m, 2))                  # nearly smooth: no bound active
            else:
                pts = base + rng.uniform(-0.2, 0.2, (m, 2)) * 1.0000001   # deviations right at the box size
            cases.append(pts)
    B = len(cases)
    xy = np.zeros((B, 64, 2))
    n_pts = np.array([len(c) for c in cases], np.int32)
    for b, c in enumerate(cases):
        xy[b, :len(c)] = c
    out, iters, st = planner.smooth_line(smooth_params(), xy, n_pts)
    assert (st == 0).all()
    at_bound = 0
    for b, c in enumerate(cases):
        n = len(c)
        want = np.asarray(op.smooth_reference_line([tuple(p) for p in c]), dtype=np.float64)
        assert_rel(out[b, :n, :2], want[:, :2], RTOL, f"case {b} smoothed xy")
        dev = np.abs(out[b, :n, :2] - c)
        assert (dev <= 0.2 + 1e-12).all()
        at_bound += int((dev >= 0.2 - 1e-12).sum())
    assert at_bound > 500, "the hard cases really end on their bounds"
    assert iters.max() <= 16, "the active-set iteration settled everywhere (no interior-point fallback)"


def test_full_cycle_on_the_wide_lattice_stage_by_stage(planner):
    """BASELINE config 5's 120x21 lattice with 16 obstacles through the whole cycle: 121-point DP paths, 61-station path
    QPs (one scene per wavefront) and 62-point trajectories (the wide Cartesian kernel).  The reference's own quintic is
    too ill-conditioned on this lattice to be the yardstick for the DP (the DP is judged bit for
    bit against oracle/exact.py in tests/test_gpu_dp.py), so the stages BEHIND the DP are checked against the port fed
    with the GPU's DP path: bounds + path QP, midpoints, Frenet -> Cartesian + smoothing + heading / curvature."""
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params, max_path_points
    cfg = S.CFG5
    b = S.make_batch(range(70, 82), cfg)
    B, P = b.ref.shape[:2]
    p = dp_params_from_cfg(cfg)
    M = max_path_points(p)
    r = planner.plan_cycle(p, qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params(), max_pts=M,
                           ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy,
                           start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
    sm, os_, ol_, bsl, start = planner.frenet_project(b.ref, np.full(B, P, np.int32), b.origin_xy, b.start_xy, b.start_v,
                                                      b.start_a, b.obs_xy, b.n_obs)
    checked = 0
    for i in range(B):
        n = int(r.dp_len[i])
        assert 100 <= n <= 121           # sample_s = 1.0: int(end_s - start_s) of every 1 m segment is 0 or 1 on a last bit
        ds, dl = list(r.dp_s[i, :n:2]), list(r.dp_l[i, :n:2])
        k = int(b.n_obs[i])
        try:
            l_min, l_max = op.cal_lmin_lmax(ds, dl, list(os_[i, :k]), list(ol_[i, :k]), cfg.obs_length, cfg.obs_width)
        except IndexError:
            assert r.status[i] & 4
            continue
        try:
            ql, _, _, status = op.Quadratic_planning(l_min, l_max, start[i, 1], start[i, 2], start[i, 3], _return_status=True)
        except np.linalg.LinAlgError:                  # the dense oracle's interior point diverged: infeasible corridor
            status = "diverged"
        if status == "diverged":
            assert r.status[i] & 8, f"scene {i}: the port's QP diverged, the kernel reports {r.status[i]}"
            continue
        if status != "optimal":                           # the dense oracle could not certify its own answer: no yardstick
            continue
        assert (r.status[i] & ~1) == 0, f"scene {i}: status {r.status[i]}"
        path_s = [ds[0]] + [(ds[j] + ds[j - 1]) / 2 for j in range(1, len(ql))] + [ds[-1]]
        path_l = [ql[0]] + [(ql[j] + ql[j - 1]) / 2 for j in range(1, len(ql))] + [ql[-1]]
        m = len(path_s)
        assert r.path_len[i] == m and 50 <= m <= 62
        assert_rel(r.path_l[i, :m], np.asarray(path_l), RTOL, f"scene {i} path l")
        want = np.asarray(op.frenet_2_x_y_theta_kappa(bsl[i, 0], bsl[i, 1], path_s, path_l, [tuple(x) for x in b.ref[i]],
                                                      list(sm[i])), dtype=np.float64)
        t = len(want)
        assert r.traj_len[i] == t
        assert_rel(r.traj[i, :t, :3], want[:, :3], RTOL, f"scene {i} trajectory")
        assert_rel(r.traj[i, 2:t, 3], want[2:, 3], RTOL, f"scene {i} curvature")
        checked += 1
    assert checked >= 4


@pytest.mark.parametrize("col,max_pts,stations", [(33, None, 33), (34, 68, 34), (34, None, 34)])
def test_path_qp_at_its_size_limits(planner, col, max_pts, stations):
    """The cycle's path QP packs eight scenes into a wavefront while a scene has at most 34 stations (R = 4 stations per lane
    of an 8-lane group: every lane busy), four up to 66 (16-lane groups); the two-scenes-per-wavefront form
    (EMP_OPT_PATH_QP_FORM = 1, the child run of tests/test_gpu_fuzz.py) holds 34 stations per half-wave, with LDS arrays
    addressed by fixed strides and unclamped neighbour indices.  33 and 34 stations through the narrow kernel, 34 through the
    wide one (the output capacity decides), each against oracle/ref_port.plan_cycle."""
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    cfg = S.LatticeConfig(f"limits_{col}x5", row=5, col=col, sample_s=2.0, sample_l=1.0, sampling_res=1, n_obs=4)
    seeds = list(range(300, 316))
    b = S.make_batch(seeds, cfg)
    B, P = b.ref.shape[:2]
    kwargs = {} if max_pts is None else {"max_pts": max_pts}
    r = planner.plan_cycle(dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width),
                           smooth_params(), ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy,
                           start_xy=b.start_xy, start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs,
                           **kwargs)
    kw = dict(sampling_res=cfg.sampling_res, row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l)
    compared = at_size = 0
    for i in range(B):
        try:
            want = op.plan_cycle([tuple(x) for x in b.ref[i]], b.origin_xy[i], b.start_xy[i], b.start_v[i], b.start_a[i],
                                 b.obs_xy[i, :b.n_obs[i]], dp_kwargs=kw, obs_length=cfg.obs_length,
                                 obs_width=cfg.obs_width, verbose=False)
        except IndexError:
            assert r.status[i] & (2 | 4), f"scene {i}: the reference raises IndexError"
            continue
        assert np.array_equal(r.dp_rows[i], np.asarray(want["dp_rows"], dtype=np.float64)), f"scene {i}: DP rows"
        if want.get("qp_status") not in (None, "optimal"):
            assert r.status[i] & 8
            continue
        if max_pts is not None and len(want["dp_s"]) > max_pts:
            assert r.status[i] & 32, f"scene {i}: {len(want['dp_s'])} DP points do not fit {max_pts}: EMP_ST_TRUNCATED"
            continue
        assert (r.status[i] & ~1) == 0, f"scene {i}: status {r.status[i]}"
        n = len(want["qp_l"])
        at_size += n == stations
        m = len(want["trajectory"])
        assert r.traj_len[i] == m
        assert_rel(r.traj[i, :m, :3], np.asarray(want["trajectory"], dtype=np.float64)[:, :3], RTOL, f"scene {i} trajectory")
        compared += 1
    assert compared >= 10 and at_size >= 8


@pytest.mark.gpu
def test_cycle_graph_replays_the_same_bits():
    """EMP_OPT_CYCLE_GRAPH: consecutive emp_plan_cycle calls with one signature (same sizes, parameters, options, input and
    output memory) are captured into a hipGraph at the third call and replayed from the fourth on.  The graph carries pointers,
    not data: new scene contents in the same input tensors must give the plain path's results for THOSE scenes bit for bit; a
    call with another signature (other tensors, another batch size, another parameter) drops the graph and is itself correct; so
    is the call after a different entry point has used the context in between."""
    import torch
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
    cfg = S.CFG2
    p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
    M = max_path_points(p)
    fields = ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status")

    def host(seeds):
        b = S.make_batch(seeds, cfg, start_ahead=S.BENCH_START_AHEAD)
        return dict(ref_line=b.ref, n_ref=np.full(len(seeds), b.ref.shape[1], np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy,
                    start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)

    plain, pl = Planner(0), Planner(0)
    try:
        def expect(h):
            r = plain.plan_cycle(p, q, sp, max_pts=M, **h)
            return {k: np.asarray(getattr(r, k)) for k in fields}

        def same(r, e, what):
            for k in fields:
                assert np.array_equal(getattr(r, k).cpu().numpy(), e[k], equal_nan=True), (what, k)

        pl.set_option("cycle_graph", 1)
        replays = 0
        for n in (1, 5, 64):
            sets = [host(range(100 * n + 7 * k, 100 * n + 7 * k + n)) for k in range(6)]
            dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sets[0].items()}
            out = None
            for k, h in enumerate(sets):                       # calls 0-1 plain, call 2 captured, calls 3-5 replayed
                for name, v in h.items():
                    dev[name].copy_(torch.from_numpy(np.ascontiguousarray(v)))
                torch.cuda.synchronize()
                out = pl.plan_cycle(p, q, sp, max_pts=M, out=out, **dev)
                pl.synchronize()
                same(out, expect(h), f"{n} scenes, call {k}")
            assert pl.cycle_graph_replays() - replays == 3, "calls 3, 4 and 5 must have been graph replays"
            replays = pl.cycle_graph_replays()
            # another entry point in between (its temporaries come from the same pool), then the same signature again
            pl.frenet_project(sets[0]["ref_line"], sets[0]["n_ref"], sets[0]["origin_xy"], sets[0]["start_xy"], sets[0]["start_v"],
                              sets[0]["start_a"], sets[0]["obs_xy"], sets[0]["n_obs"])
            out = pl.plan_cycle(p, q, sp, max_pts=M, out=out, **dev)
            pl.synchronize()
            same(out, expect(sets[-1]), f"{n} scenes, after another entry point")
            # another parameter value: a new signature
            q2 = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width)
            q2.w_dl = q.w_dl * 2.0
            r2 = pl.plan_cycle(p, q2, sp, max_pts=M, out=out, **dev)
            pl.synchronize()
            e2 = plain.plan_cycle(p, q2, sp, max_pts=M, **sets[-1])
            assert np.array_equal(r2.traj.cpu().numpy(), np.asarray(e2.traj), equal_nan=True)
            # fresh output tensors: a new signature again
            r3 = pl.plan_cycle(p, q, sp, max_pts=M, **dev)
            pl.synchronize()
            same(r3, expect(sets[-1]), f"{n} scenes, fresh outputs")
            replays = pl.cycle_graph_replays()
    finally:
        pl.close()
        plain.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipe", [0, "staged", 3], ids=["no_pipeline", "staged", "lanes3"])
def test_cycle_from_the_global_path_equals_the_two_call_form(planner, pipe):
    """ABI 11: emp_plan_cycle with emp_cycle_io.global_path runs the reference's front end (test_9.py:99-110) in front of the cycle
    in ONE call - what service.plan_requests / motion_planning / the wire server now use.  Same kernels: the outputs, the match index
    and the front end's status must equal emp_reference_line followed by emp_plan_cycle bit for bit - for good requests, for a
    previous match index outside the path (IndexError in the reference), for a global path shorter than 51 nodes, with and without
    a dynamic obstacle, on host arrays and on device tensors, in every pipeline form."""
    import torch
    from emplanner_carla_amd import service
    from emplanner_carla_amd.api import dp_params, qp_params, smooth_params, max_path_points
    g = load_golden("driver_s147.npz")
    from tests.test_wire import _driver_request
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    bad = list(reqs[0])
    bad[7] = [10 ** 6]                                   # pre_match_index far outside the path
    short = list(reqs[1])
    short[6] = short[6][:40]                             # fewer nodes than the 51-point window
    short[7] = [5]
    reqs = reqs + [tuple(bad), tuple(short)]
    a = service.pack_requests(reqs)
    dp, qp, sp = dp_params(sample_s=14.7), qp_params(), smooth_params()
    stages = {}
    st_ref, match, res2, M = service.plan_arrays(planner, a, dp, qp, sp, stages=stages)            # the two-call form
    assert (st_ref[-2:] != 0).all() and (st_ref[:-2] == 0).all()
    planner.set_pipeline(pipe)
    try:
        for rep in range(3):
            res = planner.plan_cycle(dp, qp, sp, None, None, max_pts=M, origin_xy=a["veh"], start_xy=a["pred"], start_v=a["v"],
                                     start_a=a["a"], obs_xy=a["obs_xy"], n_obs=a["n_obs"], dyn_dis_speed=a["dyn"],
                                     global_path=a["global_path"], n_global=a["n_global"], pre_match_index=a["pre_match"])
            planner.synchronize()
            assert np.array_equal(res.ref_status, st_ref) and np.array_equal(res.match_index, match)
            for k in ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status"):
                assert np.array_equal(getattr(res, k), getattr(res2, k), equal_nan=True), (k, rep)
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in a.items()}
        rd = planner.plan_cycle(dp, qp, sp, None, None, max_pts=M, origin_xy=dev["veh"], start_xy=dev["pred"], start_v=dev["v"],
                                start_a=dev["a"], obs_xy=dev["obs_xy"], n_obs=dev["n_obs"], dyn_dis_speed=dev["dyn"],
                                global_path=dev["global_path"], n_global=dev["n_global"], pre_match_index=dev["pre_match"])
        planner.synchronize()
        torch.cuda.synchronize()
        assert np.array_equal(rd.ref_status.cpu().numpy(), st_ref) and np.array_equal(rd.match_index.cpu().numpy(), match)
        for k in ("dp_rows", "path_l", "traj", "traj_len", "status"):
            assert np.array_equal(getattr(rd, k).cpu().numpy(), getattr(res2, k), equal_nan=True), k
    finally:
        planner.set_pipeline(0)
    # and the one-request fast path of the planning process: the same reply as plan_requests
    one = service.RequestPlanner(planner, dp=dp)
    want = service.plan_requests(planner, reqs, dp=dp)
    for c, req in enumerate(reqs):
        reply, status, m = one.plan(req)
        assert status == want[c][1] and m == int(match[c]), c
        assert reply == want[c][0], c
    one.close()
