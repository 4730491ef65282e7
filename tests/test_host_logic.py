"""CPU checks of the kernels' scalar building blocks (csrc/emp_core.h, emp_qp_core.h, emp_frenet_core.h).

The headers are plain C++ that hipcc compiles into the kernels; here g++ compiles the same text
into a test-only library (tests/host_check) so the *logic* - the B-spline reformulation of the path
QP, the banded interior point, the Frenet helpers, the edge arithmetic - is compared with the oracle
without a GPU.  This is host-logic coverage, not the GPU parity proof (tests/test_gpu_*.py).
"""
import ctypes as C

import numpy as np
import pytest

from emplanner_carla_amd import scenes as S
from oracle import exact as ex
from oracle import qp_dense
from oracle import ref_port as op
from tests import host_check
from tests.conftest import assert_rel, load_golden

QP_PRM = np.array([2.0, 1000.0, 3000.0, 150.0, 250.0, 3.0, 3.0, 3.0])   # ds, w_l, w_ddl, w_dddl, w_centre, d1, d2, w


@pytest.fixture(scope="module")
def hc():
    return host_check.load()


def _path_qp(hc, l_min, l_max, start3, prm=QP_PRM, solver="ipm"):
    n = len(l_min)
    l_min = np.ascontiguousarray(l_min, dtype=np.float64)
    l_max = np.ascontiguousarray(l_max, dtype=np.float64)
    prm = np.ascontiguousarray(prm, dtype=np.float64)
    out = [np.zeros(n) for _ in range(3)]
    it = C.c_int(0)
    fn = hc.hc_path_qp if solver == "ipm" else hc.hc_path_qp_gi      # interior point / dual active set
    rc = fn(n, l_min.ctypes.data, l_max.ctypes.data, *[float(v) for v in start3], prm.ctypes.data,
                       out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, C.byref(it))
    return rc, out, it.value


@pytest.mark.parametrize("fname", ["cycle_cfg2_40x9_8obs.npz", "cycle_default_6x12_3obs.npz",
                                   "cycle_cfg1_20x5_0obs.npz", "cycle_default_6x12_3obs_t7.npz"])
@pytest.mark.parametrize("solver", ["ipm", "gi"])
def test_bspline_path_qp_matches_reference_formulation(hc, fname, solver):
    """Banded B-spline solvers (interior point, dual active set) == dense solve of the reference's own (H, f, G, h, Aeq, beq)."""
    g = load_golden(fname)
    n_ok = 0
    for i in range(len(g["seeds"])):
        if np.isnan(g["l_min"][i, 0]):
            continue
        nq = int(g["n_qp"][i])
        rc, (l, dl, ddl), iters = _path_qp(hc, g["l_min"][i, :nq], g["l_max"][i, :nq], g["start"][i, 1:], solver=solver)
        if g["status"][i] == 4:
            assert rc != 0, "oracle says infeasible, banded solver must not claim success"
            continue
        assert rc == 0 and iters <= 40
        assert_rel(l, g["qp_l"][i, :nq], 1e-8, "qp_l", scale=1.0)
        assert_rel(dl, g["qp_dl"][i, :nq], 1e-8, "qp_dl", scale=1.0)
        assert_rel(ddl, g["qp_ddl"][i, :nq], 1e-8, "qp_ddl", scale=1.0)
        n_ok += 1
    assert n_ok >= 6


@pytest.mark.parametrize("solver", ["ipm", "gi"])
def test_path_qp_kkt_certificate_on_random_corridors(hc, solver):
    """Certificate against the reference's dense formulation on corridors the goldens do not hold."""
    rng = np.random.default_rng(11)
    checked = 0
    for trial in range(40):
        n = int(rng.integers(8, 60))
        l_min = -10.0 * np.ones(n)
        l_max = 10.0 * np.ones(n)
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(4, max(5, n - 6)))
            w = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                l_max[a:a + w] = rng.uniform(0.5, 4.0)
            else:
                l_min[a:a + w] = rng.uniform(-4.0, -0.5)
        start3 = (rng.uniform(-0.4, 0.4), rng.uniform(-0.05, 0.05), rng.uniform(-0.01, 0.01))
        rc, (l, dl, ddl), iters = _path_qp(hc, l_min, l_max, start3, solver=solver)
        H, f, G, h, A, b = op.path_qp_matrices(l_min, l_max, *start3)
        ref = qp_dense.solve_qp(H, f, G, h, A, b)
        if ref.status != "optimal":
            assert rc != 0
            continue
        assert rc == 0
        x = np.stack([l, dl, ddl], axis=1).reshape(-1)
        cert = qp_dense.kkt_certificate(H, f, G, h, A, b, x)
        assert cert["stationarity"] < 1e-7 and cert["ineq_violation"] < 1e-9 and cert["eq_violation"] < 1e-9
        assert_rel(x, ref.x, 1e-7, "x vs dense oracle", scale=1.0)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("solver", ["ipm", "gi"])
def test_path_qp_rejects_infeasible_and_tiny(hc, solver):
    import functools
    _path_qp_s = functools.partial(_path_qp, solver=solver)
    n = 12
    rc, _, _ = _path_qp_s(hc, 2.0 * np.ones(n), -2.0 * np.ones(n), (0, 0, 0))       # empty corridor
    assert rc == 1
    rc, _, _ = _path_qp_s(hc, -10 * np.ones(n), 10 * np.ones(n), (9.5, 0.0, 0.0))   # pinned start outside
    assert rc == 1
    lmax = 10 * np.ones(n)
    lmax[n - 1] = -5.0                               # pinned end (l = 0) above the last stations' upper bound
    rc, _, _ = _path_qp_s(hc, -10 * np.ones(n), lmax, (0, 0, 0))
    assert rc == 1
    rc, _, _ = _path_qp_s(hc, -10 * np.ones(3), 10 * np.ones(3), (0, 0, 0))         # n < 4: start/end overlap
    assert rc == 2
    rc, (l, dl, ddl), _ = _path_qp_s(hc, -10 * np.ones(4), 10 * np.ones(4), (0.3, 0.01, 0.0))   # n = 4: no freedom
    assert rc == 0 and abs(l[0] - 0.3) < 1e-15 and l[3] == 0.0


def _box_qp(hc, ref, thr=0.2, w=(0.4, 0.3, 0.3)):
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    out = np.zeros(len(ref))
    it = C.c_int(0)
    rc = hc.hc_box_qp(len(ref), ref.ctypes.data, 1, w[0], w[1], w[2], thr, out.ctypes.data, C.byref(it))
    return rc, out, it.value


def test_box_qp_matches_reference_smoothing(hc):
    """x and y solved separately == the reference's joint 2m-variable QP (unique minimiser)."""
    rng = np.random.default_rng(3)
    for m in (2, 3, 5, 23, 51, 130):
        pts = np.cumsum(rng.normal(0, 1.2, (m, 2)), axis=0) + rng.uniform(-300, 300, 2)
        H, f, G, h = op.smooth_qp_matrices([tuple(p) for p in pts])
        ref = qp_dense.solve_qp(H, f, G, h)
        assert ref.status == "optimal"
        for c in range(2):
            rc, x, iters = _box_qp(hc, pts[:, c])
            assert rc == 0 and iters <= 40
            assert_rel(x, ref.x[c::2], 1e-9, f"coordinate {c}, m={m}", scale=1.0)
    # golden: the smoothed trajectory of the reference cycle (x, y columns)
    g = load_golden("cycle_cfg2_40x9_8obs.npz")
    f0 = load_golden("qp_formulation.npz")
    tgt = (-f0["smooth_q"].reshape(-1) / 0.6)          # f = -2 * 0.3 * x_ref
    for c in range(2):
        rc, x, _ = _box_qp(hc, tgt[c::2])
        m = int(g["traj_len"][0])
        assert rc == 0
        assert_rel(x, g["traj"][0, :m, c], 1e-9, "golden trajectory", scale=1.0)


def test_heading_kappa_and_s_map(hc):
    g = load_golden("functions.npz")
    xy = np.ascontiguousarray(g["hk_xy"])
    th = np.zeros(len(xy))
    kp = np.zeros(len(xy))
    hc.hc_heading_kappa(xy.ctypes.data, len(xy), th.ctypes.data, kp.ctypes.data)
    assert_rel(th, g["hk_theta"], 1e-12, "theta", scale=1.0)
    assert_rel(kp, g["hk_kappa"], 1e-9, "kappa", scale=1.0)
    line = np.ascontiguousarray(g["mp_path"][:80])
    sm = np.zeros(80)
    hc.hc_s_map(line.ctypes.data, 80, 7.3, 2.0, sm.ctypes.data)
    assert_rel(sm, g["sm_out"], 1e-12, "s_map", scale=1.0)
    # matching with both early exits (50 on a first run, 5 in windowed mode)
    full = np.ascontiguousarray(g["mp_path"])
    for (x, y), want in zip(g["mp_pts"], g["mp_index"]):
        assert hc.hc_match(full.ctypes.data, len(full), x, y, 0, 1, 50) == want


def test_segment_cost_bit_exact_with_exact_oracle(hc):
    g = load_golden("cycle_default_6x12_3obs.npz")
    cfg = S.CFG_DEFAULT
    for sd in range(4):
        k = int(g["in_n_obs"][sd])
        obs_s = np.ascontiguousarray(g["obs_s"][sd, :k])
        obs_l = np.ascontiguousarray(g["obs_l"][sd, :k])
        c0, e = ex.edge_costs(obs_s[None], obs_l[None], np.array([k]), g["start"][sd:sd + 1], cfg.row, cfg.col,
                              cfg.sample_s, cfg.sample_l)
        ps, pl, pdl, pddl = g["start"][sd]
        for i in range(cfg.row):
            l1 = ((cfg.row + 1) / 2 - 1 - i) * cfg.sample_l
            v = hc.hc_segment_cost(pl, pdl, pddl, l1, ps, cfg.sample_s, obs_s.ctypes.data, obs_l.ctypes.data, k,
                                   1e12, 300.0, 1000.0, 5000.0, 20.0)
            assert v == c0[0, i]
        for (j, i, kk) in ((1, 0, 0), (3, 11, 2), (5, 4, 9)):
            l0 = ((cfg.row + 1) / 2 - 1 - kk) * cfg.sample_l
            l1 = ((cfg.row + 1) / 2 - 1 - i) * cfg.sample_l
            # neighbour edges: the factorised jerk term (emp_core.h neighbour_cost = what the edge kernels tabulate)
            v = hc.hc_neighbour_cost(l0, l1, ps + j * cfg.sample_s, cfg.sample_s, obs_s.ctypes.data,
                                     obs_l.ctypes.data, k, 1e12, 300.0, 1000.0, 5000.0, 20.0)
            assert v == e[0, j - 1, i, kk]
            # ... which is the general form (start edges) to a few units in the last place
            w = hc.hc_segment_cost(l0, 0.0, 0.0, l1, ps + j * cfg.sample_s, cfg.sample_s, obs_s.ctypes.data,
                                   obs_l.ctypes.data, k, 1e12, 300.0, 1000.0, 5000.0, 20.0)
            assert abs(v - w) <= 1e-13 * abs(w)


# ---- S-T speed DP scalar pieces (csrc/emp_st_core.h; reference planner/speed_planning_test.py) ---------------
def test_st_core_grid_graph_and_edges_vs_reference_golden(hc):
    from oracle import st_speed
    g = load_golden("speed.npz")
    ptr = lambda a: a.ctypes.data
    s_rows, t_cols = np.zeros(40), np.zeros(16)
    hc.hc_st_grid(ptr(s_rows), ptr(t_cols))
    np.testing.assert_array_equal(s_rows, g["s_list"][::-1])
    np.testing.assert_array_equal(t_cols, g["t_list"])
    # generate_st_graph: bit-exact against the reference on all 64 obstacle sets
    for b in range(g["graph_in"].shape[1]):
        ins = [np.ascontiguousarray(g["graph_in"][i, b]) for i in range(4)]
        outs = [np.zeros(16) for _ in range(4)]
        hc.hc_st_graph(16, *[ptr(a) for a in ins], *[ptr(a) for a in outs])
        for i in range(4):
            np.testing.assert_array_equal(outs[i], g["graph_out"][i, b])
    # CalcObsCost / CalcDpCost on the reference's edges
    w4 = np.array([50.0, 4000.0, 100.0, 10000000.0])
    for n, b in enumerate(g["edge_sets"]):
        sets = [np.ascontiguousarray(g["graph_out"][i, b]) for i in range(4)]
        for e, want in zip(g["obs_edges"][n], g["obs_cost"][n]):
            edge = np.array([e[0], e[1], 0.0, e[2], e[3]])
            obs = C.c_double(0)
            hc.hc_st_edge_cost(ptr(w4), ptr(edge), 16, *[ptr(a) for a in sets], C.byref(obs))
            assert abs(obs.value - want) <= 1e-12 * max(1.0, abs(want))
        tab = g["dp_s_dot_table"]
        for rc, want in zip(g["dp_idx"][n], g["dp_cost"][n]):
            origin = rc[0] == 0
            edge = np.array([0.0 if origin else g["s_list"][39 - rc[0]], 0.0 if origin else g["t_list"][rc[1]],
                             7.5 if origin else tab[rc[0], rc[1]], g["s_list"][39 - rc[2]], g["t_list"][rc[3]]])
            got = hc.hc_st_edge_cost(ptr(w4), ptr(edge), 16, *[ptr(a) for a in sets], None)
            assert abs(got - want) <= 1e-12 * abs(want)
    # terminal node rule on the reference's own cost tables
    for n in range(len(g["tables_out"])):
        cost = np.ascontiguousarray(g["tables_out"][n][0])
        r, c = C.c_int(-9), C.c_int(-9)
        assert hc.hc_st_terminal(ptr(cost), C.byref(r), C.byref(c)) == 1
        assert (r.value, c.value) == st_speed.terminal_node(cost)


def test_st_core_pruning_is_exact(hc):
    """Obstacle pruning by bounding boxes must not change any bit: compare with the unpruned NumPy statement."""
    from oracle import st_speed
    rng = np.random.default_rng(11)
    ptr = lambda a: a.ctypes.data
    w4 = np.array([50.0, 4000.0, 100.0, 10000000.0])
    worst = 0.0
    for trial in range(600):
        s_in = rng.uniform(0, 55, 16)
        s_out = s_in + rng.uniform(-5, 25, 16)
        t_in = rng.uniform(0, 7, 16)
        t_out = t_in + rng.uniform(0.5, 6, 16)
        s_in[rng.integers(0, 16, 3)] = np.nan
        s0, s1 = rng.uniform(0, 55, 2)
        t0 = rng.choice(np.arange(0, 8, 0.5))
        if trial % 3 == 0:          # put a few segments right next to the edge: the 1.5 .. 1.6 zone matters
            for j in range(0, 16, 4):
                off = rng.uniform(1.3, 1.8) * rng.choice([-1, 1])
                s_in[j], t_in[j] = s0 + off, t0 + rng.uniform(-0.3, 0.3)
                s_out[j], t_out[j] = s_in[j] + rng.uniform(-3, 3), t_in[j] + rng.uniform(0.2, 3)
        edge = np.array([s0, t0, rng.uniform(0, 20), s1, t0 + 0.5])
        obs = C.c_double(0)
        hc.hc_st_edge_cost(ptr(w4), ptr(edge), 16, ptr(s_in), ptr(s_out), ptr(t_in), ptr(t_out), C.byref(obs))
        want = float(st_speed.exact_obs_cost(s0, t0, s1, t0 + 0.5, s_in, s_out, t_in, t_out, 10000000.0))
        worst = max(worst, abs(obs.value - want) / max(1.0, abs(want)))
    assert worst <= 1e-13


def test_st_core_reach_interval_and_flat_pair_cost(hc):
    """The two pieces the round-3 speed DP kernel adds (emp_st_core.h): the branch-free pair cost equals the reference-order
    one bit for bit, and a pair that costs anything lies strictly inside the segment's reach interval at its sample time
    (so skipping everything outside the interval never changes a result).  Points are drawn around the 1.5 m reach, at
    the segment's ends, along it, and against steep, flat, short and degenerate segments."""
    rng = np.random.default_rng(23)
    n = 400000
    s_in = rng.uniform(0, 55, n)
    t_in = rng.uniform(0, 7, n)
    kind = rng.integers(0, 6, n)
    ds = np.where(kind == 0, 0.0, rng.uniform(-30, 30, n))               # stationary in s
    dt = np.where(kind == 1, rng.uniform(1e-9, 1e-3, n), rng.uniform(0.2, 8, n))   # nearly instantaneous
    ds = np.where(kind == 2, rng.uniform(-1e-6, 1e-6, n), ds)
    both = kind == 3                                                      # degenerate: a point
    ds = np.where(both, 0.0, ds)
    dt = np.where(both, 0.0, dt)
    s_out, t_out = s_in + ds, t_in + dt
    # a point at parameter u along the segment, offset by r across it (or beyond an end)
    u = rng.uniform(-0.3, 1.3, n)
    r = np.where(rng.uniform(size=n) < 0.7, rng.uniform(1.3, 1.7, n), rng.uniform(0, 3, n)) * rng.choice([-1.0, 1.0], n)
    L = np.hypot(ds, dt)
    with np.errstate(invalid="ignore", divide="ignore"):
        nx, ny = np.where(L > 0, -dt / L, 1.0), np.where(L > 0, ds / L, 0.0)
    pt = np.stack([s_in + u * ds + r * nx, t_in + u * dt + r * ny], axis=1)
    seg = np.ascontiguousarray(np.stack([s_in, t_in, s_out, t_out], axis=1))
    pt = np.ascontiguousarray(pt)
    cost, flat, lohi = np.zeros(n), np.zeros(n), np.zeros((n, 2))
    inside = np.zeros(n, np.int32)
    ptr = lambda a: a.ctypes.data
    hc.hc_st_reach(n, 10000000.0, ptr(seg), ptr(pt), ptr(cost), ptr(flat), ptr(inside), ptr(lohi))
    assert np.array_equal(cost, flat, equal_nan=True), "point_cost_flat differs from point_cost"
    costly = cost != 0.0
    assert costly.sum() > 0.2 * n and (~costly).sum() > 0.2 * n          # both sides of the reach are sampled
    assert inside[costly].all(), "a pair with a cost lies outside its reach interval"
    # and the interval is not vacuous: most free pairs that lie across the segment's band are outside it
    assert (inside[~costly] == 0).mean() > 0.5


# --------------------------------------------------------------------------------------
# S-T speed planning back end: the scalar core the kernels call (emp_st_backend_core.h) against the reference's
# golden vectors and oracle/st_backend.py - no GPU involved
# --------------------------------------------------------------------------------------
def test_stb_core_convex_space_and_interp_vs_reference_golden(hc):
    g = load_golden("speed_backend.npz")
    ptr = lambda a: a.ctypes.data
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    ran = 0
    for b in range(len(g["v0"])):
        n = int(g["path_len"][b])
        outs = [np.zeros(16) for _ in range(4)]
        ins = [c(g[k][b]) for k in ("dp_s", "dp_t", "path_index2s", "path_kappa")]
        sets = [c(g[k][b]) for k in ("s_in", "s_out", "t_in", "t_out")]
        st = hc.hc_stb_convex_space(*[ptr(a) for a in ins], n, *[ptr(a) for a in sets], 16, 0.2 * 9.8, *[ptr(a) for a in outs])
        assert st == [0, 2, 4][int(g["cs_raise"][b])]
        if st == 0:
            np.testing.assert_array_equal(np.stack(outs), g["cs_out"][b])
            ran += 1
    assert ran >= 60
    # numpy.interp, knot hits and clamped ends included
    rng = np.random.default_rng(3)
    xp = np.cumsum(rng.uniform(0.5, 1.5, 40))
    fp = rng.normal(size=40)
    xs = np.concatenate([xp[[0, 7, 39]], rng.uniform(xp[0] - 2, xp[-1] + 2, 500), [np.nan]])
    got = np.array([hc.hc_stb_np_interp(ptr(xp), ptr(fp), 40, float(x)) for x in xs])
    np.testing.assert_array_equal(got, np.interp(xs, xp, fp))


def test_stb_core_speed_qp_and_increase_points(hc):
    from oracle import st_backend as be
    g = load_golden("speed_backend.npz")
    ptr = lambda a: a.ctypes.data
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    w4 = np.array([10.0, 50.0, 500.0, 50.0])
    solved = 0
    for b in np.nonzero(g["qp_code"] >= 0)[0]:
        outs = [np.zeros(17) for _ in range(4)]
        iters = C.c_int(0)
        cs = [c(g["cs_out"][b, k]) for k in range(4)]
        dps, dpt = c(g["dp_s"][b]), c(g["dp_t"][b])         # named: the arrays must outlive the call
        st = hc.hc_stb_speed_qp(ptr(dps), ptr(dpt), float(g["v0"][b]), float(g["qp_a0"][b]),
                                *[ptr(a) for a in cs], ptr(w4), *[ptr(a) for a in outs], C.byref(iters))
        if g["qp_code"][b] == 2:
            assert st == 4
            continue
        if np.isnan(g["prof"][b]).all():                  # the dense oracle found the corridor infeasible
            assert st == 8
            continue
        assert st == 0 and 0 < iters.value < 60
        k = int(g["qp_size"][b])
        assert_rel(np.stack(outs)[:, :k], g["prof"][b][:, :k], 1e-6)
        assert np.isnan(np.stack(outs)[:, k:]).all()
        solved += 1
    assert solved >= 25
    for b in np.nonzero(g["dense_raise"] == 0)[0]:
        prof = [c(g["prof"][b, k]) for k in range(4)]
        outs = [np.zeros(401) for _ in range(4)]
        assert hc.hc_stb_increase_points(*[ptr(a) for a in prof], *[ptr(a) for a in outs]) == 0
        np.testing.assert_array_equal(outs[3], g["dense_out"][b, 3])
        assert_rel(np.stack(outs[:3]), g["dense_out"][b, :3], 1e-12, scale=1.0)


def test_host_logic_under_address_and_undefined_behaviour_sanitizers():
    """SURVEY.md section 5 (sanitizers): the scalar cores the kernels are built from, compiled with g++
    -fsanitize=address,undefined and driven with sizes at and beside every capacity, infeasible QPs, NaN / Inf
    coordinates, repeated points, empty and full obstacle slots (tests/host_check/sanitize_main.cpp).  Any report aborts
    the program (-fno-sanitize-recover)."""
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_check")
    exe = os.path.join(here, "_build", "sanitize_main")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", os.path.join(here, "host_check.cpp"),
                    os.path.join(here, "sanitize_main.cpp"), "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "sanitize_main:" in run.stdout


def test_projection_dot_product_rounds_like_numpy(hc):
    """emp_frenet_core.h dot2 = fma(a1, b1, a0 * b0): bit for bit what numpy.dot returns for two 2-vectors on an x86 host
    whose OpenBLAS accumulates with fused multiply-adds (the reference evaluates every projection with np.dot,
    planning_utils.py:107, :443, :507, :546-578).  The fused form itself is checked against libm's fma on any host; numpy
    only where its dot is of that form (another BLAS may round the plain way - that is the reference's host dependence,
    HISTORY.md section 4)."""
    import ctypes
    rng = np.random.default_rng(11)
    n = 20000
    a = np.ascontiguousarray(rng.normal(0, 30, (n, 2)))
    b = np.ascontiguousarray(rng.normal(0, 1, (n, 2)))
    out = np.zeros(n)
    hc.hc_dot2(n, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    libm = ctypes.CDLL("libm.so.6")
    libm.fma.restype = ctypes.c_double
    libm.fma.argtypes = [ctypes.c_double] * 3
    fused = np.array([libm.fma(a[k, 1], b[k, 1], a[k, 0] * b[k, 0]) for k in range(n)])
    np.testing.assert_array_equal(out, fused)
    plain = a[:, 0] * b[:, 0] + a[:, 1] * b[:, 1]
    assert (fused != plain).mean() > 0.1                       # the two forms really differ in the last bit
    dots = np.array([np.dot(a[k], b[k]) for k in range(n)])
    if (dots == plain).all():
        pytest.skip("this host's numpy rounds 2-vector dot products the plain way")
    np.testing.assert_array_equal(out, dots)
