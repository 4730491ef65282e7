"""Generate the golden vectors under tests/golden/*.npz from the IMPORTED reference.

Run in the build container only (needs /root/reference, which never travels):

    python -B tests/golden/make_golden.py

Every array written here is either an input synthesised by ``emplanner_carla_amd.scenes`` or
an output of the reference's own functions (reference planner/path_planning.py and
planner/planning_utils.py, imported through ``ref_loader`` with stub ``carla``/``cvxopt``
modules).  The per-cycle call sequence replays reference test_9.py:113-218 (the body of
``motion_planning`` after the reference line is available).  No reference source text is
stored - only numbers.

QP note: the two ``cvxopt.solvers.qp`` calls are served by ``oracle/qp_dense.py`` through
the stub (cvxopt is absent; see ref_loader).  The dense matrices the reference built are
recorded for a few cases (``qp_formulation.npz``) so the tests can pin the formulation.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
#: where the .npz files go: this directory, or a scratch one for tests/test_reference_live.py (regenerate and compare)
OUT = os.environ.get("EMP_GOLDEN_OUT", HERE)
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from emplanner_carla_amd import scenes as S  # noqa: E402
from oracle import qp_dense  # noqa: E402

warnings.filterwarnings("ignore", category=DeprecationWarning)

NPTS = 96      # padded length of per-scene path arrays
NTRAJ = 64     # padded length of trajectories


def pad(a, n, fill=np.nan):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    out = np.full(n, fill)
    out[:len(a)] = a
    return out


def tl(a):
    return [tuple(float(v) for v in r) for r in a]


def run_cycle(pp, pu, sc, cfg, decimate=2, use_qp=True, midpoint=True):
    """Replay of reference test_9.py:113-218 on one synthetic scene; returns dict of arrays."""
    ref = tl(sc.ref)
    kw = dict(sampling_res=cfg.sampling_res, row=cfg.row, col=cfg.col, sample_s=cfg.sample_s,
              sample_l=cfg.sample_l, w_collision_cost=cfg.w_collision_cost,
              w_smooth_cost=list(cfg.w_smooth_cost), w_reference_cost=cfg.w_reference_cost)
    out = {}
    s_map = pu.cal_s_map_fun(ref, origin_xy=tuple(sc.origin_xy))
    if len(sc.obs_xy):
        obs_s, obs_l = pu.cal_s_l_fun(tl(sc.obs_xy), ref, s_map)
    else:
        obs_s, obs_l = [], []
    begin_s, begin_l = pu.cal_s_l_fun([tuple(sc.start_xy)], ref, s_map)
    deri = pu.cal_s_l_deri_fun([tuple(sc.start_xy)], [tuple(sc.start_v)], [tuple(sc.start_a)], ref,
                               tuple(sc.start_xy))
    l0, dl0, ddl0 = deri[0][0], deri[4][0], deri[6][0]
    n_qp0 = len(ref_loader.QP_LOG)
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        dp_s, dp_l = pp.DP_algorithm(list(obs_s), list(obs_l), begin_s[0], l0, dl0, ddl0, **kw)
    out["dp_infeasible_banner"] = float("can't find a feasible path" in buf.getvalue())
    out.update(s_map=np.asarray(s_map), obs_s=pad(obs_s, max(cfg.n_obs, 1)), obs_l=pad(obs_l, max(cfg.n_obs, 1)),
               begin=np.array([begin_s[0], begin_l[0]]), start=np.array([begin_s[0], l0, dl0, ddl0]),
               deri=np.array([d[0] for d in deri], dtype=np.float64),
               dp_len=float(len(dp_s)), dp_s=pad(dp_s, NPTS), dp_l=pad(dp_l, NPTS))
    d_s, d_l = dp_s[::decimate], dp_l[::decimate]
    status = 0.0  # 0 ok, 3 bound index out of range (IndexError), 4 QP not optimal
    qp_l = list(d_l)
    if use_qp:
        try:
            l_min, l_max = pp.cal_lmin_lmax(d_s, d_l, list(obs_s), list(obs_l), cfg.obs_length, cfg.obs_width)
        except IndexError:
            status = 3.0
            l_min = l_max = None
        if l_min is not None:
            out.update(l_min=pad(l_min, NPTS), l_max=pad(l_max, NPTS))
            qp_l, qp_dl, qp_ddl = pp.Quadratic_planning(l_min, l_max, l0, dl0, ddl0)
            rec = ref_loader.QP_LOG[-1]
            if rec["status"] != "optimal":
                # cvxopt would hand back its last iterate here; what the stand-in solver's last iterate is depends on the
                # checker's internals (oracle/qp_dense.py) and nothing compares it: blank it, so that the fixture is a
                # function of the reference and the scene alone
                status = 4.0
            else:
                out.update(qp_l=pad(qp_l, NPTS), qp_dl=pad(qp_dl, NPTS), qp_ddl=pad(qp_ddl, NPTS),
                           qp_stationarity=rec["stationarity"], qp_violation=rec["violation"])
    out["n_qp"] = float(len(d_s))
    if status == 0.0:
        if midpoint:
            path_s = [d_s[0]] + [(d_s[i] + d_s[i - 1]) / 2 for i in range(1, len(qp_l))] + [d_s[-1]]
            path_l = [qp_l[0]] + [(qp_l[i] + qp_l[i - 1]) / 2 for i in range(1, len(qp_l))] + [qp_l[-1]]
        else:
            path_s, path_l = list(d_s), list(qp_l)
        traj = pp.frenet_2_x_y_theta_kappa(begin_s[0], begin_l[0], path_s, path_l, ref, s_map)
        rec = ref_loader.QP_LOG[-1]
        if rec["status"] != "optimal":
            status = 5.0
        t = np.full((NTRAJ, 4), np.nan)
        t[:len(traj)] = np.asarray(traj, dtype=np.float64)
        out.update(path_s=pad(path_s, NPTS), path_l=pad(path_l, NPTS), traj=t, traj_len=float(len(traj)),
                   smooth_stationarity=rec["stationarity"])
    out["status"] = status
    out["n_qp_calls"] = float(len(ref_loader.QP_LOG) - n_qp0)
    return out


def stack(dicts, keys_shapes):
    res = {}
    for k, shape in keys_shapes.items():
        arrs = []
        for d in dicts:
            v = d.get(k)
            arrs.append(np.full(shape, np.nan) if v is None else np.asarray(v, dtype=np.float64).reshape(shape))
        res[k] = np.stack(arrs)
    return res


def cycles_for(pp, pu, cfg, seeds, scene_kw=None, per_seed=None, **mode):
    """``scene_kw`` / ``per_seed``: options of ``scenes.make_scene`` for every scene / as a function of the seed."""
    scene_kw = dict(scene_kw or {})
    scenes = [S.make_scene(int(s), cfg, **{**scene_kw, **(per_seed(int(s)) if per_seed else {})}) for s in seeds]
    outs = [run_cycle(pp, pu, sc, cfg, **mode) for sc in scenes]
    mo = max(cfg.n_obs, 1)
    shapes = dict(s_map=(cfg.n_ref,), obs_s=(mo,), obs_l=(mo,), begin=(2,), start=(4,), deri=(7,), dp_len=(),
                  dp_s=(NPTS,), dp_l=(NPTS,), l_min=(NPTS,), l_max=(NPTS,), qp_l=(NPTS,), qp_dl=(NPTS,),
                  qp_ddl=(NPTS,), qp_stationarity=(), qp_violation=(), n_qp=(), path_s=(NPTS,), path_l=(NPTS,),
                  traj=(NTRAJ, 4), traj_len=(), smooth_stationarity=(), status=(), dp_infeasible_banner=(),
                  n_qp_calls=())
    res = stack(outs, shapes)
    batch = S.make_batch(seeds, cfg, per_seed=per_seed, **scene_kw)
    res.update(seeds=batch.seeds, in_ref=batch.ref, in_origin_xy=batch.origin_xy, in_start_xy=batch.start_xy,
               in_start_v=batch.start_v, in_start_a=batch.start_a, in_obs_xy=batch.obs_xy, in_n_obs=batch.n_obs)
    return res


def edge_tensor(pp, cfg, obs_s, obs_l, start):
    """Reference cal_start_cost / cal_neighbor_cost for every lattice edge of one scene."""
    row, col = cfg.row, cfg.col
    w = (cfg.w_collision_cost, list(cfg.w_smooth_cost), cfg.w_reference_cost)
    c0 = np.zeros(row)
    e = np.zeros((col - 1, row, row))
    ps, pl, pdl, pddl = (float(v) for v in start)
    for i in range(row):
        c0[i] = float(np.asarray(pp.cal_start_cost(obs_s, obs_l, ps, pl, pdl, pddl, i, row, cfg.sample_s,
                                                   cfg.sample_l, *w)).reshape(-1)[0])
    for j in range(1, col):
        for i in range(row):
            for k in range(row):
                cur_s = ps + (j + 1) * cfg.sample_s
                pre_s = ps + j * cfg.sample_s
                cur_l = ((row + 1) / 2 - 1 - i) * cfg.sample_l
                pre_l = ((row + 1) / 2 - 1 - k) * cfg.sample_l
                e[j - 1, i, k] = float(np.asarray(pp.cal_neighbor_cost(obs_s, obs_l, pre_s, pre_l, cur_s, cur_l,
                                                                       cfg.sample_s, *w)).reshape(-1)[0])
    return c0, e


def function_level(pp, pu):
    """Small known-input/known-output vectors for the individual functions of SURVEY.md 8a."""
    rng = np.random.default_rng(20240607)
    g = {}
    # a6 cal_quintic_coefficient: boundary conditions + the polynomial sampled on the segment
    bc = []
    vals = []
    for _ in range(24):
        s0 = rng.uniform(0.0, 90.0)
        T = rng.choice([2.5, 5.0, 15.0])
        b = [rng.uniform(-3, 3), rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05), rng.uniform(-6, 6),
             rng.uniform(-0.1, 0.1), rng.uniform(-0.02, 0.02), s0, s0 + T]
        c = pu.cal_quintic_coefficient(*b)
        ts = np.linspace(b[6], b[7], 11)
        vals.append(np.polyval(np.asarray(c, dtype=np.float64)[::-1], ts))
        bc.append(b)
    g["quintic_bc"] = np.asarray(bc)
    g["quintic_vals"] = np.asarray(vals)
    # a4 cal_obs_cost on hand-built rows hitting every branch and boundary (d2 = 16, 36)
    sq = np.array([
        [50, 40, 37, 36, 36.0001, 35.9, 30, 20, 17, 16.0001],
        [50, 40, 30, 16, 10, 5, 1, 0.5, 30, 30],
        [100, 100, 100, 100, 100, 100, 100, 100, 100, 100],
        [15.9, 20, 20, 20, 20, 20, 20, 20, 20, 20],
        [20, 20, 20, 20, 20, 20, 20, 20, 20, 16.0],
        [35.999, 36.0, 36.001, 16.001, 25, 25, 25, 25, 25, 25]], dtype=np.float64)
    g["obs_sq"] = sq
    g["obs_cost"] = np.array([float(pp.cal_obs_cost(1e12, r.reshape(10, 1))) for r in sq])
    g["obs_cost_w3"] = np.array([float(pp.cal_obs_cost(7.5, r.reshape(10, 1), danger_dis=3, safe_dis=5))
                                 for r in sq])
    # a13 cal_heading_kappa on a wiggly polyline that crosses the +-pi branch cut
    t = np.linspace(0, 4.0, 37)
    xy = np.stack([-3.0 * t + 0.3 * np.sin(3 * t), 0.8 * np.sin(1.7 * t) - 0.2 * t], axis=1)
    th, kp = pu.cal_heading_kappa(tl(xy))
    g["hk_xy"] = xy
    g["hk_theta"] = np.asarray(th)
    g["hk_kappa"] = np.asarray(kp)
    # a14 / a19 matching on a long path (so the 50- and 5-step early exits trigger)
    tt = np.arange(300) * 1.0
    path_xy = np.stack([tt, 20.0 * np.sin(tt / 40.0)], axis=1)
    pth, pk = pu.cal_heading_kappa(tl(path_xy))
    path = np.column_stack([path_xy, pth, pk])
    pts = np.array([[5.2, 3.0], [120.4, -7.0], [250.0, 10.0], [60.0, 30.0], [-4.0, 1.0], [299.5, 18.0]])
    mi, pr = pu.match_projection_points(tl(pts), tl(path))
    g["mp_path"] = path
    g["mp_pts"] = pts
    g["mp_index"] = np.asarray(mi, dtype=np.int64)
    g["mp_proj"] = np.asarray(pr, dtype=np.float64)
    fm = []
    for first, pre in ((True, 0), (False, 100), (False, 130), (False, 3), (False, 290)):
        m, p = pu.find_match_points(tl(pts[:3]), tl(path), first, pre)
        fm.append(np.concatenate([np.asarray(m, dtype=np.float64), np.asarray(p, dtype=np.float64).reshape(-1)]))
    g["fm_out"] = np.asarray(fm)
    g["fm_modes"] = np.array([[1, 0], [0, 100], [0, 130], [0, 3], [0, 290]], dtype=np.float64)
    # sampling: (match index -> first index, length) near both ends and in the middle
    samp = []
    for m in (0, 4, 10, 150, 259, 270, 299):
        loc = pu.sampling(m, tl(path), back_length=10, forward_length=50)
        samp.append([m, len(loc), loc[0][0], loc[-1][0]])
    g["sampling"] = np.asarray(samp, dtype=np.float64)
    # a15/a16/a17 on the same long path
    s_map = pu.cal_s_map_fun(tl(path[:80]), origin_xy=(7.3, 2.0))
    g["sm_out"] = np.asarray(s_map)
    s_l = pu.cal_s_l_fun(tl(pts[:1]) + [(30.0, 12.0), (70.0, 25.0)], tl(path[:80]), s_map)
    g["sl_pts"] = np.array([pts[0], [30.0, 12.0], [70.0, 25.0]])
    g["sl_out"] = np.asarray(s_l, dtype=np.float64)
    # a11 cal_proj_point / cal_proj_point_1 walking forward
    pp_out = []
    idx = 0
    for s in (0.5, 3.0, 3.1, 17.9, 44.4, 60.0):
        r = pp.cal_proj_point(s, idx, tl(path[:80]), s_map)
        r1 = pu.cal_proj_point_1(s, idx, tl(path[:80]), s_map)
        assert tuple(r) == tuple(r1)
        idx = r[4]
        pp_out.append([s] + [float(v) for v in r])
    g["projpt"] = np.asarray(pp_out)
    # a18 with zero speed (the |ds| < 1e-6 branch) and with speed
    d1 = pu.cal_s_l_deri_fun([(30.0, 12.0)], [(0.0, 0.0)], [(0.3, -0.2)], tl(path[:80]), (30.0, 12.0))
    d2 = pu.cal_s_l_deri_fun([(30.0, 12.0), (50.0, 20.0)], [(6.0, 2.0), (5.0, 1.0)], [(0.3, -0.2), (0.1, 0.4)],
                             tl(path[:80]), (31.0, 12.5))
    g["deri_zero"] = np.asarray([v[0] for v in d1], dtype=np.float64)
    g["deri_two"] = np.asarray(d2, dtype=np.float64)
    # a5 enrich_DP_s_l with integer sample_s where int() truncation flips the sample count
    en = []
    for ps in (0.0, 1.9999976, 2.0000004, 3.3, 7.1):
        DP_s = [ps + (i + 1) * 15 for i in range(6)]
        DP_l = [0.0, 1.5, 1.5, -3.0, 0.0, 0.0]
        for res in (1, 2, 0.5):
            es, el = pp.enrich_DP_s_l(DP_s, DP_l, ps, 0.2, 0.01, -0.003, resolution=res)
            en.append(np.concatenate([[ps, res, len(es)], pad(es, 200), pad(el, 200)]))
    g["enrich"] = np.asarray(en)
    # helper functions beside the path
    fx, fy, fh, fk = path[:80, 0], path[:80, 1], path[:80, 2], path[:80, 3]
    idx2s = pu.trajectory_index2s(np.append(fx, np.nan), np.append(fy, np.nan))
    g["idx2s"] = np.asarray(idx2s)
    sset = np.array([2.0, 10.5, 33.3, np.nan, 40.0])
    lset = np.array([0.5, -1.0, 2.0, 0.0, 0.0])
    dls = np.array([0.1, -0.05, 0.0, 0.0, 0.0])
    ddls = np.array([0.01, 0.0, -0.02, 0.0, 0.0])
    f2c = pu.Frenet2Cartesian(sset, lset, dls, ddls, fx, fy, fh, fk, idx2s[:80])
    g["f2c_in"] = np.stack([sset, lset, dls, ddls])
    g["f2c_out"] = np.stack([a[:5, 0] for a in f2c])
    cp = pu.CalcProjPoint(21.7, fx, fy, fh, fk, idx2s[:80])
    g["calcproj"] = np.asarray([21.7] + [float(v) for v in cp])
    dy = pu.cal_dy_obs_deri(np.array([1.0, -2.0, np.nan]), np.array([5.0, 0.0, 1.0]), np.array([1.0, 0.0, 1.0]),
                            np.array([0.1, 0.2, 0.3]), np.array([0.01, -0.02, 0.0]))
    g["dyobs"] = np.stack([a[:4] for a in dy])
    # (round 5) cal_obs_cost on rows that are NOT ten samples long (the reference loops over whatever it gets, :601), and
    # cal_neighbor_cost with an end station that is not pre_node_s + sample_s (the quintic ends at cur_node_s, :553, the
    # samples step by sample_s / 10, :565-566) - what a caller of the drop-in functions may do and the DP never does
    r2 = np.random.default_rng(20250930)
    for n in (7, 23):
        sq = r2.uniform(10.0, 50.0, (5, n))
        sq[1, n // 2] = 15.0
        sq[3, 0] = 16.0
        g[f"obs_sq{n}"] = sq
        g[f"obs_cost{n}"] = np.array([float(pp.cal_obs_cost(1e12, r.reshape(n, 1))) for r in sq])
    cases, costs = [], []
    obs_s, obs_l = [22.0, 31.5, 60.0], [1.0, -2.5, 4.0]
    for _ in range(12):
        pre_s = r2.uniform(5.0, 60.0)
        sample_s = float(r2.choice([2.5, 5.0, 15.0]))
        span = sample_s * r2.uniform(0.6, 1.7)
        pre_l, cur_l = r2.uniform(-4, 4), r2.uniform(-4, 4)
        c = pp.cal_neighbor_cost(obs_s, obs_l, pre_s, pre_l, pre_s + span, cur_l, sample_s, 1e12, [300, 1000, 5000], 20)
        cases.append([pre_s, pre_l, pre_s + span, cur_l, sample_s])
        costs.append(float(np.asarray(c).reshape(-1)[0]))
    g["nbr_general_in"] = np.asarray(cases)
    g["nbr_general_obs"] = np.array([obs_s, obs_l])
    g["nbr_general_cost"] = np.asarray(costs)
    return g


def main():
    pp, pu = ref_loader.load_reference()
    outdir = OUT
    # ---- per-config full cycles (test_9 form) -------------------------------------------
    plan = ((S.CFG1, range(8)), (S.CFG_DEFAULT, range(16)), (S.CFG2, range(32)))
    for cfg, seeds in plan:
        res = cycles_for(pp, pu, cfg, list(seeds))
        np.savez_compressed(os.path.join(outdir, f"cycle_{cfg.name}.npz"), **res)
        st = res["status"]
        print(cfg.name, "scenes", len(st), "status counts", {int(k): int((st == k).sum()) for k in np.unique(st)},
              "dp infeasible", int(res["dp_infeasible_banner"].sum()))
    # ---- the metric's lattice on SURVEY.md 8(d)'s own geometry (arc radii 150-1000 m, half of the scenes with the
    # survey's slalom layout, half of them started off the reference-line nodes), and the first scenes of the
    # benchmark batch (gentle arcs, corridor layout, start off the nodes: scenes.BENCH_START_AHEAD)
    # ... and its first scenes with the "worst" obstacle layout (every obstacle within reach of the same columns: edges with
    # several obstacles in reach at once - the several-obstacle ring of the edge kernel against the reference itself)
    for tag, kw in (("tight", dict(per_seed=S.survey_geometry_kwargs)),
                    ("bench", dict(scene_kw=dict(start_ahead=S.BENCH_START_AHEAD))),
                    ("worst", dict(scene_kw=dict(start_ahead=S.BENCH_START_AHEAD, dist="worst")))):
        res = cycles_for(pp, pu, S.CFG2, list(range(32)), **kw)
        np.savez_compressed(os.path.join(outdir, f"cycle_{S.CFG2.name}_{tag}.npz"), **res)
        st = res["status"]
        print(S.CFG2.name, tag, "status counts", {int(k): int((st == k).sum()) for k in np.unique(st)},
              "dp infeasible", int(res["dp_infeasible_banner"].sum()))
    # ---- driver variants on the default lattice (test_7: no decimation / no midpoint; test_5/6: no QP)
    for tag, mode in (("t7", dict(decimate=1, use_qp=True, midpoint=False)),
                      ("t6", dict(decimate=1, use_qp=False, midpoint=False))):
        res = cycles_for(pp, pu, S.CFG_DEFAULT, list(range(6)), **mode)
        np.savez_compressed(os.path.join(outdir, f"cycle_{S.CFG_DEFAULT.name}_{tag}.npz"), **res)
        print("variant", tag, "status", res["status"])
    # ---- edge tensors -------------------------------------------------------------------
    edges = {}
    for cfg, seeds in ((S.CFG_DEFAULT, (0, 1, 2)), (S.CFG2, (0, 9))):
        cyc = np.load(os.path.join(outdir, f"cycle_{cfg.name}.npz"))
        for sd in seeds:
            k = int(cyc["in_n_obs"][sd])
            c0, e = edge_tensor(pp, cfg, list(cyc["obs_s"][sd, :k]), list(cyc["obs_l"][sd, :k]), cyc["start"][sd])
            edges[f"{cfg.name}__{sd}__c0"] = c0
            edges[f"{cfg.name}__{sd}__e"] = e
    np.savez_compressed(os.path.join(outdir, "edges.npz"), **edges)
    print("edge tensors", sorted(edges))
    # ---- QP formulation pins ------------------------------------------------------------
    ref_loader.QP_LOG.clear()
    sc = S.make_scene(0, S.CFG2)
    run_cycle(pp, pu, sc, S.CFG2)
    form = {}
    for name, rec in zip(("path", "smooth"), ref_loader.QP_LOG[-2:]):
        for k in ("P", "q", "G", "h", "A", "b", "x"):
            if rec[k] is not None:
                form[f"{name}_{k}"] = rec[k]
        cert = qp_dense.kkt_certificate(rec["P"], rec["q"], rec["G"], rec["h"], rec["A"], rec["b"], rec["x"])
        form[f"{name}_cert"] = np.array([cert["stationarity"], cert["ineq_violation"], cert["eq_violation"]])
        print("QP", name, rec["P"].shape, rec["G"].shape, cert)
    np.savez_compressed(os.path.join(outdir, "qp_formulation.npz"), **form)
    # ---- function-level vectors ---------------------------------------------------------
    np.savez_compressed(os.path.join(outdir, "functions.npz"), **function_level(pp, pu))
    for f in sorted(os.listdir(outdir)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(outdir, f)), "bytes")


if __name__ == "__main__":
    main()
