"""BASELINE.json's full sizes: configs[2] (4096 scenes on one GPU) and configs[3] (32 768 scenes, here on one GPU
and as the contiguous shards the ranks of a 2/4/8-GPU run take).

The oracle cannot plan 32 768 scenes in seconds, so the full batches are checked through
  * the committed reference outputs of seeds 0..31, which sit inside both batches (other tile positions and
    neighbours than in the 32-scene run that the small-batch parity tests do),
  * the exact DP oracle, index for index, on the whole 4096 batch and on a random sample of the 32 768 one,
  * size-independent properties: a permuted batch gives bit-identical per-scene results (scenes are independent:
    nothing leaks between the scenes that share a wavefront or a tile), a shard planned alone equals its slice of
    the whole batch (what the multi-GPU run relies on), a repeated call is bit-identical, and every planned
    trajectory honours the path QP's pinned end state and the smoothing box.
"""
from __future__ import annotations

import os
import numpy as np
import pytest

from emplanner_carla_amd import scenes as S
from oracle import exact as ex
from tests.conftest import assert_dp_l_vs_reference, assert_rel, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-6
#: the benchmark batch (bench.py's default workload): gentle arcs, corridor layout, planning start OFF the reference-line
#: nodes.  Until round 4 it started ON node 6 (start_ahead = 2.0), where `s_map[idx + 1] < s` (path_planning.py:63) is a
#: tie decided by the host's libm: that batch is kept as the dedicated tie test below.
BENCH = dict(start_ahead=S.BENCH_START_AHEAD)
OUTPUTS = ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status")


@pytest.fixture(scope="module")
def planner():
    from emplanner_carla_amd.api import Planner
    pl = Planner(0)
    yield pl
    pl.close()


def _params(cfg):
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    return dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()


def _host_inputs(b):
    B, P = b.ref.shape[:2]
    return dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy,
                start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)


def _plan_resident(planner, cfg, host, index=None):
    """Plan with inputs resident in HBM (torch tensors); returns NumPy copies of every output."""
    import torch
    p, q, sp = _params(cfg)
    sel = (lambda a: a) if index is None else (lambda a: a[index])
    dev = {k: torch.from_numpy(np.ascontiguousarray(sel(v))).cuda() for k, v in host.items()}
    r = planner.plan_cycle(p, q, sp, **dev)
    planner.synchronize()
    return {k: getattr(r, k).cpu().numpy() for k in OUTPUTS}


def _masked(out):
    """Outputs with the padding beyond each scene's length zeroed (padding content is unspecified)."""
    res = dict(out)
    for arr, ln in (("dp_s", "dp_len"), ("dp_l", "dp_len"), ("path_s", "path_len"), ("path_l", "path_len"),
                    ("traj", "traj_len")):
        a = out[arr].copy()
        idx = np.arange(a.shape[1])[None, :] >= out[ln][:, None]
        a[idx] = 0.0
        res[arr] = a
    return res


def _assert_same(a, b, what):
    a, b = _masked(a), _masked(b)
    for k in OUTPUTS:
        ok = (a["status"] & ~1) == 0                       # refused scenes: only the status is specified
        x, y = (a[k], b[k]) if k in ("status", "dp_rows") else (a[k][ok], b[k][ok])
        if not np.array_equal(x, y):
            rows = np.nonzero(np.any(x.reshape(len(x), -1) != y.reshape(len(y), -1), axis=1))[0]
            sel = np.arange(len(a["status"])) if k in ("status", "dp_rows") else np.nonzero(ok)[0]
            i = int(sel[rows[0]])                                  # scene index in the batch
            detail = "; ".join(f"{f}: {np.array2string(np.asarray(a[f][i]).ravel()[:48], precision=17)} vs "
                               f"{np.array2string(np.asarray(b[f][i]).ravel()[:48], precision=17)}"
                               for f in (k, "status", "dp_len", "path_len", "traj_len"))
            raise AssertionError(f"{what}: {k} differs in {rows.size} of {len(x)} scenes, scenes {sel[rows[:8]].tolist()}; "
                                 f"scene {i}: {detail}")


def _check_golden_subset(out, fname="cycle_cfg2_40x9_8obs_bench.npz", min_checked=20):
    """The imported reference's own outputs for seeds 0..31 (tests/golden/make_golden.py), which are the head of `out`."""
    g = load_golden(fname)
    n = len(g["seeds"])
    assert np.array_equal(g["seeds"], np.arange(n)), "the fixture holds seeds 0..n-1, the head of the full batch"
    checked = 0
    for b in range(n):
        k = int(g["dp_len"][b])
        assert out["dp_len"][b] == k
        assert_dp_l_vs_reference(out["dp_l"][b, :k], g["dp_l"][b, :k], f"scene {b}: DP path")
        assert bool(out["status"][b] & 1) == bool(g["dp_infeasible_banner"][b])
        if g["status"][b] != 0:
            assert out["status"][b] & {3: 4, 4: 8, 5: 16}[int(g["status"][b])], f"scene {b}: refused for the reference's reason"
            continue
        assert (out["status"][b] & ~1) == 0, f"scene {b}: status {out['status'][b]}"
        m = int(g["traj_len"][b])
        assert out["traj_len"][b] == m
        assert_rel(out["traj"][b, :m, :3], g["traj"][b, :m, :3], RTOL, f"scene {b} trajectory")
        assert_rel(out["traj"][b, :m, 3], g["traj"][b, :m, 3], RTOL, f"scene {b} curvature")
        checked += 1
    assert checked >= min_checked


def _check_properties(planner, cfg, host, out):
    ok = (out["status"] & ~1) == 0
    assert 0.8 < ok.mean() < 0.95, "the generator's share of walls / blocked corridors"
    assert ((out["status"] & 1) != 0).any()
    n, m = out["path_len"][ok], out["traj_len"][ok]
    assert (m == n + 1).all() and (m == 23).all()
    last = out["path_l"][ok, n - 1]
    assert np.abs(last).max() < 1e-9, "pinned end state of the path QP"
    traj = out["traj"][ok][:, :23]
    assert np.isfinite(traj).all()
    sm, _, _, bsl, _ = planner.frenet_project(**host)
    tgt, cnt, st = planner.frenet_path_to_xy(host["ref_line"], sm, host["n_ref"], bsl, out["path_s"], out["path_l"],
                                             out["path_len"])
    assert (np.abs(traj[:, :, :2] - tgt[ok][:, :23]) <= 0.2 + 1e-9).all(), "smoothed point left its box"
    assert (np.abs(traj[:, 2:, 3]) < 0.5).all()


def test_configs2_4096_scenes(planner):
    cfg = S.CFG2
    B = 4096
    batch = S.make_batch(range(B), cfg, **BENCH)
    host = _host_inputs(batch)
    out = _plan_resident(planner, cfg, host)
    _check_golden_subset(out)
    # the DP of the whole batch, index for index, against the exact oracle (projection -> DP through the cycle's
    # own obstacle projection is covered by the golden subset; here the S-L inputs are the generator's)
    from emplanner_carla_amd.api import dp_params_from_cfg
    rows, mc, st = planner.dp_plan(dp_params_from_cfg(cfg), batch.sl_obs_s, batch.sl_obs_l, batch.n_obs, batch.sl_start)
    xrows, xfeas, _ = ex.dp_plan(batch.sl_obs_s, batch.sl_obs_l, batch.n_obs, batch.sl_start, cfg.row, cfg.col,
                                 cfg.sample_s, cfg.sample_l, cfg.sampling_res)
    assert np.array_equal(rows, xrows)
    assert np.array_equal(st == 1, ~xfeas)
    # repeat + permutation
    _assert_same(out, _plan_resident(planner, cfg, host), "repeated call")
    perm = np.random.default_rng(7).permutation(B)
    outp = _plan_resident(planner, cfg, host, perm)
    _assert_same({k: v[perm] for k, v in out.items()}, outp, "permuted batch")
    _check_properties(planner, cfg, host, out)


def _port_cycle(cfg, batch, i, **kw):
    from oracle import ref_port as op
    nk = int(batch.n_obs[i])
    return op.plan_cycle(batch.ref[i], tuple(batch.origin_xy[i]), tuple(batch.start_xy[i]), tuple(batch.start_v[i]),
                         tuple(batch.start_a[i]), [tuple(o) for o in batch.obs_xy[i, :nk]],
                         dp_kwargs=dict(row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l,
                                        sampling_res=cfg.sampling_res), obs_length=cfg.obs_length, obs_width=cfg.obs_width,
                         verbose=False, **kw)


def test_knot_tie_scenes_of_the_on_node_batch_are_the_other_branch_and_nothing_else(planner):
    """The dedicated tie test: seeds 177 and 3204 of the ON-NODE batch (start_ahead = 2.0; rounds 1-4 benchmarked it)
    and a few of their neighbours as controls.  There the scene generator puts the planning start on the normal through reference-line node 6, so `while s_map[idx + 1] < s`
    (path_planning.py:62-63) compares two numbers that agree to an ulp and the segment the first trajectory points are
    extrapolated from is decided by the last bit of cos / sin / dot on the machine at hand (0.45 mm apart: kappa ds^2).
    On those two scenes the device lands on the other side than the port on this round's host.  What must hold, whatever
    the host's libm does: each scene is within SURVEY 8(d)'s rule of the port - or, where the start IS tied, of the
    port with that one comparison answered the other way (oracle/ref_port.py `_flip_ties`); a scene that is beyond
    tolerance of both is a failure."""
    cfg = S.CFG2
    seeds = [176, 177, 178, 3203, 3204, 3205, 5, 1024]
    batch = S.make_batch(seeds, cfg, start_ahead=2.0)
    out = _plan_resident(planner, cfg, _host_inputs(batch))
    tie_tol = 8e-15
    flipped_needed = 0
    compared = 0
    for i, seed in enumerate(seeds):
        port = _port_cycle(cfg, batch, i)
        ok = port.get("qp_status", "optimal") == "optimal" and port["smooth_status"] == "optimal"
        assert ok == ((int(out["status"][i]) & ~1) == 0), f"seed {seed}: outcome"
        if not ok:
            continue
        want = np.asarray(port["trajectory"], dtype=np.float64)
        m = len(want)
        assert out["traj_len"][i] == m
        got = out["traj"][i, :m]
        gap = float(np.abs(np.asarray(port["s_map"]) - port["begin_s"]).min())
        assert gap <= tie_tol, f"seed {seed}: the generator no longer puts the start on a node (gap {gap:.2e}); this test needs one"
        near = np.abs(got - want) <= np.maximum(1e-6 * np.abs(want), 1e-9)
        if not near.all():
            other = np.asarray(_port_cycle(cfg, batch, i, _flip_ties=tie_tol)["trajectory"], dtype=np.float64)
            assert len(other) == m
            assert_rel(got, other, RTOL, f"seed {seed}: beyond tolerance of the port AND of its flipped tie branch")
            # and the two branches really are what separates them: sub-millimetre, at the first points (the smoothing QP
            # carries a decaying trace of it along the first dozen)
            # (positions only: curvature is a second difference of them and keeps the trace visible at 1e-6 of its
            # 1e-3 1/m scale further along)
            assert np.abs(want - other)[:, :2].max() < 2e-3 and not near[:2].all() and near[m // 2:, :2].all()
            flipped_needed += 1
        compared += 1
    assert compared >= 6
    print(f"tie scenes: {compared} compared, {flipped_needed} on the other branch of the start tie")


def test_start_off_the_reference_line_nodes_has_no_tie(planner):
    """The same generator with the planning start 2.7 m ahead of the origin instead of 2.0 (not on a node): no scene
    may need the flipped branch - every planned trajectory is within SURVEY 8(d)'s rule of the port, first points
    included."""
    cfg = S.CFG2
    seeds = list(range(170, 186)) + list(range(3200, 3208))
    batch = S.make_batch(seeds, cfg, **BENCH)
    out = _plan_resident(planner, cfg, _host_inputs(batch))
    compared = 0
    for i, seed in enumerate(seeds):
        port = _port_cycle(cfg, batch, i)
        ok = port.get("qp_status", "optimal") == "optimal" and port["smooth_status"] == "optimal"
        assert ok == ((int(out["status"][i]) & ~1) == 0), f"seed {seed}: outcome"
        if not ok:
            continue
        gap = float(np.abs(np.asarray(port["s_map"]) - port["begin_s"]).min())
        assert gap > 1e-3, f"seed {seed}: start on a node after all ({gap:.2e})"
        want = np.asarray(port["trajectory"], dtype=np.float64)
        assert out["traj_len"][i] == len(want)
        assert_rel(out["traj"][i, :len(want)], want, RTOL, f"seed {seed} (start off the nodes)")
        compared += 1
    assert compared >= 16


def test_on_node_batch_still_matches_its_reference_outputs(planner):
    """The round 1-4 batch (start ON node 6): its 32 reference-generated scenes inside a 1024-scene batch."""
    cfg = S.CFG2
    out = _plan_resident(planner, cfg, _host_inputs(S.make_batch(range(1024), cfg, start_ahead=2.0)))
    _check_golden_subset(out, "cycle_cfg2_40x9_8obs.npz")


def test_survey_geometry_batch(planner):
    """SURVEY 8(d)'s own geometry on the GPU: 2048 scenes on arcs of radius 150-1000 m, odd seeds with the survey's
    slalom layout, every other pair started off the nodes (scenes.survey_geometry_kwargs).  On tight arcs the reference's
    `match_point_index_list[0]` quirk (planning_utils.py:413) and its tangent-line projection (:414-424) bend the S-L
    picture by metres.  Checked: the imported reference's outputs for seeds 0..31 (the fixture), the faithful port on a
    sample of later seeds (outcome for outcome, trajectories at 1e-6), permutation / repetition bit-identical."""
    cfg = S.CFG2
    B = 2048
    batch = S.make_batch(range(B), cfg, per_seed=S.survey_geometry_kwargs)
    host = _host_inputs(batch)
    out = _plan_resident(planner, cfg, host)
    _check_golden_subset(out, "cycle_cfg2_40x9_8obs_tight.npz", min_checked=15)
    ok = (out["status"] & ~1) == 0
    assert 0.3 < ok.mean() < 0.8 and (out["status"] & 4).any() and (out["status"] & 8).any() and (out["status"] & 1).any()
    compared = 0
    for i in (1000, 1001, 1002, 1003, 1500, 1501, 1502, 1503, 2040, 2041, 2042, 2047):
        try:
            port = _port_cycle(cfg, batch, i)
        except IndexError:
            assert out["status"][i] & 4, f"seed {i}: the reference raises IndexError"
            continue
        p_ok = port.get("qp_status", "optimal") == "optimal" and port["smooth_status"] == "optimal"
        assert p_ok == bool(ok[i]), f"seed {i}: outcome"
        assert bool(out["status"][i] & 1) == (not port["dp_feasible"])
        if not p_ok:
            continue
        want = np.asarray(port["trajectory"], dtype=np.float64)
        assert out["traj_len"][i] == len(want)
        assert_rel(out["traj"][i, :len(want)], want, RTOL, f"seed {i} (tight arc)")
        compared += 1
    assert compared >= 4
    _assert_same(out, _plan_resident(planner, cfg, host), "repeated call")
    perm = np.random.default_rng(17).permutation(B)
    _assert_same({k: v[perm] for k, v in out.items()}, _plan_resident(planner, cfg, host, perm), "permuted batch")


def test_configs3_32768_scenes_and_rank_shards(planner):
    from emplanner_carla_amd import dist as emp_dist
    from emplanner_carla_amd.api import dp_params_from_cfg
    cfg = S.CFG2
    B = 32768
    batch = S.make_batch(range(B), cfg, **BENCH)
    host = _host_inputs(batch)
    out = _plan_resident(planner, cfg, host)
    _check_golden_subset(out)
    # a random sample against the exact DP oracle, through the batch's own positions
    rows, mc, st = planner.dp_plan(dp_params_from_cfg(cfg), batch.sl_obs_s, batch.sl_obs_l, batch.n_obs, batch.sl_start)
    pick = np.sort(np.random.default_rng(11).choice(B, 512, replace=False))
    xrows, xfeas, _ = ex.dp_plan(batch.sl_obs_s[pick], batch.sl_obs_l[pick], batch.n_obs[pick], batch.sl_start[pick],
                                 cfg.row, cfg.col, cfg.sample_s, cfg.sample_l, cfg.sampling_res)
    assert np.array_equal(rows[pick], xrows)
    assert np.array_equal(st[pick] == 1, ~xfeas)
    # what rank r of a W-GPU run computes is exactly its slice of the one-GPU result
    for world, rank in ((2, 1), (4, 2), (8, 7), (8, 0)):
        a, n = emp_dist.shard_range(B, rank, world)
        sl = slice(a, a + n)
        _assert_same({k: v[sl] for k, v in out.items()}, _plan_resident(planner, cfg, host, sl), f"shard {rank}/{world}")
    perm = np.random.default_rng(3).permutation(B)
    outp = _plan_resident(planner, cfg, host, perm)
    _assert_same({k: v[perm] for k, v in out.items()}, outp, "permuted batch")
    _check_properties(planner, cfg, host, out)


def _moved(host, phi, shift):
    B = len(phi)
    c, s = np.cos(phi), np.sin(phi)

    def rot(xy):                                   # (B, ..., 2) vectors
        x, y = xy[..., 0], xy[..., 1]
        cc = c.reshape((B,) + (1,) * (x.ndim - 1))
        ss = s.reshape((B,) + (1,) * (x.ndim - 1))
        return np.stack([cc * x - ss * y, ss * x + cc * y], axis=-1)

    moved = dict(host)
    ref = host["ref_line"].copy()
    ref[..., :2] = rot(host["ref_line"][..., :2]) + shift[:, None, :]
    ref[..., 2] = host["ref_line"][..., 2] + phi[:, None]
    moved["ref_line"] = ref
    for k in ("origin_xy", "start_xy"):
        moved[k] = rot(host[k]) + shift
    for k in ("start_v", "start_a"):
        moved[k] = rot(host[k])
    moved["obs_xy"] = rot(host["obs_xy"]) + shift[:, None, :]
    return moved


def _back(xy, phi, shift):
    c, s = np.cos(phi)[:, None], np.sin(phi)[:, None]
    d = xy - shift[:, None, :]
    return np.stack([c * d[..., 0] + s * d[..., 1], -s * d[..., 0] + c * d[..., 1]], axis=-1)


def test_rigid_motion_of_the_scene_moves_the_plan_with_it(planner):
    """Move every scene of the 4096 batch (reference line, ego state, obstacles) and plan again: the Frenet-frame
    results (DP rows, path s / l) must not change, the un-smoothed Cartesian path must move with the scene under any
    rigid motion, and the smoothed trajectory under a translation (its +-0.2 m boxes are aligned with the global axes,
    planning_utils.py:308-311, so a rotation legitimately changes it).  Holds for the reference up to rounding; here it
    checks projection, DP, QP and the Cartesian tail against each other on poses the golden vectors do not contain."""
    cfg = S.CFG2
    B = 4096
    m = 23
    batch = S.make_batch(range(B), cfg, **BENCH)
    host = _host_inputs(batch)
    out = _plan_resident(planner, cfg, host)
    rng = np.random.default_rng(5)

    def targets(h, o):
        sm, _, _, bsl, _ = planner.frenet_project(**h)
        return planner.frenet_path_to_xy(h["ref_line"], sm, h["n_ref"], bsl, o["path_s"], o["path_l"], o["path_len"])[0]

    for kind in ("rigid", "translation"):
        phi = rng.uniform(-np.pi, np.pi, B) if kind == "rigid" else np.zeros(B)
        shift = rng.uniform(-500.0, 500.0, (B, 2))
        moved = _moved(host, phi, shift)
        outm = _plan_resident(planner, cfg, moved)
        ok = ((out["status"] & ~1) == 0) & ((outm["status"] & ~1) == 0)
        assert (out["status"] == outm["status"]).mean() > 0.995
        same_rows = (out["dp_rows"] == outm["dp_rows"]).all(axis=1)
        assert same_rows[ok].mean() > 0.995, "a DP decision may flip on a near-tie, but rarely"
        sel = ok & same_rows
        assert sel.sum() > 3000
        assert (out["traj_len"][sel] == m).all() and (outm["traj_len"][sel] == m).all()
        assert np.abs(out["path_s"][sel][:, :m - 1] - outm["path_s"][sel][:, :m - 1]).max() < 1e-6, kind
        assert np.abs(out["path_l"][sel][:, :m - 1] - outm["path_l"][sel][:, :m - 1]).max() < 1e-6, kind
        t0, t1 = targets(host, out), targets(moved, outm)
        # A station within rounding of a reference-line knot may project from the neighbouring node in the other pose
        # (cal_proj_point walks `while s_map[idx + 1] < s`, path_planning.py:62-63), which moves it by O(kappa ds^2),
        # about a millimetre.  The synthetic scenes put the planning start on the normal through a node, so every
        # fourth station sits on a knot: those points get 5 mm, all others 1e-6.
        sm = planner.frenet_project(**host)[0]
        bsl = planner.frenet_project(**host)[3]
        st_s = np.concatenate([bsl[:, :1], out["path_s"][:, :m - 1]], axis=1)            # s of trajectory point k
        on_knot = (np.abs(st_s[:, :, None] - sm[:, None, :]).min(axis=2) < 1e-6)[sel]
        d = np.abs(_back(t1[:, :m], phi, shift)[sel] - t0[sel][:, :m]).max(axis=2)
        assert on_knot.mean() < 0.5 and d[~on_knot].max() < 1e-6 and d.max() < 5e-3, (kind, d[~on_knot].max(), d.max())
        moved_traj = _back(outm["traj"][:, :m, :2], phi, shift)[sel] - out["traj"][sel][:, :m, :2]
        if kind == "translation":                      # smoothing couples the points: millimetres everywhere
            assert np.abs(moved_traj).max() < 5e-3
            assert np.abs(outm["traj"][sel][:, 2:m, 2] - out["traj"][sel][:, 2:m, 2]).max() < 5e-3
            assert np.abs(outm["traj"][sel][:, 2:m, 3] - out["traj"][sel][:, 2:m, 3]).max() < 5e-3
        else:                                          # the boxes bound how far the smoothed points can drift apart
            assert np.abs(moved_traj).max() < 0.4 * np.sqrt(2) + 1e-6


PIPE_MODES = ["staged", 2, 3, 4]         # emp_set_pipeline: two stages on two streams, or n whole cycles on n lanes


@pytest.mark.parametrize("pipe", PIPE_MODES)
def test_pipelined_cycles_equal_plain_cycles(planner, pipe):
    """emp_set_pipeline, staged (the back stage of one batch overlaps the front stage of the next) and in lane mode (n
    whole cycles on n streams).  Six DIFFERENT batches in flight one after the other (calls that share a lane share its
    pool of temporaries) give bit-identical results to the plain calls, whether the caller works on the planner's own
    stream and reads after a synchronize, or on torch's default stream and reads right away."""
    import torch
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    dev = torch.device("cuda:0")
    batches = []
    for k in range(6):
        b = S.make_batch(range(1000 * k, 1000 * k + 1536 + 64 * k), cfg)      # different sizes as well
        batches.append({kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in _host_inputs(b).items()})
    torch.cuda.synchronize()
    plain = []
    for ins in batches:
        r = planner.plan_cycle(p, q, sp, **ins)
        planner.synchronize()
        plain.append({k: getattr(r, k).cpu().numpy() for k in OUTPUTS})
    planner.set_pipeline(pipe)
    assert planner.in_flight == (2 if pipe == "staged" else pipe)
    # emp_pipeline_depth (ABI 8): how many further calls a call's outputs must outlive - the staged form rotates four pools
    # of temporaries (two batches overlap) and waits on the host for the pool's previous user; six calls go round it
    assert planner._lib.emp_pipeline_depth(planner._h) == (4 if pipe == "staged" else pipe) == planner._retain
    try:
        with torch.cuda.stream(planner.torch_stream()):                       # the bench's way: no cross-stream waits
            res = [planner.plan_cycle(p, q, sp, **ins) for ins in batches]
        planner.synchronize()
        for k, r in enumerate(res):
            _assert_same(plain[k], {kk: getattr(r, kk).cpu().numpy() for kk in OUTPUTS}, f"pipelined batch {k}")
        for k, ins in enumerate(batches):                                       # default stream: results usable at once
            r = planner.plan_cycle(p, q, sp, **ins)
            # what the back stage writes LAST is read FIRST: a missing wait for the result stream shows here every time
            # (it once showed only as one slowest scene in twenty runs, with the fields read in declaration order)
            first = {kk: getattr(r, kk).cpu().numpy() for kk in ("status", "traj_len", "traj", "path_len")}
            rest = {kk: getattr(r, kk).cpu().numpy() for kk in OUTPUTS if kk not in first}
            _assert_same(plain[k], {**first, **rest}, f"pipelined batch {k}, default stream")
        # another entry point right behind a pipelined cycle sees its finished outputs
        r = planner.plan_cycle(p, q, sp, **batches[2])
        sm, _, _, bsl, _ = planner.frenet_project(**batches[2])
        tgt = planner.frenet_path_to_xy(batches[2]["ref_line"], sm, batches[2]["n_ref"], bsl, r.path_s, r.path_l, r.path_len)[0]
        planner.set_pipeline(False)
        r0 = planner.plan_cycle(p, q, sp, **batches[2])
        tgt0 = planner.frenet_path_to_xy(batches[2]["ref_line"], sm, batches[2]["n_ref"], bsl, r0.path_s, r0.path_l, r0.path_len)[0]
        planner.synchronize()
        ok = (plain[2]["status"] & ~1) == 0
        assert np.array_equal(tgt.cpu().numpy()[ok], tgt0.cpu().numpy()[ok], equal_nan=True)
    finally:
        planner.set_pipeline(False)


@pytest.mark.parametrize("pipe", [2, 3])
def test_lanes_of_very_different_durations_do_not_reuse_live_outputs(planner, pipe):
    """Lane mode with a caller that works on the planner's stream, packs each call's records on its result stream and
    drops the results at once (what bench.py's multi-GPU step does), on batches of VERY different sizes: a 4096-scene
    call holds its lane for many small calls of the other lanes, whose outputs torch's allocator carves out of whatever
    was released last.  Every call's records must equal the plain, unpipelined call's.  (Round-2 review: the layer
    released call k's outputs when call k + n was issued, and call k + n + 1 - on another lane, ordered against nothing
    but the main stream - could be handed that memory while call k still ran.)"""
    import torch
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    M = None
    dev = torch.device("cuda:0")
    sizes = [4096, 48, 64, 80, 48, 64, 4096, 96, 48, 64, 80, 48, 2048, 64, 48]
    batches = []
    for k, n in enumerate(sizes):
        b = S.make_batch(range(700 * k, 700 * k + n), cfg)
        batches.append({kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in _host_inputs(b).items()})
    from emplanner_carla_amd.api import max_path_points
    M = max_path_points(p)
    torch.cuda.synchronize()
    planner.set_pipeline(0)
    want = []
    for ins in batches:
        r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)
        want.append(planner.pack_records(r, p.col, M).cpu().numpy())
    planner.synchronize()
    planner.set_pipeline(pipe)
    try:
        ts = planner.torch_stream()
        for rep in range(3):
            got = []
            with torch.cuda.stream(ts):
                for ins in batches:
                    r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)
                    got.append(planner.pack_records(r, p.col, M))
                    del r                                    # the layer alone keeps the outputs of the calls in flight
            planner.synchronize()
            torch.cuda.synchronize()
            for k, (g, w) in enumerate(zip(got, want)):
                g = g.cpu().numpy()
                ok = (w[:, 0].astype(np.int64) & ~1) == 0          # refused scenes: only the status is specified
                assert np.array_equal(g[:, 0], w[:, 0]), f"round {rep}, call {k}: status"
                assert np.array_equal(g[ok], w[ok]), f"round {rep}, call {k} ({sizes[k]} scenes): records differ from the plain call"
    finally:
        planner.set_pipeline(0)


@pytest.mark.parametrize("pipe", PIPE_MODES)
def test_pipelined_records_packed_on_the_result_stream(planner, pipe):
    """What a rank of the multi-GPU bench does per step when several batches are in flight: plan on the planner's first
    stream, pack the result records on the stream on which the results become complete (the gather follows there; in
    lane mode it is a different stream from step to step).  The records of every step equal those of the plain call."""
    import torch
    from emplanner_carla_amd import dist as emp_dist
    from emplanner_carla_amd.api import max_path_points
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    M = max_path_points(p)
    dev = torch.device("cuda:0")
    batches = []
    for k in range(5):
        b = S.make_batch(range(500 * k, 500 * k + 2048), cfg)
        batches.append({kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in _host_inputs(b).items()})
    torch.cuda.synchronize()
    want = []
    for ins in batches:
        r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)
        planner.synchronize()
        want.append(emp_dist.pack_records(r, p.col, M).cpu().numpy())
    planner.set_pipeline(pipe)
    try:
        recs = []
        for ins in batches:
            with torch.cuda.stream(planner.torch_stream()):
                r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)
            with torch.cuda.stream(planner.torch_result_stream()):
                recs.append(emp_dist.gather_records(emp_dist.pack_records(r, p.col, M), len(ins["n_obs"])))
            del r                                       # the step's outputs are released while work is still queued
        planner.synchronize()
        torch.cuda.synchronize()
        for k in range(5):
            got = recs[k].cpu().numpy()
            ok = (want[k][:, 0].astype(np.int64) & ~1) == 0
            assert np.array_equal(got[:, :3], want[k][:, :3])                  # status, lengths
            assert np.array_equal(got[ok], want[k][ok], equal_nan=True), f"step {k}"
    finally:
        planner.set_pipeline(False)


@pytest.mark.parametrize("pipe", ["staged", 3])
def test_native_record_packing_equals_the_torch_form(planner, pipe):
    """emp_pack_records (one launch) against the torch concatenation of emplanner_carla_amd.dist.pack_records, bit for
    bit, full and trimmed records: plain mode from torch's default stream, pipelined mode from the result stream (the
    bench's per-step pattern, the step's outputs dropped at once) and from the default stream."""
    import torch
    from emplanner_carla_amd import dist as emp_dist
    from emplanner_carla_amd.api import max_path_points
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    M = max_path_points(p)
    dev = torch.device("cuda:0")
    batches = []
    for k in range(4):
        b = S.make_batch(range(700 * k, 700 * k + 1500 + 32 * k), cfg)
        batches.append({kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in _host_inputs(b).items()})
    torch.cuda.synchronize()
    same = lambda x, y: np.array_equal(x.view(np.uint64), y.view(np.uint64))
    want = []
    for ins in batches:
        r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)
        for cap in (None, emp_dist.path_capacity(M)):
            ref = emp_dist.pack_records(r, p.col, M, path_cap=cap).cpu().numpy()
            got = emp_dist.pack_records(r, p.col, M, path_cap=cap, planner=planner).cpu().numpy()
            assert got.shape == (len(ins["n_obs"]), emp_dist.record_width(p.col, M, cap)) and same(got, ref)
        want.append(ref)
    # host arrays in, host records out (staged by the library)
    from emplanner_carla_amd.api import CycleResult
    host = CycleResult(**{k: getattr(r, k).cpu().numpy() for k in OUTPUTS})
    got = emp_dist.pack_records(host, p.col, M, path_cap=emp_dist.path_capacity(M), planner=planner)
    assert isinstance(got, np.ndarray) and same(got, want[-1])
    planner.set_pipeline(pipe)
    try:
        recs = []
        for ins in batches:
            with torch.cuda.stream(planner.torch_stream()):
                r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)
            with torch.cuda.stream(planner.torch_result_stream()):
                recs.append(emp_dist.pack_records(r, p.col, M, path_cap=emp_dist.path_capacity(M), planner=planner))
            del r
        planner.synchronize()
        torch.cuda.synchronize()
        for k, ins in enumerate(batches):
            ok = (want[k][:, 0].astype(np.int64) & ~1) == 0
            got = recs[k].cpu().numpy()
            assert np.array_equal(got[:, :3], want[k][:, :3]) and same(got[ok], want[k][ok]), f"step {k}"
            r = planner.plan_cycle(p, q, sp, max_pts=M, **ins)                  # default stream, used at once
            got = planner.pack_records(r, p.col, M, emp_dist.path_capacity(M)).cpu().numpy()
            assert np.array_equal(got[:, :3], want[k][:, :3]) and same(got[ok], want[k][ok]), f"default stream, step {k}"
    finally:
        planner.set_pipeline(False)


@pytest.mark.parametrize("pipe", ["staged", 2])
def test_unfenced_calls_overlap_the_cycles_in_flight(planner, pipe):
    """emp_set_fence(0): the S-T speed planner of the same scenes, queued on the main stream while cycles are in flight,
    neither waits for them nor disturbs them - what bench.py --config cfg5 does every step.  Both halves equal their
    plain results bit for bit."""
    import torch
    from emplanner_carla_amd.api import speed_dp_params
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    sdp = speed_dp_params()
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    batches, dyns = [], []
    for k in range(4):
        b = S.make_batch(range(300 * k, 300 * k + 1024), cfg)
        batches.append({kk: t(v) for kk, v in _host_inputs(b).items()})
        d = S.make_dynamic_batch(range(300 * k, 300 * k + 1024), 16)
        dyns.append(([t(a) for a in d[:4]], t(d[4])))
    torch.cuda.synchronize()
    plain_cycle, plain_st = [], []
    for ins, (obs, v0) in zip(batches, dyns):
        r = planner.plan_cycle(p, q, sp, **ins)
        sets = planner.st_graph(*obs)
        st = planner.speed_dp(sdp, *sets, v0, tables=False)
        planner.synchronize()
        plain_cycle.append({k: getattr(r, k).cpu().numpy() for k in OUTPUTS})
        plain_st.append((st.speed_s.cpu().numpy(), st.speed_t.cpu().numpy(), st.end_node.cpu().numpy()))
    planner.set_pipeline(pipe)
    try:
        res, sts = [], []
        with torch.cuda.stream(planner.torch_stream()):
            for ins, (obs, v0) in zip(batches, dyns):
                res.append(planner.plan_cycle(p, q, sp, **ins))
                planner.set_fence(False)
                sets = planner.st_graph(*obs)
                sts.append(planner.speed_dp(sdp, *sets, v0, tables=False))
                planner.set_fence(True)
        planner.synchronize()
        torch.cuda.synchronize()
        for k in range(4):
            _assert_same(plain_cycle[k], {kk: getattr(res[k], kk).cpu().numpy() for kk in OUTPUTS}, f"cycle {k}")
            got = (sts[k].speed_s.cpu().numpy(), sts[k].speed_t.cpu().numpy(), sts[k].end_node.cpu().numpy())
            for a, b in zip(got, plain_st[k]):
                assert np.array_equal(a, b, equal_nan=True), f"speed DP {k}"
    finally:
        planner.set_fence(True)
        planner.set_pipeline(False)


def _benchmark_batch_under(planner, option, value, time_kernel=None):
    """The 4096 benchmark scenes through the whole cycle with one emp_set_option value in force (restored afterwards).
    ``time_kernel``: also return the mean duration (ms) of that kernel over the call."""
    cfg = S.CFG2
    host = _host_inputs(S.make_batch(range(4096), cfg, **BENCH))
    old = planner.get_option(option)
    planner.set_option(option, value)
    try:
        if time_kernel is None:
            return _plan_resident(planner, cfg, host)
        planner.set_timing(True, only=time_kernel)
        out = _plan_resident(planner, cfg, host)
        ms = planner.kernel_ms(time_kernel)
        planner.set_timing(False)
        return out, ms
    finally:
        planner.set_option(option, old)


def test_both_path_qp_kernels_agree_on_the_benchmark_batch(planner):
    """The cycle's path QP runs eight scenes per wavefront (emp_qp_rows.h: R stations per lane); the two-scenes-per-wavefront
    kernel of rounds 1-2 (emp_qp_wave.h) is EMP_OPT_PATH_QP_FORM = 1.  Same algorithm and stopping rule, sums associated
    differently: on all 4096 benchmark scenes the two must classify every scene alike and agree on path and trajectory far
    inside the 1e-6 bar (measured: 2e-9).  Scene 446 is the one that found the stopping rule's weak spot: its dual residual
    sits at the threshold when the complementarity has converged, one more iteration destroys the iterate (HISTORY 3.3)."""
    a = _benchmark_batch_under(planner, "path_qp_form", 0)
    b = _benchmark_batch_under(planner, "path_qp_form", 1)
    assert np.array_equal(a["status"], b["status"])
    ok = (a["status"] & ~1) == 0
    assert ok.sum() > 3000
    assert np.array_equal(a["traj_len"], b["traj_len"])
    assert np.abs(a["path_l"][ok] - b["path_l"][ok]).max() < 1e-7
    assert np.abs(a["traj"][ok] - b["traj"][ok]).max() < 1e-7
    assert not np.array_equal(a["path_l"][ok], b["path_l"][ok]), "the option must reach a different kernel"


def test_cartesian_tail_kernels_agree_on_the_benchmark_batch(planner):
    """The cycle's Cartesian tail runs four scenes per wavefront (cycle_cartesian_rows_kernel); the one-scene-per-wavefront
    kernel of rounds 1-2 is EMP_OPT_CARTESIAN_FORM = 1.  Same operations per coordinate in the same order: the
    trajectories of the 4096 benchmark scenes must be BIT-identical.  Third run: EMP_OPT_SMOOTH_FORCE_FALLBACK = 1 sends every
    scene of the new kernel through its fall-back (the half-wave interior-point / active-set solvers, the whole wavefront
    on one scene at a time) - a path no test or benchmark scene takes by itself; it must classify alike and agree to 1e-9
    (measured on the GPU box: the fall-back converges to the same active set and its final solve performs the same
    operations, so the trajectories come out bit-identical - which is why the proof that the hook reaches it is the kernel's
    duration: four scenes one after the other on the whole wavefront instead of side by side)."""
    a, ms_a = _benchmark_batch_under(planner, "cartesian_form", 0, time_kernel="to_cartesian")
    b = _benchmark_batch_under(planner, "cartesian_form", 1)
    c, ms_c = _benchmark_batch_under(planner, "smooth_force_fallback", 1, time_kernel="to_cartesian")
    for other in (b, c):
        assert np.array_equal(a["status"], other["status"]) and np.array_equal(a["traj_len"], other["traj_len"])
    ok = (a["status"] & ~1) == 0
    assert ok.sum() > 3000
    assert np.array_equal(a["traj"], b["traj"]), "the two kernels perform the same operations: bit-identical"
    assert np.abs(a["traj"][ok] - c["traj"][ok]).max() < 1e-9
    assert ms_a > 0 and ms_c > 1.3 * ms_a, f"the hook must reach the fall-back: {ms_c:.3f} ms against {ms_a:.3f} ms"


def test_small_shards_equal_their_slice_of_the_full_batch(planner):
    """What rank r of an N-GPU run computes must be its slice of the one-GPU result whatever the shard size - also when a
    shard is far smaller than the batch (4096 scenes on 8 ranks: 512 each; 16 ranks' worth: 256; a 7-scene tail).  Round 3
    chose the path-QP kernel from the batch size (two scenes per wavefront below 1024), which broke exactly this; the form is
    now an option of the caller (include/emplanner.h) and every kernel of the cycle treats a scene independently of its
    neighbours in the batch."""
    cfg = S.CFG2
    B = 4096
    host = _host_inputs(S.make_batch(range(B), cfg, **BENCH))
    out = _plan_resident(planner, cfg, host)
    for a, n in ((0, 512), (3584, 512), (1024, 256), (777, 100), (4089, 7), (5, 1)):
        sl = slice(a, a + n)
        _assert_same({k: v[sl] for k, v in out.items()}, _plan_resident(planner, cfg, host, sl), f"shard {a}+{n}")


@pytest.mark.parametrize("options", [{}, {"sweep_exclusive": 1}, {"sweep_exclusive": 2}, {"edge_after_enrich": 0},
                                     {"sweep_exclusive": 2, "edge_after_enrich": 0}],
                         ids=["default", "exclusive_sweep", "sweep_behind_the_path_qp", "edge_not_held", "exclusive2_edge_not_held"])
def test_staged_handoff_holds_when_the_sweep_carries_timing_events(planner, options):
    """bench.py's timed region brackets the sweep with HIP events, and in staged mode the back stage is released by the
    event attached to the sweep's own dispatch - the TIMING event then (emp_api.hip: front_attached).  Consecutive calls on
    DIFFERENT batches with the events on must equal the plain calls bit for bit: a back stage that started early would
    densify another batch's predecessor table (the bench itself, planning the same batch every step, could not tell).
    Every ordering option the staged pipeline still has (include/emplanner.h: EMP_OPT_SWEEP_EXCLUSIVE 0 / 1 / 2 - the default is 0 -
    and EMP_OPT_EDGE_AFTER_ENRICH - default 1) moves waits between the two queues: each one is held to the same bar."""
    import torch
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    dev = torch.device("cuda:0")
    batches = []
    for k in range(6):
        b = S.make_batch(range(5000 + 3000 * k, 5000 + 3000 * k + 2048), cfg)
        batches.append({kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in _host_inputs(b).items()})
    torch.cuda.synchronize()
    plain = []
    for ins in batches:
        r = planner.plan_cycle(p, q, sp, **ins)
        planner.synchronize()
        plain.append({k: getattr(r, k).cpu().numpy() for k in OUTPUTS})
    old = {k: planner.get_option(k) for k in options}
    for k, v in options.items():
        planner.set_option(k, v)
    planner.set_pipeline("staged")
    planner.set_timing(True, only="dp_sweep")
    try:
        for rep in range(3):
            with torch.cuda.stream(planner.torch_stream()):
                res = [planner.plan_cycle(p, q, sp, **ins) for ins in batches]
            planner.synchronize()
            for k, r in enumerate(res):
                _assert_same(plain[k], {kk: getattr(r, kk).cpu().numpy() for kk in OUTPUTS}, f"timed staged call {k}, round {rep}")
        assert planner.kernel_launches("dp_sweep") == 18 and planner.kernel_ms("dp_sweep") > 0
    finally:
        planner.set_timing(False)
        planner.set_pipeline(False)
        for k, v in old.items():
            planner.set_option(k, v)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_lane_edge_order_changes_no_result(planner, order):
    """EMP_OPT_LANE_EDGE_ORDER (lane mode: the edge-cost kernel of a call waits for the previous call's, which ran on another
    lane): consecutive calls on DIFFERENT batches on three lanes, with the sweep's timing events on as in bench.py, equal the
    plain calls bit for bit under every value (0 is the default; 2 orders calls of 4096 scenes and more - both sizes are run)."""
    import torch
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    dev = torch.device("cuda:0")
    old = planner.get_option("lane_edge_order")
    try:
        for n in (1024, 8192):
            batches = []
            for k in range(5):
                b = S.make_batch(range(7000 + 5000 * k, 7000 + 5000 * k + n), cfg, **BENCH)
                batches.append({kk: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for kk, v in _host_inputs(b).items()})
            torch.cuda.synchronize()
            plain = []
            for ins in batches:
                r = planner.plan_cycle(p, q, sp, **ins)
                planner.synchronize()
                plain.append({k: getattr(r, k).cpu().numpy() for k in OUTPUTS})
            planner.set_option("lane_edge_order", order)
            planner.set_pipeline(3)
            planner.set_timing(True, only="dp_sweep")
            for rep in range(2):
                with torch.cuda.stream(planner.torch_stream()):
                    res = [planner.plan_cycle(p, q, sp, **ins) for ins in batches]
                planner.synchronize()
                for k, r in enumerate(res):
                    _assert_same(plain[k], {kk: getattr(r, kk).cpu().numpy() for kk in OUTPUTS}, f"{n} scenes, lane call {k}, round {rep}")
            planner.set_timing(False)
            planner.set_pipeline(False)
    finally:
        planner.set_timing(False)
        planner.set_pipeline(False)
        planner.set_option("lane_edge_order", old)


def test_measurement_entry_points_of_the_sweep(planner):
    """emp_kernel_samples (the per-launch durations behind emp_kernel_ms) and the in-kernel clock probe
    (EMP_OPT_SWEEP_CLOCK_PROBE: emp_sweep_clock_mhz, emp_sweep_probe_spans) on staged steps of 2048 scenes: one sample per
    launch whose mean is the reported mean, a shader clock in the chip's range, wavefronts that start within microseconds of
    each other and finish inside the launch's event-measured duration - and the probe changes no result."""
    import torch
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    host = _host_inputs(S.make_batch(range(2048), cfg))
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in host.items()}
    plain = _plan_resident(planner, cfg, host)
    planner.set_option("sweep_clock_probe", 1)
    planner.set_pipeline("staged")
    planner.set_timing(True, only="dp_sweep")
    try:
        with torch.cuda.stream(planner.torch_stream()):
            res = [planner.plan_cycle(p, q, sp, **dev) for _ in range(8)]
        planner.synchronize()
        smp = planner.kernel_samples("dp_sweep")
        assert len(smp) == planner.kernel_launches("dp_sweep") == 8 and (smp > 0).all()
        assert abs(float(smp.mean()) - planner.kernel_ms("dp_sweep")) < 1e-6
        mhz, mean_us, max_us = planner.sweep_clock()
        assert 1000.0 < mhz < 3000.0 and 0.0 < mean_us <= max_us
        spread_us, span_us = planner.sweep_probe_spans()
        # (max_us is the longest-lived wavefront of ANY of the eight launches, span_us the launches' MEAN first-start-to-last-end
        # span: with the sweep free to overlap the previous batch's Cartesian tail - the default since round 5 - launches differ)
        assert 0.0 <= spread_us < span_us and max_us <= span_us * 1.5 + 1e-9
        assert span_us < float(smp.mean()) * 1e3 * 1.5 + 5.0          # the wavefronts live inside the launch
        _assert_same(plain, {k: getattr(res[-1], k).cpu().numpy() for k in OUTPUTS}, "staged call with the clock probe on")
    finally:
        planner.set_timing(False)
        planner.set_pipeline(False)
        planner.set_option("sweep_clock_probe", 0)


def test_auto_pipeline_picks_a_form_the_process_can_sustain():
    """emp_set_pipeline(EMP_PIPELINE_AUTO): three lanes want a hardware queue per stream; the HIP runtime has
    GPU_MAX_HW_QUEUES of them (default 4), fixed when it initialises.  With 4 queues AUTO must choose the staged form - and must
    not be slower than it - with 12 it chooses three lanes; every other stream the process declares (option foreign_streams) or
    the context owns (the copy streams of the page-locked path) counts against the lanes.  Each case in a process of its own:
    the queue count cannot change once HIP is up."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["EMP_ROOT"])
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params
cfg = S.CFG2
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
b = S.make_batch(range(4096), cfg, start_ahead=S.BENCH_START_AHEAD)
B, P = b.ref.shape[:2]
ins = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32),
       origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs).items()}
out = {}
for mode in json.loads(os.environ["EMP_MODES"]):
    pl = Planner(0)                                   # a context per mode: a process that only ever runs this form
    for k, v in json.loads(os.environ["EMP_OPTS"]).items():
        pl.set_option(k, v)
    pl.set_pipeline(mode)
    form = pl.pipeline_form()
    best = 1e9
    with torch.cuda.stream(pl.torch_stream()):
        for rep in range(4):
            for _ in range(40):
                pl.plan_cycle(p, q, sp, **ins)
            pl.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                pl.plan_cycle(p, q, sp, **ins)
            pl.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    out[str(mode)] = {"form": form, "ms": best}
    pl.set_pipeline(0)
    pl.close()
print(json.dumps(out))
'''

    def run(queues, modes, opts=None):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if queues:
            env["GPU_MAX_HW_QUEUES"] = str(queues)
        env.update(EMP_ROOT=root, EMP_MODES=json.dumps(modes), EMP_OPTS=json.dumps(opts or {}))
        r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    four = run(4, ["auto"])
    assert four["auto"]["form"] == [1, 4, 2], four            # staged: 4 queues < 3 lanes + main + one foreign stream
    four_staged = run(4, ["staged"])
    four_lanes = run(4, [3])
    assert four_lanes["3"]["form"][0] == 3
    # AUTO is the staged form there: not slower than asking for it (10 % for the noise between two processes on a shared box)
    assert four["auto"]["ms"] <= 1.10 * four_staged["staged"]["ms"], (four, four_staged)
    print("4 hardware queues: auto (staged) %.4f ms, staged %.4f, three lanes %.4f" %
          (four["auto"]["ms"], four_staged["staged"]["ms"], four_lanes["3"]["ms"]))
    default = run(None, ["auto"])                             # the variable unset: HIP's default of four
    assert default["auto"]["form"] == [1, 4, 2], default
    twelve = run(12, ["auto"])
    assert twelve["auto"]["form"] == [3, 12, 2], twelve
    assert run(12, ["auto"], {"foreign_streams": 9})["auto"]["form"] == [1, 12, 10]
    assert run(5, ["auto"], {"foreign_streams": 1})["auto"]["form"] == [3, 5, 2]
