"""Randomised parity sweep over lattice shapes the fixed tests do not touch: odd and even row counts (the generic
sweep path and the compiled ones), short and long lattices, different station spacings and densities, 0..12 obstacles
(more QP stations than one half-wave holds, trajectories of 20..130 points).

DP: rows, densified path and status bit-exact against oracle/exact.py.  Full cycle: per-scene outcome and trajectory
against the faithful port (oracle/ref_port.py) at 1e-6."""
import os

import numpy as np
import pytest

from emplanner_carla_amd import scenes as S
from oracle import exact as ex
from oracle import ref_port as op
from tests.conftest import assert_rel

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("EMP_FUZZ_SCALE", "1"))       # the same tests on SCALE times as many scenes per lattice shape

# sample_s is kept off the integers: the reference sizes its output with int(end_s - start_s)
# (path_planning.py:398), which for an integer sample_s sits on the edge of a truncation and follows the last bit of
# the projected start s - and that bit comes out of ndarray.dot, i.e. out of the BLAS the reference runs on
# (HISTORY.md "Known sensitivity"); the integer cases live in the golden fixtures.
SHAPES = [  # row, col, sample_s, sample_l, res, n_obs
    (3, 12, 7.5, 1.5, 2, 2), (4, 20, 4.5, 1.0, 1, 4), (6, 16, 5.2, 1.2, 2, 3), (7, 30, 3.2, 1.0, 1, 6),
    (9, 24, 2.5, 1.5, 2, 8), (11, 14, 6.3, 0.8, 2, 5), (15, 18, 4.2, 0.6, 2, 7), (5, 10, 9.3, 1.5, 1, 0),
    (12, 8, 10.4, 1.0, 2, 12), (21, 26, 3.5, 0.5, 1, 9),
    # wider than 32 rows (round 3): the generic one-block-per-scene kernels - 33 and 48 (one scene per wavefront), 64 (a full
    # wavefront), 80 (two wavefronts per scene), each with lateral spacings that keep the lattice inside +-7 m
    (33, 10, 4.2, 0.4, 2, 6), (48, 7, 5.3, 0.3, 1, 5), (64, 6, 6.1, 0.22, 2, 8), (80, 5, 4.6, 0.18, 2, 4),
]


@pytest.fixture(scope="module")
def planner():
    from tests.conftest import make_planner
    p = make_planner(0)
    yield p
    p.close()


def _cfg(i):
    row, col, ss, sl, res, n_obs = SHAPES[i]
    n_ref = int(np.ceil((col * ss + 30.0) / 2.0)) + 8
    return S.LatticeConfig(f"fuzz{i}_{col}x{row}", row=row, col=col, sample_s=ss, sample_l=sl, sampling_res=res, n_obs=n_obs,
                           n_ref=n_ref)


@pytest.mark.parametrize("i", range(len(SHAPES)))
def test_dp_bit_exact_on_random_lattices(planner, i):
    from emplanner_carla_amd.api import dp_params_from_cfg, max_path_points
    cfg = _cfg(i)
    b = S.make_batch(range(40 * i * SCALE, 40 * i * SCALE + 24 * SCALE), cfg)
    p = dp_params_from_cfg(cfg)
    rows, mc, st = planner.dp_plan(p, b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)
    xrows, xfeas, xpaths = ex.dp_plan(b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start, cfg.row, cfg.col, cfg.sample_s, cfg.sample_l,
                                      cfg.sampling_res)
    np.testing.assert_array_equal(rows, xrows)
    np.testing.assert_array_equal((st & 1) == 1, ~xfeas)
    M = max_path_points(p)
    ps, pl_, ln, st2 = planner.dp_enrich(p, rows, b.sl_start, M)
    for k in range(len(rows)):
        xs, xl = xpaths[k]
        assert ln[k] == len(xs)
        np.testing.assert_array_equal(ps[k, :ln[k]], xs)
        np.testing.assert_array_equal(pl_[k, :ln[k]], xl)


#: scene options of the cycle fuzz: the generator's default geometry, and SURVEY 8(d)'s (arc radii 150-1000 m, the survey's
#: slalom on odd seeds, starts off the nodes on every other pair - scenes.survey_geometry_kwargs).  Tight arcs only for the
#: tiled lattices (<= 32 rows): there the faithful port is bit-identical to the imported reference (VERDICT r04's probe)
GEOMETRIES = {"gentle": {}, "survey_arcs": dict(per_seed=S.survey_geometry_kwargs)}


@pytest.mark.parametrize("geometry", list(GEOMETRIES))
@pytest.mark.parametrize("i", range(len(SHAPES)))
def test_cycle_vs_port_on_random_lattices(planner, i, geometry):
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params, max_path_points
    cfg = _cfg(i)
    if geometry != "gentle" and cfg.row > 32:
        pytest.skip("tight arcs are run on the tiled lattices")
    seeds = list(range(1000 + 10 * i * SCALE, 1000 + 10 * i * SCALE + 5 * SCALE))
    b = S.make_batch(seeds, cfg, **GEOMETRIES[geometry])
    P = b.ref.shape[1]
    p = dp_params_from_cfg(cfg)
    r = planner.plan_cycle(p, qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params(),
                           max_pts=max_path_points(p), ref_line=b.ref, n_ref=np.full(len(seeds), P, np.int32), origin_xy=b.origin_xy,
                           start_xy=b.start_xy, start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
    compared = 0
    for k in range(len(seeds)):
        nk = int(b.n_obs[k])
        try:
            out = op.plan_cycle(b.ref[k], tuple(b.origin_xy[k]), tuple(b.start_xy[k]), tuple(b.start_v[k]), tuple(b.start_a[k]),
                                [tuple(o) for o in b.obs_xy[k, :nk]],
                                dp_kwargs=dict(row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l,
                                               sampling_res=cfg.sampling_res), obs_length=cfg.obs_length, obs_width=cfg.obs_width,
                                verbose=False)
            port_ok = out.get("qp_status", "optimal") == "optimal" and out["smooth_status"] == "optimal"
            assert bool(r.status[k] & 1) == (not out["dp_feasible"]), f"scene {k}: DP feasibility"
        except IndexError:
            port_ok = False
        dev_ok = (int(r.status[k]) & ~1) == 0
        assert port_ok == dev_ok, f"shape {SHAPES[i]} scene {k}: port ok {port_ok}, device status {int(r.status[k])}"
        if not port_ok:
            continue
        want = np.asarray(out["trajectory"], dtype=np.float64)
        m = int(r.traj_len[k])
        assert m == len(want)
        assert_rel(r.traj[k, :m, :3], want[:, :3], 1e-6, "x, y, theta")
        assert_rel(r.traj[k, :m, 3], want[:, 3], 1e-6, "kappa")
        compared += 1
    assert compared >= 1 or cfg.n_obs >= 9 or geometry != "gentle"


@pytest.mark.parametrize("row", [33, 80, 200, 257, 600])
def test_wide_lattice_edge_tensor_bit_exact_and_layout(planner, row):
    """More than 32 rows (up to 1024 since round 5: 16-bit predecessors; threads take several rows beyond 256): the generic
    kernels' edge tensor equals oracle/exact.py bit for bit, the 'tiled' layout IS the
    canonical one there, the single-kernel DP mode falls back to the two-kernel form, and the sweep entry point consumes
    the tensor the edge entry point produced."""
    from emplanner_carla_amd import _lib as L
    from emplanner_carla_amd.api import dp_params_from_cfg
    cfg = S.LatticeConfig(f"wide_{row}", row=row, col=4, sample_s=5.7, sample_l=13.0 / row, sampling_res=2, n_obs=5, n_ref=40)
    b = S.make_batch(range(300, 300 + 6), cfg)
    p = dp_params_from_cfg(cfg)
    c0, e = planner.dp_edge_costs(p, b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)
    x0, xe = ex.edge_costs(b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start, cfg.row, cfg.col, cfg.sample_s, cfg.sample_l)
    np.testing.assert_array_equal(c0, x0)
    np.testing.assert_array_equal(e, xe)
    c0t, et = planner.dp_edge_costs(p, b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start, layout=L.EMP_EDGE_TILED)
    np.testing.assert_array_equal(et, e)
    assert planner.edge_tensor_elems(p, 6, L.EMP_EDGE_TILED) == 6 * 3 * row * row
    rows, mc, st = planner.dp_plan(p, b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)
    rows_f, mc_f, st_f = planner.dp_plan(p, b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start, mode=L.EMP_DP_FUSED)
    np.testing.assert_array_equal(rows, rows_f)
    np.testing.assert_array_equal(mc, mc_f)
    xrows, xfeas, _ = ex.dp_plan(b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start, cfg.row, cfg.col, cfg.sample_s, cfg.sample_l, cfg.sampling_res)
    np.testing.assert_array_equal(rows, xrows)
    np.testing.assert_array_equal((st & 1) == 1, ~xfeas)
    if hasattr(planner, "dp_sweep"):
        r2, m2, s2 = planner.dp_sweep(p, c0, np.ascontiguousarray(et).reshape(-1))
        sel = b.n_obs > 0                                  # the sweep entry point knows no bypass
        np.testing.assert_array_equal(r2[sel], rows[sel])
    with pytest.raises(Exception):
        planner.dp_plan(dp_params_from_cfg(S.LatticeConfig("too_wide", row=1025, col=3, sample_s=5.0, sample_l=0.0125, sampling_res=2,
                                                           n_obs=0, n_ref=30)), b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)


def test_pair_path_qp_form_on_the_small_batches_of_this_file():
    """The path QP's kernel form is the caller's choice (emp_set_option, EMP_OPT_PATH_QP_FORM) and never follows the batch
    size: every test above runs the default eight-scenes-per-wavefront kernel (R = 3 and R = 4 instantiations, ragged last
    wavefronts), whatever its batch.  The two-scenes-per-wavefront kernel of rounds 1-2 stays available for latency-bound
    callers: the cycle tests of this file and the golden cycle tests run again on it in a child process (the fixtures read
    EMP_TEST_PATH_QP_FORM, tests/conftest.py make_planner)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                          os.path.join(root, "tests", "test_gpu_fuzz.py"), os.path.join(root, "tests", "test_gpu_cycle.py"),
                          "-k", "(cycle_vs_port or full_cycle or path_qp_at_its_size_limits) and not pair_path_qp_form"],
                         cwd=root, env=dict(os.environ, EMP_TEST_PATH_QP_FORM="1"), capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
    assert " passed" in run.stdout
