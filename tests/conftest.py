"""pytest configuration: markers, paths and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Safety net (pytest-timeout, when installed): no single test may hold the suite for more than 15 minutes - a rendezvous
    # that never completes or a wedged child process fails its test instead of hanging the run.
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 900


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


ATOL_SURVEY = 1e-9     # "within 1e-6 relative (abs floor 1e-9)"


def tolerance(b, rtol, scale=None):
    """The suite's closeness rule.  For the 1e-6 bar it is SURVEY.md 8(d)'s, to the letter:
    |a - b| <= max(rtol |b|, 1e-9 (rtol / 1e-6)) - no magnitude floor (until round 3 the floor was max(|b|, 1.0), i.e.
    1e-6 m absolute near l = 0: a thousand times looser than what the whole GPU suite measures, see HISTORY.md section 4).
    For the tight comparisons (rtol < 1e-7: S-T cost tables, MPC error model ...) `scale` is an explicit magnitude
    floor, rtol max(|b|, scale), because there an absolute 1e-9 would be the LOOSER rule."""
    b = np.abs(np.asarray(b, dtype=np.float64))
    if rtol >= 1e-7 and scale is None:
        return np.maximum(rtol * b, ATOL_SURVEY * (rtol / 1e-6))
    return rtol * np.maximum(b, 1.0 if scale is None else scale)


def rel_close(a, b, rtol, scale=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) <= tolerance(b, rtol, scale)


def assert_rel(a, b, rtol, what="", scale=None):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    tol = tolerance(b, rtol, scale)
    if os.environ.get("EMP_TOL_LOG") and a.size:      # development: what every comparison of the suite actually measures
        import json
        d = np.abs(a - b)
        fin = np.isfinite(d)
        rec = {"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "what": what, "rtol": rtol, "scale": scale,
               "n": int(a.size), "max_abs": float(d[fin].max(initial=0.0)),
               "max_over_tolerance": float((d[fin] / tol[fin]).max(initial=0.0)),
               "median_abs_b": float(np.median(np.abs(b[fin]))) if fin.any() else None}
        with open(os.environ["EMP_TOL_LOG"], "a") as f:
            f.write(json.dumps(rec) + "\n")
    ok = np.abs(a - b) <= tol
    if not ok.all():
        err = np.abs(a - b) / tol
        raise AssertionError(f"{what}: worst error is {np.nanmax(err):.3g} x the tolerance (rtol {rtol:.1e}) "
                             f"at {np.unravel_index(np.nanargmax(err), err.shape)}")


DP_L_ZERO_FLOOR = 1e-8


def assert_dp_l_vs_reference(a, b, what="dp_l vs reference"):
    """Densified DP path against the REFERENCE's own values: SURVEY 8(d)'s rule with the absolute floor at 1e-8 m
    instead of 1e-9.  Named exception, measured: where the lattice path sits on l = 0 exactly, the reference evaluates
    its quintic in ABSOLUTE s (path_planning.py:405-420, coefficients from a 6x6 inverse with cond ~ 1e15) and returns
    +-1e-9 .. 4e-9 instead of 0 (3.26e-9 at scene 23 of the tight-arc fixture; oracle/ref_port.py reproduces that value
    bit for bit, oracle/exact.py and the GPU return 0.0).  Everywhere else the rule is unchanged."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    tol = np.maximum(1e-6 * np.abs(b), DP_L_ZERO_FLOOR)
    bad = ~(np.abs(a - b) <= tol)
    assert not bad.any(), f"{what}: worst error is {np.nanmax(np.abs(a - b) / tol):.3g} x the tolerance"


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def make_planner(device_id=0):
    """The GPU suites' context.  EMP_TEST_PATH_QP_FORM (an environment variable of the TEST HARNESS - the library reads
    none) selects the path-QP kernel form through emp_set_option, so that a child pytest can rerun whole test files on the
    two-scenes-per-wavefront kernel (tests/test_gpu_fuzz.py)."""
    from emplanner_carla_amd.api import Planner
    pl = Planner(device_id)
    form = os.environ.get("EMP_TEST_PATH_QP_FORM")
    if form:
        pl.set_option("path_qp_form", int(form))
    return pl
