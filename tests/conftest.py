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


def rel_close(a, b, rtol, scale=1.0):
    """|a-b| <= rtol * max(|b|, scale): relative tolerance with an explicit magnitude floor."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) <= rtol * np.maximum(np.abs(b), scale)


def assert_rel(a, b, rtol, scale=1.0, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ok = rel_close(a, b, rtol, scale)
    if not ok.all():
        err = np.abs(a - b) / np.maximum(np.abs(b), scale)
        raise AssertionError(f"{what}: max scaled error {np.nanmax(err):.3e} > {rtol:.1e} "
                             f"at {np.unravel_index(np.nanargmax(err), err.shape)}")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
