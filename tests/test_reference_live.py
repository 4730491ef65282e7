"""Live cross-check against the reference tree: every committed fixture under tests/golden/ regenerates, bit for bit, from
the IMPORTED reference through the committed generator scripts.  Runs only where /root/reference exists (the build
container); skipped everywhere else (the GPU box never has the tree, and nothing in the `-m gpu` suite reads it).

What this pins: a fixture cannot drift from its generator (a checker rewrite, a scene-generator change, a numpy upgrade)
without this test saying which array of which file moved.  The generators run as child processes: importing the reference
installs stub ``carla`` / ``cvxopt`` modules and a top-level ``planner`` package that must not leak into this process.
"""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLDEN)
import ref_loader  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_loader.reference_available(),
                                reason="/root/reference is not present (it never travels to the GPU box)")

#: generator script -> the fixtures it writes
GENERATORS = {
    "make_golden.py": ["cycle_cfg1_20x5_0obs.npz", "cycle_default_6x12_3obs.npz", "cycle_cfg2_40x9_8obs.npz",
                       "cycle_cfg2_40x9_8obs_tight.npz", "cycle_cfg2_40x9_8obs_bench.npz", "cycle_cfg2_40x9_8obs_worst.npz",
                       "cycle_default_6x12_3obs_t7.npz", "cycle_default_6x12_3obs_t6.npz", "edges.npz", "functions.npz",
                       "qp_formulation.npz"],
    "make_golden_speed.py": ["speed.npz"],
    "make_golden_speed_backend.py": ["speed_backend.npz"],
    "make_golden_driver.py": ["driver.npz", "driver_s147.npz"],
    "make_golden_mpc.py": ["mpc.npz"],
}


@pytest.fixture(scope="module")
def regenerated(tmp_path_factory):
    """All five generators at once (they are single-threaded Python; about 45 s wall on this container's cores)."""
    out = str(tmp_path_factory.mktemp("golden_live"))
    env = dict(os.environ, EMP_GOLDEN_OUT=out, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    procs = {g: subprocess.Popen([sys.executable, "-B", os.path.join(GOLDEN, g)], env=env, cwd=ROOT,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for g in GENERATORS}
    logs = {}
    for g, p in procs.items():
        logs[g], _ = p.communicate(timeout=840)
        assert p.returncode == 0, f"{g} failed:\n{logs[g][-2000:]}"
    return out


def _differences(want, got):
    diffs = []
    if sorted(want.files) != sorted(got.files):
        diffs.append(f"arrays differ: only committed {sorted(set(want.files) - set(got.files))}, "
                     f"only regenerated {sorted(set(got.files) - set(want.files))}")
    for k in want.files:
        if k not in got.files:
            continue
        a, b = want[k], got[k]
        if a.shape != b.shape or a.dtype != b.dtype:
            diffs.append(f"{k}: {a.dtype}{a.shape} committed, {b.dtype}{b.shape} regenerated")
        elif not np.array_equal(a, b, equal_nan=a.dtype.kind == "f"):
            bad = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == "f" else a != b
            diffs.append(f"{k}: {int(bad.sum())} of {a.size} entries differ (first at {tuple(np.argwhere(bad)[0])})")
    return diffs


@pytest.mark.parametrize("generator", list(GENERATORS))
def test_committed_fixtures_regenerate_bit_for_bit(regenerated, generator):
    problems = []
    for name in GENERATORS[generator]:
        path = os.path.join(regenerated, name)
        assert os.path.exists(path), f"{generator} did not write {name}"
        for d in _differences(np.load(os.path.join(GOLDEN, name)), np.load(path)):
            problems.append(f"{name}: {d}")
    assert not problems, "\n".join(problems)


def test_every_committed_fixture_has_a_generator():
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert committed == sorted(n for names in GENERATORS.values() for n in names)


def test_signature_fixture_is_the_reference_surface():
    import make_signatures
    want = json.load(open(os.path.join(GOLDEN, "signatures.json")))
    got = json.loads(json.dumps(make_signatures.reference_surface()))
    assert got == want
