"""Import the *actual reference* (read-only tree at /root/reference) in this container.

Used only by the ``make_golden*.py`` scripts (fixture generation) and by the live
cross-check (tests/test_reference_live.py), which is skipped when /root/reference is absent (it never exists on the
GPU box).  Nothing here is copied from the reference: we only register two stub modules
so that its ``import carla`` / ``import cvxopt`` lines succeed:

  * ``carla``   - the reference evaluates annotations such as ``loc_1: carla.Location``
                   at def time (reference planner/planning_utils.py:14), so the stub
                   carries dummy attributes of those names;
  * ``cvxopt``  - ``solvers.qp`` RECORDS the dense (P, q, G, h, A, b) the reference built
                   (pinning the QP *formulation*) and returns the KKT-certified solution
                   of ``oracle.qp_dense`` (the QP *arithmetic* is unpinned: cvxopt is not
                   vendored, not version-pinned and not installable here).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

#: every call of the stub ``cvxopt.solvers.qp`` appends a dict here
QP_LOG: list = []


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "planner", "path_planning.py"))


class _Matrix:
    """Minimal stand-in for cvxopt.matrix: dense column-major semantics are not needed
    because the reference only converts numpy arrays and indexes the 1-D solution."""

    def __init__(self, a):
        self.a = np.array(a, dtype=np.float64)
        if self.a.ndim == 1:                       # cvxopt turns a 1-D array into a column vector
            self.a = self.a.reshape(-1, 1)

    def __array__(self, dtype=None, copy=None):
        return self.a if dtype is None else self.a.astype(dtype)

    # item access as far as the reference's speed_QP uses it (speed_planning_test.py:446-482): 2-D slices with an
    # array on the right, single indices into column vectors (column-major linear index, like cvxopt)
    @property
    def size(self):
        return self.a.shape

    def __setitem__(self, key, value):
        if isinstance(key, tuple):
            self.a[key] = np.asarray(value, dtype=np.float64)
        else:
            self.a.reshape(-1, order="F")[key] = value

    def __getitem__(self, key):
        if isinstance(key, tuple):
            return self.a[key]
        return self.a.reshape(-1, order="F")[key]


class _Solution:
    """``res['x']`` in the reference is sliced ([0::3]), indexed ([i]) and len()-ed."""

    def __init__(self, x):
        self._x = [float(v) for v in np.asarray(x).reshape(-1)]

    def __getitem__(self, k):
        return self._x[k]

    def __len__(self):
        return len(self._x)

    def __iter__(self):
        return iter(self._x)


def _install_stubs():
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    from oracle import qp_dense

    carla = types.ModuleType("carla")
    for name in ("Location", "Waypoint", "Vehicle", "Map", "World", "Transform", "Rotation",
                 "Vector3D", "VehicleControl", "Client", "Color", "Actor"):
        setattr(carla, name, type(name, (), {}))

    cvxopt = types.ModuleType("cvxopt")
    solvers = types.ModuleType("cvxopt.solvers")
    solvers.options = {}

    def qp(P, q, G=None, h=None, A=None, b=None, **_kw):
        arr = lambda m: None if m is None else np.asarray(m, dtype=np.float64)
        rec = {"P": arr(P), "q": arr(q), "G": arr(G), "h": arr(h), "A": arr(A), "b": arr(b)}
        if rec["A"] is not None and rec["A"].ndim == 2 and rec["A"].shape[1] != rec["P"].shape[0]:
            # what cvxopt does with the reference's speed_QP call (an untransposed equality matrix, :503)
            rec["status"] = "rejected"
            QP_LOG.append(rec)
            raise TypeError("'A' must be a 'd' matrix with %d columns" % rec["P"].shape[0])
        res = qp_dense.solve_qp(rec["P"], rec["q"], rec["G"], rec["h"], rec["A"], rec["b"])
        rec["x"] = res.x.copy()
        rec["status"] = res.status
        rec["stationarity"] = res.stationarity
        rec["violation"] = res.violation
        rec["polished"] = res.polished
        QP_LOG.append(rec)
        return {"x": _Solution(res.x), "status": res.status}

    solvers.qp = qp
    cvxopt.matrix = _Matrix
    cvxopt.solvers = solvers
    sys.modules["carla"] = carla
    sys.modules["cvxopt"] = cvxopt
    sys.modules["cvxopt.solvers"] = solvers


def load_reference():
    """Return (path_planning, planning_utils) modules of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the tree is read-only; never leave .pyc behind
    _install_stubs()
    # The reference's package is called ``planner``; keep it off sys.path except while importing
    # so that it cannot shadow anything of ours.
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        for name in ("planner", "planner.planning_utils", "planner.path_planning"):
            sys.modules.pop(name, None)
        pu = importlib.import_module("planner.planning_utils")
        pp = importlib.import_module("planner.path_planning")
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return pp, pu


def load_speed_reference():
    """Return the reference's ``planner.speed_planning_test`` module (S-T speed DP, config 5).  It imports
    ``cvxopt`` (stubbed above) and ``scipy.interpolate`` (installed)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        for name in ("planner", "planner.speed_planning_test"):
            sys.modules.pop(name, None)
        sp = importlib.import_module("planner.speed_planning_test")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        sys.modules.pop("planner", None)
    return sp


def load_controller_reference():
    """Return the reference's ``controller.controller`` module (lateral MPC / LQR, longitudinal PID).  It imports
    ``carla`` and ``cvxopt`` (both stubbed above) and ``planner.planning_utils`` (the reference's own)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        for name in ("planner", "planner.planning_utils", "controller", "controller.controller"):
            sys.modules.pop(name, None)
        mod = importlib.import_module("controller.controller")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for name in ("planner", "planner.planning_utils", "controller"):
            sys.modules.pop(name, None)
    return mod


def load_driver_reference():
    """Return the reference's ``test_9`` driver module so that its ``motion_planning(conn)`` (test_9.py:92-220, the
    planning process body) can be run against a fake Pipe.  The driver's other imports (CARLA agents, sensors, the
    global planner, the controller) are irrelevant to that function and are stubbed as empty modules carrying the
    imported names; ``planner.planning_utils`` / ``planner.path_planning`` are the reference's own."""
    if not reference_available():
        raise RuntimeError("reference tree not present at " + REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _install_stubs()
    stubs = {"planner.global_planning": ("global_path_planner",), "sensors": (), "sensors.Sensors_detector_lib": ("Obstacle_detector",),
             "agents": (), "agents.navigation": (), "agents.navigation.behavior_agent": ("BehaviorAgent",),
             "controller": (), "controller.controller": ("Vehicle_control",)}
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        for name in ("planner", "planner.planning_utils", "planner.path_planning", "test_9"):
            sys.modules.pop(name, None)
        importlib.import_module("planner.planning_utils")
        importlib.import_module("planner.path_planning")
        for name, attrs in stubs.items():
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (), {}))
            sys.modules[name] = m
        mod = importlib.import_module("test_9")
    finally:
        sys.path.remove(REFERENCE_ROOT)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for name in ("planner", "planner.planning_utils", "planner.path_planning", "test_9"):
            sys.modules.pop(name, None)
    return mod
