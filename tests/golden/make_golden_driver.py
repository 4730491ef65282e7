"""Golden vectors of the reference's planning process body ``motion_planning(conn)`` (test_9.py:92-220), obtained by
importing the real driver module (tests/golden/ref_loader.load_driver_reference) and running the function against a
fake Pipe: one request in, one reply out, then the fake raises to leave the reference's ``while 1`` loop.

A request is the tuple the driver sends (test_9.py:390-392): static obstacles [(x, y, dis)], dynamic obstacles
[(x, y, dis, speed)], vehicle location, predicted location, velocity, acceleration, the global path [(x, y, theta,
kappa)] and the previous match index list.  The reply (test_9.py:220) is (trajectory [(x, y, theta, kappa)],
match_point_list, path_s, path_l).  The QPs go through the stub cvxopt (oracle/qp_dense.py), as everywhere.

Run:  python tests/golden/make_golden_driver.py      (writes tests/golden/driver.npz)
"""
from __future__ import annotations

import io
import contextlib
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
#: where the .npz files go: this directory, or a scratch one for tests/test_reference_live.py (regenerate and compare)
OUT = os.environ.get("EMP_GOLDEN_OUT", HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402


class _Done(Exception):
    pass


class FakeConn:
    def __init__(self, request):
        self.request, self.reply, self.served = request, None, False

    def recv(self):
        if self.served:
            raise _Done()
        self.served = True
        return self.request

    def send(self, obj):
        self.reply = obj


def global_path(rng, n, pu):
    t = np.arange(n) * 2.0
    xy = np.stack([t + rng.normal(0, 0.03, n), 30.0 * np.sin(t / 70.0 + rng.uniform(0, 3)) + rng.normal(0, 0.03, n)], axis=1)
    th, ka = pu.cal_heading_kappa([tuple(p) for p in xy])
    return [(float(x), float(y), float(a), float(k)) for (x, y), a, k in zip(xy, th, ka)]


def make_request(rng, pu, case):
    """case: 0 no obstacles, 1 static only (near), 2 static only (nearest farther than 30 m: ignored, test_9.py:117),
    3 dynamic only, 4 static + dynamic, 5 dynamic that leaves beyond 80 m (ignored, test_9.py:163)."""
    n = 240
    path = global_path(rng, n, pu)
    at = int(rng.integers(20, 150))
    px, py, pth, _ = path[at]
    off = rng.normal(0, 0.25)
    veh = (px - off * math.sin(pth) - 1.2 * math.cos(pth), py + off * math.cos(pth) - 1.2 * math.sin(pth))
    pred = (px - off * math.sin(pth) + rng.normal(0, 0.05), py + off * math.cos(pth) + rng.normal(0, 0.05))
    speed = rng.uniform(6.0, 11.0)
    v = (speed * math.cos(pth + 0.02), speed * math.sin(pth + 0.02))
    a = (rng.normal(0, 0.3), rng.normal(0, 0.3))

    def at_offset(ahead_nodes, lat):
        x, y, th, _ = path[at + ahead_nodes]
        return x - lat * math.sin(th), y + lat * math.cos(th)

    static, dynamic = [], []
    if case in (1, 4):
        for ahead, lat in ((8, 4.6), (16, -4.8), (24, 5.2)):
            x, y = at_offset(ahead, lat)
            static.append((x, y, math.hypot(x - veh[0], y - veh[1])))
    if case == 2:
        for ahead, lat in ((18, 4.6), (24, -4.8)):
            x, y = at_offset(ahead, lat)
            static.append((x, y, math.hypot(x - veh[0], y - veh[1])))
    static.sort(key=lambda tup: tup[2])
    if case in (3, 4):
        x, y = at_offset(10, 0.3)
        dynamic.append((x, y, 20.0 + rng.uniform(0, 3), speed - rng.uniform(2.5, 4.0)))
        x, y = at_offset(30, -0.2)
        dynamic.append((x, y, 60.0, speed - 1.0))              # only the FIRST dynamic obstacle is used (:141-142)
    if case == 5:
        x, y = at_offset(10, 0.3)
        dynamic.append((x, y, 30.0, speed - 0.4))               # slow closing speed: leaves far beyond 80 m
    return (static, dynamic, veh, pred, v, a, path, [max(0, at - int(rng.integers(0, 3)))])


def run(t9, sample_s, fname):
    """sample_s None: the driver exactly as it is (DP defaults, sample_s = 15).  Otherwise DP_algorithm's keyword default
    is overridden from outside - the reference's code, a non-default parameter: with an integer sample_s the reference
    sizes every densified segment with int(15 -+ 1 ulp) (path_planning.py:398) and its point counts follow the last
    bits of the smoothing QP's output, which no other solver reproduces; 14.7 keeps the composition checkable exactly."""
    pu = t9.planning_utils
    orig = t9.path_planning.DP_algorithm
    if sample_s is not None:
        t9.path_planning.DP_algorithm = lambda *a, **k: orig(*a, **{**dict(sample_s=sample_s), **k})
    try:
        _run(t9, pu, fname)
    finally:
        t9.path_planning.DP_algorithm = orig


def _run(t9, pu, fname):
    rng = np.random.default_rng(2024)
    recs = []
    for c in range(18):
        case = c % 6
        req = make_request(rng, pu, case)
        conn = FakeConn(req)
        ref_loader.QP_LOG.clear()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                t9.motion_planning(conn)
        except _Done:
            pass
        except IndexError as e:                                   # the reference itself may raise (bound index)
            print("case", c, "reference raised IndexError:", e)
        ok = conn.reply is not None
        static, dynamic, veh, pred, v, a, path, mpl = req
        # the reference ignores cvxopt's status (path_planning.py:211-218) and sends whatever iterate it got
        qp_ok = all(r["status"] == "optimal" for r in ref_loader.QP_LOG)
        rec = dict(case=case, ok=ok, qp_ok=qp_ok, static=np.array(static + [(np.nan,) * 3] * (4 - len(static)))[:4],
                   n_static=len(static), dynamic=np.array(dynamic + [(np.nan,) * 4] * (2 - len(dynamic)))[:2],
                   n_dynamic=len(dynamic), veh=np.array(veh), pred=np.array(pred), v=np.array(v), a=np.array(a),
                   path=np.array(path), pre_match=int(mpl[0]))
        if ok:
            traj, match, ps, pl = conn.reply
            rec.update(traj=np.pad(np.array(traj, dtype=np.float64), ((0, 64 - len(traj)), (0, 0))), n_traj=len(traj),
                       match=int(match[0]), path_s=np.pad(np.array(ps, dtype=np.float64), (0, 64 - len(ps))), n_path=len(ps),
                       path_l=np.pad(np.array(pl, dtype=np.float64), (0, 64 - len(pl))))
        else:
            rec.update(traj=np.zeros((64, 4)), n_traj=0, match=-1, path_s=np.zeros(64), n_path=0, path_l=np.zeros(64))
        recs.append(rec)
        print("case", c, "kind", case, "reply" if ok else "no reply", "traj points", rec["n_traj"])
    out = {k: np.stack([np.asarray(r[k]) for r in recs]) for k in recs[0]}
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, os.path.getsize(os.path.join(OUT, fname)), "bytes")


def main():
    t9 = ref_loader.load_driver_reference()
    run(t9, None, "driver.npz")
    run(t9, 14.7, "driver_s147.npz")


if __name__ == "__main__":
    main()
