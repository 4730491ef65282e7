"""Golden vectors for the S-T speed planning back end (reference planner/speed_planning_test.py:308-620), generated
by importing the actual reference in this container (tests/golden/ref_loader.py).  Fixtures hold inputs and the
reference's outputs only.

* ``generate_convex_space``, ``increase_points``, ``path_speed_merge`` run as they are; where the reference raises
  (interp1d's ValueError, IndexError at s_ub[16] or at a trajectory without NaN padding) the exception type is
  recorded instead of outputs.
* ``speed_QP`` cannot get past its solver call (cvxopt rejects the untransposed equality matrix, see
  oracle/st_backend.py); the stub ``cvxopt.solvers.qp`` records the matrices it was handed and raises cvxopt's
  TypeError, and the remaining locals (dt, qp_size, the aliased bound vector) are read from the frame.

Inputs: DP speed profiles are the oracle's (oracle/st_speed.exact_speed_dp - the reference's own speed_DP raises in
its backtrack), obstacle S-T segments the reference's generate_st_graph, paths are synthetic arcs.

Run:  python tests/golden/make_golden_speed_backend.py        (writes tests/golden/speed_backend.npz)
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
#: where the .npz files go: this directory, or a scratch one for tests/test_reference_live.py (regenerate and compare)
OUT = os.environ.get("EMP_GOLDEN_OUT", HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from emplanner_carla_amd import scenes as S  # noqa: E402
from oracle import st_backend, st_speed  # noqa: E402

RAISES = {None: 0, "ValueError": 1, "IndexError": 2, "TypeError": 3}
MAX_PATH = 96


def call(fn, *a):
    try:
        return fn(*a), 0
    except (ValueError, IndexError) as exc:
        return None, RAISES[type(exc).__name__]


def speed_qp_locals(sp, *a):
    """Run the reference's speed_QP to its solver call; return the recorded matrices and the frame's locals."""
    n0 = len(ref_loader.QP_LOG)
    try:
        sp.speed_QP(*a)
    except TypeError as exc:
        tb = exc.__traceback__
        while tb is not None and tb.tb_frame.f_code.co_name != "speed_QP":
            tb = tb.tb_next
        loc = tb.tb_frame.f_locals
        rec = ref_loader.QP_LOG[-1]
        assert len(ref_loader.QP_LOG) == n0 + 1 and rec["status"] == "rejected"
        return dict(H=rec["P"], f=rec["q"], A=rec["G"], b=rec["h"], Aeq=rec["A"], beq=rec["b"], dt=loc["dt"],
                    qp_size=loc["qp_size"], bound=np.asarray(loc["lb"]).reshape(-1), aliased=loc["lb"] is loc["ub"]), 3
    except IndexError:
        return None, 2
    raise RuntimeError("speed_QP returned")


def main():
    sp = ref_loader.load_speed_reference()
    rng = np.random.default_rng(2024)
    B = 96
    o = S.make_dynamic_batch(range(300, 300 + B))
    sets = [np.stack([np.asarray(sp.generate_st_graph(o[0][b], o[1][b], o[2][b], o[3][b])[i]) for b in range(B)])
            for i in range(4)]
    ex = st_speed.exact_speed_dp(*sets, o[4])
    dp_s, dp_t = ex["speed_s"].copy(), ex["speed_t"].copy()
    # a third of the profiles get a shorter horizon (NaN tail), which is the only shape speed_QP accepts
    for b in range(0, B, 3):
        cut = int(rng.integers(6, 15))
        dp_s[b, cut:] = np.nan
        dp_t[b, cut:] = np.nan
    # obstacles that cross the ego's lane after the (shortened) DP horizon make the reference's interp1d raise;
    # in two cases of three they are removed so that enough cases run through
    for b in range(B):
        if b % 3 == 2:
            continue
        valid = dp_t[b][~np.isnan(dp_t[b])]
        horizon = valid[-2] if len(valid) < 16 else valid[-1]
        late = (sets[2][b] + sets[3][b]) / 2 > horizon
        for a in sets:
            a[b, late] = np.nan
    g = dict(v0=o[4], dp_s=dp_s, dp_t=dp_t, s_in=sets[0], s_out=sets[1], t_in=sets[2], t_out=sets[3])
    # paths: index2s (cumulative arc length, zero padded in half of the cases) and curvature
    n_path = rng.integers(60, 90, B)
    idx2s = np.zeros((B, MAX_PATH))
    kappa = np.zeros((B, MAX_PATH))
    path_len = np.zeros(B, np.int32)                     # len() of the list handed to the reference
    for b in range(B):
        step = rng.uniform(0.8, 1.3)
        s = np.concatenate(([0.0], np.cumsum(rng.uniform(0.9, 1.1, n_path[b] - 1) * step)))
        idx2s[b, :n_path[b]] = s
        kappa[b, :n_path[b]] = 0.02 * np.sin(s / rng.uniform(8, 25) + rng.uniform(0, 6))
        path_len[b] = MAX_PATH if b % 2 else n_path[b]   # odd cases: zero-padded buffer, even: exact length
    g.update(path_index2s=idx2s, path_kappa=kappa, path_len=path_len)
    cs = np.full((B, 4, 16), np.nan)
    cs_raise = np.zeros(B, np.int32)
    for b in range(B):
        n = path_len[b]
        out, code = call(sp.generate_convex_space, dp_s[b], dp_t[b], idx2s[b, :n], sets[0][b], sets[1][b], sets[2][b],
                         sets[3][b], kappa[b, :n])
        cs_raise[b] = code
        if out is not None:
            cs[b] = np.stack(out)
    g.update(cs_out=cs, cs_raise=cs_raise)
    print("convex space: raises", np.bincount(cs_raise, minlength=3).tolist())

    # speed_QP formulation on every case whose convex space exists
    qp_code = np.zeros(B, np.int32)
    qp_size = np.zeros(B, np.int32)
    qp_dt = np.full(B, np.nan)
    H = np.zeros((B, 51, 51)); f = np.zeros((B, 51)); A = np.zeros((B, 16, 51)); Aeq = np.zeros((B, 51, 32))
    bound = np.full((B, 51), np.nan)
    a0 = rng.uniform(-1.0, 1.0, B)
    for b in range(B):
        if cs_raise[b]:
            qp_code[b] = -1
            continue
        rec, code = speed_qp_locals(sp, float(o[4][b]), float(a0[b]), dp_s[b], dp_t[b], *cs[b])
        qp_code[b] = code
        if rec is None:
            continue
        n = rec["qp_size"]
        qp_size[b], qp_dt[b] = n, rec["dt"]
        assert rec["aliased"] and rec["H"].shape == (3 * n, 3 * n) and rec["Aeq"].shape == (3 * n, 2 * n - 2)
        assert not rec["b"].any() and not rec["beq"].any()
        H[b, :3 * n, :3 * n] = rec["H"]
        f[b, :3 * n] = rec["f"].reshape(-1)
        A[b, :n - 1, :3 * n] = rec["A"]
        Aeq[b, :3 * n, :2 * n - 2] = rec["Aeq"]
        bound[b, :3 * n] = rec["bound"]
    g.update(qp_a0=a0, qp_code=qp_code, qp_size=qp_size, qp_dt=qp_dt, qp_H=H, qp_f=f, qp_A=A, qp_Aeq=Aeq, qp_bound=bound)
    print("speed_QP: codes", {int(c): int((qp_code == c).sum()) for c in np.unique(qp_code)})

    # increase_points / path_speed_merge on profiles of the intended QP (any smooth profile would do)
    prof = np.full((B, 4, 17), np.nan)
    dense = np.full((B, 4, 401), np.nan)
    dense_raise = np.zeros(B, np.int32)
    for b in range(B):
        if qp_code[b] != 3:
            dense_raise[b] = -1
            continue
        (qs, qv, qa, qt), res, _ = st_backend.speed_qp(float(o[4][b]), float(a0[b]), dp_s[b], dp_t[b], *cs[b])
        if res is None or res.status != "optimal":        # infeasible corridor: no profile to densify
            dense_raise[b] = -1
            continue
        prof[b] = np.stack([qs, qv, qa, qt])
        out, code = call(sp.increase_points, qs, qv, qa, qt)
        dense_raise[b] = code
        if out is not None:
            dense[b] = np.stack(out)
    g.update(prof=prof, dense_out=dense, dense_raise=dense_raise)
    print("increase_points: cases", int((dense_raise == 0).sum()))

    # merge: path arrays NaN padded to MAX_PATH (one case without padding -> IndexError)
    px = np.full((B, MAX_PATH), np.nan); py = px.copy(); ph = px.copy(); pk = px.copy(); ps = np.zeros((B, MAX_PATH))
    merged = np.full((B, 7, 401), np.nan)
    merge_raise = np.zeros(B, np.int32)
    now = rng.uniform(0, 100, B)
    for b in range(B):
        if dense_raise[b] != 0:
            merge_raise[b] = -1
            continue
        n = n_path[b]
        s = idx2s[b, :n]
        th = np.cumsum(kappa[b, :n] * np.gradient(s))
        px[b, :n] = np.cumsum(np.cos(th) * np.gradient(s))
        py[b, :n] = np.cumsum(np.sin(th) * np.gradient(s))
        ph[b, :n] = th
        pk[b, :n] = kappa[b, :n]
        ps[b, :n] = s
        width = n if b == 1 else MAX_PATH                 # case 1: no NaN padding
        out, code = call(sp.path_speed_merge, *dense[b], float(now[b]), ps[b, :width], px[b, :width], py[b, :width],
                         ph[b, :width], pk[b, :width])
        merge_raise[b] = code
        if out is not None:
            merged[b] = np.stack(out)
    g.update(merge_now=now, merge_path_s=ps, merge_x=px, merge_y=py, merge_heading=ph, merge_kappa=pk, merge_n=n_path,
             merge_out=merged, merge_raise=merge_raise)
    print("path_speed_merge: raises", {int(c): int((merge_raise == c).sum()) for c in np.unique(merge_raise)})
    path = os.path.join(OUT, "speed_backend.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
