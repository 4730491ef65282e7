"""CPU oracle, mode "exact": the arithmetic the HIP kernels implement, in vectorised NumPy.

TEST INFRASTRUCTURE ONLY (same import rule as ``oracle/ref_port.py``).

The reference fits every lattice quintic by numerically inverting a 6x6 matrix in
*absolute* s (reference planner/planning_utils.py:692-703), which is ill-conditioned
(cond ~ 5e15 at s = 75 m) and therefore carries machine-dependent noise of up to ~1e-6
relative in its own edge costs at s ~ 100 m.  This module restates the same
mathematics with the closed-form quintic in the shifted coordinate ``t = s - s0`` and
rebuilds only what the reference's ``dddl`` quirk needs (the absolute-basis c3, c4, c5;
reference planner/path_planning.py:498,571).  It is validated against the golden
vectors of the imported reference (index-exact rows, 1e-6 relative values) and is in turn
the *bit-exact* target for the GPU DP kernels: every expression below is evaluated in the
written order with separately rounded IEEE-754 multiplies and adds (NumPy never fuses),
and the kernels are compiled with floating-point contraction off for these expressions.

Operation order (one lattice edge from (s0, l0, dl0, ddl0) to (s0 + T, l1, 0, 0)), T = sample_s:
    h  = l1 - l0
    a3 = ((20 h - (12 dl0) T) - (3 ddl0) T^2) / (2 T^3)      a0 = l0
    a4 = ((-30 h + (16 dl0) T) + (3 ddl0) T^2) / (2 T^4)      a1 = dl0
    a5 = ((12 h - (6 dl0) T) - ddl0 T^2) / (2 T^5)            a2 = 0.5 ddl0
    t_i = (i * sample_s) / 10,  s_i = s0 + t_i,  i = 0..9     (reference :493/:566)
    l, dl, ddl by Horner in t_i (highest coefficient first)
    c5 = a5,  c4 = a4 - (5 a5) s0,  c3 = (a3 - (4 a4) s0) + ((10 a5) s0) s0
    dddl_i = 6 c3 + (24 c4) s_i + (60 c5) (s_i * 2)           (the quirk) is linear in s_i: dddl_i = A + K1 t_i with
        K1 = 24 c4 + (60 c5) 2,  A = 6 c3 + K1 s0;  its sum of squares in closed form, T1 = sum t_i, T2 = sum t_i^2 (ascending):
        S_dddl = (10 (A A) + (2 A) (K1 T1)) + (K1 K1) T2
    (until round 2 the kernels summed dddl_i^2 sample by sample as the reference does; the closed form is the same
     mathematics, differs in the last bits, and is a quarter of the edge kernel's instructions cheaper)
    other sums over i ascending; cost = ((w0 S_dl + w1 S_ddl) + w2 S_dddl + collision) + w_ref S_l
    NEIGHBOUR edges (dl0 = ddl0 = 0; round 5): their coefficients are h times the unit quintic's (u3, u4, u5) =
    quintic_shifted(0, 0, 0, 1, T), and the quirk term is linear in them, so S_dddl = (h h) F(s0) with
        F(s0) = S_dddl of the unit quintic at s0 (the closed form above, operation for operation),
    evaluated once per (scene, column); the edge's smoothness term is (w0 S_dl + w1 S_ddl) + (w2 (h h)) F(s0).
    Same mathematics as the closed form on the edge's own coefficients (rounds 2-4), different in the last bits, and 26 of
    the ~100 vector instructions an edge costs the kernel outside its obstacle scans.  Start edges keep the general form.
    collision = sum over obstacles in order of the ordered, early-breaking scan (:588-609)
"""
from __future__ import annotations

import numpy as np


def quintic_shifted(l0, dl0, ddl0, l1, T):
    """Closed-form coefficients a0..a5 in t = s - s0 for end state (l1, 0, 0) at t = T."""
    l0, dl0, ddl0, l1, T = np.broadcast_arrays(*(np.asarray(v, dtype=np.float64) for v in (l0, dl0, ddl0, l1, T)))
    h = l1 - l0
    T2 = T * T
    T3 = T2 * T
    T4 = T3 * T
    T5 = T4 * T
    a3 = ((20.0 * h - (12.0 * dl0) * T) - (3.0 * ddl0) * T2) / (2.0 * T3)
    a4 = ((-30.0 * h + (16.0 * dl0) * T) + (3.0 * ddl0) * T2) / (2.0 * T4)
    a5 = ((12.0 * h - (6.0 * dl0) * T) - ddl0 * T2) / (2.0 * T5)
    return l0, dl0, 0.5 * ddl0, a3, a4, a5


def _segment_cost(a, s0, sample_s, obs_s, obs_l, n_obs, w_coll, w_smooth, w_ref, weighted_jerk=None):
    """Edge cost for arrays of edges.  ``weighted_jerk``: the term w2 S_dddl computed by the caller (neighbour edges:
    (w2 (h h)) F(s0), see the module docstring) instead of the general closed form.

    a        : tuple of 6 arrays, each shape E (any shape), shifted coefficients
    s0       : start s per edge (same shape)
    obs_s/l  : shape E + (max_obs,) broadcastable; n_obs shape E broadcastable
    """
    a0, a1, a2, a3, a4, a5 = a
    shape = np.broadcast(a0, s0).shape
    c5 = a5
    c4 = a4 - (5.0 * a5) * s0
    c3 = (a3 - (4.0 * a4) * s0) + ((10.0 * a5) * s0) * s0
    k_dl = (5.0 * a5, 4.0 * a4, 3.0 * a3, 2.0 * a2)
    k_ddl = (20.0 * a5, 12.0 * a4, 6.0 * a3, 2.0 * a2)
    k_d3 = (6.0 * c3, 24.0 * c4, 60.0 * c5)
    S_l = np.zeros(shape)
    S_dl = np.zeros(shape)
    S_ddl = np.zeros(shape)
    T1 = 0.0
    T2 = 0.0
    for i in range(10):
        t = (i * sample_s) / 10.0
        T1 = T1 + t
        T2 = T2 + t * t
    K1 = k_d3[1] + k_d3[2] * 2.0
    A = k_d3[0] + K1 * s0
    S_d3 = (10.0 * (A * A) + (2.0 * A) * (K1 * T1)) + (K1 * K1) * T2
    max_obs = obs_s.shape[-1]
    coll_each = np.zeros(shape + (max_obs,))
    alive = np.ones(shape + (max_obs,), dtype=bool)           # obstacle scan not yet broken
    for i in range(10):
        t = (i * sample_s) / 10.0
        s = s0 + t
        p = a5
        for c in (a4, a3, a2, a1, a0):
            p = c + t * p
        q = k_dl[0]
        for c in (k_dl[1], k_dl[2], k_dl[3], a1):
            q = c + t * q
        r = k_ddl[0]
        for c in (k_ddl[1], k_ddl[2], k_ddl[3]):
            r = c + t * r
        S_l = S_l + p * p
        S_dl = S_dl + q * q
        S_ddl = S_ddl + r * r
        d_lon = obs_s - np.asarray(s)[..., None]
        d_lat = obs_l - np.asarray(p)[..., None]
        d2 = d_lon * d_lon + d_lat * d_lat
        hard = alive & (d2 <= 16.0)
        soft = alive & (d2 > 16.0) & (d2 < 36.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            coll_each = np.where(hard, coll_each + w_coll, np.where(soft, coll_each + 5000.0 / d2, coll_each))
        alive = alive & ~hard
    valid = np.arange(max_obs) < np.asarray(n_obs)[..., None]
    coll = np.zeros(shape)
    for m in range(max_obs):                                   # obstacles in order
        coll = coll + np.where(valid[..., m], coll_each[..., m], 0.0)
    smooth = (w_smooth[0] * S_dl + w_smooth[1] * S_ddl) + (w_smooth[2] * S_d3 if weighted_jerk is None else weighted_jerk)
    return (smooth + coll) + w_ref * S_l


def jerk_quirk_sum_sample_loop(a3, a4, a5, s0, sample_s):
    """The reference's third-derivative sum in ITS order (path_planning.py:492-499 / :565-572): per sample
    s_i = s0 + i sample_s / 10 the term 6 c3 + 24 c4 s_i + 60 c5 (s_i * 2), squared and accumulated sample by sample -
    from the same exact absolute-s coefficients as the closed form in `_segment_cost`.  Not used by any parity test of
    the kernels (they and `_segment_cost` use the closed form 10 A^2 + 2 A K1 T1 + K1^2 T2, a different rounding of the
    same number); tests/test_oracle_golden.py checks that the two agree to a few ulp of the sum's largest term, so that
    the closed form cannot drift from the reference's arithmetic unnoticed."""
    a3, a4, a5, s0 = np.broadcast_arrays(*(np.asarray(v, dtype=np.float64) for v in (a3, a4, a5, s0)))
    c5 = a5
    c4 = a4 - (5.0 * a5) * s0
    c3 = (a3 - (4.0 * a4) * s0) + ((10.0 * a5) * s0) * s0
    total = np.zeros(a3.shape)
    for i in range(10):
        s = s0 + (i * sample_s) / 10.0
        d3 = (6.0 * c3 + (24.0 * c4) * s) + (60.0 * c5) * (s * 2.0)
        total = total + d3 * d3
    return total


def jerk_quirk_sum_closed_form(a3, a4, a5, s0, sample_s):
    """The closed form as `_segment_cost` and csrc/emp_core.h (jerk_quirk_sum) evaluate it, operation for operation."""
    a3, a4, a5, s0 = np.broadcast_arrays(*(np.asarray(v, dtype=np.float64) for v in (a3, a4, a5, s0)))
    c4 = a4 - (5.0 * a5) * s0
    c3 = (a3 - (4.0 * a4) * s0) + ((10.0 * a5) * s0) * s0
    T1 = T2 = 0.0
    for i in range(10):
        t = (i * sample_s) / 10.0
        T1 = T1 + t
        T2 = T2 + t * t
    K1 = 24.0 * c4 + (60.0 * a5) * 2.0
    A = 6.0 * c3 + K1 * s0
    return (10.0 * (A * A) + (2.0 * A) * (K1 * T1)) + (K1 * K1) * T2


def lattice_l(row, sample_l):
    """Lateral offset of lattice row i (reference path_planning.py:326)."""
    return ((row + 1) / 2 - 1 - np.arange(row)) * sample_l


def edge_costs(obs_s, obs_l, n_obs, start, row, col, sample_s, sample_l,
               w_coll=1e12, w_smooth=(300.0, 1000.0, 5000.0), w_ref=20.0):
    """All lattice edge costs for a batch.

    obs_s, obs_l : (B, max_obs);  n_obs : (B,);  start : (B, 4) = s, l, dl, ddl
    returns start_cost (B, row) and edge (B, col-1, row_i, row_k): edge[b, j-1, i, k] is the
    cost of moving from row k of column j-1 to row i of column j (k fastest, the layout the
    sweep kernel streams).
    """
    obs_s = np.asarray(obs_s, dtype=np.float64)
    obs_l = np.asarray(obs_l, dtype=np.float64)
    start = np.asarray(start, dtype=np.float64)
    B = start.shape[0]
    n_obs = np.asarray(n_obs)
    ll = lattice_l(row, sample_l)
    T = float(sample_s)
    ps, pl, pdl, pddl = (start[:, k] for k in range(4))
    # start edges: (B, row)
    a = quintic_shifted(pl[:, None], pdl[:, None], pddl[:, None], ll[None, :], T)
    c0 = _segment_cost(a, ps[:, None], sample_s, obs_s[:, None, :], obs_l[:, None, :], n_obs[:, None],
                       w_coll, w_smooth, w_ref)
    # neighbour edges: (B, col-1, row_i, row_k)
    j = np.arange(1, col)
    s0 = ps[:, None] + j[None, :] * sample_s                               # pre_node_s (:330)
    a = quintic_shifted(ll[None, None, None, :], 0.0, 0.0, ll[None, None, :, None], T)
    a = tuple(np.broadcast_to(x, (1, 1, row, row)) for x in a)
    h = ll[:, None] - ll[None, :]                                          # l_cur - l_pre, [i][k] (as quintic_shifted)
    e = _segment_cost(a, s0[:, :, None, None], sample_s, obs_s[:, None, None, None, :],
                      obs_l[:, None, None, None, :], n_obs[:, None, None, None], w_coll, w_smooth, w_ref,
                      weighted_jerk=(w_smooth[2] * (h * h))[None, None] * neighbour_jerk_factor(s0, sample_s)[:, :, None, None])
    return c0, e


def neighbour_jerk_factor(s0, sample_s):
    """F(s0): the quirked third-derivative sum of the UNIT neighbour quintic (h = 1) that starts at s0 - what
    csrc/emp_dp_kernels.h jerk_unit_sum evaluates once per (scene, column)."""
    _, _, _, u3, u4, u5 = quintic_shifted(0.0, 0.0, 0.0, 1.0, float(sample_s))
    return jerk_quirk_sum_closed_form(u3, u4, u5, s0, sample_s)


def dp_sweep(c0, edge, row):
    """Min-plus sweep of reference path_planning.py:301-346 on a materialised edge tensor.

    returns cost (B, row, col) and pre_node_index (B, row, col) int32 (initialised to ones).
    """
    B = c0.shape[0]
    col = edge.shape[1] + 1
    left = (np.arange(row) < (row >> 1))
    cost = np.full((B, row, col), np.inf)
    pre = np.ones((B, row, col), dtype=np.int32)
    cost[:, :, 0] = np.where(left[None, :], c0 + 10000.0, c0)
    for j in range(1, col):
        cand = cost[:, None, :, j - 1] + edge[:, j - 1]                   # (B, i, k)
        cand = np.where(left[None, :, None], cand + 10000.0, cand)
        best = np.full((B, row), np.inf)
        arg = np.ones((B, row), dtype=np.int32)
        for k in range(row):                                               # strict <, k ascending
            better = cand[:, :, k] < best
            best = np.where(better, cand[:, :, k], best)
            arg = np.where(better, k, arg)
        cost[:, :, j] = best
        pre[:, :, j] = arg
    return cost, pre


def dp_backtrack(cost, pre, w_coll=1e12):
    """Reference path_planning.py:348-361: first minimum of the last column, then follow pre."""
    B, row, col = cost.shape
    rows = np.zeros((B, col), dtype=np.int32)
    idx = cost[:, :, -1].argmin(axis=1)
    feasible = ~(cost[:, :, -1].min(axis=1) > w_coll)
    rows[:, col - 1] = idx
    b = np.arange(B)
    for j in range(col - 1, 0, -1):
        idx = pre[b, idx, j]
        rows[:, j - 1] = idx
    return rows, feasible


def enrich(rows, start, row, col, sample_s, sample_l, resolution):
    """Reference path_planning.py:364-432 for ONE scene with the closed-form quintic.

    rows may be float (the no-obstacle bypass yields (row+1)/2-1, possibly x.5).
    returns (s list, l list).
    """
    ps, pl, pdl, pddl = (float(v) for v in start)
    dp_s = [ps + (i + 1) * sample_s for i in range(col)]
    dp_l = [((row + 1) / 2 - 1 - float(rows[i])) * sample_l for i in range(col)]
    out_s, out_l = [], []
    seg = (ps, pl, pdl, pddl)
    for i in range(col):
        if i > 0:
            seg = (dp_s[i - 1], dp_l[i - 1], 0.0, 0.0)
        s0, l0, dl0, ddl0 = seg
        span = dp_s[i] - s0
        count = len(np.arange(0, int(span), resolution))                   # :405 / :423
        a0, a1, a2, a3, a4, a5 = (float(v) for v in quintic_shifted(l0, dl0, ddl0, dp_l[i], span))
        for k in range(count):
            t = float(k * resolution)
            p = a5
            for c in (a4, a3, a2, a1, a0):
                p = c + t * p
            out_s.append(s0 + t)
            out_l.append(p)
    out_s.append(dp_s[-1])
    out_l.append(dp_l[-1])
    return out_s, out_l


def dp_plan(obs_s, obs_l, n_obs, start, row, col, sample_s, sample_l, sampling_res,
            w_coll=1e12, w_smooth=(300.0, 1000.0, 5000.0), w_ref=20.0, chunk=128):
    """Batched DP_algorithm: returns rows (B, col) float64, feasible (B,), list of (s, l) arrays."""
    start = np.asarray(start, dtype=np.float64)
    B = start.shape[0]
    n_obs = np.asarray(n_obs)
    rows = np.zeros((B, col))
    feasible = np.ones(B, dtype=bool)
    for lo in range(0, B, chunk):
        sl = slice(lo, min(B, lo + chunk))
        act = np.nonzero(n_obs[sl] > 0)[0] + lo
        byp = np.nonzero(n_obs[sl] == 0)[0] + lo
        rows[byp] = (row + 1) / 2 - 1                                      # bypass :363
        if len(act):
            c0, e = edge_costs(np.asarray(obs_s)[act], np.asarray(obs_l)[act], n_obs[act], start[act],
                               row, col, sample_s, sample_l, w_coll, w_smooth, w_ref)
            cost, pre = dp_sweep(c0, e, row)
            r, f = dp_backtrack(cost, pre, w_coll)
            rows[act] = r
            feasible[act] = f
    paths = [enrich(rows[b], start[b], row, col, sample_s, sample_l, sampling_res) for b in range(B)]
    return rows, feasible, paths
