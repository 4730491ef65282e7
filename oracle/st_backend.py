"""CPU oracle for the S-T speed planning back end.

TEST INFRASTRUCTURE ONLY (same import rule as ``oracle/ref_port.py``).

Restates reference ``planner/speed_planning_test.py``:

    generate_convex_space   :308-407   s / s_dot bounds per DP column from curvature and the obstacles' S-T segments
    speed_QP                :410-511   3 x qp_size variable QP (s, s_dot, s_dot2 per time station)
    increase_points         :514-566   densification to 401 samples
    path_speed_merge        :569-620   path x speed profile -> 401-point trajectory

Pinning.  ``generate_convex_space``, ``increase_points`` and ``path_speed_merge`` run in the imported reference;
``tests/golden/make_golden_speed_backend.py`` commits their inputs and outputs (and the exception type where they
raise), and the ``port_*`` functions below are bit-identical to them.  ``speed_QP`` has NEVER run anywhere: it hands
``cvxopt.solvers.qp`` the equality matrix untransposed (30 x 18 instead of 18 x 30, :503), which cvxopt rejects with
a TypeError before solving, it never passes the bounds it builds, and ``ub = lb`` (:443) makes the two bound vectors
one object.  What IS pinned is everything the reference computes before that call: H, f, A, b, Aeq, beq, dt and
qp_size of the golden cases are matched bit for bit by ``speed_qp_formulation``.  **Parity unpinned** beyond that:
``speed_qp`` solves the problem the code evidently means (its comments and the MATLAB call it was translated from,
``quadprog(H, f, A, b, Aeq', beq, lb, ub)``):

    minimise   sum_i  w_a s_dot2_i^2 + w_v (s_dot_i - v_ref)^2  +  sum_i w_j (s_dot2_{i+1} - s_dot2_i)^2
    subject to Aeq' X = beq   (piecewise-linear acceleration: s, s_dot continuous, :452-458)
               s_i - s_{i+1} <= 0                                        (:463-468)
               lb <= X <= ub with SEPARATE vectors: station 0 pinned to (0, v0, a0), station i >= 1 bounded by
               s_lb / s_ub / s_dot_lb / s_dot_ub of DP column i-1 and -6 <= s_dot2 <= 4   (:472-487)

The minimiser is unique (H is positive definite on the constraint space) and is certified through
``oracle/qp_dense.py``.

scipy's ``interp1d`` (linear, bounds_error=True, assume_sorted=False) is restated in ``interp1d_linear`` with its
exact arithmetic: ``slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope * (x_new - x_lo) + y_lo`` on the interval chosen
by ``searchsorted(x, x_new)`` clipped to [1, len-1].
"""
from __future__ import annotations

import numpy as np

from . import qp_dense

N_DP = 16          # DP columns (:318)
N_QP = 17          # QP stations incl. the planning start (:428)
N_DENSE = 401      # :541, :576


class OutOfRange(ValueError):
    """interp1d's bounds error (ValueError in scipy)."""


def interp1d_linear(x, y, x_new):
    """scipy.interpolate.interp1d(x, y)(x_new) for scalar x_new."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if len(x) != len(y):
        raise ValueError("x and y arrays must be equal in length along interpolation axis.")
    if len(x) < 1:
        raise ValueError("x and y arrays must have at least 1 entry")
    order = np.argsort(x, kind="mergesort")
    x, y = x[order], y[order]
    if x_new < x[0]:
        raise OutOfRange("A value in x_new is below the interpolation range.")
    if x_new > x[-1]:
        raise OutOfRange("A value in x_new is above the interpolation range.")
    if np.isnan(x_new):
        return np.nan
    idx = int(np.searchsorted(x, x_new))
    idx = min(max(idx, 1), len(x) - 1)
    lo, hi = idx - 1, idx
    slope = (y[hi] - y[lo]) / (x[hi] - x[lo])
    return slope * (x_new - x[lo]) + y[lo]


def port_generate_convex_space(dp_speed_s, dp_speed_t, path_index2s, s_in, s_out, t_in, t_out, kappa,
                               max_lateral_accel=0.2 * 9.8):
    """:308-407.  Raises what the reference raises (OutOfRange = interp1d's ValueError, IndexError at s_ub[16])."""
    n = N_DP
    s_lb = np.ones(n) * -np.inf
    s_ub = np.ones(n) * np.inf
    sd_lb = np.ones(n) * -np.inf
    sd_ub = np.ones(n) * np.inf
    path_end = len(path_index2s)
    dp_end = len(dp_speed_s)
    for k in range(1, len(path_index2s)):                       # :326-330
        if path_index2s[k] == 0 and path_index2s[k - 1] != 0:
            path_end = k - 1
            break
        path_end = k
    for k in range(len(dp_speed_s)):                            # :333-336
        if np.isnan(dp_speed_s[k]):
            dp_end = k - 1
            break
    for i in range(n):                                          # :339-347
        if np.isnan(dp_speed_s[i]):
            break
        cur_kappa = interp1d_linear(path_index2s[0:path_end], kappa[0:path_end], dp_speed_s[i])
        sd_lb[i] = 0
        sd_ub[i] = np.sqrt(max_lateral_accel / (abs(cur_kappa) + 1e-10))
    for i in range(len(s_in)):                                  # :349-405
        if np.isnan(s_in[i]):
            continue
        obs_t = (t_in[i] + t_out[i]) / 2
        obs_s = (s_in[i] + s_out[i]) / 2
        obs_speed = (s_out[i] - s_in[i]) / (t_out[i] - t_in[i])
        dp_s = interp1d_linear([0] + list(dp_speed_t[0:dp_end]), [0] + list(dp_speed_s[0:dp_end]), obs_t)
        t_lb_index = _time_index(dp_speed_t, t_in[i])
        t_ub_index = _time_index(dp_speed_t, t_out[i])
        t_lb_index = max(t_lb_index - 2, 3)
        t_ub_index = min(t_ub_index + 2, dp_end)
        if obs_s > dp_s:
            for m in range(t_lb_index, t_ub_index + 1):
                s_ub[m] = min(s_ub[m], s_in[i] + obs_speed * (dp_speed_t[m] - t_in[i]))
        else:
            for m in range(t_lb_index, t_ub_index + 1):
                s_lb[m] = max(s_lb[m], s_in[i] + obs_speed * (dp_speed_t[m] - t_in[i]))
    return s_lb, s_ub, sd_lb, sd_ub


def _time_index(dp_speed_t, t):
    """:361-382 - the DP column whose [t_j, t_j+1) holds t; 0 if t lies before the first or after the last column
    (comparisons with a NaN column time are False, so NaN columns never match)."""
    for j in range(len(dp_speed_t) - 1):
        if dp_speed_t[0] > t:
            return j
        if dp_speed_t[j] <= t < dp_speed_t[j + 1]:
            return j
    return 0


def speed_qp_formulation(plan_start_s_dot, plan_start_s_dot2, dp_speed_s, dp_speed_t, s_lb, s_ub, s_dot_lb, s_dot_ub,
                         w_cost_s_dot2=10, w_cost_v_ref=50, w_cost_jerk=500, reference_speed=50):
    """:424-500 - everything the reference builds before its (failing) solver call, plus the bounds as they were
    meant (separate lb / ub).  Raises IndexError like the reference when the DP result has no NaN tail (:435)."""
    dp_speed_end = 16
    for i in range(len(dp_speed_s)):
        if np.isnan(dp_speed_s[i]):
            dp_speed_end = i - 1
            break
    s_end = dp_speed_s[dp_speed_end]                             # IndexError for a full-length DP result
    recommend_T = dp_speed_t[dp_speed_end]
    qp_size = dp_speed_end + 1
    nx = 3 * qp_size
    Aeq = np.zeros((nx, 2 * qp_size - 2))
    beq = np.zeros((2 * qp_size - 2, 1))
    dt = recommend_T / dp_speed_end
    A_sub = np.array([[1, 0], [dt, 1], [(1 / 3) * dt ** 2, (1 / 2) * dt], [-1, 0], [0, -1], [(1 / 6) * dt ** 2, dt / 2]])
    for i in range(qp_size - 1):
        Aeq[3 * i:3 * i + 6, 2 * i:2 * i + 2] = A_sub
    A = np.zeros((qp_size - 1, nx))
    b = np.zeros((qp_size - 1, 1))
    for i in range(qp_size - 1):
        A[i, 3 * i] = 1
        A[i, 3 * i + 3] = -1
    lb = np.ones(nx)
    ub = np.ones(nx)
    for i in range(1, qp_size):
        lb[3 * i], lb[3 * i + 1], lb[3 * i + 2] = s_lb[i - 1], s_dot_lb[i - 1], -6
        ub[3 * i], ub[3 * i + 1], ub[3 * i + 2] = s_ub[i - 1], s_dot_ub[i - 1], 4
    lb[0:3] = (0, plan_start_s_dot, plan_start_s_dot2)
    ub[0:3] = lb[0:3]
    A_s_dot2 = np.zeros((nx, nx))
    A_jerk = np.zeros((nx, qp_size - 1))
    A_ref = np.zeros((nx, nx))
    A4_sub = np.array([[0], [0], [1], [0], [0], [-1]])
    for i in range(1, qp_size + 1):
        A_s_dot2[3 * i - 1, 3 * i - 1] = 1
        A_ref[3 * i - 2, 3 * i - 2] = 1
    for i in range(1, qp_size):
        A_jerk[3 * i - 3:3 * i + 3, i - 1:i] = A4_sub
    H = w_cost_s_dot2 * (A_s_dot2 @ A_s_dot2.T) + w_cost_jerk * (A_jerk @ A_jerk.T) + w_cost_v_ref * (A_ref @ A_ref.T)
    H = 2 * H
    f = np.zeros((nx, 1))
    for i in range(1, qp_size + 1):
        f[3 * i - 2] = -2 * w_cost_v_ref * reference_speed
    return dict(H=H, f=f, A=A, b=b, Aeq=Aeq, beq=beq, lb=lb, ub=ub, dt=dt, qp_size=qp_size, dp_speed_end=dp_speed_end,
                s_end=s_end, recommend_T=recommend_T)


def speed_qp(plan_start_s_dot, plan_start_s_dot2, dp_speed_s, dp_speed_t, s_lb, s_ub, s_dot_lb, s_dot_ub, **weights):
    """The intended speed QP (see the module header) -> (qp_s, qp_s_dot, qp_s_dot2, relative_time) [17], NaN padded,
    the dense solver's result record (None if the problem is infeasible) and the formulation."""
    F = speed_qp_formulation(plan_start_s_dot, plan_start_s_dot2, dp_speed_s, dp_speed_t, s_lb, s_ub, s_dot_lb,
                             s_dot_ub, **weights)
    nx = 3 * F["qp_size"]
    eye = np.eye(nx)
    fixed = F["lb"] == F["ub"]
    rows_eq = [F["Aeq"].T] + [eye[fixed]]
    rhs_eq = [F["beq"].reshape(-1)] + [F["lb"][fixed]]
    upper = np.isfinite(F["ub"]) & ~fixed
    lower = np.isfinite(F["lb"]) & ~fixed
    G = np.concatenate([F["A"], eye[upper], -eye[lower]])
    h = np.concatenate([F["b"].reshape(-1), F["ub"][upper], -F["lb"][lower]])
    out = [np.ones(N_QP) * np.nan for _ in range(4)]
    if (F["lb"] > F["ub"]).any():                               # an empty corridor: nothing to solve
        return tuple(out), None, F
    try:
        res = qp_dense.solve_qp(F["H"], F["f"].reshape(-1), G, h, np.concatenate(rows_eq), np.concatenate(rhs_eq))
    except np.linalg.LinAlgError:                               # the dense IPM diverged: infeasible problem
        return tuple(out), None, F
    X = np.asarray(res.x).reshape(-1)
    for i in range(1, F["qp_size"] + 1):                       # :506-510
        out[0][i - 1] = X[3 * i - 3]
        out[1][i - 1] = X[3 * i - 2]
        out[2][i - 1] = X[3 * i - 1]
        out[3][i - 1] = (i - 1) * F["dt"]
    return tuple(out), res, F


def port_increase_points(s_init, s_dot_init, s_dot2_init, relative_time_init):
    """:514-566 (the first sample lies at -dt, the last interval is never searched: both kept)."""
    t_end = len(relative_time_init)
    for i in range(len(relative_time_init)):
        if np.isnan(relative_time_init[i]):
            t_end = i - 1
            break
    T = relative_time_init[t_end]
    n = N_DENSE
    dt = T / (n - 1)
    s, s_dot, s_dot2, rel = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    tmp = 0
    for i in range(n):
        current_t = (i - 1) * dt
        for j in range(t_end - 1):
            if relative_time_init[j] <= current_t < relative_time_init[j + 1]:
                tmp = j
                break
        x = current_t - relative_time_init[tmp]
        s[i] = s_init[tmp] + s_dot_init[tmp] * x + (1 / 3) * s_dot2_init[tmp] * x ** 2 + (1 / 6) * s_dot2_init[tmp + 1] * x ** 2
        s_dot[i] = s_dot_init[tmp] + 0.5 * s_dot2_init[tmp] * x + 0.5 * s_dot2_init[tmp + 1] * x
        s_dot2[i] = s_dot2_init[tmp] + (s_dot2_init[tmp + 1] - s_dot2_init[tmp]) * x / (
            relative_time_init[tmp + 1] - relative_time_init[tmp])
        rel[i] = current_t
    return s, s_dot, s_dot2, rel


def np_interp(x, xp, fp):
    """numpy.interp for scalar x (compiled loop of numpy/core/src/multiarray/compiled_base.c: clamped ends, exact hit
    on a knot returns the knot value, otherwise slope * (x - xp[j]) + fp[j])."""
    n = len(xp)
    if n == 0:
        raise ValueError("array of sample points is empty")
    if np.isnan(x):
        return x
    if x <= xp[0] if n == 1 else x < xp[0]:
        return fp[0]
    if x > xp[n - 1]:
        return fp[n - 1]
    if n == 1:
        return fp[0]
    j = int(np.searchsorted(xp, x, side="right")) - 1
    if j >= n - 1:
        return fp[n - 1]
    if xp[j] == x:
        return fp[j]
    slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j])
    r = slope * (x - xp[j]) + fp[j]
    if np.isnan(r):
        r = slope * (x - xp[j + 1]) + fp[j + 1]
        if np.isnan(r) and fp[j] == fp[j + 1]:
            r = fp[j]
    return r


def port_path_speed_merge(s, s_dot, s_dot2, relative_time, current_time, path_s, x_init, y_init, heading_init, kappa_init):
    """:569-620.  IndexError if ``x_init`` has no NaN (the reference scans for one, :586)."""
    n = N_DENSE
    out = [np.zeros(n) for _ in range(7)]      # x, y, heading, kappa, speed, accel, time
    index = 0
    while not np.isnan(x_init[index]):
        index += 1
    index -= 1
    for i in range(n - 1):
        out[0][i] = np_interp(s[i], path_s[:index], x_init[:index])
        out[1][i] = np_interp(s[i], path_s[:index], y_init[:index])
        out[2][i] = np_interp(s[i], path_s[:index], heading_init[:index])
        out[3][i] = np_interp(s[i], path_s[:index], kappa_init[:index])
        out[6][i] = relative_time[i] + current_time
        out[4][i] = s_dot[i]
        out[5][i] = s_dot2[i]
    out[0][-1] = x_init[-1]
    out[1][-1] = y_init[-1]
    out[2][-1] = heading_init[-1]
    out[3][-1] = kappa_init[-1]
    out[6][-1] = relative_time[-1] + current_time
    out[4][-1] = s_dot[-1]
    out[5][-1] = s_dot2[-1]
    return tuple(out)
