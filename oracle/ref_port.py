"""CPU oracle, mode "faithful": a restatement of the reference's hot path in NumPy.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; nothing under
``emplanner_carla_amd/`` does.  It follows the reference's *control flow and floating-point
route* function by function (per-edge 6x6 ``inv``, absolute-``s`` polynomials, Python
loops with the same early exits), so that on the machine that produced the golden
vectors it agrees with the imported reference to round-off, and so that timing it gives
the "reference-structured CPU path" of BASELINE.md section 3.  The arithmetic the GPU
uses (closed-form quintic in a shifted coordinate) lives in ``oracle/exact.py``.

Parity status: every function below except the two QP solves is pinned by golden vectors
generated from the imported reference (tests/golden/make_golden.py).  The QP *formulation*
is pinned the same way; the QP *arithmetic* is **parity unpinned** - the reference calls
``cvxopt.solvers.qp`` (reference planner/path_planning.py:211, planning_utils.py:353),
cvxopt is absent from /root/reference, carries no pinned version and cannot be installed
here - and is replaced by ``oracle/qp_dense.py`` (KKT-certified, unique minimiser).

Each function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math

import numpy as np

from . import qp_dense

INFEASIBLE_BANNER = "********************     can't find a feasible path      ********************"


# --------------------------------------------------------------------------------------
# quintic + edge costs  (planner/planning_utils.py:671-703, planner/path_planning.py:435-609)
# --------------------------------------------------------------------------------------
def cal_quintic_coefficient(start_l, start_dl, start_ddl, end_l, end_dl, end_ddl, start_s, end_s):
    """planning_utils.py:671-703 - boundary-value matrix in absolute s, inverted numerically."""
    def rows(s):
        p2, p3, p4, p5 = pow(s, 2), pow(s, 3), pow(s, 4), pow(s, 5)
        return [[1, s, p2, p3, p4, p5],
                [0, 1, 2 * s, 3 * p2, 4 * p3, 5 * p4],
                [0, 0, 2, 6 * s, 12 * p2, 20 * p3]]
    M = np.array(rows(start_s) + rows(end_s))
    rhs = np.array([start_l, start_dl, start_ddl, end_l, end_dl, end_ddl]).reshape((6, 1))
    return list((np.linalg.inv(M) @ rhs).squeeze())


def cal_obs_cost(w_cost_collision, square_d, danger_dis=4, safe_dis=6):
    """path_planning.py:588-609 - ordered scan with early break on the first hard hit."""
    total = 0
    for d2 in np.asarray(square_d).squeeze():
        if d2 <= danger_dis ** 2:
            total += w_cost_collision
            break
        elif danger_dis ** 2 < d2 < safe_dis ** 2:
            total += 5000 / d2
    return total


def _segment_cost(obs_s_list, obs_l_list, c, start_s, sample_s, w_cost_collision, w_cost_smooth, w_cost_ref):
    """Shared tail of path_planning.py:486-514 and :559-585 (identical arithmetic)."""
    s = np.zeros(shape=(10, 1))
    for i in range(10):
        s[i][0] = start_s + i * sample_s / 10                              # :493 / :566
    l = c[0] + c[1] * s + c[2] * (s ** 2) + c[3] * (s ** 3) + c[4] * (s ** 4) + c[5] * (s ** 5)
    dl = c[1] + 2 * c[2] * s + 3 * c[3] * (s ** 2) + 4 * c[4] * (s ** 3) + 5 * c[5] * (s ** 4)
    ddl = 2 * c[2] + 6 * c[3] * s + 12 * c[4] * (s ** 2) + 20 * c[5] * (s ** 3)
    dddl = 6 * c[3] + 24 * c[4] * s + 60 * c[5] * (s * 2)                   # quirk :498 / :571
    smooth = (w_cost_smooth[0] * (dl.T @ dl) + w_cost_smooth[1] * (ddl.T @ ddl)
              + w_cost_smooth[2] * (dddl.T @ dddl))
    ref = w_cost_ref * (l.T @ l)
    collision = 0
    for k in range(len(obs_s_list)):
        d_lon = obs_s_list[k] - s
        d_lat = obs_l_list[k] - l
        collision += cal_obs_cost(w_cost_collision, d_lon ** 2 + d_lat ** 2)
    return smooth + collision + ref                                        # association of :514 / :585


def cal_start_cost(obs_s_list, obs_l_list, begin_s, begin_l, begin_dl, begin_ddl, cur_node_row, row,
                   sample_s, sample_l, w_cost_collision, w_cost_smooth, w_cost_ref):
    """path_planning.py:435-514."""
    end_l = ((row + 1) / 2 - 1 - cur_node_row) * sample_l                  # :478
    c = cal_quintic_coefficient(begin_l, begin_dl, begin_ddl, end_l, 0, 0, begin_s, begin_s + sample_s)
    return _segment_cost(obs_s_list, obs_l_list, c, begin_s, sample_s, w_cost_collision, w_cost_smooth,
                         w_cost_ref)


def cal_neighbor_cost(obs_s_list, obs_l_list, pre_node_s, pre_node_l, cur_node_s, cur_node_l, sample_s,
                      w_cost_collision, w_cost_smooth, w_cost_ref):
    """path_planning.py:517-585."""
    c = cal_quintic_coefficient(pre_node_l, 0, 0, cur_node_l, 0, 0, pre_node_s, cur_node_s)
    return _segment_cost(obs_s_list, obs_l_list, c, pre_node_s, sample_s, w_cost_collision, w_cost_smooth,
                         w_cost_ref)


# --------------------------------------------------------------------------------------
# DP + densification  (planner/path_planning.py:276-432)
# --------------------------------------------------------------------------------------
def dp_tables(obs_s_list, obs_l_list, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl,
              w_collision_cost=1e12, w_smooth_cost=(300, 1000, 5000), w_reference_cost=20,
              row=12, col=6, sample_s=15, sample_l=1.5):
    """Forward sweep of path_planning.py:301-346; returns (cost[row,col], pre_node_index[row,col])."""
    cost = np.ones(shape=(row, col)) * np.inf
    pre = np.ones(shape=(row, col), dtype="int32")                         # quirk :304 (ones)
    for i in range(row):
        c0 = cal_start_cost(obs_s_list, obs_l_list, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl,
                            i, row, sample_s, sample_l, w_collision_cost, w_smooth_cost, w_reference_cost)
        cost[i][0] = float(np.asarray(c0).reshape(-1)[0])
        if i < (row >> 1):                                                 # :317
            cost[i][0] += 10000
    for j in range(1, col):
        cur_s = plan_start_s + (j + 1) * sample_s                          # :325
        pre_s = plan_start_s + j * sample_s                                # :330
        for i in range(row):
            cur_l = ((row + 1) / 2 - 1 - i) * sample_l
            for k in range(row):
                pre_l = ((row + 1) / 2 - 1 - k) * sample_l
                nb = cal_neighbor_cost(obs_s_list, obs_l_list, pre_s, pre_l, cur_s, cur_l, sample_s,
                                       w_collision_cost, w_smooth_cost, w_reference_cost)
                tmp = cost[k][j - 1] + float(np.asarray(nb).reshape(-1)[0])  # :340
                if i < (row >> 1):
                    tmp += 10000                                           # :342
                if tmp < cost[i][j]:                                       # strict, k ascending :344
                    cost[i][j] = tmp
                    pre[i][j] = k
    return cost, pre


def dp_backtrack(cost, pre, w_collision_cost=1e12, verbose=True):
    """path_planning.py:348-361; returns (row index per column, feasible flag)."""
    col = cost.shape[1]
    idx = int(cost[:, -1].argmin())
    feasible = not (cost[:, -1].min() > w_collision_cost)
    if not feasible and verbose:
        print(INFEASIBLE_BANNER)
    chain = [idx]
    for j in range(col - 1, 0, -1):
        idx = int(pre[idx][j])
        chain.append(idx)
    chain.reverse()
    return chain, feasible


def enrich_DP_s_l(DP_s_list, DP_l_list, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl, resolution=1):
    """path_planning.py:378-432 - sample count from int() truncation of a float difference."""
    out_s, out_l = [], []
    seg_start = (plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl)
    end_s = end_l = None
    for i in range(len(DP_s_list)):
        if i > 0:
            seg_start = (DP_s_list[i - 1], DP_l_list[i - 1], 0, 0)
        s0, l0, dl0, ddl0 = seg_start
        end_s, end_l = DP_s_list[i], DP_l_list[i]
        c = cal_quintic_coefficient(l0, dl0, ddl0, end_l, 0, 0, s0, end_s)
        s = s0 + np.arange(0, int(end_s - s0), resolution)                 # :405 / :423
        l = c[0] + c[1] * s + c[2] * (s ** 2) + c[3] * (s ** 3) + c[4] * (s ** 4) + c[5] * (s ** 5)
        out_s += list(s)
        out_l += list(l)
    out_s += [end_s]
    out_l += [end_l]
    return out_s, out_l


def DP_algorithm(obs_s_list, obs_l_list, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl,
                 sampling_res=2, w_collision_cost=1e12, w_smooth_cost=[300, 1000, 5000], w_reference_cost=20,
                 row=12, col=6, sample_s=15, sample_l=1.5, _return_rows=False, _verbose=True):
    """path_planning.py:276-375."""
    feasible = True
    if len(obs_s_list):
        cost, pre = dp_tables(obs_s_list, obs_l_list, plan_start_s, plan_start_l, plan_start_dl,
                              plan_start_ddl, w_collision_cost, w_smooth_cost, w_reference_cost,
                              row, col, sample_s, sample_l)
        rows, feasible = dp_backtrack(cost, pre, w_collision_cost, _verbose)
    else:
        rows = list(np.ones(col) * ((row + 1) / 2 - 1))                    # bypass :363 (may be x.5)
    dp_s = [plan_start_s + (i + 1) * sample_s for i in range(len(rows))]   # :369
    dp_l = [((row + 1) / 2 - 1 - rows[i]) * sample_l for i in range(len(rows))]
    es, el = enrich_DP_s_l(dp_s, dp_l, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl,
                           resolution=sampling_res)
    if _return_rows:
        return es, el, rows, feasible
    return es, el


# --------------------------------------------------------------------------------------
# QP bounds and path QP  (planner/path_planning.py:78-273)
# --------------------------------------------------------------------------------------
def cal_lmin_lmax(dp_path_s, dp_path_l, obs_s_list, obs_l_list, obs_length, obs_width):
    """path_planning.py:222-273 - argmin+2 station offsets, unchecked upper index."""
    n = len(dp_path_s)
    lmin = -10 * np.ones(n)
    lmax = 10 * np.ones(n)
    s_arr = np.array(dp_path_s)
    for k in range(len(obs_s_list)):
        lo = np.argmin(np.abs(s_arr - (obs_s_list[k] - obs_length / 2))) + 2
        hi = np.argmin(np.abs(s_arr - (obs_s_list[k] + obs_length / 2))) + 2
        centre = np.argmin(np.abs(s_arr - obs_s_list[k]))
        if dp_path_l[centre] < obs_l_list[k]:
            for j in range(lo, hi + 1):
                lmax[j] = min(lmax[j], obs_l_list[k] - obs_width / 2)      # IndexError past n, like :267
        else:
            for j in range(lo, hi + 1):
                lmin[j] = max(lmin[j], obs_l_list[k] + obs_width / 2)
    return lmin, lmax


def path_qp_matrices(l_min, l_max, plan_start_l, plan_start_dl, plan_start_ddl, dp_sampling_res=2,
                     w_cost_l=1000, w_cost_dl=10000, w_cost_ddl=3000, w_cost_dddl=150, w_cost_centre=250,
                     w_cost_end_l=40, w_cost_end_dl=40, w_cost_end_ddl=40, host_d1=3, host_d2=3, host_w=3):
    """Dense (H, f, G, h, Aeq, beq) exactly as path_planning.py:103-205 assembles them."""
    n = len(l_min)
    ds = dp_sampling_res
    Aeq = np.zeros((2 * n - 2, 3 * n))
    beq = np.zeros((2 * n - 2, 1))
    link = np.array([[1, ds, ds ** 2 / 3, -1, 0, ds ** 2 / 6],
                     [0, 1, ds / 2, 0, -1, ds / 2]])
    for i in range(n - 1):
        Aeq[2 * i:2 * i + 2, 3 * i:3 * i + 6] = link
    A = np.zeros((8 * n, 3 * n))
    b = np.zeros((8 * n, 1))
    corner = np.array([[1, host_d1, 0], [1, host_d1, 0], [1, -host_d2, 0], [1, -host_d2, 0],
                       [-1, -host_d1, 0], [-1, -host_d1, 0], [-1, host_d2, 0], [-1, host_d2, 0]])
    fwd = math.ceil(host_d1 / ds)
    back = math.ceil(host_d2 / ds)
    for i in range(n):
        A[8 * i:8 * i + 8, 3 * i:3 * i + 3] = corner
        up = l_max[min(i + fwd, n - 1)]
        lo = l_min[max(i - back, 0)]
        b[8 * i:8 * i + 8, 0] = [up - host_w / 2, up + host_w / 2, up - host_w / 2, up + host_w / 2,
                                 -lo + host_w / 2, -lo - host_w / 2, -lo + host_w / 2, -lo - host_w / 2]
    lb = np.ones((3 * n, 1)) * (-100000)
    ub = np.ones((3 * n, 1)) * 100000
    lb[0:3, 0] = ub[0:3, 0] = [plan_start_l, plan_start_dl, plan_start_ddl]
    lb[3 * n - 3:3 * n, 0] = ub[3 * n - 3:3 * n, 0] = 0
    G = np.concatenate((A, np.identity(3 * n), -np.identity(3 * n)))
    h = np.concatenate((b, ub, -lb))
    sel_l = np.zeros((3 * n, 3 * n))
    sel_dl = np.zeros((3 * n, 3 * n))
    sel_ddl = np.zeros((3 * n, 3 * n))
    for i in range(n):
        sel_l[3 * i, 3 * i] = 1
        sel_dl[3 * i + 1, 3 * i + 1] = 1
        sel_ddl[3 * i + 2, 3 * i + 2] = 1
    jerk = np.zeros((n - 1, 3 * n))
    for i in range(n - 1):
        jerk[i, 3 * i:3 * i + 6] = [0, 0, -1, 0, 0, 1]
    end_l = np.zeros((3 * n, 3 * n)); end_l[3 * n - 3, 3 * n - 3] = 1
    end_dl = np.zeros((3 * n, 3 * n)); end_dl[3 * n - 2, 3 * n - 2] = 1
    end_ddl = np.zeros((3 * n, 3 * n)); end_ddl[3 * n - 1, 3 * n - 1] = 1
    # the w_cost_dl product is sel_dl' @ sel_l == 0: quirk of path_planning.py:193
    H = (w_cost_l * (sel_l.T @ sel_l) + w_cost_dl * (sel_dl.T @ sel_l) + w_cost_ddl * (sel_ddl.T @ sel_ddl)
         + w_cost_dddl * (jerk.T @ jerk) + w_cost_centre * (sel_l.T @ sel_l)
         + w_cost_end_l * (end_l.T @ end_l) + w_cost_end_dl * (end_dl.T @ end_dl)
         + w_cost_end_ddl * (end_ddl.T @ end_ddl))
    H = 2 * H
    f = np.zeros((3 * n, 1))
    centre = (np.array(l_min) + np.array(l_max)) / 2
    for i in range(n):
        f[3 * i] = -2 * centre[i]
    f = w_cost_centre * f
    return H, f, G, h, Aeq, beq


def Quadratic_planning(l_min, l_max, plan_start_l, plan_start_dl, plan_start_ddl, dp_sampling_res=2,
                       w_cost_l=1000, w_cost_dl=10000, w_cost_ddl=3000, w_cost_dddl=150, w_cost_centre=250,
                       w_cost_end_l=40, w_cost_end_dl=40, w_cost_end_ddl=40, host_d1=3, host_d2=3, host_w=3,
                       _return_status=False):
    """path_planning.py:78-219 with ``cvxopt.solvers.qp`` replaced by oracle.qp_dense (unpinned)."""
    H, f, G, h, Aeq, beq = path_qp_matrices(l_min, l_max, plan_start_l, plan_start_dl, plan_start_ddl,
                                             dp_sampling_res, w_cost_l, w_cost_dl, w_cost_ddl, w_cost_dddl,
                                             w_cost_centre, w_cost_end_l, w_cost_end_dl, w_cost_end_ddl,
                                             host_d1, host_d2, host_w)
    res = qp_dense.solve_qp(H, f, G, h, Aeq, beq)
    x = [float(v) for v in res.x]
    out = (x[0::3], x[1::3], x[2::3])
    return out + (res.status,) if _return_status else out


# --------------------------------------------------------------------------------------
# heading / curvature, reference-line smoothing  (planner/planning_utils.py:185-361)
# --------------------------------------------------------------------------------------
def cal_heading_kappa(frenet_path_xy_list):
    """planning_utils.py:185-228 - midpoint-averaged forward differences, sin() on d_theta."""
    pts = frenet_path_xy_list
    dx_ = [pts[i + 1][0] - pts[i][0] for i in range(len(pts) - 1)]
    dy_ = [pts[i + 1][1] - pts[i][1] for i in range(len(pts) - 1)]
    dx = (np.array([dx_[0]] + dx_) + np.array(dx_ + [dx_[-1]])) / 2
    dy = (np.array([dy_[0]] + dy_) + np.array(dy_ + [dy_[-1]])) / 2
    theta = np.arctan2(dy, dx)
    dth = np.diff(theta)
    dth_pre = np.insert(dth, 0, dth[0])
    dth_aft = np.insert(dth, -1, dth[-1])                                  # == append of last (:223)
    k = np.sin((dth_pre + dth_aft) / 2) / np.sqrt(dx ** 2 + dy ** 2)
    return list(theta), list(k)


def smooth_qp_matrices(local_frenet_path_xy, w_cost_smooth=0.4, w_cost_length=0.3, w_cost_ref=0.3,
                       x_thre=0.2, y_thre=0.2):
    """Dense (H, f, G, h) as planning_utils.py:300-349 assembles them."""
    n = len(local_frenet_path_xy)
    ref = np.zeros((2 * n, 1))
    lb = np.zeros((2 * n, 1))
    ub = np.zeros((2 * n, 1))
    for i in range(n):
        px, py = local_frenet_path_xy[i][0], local_frenet_path_xy[i][1]
        ref[2 * i], ref[2 * i + 1] = px, py
        lb[2 * i], lb[2 * i + 1] = px - x_thre, py - y_thre
        ub[2 * i], ub[2 * i + 1] = px + x_thre, py + y_thre
    A1 = np.zeros((2 * n - 4, 2 * n))
    for i in range(2 * n - 4):
        A1[i, i], A1[i, i + 2], A1[i, i + 4] = 1, -2, 1
    A2 = np.zeros((2 * n - 2, 2 * n))
    for i in range(2 * n - 2):
        A2[i, i], A2[i, i + 2] = 1, -1
    H = 2 * (w_cost_smooth * np.dot(A1.transpose(), A1) + w_cost_length * np.dot(A2.transpose(), A2)
             + w_cost_ref * np.identity(2 * n))
    f = -2 * w_cost_ref * ref
    G = np.concatenate((np.identity(2 * n), -np.identity(2 * n)))
    h = np.concatenate((ub, -lb))
    return H, f, G, h


def smooth_reference_line(local_frenet_path_xy, w_cost_smooth=0.4, w_cost_length=0.3, w_cost_ref=0.3,
                          x_thre=0.2, y_thre=0.2, _return_status=False):
    """planning_utils.py:262-361 with the cvxopt call replaced by oracle.qp_dense (unpinned)."""
    H, f, G, h = smooth_qp_matrices(local_frenet_path_xy, w_cost_smooth, w_cost_length, w_cost_ref,
                                    x_thre, y_thre)
    res = qp_dense.solve_qp(H, f, G, h)
    x = [float(v) for v in res.x]
    xy = [(x[i], x[i + 1]) for i in range(0, len(x), 2)]
    theta, kappa = cal_heading_kappa(xy)
    out = [xy[i] + (theta[i], kappa[i]) for i in range(len(xy))]
    return (out, res.status) if _return_status else out


# --------------------------------------------------------------------------------------
# Cartesian <-> Frenet  (planner/planning_utils.py:49-182, 231-259, 364-588, 647-668;
#                         planner/path_planning.py:15-75)
# --------------------------------------------------------------------------------------
def _project_on(node, x, y):
    """Tangent-line projection used at planning_utils.py:104-114, :170-180, :414-424."""
    x_m, y_m, theta_m, k_m = node
    d_v = np.array([x - x_m, y - y_m])
    tou_v = np.array([np.cos(theta_m), np.sin(theta_m)])
    ds = np.dot(d_v, tou_v)
    x_r, y_r = np.array([x_m, y_m]) + ds * tou_v
    return (x_r, y_r, theta_m + k_m * ds, k_m)


def match_projection_points(xy_list, frenet_path_node_list):
    """planning_utils.py:364-426 - scan from 0, stop after 50 non-improvements; projection
    of EVERY point uses the match index of point 0 (quirk :413)."""
    k = len(xy_list)
    P = len(frenet_path_node_list)
    match = np.zeros(k, dtype="int32")
    proj = []
    for j in range(k):
        x, y = xy_list[j][0], xy_list[j][1]
        worse = 0
        best = float("inf")
        for i in range(P):
            nx, ny = frenet_path_node_list[i][0], frenet_path_node_list[i][1]
            d = math.sqrt((nx - x) ** 2 + (ny - y) ** 2)
            if d < best:
                best = d
                match[j] = i
                worse = 0
            else:
                worse += 1
                if worse >= 50:
                    break
        proj.append(_project_on(frenet_path_node_list[match[0]], x, y))
    return list(match), proj


def find_match_points(xy_list, frenet_path_node_list, is_first_run, pre_match_index):
    """planning_utils.py:49-182 - windowed search around the previous match (limit 5), direction
    from the sign of the projection on the previous tangent; same [0] quirk (:103, :169)."""
    k = len(xy_list)
    P = len(frenet_path_node_list)
    match = np.zeros(k, dtype="int32")
    proj = []
    for j in range(k):
        x, y = xy_list[j]
        if is_first_run is True:
            order, limit = range(0, P), 50
        else:
            start = pre_match_index
            px, py, pth = frenet_path_node_list[start][0], frenet_path_node_list[start][1], \
                frenet_path_node_list[start][2]
            flag = np.dot(np.array([x - px, y - py]), np.array([np.cos(pth), np.sin(pth)]))
            order = range(start, P) if flag > 0 else range(start, -1, -1)
            limit = 5
        worse = 0
        best = float("inf")
        for i in order:
            nx, ny = frenet_path_node_list[i][0], frenet_path_node_list[i][1]
            d = math.sqrt((nx - x) ** 2 + (ny - y) ** 2)
            if d < best:
                best = d
                match[j] = i
                worse = 0
            else:
                worse += 1
                if worse >= limit:
                    break
        proj.append(_project_on(frenet_path_node_list[match[0]], x, y))
    return list(match), proj


def sampling(match_point_index, frenet_path_node_list, back_length=10, forward_length=50):
    """planning_utils.py:231-259 - arguments are overwritten with 10/40 (:244-245)."""
    back, fwd = 10, 40
    total = back + fwd
    if match_point_index < back:
        back = match_point_index
        fwd = total - back
    if (len(frenet_path_node_list) - match_point_index) - 1 < fwd:
        fwd = len(frenet_path_node_list) - match_point_index - 1
        back = total - fwd
    return (frenet_path_node_list[match_point_index - back:match_point_index]
            + frenet_path_node_list[match_point_index:match_point_index + fwd + 1])


def cal_projection_s_fun(local_path_opt, match_index_list, xy_list, s_map):
    """planning_utils.py:429-445 - each point's OWN match index."""
    out = []
    for i in range(len(match_index_list)):
        x, y, theta, _ = local_path_opt[match_index_list[i]]
        d_v = np.array([xy_list[i][0] - x, xy_list[i][1] - y])
        tou_v = np.array([math.cos(theta), math.sin(theta)])
        out.append(s_map[match_index_list[i]] + np.dot(d_v, tou_v))
    return out


def cal_s_map_fun(local_path_opt, origin_xy):
    """planning_utils.py:448-472."""
    m, _ = match_projection_points([origin_xy], local_path_opt)
    acc = [0]
    for i in range(1, len(local_path_opt)):
        acc.append(math.sqrt((local_path_opt[i][0] - local_path_opt[i - 1][0]) ** 2
                             + (local_path_opt[i][1] - local_path_opt[i - 1][1]) ** 2) + acc[-1])
    s0 = cal_projection_s_fun(local_path_opt, [m[0]], [origin_xy], acc)
    return list(np.array(acc) - s0[0])


def cal_s_l_fun(obs_xy_list, local_path_opt, s_map):
    """planning_utils.py:475-509."""
    match, proj = match_projection_points(obs_xy_list, local_path_opt)
    s_list = cal_projection_s_fun(local_path_opt, match, obs_xy_list, s_map)
    l_list = []
    for i in range(len(obs_xy_list)):
        px, py, theta, _ = proj[i]
        n_r = np.array([-math.sin(theta), math.cos(theta)])
        r_h = np.array([obs_xy_list[i][0], obs_xy_list[i][1]])
        l_list.append(np.dot(r_h - np.array([px, py]), n_r))
    return s_list, l_list


def cal_s_l_deri_fun(xy_list, V_xy_list, a_xy_list, local_path_xy_opt, origin_xy):
    """planning_utils.py:512-588 - the position used is origin_xy, not xy_list[i] (:542)."""
    _, proj = match_projection_points(xy_list, local_path_xy_opt)
    L, DL, DS, DDL, LDS, DDS, LDDS = [], [], [], [], [], [], []
    for i in range(len(xy_list)):
        x, y, theta, kappa = proj[i]
        nor = np.array([-math.sin(theta), math.cos(theta)])
        tou = np.array([math.cos(theta), math.sin(theta)])
        l = np.dot(np.array([origin_xy[0], origin_xy[1]]) - np.array([x, y]), nor)
        L.append(l)
        V_h = np.array([V_xy_list[i][0], V_xy_list[i][1]])
        dl = np.dot(V_h, nor)
        DL.append(dl)
        ds = np.dot(V_h, tou) / (1 - kappa * L[i])
        DS.append(ds)
        a_h = np.array([a_xy_list[i][0], a_xy_list[i][1]])
        ddl = np.dot(a_h, nor) - kappa * (1 - kappa * l) * (ds ** 2)
        DDL.append(ddl)
        l_ds = 0 if abs(ds) < 1e-6 else DL[i] / ds
        LDS.append(l_ds)
        dds = (np.dot(a_h, tou) + 2 * (ds ** 2 * kappa * l_ds) + ds ** 2 * 0 * l) / (1 - kappa * l)
        DDS.append(dds)
        LDDS.append(0 if abs(ds) < 1e-6 else (ddl - l_ds * dds) / (ds ** 2))
    return L, DL, DS, DDL, LDS, DDS, LDDS


def cal_proj_point(s, pre_match_index, frenet_path_opt, s_map, _flip_ties=None):
    """path_planning.py:52-75 (twin planning_utils.py:647-668) - monotone walk, IndexError past end.

    ``_flip_ties`` (checker only, not in the reference): the walk's comparison ``s_map[idx + 1] < s`` decides which
    node a station is extrapolated from; when the two sides agree to within ``_flip_ties`` (a station that sits on
    a node up to rounding) its outcome is the last bit of cos / sin / dot on the machine at hand, and either answer
    is "the reference's".  With a tolerance given, such a comparison is answered the OTHER way - the reference run
    on a machine that rounds the other way; tests use it to show that a device result beyond 1e-6 of the port is
    the other branch of such a tie and nothing else."""
    idx = pre_match_index
    while (s_map[idx + 1] < s) != (_flip_ties is not None and abs(s_map[idx + 1] - s) <= _flip_ties):
        idx += 1
    mx, my, mth, mk = frenet_path_opt[idx]
    ds = s - s_map[idx]
    px, py = np.array([mx, my]) + ds * np.array([math.cos(mth), math.sin(mth)])
    return (px, py, mth + mk * ds, mk, idx)


cal_proj_point_1 = cal_proj_point


def frenet_path_to_xy(plan_start_s, plan_start_l, enriched_s_list, enriched_l_list, frenet_path_opt, s_map,
                      _flip_ties=None):
    """The un-smoothed target_xy of path_planning.py:29-46 (first tuple 2-long, rest 4-long)."""
    target = []
    px, py, pth, _, idx = cal_proj_point(plan_start_s, 0, frenet_path_opt, s_map, _flip_ties)
    nor = np.array([-math.sin(pth), math.cos(pth)])
    cx, cy = np.array([px, py]) + plan_start_l * nor
    target.append((cx, cy))
    for i in range(len(enriched_l_list)):
        if enriched_s_list[i] > s_map[-1]:                                 # truncation :40
            break
        px, py, pth, pk, idx = cal_proj_point(enriched_s_list[i], idx, frenet_path_opt, s_map, _flip_ties)
        nor = np.array([-math.sin(pth), math.cos(pth)])
        cx, cy = np.array([px, py]) + enriched_l_list[i] * nor
        target.append((cx, cy, pth, pk))
    return target


def frenet_2_x_y_theta_kappa(plan_start_s, plan_start_l, enriched_s_list, enriched_l_list, frenet_path_opt,
                             s_map):
    """path_planning.py:15-49."""
    return smooth_reference_line(frenet_path_to_xy(plan_start_s, plan_start_l, enriched_s_list,
                                                   enriched_l_list, frenet_path_opt, s_map))


# --------------------------------------------------------------------------------------
# small utilities of planning_utils.py that sit beside the path (:706-808)
# --------------------------------------------------------------------------------------
def CalcProjPoint(s, frenet_path_x, frenet_path_y, frenet_path_heading, frenet_path_kappa, s_map):
    """planning_utils.py:736-755 - starts at index 1, uses the first s_map[idx] >= s."""
    idx = 1
    while s_map[idx] < s:
        idx += 1
    ds = s - s_map[idx]
    hd = frenet_path_heading[idx]
    p = np.array([frenet_path_x[idx], frenet_path_y[idx]]) + ds * np.array([np.cos(hd), np.sin(hd)])
    return p[0], p[1], hd + ds * frenet_path_kappa[idx], frenet_path_kappa[idx]


def Frenet2Cartesian(s_set, l_set, dl_set, ddl_set, frenet_path_x, frenet_path_y, frenet_path_heading,
                     frenet_path_kappa, index2s):
    """planning_utils.py:706-733 - 600-slot NaN-padded outputs, stops at the first NaN s."""
    xs = np.ones((600, 1)) * np.nan
    ys = np.ones((600, 1)) * np.nan
    hs = np.ones((600, 1)) * np.nan
    ks = np.ones((600, 1)) * np.nan
    for i in range(len(s_set)):
        if np.isnan(s_set[i]):
            break
        px, py, ph, pk = CalcProjPoint(s_set[i], frenet_path_x, frenet_path_y, frenet_path_heading,
                                       frenet_path_kappa, index2s)
        pt = np.array([px, py]) + l_set[i] * np.array([-np.sin(ph), np.cos(ph)])
        xs[i], ys[i] = pt[0], pt[1]
        hs[i] = ph + np.arctan(dl_set[i] / (1 - pk * l_set[i]))
        dth = hs[i] - ph
        ks[i] = ((ddl_set[i] + pk * dl_set[i] * np.tan(dth)) * (np.cos(dth) ** 2) / (1 - pk * l_set[i]) + pk) \
            * np.cos(dth) / (1 - pk * l_set[i])
    return xs, ys, hs, ks


def trajectory_index2s(trajectory_x, trajectory_y):
    """planning_utils.py:758-780 - cumulative chord length until the first NaN x."""
    n = len(trajectory_x)
    out = np.zeros(n)
    s = 0
    for i in range(1, n):
        if np.isnan(trajectory_x[i]):
            break
        s += np.sqrt((trajectory_x[i] - trajectory_x[i - 1]) ** 2 + (trajectory_y[i] - trajectory_y[i - 1]) ** 2)
        out[i] = s
    return out


def cal_dy_obs_deri(l_set, vx_set, vy_set, proj_heading_set, proj_kappa_set):
    """planning_utils.py:783-808 - 128-slot NaN-padded outputs."""
    s_dot = np.ones(128) * np.nan
    l_dot = np.ones(128) * np.nan
    dl = np.ones(128) * np.nan
    for i in range(len(l_set)):
        if np.isnan(l_set[i]):
            break
        v = np.array([vx_set[i], vy_set[i]])
        hd = proj_heading_set[i]
        l_dot[i] = np.dot(v, np.array([-np.sin(hd), np.cos(hd)]))
        s_dot[i] = np.dot(v, np.array([np.cos(hd), np.sin(hd)])) / (1 - proj_kappa_set[i] * l_set[i])
        dl[i] = 0 if abs(s_dot[i]) < 1e-6 else l_dot[i] / s_dot[i]
    return s_dot, l_dot, dl


# --------------------------------------------------------------------------------------
# one planning cycle  (test_9.py:92-221, the reference's motion_planning without the Pipe)
# --------------------------------------------------------------------------------------
def virtual_obstacles(begin_s, start_v, dyn_dis_speed):
    """test_9.py:137-169 - the FIRST dynamic obstacle (distance, speed) becomes three static obstacles on the centre
    line (l = 0) covering the stretch where it and the ego meet, unless they part beyond s = 80 m."""
    if dyn_dis_speed is None:
        return []
    Len_vehicle, Len_obs = 2.910, 3
    Dis, V_obs = dyn_dis_speed
    V_ego = math.sqrt(start_v[0] ** 2 + start_v[1] ** 2)
    delta_v = V_ego - V_obs
    meet_t = (Dis - Len_vehicle / 2 - Len_obs / 2) / delta_v
    delta_t = (Len_vehicle + Len_obs) / delta_v
    leave_t = meet_t + delta_t
    meet_s = begin_s + Dis + V_obs * meet_t - Len_obs / 2
    leave_s = begin_s + Dis + V_obs * leave_t + Len_obs / 2
    delta_s = leave_s - meet_s
    obs_pos = meet_s + delta_s / 2
    if leave_s < 80:
        return [(meet_s - 10, 0), (obs_pos, 0), (leave_s, 0)]
    return []


def plan_cycle(ref_line, origin_xy, start_xy, start_v, start_a, static_obs_xy, dp_kwargs=None,
               obs_length=5, obs_width=5, decimate=2, use_qp=True, midpoint=True, verbose=True, dyn_dis_speed=None,
               _flip_ties=None):
    """test_9.py:113-218 from the smoothed reference line onward; returns a dict of every stage.
    ``_flip_ties``: see cal_proj_point (checker only)."""
    dp_kwargs = dict(dp_kwargs or {})
    ref_line = [tuple(p) for p in ref_line]
    s_map = cal_s_map_fun(ref_line, origin_xy=tuple(origin_xy))                                  # :113
    if len(static_obs_xy):
        obs_s, obs_l = cal_s_l_fun([tuple(p) for p in static_obs_xy], ref_line, s_map)           # :122
    else:
        obs_s, obs_l = [], []
    begin_s, begin_l = cal_s_l_fun([tuple(start_xy)], ref_line, s_map)                           # :134
    for vs, vl in virtual_obstacles(begin_s[0], start_v, dyn_dis_speed):                         # :137-169
        obs_s = list(obs_s) + [vs]
        obs_l = list(obs_l) + [vl]
    l0, _, _, _, dl0, _, ddl0 = cal_s_l_deri_fun([tuple(start_xy)], [tuple(start_v)], [tuple(start_a)],
                                                 ref_line, tuple(start_xy))                     # :172
    dp_s, dp_l, rows, feasible = DP_algorithm(obs_s, obs_l, begin_s[0], l0[0], dl0[0], ddl0[0],
                                              _return_rows=True, _verbose=verbose, **dp_kwargs)  # :180
    out = dict(s_map=s_map, obs_s=obs_s, obs_l=obs_l, begin_s=begin_s[0], begin_l=begin_l[0],
               start_l=l0[0], start_dl=dl0[0], start_ddl=ddl0[0], dp_s=dp_s, dp_l=dp_l, dp_rows=rows,
               dp_feasible=feasible)
    dp_s, dp_l = dp_s[::decimate], dp_l[::decimate]                                              # :187
    if use_qp:
        l_min, l_max = cal_lmin_lmax(dp_s, dp_l, obs_s, obs_l, obs_length, obs_width)            # :189
        ql, qdl, qddl, status = Quadratic_planning(l_min, l_max, l0[0], dl0[0], ddl0[0], _return_status=True)
        out.update(l_min=l_min, l_max=l_max, qp_l=ql, qp_dl=qdl, qp_ddl=qddl, qp_status=status)
    else:
        ql = list(dp_l)
    if midpoint:                                                                                 # :204-210
        path_s = [dp_s[0]] + [(dp_s[i] + dp_s[i - 1]) / 2 for i in range(1, len(ql))] + [dp_s[-1]]
        path_l = [ql[0]] + [(ql[i] + ql[i - 1]) / 2 for i in range(1, len(ql))] + [ql[-1]]
    else:
        path_s, path_l = list(dp_s), list(ql)
    target_xy = frenet_path_to_xy(begin_s[0], begin_l[0], path_s, path_l, ref_line, s_map, _flip_ties)   # :212
    traj, smooth_status = smooth_reference_line(target_xy, _return_status=True)
    out.update(path_s=path_s, path_l=path_l, target_xy=target_xy, trajectory=traj,
               smooth_status=smooth_status)
    return out


def motion_planning_body(request, dp_kwargs=None, verbose=False):
    """One pass of the planning process body, test_9.py:92-220: request tuple in (:95-96), reply tuple out (:220)."""
    static_obs, dynamic_obs, vehicle_loc, pred_loc, vehicle_v, vehicle_a, global_path, match_list = request
    match_list, _ = find_match_points([tuple(pred_loc)], global_path, False, match_list[0])      # :99-102
    local = sampling(match_list[0], global_path)                                                  # :104
    line = smooth_reference_line(local)                                                           # :110
    static_xy = [(x, y) for x, y, _ in static_obs] if len(static_obs) != 0 and static_obs[0][-1] <= 30 else []   # :116-124
    dyn = (dynamic_obs[0][2], dynamic_obs[0][3]) if len(dynamic_obs) != 0 else None               # :126-131, :141-142
    out = plan_cycle(line, vehicle_loc, pred_loc, vehicle_v, vehicle_a, static_xy, dp_kwargs=dp_kwargs, verbose=verbose,
                     dyn_dis_speed=dyn)
    return out["trajectory"], match_list, out["path_s"], out["path_l"], out
