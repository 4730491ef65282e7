"""Drop-in for reference planner/path_planning.py: S-L lattice DP, densification, QP bounds, path QP,
Frenet->Cartesian - same names, argument order, keyword names and defaults; results from the HIP
kernels (batch of one scene).  Line numbers cite the reference file this module replaces."""
from __future__ import annotations

import numpy as np

from .. import _lib as L
from ..api import dp_params, max_path_points, qp_params, smooth_params
from . import planning_utils
from ._runtime import f64, line_array, planner

INFEASIBLE_BANNER = "********************     can't find a feasible path      ********************"


def _sl_arrays(obs_s_list, obs_l_list):
    k = len(obs_s_list)
    obs_s = np.zeros((1, max(k, 1)))
    obs_l = np.zeros((1, max(k, 1)))
    obs_s[0, :k] = [float(v) for v in obs_s_list]
    obs_l[0, :k] = [float(v) for v in obs_l_list]
    return obs_s, obs_l, np.array([k], np.int32)


def cal_obs_cost(w_cost_collision, square_d, danger_dis=4, safe_dis=6):
    """ref :588-609 (any number of samples: the reference loops over what it is handed, :601)."""
    sq = np.asarray(square_d, dtype=np.float64).reshape(1, -1)
    return float(planner().obs_cost(sq, w_cost_collision, danger_dis, safe_dis)[0])


def _edge(obs_s_list, obs_l_list, start, end_s, end_l, sample_s, w_cost_collision, w_cost_smooth, w_cost_ref):
    """One free edge through emp_free_edge_costs: the quintic from the start state (s, l, dl, ddl) to (end_l, 0, 0) at end_s,
    sampled every sample_s / 10 from the start (ref :475 / :553 and :492-493 / :565-566)."""
    obs_s, obs_l, n = _sl_arrays(obs_s_list, obs_l_list)
    # on the lattice end_s IS start_s + sample_s: the span is then sample_s itself, to the last bit, as in the DP kernels
    span = float(sample_s) if float(end_s) == start[0] + float(sample_s) else float(end_s) - start[0]
    edge = np.array([[start[0], start[1], start[2], start[3], span, float(end_l), float(sample_s), 0.0]], dtype=np.float64)
    c = planner().free_edge_costs(edge, obs_s, obs_l, n, w_cost_collision, w_cost_smooth, w_cost_ref)
    return np.array([[c[0]]])                                       # the reference returns a 1x1 array


def cal_start_cost(obs_s_list, obs_l_list, begin_s, begin_l, begin_dl, begin_ddl, cur_node_row, row, sample_s,
                   sample_l, w_cost_collision, w_cost_smooth, w_cost_ref):
    """ref :435-514."""
    p = dp_params(row=row, col=1, sample_s=sample_s, sample_l=sample_l, w_collision_cost=w_cost_collision,
                  w_smooth_cost=w_cost_smooth, w_reference_cost=w_cost_ref)
    obs_s, obs_l, n = _sl_arrays(obs_s_list, obs_l_list)
    c0, _ = planner().dp_edge_costs(p, obs_s, obs_l, n,
                                    np.array([[begin_s, begin_l, begin_dl, begin_ddl]], dtype=np.float64))
    return np.array([[c0[0, int(cur_node_row)]]])


def cal_neighbor_cost(obs_s_list, obs_l_list, pre_node_s, pre_node_l, cur_node_s, cur_node_l, sample_s,
                      w_cost_collision, w_cost_smooth, w_cost_ref):
    """ref :517-585: the quintic ends at cur_node_s (:553), the samples step by sample_s / 10 from pre_node_s (:565-566)."""
    return _edge(obs_s_list, obs_l_list, (float(pre_node_s), float(pre_node_l), 0.0, 0.0), float(cur_node_s), float(cur_node_l),
                 sample_s, w_cost_collision, w_cost_smooth, w_cost_ref)


def enrich_DP_s_l(DP_s_list, DP_l_list, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl, resolution=1):
    """ref :378-432."""
    col = len(DP_s_list)
    node_s = np.asarray(DP_s_list, dtype=np.float64).reshape(1, col)
    node_l = np.asarray(DP_l_list, dtype=np.float64).reshape(1, col)
    start = np.array([[plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl]], dtype=np.float64)
    spans = np.diff(np.concatenate([[float(plan_start_s)], node_s[0]]))
    cap = int(sum(int(np.ceil(max(int(v), 0) / resolution)) for v in spans)) + 1
    s, l, n, st = planner().enrich_nodes(node_s, node_l, np.array([col], np.int32), start, resolution, max(cap, 1))
    n = int(n[0])
    return list(s[0, :n]), list(l[0, :n])


def DP_algorithm(obs_s_list, obs_l_list, plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl, sampling_res=2,
                 w_collision_cost=1e12, w_smooth_cost=[300, 1000, 5000], w_reference_cost=20, row=12, col=6,
                 sample_s=15, sample_l=1.5):
    """ref :276-375."""
    p = dp_params(row, col, sample_s, sample_l, sampling_res, w_collision_cost, w_smooth_cost, w_reference_cost)
    obs_s, obs_l, n_obs = _sl_arrays(obs_s_list, obs_l_list)
    start = np.array([[plan_start_s, plan_start_l, plan_start_dl, plan_start_ddl]], dtype=np.float64)
    pl = planner()
    rows, min_cost, status = pl.dp_plan(p, obs_s, obs_l, n_obs, start)
    if status[0] & L.ST_DP_INFEASIBLE:
        print(INFEASIBLE_BANNER)                                   # ref :351-352 prints and carries on
    s, l, n, st = pl.dp_enrich(p, rows, start, max_path_points(p))
    n = int(n[0])
    return list(s[0, :n]), list(l[0, :n])


def cal_lmin_lmax(dp_path_s, dp_path_l, obs_s_list, obs_l_list, obs_length, obs_width):
    """ref :222-273 (returns two ndarrays; IndexError where the reference indexes past the path)."""
    n = len(dp_path_s)
    obs_s, obs_l, n_obs = _sl_arrays(obs_s_list, obs_l_list)
    lo, hi, st = planner().lmin_lmax(np.asarray(dp_path_s, dtype=np.float64).reshape(1, n),
                                     np.asarray(dp_path_l, dtype=np.float64).reshape(1, n),
                                     np.array([n], np.int32), obs_s, obs_l, n_obs, obs_length, obs_width)
    if st[0] & L.ST_BOUND_INDEX:
        raise IndexError(f"index out of bounds for axis 0 with size {n}")      # ref :267 / :272
    return lo[0].copy(), hi[0].copy()


def Quadratic_planning(l_min, l_max, plan_start_l, plan_start_dl, plan_start_ddl, dp_sampling_res=2, w_cost_l=1000,
                       w_cost_dl=10000, w_cost_ddl=3000, w_cost_dddl=150, w_cost_centre=250, w_cost_end_l=40,
                       w_cost_end_dl=40, w_cost_end_ddl=40, host_d1=3, host_d2=3, host_w=3):
    """ref :78-219.  Raises ValueError when the QP is infeasible (the reference hands back whatever cvxopt's
    last iterate was without looking at its status, :211-218)."""
    n = len(l_min)
    q = qp_params(dp_sampling_res=float(dp_sampling_res), w_cost_l=w_cost_l, w_cost_dl=w_cost_dl,
                  w_cost_ddl=w_cost_ddl, w_cost_dddl=w_cost_dddl, w_cost_centre=w_cost_centre,
                  w_cost_end_l=w_cost_end_l, w_cost_end_dl=w_cost_end_dl, w_cost_end_ddl=w_cost_end_ddl,
                  host_d1=host_d1, host_d2=host_d2, host_w=host_w)
    l, dl, ddl, it, st = planner().path_qp(q, np.asarray(l_min, dtype=np.float64).reshape(1, n),
                                           np.asarray(l_max, dtype=np.float64).reshape(1, n), np.array([n], np.int32),
                                           np.array([[plan_start_l, plan_start_dl, plan_start_ddl]], dtype=np.float64))
    if st[0] & L.ST_QP_FAILED:
        raise ValueError("Quadratic_planning: the QP is infeasible for these bounds / start state")
    return [float(v) for v in l[0]], [float(v) for v in dl[0]], [float(v) for v in ddl[0]]


def cal_proj_point(s, pre_match_index, frenet_path_opt, s_map):
    """ref :52-75."""
    line, n_ref = line_array(frenet_path_opt)
    sm = np.asarray(s_map, dtype=np.float64).reshape(1, -1)
    out, idx, st = planner().proj_point(line, sm, n_ref, np.array([float(s)]), np.array([int(pre_match_index)], np.int32))
    if st[0]:
        raise IndexError("list index out of range")                 # ref :63
    return (f64(out[0, 0]), f64(out[0, 1]), f64(out[0, 2]), f64(out[0, 3]), int(idx[0]))


def frenet_2_x_y_theta_kappa(plan_start_s, plan_start_l, enriched_s_list, enriched_l_list, frenet_path_opt, s_map):
    """ref :15-49: Frenet->Cartesian of the path, then smooth_reference_line on the result."""
    line, n_ref = line_array(frenet_path_opt)
    sm = np.asarray(s_map, dtype=np.float64).reshape(1, -1)
    n = len(enriched_l_list)
    pl = planner()
    tgt, cnt, st = pl.frenet_path_to_xy(line, sm, n_ref, np.array([[plan_start_s, plan_start_l]], dtype=np.float64),
                                        np.asarray(enriched_s_list[:n], dtype=np.float64).reshape(1, n),
                                        np.asarray(enriched_l_list, dtype=np.float64).reshape(1, n),
                                        np.array([n], np.int32))
    if st[0] & L.ST_S_OUT_OF_RANGE:
        raise IndexError("list index out of range")                 # ref :63 via :31 / :42
    m = int(cnt[0])
    return planning_utils.smooth_reference_line([tuple(p) for p in tgt[0, :m]])
