"""Drop-in for the S-T speed-DP part of reference planner/speed_planning_test.py (:23-305): same function
names, argument order, keyword names and defaults; results from the HIP kernels (batch of one scene).
Line numbers cite the reference file.  The back end (:308-620: ``generate_convex_space``, ``speed_QP``,
``increase_points``, ``path_speed_merge``) is at the bottom of this module.

``speed_DP`` in the reference cannot return: its backtrack indexes ``s_list`` with a float read from
``dp_st_node`` and raises ``IndexError`` (:184) whenever the terminal column is not 0, and its two outputs
are one aliased array (:156).  Here the backtrack uses integer predecessors and returns separate arrays;
pass ``reference_behaviour=True`` to get the reference's observable behaviour instead.
"""
from __future__ import annotations

import numpy as np

from ..api import STB_INDEX, STB_NO_PROFILE, STB_QP_FAILED, STB_RANGE, speed_dp_params, speed_qp_params, st_grid
from ._runtime import planner

_MAX_OBS = 64


def _sets(*arrays):
    k = len(arrays[0])
    if k > _MAX_OBS:
        raise ValueError(f"at most {_MAX_OBS} obstacle slots are supported")
    out = []
    for a in arrays:
        row = np.full((1, max(k, 1)), np.nan)
        row[0, :k] = [float(v) for v in a]
        out.append(row)
    return out, k


def calc_speed_planning_start_condition(plan_start_vx, plan_start_vy, plan_start_ax, plan_start_ay, plan_start_heading):
    """ref :23-35 (called by the driver at test_10.py:249) -> plan_start_s_dot, plan_start_s_dot2."""
    one = lambda v: np.array([float(v)])
    s_dot, s_dot2 = planner().speed_start_condition(one(plan_start_vx), one(plan_start_vy), one(plan_start_ax),
                                                    one(plan_start_ay), one(plan_start_heading))
    return np.float64(s_dot[0]), np.float64(s_dot2[0])


def generate_st_graph(dynamic_obs_s_set, dynamic_obs_l_set, dynamic_obs_s_dot_set, dynamic_obs_l_dot_set):
    """ref :38-98."""
    sets, k = _sets(dynamic_obs_s_set, dynamic_obs_l_set, dynamic_obs_s_dot_set, dynamic_obs_l_dot_set)
    outs = planner().st_graph(*sets)
    return tuple(np.array(o[0, :k]) for o in outs)


def CalcSTCoordinate(row, col, s_list, t_list):
    """ref :287-305 - a table lookup (row 0 is the largest s); a float ``row`` raises IndexError as in the reference."""
    return s_list[len(s_list) - row - 1], t_list[col]


def CalcCollisionCost(w_cost_obs, min_dis):
    """ref :274-284."""
    return float(planner().st_collision_cost(w_cost_obs, np.array([float(min_dis)]))[0])


def _edge_costs(s_start, t_start, s_dot_start, s_end, t_end, sets, p):
    edges = np.array([[[float(s_start), float(t_start), float(s_dot_start), float(s_end), float(t_end)]]])
    return planner().st_edge_costs(p, edges, *sets)


def CalcObsCost(s_start, t_start, s_end, t_end, obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set, obs_st_t_out_set,
                w_cost_obs):
    """ref :234-271."""
    sets, _ = _sets(obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set, obs_st_t_out_set)
    _, obs = _edge_costs(s_start, t_start, 0.0, s_end, t_end, sets, speed_dp_params(w_cost_obs=w_cost_obs))
    return float(obs[0, 0])


def CalcDpCost(row_start, col_start, row_end, col_end, obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set,
               obs_st_t_out_set, w_cost_ref_speed, reference_speed, w_cost_accel, w_cost_obs, plan_start_s_dot,
               s_list, t_list, dp_st_s_dot):
    """ref :191-231 - ``row_start == 0`` selects the DP origin (0, 0, plan_start_s_dot)."""
    s_end, t_end = CalcSTCoordinate(row_end, col_end, s_list, t_list)
    if row_start == 0:
        s_start, t_start, s_dot_start = 0.0, 0.0, plan_start_s_dot
    else:
        s_start, t_start = CalcSTCoordinate(row_start, col_start, s_list, t_list)
        s_dot_start = dp_st_s_dot[row_start][col_start]
    sets, _ = _sets(obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set, obs_st_t_out_set)
    p = speed_dp_params(reference_speed, w_cost_ref_speed, w_cost_accel, w_cost_obs)
    total, _ = _edge_costs(s_start, t_start, s_dot_start, s_end, t_end, sets, p)
    return float(total[0, 0])


def speed_DP(obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set, obs_st_t_out_set, plan_start_s_dot,
             reference_speed=50, w_cost_ref_speed=4000, w_cost_accel=100, w_cost_obs=10000000,
             reference_behaviour=False):
    """ref :101-188.  Returns (dp_speed_s, dp_speed_t), NaN after the terminal column."""
    sets, _ = _sets(obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set, obs_st_t_out_set)
    p = speed_dp_params(reference_speed, w_cost_ref_speed, w_cost_accel, w_cost_obs)
    res = planner().speed_dp(p, *sets, np.array([float(plan_start_s_dot)]), tables=False)
    if reference_behaviour:
        row, col = (int(v) for v in res.end_node[0])
        if col != 0:                                            # ref :182-184
            raise IndexError("only integers, slices (`:`), ellipsis (`...`), numpy.newaxis (`None`) and integer or "
                             "boolean arrays are valid indices")
        both = np.ones(len(st_grid()[1])) * np.nan              # ref :155-156, :178: one array, t written last
        both[col] = res.speed_t[0, col]
        return both, both
    return np.array(res.speed_s[0]), np.array(res.speed_t[0])


# ---------------------------------------------------------------------------------------------
# back end (ref :308-620)
# ---------------------------------------------------------------------------------------------
def _row(values, width=None, fill=np.nan):
    vals = [float(v) for v in values]
    w = max(len(vals), 1) if width is None else max(width, 1)
    row = np.full((1, w), fill)
    k = min(len(vals), w)
    row[0, :k] = vals[:k]
    return row


def _raise_like_reference(status, what):
    """Turn a status bit back into the exception the reference raises on the same input."""
    if status & STB_RANGE:
        raise ValueError(f"{what}: a value in x_new is outside the interpolation range")
    if status & (STB_INDEX | STB_NO_PROFILE):
        raise IndexError(f"{what}: index out of bounds")
    if status & STB_QP_FAILED:
        raise ValueError(f"{what}: the speed QP is infeasible or did not converge")


def generate_convex_space(dp_speed_s, dp_speed_t, path_index2s, obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set,
                          obs_st_t_out_set, trajectory_kappa_init, max_lateral_accel=0.2 * 9.8):
    """ref :308-407 -> s_lb, s_ub, s_dot_lb, s_dot_ub (16 each).  ``path_index2s`` must ascend (the reference sorts
    it inside scipy's interp1d; an ascending array is what ``trajectory_index2s`` produces)."""
    if len(dp_speed_s) != 16 or len(dp_speed_t) != 16:
        raise IndexError("dp_speed_s / dp_speed_t must hold 16 columns")           # the reference indexes [i] for i < 16
    sets, _ = _sets(obs_st_s_in_set, obs_st_s_out_set, obs_st_t_in_set, obs_st_t_out_set)
    n = len(path_index2s)
    out = planner().speed_convex_space(_row(dp_speed_s), _row(dp_speed_t), _row(path_index2s, fill=0.0),
                                       _row(trajectory_kappa_init, width=max(n, 1), fill=0.0), np.array([n], np.int32),
                                       *sets, max_lateral_accel=max_lateral_accel)
    _raise_like_reference(int(out[4][0]), "generate_convex_space")
    return tuple(np.array(o[0]) for o in out[:4])


def speed_QP(plan_start_s_dot, plan_start_s_dot2, dp_speed_s, dp_speed_t, s_lb, s_ub, s_dot_lb, s_dot_ub,
             w_cost_s_dot2=10, w_cost_v_ref=50, w_cost_jerk=500, reference_speed=50):
    """ref :410-511 -> qp_s_init, qp_s_dot_init, qp_s_dot2_init, relative_time_init (17 each, NaN padded).

    The reference's own call raises TypeError inside cvxopt (it passes the equality matrix untransposed, never passes
    its bounds, and aliases ub to lb); this solves the problem the function states, see include/emplanner.h."""
    p = speed_qp_params(w_cost_s_dot2, w_cost_v_ref, w_cost_jerk, reference_speed)
    out = planner().speed_qp(p, np.array([float(plan_start_s_dot)]), np.array([float(plan_start_s_dot2)]), _row(dp_speed_s),
                             _row(dp_speed_t), _row(s_lb), _row(s_ub), _row(s_dot_lb), _row(s_dot_ub))
    _raise_like_reference(int(out[5][0]), "speed_QP")
    return tuple(np.array(o[0]) for o in out[:4])


def increase_points(s_init, s_dot_init, s_dot2_init, relative_time_init):
    """ref :514-566 -> s, s_dot, s_dot2, relative_time (401 each)."""
    out = planner().speed_increase_points(_row(s_init, 17), _row(s_dot_init, 17), _row(s_dot2_init, 17),
                                          _row(relative_time_init, 17))
    _raise_like_reference(int(out[4][0]), "increase_points")
    return tuple(np.array(o[0]) for o in out[:4])


def path_speed_merge(s, s_dot, s_dot2, relative_time, current_time, path_s, trajectory_x_init, trajectory_y_init,
                     trajectory_heading_init, trajectory_kappa_init):
    """ref :569-620 -> trajectory_x, _y, _heading, _kappa, _speed, _accel, _time (401 each)."""
    n = len(trajectory_x_init)
    out, st = planner().path_speed_merge(_row(s, 401), _row(s_dot, 401), _row(s_dot2, 401), _row(relative_time, 401),
                                         np.array([float(current_time)]), _row(path_s, width=n, fill=0.0),
                                         _row(trajectory_x_init), _row(trajectory_y_init), _row(trajectory_heading_init),
                                         _row(trajectory_kappa_init), np.array([n], np.int32))
    _raise_like_reference(int(st[0]), "path_speed_merge")
    return tuple(np.array(out[0, c]) for c in range(7))
