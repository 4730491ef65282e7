"""Drop-in for the S-T speed-DP part of reference planner/speed_planning_test.py (:38-305): same function
names, argument order, keyword names and defaults; results from the HIP kernels (batch of one scene).
Line numbers cite the reference file.  The speed QP and the path/speed merge (:308-611) are not part of
the hot path and are not provided.

``speed_DP`` in the reference cannot return: its backtrack indexes ``s_list`` with a float read from
``dp_st_node`` and raises ``IndexError`` (:184) whenever the terminal column is not 0, and its two outputs
are one aliased array (:156).  Here the backtrack uses integer predecessors and returns separate arrays;
pass ``reference_behaviour=True`` to get the reference's observable behaviour instead.
"""
from __future__ import annotations

import numpy as np

from ..api import speed_dp_params, st_grid
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
