"""Drop-in for reference planner/planning_utils.py: Cartesian<->Frenet projection, s-map, quintic fit,
reference-line smoothing, heading/curvature - same names, argument order, keyword names and defaults;
numeric results from the HIP kernels (batch of one scene).  The four CARLA-bound helpers stay thin
duck-typed Python (they only read attributes of carla objects) and this module does NOT import carla,
whose import at reference planning_utils.py:10 is what ties the reference to a simulator install.
Line numbers cite the reference file this module replaces."""
from __future__ import annotations

import math

import numpy as np

from .. import _lib as L
from ..api import smooth_params
from ._runtime import f64, line_array, planner, xy_array
from .vehicle_state import planar_state


# ---- CARLA-bound helpers (host glue only: attribute access and list building) ------------------
def Vector_fun(loc_1, loc_2):
    """ref :14-26 - unit vector loc_1 -> loc_2, rounded to 4 decimals."""
    d = [loc_2.x - loc_1.x, loc_2.y - loc_1.y, loc_2.z - loc_1.z]
    norm = math.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2) + np.finfo(float).eps
    return np.round([d[0] / norm, d[1] / norm, d[2] / norm], 4)


def waypoint_list_2_target_path(pathway):
    """ref :29-46 - [(waypoint, option), ...] -> [(x, y, theta, kappa), ...]."""
    xy = [(w[0].transform.location.x, w[0].transform.location.y) for w in pathway]
    theta, kappa = cal_heading_kappa(xy)
    return [(xy[i][0], xy[i][1], theta[i], kappa[i]) for i in range(len(theta))]


def predict_block(ego_vehicle, ts=0.1):
    """ref :591-614 - where the vehicle is ``ts`` seconds from now at constant body-frame velocity and yaw rate."""
    st = planar_state(ego_vehicle)
    c, s_ = math.cos(st.yaw), math.sin(st.yaw)
    return (st.x + st.v_long * ts * c - st.v_lat * ts * s_, st.y + st.v_lat * ts * c + st.v_long * ts * s_,
            st.yaw + st.yaw_rate * ts)


def predict_block_based_on_frenet(vehicle_loc, vehicle_velocity, local_frenet_path_opt, cur_path_s, cur_path_l, ts=0.1):
    """ref :617-644."""
    speed = math.sqrt(vehicle_velocity.x ** 2 + vehicle_velocity.y ** 2 + vehicle_velocity.z ** 2)
    s_map = cal_s_map_fun(local_frenet_path_opt, origin_xy=(vehicle_loc.x, vehicle_loc.y))
    s = max(cur_path_s[0], 0) + speed * ts
    index = int(np.argmin(abs(np.array(cur_path_s) - s)))
    l = cur_path_l[index]
    px, py, pth, _, _ = cal_proj_point_1(s, 0, local_frenet_path_opt, s_map)
    return px + l * (-math.sin(pth)), py + l * math.cos(pth)


# ---- matching / projection --------------------------------------------------------------------
def find_match_points(xy_list, frenet_path_node_list, is_first_run, pre_match_index):
    """ref :49-182."""
    if len(frenet_path_node_list) == 0:
        raise IndexError("list index out of range")                 # ref :103 / :123 on an empty path
    line, n_ref = line_array(frenet_path_node_list)
    xy, n = xy_array(xy_list)
    mi, pr = planner().find_match_points(line, n_ref, xy, n, np.array([1 if is_first_run is True else 0], np.int32),
                                         np.array([int(pre_match_index)], np.int32))
    if len(mi[0]) and mi[0][0] < 0:
        raise IndexError("list index out of range")                 # ref :123 frenet_path_node_list[pre_match_index]
    return list(mi[0].astype(np.int32)), [tuple(f64(v) for v in p) for p in pr[0]]


def match_projection_points(xy_list, frenet_path_node_list):
    """ref :364-426."""
    if len(frenet_path_node_list) == 0:
        raise IndexError("list index out of range")                 # ref :413 frenet_path_node_list[match_point_index_list[0]]
    line, n_ref = line_array(frenet_path_node_list)
    xy, n = xy_array(xy_list)
    mi, pr = planner().match_projection(line, n_ref, xy, n)
    if len(mi[0]) and mi[0][0] < 0:
        raise IndexError("list index out of range")                 # ref :383 on an empty path
    return list(mi[0].astype(np.int32)), [tuple(f64(v) for v in p) for p in pr[0]]


def cal_heading_kappa(frenet_path_xy_list):
    """ref :185-228."""
    xy, n = xy_array(frenet_path_xy_list)
    if n[0] < 2:
        raise IndexError("list index out of range")                 # ref :212 dx_[0] on an empty list
    th, kp = planner().heading_kappa(xy, n)
    return list(th[0]), list(kp[0])


def sampling(match_point_index, frenet_path_node_list, back_length=10, forward_length=50):
    """ref :231-259 - list slicing only; the arguments are overwritten with 10 / 40 exactly like :244-245."""
    back_length, forward_length = 10, 40
    total = back_length + forward_length
    if match_point_index < back_length:
        back_length = match_point_index
        forward_length = total - back_length
    if (len(frenet_path_node_list) - match_point_index) - 1 < forward_length:
        forward_length = len(frenet_path_node_list) - match_point_index - 1
        back_length = total - forward_length
    return (frenet_path_node_list[match_point_index - back_length:match_point_index]
            + frenet_path_node_list[match_point_index:match_point_index + forward_length + 1])


def smooth_reference_line(local_frenet_path_xy, w_cost_smooth=0.4, w_cost_length=0.3, w_cost_ref=0.3, x_thre=0.2,
                          y_thre=0.2):
    """ref :262-361 - returns [(x, y, theta, kappa), ...] with x, y Python floats like the reference."""
    xy, n = xy_array(local_frenet_path_xy)
    out, it, st = planner().smooth_line(smooth_params(w_cost_smooth, w_cost_length, w_cost_ref, x_thre, y_thre), xy, n)
    if st[0] & L.ST_SMOOTH_FAILED:
        raise ValueError("smooth_reference_line: the smoothing QP did not converge")
    o = out[0]                                                      # x, y Python floats; theta, kappa np.float64 (ref :355-361)
    return list(zip(o[:, 0].tolist(), o[:, 1].tolist(), list(o[:, 2]), list(o[:, 3])))


def cal_projection_s_fun(local_path_opt, match_index_list, xy_list, s_map):
    """ref :429-445."""
    line, n_ref = line_array(local_path_opt)
    k = len(match_index_list)
    xy, n = xy_array(xy_list[:k])
    s, _ = planner().s_l(line, np.asarray(s_map, dtype=np.float64).reshape(1, -1), n_ref, xy, n,
                         match_index=np.asarray(match_index_list, dtype=np.int32).reshape(1, k), want_l=False)
    return list(s[0])


def cal_s_map_fun(local_path_opt, origin_xy):
    """ref :448-472."""
    line, n_ref = line_array(local_path_opt)
    sm = planner().s_map(line, n_ref, np.array([[float(origin_xy[0]), float(origin_xy[1])]]))
    return list(sm[0])


def cal_s_l_fun(obs_xy_list, local_path_opt, s_map):
    """ref :475-509."""
    line, n_ref = line_array(local_path_opt)
    xy, n = xy_array(obs_xy_list)
    s, l = planner().s_l(line, np.asarray(s_map, dtype=np.float64).reshape(1, -1), n_ref, xy, n)
    return list(s[0]), list(l[0])


def cal_s_l_deri_fun(xy_list, V_xy_list, a_xy_list, local_path_xy_opt, origin_xy):
    """ref :512-588 - seven lists: l, dl/dt, ds/dt, d2l/dt2, dl/ds, d2s/dt2, d2l/ds2."""
    line, n_ref = line_array(local_path_xy_opt)
    xy, n = xy_array(xy_list)
    v, _ = xy_array(V_xy_list)
    a, _ = xy_array(a_xy_list)
    o = planner().s_l_deri(line, n_ref, xy, v, a, n, np.array([[float(origin_xy[0]), float(origin_xy[1])]]))
    return tuple(list(o[0, :, c]) for c in range(7))


def cal_proj_point_1(s, pre_match_index, frenet_path_opt, s_map):
    """ref :647-668 (twin of path_planning.cal_proj_point)."""
    from . import path_planning
    return path_planning.cal_proj_point(s, pre_match_index, frenet_path_opt, s_map)


def cal_quintic_coefficient(start_l, start_dl, start_ddl, end_l, end_dl, end_ddl, start_s, end_s):
    """ref :671-703 - coefficients [c0..c5] of l(s) in ABSOLUTE s.  Computed in closed form in the shifted
    coordinate and re-expanded, i.e. without the reference's ill-conditioned 6x6 inverse: the polynomial agrees
    with the reference's on the segment to its own noise ("Reference noise floor")."""
    c = planner().quintic_coefficients(np.array([[start_l, start_dl, start_ddl, end_l, end_dl, end_ddl, start_s,
                                                  end_s]], dtype=np.float64))
    return list(c[0])


# ---- helpers beside the path (used by the reference's speed-planning drivers) ---------------------
def _line_from_columns(fx, fy, fh, fk):
    a = np.stack([np.asarray(fx, dtype=np.float64).reshape(-1), np.asarray(fy, dtype=np.float64).reshape(-1),
                  np.asarray(fh, dtype=np.float64).reshape(-1), np.asarray(fk, dtype=np.float64).reshape(-1)], axis=1)
    return a.reshape(1, -1, 4), np.array([a.shape[0]], np.int32)


def Frenet2Cartesian(s_set, l_set, dl_set, ddl_set, frenet_path_x, frenet_path_y, frenet_path_heading,
                     frenet_path_kappa, index2s):
    """ref :706-733 - four (600, 1) NaN-padded arrays."""
    line, n_ref = _line_from_columns(frenet_path_x, frenet_path_y, frenet_path_heading, frenet_path_kappa)
    k = len(s_set)
    sl = np.stack([np.asarray(v, dtype=np.float64).reshape(-1)[:k] for v in (s_set, l_set, dl_set, ddl_set)], axis=1)
    o, st = planner().frenet2cartesian(line, np.asarray(index2s, dtype=np.float64).reshape(1, -1)[:, :n_ref[0]], n_ref,
                                       sl.reshape(1, k, 4), np.array([k], np.int32))
    if st[0]:
        raise IndexError("index out of bounds")                     # ref :743 walks past index2s
    outs = [np.ones((600, 1)) * np.nan for _ in range(4)]
    for c in range(4):
        outs[c][:k, 0] = o[0, :, c]
    return tuple(outs)


def CalcProjPoint(s, frenet_path_x, frenet_path_y, frenet_path_heading, frenet_path_kappa, s_map):
    """ref :736-755."""
    line, n_ref = _line_from_columns(frenet_path_x, frenet_path_y, frenet_path_heading, frenet_path_kappa)
    sl = np.array([[[float(s), 0.0, 0.0, 0.0]]])
    o, st = planner().frenet2cartesian(line, np.asarray(s_map, dtype=np.float64).reshape(1, -1)[:, :n_ref[0]], n_ref, sl,
                                       np.array([1], np.int32), proj_only=True)
    if st[0]:
        raise IndexError("index out of bounds")
    return f64(o[0, 0, 0]), f64(o[0, 0, 1]), f64(o[0, 0, 2]), f64(o[0, 0, 3])


def trajectory_index2s(trajectory_x, trajectory_y):
    """ref :758-780."""
    x = np.asarray(trajectory_x, dtype=np.float64).reshape(1, -1)
    y = np.asarray(trajectory_y, dtype=np.float64).reshape(1, -1)
    return planner().trajectory_index2s(x, y, np.array([x.shape[1]], np.int32))[0].copy()


def cal_dy_obs_deri(l_set, vx_set, vy_set, proj_heading_set, proj_kappa_set):
    """ref :783-808 - three 128-slot NaN-padded arrays, stopping at the first NaN l."""
    n = 128
    outs = [np.ones(n) * np.nan for _ in range(3)]
    l = np.asarray(l_set, dtype=np.float64).reshape(-1)
    k = 0
    while k < len(l) and not np.isnan(l[k]):
        k += 1
    if k:
        rows = np.stack([l[:k]] + [np.asarray(v, dtype=np.float64).reshape(-1)[:k]
                                   for v in (vx_set, vy_set, proj_heading_set, proj_kappa_set)], axis=1)
        o = planner().dy_obs_deri(rows)
        for c in range(3):
            outs[c][:k] = o[:, c]
    return tuple(outs)
