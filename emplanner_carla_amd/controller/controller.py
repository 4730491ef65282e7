"""Drop-in for reference controller/controller.py class Lateral_MPC_controller (:65-337).  Line numbers cite the
reference file.  ``cal_vehicle_info`` (:90-113) is the only part that talks to CARLA: it stays on the host, duck-typed;
everything `_control` does after it is one call of ``emp_mpc_lateral`` (batch of one vehicle)."""
from __future__ import annotations

import math

import numpy as np

from ..api import mpc_params
from ..planner._runtime import planner
from ..planner.vehicle_state import planar_state


class Lateral_MPC_controller(object):
    def __init__(self, ego_vehicle, vehicle_para, pathway_xy_theta_kappa):
        self._vehicle_state = None
        self._vehicle_para = vehicle_para
        self._vehicle = ego_vehicle
        self._vehicle_Vx = 0
        self._target_path = pathway_xy_theta_kappa
        self._N, self._P, self._n = 6, 2, 4                  # :72-74
        self.k_r = None
        self.e_rr = None
        self.min_index = 0
        self.x_pre = self.y_pre = self.x_pro = self.y_pro = 0

    def cal_vehicle_info(self):
        """:90-113 - the longitudinal speed keeps its sign but never drops below 0.005 in magnitude (the model divides by it)."""
        st = planar_state(self._vehicle)
        self._vehicle_state = (st.x, st.y, st.yaw, st.v_lat, st.yaw_rate)
        self._vehicle_Vx = math.copysign(max(abs(st.v_long), 0.005), st.v_long) if st.v_long != 0 else 0.005

    def _control(self):
        """:313-337 - returns the first control of the horizon (the raw steering command)."""
        self.cal_vehicle_info()
        path = np.asarray([[float(p[0]), float(p[1]), float(p[2]), float(p[3])] for p in self._target_path], dtype=np.float64)
        if not 0 <= self.min_index < len(path):
            raise IndexError("list index out of range")         # what the reference's self._target_path[min_index] does
        res = planner().mpc_lateral(mpc_params(vehicle_para=self._vehicle_para), path[None], np.array([len(path)], np.int32),
                                    np.array([self._vehicle_state], dtype=np.float64), np.array([self._vehicle_Vx]),
                                    np.array([self.min_index], np.int32))
        if int(res.status[0]) != 0:
            raise ValueError("lateral MPC: the box QP did not converge")
        self.min_index = int(res.min_index[0])
        self.e_rr = tuple(float(v) for v in res.e_rr[0])
        self.k_r = float(res.k_r[0])
        self.x_pre, self.y_pre, self.x_pro, self.y_pro = (float(v) for v in res.pre_pro[0])
        return float(res.steer[0])


class Lateral_LQR_controller(object):
    """Drop-in for reference class Lateral_LQR_controller (:374-611): same constructor and ``_control()``."""

    def __init__(self, ego_vehicle, vehicle_para, pathway_xy_theta_kappa):
        self._vehicle_para = vehicle_para
        self._vehicle = ego_vehicle
        self._vehicle_state = None
        self._vehicle_Vx = 0
        self._target_path = pathway_xy_theta_kappa
        self.K = None
        self.k_r = None
        self.e_rr = None
        self.delta_f = None
        self.min_index = 0
        self.x_pre = self.y_pre = self.x_pro = self.y_pro = 0

    def cal_vehicle_info(self):
        """:405-422 - no clamp on the longitudinal speed here (cal_A_B_fun adds 0.0001 instead, :439)."""
        st = planar_state(self._vehicle)
        self._vehicle_state = (st.x, st.y, st.yaw, st.v_lat, st.yaw_rate)
        self._vehicle_Vx = st.v_long

    def _control(self):
        """:585-611 - the raw steering command -K e_rr + delta_f."""
        from ..api import lqr_params
        self.cal_vehicle_info()
        path = np.asarray([[float(p[0]), float(p[1]), float(p[2]), float(p[3])] for p in self._target_path],
                          dtype=np.float64).reshape(-1, 4)
        if len(path) == 0:
            raise IndexError("list index out of range")
        res = planner().lqr_lateral(lqr_params(vehicle_para=self._vehicle_para), path[None], np.array([len(path)], np.int32),
                                    np.array([self._vehicle_state], dtype=np.float64), np.array([self._vehicle_Vx]),
                                    np.array([min(max(self.min_index, 0), len(path) - 1)], np.int32))
        if int(res.status[0]) != 0:
            raise IndexError("list index out of range")
        self.min_index = int(res.min_index[0])
        self.K = np.asarray(res.K[0]).reshape(1, 4)
        self.e_rr = tuple(float(v) for v in res.e_rr[0])
        self.k_r = float(res.k_r[0])
        self.x_pre, self.y_pre, self.x_pro, self.y_pro = (float(v) for v in res.pre_pro[0])
        return float(res.steer[0])
