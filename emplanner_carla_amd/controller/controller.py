"""Drop-in for reference controller/controller.py class Lateral_MPC_controller (:65-337).  Line numbers cite the
reference file.  ``cal_vehicle_info`` (:90-113) is the only part that talks to CARLA: it stays on the host, duck-typed;
everything `_control` does after it is one call of ``emp_mpc_lateral`` (batch of one vehicle)."""
from __future__ import annotations

import math

import numpy as np

from ..api import mpc_params
from ..planner._runtime import planner


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
        """:90-113 - state from the (duck-typed) vehicle; |Vx| is kept >= 0.005 as in the reference."""
        loc = self._vehicle.get_location()
        x, y = loc.x, loc.y
        fi = self._vehicle.get_transform().rotation.yaw * (math.pi / 180)
        V = self._vehicle.get_velocity()
        V_length = math.sqrt(V.x * V.x + V.y * V.y + V.z * V.z)
        beta = math.atan2(V.y, V.x) - fi
        Vy = V_length * math.sin(beta)
        if V_length * math.cos(beta) < 0:
            Vx = -max(abs(V_length * math.cos(beta)), 0.005)
        else:
            Vx = max(V_length * math.cos(beta), 0.005)
        fi_dao = self._vehicle.get_angular_velocity().z * (math.pi / 180)
        self._vehicle_state = (x, y, fi, Vy, fi_dao)
        self._vehicle_Vx = Vx

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
        """:405-422 - no clamp on Vx here (cal_A_B_fun adds 0.0001 instead, :439)."""
        loc = self._vehicle.get_location()
        x, y = loc.x, loc.y
        fi = self._vehicle.get_transform().rotation.yaw * (math.pi / 180)
        V = self._vehicle.get_velocity()
        V_length = math.sqrt(V.x * V.x + V.y * V.y + V.z * V.z)
        beta = math.atan2(V.y, V.x) - fi
        Vy = V_length * math.sin(beta)
        Vx = V_length * math.cos(beta)
        fi_dao = self._vehicle.get_angular_velocity().z * (math.pi / 180)
        self._vehicle_state = (x, y, fi, Vy, fi_dao)
        self._vehicle_Vx = Vx

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
