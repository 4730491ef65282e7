"""CPU oracle for the lateral LQR controller (second controller of the same module).

TEST INFRASTRUCTURE ONLY (same import rule as ``oracle/ref_port.py``).

Restates reference ``controller/controller.py`` class ``Lateral_LQR_controller`` (:374-611) as pure functions of explicit
inputs (the reference reads its state from a live ``carla.Vehicle`` in ``cal_vehicle_info``, :405-422):

    cal_A_B_fun            :424-455   continuous error dynamics; Vx + 0.0001 guards the division
    LQR_fun(Q, R)          :457-486   bilinear discretisation (ts = 0.1), Riccati iteration until max|dP| < 0.1
                                       (at most 5000 sweeps), gain K
    cal_error_k_fun(0.1)   :488-567   one-step prediction, nearest point over the WHOLE path, Frenet errors
    forward_control_fun    :569-583   curvature feed-forward (converted "to radians" although already in radians)
    _control               :585-611   Q = diag(200, 1, 50, 1), R = 1:  u = -K e_rr + delta_f

Fully pinned: no third-party solver on this path.  Golden vectors come from the imported reference class
(tests/golden/make_golden_mpc.py).
"""
from __future__ import annotations

import math

import numpy as np

Q_DIAG = (200.0, 1.0, 50.0, 1.0)        # :593-597
R_WEIGHT = 1.0                          # :592, :598
MAX_ITR, EPS, TS = 5000, 0.1, 0.1       # :467-469


def continuous_model(vehicle_para, Vx):
    """:424-455."""
    Vx = Vx + 0.0001
    a, b, Cf, Cr, m, Iz = vehicle_para
    A = np.zeros((4, 4))
    B = np.zeros((4, 1))
    A[0][1] = 1
    A[1][1] = (Cf + Cr) / (m * Vx)
    A[1][2] = -(Cf + Cr) / m
    A[1][3] = (a * Cf - b * Cr) / (m * Vx)
    A[2][3] = 1
    A[3][1] = (a * Cf - b * Cr) / (Iz * Vx)
    A[3][2] = -(a * Cf - b * Cr) / Iz
    A[3][3] = (a * a * Cf + b * b * Cr) / (Iz * Vx)
    B[1][0] = -Cf / m
    B[3][0] = -a * Cf / Iz
    return A, B


def riccati_gain(Ac, Bc, Q=None, R=R_WEIGHT):
    """:457-486.  Returns (K (1 x 4), number of sweeps performed)."""
    Q = np.diag(Q_DIAG) if Q is None else Q
    P = Q
    P_pre = Q
    temp = np.linalg.inv(np.eye(4) - (TS * Ac) / 2)
    A = temp @ (np.eye(4) + (TS * Ac) / 2)
    B = temp @ Bc * TS
    AT, BT = A.T, B.T
    sweeps = 0
    for i in range(MAX_ITR):
        P = AT @ P @ A - (AT @ P @ B) @ np.linalg.inv(R + BT @ P @ B) @ (BT @ P @ A) + Q
        sweeps = i + 1
        if abs(P - P_pre).max() < EPS:
            break
        P_pre = P
    K = np.linalg.inv(BT @ P @ B + R) @ (BT @ P @ A)
    return K, sweeps


def tracking_error(target_path, state, Vx, min_index, ts=0.1):
    """:488-567 - as the MPC's, but the nearest point is searched over the whole path (:518)."""
    x, y, fi, Vy, fi_dot = state
    x = x + Vx * ts * math.cos(fi) - Vy * ts * math.sin(fi)
    y = y + Vy * ts * math.cos(fi) + Vx * ts * math.sin(fi)
    fi = fi + fi_dot * ts
    min_d = 10000
    idx = min_index
    for i in range(0, len(target_path)):
        d = (target_path[i][0] - x) ** 2 + (target_path[i][1] - y) ** 2
        if d < min_d:
            min_d = d
            idx = i
    px, py, pth, pk = target_path[idx][0], target_path[idx][1], target_path[idx][2], target_path[idx][3]
    tor = np.array([math.cos(pth), math.sin(pth)])
    nor = np.array([-math.sin(pth), math.cos(pth)])
    d_v = np.array([x - px, y - py])
    e_d = np.dot(nor, d_v)
    e_s = np.dot(tor, d_v)
    x_pro, y_pro = np.array([px, py]) + e_s * tor
    theta_r = pth + pk * e_s
    e_d_dot = Vy * math.cos(fi - theta_r) + Vx * math.sin(fi - theta_r)
    e_fi = math.sin(fi - theta_r)
    S_dot = (Vx * math.cos(fi - theta_r) - Vy * math.sin(fi - theta_r)) / (1 - pk * e_d)
    e_fi_dot = fi_dot - pk * S_dot
    return (e_d, e_d_dot, e_fi, e_fi_dot), pk, idx, (x, y), (float(x_pro), float(y_pro))


def feed_forward(vehicle_para, K, k_r, Vx):
    """:569-583."""
    a, b, Cf, Cr, m, Iz = vehicle_para
    K_3 = K[0][2]
    delta_f = k_r * (a + b - b * K_3 - (b / Cf + a * K_3 / Cr - a / Cr) * (m * Vx * Vx) / (a + b))
    return delta_f * np.pi / 180


def lateral_lqr(target_path, state, Vx, min_index, vehicle_para):
    """:585-611 (`_control`) from explicit inputs."""
    A, B = continuous_model(vehicle_para, Vx)
    K, sweeps = riccati_gain(A, B)
    e_rr, k_r, idx, pre, pro = tracking_error(target_path, state, Vx, min_index)
    delta_f = feed_forward(vehicle_para, K, k_r, Vx)
    u = -np.dot(K, np.array(e_rr)) + delta_f
    return dict(K=K, sweeps=sweeps, e_rr=np.array(e_rr), k_r=k_r, min_index=idx, pre=pre, pro=pro, delta_f=float(delta_f),
                steering=float(u[0]))
