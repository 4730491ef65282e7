"""CPU oracle for the lateral MPC controller (the controller input side).

TEST INFRASTRUCTURE ONLY (same import rule as ``oracle/ref_port.py``).

Restates the arithmetic of reference ``controller/controller.py`` class ``Lateral_MPC_controller``
(:65-337) as pure functions of explicit inputs - the reference reads its state from a live
``carla.Vehicle`` (``cal_vehicle_info`` :90-113), which is the caller's business here:

    cal_A_B_C_fun                        :115-148   continuous error dynamics from (a, b, Cf, Cr, m, Iz), Vx
    cal_error_k_fun(ts=0.1)              :170-251   one-step prediction, windowed nearest point, Frenet errors
    cal_coefficient_of_discretion_fun    :151-168   bilinear discretisation, ts = 0.1
    cal_control_para_fun(Q, R, F)        :253-311   condensed MPC: N = 6 steps, P = 2 controls per step, box +-1
    _control                             :313-337   Q = diag(250, 1, 50, 1), F = I, R = 1 -> first control

Pinning: golden vectors of the imported reference class (tests/golden/make_golden_mpc.py drives the methods
with ``_vehicle_state`` / ``_vehicle_Vx`` set by hand and a stub ``carla``) pin everything up to the dense
(H, f) of the QP.  The QP itself goes to ``cvxopt.solvers.qp`` in the reference (:309): **parity unpinned**
there, as for the path QP - the minimiser of the strictly convex box QP is unique and is certified through
``oracle/qp_dense.py``.
"""
from __future__ import annotations

import math

import numpy as np

from . import qp_dense

N_STEPS, N_CTRL, N_STATE = 6, 2, 4      # :72-74
TS_DISCRETE = 0.1                       # :159
TS_PREDICT = 0.1                        # _control calls cal_error_k_fun(ts=0.1), :333
WINDOW = 50                             # :204
Q_DIAG = (250.0, 1.0, 50.0, 1.0)        # :322-326
F_DIAG = (1.0, 1.0, 1.0, 1.0)           # :327
R_WEIGHT = 1.0                          # :321, :328


def continuous_model(vehicle_para, Vx):
    """:115-148."""
    a, b, Cf, Cr, m, Iz = vehicle_para
    A = np.zeros((4, 4))
    B = np.zeros((4, 1))
    C = np.zeros((4, 1))
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
    C[1][0] = (a * Cf + b * Cr) / (m * Vx) - Vx
    C[3][0] = (a ** 2 * Cf + b ** 2 * Cr) / (Iz * Vx)
    return A, B, C


def tracking_error(target_path, state, Vx, min_index, ts=TS_PREDICT):
    """:170-251.  state = (x, y, fi, Vy, fi_dot).  Returns (e_rr, k_r, min_index, (x_pre, y_pre), (x_pro, y_pro))."""
    x, y, fi, Vy, fi_dot = state
    x = x + Vx * ts * math.cos(fi) - Vy * ts * math.sin(fi)
    y = y + Vy * ts * math.cos(fi) + Vx * ts * math.sin(fi)
    fi = fi + fi_dot * ts
    x_pre, y_pre = x, y
    n = len(target_path)
    min_d = 10000
    idx = min_index
    for i in range(min_index, min(min_index + WINDOW, n)):       # squared distance, strict '<', window from the
        d = (target_path[i][0] - x) ** 2 + (target_path[i][1] - y) ** 2     # previous match (which it keeps if nothing
        if d < min_d:                                                       # is closer than 100 m)
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
    return (e_d, e_d_dot, e_fi, e_fi_dot), pk, idx, (x_pre, y_pre), (float(x_pro), float(y_pro))


def discretise(A, B, C, k_r, Vx, ts=TS_DISCRETE):
    """:151-168 - bilinear transform; the curvature term is folded into C_bar."""
    temp = np.linalg.inv(np.eye(4) - (ts * A) / 2)
    A_bar = temp @ (np.eye(4) + (ts * A) / 2)
    B_bar = temp @ B * ts
    C_bar = temp @ C * ts * k_r * Vx
    return A_bar, B_bar, C_bar


def condensed_qp(A_bar, B_bar, C_bar, e_rr, Q=None, R=R_WEIGHT, F=None):
    """:253-298 - H (12 x 12) and f (12 x 1) of  min 1/2 u'Hu + f'u,  -1 <= u <= 1."""
    N, P, n = N_STEPS, N_CTRL, N_STATE
    Q = np.diag(Q_DIAG) if Q is None else Q
    F = np.diag(F_DIAG) if F is None else F
    M = np.zeros(((N + 1) * n, n))
    M[0:n, :] = np.eye(n)
    for i in range(1, N + 1):
        M[i * n:(i + 1) * n, :] = A_bar @ M[(i - 1) * n:i * n, :]
    Cm = np.zeros(((N + 1) * n, N * P))
    Cm[n:2 * n, 0:P] = B_bar                      # the 4 x 1 B_bar is broadcast to both control columns
    for i in range(2, N + 1):
        Cm[i * n:(i + 1) * n, (i - 1) * P:i * P] = B_bar
        for j in range(i - 2, -1, -1):
            Cm[i * n:(i + 1) * n, j * P:(j + 1) * P] = A_bar @ Cm[i * n:(i + 1) * n, (j + 1) * P:(j + 2) * P]
    Cc = np.zeros(((N + 1) * n, 1))
    for i in range(1, N + 1):
        Cc[n * i:n * (i + 1), 0:1] = A_bar @ Cc[n * (i - 1):n * i, 0:1] + C_bar
    Q_bar = np.zeros(((N + 1) * n, (N + 1) * n))
    for i in range(N):
        Q_bar[i * n:(i + 1) * n, i * n:(i + 1) * n] = Q
    Q_bar[N * n:, N * n:] = F
    R_bar = np.zeros((N * P, N * P))
    for i in range(N):
        R_bar[i * P:(i + 1) * P, i * P:(i + 1) * P] = np.eye(P) * R
    H = Cm.T @ Q_bar @ Cm + R_bar
    E = Cm.T @ Q_bar @ Cc + Cm.T @ Q_bar @ M @ (np.array(e_rr).reshape(n, 1))
    return 2 * H, 2 * E


def solve_box_qp(H, f):
    """:300-311 with the certified dense solver in place of cvxopt (unique minimiser)."""
    m = H.shape[0]
    G = np.concatenate((np.identity(m), -np.identity(m)))
    h = np.ones((2 * m, 1))
    res = qp_dense.solve_qp(H, f, G, h)
    return res.x, res


def lateral_mpc(target_path, state, Vx, min_index, vehicle_para):
    """:313-337 (`_control`) from explicit inputs.  Returns a dict of every stage."""
    A, B, C = continuous_model(vehicle_para, Vx)
    e_rr, k_r, idx, pre, pro = tracking_error(target_path, state, Vx, min_index)
    A_bar, B_bar, C_bar = discretise(A, B, C, k_r, Vx)
    H, f = condensed_qp(A_bar, B_bar, C_bar, e_rr)
    u, res = solve_box_qp(H, f)
    return dict(A=A, B=B, C=C, e_rr=np.array(e_rr), k_r=k_r, min_index=idx, pre=pre, pro=pro, A_bar=A_bar, B_bar=B_bar,
                C_bar=C_bar, H=H, f=f, u=np.asarray(u).reshape(-1), steering=float(np.asarray(u).reshape(-1)[0]),
                qp=res)
