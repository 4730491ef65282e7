"""CPU oracle for the S-T speed DP (SURVEY.md section 8 row a-ST, BASELINE config 5).

TEST INFRASTRUCTURE ONLY (same import rule as ``oracle/ref_port.py``): only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Restates reference ``planner/speed_planning_test.py:38-305`` twice:

* the ``port_*`` functions follow the reference's floating-point route call by call (2-vectors
  as ``np.array`` with ``.dot``, scalar ``**``), so on the machine that generated the golden
  vectors they agree with the imported reference bit for bit.  ``ndarray.dot`` on this NumPy
  build fuses the multiply-add and scalar ``x ** 2`` is not always ``x * x`` (checked: 25 % /
  0.07 % of random operands differ in the last bit), i.e. the reference's own last bits depend
  on the BLAS / libm it runs on;
* the ``exact_*`` functions are the arithmetic the HIP kernel implements: every product and sum
  rounded separately in the written order, vectorised over scenes and edges.  They agree with
  the port to ~1e-15 relative (pinned in tests/test_oracle_golden.py) and are the target the GPU
  is compared with.

Pinning: ``generate_st_graph``, ``CalcSTCoordinate``, ``CalcCollisionCost``, ``CalcObsCost``,
``CalcDpCost`` are pinned by golden vectors of the imported reference; the forward tables
(``dp_st_cost``, ``dp_st_s_dot``, ``dp_st_node``) and the terminal node are pinned by running the
reference's own ``speed_DP`` and reading its locals from the frame in which it raises
(tests/golden/make_golden.py: the reference's backtrack indexes ``s_list`` with a float taken
from ``dp_st_node`` and raises ``IndexError`` at :184 whenever the terminal column is not 0).
The backtrack below therefore cannot be pinned: it is the loop of :178-186 with the predecessor
cast to int and with separate s / t output arrays (the reference aliases them, :156).

Operation order of one edge (s0, t0, v0) -> (s1, t1)   [exact_* and the kernel]:
    v   = (s1 - s0) / (t1 - t0)                 a = (v - v0) / (t1 - t0)
    ref = w_ref * ((v - v_ref) * (v - v_ref))
    acc = w_acc * (a * a)            if -6 < a < 4       else (100000 * w_acc) * (a * a)
    dt  = (t1 - t0) / 4,   sample m = 0..4:  f = m - 1,  t = t0 + f * dt,  s = s0 + (v * f) * dt
    per obstacle j (s_in not NaN), in order:
        v1 = (s_in - s, t_in - t)   v2 = (s_out - s, t_out - t)   v3 = v2 - v1
        p = v1.v3   q = v2.v3    (x*x' + y*y', two products, one add)
        if (p > 0 and q > 0) or (p < 0 and q < 0):  d = sqrt(min(v1.v1, v2.v2))   (first wins ties)
        else:                                       d = |v1x*v3y - v1y*v3x| / sqrt(v3.v3)
        c = w_obs if |d| < 0.5;  w_obs ** ((0.5 - d) + 1) if 0.5 < |d| < 1.5;  else 0
    obs = sum of c, m outer / j inner, from 0;     cost = (obs + acc) + ref
"""
from __future__ import annotations

import numpy as np

N_ROWS = 40
N_COLS = 16


def grid():
    """speed_planning_test.py:114,116 - the hard-coded non-uniform s samples and the t samples."""
    s_list = np.concatenate((np.arange(0, 5, 0.5), np.arange(5.5, 15, 1), np.arange(16, 30, 1.5), np.arange(32, 55, 2.5)))
    t_list = np.arange(0.5, 8.5, 0.5)
    return s_list, t_list


DEFAULTS = dict(reference_speed=50, w_cost_ref_speed=4000, w_cost_accel=100, w_cost_obs=10000000)  # :101-102


# --------------------------------------------------------------------------------------
# faithful port
# --------------------------------------------------------------------------------------
def port_st_coordinate(row, col, s_list, t_list):
    """:287-305 - row 0 is the LARGEST s."""
    return s_list[len(s_list) - row - 1], t_list[col]


def port_collision_cost(w_cost_obs, min_dis):
    """:274-284 - note that exactly 0.5 and anything >= 1.5 cost nothing."""
    if abs(min_dis) < 0.5:
        return w_cost_obs
    if 0.5 < abs(min_dis) < 1.5:
        return w_cost_obs ** ((0.5 - min_dis) + 1)
    return 0


def port_obs_cost(s_start, t_start, s_end, t_end, s_in_set, s_out_set, t_in_set, t_out_set, w_cost_obs):
    """:234-271 - five samples at t_start + (i-1)*dt (the first one lies BEFORE the edge)."""
    total = 0
    n = 5
    dt = (t_end - t_start) / (n - 1)
    k = (s_end - s_start) / (t_end - t_start)
    for i in range(n):
        t = t_start + (i - 1) * dt
        s = s_start + k * (i - 1) * dt
        here = np.array([s, t])
        for j in range(len(s_in_set)):
            if np.isnan(s_in_set[j]):
                continue
            v1 = np.array([s_in_set[j], t_in_set[j]]) - here
            v2 = np.array([s_out_set[j], t_out_set[j]]) - here
            v3 = v2 - v1
            dis1 = np.sqrt(v1.dot(v1))
            dis2 = np.sqrt(v2.dot(v2))
            dis3 = abs(v1[0] * v3[1] - v1[1] * v3[0]) / np.sqrt(v3.dot(v3))
            p, q = v1.dot(v3), v2.dot(v3)
            if (p > 0 and q > 0) or (p < 0 and q < 0):
                d = min(dis1, dis2)
            else:
                d = dis3
            total = total + port_collision_cost(w_cost_obs, d)
    return total


def port_dp_cost(row_start, col_start, row_end, col_end, s_in_set, s_out_set, t_in_set, t_out_set,
                 w_cost_ref_speed, reference_speed, w_cost_accel, w_cost_obs, plan_start_s_dot, s_list, t_list,
                 dp_st_s_dot):
    """:191-231 - ``row_start == 0`` means "the DP origin" even for a real row-0 node."""
    s_end, t_end = port_st_coordinate(row_end, col_end, s_list, t_list)
    if row_start == 0:
        s_start, t_start, s_dot_start = 0, 0, plan_start_s_dot
    else:
        s_start, t_start = port_st_coordinate(row_start, col_start, s_list, t_list)
        s_dot_start = dp_st_s_dot[row_start][col_start]
    cur_s_dot = (s_end - s_start) / (t_end - t_start)
    cur_s_dot2 = (cur_s_dot - s_dot_start) / (t_end - t_start)
    cost_ref_speed = w_cost_ref_speed * (cur_s_dot - reference_speed) ** 2
    if 4 > cur_s_dot2 > -6:
        cost_accel = w_cost_accel * cur_s_dot2 ** 2
    else:
        cost_accel = 100000 * w_cost_accel * cur_s_dot2 ** 2
    cost_obs = port_obs_cost(s_start, t_start, s_end, t_end, s_in_set, s_out_set, t_in_set, t_out_set, w_cost_obs)
    return cost_obs + cost_accel + cost_ref_speed


def port_generate_st_graph(obs_s, obs_l, obs_s_dot, obs_l_dot):
    """:38-98 - the scan STOPS at the first NaN s; slow lateral movers are ignored."""
    n = len(obs_s)
    s_in, s_out, t_in, t_out = (np.ones(n) * np.nan for _ in range(4))
    for i in range(n):
        if np.isnan(obs_s[i]):
            break
        if abs(obs_l_dot[i]) < 0.3:
            continue
        t_zero = -obs_l[i] / obs_l_dot[i]
        b1 = 2 / obs_l_dot[i] + t_zero
        b2 = -2 / obs_l_dot[i] + t_zero
        t_max, t_min = (b1, b2) if b1 > b2 else (b2, b1)
        if t_max < 1 or t_min > 8:
            continue
        if t_min < 0 and t_max > 0:
            s_in[i], t_in[i] = obs_s[i], 0
        else:
            s_in[i], t_in[i] = obs_s[i] + obs_s_dot[i] * t_min, t_min
        s_out[i], t_out[i] = obs_s[i] + obs_s_dot[i] * t_max, t_max
    return s_in, s_out, t_in, t_out


def port_speed_dp_tables(s_in_set, s_out_set, t_in_set, t_out_set, plan_start_s_dot, reference_speed=50,
                         w_cost_ref_speed=4000, w_cost_accel=100, w_cost_obs=10000000):
    """:114-172 - forward sweep and terminal node.  Slow (pure Python, ~24 000 edges)."""
    s_list, t_list = grid()
    m, n = len(s_list), len(t_list)
    cost = np.ones((m, n)) * np.inf
    s_dot = np.zeros((m, n))
    node = np.zeros((m, n))
    args = (s_in_set, s_out_set, t_in_set, t_out_set, w_cost_ref_speed, reference_speed, w_cost_accel, w_cost_obs,
            plan_start_s_dot, s_list, t_list, s_dot)
    for i in range(m):
        cost[i, 0] = port_dp_cost(0, 0, i, 0, *args)
        s_end, t_end = port_st_coordinate(i, 0, s_list, t_list)
        s_dot[i, 0] = s_end / t_end
    for c in range(1, n):
        for j in range(m):
            for k in range(m):
                cand = port_dp_cost(k, c - 1, j, c, *args) + cost[k, c - 1]
                if cand < cost[j, c]:
                    cost[j, c] = cand
                    s_start, t_start = port_st_coordinate(k, c - 1, s_list, t_list)
                    s_end, t_end = port_st_coordinate(j, c, s_list, t_list)
                    s_dot[j, c] = (s_end - s_start) / (t_end - t_start)
                    node[j, c] = k
    end_row, end_col = terminal_node(cost)
    return dict(cost=cost, s_dot=s_dot, node=node, end_row=end_row, end_col=end_col)


def terminal_node(cost):
    """:158-172 - right column top to bottom, then top row left to right, both with ``<=``."""
    m, n = cost.shape
    best, row, col = np.inf, -1, -1
    for i in range(m):
        if cost[i, n - 1] <= best:
            best, row, col = cost[i, n - 1], i, n - 1
    for j in range(n):
        if cost[0, j] <= best:
            best, row, col = cost[0, j], 0, j
    return row, col


def backtrack(node, end_row, end_col):
    """:155-186 with the float predecessor cast to int and s / t in separate arrays (UNPINNED: the
    reference raises IndexError here, see the module header)."""
    s_list, t_list = grid()
    speed_s = np.ones(len(t_list)) * np.nan
    speed_t = np.ones(len(t_list)) * np.nan
    if end_row < 0:
        return speed_s, speed_t
    row, col = int(end_row), int(end_col)
    speed_s[col], speed_t[col] = port_st_coordinate(row, col, s_list, t_list)
    while col != 0:
        row = int(node[row, col])
        col -= 1
        speed_s[col], speed_t[col] = port_st_coordinate(row, col, s_list, t_list)
    return speed_s, speed_t


# --------------------------------------------------------------------------------------
# exact (the kernel's arithmetic), vectorised
# --------------------------------------------------------------------------------------
def exact_collision_cost(w_cost_obs, d):
    d = np.asarray(d, dtype=np.float64)
    a = np.abs(d)
    with np.errstate(over="ignore", invalid="ignore"):
        mid = np.power(np.float64(w_cost_obs), (0.5 - d) + 1.0)
    return np.where(a < 0.5, np.float64(w_cost_obs), np.where((0.5 < a) & (a < 1.5), mid, 0.0))


def exact_obs_cost(s0, t0, s1, t1, s_in, s_out, t_in, t_out, w_cost_obs):
    """Obstacle cost of edges.  s0..t1 broadcast to a common shape ``E``; the obstacle arrays have shape
    ``E + (n_obs,)`` or broadcast to it."""
    s0, t0, s1, t1 = np.broadcast_arrays(*(np.asarray(v, dtype=np.float64) for v in (s0, t0, s1, t1)))
    s_in, s_out, t_in, t_out = (np.asarray(v, dtype=np.float64) for v in (s_in, s_out, t_in, t_out))
    with np.errstate(divide="ignore", invalid="ignore"):
        dt = (t1 - t0) / 4.0
        k = (s1 - s0) / (t1 - t0)
        total = np.zeros(s0.shape)
        valid = ~np.isnan(s_in)
        for m in range(5):
            f = float(m - 1)
            t = (t0 + f * dt)[..., None]
            s = (s0 + (k * f) * dt)[..., None]
            v1x, v1y = s_in - s, t_in - t
            v2x, v2y = s_out - s, t_out - t
            v3x, v3y = v2x - v1x, v2y - v1y
            p = v1x * v3x + v1y * v3y
            q = v2x * v3x + v2y * v3y
            d11 = v1x * v1x + v1y * v1y
            d22 = v2x * v2x + v2y * v2y
            ends = np.sqrt(np.where(d22 < d11, d22, d11))
            perp = np.abs(v1x * v3y - v1y * v3x) / np.sqrt(v3x * v3x + v3y * v3y)
            outside = ((p > 0) & (q > 0)) | ((p < 0) & (q < 0))
            c = exact_collision_cost(w_cost_obs, np.where(outside, ends, perp))
            c = np.where(valid, c, 0.0)
            for j in range(c.shape[-1]):       # ordered accumulation (m outer, j inner)
                total = total + c[..., j]
    return total


def exact_edge_cost(s0, t0, v0, s1, t1, s_in, s_out, t_in, t_out, reference_speed=50, w_cost_ref_speed=4000,
                    w_cost_accel=100, w_cost_obs=10000000, with_parts=False):
    s0, t0, v0, s1, t1 = np.broadcast_arrays(*(np.asarray(v, dtype=np.float64) for v in (s0, t0, v0, s1, t1)))
    with np.errstate(divide="ignore", invalid="ignore"):
        v = (s1 - s0) / (t1 - t0)
        a = (v - v0) / (t1 - t0)
        e = v - np.float64(reference_speed)
        ref = np.float64(w_cost_ref_speed) * (e * e)
        a2 = a * a
        acc = np.where((4 > a) & (a > -6), np.float64(w_cost_accel) * a2, (100000.0 * np.float64(w_cost_accel)) * a2)
    obs = exact_obs_cost(s0, t0, s1, t1, s_in, s_out, t_in, t_out, w_cost_obs)
    total = (obs + acc) + ref
    return (total, obs) if with_parts else total


def exact_generate_st_graph(obs_s, obs_l, obs_s_dot, obs_l_dot):
    """Batched [B, n] version of :38-98 (same expressions; the break at the first NaN s becomes a mask)."""
    obs_s, obs_l, obs_s_dot, obs_l_dot = (np.atleast_2d(np.asarray(v, dtype=np.float64)) for v in
                                          (obs_s, obs_l, obs_s_dot, obs_l_dot))
    alive = np.cumsum(np.isnan(obs_s), axis=1) == 0
    with np.errstate(divide="ignore", invalid="ignore"):
        t_zero = -obs_l / obs_l_dot
        b1 = 2 / obs_l_dot + t_zero
        b2 = -2 / obs_l_dot + t_zero
        t_max = np.where(b1 > b2, b1, b2)
        t_min = np.where(b1 > b2, b2, b1)
        keep = alive & ~(np.abs(obs_l_dot) < 0.3) & ~((t_max < 1) | (t_min > 8))
        inside = (t_min < 0) & (t_max > 0)
        s_in = np.where(inside, obs_s, obs_s + obs_s_dot * t_min)
        t_in = np.where(inside, 0.0, t_min)
        s_out = obs_s + obs_s_dot * t_max
    nan = np.nan
    return (np.where(keep, s_in, nan), np.where(keep, s_out, nan), np.where(keep, t_in, nan),
            np.where(keep, t_max, nan))


def exact_speed_dp(s_in, s_out, t_in, t_out, plan_start_s_dot, reference_speed=50, w_cost_ref_speed=4000,
                   w_cost_accel=100, w_cost_obs=10000000):
    """Batched forward sweep + terminal node + int-cast backtrack.  Obstacle arrays [B, n_obs], start [B].
    Returns dict(cost [B,40,16], s_dot, node (int32), end [B,2], speed_s [B,16], speed_t [B,16])."""
    s_list, t_list = grid()
    s_in, s_out, t_in, t_out = (np.atleast_2d(np.asarray(v, dtype=np.float64)) for v in (s_in, s_out, t_in, t_out))
    v_start = np.atleast_1d(np.asarray(plan_start_s_dot, dtype=np.float64))
    B = s_in.shape[0]
    kw = dict(reference_speed=reference_speed, w_cost_ref_speed=w_cost_ref_speed, w_cost_accel=w_cost_accel,
              w_cost_obs=w_cost_obs)
    s_node = s_list[::-1].copy()                       # s of row r
    cost = np.full((B, N_ROWS, N_COLS), np.inf)
    s_dot = np.zeros((B, N_ROWS, N_COLS))
    node = np.zeros((B, N_ROWS, N_COLS), dtype=np.int32)
    ob = lambda a: a[:, None, :]                       # [B, 1, n_obs] against edges [B, 40]
    zero = np.zeros((B, N_ROWS))
    cost[:, :, 0] = exact_edge_cost(zero, zero, v_start[:, None], s_node[None, :], t_list[0], ob(s_in), ob(s_out),
                                    ob(t_in), ob(t_out), **kw)
    s_dot[:, :, 0] = s_node[None, :] / t_list[0]
    ob2 = lambda a: a[:, None, None, :]                # against edges [B, j, k]
    for c in range(1, N_COLS):
        # edge (k, c-1) -> (j, c); k == 0 starts at the DP origin (:208-212)
        s0 = np.broadcast_to(s_node[None, None, :], (B, N_ROWS, N_ROWS)).copy()
        t0 = np.full((B, N_ROWS, N_ROWS), t_list[c - 1])
        v0 = np.broadcast_to(s_dot[:, None, :, c - 1], (B, N_ROWS, N_ROWS)).copy()
        s0[:, :, 0] = 0.0
        t0[:, :, 0] = 0.0
        v0[:, :, 0] = v_start[:, None]
        s1 = np.broadcast_to(s_node[None, :, None], (B, N_ROWS, N_ROWS))
        e = exact_edge_cost(s0, t0, v0, s1, t_list[c], ob2(s_in), ob2(s_out), ob2(t_in), ob2(t_out), **kw)
        cand = e + cost[:, None, :, c - 1]
        # ordered strict-< scan from +inf == first minimum, unless nothing beats +inf
        with np.errstate(invalid="ignore"):
            k_best = np.argmin(np.where(np.isnan(cand), np.inf, cand), axis=2)
        best = np.take_along_axis(cand, k_best[:, :, None], axis=2)[:, :, 0]
        took = best < np.inf
        cost[:, :, c] = np.where(took, best, np.inf)
        node[:, :, c] = np.where(took, k_best, 0)
        s_dot[:, :, c] = np.where(took, (s_node[None, :] - s_node[k_best]) / (t_list[c] - t_list[c - 1]), 0.0)
    end = np.zeros((B, 2), dtype=np.int32)
    speed_s = np.full((B, N_COLS), np.nan)
    speed_t = np.full((B, N_COLS), np.nan)
    for b in range(B):
        r, c = terminal_node(cost[b])
        end[b] = (r, c)
        speed_s[b], speed_t[b] = backtrack(node[b], r, c)
    return dict(cost=cost, s_dot=s_dot, node=node, end=end, speed_s=speed_s, speed_t=speed_t)
