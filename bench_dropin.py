#!/usr/bin/env python
"""bench_dropin.py - the REFERENCE'S OWN CALL SHAPE, timed (`dropin_leg` of bench.py's default line).

The API the north star keeps is one planning request at a time, Python lists in, tuples out (reference test_9.py:92-96, 220,
390-395).  Two ways an unmodified driver reaches the GPU:

  (i)  pipe        the planning process: `multiprocessing.Process(target=motion_planning, args=(conn,))` with the package's
                   `emplanner_carla_amd.service.motion_planning` in place of the reference's (test_9.py:225-227), the request tuple
                   of test_9.py:390-392 sent down a real `multiprocessing.Pipe`, the reply tuple of :220 received - round-trip
                   wall time per request in the DRIVER process, pickling both ways included;
  (ii) functions   the body of the reference's planning loop (test_9.py:113-218) written against the drop-in modules
                   `emplanner_carla_amd.planner.planning_utils` / `path_planning`: find_match_points, sampling,
                   smooth_reference_line, cal_s_map_fun, cal_s_l_fun (obstacles, start), cal_s_l_deri_fun, DP_algorithm, [::2],
                   cal_lmin_lmax, Quadratic_planning, the midpoint re-interleave, frenet_2_x_y_theta_kappa - eleven synchronous
                   library calls with list <-> array conversion around each;
  and, for scale, (iii) the same request through `service.plan_requests` in-process (two device calls).

Lattice: the reference's own keyword defaults (path_planning.py:277-279: 6 x 12, sample_s 15), which is what its driver plans on.
Prints ONE JSON line.  `python bench_dropin.py [--requests 200]`."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_request(seed, n_static=3):
    """A request tuple as the reference's driver sends it (test_9.py:390-392): ([(x, y, dis)], [(x, y, dis, speed)], vehicle_loc,
    pred_loc, vehicle_v, vehicle_a, global path [(x, y, theta, kappa)], [previous match index]).  A gently winding 2 m-spaced
    global path of 240 nodes, the vehicle near node 20-150, static obstacles 16-50 m ahead beside the lane (the nearest within 30 m:
    test_9.py:117 counts them), no dynamic obstacle."""
    rng = np.random.default_rng(seed)
    n = 240
    t = np.arange(n) * 2.0
    ph = rng.uniform(0, 3)
    x, y = t, 30.0 * np.sin(t / 70.0 + ph)
    dx, dy = np.gradient(x), np.gradient(y)
    th = np.arctan2(dy, dx)
    ka = np.gradient(th) / np.hypot(dx, dy)
    path = [(float(a), float(b), float(c), float(d)) for a, b, c, d in zip(x, y, th, ka)]
    at = int(rng.integers(20, 150))
    px, py, pth, _ = path[at]
    off = rng.normal(0, 0.25)
    veh = (px - off * math.sin(pth) - 1.2 * math.cos(pth), py + off * math.cos(pth) - 1.2 * math.sin(pth))
    pred = (px - off * math.sin(pth), py + off * math.cos(pth))
    speed = rng.uniform(6.0, 11.0)
    v = (speed * math.cos(pth + 0.02), speed * math.sin(pth + 0.02))
    a = (rng.normal(0, 0.3), rng.normal(0, 0.3))
    static = []
    for k, (ahead, lat) in enumerate(((8, 4.6), (16, -4.8), (24, 5.2), (12, -5.5), (20, 5.0))[:n_static]):
        ox, oy, oth, _ = path[at + ahead]
        ox, oy = ox - lat * math.sin(oth), oy + lat * math.cos(oth)
        static.append((ox, oy, math.hypot(ox - veh[0], oy - veh[1])))
    static.sort(key=lambda o: o[2])
    return (static, [], veh, pred, v, a, path, [max(at - 2, 0)])


def plan_by_functions(request, pu, pp):
    """The planning loop body of the reference (test_9.py:113-218) against the drop-in modules `pu` (planning_utils) and `pp`
    (path_planning): the same functions in the same order with the same arguments, one request."""
    static, dynamic, veh, pred, v, a, gpath, match = request
    match, _ = pu.find_match_points(xy_list=[pred], frenet_path_node_list=gpath, is_first_run=False, pre_match_index=match[0])
    local = pu.sampling(match[0], gpath, back_length=10, forward_length=50)
    line = pu.smooth_reference_line(local)
    s_map = pu.cal_s_map_fun(line, origin_xy=veh)
    if len(static) != 0 and static[0][-1] <= 30:
        obs_s, obs_l = pu.cal_s_l_fun([(o[0], o[1]) for o in static], line, s_map)
    else:
        obs_s, obs_l = [], []
    begin_s, begin_l = pu.cal_s_l_fun([pred], line, s_map)
    if len(dynamic) != 0:                                        # the virtual obstacles of the first dynamic obstacle (:137-169)
        dis, v_obs = dynamic[0][2], dynamic[0][3]
        dv = math.hypot(v[0], v[1]) - v_obs
        meet_t = (dis - 2.910 / 2 - 3 / 2) / dv
        leave_t = meet_t + (2.910 + 3) / dv
        meet_s = begin_s[0] + dis + v_obs * meet_t - 3 / 2
        leave_s = begin_s[0] + dis + v_obs * leave_t + 3 / 2
        if leave_s < 80:
            obs_s += [meet_s - 10, meet_s + (leave_s - meet_s) / 2, leave_s]
            obs_l += [0, 0, 0]
    l0, _, _, _, dl0, _, ddl0 = pu.cal_s_l_deri_fun(xy_list=[pred], V_xy_list=[v], a_xy_list=[a], local_path_xy_opt=line,
                                                   origin_xy=pred)
    dp_s, dp_l = pp.DP_algorithm(obs_s, obs_l, plan_start_s=begin_s[0], plan_start_l=l0[0], plan_start_dl=dl0[0],
                                 plan_start_ddl=ddl0[0])
    dp_l, dp_s = dp_l[::2], dp_s[::2]
    l_min, l_max = pp.cal_lmin_lmax(dp_path_s=dp_s, dp_path_l=dp_l, obs_s_list=obs_s, obs_l_list=obs_l, obs_length=5, obs_width=5)
    ql, _, _ = pp.Quadratic_planning(l_min, l_max, plan_start_l=l0[0], plan_start_dl=dl0[0], plan_start_ddl=ddl0[0])
    path_s, path_l = [dp_s[0]], [ql[0]]
    for i in range(1, len(ql)):
        path_s.append((dp_s[i] + dp_s[i - 1]) / 2)
        path_l.append((ql[i] + ql[i - 1]) / 2)
    path_s.append(dp_s[-1])
    path_l.append(ql[-1])
    traj = pp.frenet_2_x_y_theta_kappa(plan_start_s=begin_s[0], plan_start_l=begin_l[0], enriched_s_list=path_s,
                                       enriched_l_list=path_l, frenet_path_opt=line, s_map=s_map)
    return traj, match, path_s, path_l


def _stats(ms):
    ms = np.sort(np.asarray(ms))
    return {"median_ms": round(float(np.median(ms)), 4), "p95_ms": round(float(ms[int(0.95 * (len(ms) - 1))]), 4),
            "mean_ms": round(float(ms.mean()), 4), "min_ms": round(float(ms[0]), 4), "requests": int(len(ms))}


def pipe_leg(requests, warm=20):
    """(i): a spawned planning process on the package's motion_planning, a real Pipe, one request in flight."""
    import multiprocessing as mp
    from emplanner_carla_amd import service
    ctx = mp.get_context("spawn")          # the driver process may hold a HIP context already: the planner process starts clean
    parent, child = ctx.Pipe()
    proc = ctx.Process(target=service.motion_planning, args=(child,), kwargs={"on_infeasible": "sentinel"}, daemon=True)
    proc.start()
    try:
        ms, replies = [], []
        for k, req in enumerate([requests[i % len(requests)] for i in range(warm)] + list(requests)):
            t0 = time.perf_counter()
            parent.send(req)
            if not parent.poll(120.0):
                raise TimeoutError("the planning process did not answer within 120 s")
            rep = parent.recv()
            if k >= warm:
                ms.append((time.perf_counter() - t0) * 1e3)
                replies.append(rep)
        return _stats(ms), replies
    finally:
        proc.terminate()
        proc.join(10)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=200)
    ap.add_argument("--skip-pipe", action="store_true")
    args = ap.parse_args(argv)
    t_all = time.perf_counter()
    reqs = [make_request(1000 + k) for k in range(args.requests)]
    line = {"workload": "one planning request at a time in the reference's own call shape (test_9.py:92-220, 390-395): Python lists "
                        "in, tuples out, the reference's default lattice (6 x 12, sample_s 15), 3 static obstacles, 240-node global path",
            "requests": args.requests}
    pipe_replies = None
    if not args.skip_pipe:
        line["pipe_motion_planning"], pipe_replies = pipe_leg(reqs)
        line["pipe_motion_planning"]["what"] = ("round trip in the driver process: conn.send(request) -> a spawned process running "
                                                "emplanner_carla_amd.service.motion_planning(conn) -> conn.recv(); pickling both ways included")
    from emplanner_carla_amd import service
    from emplanner_carla_amd.planner import _runtime, path_planning as pp, planning_utils as pu
    import contextlib
    import io
    # (ii) the explicit function sequence
    ms, planned, refused, fn_replies = [], 0, 0, []
    for k, req in enumerate(reqs[:10] + reqs):
        t0 = time.perf_counter()
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                rep = plan_by_functions(req, pu, pp)
        except (ValueError, IndexError):
            rep = None
        if k >= 10:
            ms.append((time.perf_counter() - t0) * 1e3)
            fn_replies.append(rep)
            planned += rep is not None
            refused += rep is None
    line["function_sequence"] = dict(_stats(ms), planned=planned, refused=refused,
                                     what="the body of test_9.py:113-218 against emplanner_carla_amd.planner.* : eleven synchronous library "
                                          "calls, list <-> array conversion around each")
    # (iii) the same requests through service.plan_requests in-process (two device calls per request)
    pl = _runtime.planner()
    ms = []
    for k, req in enumerate(reqs[:10] + reqs):
        t0 = time.perf_counter()
        service.plan_requests(pl, [req])
        if k >= 10:
            ms.append((time.perf_counter() - t0) * 1e3)
    line["plan_requests_in_process"] = dict(_stats(ms), what="service.plan_requests(planner, [request]): emp_reference_line + emp_plan_cycle, host arrays")
    # (iv) the planning process's own fast path, in-process: service.RequestPlanner (what motion_planning runs per request)
    one = service.RequestPlanner(pl)
    ms = []
    for k, req in enumerate(reqs[:10] + reqs):
        t0 = time.perf_counter()
        one.plan(req)
        if k >= 10:
            ms.append((time.perf_counter() - t0) * 1e3)
    one.close()
    line["request_planner_in_process"] = dict(_stats(ms), what="service.RequestPlanner.plan(request): the request written into one page-locked block, one emp_plan_cycle call")
    # the three routes agree (same kernels underneath; the function sequence hands float64 lists from call to call)
    worst, compared = 0.0, 0
    if pipe_replies is not None:
        for rp, rf in zip(pipe_replies, fn_replies):
            if rp is None or rf is None or rp[0] is None:
                continue
            a, b = np.asarray(rp[0], dtype=np.float64), np.asarray(rf[0], dtype=np.float64)
            if a.shape == b.shape:
                worst = max(worst, float(np.abs(a[:, :2] - b[:, :2]).max()))
                compared += 1
        line["pipe_vs_function_sequence"] = {"trajectories_compared": compared, "max_abs_xy_difference_m": worst}
    line["wall_s"] = round(time.perf_counter() - t_all, 2)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
