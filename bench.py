#!/usr/bin/env python
"""bench.py - planning cycles/s of the MI355X EM-Planner hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one pass of the whole planning cycle (projection -> S-L lattice DP -> QP bounds -> path QP ->
midpoints -> Frenet->Cartesian -> smoothing QP -> heading/kappa; reference test_9.py:113-218) over one batch
of synthetic scenes per GPU, with every input already resident in HBM.  Workload = BASELINE.json
configs[2] on one GPU (4096 scenes, 40x9 lattice, 8 obstacles) and configs[3] across GPUs (weak scaling:
4096 scenes per GPU, i.e. 32768 at 8 GPUs), plus the RCCL gather of the result records when N > 1.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      the DP min-plus sweep kernel (HBM bound): algorithmic bytes / mean HIP-event duration of its
                launches INSIDE the timed region (the only kernel bracketed by events there)
  cpu_baseline  the reference-structured CPU port (oracle/ref_port.py), one core, bounded sample
  cpu_baseline_pool   the same port on a pool of single-threaded processes (up to 64 host cores), whole-pool rate
  kernels_ms    mean duration of every kernel of the cycle, from a short diagnostic pass after the timed region
                with every kernel bracketed by events (the brackets themselves cost ~7 % of a step)
  roofline_dp_edge, dp_only   the FP64-VALU-bound edge kernel against the vector peak, and the DP alone (same pass)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # half the 157.3 TFLOP/s FP32 vector peak of MI355X_MICROARCH.md (public spec figure)


def cpu_baseline(cfg, n_scenes, seed0):
    """Time the reference-structured CPU path (the oracle's faithful port) on a bounded sample."""
    import contextlib
    import io
    from emplanner_carla_amd import scenes as S
    from oracle import ref_port as op
    kw = dict(sampling_res=cfg.sampling_res, row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l)
    scenes = [S.make_scene(seed0 + i, cfg) for i in range(n_scenes)]
    t0 = time.perf_counter()
    done = 0
    for sc in scenes:
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                op.plan_cycle([tuple(r) for r in sc.ref], sc.origin_xy, sc.start_xy, sc.start_v, sc.start_a, sc.obs_xy,
                              dp_kwargs=kw, obs_length=cfg.obs_length, obs_width=cfg.obs_width, verbose=False)
        except IndexError:
            pass
        done += 1
        if time.perf_counter() - t0 > 25.0:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "planning cycles/s", "cores": 1, "kind": "port",
            "sample": f"{done} scenes of the same workload (seeds {seed0}..{seed0 + done - 1}), oracle/ref_port.py "
                      f"plan_cycle (reference-structured NumPy path; QP by oracle/qp_dense.py, not cvxopt), "
                      f"{dt:.1f} s on 1 of {os.cpu_count()} host cores"}


def usable_cores(cap):
    """Host cores this process may actually use: the scheduler affinity, cut by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def cpu_baseline_pool(cfg_name, workers, per_worker):
    """The same CPU path on `workers` host cores at once (one single-threaded process per core, spawned so that no
    worker inherits this process's HIP state): whole-pool throughput over workers * per_worker scenes."""
    import multiprocessing as mp
    from oracle import cpu_pool
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
        os.environ[k] = "1"
    try:
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(cpu_pool.warm, range(workers), chunksize=1)           # processes up, modules imported
            jobs = [("CFG2", list(range(10000 + w * per_worker, 10000 + (w + 1) * per_worker))) for w in range(workers)]
            t0 = time.perf_counter()
            done = pool.map(cpu_pool.plan_seeds, jobs, chunksize=1)
            dt = time.perf_counter() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    n = sum(d[0] for d in done)
    busy = sum(d[1] for d in done)                  # CPU-seconds the workers spent planning
    quota = "unknown"
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota = fh.read().strip()
    except OSError:
        pass
    return {"value": n / dt, "per_core_value": n / busy, "affinity_cores": len(os.sched_getaffinity(0)),
            "cgroup_cpu_max": quota, "unit": "planning cycles/s", "cores": workers, "kind": "port",
            "sample": f"{n} scenes (seeds 10000..{10000 + n - 1}) over a pool of {workers} single-threaded processes "
                      f"(os.cpu_count() = {os.cpu_count()}), {per_worker} scenes each, oracle/ref_port.py plan_cycle, "
                      f"{dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scenes-per-gpu", type=int, default=4096)
    ap.add_argument("--dp-mode", choices=["two_kernel", "fused"], default="two_kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one batch in flight instead of two (emp_set_pipeline off)")
    ap.add_argument("--force-gather-path", action="store_true",
                    help="run the N > 1 per-step code (pack + gather streams) on one GPU; the gather itself is then the identity")
    ap.add_argument("--cpu-sample", type=int, default=48)
    ap.add_argument("--cpu-pool", type=int, default=-1, help="processes of the multi-core CPU baseline (0 = skip, "
                    "-1 = the cores this process may use - affinity and cgroup quota - up to 64)")
    ap.add_argument("--cpu-pool-scenes", type=int, default=16, help="scenes per pool process")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from emplanner_carla_amd import _lib as L
    from emplanner_carla_amd import dist as emp_dist
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import (Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    cfg = S.CFG2
    B = args.scenes_per_gpu
    total = B * world
    start, count = emp_dist.shard_range(total, rank, world)
    batch = S.make_batch(range(start, start + count), cfg)
    P = batch.ref.shape[1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(count, P, np.int32)), origin_xy=t(batch.origin_xy),
                  start_xy=t(batch.start_xy), start_v=t(batch.start_v), start_a=t(batch.start_a),
                  obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
    torch.cuda.synchronize()

    pl = Planner(local_rank)
    p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
    M = max_path_points(p)
    mode = L.EMP_DP_TWO_KERNEL if args.dp_mode == "two_kernel" else L.EMP_DP_FUSED
    # Two batches in flight: the back stage (path QP, Cartesian tail) of step k runs on the planner's second stream
    # while the front stage (projection, DP) of step k+1 runs on its first (include/emplanner.h, emp_set_pipeline).
    # Every step is a complete pass over the batch; the K timed steps are all finished at the closing fence.
    pipelined = not args.no_pipeline
    pl.set_pipeline(pipelined)
    ts = pl.torch_stream()

    gather_path = world > 1 or args.force_gather_path
    gs = torch.cuda.Stream(device=device) if gather_path else None        # the gather's own stream
    in_flight = []                                                        # (tensors, event) of the last gathers

    def step():
        # torch work of a step runs on the planner's own streams, ordered with its kernels without any cross-stream
        # event: output allocation on the first; for N > 1 the records are packed on the stream on which the cycle's
        # results become complete (the second one when pipelined) and gathered over RCCL on a third stream, so that
        # the gather of step k overlaps the back stage of step k+1 as well as its front stage.
        with torch.cuda.stream(ts):
            res = pl.plan_cycle(p, q, sp, max_pts=M, mode=mode, **inputs)
        if not gather_path:
            return res
        rs = pl.torch_result_stream()
        with torch.cuda.stream(rs):
            rec = emp_dist.pack_records(res, p.col, M, path_cap=emp_dist.path_capacity(M), planner=pl)
        gs.wait_stream(rs)
        with torch.cuda.stream(gs):
            out = emp_dist.gather_records(rec, total)
            done = torch.cuda.Event()
            done.record(gs)
        # the packed records and the gathered matrix stay referenced until their gather has certainly finished
        in_flight.append((rec, out, done))
        if len(in_flight) > 3:
            in_flight.pop(0)[2].synchronize()          # three steps old: long done, costs nothing
        return out

    def fence():
        pl.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    # Inside the timed region only the roofline kernel is bracketed by HIP events (an event pair costs a few
    # microseconds of stream time per launch; six bracketed kernels per step cost ~7 % of the step).
    pl.set_timing(True, only="dp_sweep")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    sweep_ms, sweep_launches = pl.kernel_ms("dp_sweep"), pl.kernel_launches("dp_sweep")
    # Per-kernel breakdown: a separate diagnostic pass AFTER the timed region, every kernel bracketed, one batch in
    # flight (the durations of overlapping kernels would not add up to anything)
    fence()
    pl.set_pipeline(False)
    for _ in range(2):
        out = step()
    fence()
    pl.set_timing(True)
    for _ in range(min(args.steps, 5)):
        out = step()
    fence()
    kernels = {}
    for name in ("project", "dp_edge", "dp_sweep", "dp_fused", "dp_enrich", "path_qp", "to_cartesian", "heading"):
        ms = pl.kernel_ms(name)
        if ms >= 0:
            kernels[name] = round(ms, 6)
    pl.set_timing(False)

    # outcome statistics of the last step (sanity: the work was really done)
    if gather_path:
        st = emp_dist.unpack_records(out, p.col, M, path_cap=emp_dist.path_capacity(M))["status"].cpu().numpy()
    else:
        st = out.status.cpu().numpy()
    ok_frac = float(((st & ~1) == 0).mean())

    if rank == 0:
        E = cfg.row + (cfg.col - 1) * cfg.row ** 2
        bytes_dp = (8 * E + 4 * cfg.row * cfg.col + 4 * cfg.col) * count          # SURVEY.md 8(d), per launch
        roof = None
        if sweep_ms > 0:
            ach = bytes_dp / (sweep_ms * 1e-3) / 1e9
            traffic = None
            side = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(side):
                try:
                    tj = json.load(open(side))
                    if tj.get("scenes_per_gpu") == count and tj.get("kernel") == "dp_sweep":
                        traffic = tj.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"kernel": "dp_sweep_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": bytes_dp, "launches_timed": sweep_launches,
                    "mean_launch_us": round(sweep_ms * 1e3, 2)}
        # Secondary figures from the diagnostic pass (event-bracketed kernels; not part of the timed region):
        # the edge-cost kernel against the FP64 vector peak with SURVEY.md 8(d)'s ALGORITHMIC flop count (obstacles
        # out of reach are skipped at run time, so fewer are executed), and the DP alone.
        extra = {}
        if "dp_edge" in kernels:
            flops = E * (40 + 10 * (36 + 7 * cfg.n_obs)) * count
            tf = flops / (kernels["dp_edge"] * 1e-3) / 1e12
            extra["roofline_dp_edge"] = {"kernel": "dp_edge_kernel", "bound": "fp64_valu", "achieved": round(tf, 2),
                                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(tf / FP64_VECTOR_PEAK_TFLOPS, 4),
                                         "algorithmic_flops_per_launch": flops,
                                         "mean_launch_us": round(kernels["dp_edge"] * 1e3, 2),
                                         "source": "diagnostic pass after the timed region"}
        if all(k in kernels for k in ("dp_edge", "dp_sweep", "dp_enrich")):
            dp_ms = kernels["dp_edge"] + kernels["dp_sweep"] + kernels["dp_enrich"]
            extra["dp_only"] = {"value": round(count / (dp_ms * 1e-3), 1), "unit": "DP plans/s per GPU",
                                "source": "sum of the three DP kernels' mean durations in the diagnostic pass"}
        value = total * args.steps / elapsed
        line = {
            "metric": "planning cycles/sec (DP+QP, 40x9 S-L lattice, 8 obs)", "value": round(value, 1),
            "unit": "planning cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]/[3]: full planning cycle per scene (projection, S-L DP, path QP, "
                                   "Frenet->Cartesian, smoothing QP, heading/kappa), inputs resident in HBM"
                                   + ("; + RCCL all_gather of result records" if world > 1 else ""),
                       "scenes_per_gpu": count, "total_scenes": total, "lattice": f"col={cfg.col} x row={cfg.row}",
                       "sample_s": cfg.sample_s, "sample_l": cfg.sample_l, "obstacles": cfg.n_obs,
                       "ref_line_points": int(P), "qp_stations": 21, "dp_mode": args.dp_mode,
                       "batches_in_flight": 2 if pipelined else 1,
                       "parallelism": f"scenes sharded over {world} GPU(s), one process per GPU"},
            "roofline": roof,
            **extra,
            "kernels_ms": kernels,
            "scenes_fully_planned_frac": round(ok_frac, 4),
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample, 0)
            workers = usable_cores(64) if args.cpu_pool < 0 else args.cpu_pool
            if workers > 0:
                try:
                    line["cpu_baseline_pool"] = cpu_baseline_pool("CFG2", workers, args.cpu_pool_scenes)
                except Exception as exc:                                       # informational: never fails the bench
                    line["cpu_baseline_pool"] = {"error": f"{type(exc).__name__}: {exc}"}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    pl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
