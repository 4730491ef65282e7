#!/usr/bin/env python
"""bench.py - planning cycles/s of the MI355X EM-Planner hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one pass of the whole planning cycle (projection -> S-L lattice DP -> QP bounds -> path QP ->
midpoints -> Frenet->Cartesian -> smoothing QP -> heading/kappa; reference test_9.py:113-218) over one batch
of synthetic scenes per GPU, with every input already resident in HBM.  Default workload = BASELINE.json
configs[2] on one GPU (4096 scenes, 40x9 lattice, 8 obstacles) and configs[3] across GPUs (weak scaling:
4096 scenes per GPU, i.e. 32768 at 8 GPUs), plus the RCCL gather of the result records when N > 1.

Other lines (one JSON line per invocation, same keys):
    --config cfg5      BASELINE configs[4]: 120x21 lattice, 16 obstacles, full cycle + the S-T speed DP (40x16 grid, 16
                       dynamic-obstacle slots; reference speed_planning_test.py:38-188) per scene, 4096 scenes per GPU
    --latency          BASELINE configs[1]: ONE scene on the 40x9 lattice, synchronous calls (metric: ms per cycle)
    --scene-dist X     obstacle layout of the synthetic scenes: corridor (default, emplanner_carla_amd/scenes.py), survey
                       (s_k = 12 + 11 k +- 2, l_k = +-U(2.5, 5)), worst (all obstacles within reach
                       of the same columns)
    --dp-mode fused    the single-kernel DP (no HBM edge tensor; no sweep kernel, so `roofline` is null)

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      the DP min-plus sweep kernel (HBM bound): algorithmic bytes / mean HIP-event duration of its
                launches INSIDE the timed region (the only kernel bracketed by events there).  The timed region runs three
                batches on three lanes: the sweep shares the chip with two other batches' edge-cost kernels, so `frac` is the
                schedule's figure; `frac_alone` (diagnostic pass), `staged_leg` and `exclusive_sweep_leg` are the kernel's.
                `traffic` = HBM bytes
                per launch from the rocprofv3 PMC passes named in `traffic_source`, or null when no profile of this
                workload is committed
  cpu_baseline  the reference-structured CPU port (oracle/ref_port.py), one core, bounded sample
  cpu_baseline_pool   the same port on a pool of single-threaded processes (up to 64 host cores), whole-pool rate
  kernels_ms    mean duration of every kernel of the cycle, from a short diagnostic pass after the timed region
                with every kernel bracketed by events, one batch in flight
  roofline_dp_edge, dp_only   the FP64-issue-bound edge kernel (algorithmic flops of SURVEY.md 8d AND the executed
                wave-level instruction / active-lane counts of the committed SQ counter profile), and the DP alone
  value / all_scenes_cycles_per_s   `value` counts the scenes whose cycle ran to the end (scenes_fully_planned_frac of the batch;
                the rest are refused at the end of the cycle: walls, blocked corridors, infeasible QPs - they were computed
                too and cost the same time); all_scenes_cycles_per_s = scenes per step / ms_per_step
  roofline_step   the whole step against the chip's vector-issue capacity: sum over the step's kernels of their VALU-busy
                quad-cycles (committed SQ counter pass, source named) / (1024 SIMDs x clock / 4 x ms_per_step), with the measured
                cost of a wave64 FP64 instruction (tools/fp64_pipe_bench.hip: 4.3 cycles, not the 4 the counter charges)
  staged_leg, exclusive_sweep_leg, rccl_gather_leg, dram_leg, cfg5_leg, latency_leg, survey_leg, tight_corridor_leg, host_io_leg, gather_path_leg, dropin_leg
                (default run only: N = 1, cfg2, 4096 scenes; --no-legs skips them) short secondary measurements after the
                headline: the staged pipeline of rounds 2-5 (two batches, front stage beside back stage: ~9 % longer steps, the
                sweep beside the Cartesian tail only) and the same with the sweep held back behind the previous batch's path QP
                (EMP_OPT_SWEEP_EXCLUSIVE = 2: the sweep's bandwidth at its best); 32768 scenes (the edge tensor streams from HBM instead of the
                Infinity Cache); BASELINE configs[4] (120x21 lattice + S-T speed DP) on 4096 scenes; configs[1] (one scene per
                synchronous call); SURVEY 8(d)'s own geometry (arc radii 150-1000 m) with its slalom layout and with the
                corridor layout; the host path (NumPy arrays in and out through the page-locked ring, PCIe included); the
                per-step code of an N > 1 rank (record packing + gather streams) on this one GPU - without a process group
                (gather_path_leg: the identity) and, in a process of its own, with a one-rank ProcessGroupNCCL whose dist.gather is a
                real RCCL call every step (rccl_gather_leg)
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



def cpu_baseline(cfg, n_scenes, seed0, scene_kw=None, budget_s=25.0):
    """Time the reference-structured CPU path (the oracle's faithful port) on a bounded sample."""
    scene_kw = dict(scene_kw or {})
    import contextlib
    import io
    from emplanner_carla_amd import scenes as S
    from oracle import ref_port as op
    kw = dict(sampling_res=cfg.sampling_res, row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l)
    scenes = [S.make_scene(seed0 + i, cfg, **scene_kw) for i in range(n_scenes)]
    t0 = time.perf_counter()
    done = 0
    for sc in scenes:
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                op.plan_cycle([tuple(r) for r in sc.ref], sc.origin_xy, sc.start_xy, sc.start_v, sc.start_a, sc.obs_xy,
                              dp_kwargs=kw, obs_length=cfg.obs_length, obs_width=cfg.obs_width, verbose=False)
        except (IndexError, np.linalg.LinAlgError):
            pass
        done += 1
        if time.perf_counter() - t0 > budget_s:
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


def cpu_baseline_pool(cfg_name, workers, per_worker, scene_kw=None):
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
            jobs = [(cfg_name, list(range(10000 + w * per_worker, 10000 + (w + 1) * per_worker)), dict(scene_kw or {}))
                    for w in range(workers)]
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


def latency_main(args):
    """BASELINE configs[1]: one scene on the 40x9 lattice, one synchronous call per cycle (what a driver that plans for
    one vehicle sees).  Prints one JSON line; `value` is the mean wall time of a cycle in milliseconds."""
    import torch
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
    cfg = S.CFG2
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    batch = S.make_batch([7], cfg, **scene_kwargs(args))
    P = batch.ref.shape[1]
    host = dict(ref_line=batch.ref, n_ref=np.full(1, P, np.int32), origin_xy=batch.origin_xy, start_xy=batch.start_xy,
                start_v=batch.start_v, start_a=batch.start_a, obs_xy=batch.obs_xy, n_obs=batch.n_obs)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in host.items()}
    pl = Planner(0)
    p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
    M = max_path_points(p)
    steps, warm = max(args.steps, 20), max(args.warmup, 5)
    out = {}
    for name, inputs in (("device_resident_inputs", dev), ("host_arrays_in_and_out", host)):
        for _ in range(warm):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **inputs)
            pl.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **inputs)
            pl.synchronize()
        out[name] = (time.perf_counter() - t0) / steps * 1e3
    pl.set_timing(True)
    for _ in range(5):
        r = pl.plan_cycle(p, q, sp, max_pts=M, **dev)
        pl.synchronize()
    kernels = {n: round(pl.kernel_ms(n), 6) for n in ("project", "dp_edge", "dp_sweep", "dp_enrich", "path_qp", "to_cartesian")
               if pl.kernel_ms(n) >= 0}
    st = int(np.asarray(r.status.cpu() if hasattr(r.status, "cpu") else r.status)[0])
    line = {"metric": "planning cycle latency (DP+QP, 40x9 S-L lattice, 8 obs, ONE scene)", "value": round(out["device_resident_inputs"], 4),
            "unit": "ms per planning cycle", "n_gpus": 1, "steps": steps, "warmup": warm,
            "ms_per_step": round(out["device_resident_inputs"], 4), "higher_is_better": False, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: one scene, full planning cycle, one synchronous call per cycle, inputs "
                                   "resident in HBM", "scenes_per_gpu": 1, "lattice": f"col={cfg.col} x row={cfg.row}",
                       "obstacles": cfg.n_obs, "scene_dist": args.scene_dist},
            "roofline": None, "latency_ms": {k: round(v, 4) for k, v in out.items()}, "kernels_ms": kernels,
            "scene_status": st,
            "note": "a single scene occupies one wavefront per kernel: the cycle is launch and dependent-instruction latency, "
                    "no roofline applies"}
    if not args.no_cpu_baseline:
        cb = cpu_baseline(cfg, 8, 7, scene_kwargs(args), budget_s=10.0)
        line["cpu_baseline"] = {**cb, "value": round(1e3 / cb["value"], 2), "unit": "ms per planning cycle"}
    print(json.dumps(line), flush=True)
    pl.close()


def scene_kwargs(args):
    """scenes.make_scene options of the run: obstacle layout, planning start (default: off the reference-line nodes), arc radii."""
    from emplanner_carla_amd import scenes as S
    return dict(dist=args.scene_dist, start_ahead=args.start_ahead,
                radius_range=S.SURVEY_ARCS if args.arcs == "survey" else S.GENTLE_ARCS)


DEFAULT_PIPELINE = "3"      # --pipeline auto: three lanes (see main)
MAX_TIMED_BLOCKS = 64       # the timed block is repeated until --min-timed-ms are covered, this often at most

from bench_legs import (HBM_PEAK_GBS, _device_inputs, committed_parity_sweeps, committed_profile, dropin_subprocess_leg, gather_path_leg, host_io_leg,  # noqa: E402
                        latency_leg, rccl_gather_subprocess_leg, roofline_step, secondary_leg, staged_subprocess_leg)


def self_launch_argv(argv, n, port, python=None):
    """The command a bare `python bench.py --gpus N ...` (N > 1, no WORLD_SIZE in the environment) re-executes itself as: the
    driver's own multi-rank form, one process per GPU under torch.distributed.run on the loopback address, `argv` (sys.argv[1:])
    passed through unchanged."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__), *argv]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and nobody having launched the ranks: launch them (reference test_9.py:225-227 is its
    only process split - a planner process beside the driver; here N planner processes, one per GPU).  Rank 0's ONE JSON line is
    passed through on stdout, everything else the children print goes to stderr; the exit code is torch.distributed.run's."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = self_launch_argv(argv, args.gpus, port)
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = []
    for ln in proc.stdout:
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            lines.append(ln.rstrip())
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 1
        print("[bench] the ranks exited without a JSON line", file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=["cfg2", "cfg5"], default="cfg2", help="cfg2 = BASELINE configs[2]/[3] (default); cfg5 = configs[4]")
    ap.add_argument("--scenes-per-gpu", type=int, default=0, help="default 4096")
    ap.add_argument("--total-scenes", type=int, default=0,
                    help="STRONG scaling: this many scenes in all, sharded contiguously over the ranks (BASELINE configs[3] as "
                         "written: --total-scenes 32768 = 16384 / 8192 / 4096 per GPU at 2 / 4 / 8); the line then says "
                         "\"scaling\": \"strong\".  Default 0 = weak scaling at --scenes-per-gpu per GPU")
    ap.add_argument("--scene-dist", choices=["corridor", "survey", "worst"], default="corridor")
    ap.add_argument("--start-ahead", type=float, default=2.7,
                    help="planning start, metres ahead of the ego (scenes.BENCH_START_AHEAD = 2.7: off the reference-line nodes; 2.0 "
                         "puts it ON node 6 in three scenes of four - the batch rounds 1-4 benchmarked)")
    ap.add_argument("--input-batches", type=int, default=0,
                    help="resident input batches the steps rotate through (different scenes every step); 0 = 4 up to 8192 scenes per "
                         "GPU, else 1")
    ap.add_argument("--arcs", choices=["gentle", "survey"], default="gentle",
                    help="arc radii of the reference lines: gentle = U(1500, 6000) m (default), survey = SURVEY 8(d)'s U(150, 1000) m")
    ap.add_argument("--latency", action="store_true", help="BASELINE configs[1]: one scene, synchronous calls")
    ap.add_argument("--dp-mode", choices=["two_kernel", "fused"], default="two_kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", default="auto", help="emp_set_pipeline mode: 'staged' (two batches, back stage of one over "
                    "the front stage of the next; the sweep keeps its bandwidth), 'off', or n >= 2 = n batches on n lanes "
                    "(more overlap: shorter steps, every kernel's own launches longer); 'auto' (default) = emp_set_pipeline(EMP_PIPELINE_AUTO): "
                    "three lanes when every stream gets a hardware queue (bench.py asks for 12), else staged")
    ap.add_argument("--no-pipeline", action="store_true", help="the same as --pipeline off")
    ap.add_argument("--one-rank-rccl", action="store_true", help="with --force-gather-path on one GPU: a one-rank ProcessGroupNCCL, "
                    "so that every step's gather is a real RCCL call (rccl_gather_leg of the default run)")
    ap.add_argument("--settle-steps", type=int, default=-1, help="untimed steps before the timed region, warm-up included "
                    "(clock settling; 0 = only the --warmup steps; default: ~50 ms of work - 150 steps at 4096 scenes of "
                    "config 2, fewer for bigger steps)")
    ap.add_argument("--alt-pipeline", default="none", help="a second timed region in this pipeline mode (e.g. 3), reported as "
                    "'alt_pipeline' next to the headline (N = 1 only; off by default so that a profile of the default "
                    "command holds one mode's launches only)")
    ap.add_argument("--force-gather-path", action="store_true",
                    help="run the N > 1 per-step code (pack + gather streams) on one GPU; the gather itself is then the identity")
    ap.add_argument("--gather", choices=["rank0", "all", "none"], default="rank0",
                    help="N > 1: gather the records to rank 0 (default: what BASELINE's 'RCCL gather' asks for) or all_gather them")
    ap.add_argument("--records", choices=["full", "trajectory"], default="full",
                    help="what a record carries: everything a cycle returns (179 doubles per scene at 40x9) or status + "
                         "trajectory only (94)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="emp_set_option before the pipeline is set up (include/emplanner.h emp_option; names: "
                         "emplanner_carla_amd._lib.OPTIONS), e.g. --opt sweep_exclusive=2 --opt lane_edge_order=1; repeatable")
    ap.add_argument("--no-legs", action="store_true",
                    help="skip the secondary legs of the default run (N = 1, config cfg2, default batch): exclusive_sweep_leg "
                         "(the same steps with the sweep held back behind the previous batch's path QP, --opt sweep_exclusive=2), "
                         "dram_leg (32768 scenes: the edge tensor streams from HBM), cfg5_leg (BASELINE configs[4], 4096 scenes), "
                         "latency_leg (configs[1], one scene per call), survey_leg / tight_corridor_leg (SURVEY 8(d)'s arc radii), "
                         "host_io_leg (NumPy in and out, PCIe included), gather_path_leg (the N > 1 per-step code on one GPU)")
    ap.add_argument("--min-timed-ms", type=float, default=50.0,
                    help="repeat the timed block of --steps steps until this much time is covered (at most 64 blocks) and report the "
                         "median block (ms_per_step), the fastest and the slowest (ms_per_step_min_max, timed_blocks); 0 = one block")
    ap.add_argument("--cpu-sample", type=int, default=48)
    ap.add_argument("--cpu-pool", type=int, default=-1, help="processes of the multi-core CPU baseline (0 = skip, "
                    "-1 = the cores this process may use - affinity and cgroup quota - up to 64)")
    ap.add_argument("--cpu-pool-scenes", type=int, default=16, help="scenes per pool process")
    args = ap.parse_args()
    if args.latency:
        return latency_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args, sys.argv[1:])

    from emplanner_carla_amd import _lib as L
    # lane mode: a hardware queue per stream; must precede HIP's initialisation (_lib.py).  Twelve since round 5: three lanes, the
    # main stream, the gather stream and torch's fit into eight - until RCCL brings its own streams (N > 1): a one-rank
    # ProcessGroupNCCL with an all_gather per step beside three lanes measured 0.243 ms per step on 8 queues, 0.211 on 12 or 16 or
    # 24 (staged: 0.227 either way; without the collective 0.197 whatever the count) - tools/rccl_lanes_probe.py.  Not more than
    # needed: the queues are the GPU's, and two processes of 16 on ONE GPU (the two-rank test) ran four times slower than of 12
    L.configure_hw_queues(12)
    import torch
    import torch.distributed as dist
    from emplanner_carla_amd import dist as emp_dist
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import (Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params,
                                         speed_dp_params)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # (test hook: EMP_BENCH_BACKEND=gloo runs the N > 1 code with several ranks on ONE GPU - RCCL refuses two ranks on a
    # device, gloo moves CUDA tensors through the host - so that the whole multi-rank step loop can be exercised on a
    # one-GPU box; tests/test_gpu_bench.py.  The driver's runs never set it.)
    backend = os.environ.get("EMP_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"[bench] {world} ranks on {torch.cuda.device_count()} visible GPU(s): RCCL wants one device per rank "
                         f"(--gpus N must not exceed the node's GPUs)")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # RCCL prints a version banner on the process's C stdout when its first communicator comes up; rank 0's stdout is ONE JSON
    # line, so file descriptor 1 points at stderr while a process group initialises
    class _StdoutToStderr:
        def __enter__(self):
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        def __exit__(self, *exc):
            import ctypes
            try:
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(self.saved, 1)
            os.close(self.saved)
    pg_one = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with _StdoutToStderr():
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=device)
                dist.barrier()
            else:
                dist.init_process_group(backend=backend)
    elif args.one_rank_rccl:
        # a process group of ONE rank on the real backend: the per-step gather below is then an RCCL dist.gather (the identity,
        # through ProcessGroupNCCL, its streams and its kernels) - what a one-GPU box can exercise of the N > 1 exchange
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        with _StdoutToStderr():
            dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
            dist.barrier()
        pg_one = True

    wide = args.config == "cfg5"
    cfg = S.CFG5 if wide else S.CFG2
    if args.total_scenes > 0:                         # strong scaling: the total is fixed, a rank's shard shrinks with N
        if args.scenes_per_gpu:
            raise SystemExit("--total-scenes and --scenes-per-gpu are two ways of saying the size: give one")
        total = args.total_scenes
        B = emp_dist.shard_range(total, 0, world)[1]  # the largest shard (rank 0's)
    else:
        B = args.scenes_per_gpu or 4096
        total = B * world
    scaling = "strong" if args.total_scenes > 0 else "weak"
    start, count = emp_dist.shard_range(total, rank, world)
    scene_kw = scene_kwargs(args)
    # DIFFERENT SCENES EVERY STEP: --input-batches resident batches (default 4 up to 8192 scenes per GPU, else 1), step i plans
    # batch i mod n.  Batch b holds seeds [b * total, (b + 1) * total): batch 0 is the one the parity tests and sweeps cover.
    n_in = args.input_batches if args.input_batches > 0 else (4 if B <= 8192 else 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    inputs_ring, st_ring = [], []
    for ib in range(n_in):
        lo = ib * total + start
        batch = S.make_batch(range(lo, lo + count), cfg, **scene_kw)
        P = batch.ref.shape[1]
        inputs_ring.append(dict(ref_line=t(batch.ref), n_ref=t(np.full(count, P, np.int32)), origin_xy=t(batch.origin_xy),
                                start_xy=t(batch.start_xy), start_v=t(batch.start_v), start_a=t(batch.start_a),
                                obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs)))
        if wide:    # the S-T half of configs[4]: 16 dynamic-obstacle slots per scene (reference speed_planning_test.py:38-188)
            dyn = S.make_dynamic_batch(range(lo, lo + count), 16)
            st_ring.append(([t(a) for a in dyn[:4]], t(dyn[4])))
    step_no = [0]
    torch.cuda.synchronize()

    pl = Planner(dev_index)
    p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
    sdp = speed_dp_params()
    M = max_path_points(p)
    mode = L.EMP_DP_TWO_KERNEL if args.dp_mode == "two_kernel" else L.EMP_DP_FUSED
    # Several batches in flight (include/emplanner.h, emp_set_pipeline).  Every step is a complete pass over the batch;
    # the K timed steps are all finished at the closing fence.
    # Default: three lanes - each batch runs its kernels in order on a stream of its own, three batches deep.  Until round 5 the
    # headline was taken in the STAGED form (front stage of batch k+1 beside the back stage of batch k), which keeps the sweep
    # nearly alone on the chip; lanes are the faster form on every workload measured (4096 scenes 0.220 -> 0.201 ms per step,
    # 1024 scenes 0.19 -> 0.12, 8192 0.416 -> 0.368, 32768 1.47 -> 1.45, configs[4] the same), so they are what the headline
    # runs - the staged form and its sweep are the `staged_leg` / `exclusive_sweep_leg` of the same line.
    options = {}
    for kv in args.opt:
        name, val = kv.split("=")
        pl.set_option(name, int(val))
        options[name] = int(val)
    auto_choice = None
    if args.pipeline == "auto" and not args.no_pipeline:
        # emp_set_pipeline(EMP_PIPELINE_AUTO): three lanes if every stream of this process gets a hardware queue of its own - the
        # queues configure_hw_queues(12) asked for above, unless the environment already said otherwise - else the staged form.
        # Beside the lanes and the planner's own streams the process runs torch's stream and, with the exchange, the gather
        # stream and RCCL's.
        will_gather = (world > 1 or args.force_gather_path) and args.gather != "none"
        if "foreign_streams" not in options:
            pl.set_option("foreign_streams", 1 + (2 if will_gather else 0))
        pl.set_pipeline("auto")
        form, queues, others = pl.pipeline_form()
        auto_choice = {"form": "staged" if form == 1 else f"{form} lanes", "hardware_queues": queues, "streams_beside_the_lanes": others}
        args.pipeline = "staged" if form == 1 else str(form)
        pmode = form
    else:
        pmode = 0 if (args.no_pipeline or args.pipeline == "off") else (1 if args.pipeline == "staged" else int(args.pipeline))
        pl.set_pipeline(pmode)
    pipelined, in_flight = pl.pipelined, pl.in_flight
    ts = pl.torch_stream()

    gather_path = (world > 1 or args.force_gather_path) and args.gather != "none"     # "none": compute scaling only
    sg = None
    if gather_path:      # the per-step result exchange (emplanner_carla_amd/dist.py StepGather): pack on the result stream,
        sg = emp_dist.StepGather(p.col, M, total, planner=pl, fields=args.records, device=device,     # gather on its own
                                 dst=0 if args.gather == "rank0" else None, timing=True, alone_too=pg_one)
    with_gather = [gather_path]

    def step():
        # torch work of a step runs on the planner's own streams, ordered with its kernels without any cross-stream
        # event: output allocation on the first; for N > 1 the records are packed on the stream on which the cycle's
        # results become complete (the step's lane when pipelined) and gathered over RCCL on a stream of its own, so that
        # the gather of step k overlaps the steps behind it.
        ib = step_no[0] % n_in
        step_no[0] += 1
        with torch.cuda.stream(ts):
            res = pl.plan_cycle(p, q, sp, max_pts=M, mode=mode, **inputs_ring[ib])
            if wide:      # the S-T half reads nothing of the cycle: it need not wait for the cycles in flight
                pl.set_fence(False)
                sets = pl.st_graph(*st_ring[ib][0])
                pl.speed_dp(sdp, *sets, st_ring[ib][1], tables=False)
                pl.set_fence(True)
        return sg.submit(res) if with_gather[0] else res, res

    def fence():
        pl.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out, res = step()
    # Settling, untimed like the warm-up: a handful of warm-up steps is ~2 ms of GPU work, after which the chip is not yet at
    # its sustained clocks - 20 timed steps measured 0.354 ms per step behind 5 warm-up steps and 0.330 behind 150 (the
    # same 0.330 that 100 timed steps give either way).  The same fixed number of extra steps on every rank.
    settle_total = args.settle_steps if args.settle_steps >= 0 else (16 if wide else max(16, min(150, 150 * 4096 // max(B, 1))))
    settle = max(0, settle_total - args.warmup)
    for _ in range(settle):
        out, res = step()
    fence()
    # Inside the timed region only the roofline kernel is bracketed by HIP events (an event pair costs a few
    # microseconds of stream time per launch; six bracketed kernels per step cost ~7 % of the step).
    pl.set_timing(True, only="dp_sweep")
    if sg is not None:
        sg.timed = []                    # the gathers of the timed steps only
    # The timed region: EXACTLY args.steps steps between two fences, the maximum over the ranks.  The driver's 20 steps of 0.2 ms are
    # 4 ms of GPU time - one number without a spread, 4-5 % above what 100 steps give - so a block that short is
    # REPEATED (the same args.steps steps between the same fences) until MIN_TIMED_MS are covered, and the line reports the MEDIAN
    # block with the fastest and the slowest next to it.  Every rank sees the same max-reduced times: the same decision everywhere.
    block_s, local_block_s = [], []
    while True:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, res = step()
        fence()
        el_local = time.perf_counter() - t0
        el_max = el_local
        if world > 1:
            el = torch.tensor([el_local], dtype=torch.float64, device=device)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            el_max = float(el.item())
        block_s.append(el_max)
        local_block_s.append(el_local)
        if sum(block_s) * 1e3 >= args.min_timed_ms or len(block_s) >= MAX_TIMED_BLOCKS:
            break
    elapsed = float(np.median(block_s))                  # seconds per block of args.steps steps
    local_elapsed = float(np.median(local_block_s))

    sweep_ms, sweep_launches = pl.kernel_ms("dp_sweep"), pl.kernel_launches("dp_sweep")
    sweep_samples = pl.kernel_samples("dp_sweep") * 1e3 if sweep_launches > 0 else None
    # ---- what a first multi-GPU run needs to explain itself (N > 1, or the N > 1 code forced onto one GPU) -------------
    diag = None
    if gather_path:
        sg.drain()
        gms = sg.gather_ms()
        per_rank = [elapsed / args.steps * 1e3]
        if world > 1:
            mine = torch.tensor([local_elapsed / args.steps * 1e3], dtype=torch.float64, device=device)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [float(x.item()) for x in allr]
        # the same K steps WITHOUT pack and gather, between the same fences: what the exchange costs the step
        fence()
        pl.set_timing(False)
        with_gather[0] = False
        for _ in range(max(2, 2 * in_flight)):
            step()
        fence()
        nogs = []
        for _ in range(len(block_s)):                   # as many blocks as the timed region took, the median again
            n0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            nog = time.perf_counter() - n0
            if world > 1:
                el2 = torch.tensor([nog], dtype=torch.float64, device=device)
                dist.all_reduce(el2, op=dist.ReduceOp.MAX)
                nog = float(el2.item())
            nogs.append(nog)
        nog = float(np.median(nogs))
        with_gather[0] = True
        ms_with, ms_without = elapsed / args.steps * 1e3, nog / args.steps * 1e3
        diag = {"backend": (dist.get_backend() if (world > 1 or pg_one) else "none (one process)"),
                "world_size_seen_by_the_process_group": (dist.get_world_size() if (world > 1 or pg_one) else 1),
                "ms_per_step_per_rank": [round(v, 4) for v in per_rank],
                "ms_per_step_min_max_over_ranks": [round(min(per_rank), 4), round(max(per_rank), 4)],
                # the destination rank takes in (N - 1) blocks per step on top of its own shard: what that costs it
                "rank0_extra_ms_over_the_slowest_other_rank": (round(per_rank[0] - max(per_rank[1:]), 4) if len(per_rank) > 1 else None),
                "ms_per_step_without_pack_and_gather": round(ms_without, 4),
                "gather_ms_on_its_stream": None if gms is None else {"mean": round(gms[0], 4), "min": round(gms[1], 4), "max": round(gms[2], 4), "count": gms[3]},
                "gather_hidden_behind_compute_frac": None}
        if gms is not None and gms[0] > 0:
            exposed = max(0.0, ms_with - ms_without)
            diag["gather_hidden_behind_compute_frac"] = round(max(0.0, 1.0 - exposed / gms[0]), 3)
    # The other pipeline form, as a second, separately reported measurement (same steps, same fences; N = 1 only): the
    # headline is taken in the form that keeps the sweep's bandwidth, lane mode trades it for throughput.
    alt = None
    if world == 1 and not gather_path and args.alt_pipeline != "none" and args.alt_pipeline != args.pipeline:
        fence()
        pl.set_timing(False)
        amode = 0 if args.alt_pipeline == "off" else (1 if args.alt_pipeline == "staged" else int(args.alt_pipeline))
        pl.set_pipeline(amode)
        for _ in range(max(args.warmup, 2 * pl.in_flight)):
            out, res = step()
        fence()
        pl.set_timing(True, only="dp_sweep")
        a0 = time.perf_counter()
        for _ in range(args.steps):
            out, res = step()
        fence()
        a_el = time.perf_counter() - a0
        a_sweep = pl.kernel_ms("dp_sweep")
        alt = {"pipeline": "off" if amode == 0 else "staged" if amode == 1 else f"{amode} lanes", "batches_in_flight": pl.in_flight,
               "all_scenes_cycles_per_s": round(total * args.steps / a_el, 1), "unit": "planning cycles/s",
               "value": round(total * args.steps / a_el, 1),
               "ms_per_step": round(a_el / args.steps * 1e3, 4), "sweep_mean_launch_us": round(a_sweep * 1e3, 2)}
        pl.set_timing(False)
    # The shader clock the chip holds in THIS schedule: a few more steps of it with the edge-cost kernel's clock probe on (two
    # counter reads per wavefront; untimed) - what roofline_step prices the step's vector-issue capacity at
    edge_clock_mhz = None
    probe_steps = 0
    if args.dp_mode == "two_kernel" and pl.get_option("edge_form") == 0:
        fence()
        pl.set_timing(False)
        pl.set_option("edge_clock_probe", 1)
        clocks = []
        for _ in range(3):
            for _ in range(2 * max(in_flight, 1)):
                out, res = step()
                probe_steps += 1
            fence()
            c = pl.edge_clock_mhz()
            if c:
                clocks.append(c)
        pl.set_option("edge_clock_probe", 0)
        edge_clock_mhz = float(np.median(clocks)) if clocks else None
    # Per-kernel breakdown: a separate diagnostic pass AFTER the timed region, every kernel bracketed, one batch in
    # flight (the durations of overlapping kernels would not add up to anything)
    fence()
    pl.set_pipeline(False)
    for _ in range(2):
        out, res = step()
    fence()
    pl.set_timing(True)
    for _ in range(min(args.steps, 5)):
        out, res = step()
    fence()
    kernels = {}
    for name in ("project", "dp_edge", "dp_sweep", "dp_fused", "dp_enrich", "path_qp", "to_cartesian", "heading", "st_graph",
                 "speed_dp"):
        ms = pl.kernel_ms(name)
        if ms >= 0:
            kernels[name] = round(ms, 6)
    pl.set_timing(False)

    # ---- secondary legs (N = 1, the default workload only): other workloads, observed by the same command, after the headline
    # and its diagnostic pass; each is a short measurement of its own and never touches the headline's numbers
    legs = {}
    if (world == 1 and not gather_path and not wide and not args.no_legs and args.scenes_per_gpu in (0, 4096) and args.total_scenes in (0, 4096)
            and args.dp_mode == "two_kernel" and (auto_choice is not None or args.pipeline == DEFAULT_PIPELINE) and args.scene_dist == "corridor"):
        # (a) the staged pipeline of rounds 2-5 (two batches: the front stage of one beside the back stage of the other; the sweep
        # overlaps only the previous batch's Cartesian tail), with the library's default options and with the sweep held back
        # behind the previous batch's path QP (EMP_OPT_SWEEP_EXCLUSIVE = 2).  Each in a process of its own: which hardware queue
        # a stream lands on depends on the streams the process created before it, and a staged pipeline set up behind three lane
        # streams finds its two stages on one queue (0.43 ms a step instead of 0.22) - a fresh process is what a staged user has.
        legs["staged_leg"] = staged_subprocess_leg(args.steps, {})
        legs["exclusive_sweep_leg"] = staged_subprocess_leg(args.steps, {"sweep_exclusive": 2})
        legs["rccl_gather_leg"] = rccl_gather_subprocess_leg(max(args.steps, 20))
        legs["gather_path_leg"] = gather_path_leg(pl, torch, emp_dist, S.CFG2, 4096, max(args.steps, 20), device, scene_kw)
        legs["dram_leg"] = secondary_leg(pl, torch, S.CFG2, 32768, 10, 12, device, scene_kw)
        legs["cfg5_leg"] = secondary_leg(pl, torch, S.CFG5, 4096, 12, 6, device, scene_kw, speed=True)   # 55 ms timed: like the headline's blocks
        legs["latency_leg"] = latency_leg(pl, torch, device, scene_kw=scene_kw)
        # SURVEY 8(d)'s own geometry: arcs of radius 150-1000 m, with its slalom layout (the reference refuses nearly every
        # scene there: status paths) and with the corridor layout (mostly plannable)
        tight = dict(scene_kw, radius_range=S.SURVEY_ARCS)
        legs["survey_leg"] = secondary_leg(pl, torch, S.CFG2, 4096, 10, 20, device, dict(tight, dist="survey"))
        legs["tight_corridor_leg"] = secondary_leg(pl, torch, S.CFG2, 4096, 10, 20, device, dict(tight, dist="corridor"))
        legs["host_io_leg"] = host_io_leg(pl, torch, S.CFG2, 4096, 60, scene_kw)
        # the reference's OWN call shape - one request, Python lists in, tuples out - through a real Pipe and as the explicit function
        # sequence of its planning loop (bench_dropin.py, a process of its own: the drop-in modules own a context per process)
        legs["dropin_leg"] = dropin_subprocess_leg(200)

    # outcome statistics (sanity: the work was really done): the planned fraction of every input batch the steps rotate through
    # (one more pass each, untimed); every rank looks at its own shard and the fractions are averaged over the ranks
    fence()
    pl.set_timing(False)
    oks = []
    for ib in range(n_in):
        with torch.cuda.stream(ts):
            r_ib = pl.plan_cycle(p, q, sp, max_pts=M, mode=mode, **inputs_ring[ib])
        fence()
        oks.append(float(((r_ib.status.cpu().numpy() & ~1) == 0).mean()))
    ok_frac = float(np.mean(oks))
    if world > 1:
        okt = torch.tensor([ok_frac * count, float(count)], dtype=torch.float64, device=device)
        dist.all_reduce(okt)
        ok_frac = float(okt[0].item() / okt[1].item())
    gathered_ok = None
    if gather_path and out is not None:                    # on the destination rank: the gathered records are complete
        gst = sg.unpack(out)["status"].cpu().numpy()
        gathered_ok = bool(gst.shape[0] == total)
    if sg is not None:
        sg.drain()

    if rank == 0:
        E = cfg.row + (cfg.col - 1) * cfg.row ** 2
        bytes_dp = (8 * E + 4 * cfg.row * cfg.col + 4 * cfg.col) * count          # SURVEY.md 8(d), per launch
        roof = None
        if sweep_ms > 0:
            ach = bytes_dp / (sweep_ms * 1e-3) / 1e9
            prof = committed_profile("dp_sweep_traffic", config=cfg.name, scenes_per_gpu=count)
            tprof = committed_profile("dp_sweep_trace", config=cfg.name, scenes_per_gpu=count, pipeline=("off" if pmode == 0 else "staged" if pmode == 1 else f"{pmode} lanes"))
            roof = {"kernel": "dp_sweep_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": prof["hbm_bytes_per_launch"] if prof else None,
                    "traffic_source": prof["source"] if prof else None,
                    "algorithmic_bytes_per_launch": bytes_dp, "launches_timed": sweep_launches,
                    "mean_launch_us": round(sweep_ms * 1e3, 2),
                    "launch_us_min_median_max": [round(float(v), 2) for v in (sweep_samples.min(), np.median(sweep_samples), sweep_samples.max())]
                    if sweep_samples is not None and len(sweep_samples) else None,
                    # the same three numbers from the COMMITTED rocprofv3 kernel trace of this command (another box, another day):
                    # what the files under profiles/ say next to what this run's events say
                    "committed_trace_launch_us_min_median_max": (tprof or {}).get("timed_launch_us_min_median_max"),
                    "committed_trace_source": (tprof or {}).get("source"),
                    # the same kernel with nothing beside it: the diagnostic pass after the timed region, one batch in flight
                    "frac_alone": (round(bytes_dp / (kernels["dp_sweep"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if "dp_sweep" in kernels else None),
                    "alone_mean_launch_us": (round(kernels["dp_sweep"] * 1e3, 2) if "dp_sweep" in kernels else None)}
        if alt and alt["sweep_mean_launch_us"] > 0:
            alt["sweep_roofline_frac"] = round(bytes_dp / (alt["sweep_mean_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        if roof is not None and pmode >= 2:
            # `frac` above is the kernel inside THIS schedule (lanes: two other batches' edge-cost kernels write their tensors
            # through the caches the sweep reads its own from); the same kernel in the schedules that leave it (nearly) alone,
            # measured by the legs of this very run, next to it - null when the legs did not run
            roof["frac_is"] = "the sweep inside the lane schedule; the kernel's own figures are frac_alone and the two below"
            roof["frac_in_the_staged_form"] = (legs.get("staged_leg") or {}).get("sweep_frac")
            roof["frac_held_back_in_the_staged_form"] = (legs.get("exclusive_sweep_leg") or {}).get("sweep_frac")
        # Secondary figures from the diagnostic pass (event-bracketed kernels; not part of the timed region):
        # the edge-cost kernel against the FP64 vector peak with SURVEY.md 8(d)'s ALGORITHMIC flop count next to what
        # the committed SQ counter profile says was executed (obstacles out of reach are skipped at run time and
        # 40-45 % of the lanes of an executed instruction are masked off), and the DP alone.
        extra = {}
        if "dp_edge" in kernels:
            flops = E * (40 + 10 * (36 + 7 * cfg.n_obs)) * count
            tf = flops / (kernels["dp_edge"] * 1e-3) / 1e12
            cprof = committed_profile("dp_edge_counters", config=cfg.name, scenes_per_gpu=count, scene_dist=args.scene_dist)
            # `frac` is what the committed SQ pass of this workload says: the fraction of the launch's FP64 issue slots that
            # carried work, busy x active lanes (null without a profile).  SURVEY 8(d)'s ALGORITHMIC flop count over the
            # duration reads 1.0 and more of the vector peak, because obstacles out of reach are skipped at run time: it stays
            # as a secondary key, it is not a roofline.
            e = {"kernel": "dp_edge_ring_kernel" if pl.get_option("edge_form") == 0 else "dp_edge_kernel", "bound": "fp64_valu_issue", "frac": None, "unit": "fraction of the FP64 issue slots doing work",
                 "mean_launch_us": round(kernels["dp_edge"] * 1e3, 2), "source": "diagnostic pass after the timed region",
                 "survey_8d_flop_count_per_launch": flops, "survey_8d_flop_count_over_duration_tflops": round(tf, 2),
                 "survey_8d_flop_count_note": "SURVEY 8(d)'s count prices every (edge, obstacle) pair; the kernel skips pairs out of reach "
                                              "exactly, so this is not a rate the vector pipe sustained and never a fraction of its peak"}
            if cprof:
                busy = cprof.get("valu_issue_busy_frac")
                if busy is None:            # round-2 entries: quad-cycles against the 2.4 GHz peak clock and this run's duration
                    busy = round(cprof["valu_busy_quad_cycles"] * 4 / (1024 * 2.4e3) / (kernels["dp_edge"] * 1e3), 3)
                e.update(frac=round(busy * cprof["lanes_active_frac"], 3), valu_issue_busy_frac=busy,
                         active_lane_frac=cprof["lanes_active_frac"], executed_wave_instructions_valu=cprof["insts_valu"],
                         executed_lane_ops=int(cprof["insts_valu"] * 64 * cprof["lanes_active_frac"]),
                         counters_source=cprof["source"])
            extra["roofline_dp_edge"] = e
        if all(k in kernels for k in ("dp_edge", "dp_sweep", "dp_enrich")):
            dp_ms = kernels["dp_edge"] + kernels["dp_sweep"] + kernels["dp_enrich"]
            extra["dp_only"] = {"value": round(count / (dp_ms * 1e-3), 1), "unit": "DP plans/s per GPU",
                                "source": "sum of the three DP kernels' mean durations in the diagnostic pass"}
        if wide and "speed_dp" in kernels:
            # The S-T speed DP is FP64-issue bound with no HBM traffic to speak of (0.8 KB per scene).  SURVEY.md 8(d)'s
            # algorithmic count, E_st (30 + 5 n_obs 45) flops per scene, prices every (sample, obstacle) pair, of which the
            # kernel costs only the 8 % within reach: like roofline_dp_edge's it says little.  The counter-backed figures are
            # what the kernel executed (committed SQ pass of the same kernel alone at this batch size).
            e_st = 40 + 15 * 40 * 40
            flops = e_st * (30 + 5 * 16 * 45) * count
            tf = flops / (kernels["speed_dp"] * 1e-3) / 1e12
            # `frac` is the counter-backed useful-issue fraction (null without a committed SQ pass of this workload); the
            # algorithmic count reads MORE than the vector peak because the kernel prunes most of what it prices - a
            # secondary key, never a fraction of anything
            e = {"kernel": "speed_dp_kernel", "bound": "fp64_valu_issue", "frac": None,
                 "unit": "fraction of the FP64 issue slots doing work",
                 "survey_8d_flop_count_per_launch": flops, "survey_8d_flop_count_over_duration_tflops": round(tf, 2),
                 "survey_8d_flop_count_note": "SURVEY 8(d)'s flop count with all 16 obstacle slots present; the kernel skips the 92 % of "
                                              "(sample, obstacle) pairs out of reach, so this is not a roofline",
                 "mean_launch_us": round(kernels["speed_dp"] * 1e3, 1), "speed_dps_per_s": round(count / (kernels["speed_dp"] * 1e-3), 1),
                 "edges_per_dp": e_st, "hbm_bytes_per_scene": 16 * 4 * 8 + 8 + 2 * 16 * 8 + 8,
                 "source": "diagnostic pass after the timed region (one batch in flight); tables stay in LDS (emp_st_kernels.h)"}
            cprof = committed_profile("speed_dp_counters", scenes_per_gpu=count, obstacle_slots=16)
            if cprof:
                e.update(executed_wave_instructions_valu=cprof["insts_valu"], active_lane_frac=cprof["lanes_active_frac"],
                         executed_lane_ops=int(cprof["insts_valu"] * 64 * cprof["lanes_active_frac"]),
                         valu_issue_busy_frac=cprof["valu_issue_busy_frac"],
                         frac=round(cprof["valu_issue_busy_frac"] * cprof["lanes_active_frac"], 3),
                         useful_issue_frac=round(cprof["valu_issue_busy_frac"] * cprof["lanes_active_frac"], 3),
                         mean_waves_per_simd=cprof["mean_waves_per_simd"], counters_source=cprof["source"])
            extra["roofline_speed_dp"] = e
        all_scenes_rate = total * args.steps / elapsed
        value = all_scenes_rate * ok_frac                  # what "planning cycles/sec" means: cycles planned to the end
        gather_note = ""
        if world > 1:
            gather_note = f"; + RCCL {'gather to rank 0' if args.gather == 'rank0' else 'all_gather'} of {args.records} result records"
        line = {
            "metric": ("planning cycles/sec (DP+QP, 40x9 S-L lattice, 8 obs)" if not wide else
                       "planning cycles/sec (DP+QP on the 120x21 S-L lattice, 16 obs, + S-T speed DP 40x16, 16 dynamic obstacles)"),
            "metric_note": "value counts the scenes PLANNED TO THE END: scenes per step x scenes_fully_planned_frac / ms_per_step.  "
                           "Every scene of the batch runs the whole cycle; those whose path QP turns out infeasible (walls and "
                           "blocked corridors the generator puts there on purpose) are refused with a status bit at the end of "
                           "it and are not counted, although they cost the same time: all_scenes_cycles_per_s = scenes per "
                           "step / ms_per_step is the rate at which scenes go through the pipeline",
            "value": round(value, 1),
            "value_definition": "fully_planned_cycles_per_s (since round 4; rounds 1-3 reported all_scenes_cycles_per_s as value: "
                                "compare rounds on all_scenes_cycles_per_s)",
            "fully_planned_cycles_per_s": round(value, 1),
            "all_scenes_cycles_per_s": round(all_scenes_rate, 1),
            "unit": "planning cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "rccl_world_size": (dist.get_world_size() if world > 1 else None),     # what the process group itself reports
            "process_group_backend": (dist.get_backend() if world > 1 else None),
            "untimed_steps_before_the_timed_region": args.warmup + settle,
            # (what a kernel trace of this command holds, in start order: the untimed steps, timed_blocks x steps timed steps, these -
            # the clock probe's - then the diagnostic pass (2 + min(steps, 5) steps) and one pass per input batch)
            "untimed_steps_between_the_timed_region_and_the_diagnostic_pass": probe_steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_is": f"the median of timed_blocks blocks of {args.steps} steps each (every block between two fences, max over ranks)",
            "ms_per_step_min_max": [round(min(block_s) / args.steps * 1e3, 4), round(max(block_s) / args.steps * 1e3, 4)],
            "timed_blocks": len(block_s), "timed_ms_total": round(sum(block_s) * 1e3, 2),
            "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[2]/[3]" if not wide else "BASELINE configs[4]")
                                   + ": full planning cycle per scene (projection, S-L DP, path QP, Frenet->Cartesian, "
                                     "smoothing QP, heading/kappa)" + (", then generate_st_graph + the S-T speed DP" if wide else "")
                                   + ", inputs resident in HBM" + gather_note,
                       "scenes_per_gpu": count, "total_scenes": total,
                       "scenes_per_rank": [emp_dist.shard_range(total, r, world)[1] for r in range(world)], "input_batches": n_in,
                       "input_batches_note": "step i plans resident batch i mod input_batches: different scenes every step",
                       "lattice": f"col={cfg.col} x row={cfg.row}",
                       "sample_s": cfg.sample_s, "sample_l": cfg.sample_l, "obstacles": cfg.n_obs,
                       "scene_dist": args.scene_dist, "start_ahead_m": args.start_ahead, "arc_radii_m": list(scene_kw["radius_range"]),
                       "ref_line_points": int(P), "dp_mode": args.dp_mode,
                       "batches_in_flight": in_flight, "pipeline": "off" if pmode == 0 else "staged" if pmode == 1 else f"{pmode} lanes",
                       "pipeline_chosen_by_the_library": auto_choice,
                       "parallelism": f"scenes sharded over {world} GPU(s), one process per GPU"},
            "roofline": roof,
            "roofline_step": (roofline_step(cfg, count, args.scene_dist, elapsed / args.steps * 1e3, speed=wide, measured_clock_mhz=edge_clock_mhz) if pmode != 0 else None),
            **extra,
            "kernels_ms": kernels,
            "alt_pipeline": alt,
            "scenes_fully_planned_frac": round(ok_frac, 4),
            "parity_sweeps_committed": committed_parity_sweeps() if (world == 1 and not args.no_legs) else None,
            "options": {**{k: pl.get_option(k) for k in ("sweep_exclusive", "edge_after_enrich", "lane_edge_order", "path_qp_form", "edge_form")}, **options},
            **legs,
        }
        if gather_path:
            line["gather"] = {"mode": args.gather, "records": args.records, "doubles_per_scene": sg.width,
                              "bytes_sent_per_rank_and_step": sg.bytes_per_rank_and_step(count),
                              "bytes_received_by_rank0_per_step": sg.bytes_per_rank_and_step(count) * (world - 1)
                              if args.gather == "rank0" else sg.bytes_per_rank_and_step(count) * (world - 1),
                              "records_complete_on_rank0": gathered_ok, **(diag or {})}
        elif world > 1 or args.gather == "none":
            line["gather"] = {"mode": "none", "note": "no pack, no gather: compute scaling only",
                              "world_size_seen_by_the_process_group": (dist.get_world_size() if world > 1 else 1)}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample if not wide else 2, 0, scene_kw)
            workers = usable_cores(64) if args.cpu_pool < 0 else args.cpu_pool
            if workers > 0 and not wide:
                try:
                    line["cpu_baseline_pool"] = cpu_baseline_pool("CFG2", workers, args.cpu_pool_scenes, scene_kw)
                except Exception as exc:                                       # informational: never fails the bench
                    line["cpu_baseline_pool"] = {"error": f"{type(exc).__name__}: {exc}"}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    pl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    elif pg_one:
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
