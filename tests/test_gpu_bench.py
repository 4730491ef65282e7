"""bench.py end to end on the GPU, short: the one-GPU line and the per-step code of an N > 1 rank (record packing on the
result stream, gather on its own stream - the gather itself is the identity without a process group)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--no-cpu-baseline",
           *(() if "--legs" in extra else ("--no-legs",)), *[e for e in extra if e != "--legs"]]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("extra", [(), ("--no-pipeline",), ("--force-gather-path",), ("--force-gather-path", "--no-pipeline"),
                                   ("--pipeline", "staged"), ("--pipeline", "2", "--force-gather-path"), ("--alt-pipeline", "staged")])
def test_bench_line(extra):
    d = _run(*extra)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    # a four-step block is under a millisecond: it is repeated until 50 ms are covered and the line carries the median and the spread
    assert d["timed_blocks"] > 1 and d["timed_ms_total"] >= 50.0 or d["timed_blocks"] == 64
    assert d["ms_per_step_min_max"][0] <= d["ms_per_step"] <= d["ms_per_step_min_max"][1]
    pipe = extra[extra.index("--pipeline") + 1] if "--pipeline" in extra else "3"       # bench.DEFAULT_PIPELINE
    lanes = 0 if pipe == "staged" else int(pipe)
    # (lane mode overlaps the sweep with other batches' edge kernels: it takes two to three times as long there)
    assert d["value"] > 1e6 and (0.05 if lanes else 0.3) < d["roofline"]["frac"] < 1.0 and d["roofline"]["bound"] == "hbm"
    assert 0.8 < d["scenes_fully_planned_frac"] < 0.95
    # `value` is the fully planned rate: scenes per step x planned fraction / step time
    assert abs(d["value"] - d["all_scenes_cycles_per_s"] * d["scenes_fully_planned_frac"]) <= 1e-3 * d["value"]
    assert abs(d["all_scenes_cycles_per_s"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 2e-3 * d["all_scenes_cycles_per_s"]
    # (with the gather path the diagnostic pass still packs and gathers every step: the "alone" sweep has the copy kernels beside it)
    assert d["roofline"]["frac_alone"] > (0.15 if "--force-gather-path" in extra else 0.3) and len(d["roofline"]["launch_us_min_median_max"]) == 3
    assert d["config"]["batches_in_flight"] == (1 if "--no-pipeline" in extra else lanes or 2)
    assert d["config"]["pipeline"] == ("off" if "--no-pipeline" in extra else f"{lanes} lanes" if lanes else "staged")
    if "--pipeline" not in extra and "--no-pipeline" not in extra:      # the default: the library chose, on the 12 queues bench.py asks for
        c = d["config"]["pipeline_chosen_by_the_library"]
        assert c["form"] == "3 lanes" and c["hardware_queues"] == 12 and c["streams_beside_the_lanes"] == (4 if "--force-gather-path" in extra else 2)
    else:
        assert d["config"]["pipeline_chosen_by_the_library"] is None
    if "--alt-pipeline" in extra:      # the second timed region, in the staged form
        alt = d["alt_pipeline"]
        assert alt["pipeline"] == "staged" and alt["all_scenes_cycles_per_s"] > 1e6 and 0.05 < alt["sweep_roofline_frac"] < 1.0
    else:
        assert d["alt_pipeline"] is None


@pytest.mark.parametrize("extra,metric_part", [
    (("--latency",), "latency"),
    (("--config", "cfg5", "--scenes-per-gpu", "96"), "120x21"),
    (("--scene-dist", "survey", "--scenes-per-gpu", "512"), "40x9"),
    (("--scene-dist", "worst", "--scenes-per-gpu", "512"), "40x9"),
    (("--dp-mode", "fused", "--scenes-per-gpu", "512"), "40x9"),
    (("--force-gather-path", "--records", "trajectory", "--gather", "all", "--scenes-per-gpu", "512"), "40x9"),
], ids=["latency", "cfg5", "survey", "worst", "fused", "gather_all_trajectory"])
def test_other_bench_lines(extra, metric_part):
    """The other lines bench.py can print (BASELINE configs[1] and [4], the survey's and the worst-case scene layouts, the
    fused DP, the all-gather / trajectory-only exchange): each runs and carries the contract's keys."""
    d = _run(*extra)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert metric_part in d["metric"] and d["value"] > 0 and d["dtype"] == "f64" and d["vs_baseline"] is None
    if "--latency" in extra:
        assert d["higher_is_better"] is False and d["roofline"] is None and 0.05 < d["value"] < 5.0
        return
    assert d["higher_is_better"] is True and "all_scenes_cycles_per_s" in d and d["value"] <= d["all_scenes_cycles_per_s"]
    if "fused" in extra:
        assert d["roofline"] is None and "dp_fused" in d["kernels_ms"]
    else:
        assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0.01
    if "cfg5" in extra:
        assert "speed_dp" in d["kernels_ms"] and d["config"]["lattice"] == "col=120 x row=21"
    if "worst" in extra:
        assert d["scenes_fully_planned_frac"] > 0.9
    if "--gather" in extra:
        assert d["gather"]["mode"] == "all" and d["gather"]["doubles_per_scene"] == 94 and d["gather"]["records_complete_on_rank0"]


def test_default_run_carries_the_secondary_legs():
    """The driver's one command (no flags beyond steps / warm-up) also observes the other workloads: the sweep held back behind
    the previous batch's path QP, 32768 scenes (edge tensor from HBM), BASELINE configs[4] on 4096 scenes, configs[1], SURVEY
    8(d)'s arc radii with both obstacle layouts, the host path (NumPy in and out) and the N > 1 per-step code - each a short
    leg behind the headline, none of which may fail or change the headline's keys."""
    d = _run("--legs")
    assert d["options"]["sweep_exclusive"] == 0 and d["options"]["edge_form"] == 0          # the library's defaults: the fastest step
    assert d["value_definition"].startswith("fully_planned") and d["fully_planned_cycles_per_s"] == d["value"]
    assert d["config"]["start_ahead_m"] == 2.7 and d["config"]["arc_radii_m"] == [1500.0, 6000.0]
    o = d["exclusive_sweep_leg"]
    assert "error" not in o and o["options"] == {"sweep_exclusive": 2} and o["all_scenes_cycles_per_s"] > 1e6 and 0.3 < o["sweep_frac"] < 1.0
    # the headline runs three lanes; the staged form of rounds 2-5 is a leg in a process of its own, its sweep nearly alone
    sl = d["staged_leg"]
    assert d["config"]["pipeline"] == "3 lanes" and d["roofline_step"] is not None
    assert "error" not in sl and sl["batches_in_flight"] == 2 and sl["options"] == {} and sl["all_scenes_cycles_per_s"] > 1e6
    assert 0.3 < sl["sweep_frac"] < 1.0 and sl["sweep_frac"] > d["roofline"]["frac"]
    # SURVEY 8(d)'s own geometry, with its slalom layout (the reference refuses nearly everything) and with the corridor layout
    sv, tc = d["survey_leg"], d["tight_corridor_leg"]
    assert "error" not in sv and "error" not in tc, (sv, tc)
    assert sv["scenes"]["radius_range"] == [150.0, 1000.0] and sv["scenes"]["dist"] == "survey" and tc["scenes"]["dist"] == "corridor"
    assert sv["all_scenes_cycles_per_s"] > 1e6 and 0.0 < sv["scenes_fully_planned_frac"] < 0.5 < tc["scenes_fully_planned_frac"] < 0.95
    # the host path: NumPy in and out through the page-locked ring beats the synchronous pageable path, with the same results
    h = d["host_io_leg"]
    assert "error" not in h, h
    assert h["ring_outputs_equal_the_synchronous_path"] is True and h["bytes_in_per_step"] > 8e6 and h["bytes_out_per_step"] > 1e7
    # (a four-step run measures the synchronous path cold - three calls - so only the order of the two is asserted here; the
    # figures themselves are bench.py's default run: profiles/r05_bench_default.json)
    assert h["host_ring"]["all_scenes_cycles_per_s"] > h["synchronous_pageable_path"]["all_scenes_cycles_per_s"] > 1e5
    assert 0.05 < h["one_scene_host_latency_ms"] < 5.0
    # the N > 1 per-step code with a real (one-rank) RCCL process group, in a process of its own
    rg = d["rccl_gather_leg"]
    assert "error" not in rg, rg
    assert rg["backend"] == "nccl" and rg["world_size_seen_by_the_process_group"] == 1 and rg["records_complete_on_rank0"] is True
    assert rg["doubles_per_scene"] == 179 and rg["ms_per_step"] > 0 and rg["gather_ms_on_its_stream"]["count"] > 0
    # the N > 1 per-step code on this one GPU
    gp = d["gather_path_leg"]
    assert "error" not in gp, gp
    assert gp["records_complete"] is True and gp["doubles_per_scene"] == 179 and gp["bytes_sent_per_rank_and_step"] == 4096 * 179 * 8
    assert gp["ms_per_step"] > 0 and gp["ms_per_step_without_pack_and_gather"] > 0 and gp["pack_kernel_us"] > 0 and 0.05 < gp["sweep_frac"] < 1.0
    g = d["dram_leg"]
    assert "error" not in g, g
    # (three lanes: the sweep streams its 883 MB from DRAM beside two other batches' edge kernels; alone it keeps its bandwidth)
    assert g["ms_per_step"] > 4 * d["ms_per_step"] and 0.15 < g["sweep"]["frac"] < 1.0 and g["sweep"]["frac_alone"] > 0.4
    assert g["sweep"]["algorithmic_bytes_per_launch"] == 26944 * 32768
    assert 0.8 < g["scenes_fully_planned_frac"] < 0.95
    c = d["cfg5_leg"]
    assert "error" not in c, c
    assert c["speed_dp_us"] > 100 and 0.15 < c["sweep"]["frac"] < 1.0 and c["all_scenes_cycles_per_s"] > 1e5
    # the reference's own call shape: a request down a real Pipe, the function sequence of its planning loop, plan_requests in-process
    dr = d["dropin_leg"]
    assert "error" not in dr, dr
    assert dr["pipe_motion_planning"]["requests"] == 200 and 0.05 < dr["pipe_motion_planning"]["median_ms"] < 5.0
    assert dr["function_sequence"]["planned"] > 150 and 0.2 < dr["function_sequence"]["median_ms"] < 10.0
    assert dr["pipe_vs_function_sequence"]["trajectories_compared"] > 150 and dr["pipe_vs_function_sequence"]["max_abs_xy_difference_m"] < 1e-6
    assert dr["pipe_motion_planning"]["median_ms"] < dr["function_sequence"]["median_ms"]
    lat = d["latency_leg"]
    assert "error" not in lat and 0.05 < lat["ms_per_cycle_median"] < 5.0 and lat["calls"] == 50
    g = lat["as_one_hipgraph"]
    assert g["results_equal_the_plain_calls"] is True and 0.05 < g["ms_per_cycle_median"] < 5.0


def test_bench_two_ranks_without_gather_separates_compute_scaling():
    """--gather none: the same two ranks with no pack and no gather at all - the line that tells compute scaling from
    the cost of the exchange."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--settle-steps", "4", "--scenes-per-gpu", "1024", "--no-cpu-baseline", "--gather", "none"]
    env = dict(os.environ, EMP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["gather"]["mode"] == "none" and d["rccl_world_size"] == 2 and d["value"] > 1e4


@pytest.mark.parametrize("extra", [("--gather", "rank0"), ("--gather", "all", "--records", "trajectory"), ("--pipeline", "staged")],
                         ids=["gather_rank0", "all_gather_trajectory", "staged"])
def test_bench_two_ranks_on_one_gpu(extra):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), with both ranks on the
    one GPU and gloo instead of RCCL (EMP_BENCH_BACKEND: RCCL refuses two ranks on a device): shards, the per-step pack on
    the result stream, the gather on its own stream with the in-flight ring, barriers and the max-over-ranks time are the
    code the 2/4/8-GPU runs execute.  Rank 0 prints the one JSON line; the records it gathered are complete."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:      # a port nobody listens on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--settle-steps", "4", "--scenes-per-gpu", "1024", "--no-cpu-baseline", *extra]
    env = dict(os.environ, EMP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["total_scenes"] == 2048 and d["config"]["scenes_per_gpu"] == 1024
    assert d["value"] > 1e4 and d["gather"]["records_complete_on_rank0"] is True      # (gloo moves the records through the host)
    assert d["gather"]["mode"] == ("all" if "all" in extra else "rank0")
    # the self-diagnosis of a first multi-GPU run: what the process group reports, every rank's own step time, the gather's
    # duration on its stream, the step without pack and gather, and how much of the gather hides behind compute
    g = d["gather"]
    assert d["rccl_world_size"] == 2 and d["process_group_backend"] == "gloo" and g["world_size_seen_by_the_process_group"] == 2
    assert len(g["ms_per_step_per_rank"]) == 2 and g["ms_per_step_min_max_over_ranks"][0] <= g["ms_per_step_min_max_over_ranks"][1]
    assert g["ms_per_step_min_max_over_ranks"][1] <= d["ms_per_step"] * 1.0001
    assert g["ms_per_step_without_pack_and_gather"] > 0 and g["gather_ms_on_its_stream"]["count"] == 6 * d["timed_blocks"]
    assert g["gather_ms_on_its_stream"]["mean"] > 0 and 0.0 <= g["gather_hidden_behind_compute_frac"] <= 1.0
    assert 0.8 < d["scenes_fully_planned_frac"] < 0.95


def test_rccl_calls_of_the_multi_gpu_path_with_one_rank():
    """The only RCCL run a one-GPU box allows: tools/rccl_smoke.py under torch.distributed.run with ONE rank and the real
    "nccl" (= RCCL) backend issues the torch.distributed calls of bench.py's N > 1 path in the same forms - init with device_id,
    barrier, all_reduce MAX, all_gather, dist.gather into views of one tensor on a side stream behind the record-packing
    kernel of a staged planning step, all_gather_into_tensor - and checks the gathered records.  It cannot show scaling; it
    shows that RCCL / ProcessGroupNCCL on this stack accept the calls, their argument forms and the stream usage."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "rccl_smoke.py")]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(port))
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "backend nccl world 1" in out.stdout and "rccl smoke ok" in out.stdout


@pytest.mark.parametrize("size", [("--scenes-per-gpu", "1024"), ("--total-scenes", "2051")], ids=["weak", "strong_ragged"])
def test_bare_command_with_gpus_2_launches_two_ranks(size):
    """The BARE command - `python bench.py --gpus 2 ...`, no torch.distributed.run, no WORLD_SIZE - must yield a two-rank line:
    bench.py re-executes itself under torch.distributed.run (gloo here: both ranks share the one GPU).  With --total-scenes the
    total is fixed and sharded (BASELINE configs[3] as written: strong scaling), ragged shards included."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--settle-steps", "4",
           "--no-cpu-baseline", *size]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(EMP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout[-2000:]                         # stdout is rank 0's ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["gather"]["records_complete_on_rank0"] is True
    g = d["gather"]
    assert len(g["ms_per_step_per_rank"]) == 2 and g["rank0_extra_ms_over_the_slowest_other_rank"] is not None
    assert 0.0 <= g["gather_hidden_behind_compute_frac"] <= 1.0 and g["world_size_seen_by_the_process_group"] == 2
    if "--total-scenes" in size:
        assert d["scaling"] == "strong" and d["config"]["total_scenes"] == 2051 and d["config"]["scenes_per_rank"] == [1026, 1025]
    else:
        assert d["scaling"] == "weak" and d["config"]["total_scenes"] == 2048 and d["config"]["scenes_per_rank"] == [1024, 1024]
    assert d["value"] > 1e4 and 0.8 < d["scenes_fully_planned_frac"] < 0.95


def test_default_bench_takes_the_staged_form_when_the_process_has_four_hardware_queues():
    """`python bench.py` asks the library for the pipeline form (EMP_PIPELINE_AUTO).  With GPU_MAX_HW_QUEUES=4 already in the
    environment - bench.py does not override it - three lanes would share queues and run at 0.28 ms per step; the line must show
    the staged form, chosen by the library, and a step near the staged form's."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-legs"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, GPU_MAX_HW_QUEUES="4"))
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["config"]["pipeline"] == "staged" and d["config"]["batches_in_flight"] == 2
    assert d["config"]["pipeline_chosen_by_the_library"] == {"form": "staged", "hardware_queues": 4, "streams_beside_the_lanes": 2}
    assert d["ms_per_step"] < 0.26 and d["roofline"]["frac"] > 0.45       # (three lanes on four queues: 0.28 ms, sweep 0.45 by accident of sharing)
