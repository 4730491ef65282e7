"""bench_legs.py - the secondary measurements of bench.py's default run, and the step roofline.

Imported by bench.py only (kept beside it so that bench.py itself stays the contract: arguments, the timed region, the JSON
line, the CPU baselines).  Nothing here imports ``oracle/``: the CPU baselines are bench.py's.  Each leg is a short,
separately reported measurement behind the headline's timed region; a leg never raises - a failure is reported in the leg.

  secondary_leg        another workload through the headline's own procedure (32768 scenes, configs[4], SURVEY 8(d)'s geometry)
  latency_leg          BASELINE configs[1]: one scene per synchronous call
  host_io_leg          NumPy in, NumPy out through the page-locked ring (api.HostRing), PCIe included
  gather_path_leg      the per-step code of an N > 1 rank on one GPU (record packing + gather streams)
  roofline_step        the whole step against the chip's vector-issue capacity, from the committed SQ counter pass
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
LEG_PIPELINE = 3         # emp_set_pipeline mode of the secondary legs: three lanes, as the headline (bench.DEFAULT_PIPELINE)


def committed_profile(kind, **match):
    """An entry of profiles/counters.json (written by tools/summarize_profile.py / tools/summarize_sq.py from rocprofv3
    passes on the GPU box) whose workload keys equal `match`; None when no such profile is committed."""
    side = os.path.join(ROOT, "profiles", "counters.json")
    try:
        for e in json.load(open(side)).get(kind, []):
            if all(e.get(k) == v for k, v in match.items()):
                return e
    except Exception:
        pass
    return None


def committed_parity_sweeps(prefix="r06_parity_sweep_"):
    """What the COMMITTED large-sample parity runs of this build say (tools/parity_sweep.py on a GPU box, GPU against the oracles on
    its host cores; files profiles/<prefix>*.json): scenes compared and how many differ, next to the headline they vouch for -
    including the batch whose planning starts sit ON reference-line nodes, where `s_map[idx + 1] < s` (reference path_planning.py:63)
    is decided by the last bit.  Not re-run here: a sweep is minutes of host-core time."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", prefix + "*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        e = {"source": os.path.relpath(path, ROOT)}
        dp, cy = d.get("dp"), d.get("cycle")
        if dp:
            e["dp_scenes_bit_for_bit_against_the_exact_oracle"] = dp.get("scenes")
            e["dp_scenes_mismatching"] = int(sum((dp.get("mismatching") or {}).values()))
        if cy:
            e.update(cycles=cy.get("scenes"), cycles_planned_and_compared=cy.get("fully_planned_and_compared"),
                     outcome_mismatch=cy.get("outcome_mismatch"), start_ahead_m=cy.get("start_ahead"), geometry=cy.get("geometry"),
                     worst_error_over_the_survey_rule=cy.get("worst_error_over_survey_rule"),
                     scenes_with_the_start_on_a_node=cy.get("scenes_with_the_start_on_a_node"),
                     tie_scenes_beyond_tolerance=len(cy.get("tie_scenes_beyond_tolerance") or []),
                     tie_scenes_not_explained_by_the_flipped_branch=cy.get("tie_scenes_not_explained_by_the_flipped_branch"))
        for part in ("speed_dp", "front_end"):
            if isinstance(d.get(part), dict):
                e[part] = {k: v for k, v in d[part].items() if isinstance(v, (int, float, str)) and k != "seconds"}
        out[os.path.basename(path)[len(prefix):-5]] = e
    return out or None


def _device_inputs(torch, S, cfg, seeds, device, scene_kw=None):
    batch = S.make_batch(seeds, cfg, **(scene_kw or {}))
    P = batch.ref.shape[1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return dict(ref_line=t(batch.ref), n_ref=t(np.full(len(batch.seeds), P, np.int32)), origin_xy=t(batch.origin_xy),
                start_xy=t(batch.start_xy), start_v=t(batch.start_v), start_a=t(batch.start_a),
                obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))


def secondary_leg(pl, torch, cfg, scenes, steps, untimed, device, scene_kw=None, speed=False, options=None):
    """A short, separately reported measurement of ANOTHER workload inside the default run (the driver's one command then
    observes it too): `untimed` steps, a fence, `steps` timed steps with the sweep bracketed by events, a fence - the
    headline's own procedure - then three steps with every kernel bracketed, one batch in flight, for the kernel table.
    Staged pipeline with the library's default options.  Never raises: a failure is reported in the leg."""
    from emplanner_carla_amd import _lib as L
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import dp_params_from_cfg, max_path_points, qp_params, smooth_params, speed_dp_params
    t_leg = time.perf_counter()
    saved = {}
    try:
        for k, v in (options or {}).items():
            saved[k] = pl.get_option(k)
            pl.set_option(k, v)
        inputs = _device_inputs(torch, S, cfg, range(scenes), device, scene_kw)
        st_inputs = None
        if speed:
            dyn = S.make_dynamic_batch(range(scenes), 16)
            st_inputs = [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in dyn[:4]], torch.from_numpy(np.ascontiguousarray(dyn[4])).to(device)
        p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
        sdp, M = speed_dp_params(), max_path_points(p)
        pl.set_timing(False)
        pl.set_pipeline(LEG_PIPELINE)
        ts = pl.torch_stream()

        def step():
            with torch.cuda.stream(ts):
                res = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
                if speed:
                    pl.set_fence(False)
                    sets = pl.st_graph(*st_inputs[0])
                    pl.speed_dp(sdp, *sets, st_inputs[1], tables=False)
                    pl.set_fence(True)
            return res

        def fence():
            pl.synchronize()
            torch.cuda.synchronize()

        for _ in range(untimed):
            step()
        fence()
        pl.set_timing(True, only="dp_sweep")
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        fence()
        el = time.perf_counter() - t0
        sweep_ms = pl.kernel_ms("dp_sweep")
        pl.set_timing(False)
        pl.set_pipeline(0)
        step()
        fence()
        pl.set_timing(True)
        for _ in range(3):
            step()
        fence()
        kernels = {n: round(pl.kernel_ms(n), 6) for n in ("project", "dp_edge", "dp_sweep", "dp_enrich", "path_qp", "to_cartesian",
                                                          "st_graph", "speed_dp") if pl.kernel_ms(n) >= 0}
        pl.set_timing(False)
        ok = float(((res.status.cpu().numpy() & ~1) == 0).mean())
        E = cfg.row + (cfg.col - 1) * cfg.row ** 2
        bytes_dp = (8 * E + 4 * cfg.row * cfg.col + 4 * cfg.col) * scenes
        rate = scenes * steps / el
        out = {"workload": f"{scenes} scenes, lattice col={cfg.col} x row={cfg.row}, {cfg.n_obs} obstacles"
                           + (", + generate_st_graph and the S-T speed DP (40x16 grid, 16 dynamic-obstacle slots)" if speed else "")
                           + "; full planning cycle, inputs resident in HBM, three lanes",
               "steps": steps, "untimed_steps": untimed, "ms_per_step": round(el / steps * 1e3, 4),
               "fully_planned_cycles_per_s": round(rate * ok, 1), "all_scenes_cycles_per_s": round(rate, 1),
               "scenes_fully_planned_frac": round(ok, 4),
               "sweep": {"mean_launch_us": round(sweep_ms * 1e3, 2), "algorithmic_bytes_per_launch": bytes_dp,
                         "achieved_gbs": round(bytes_dp / (sweep_ms * 1e-3) / 1e9, 1),
                         "frac": round(bytes_dp / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_alone": (round(bytes_dp / (kernels["dp_sweep"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if "dp_sweep" in kernels else None)},
               "kernels_ms_one_batch_in_flight": kernels}
        if speed and "speed_dp" in kernels:
            out["speed_dp_us"] = round(kernels["speed_dp"] * 1e3, 1)
        if options:
            out["options"] = dict(options)
        if scene_kw:
            out["scenes"] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in scene_kw.items()}
        out["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
        return out
    except Exception as exc:                                    # a secondary leg never costs the headline
        try:
            pl.set_timing(False)
            pl.set_pipeline(0)
        except Exception:
            pass
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    finally:
        for k, v in saved.items():
            try:
                pl.set_option(k, v)
            except Exception:
                pass


def latency_leg(pl, torch, device, calls=50, scene_kw=None):
    """BASELINE configs[1] inside the default run: ONE scene on the 40x9 lattice, one synchronous call per cycle."""
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import dp_params_from_cfg, max_path_points, qp_params, smooth_params
    t_leg = time.perf_counter()
    try:
        cfg = S.CFG2
        dev = _device_inputs(torch, S, cfg, [7], device, scene_kw)
        p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
        M = max_path_points(p)
        pl.set_timing(False)
        pl.set_pipeline(0)
        for _ in range(10):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **dev)
            pl.synchronize()
        lat = []
        for _ in range(calls):
            t0 = time.perf_counter()
            r = pl.plan_cycle(p, q, sp, max_pts=M, **dev)
            pl.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.sort(np.asarray(lat))
        # the same calls as one hipGraph (EMP_OPT_CYCLE_GRAPH): same inputs' memory, the first result's arrays written again
        graph = None
        try:
            pl.set_option("cycle_graph", 1)
            ref = {k: getattr(r, k).clone() for k in ("traj", "traj_len", "status", "path_l", "dp_rows")}
            for _ in range(10):
                r = pl.plan_cycle(p, q, sp, max_pts=M, out=r, **dev)
                pl.synchronize()
            glat = []
            for _ in range(calls):
                t0 = time.perf_counter()
                r = pl.plan_cycle(p, q, sp, max_pts=M, out=r, **dev)
                pl.synchronize()
                glat.append((time.perf_counter() - t0) * 1e3)
            glat = np.sort(np.asarray(glat))
            same = all(bool(torch.equal(getattr(r, k), v)) for k, v in ref.items())
            graph = {"ms_per_cycle_mean": round(float(glat.mean()), 4), "ms_per_cycle_median": round(float(np.median(glat)), 4),
                     "ms_per_cycle_p95": round(float(glat[int(0.95 * (calls - 1))]), 4), "results_equal_the_plain_calls": same}
            # ... and with the path-QP kernel form that is shorter on a single scene's critical path (EMP_OPT_PATH_QP_FORM = 1:
            # result-affecting at the 2e-9 level, hence an option and never chosen from the batch size)
            pl.set_option("path_qp_form", 1)
            for _ in range(10):
                r = pl.plan_cycle(p, q, sp, max_pts=M, out=r, **dev)
                pl.synchronize()
            flat = []
            for _ in range(calls):
                t0 = time.perf_counter()
                r = pl.plan_cycle(p, q, sp, max_pts=M, out=r, **dev)
                pl.synchronize()
                flat.append((time.perf_counter() - t0) * 1e3)
            worst = float((r.traj - ref["traj"]).abs().max().item())
            graph["with_path_qp_form_1"] = {"ms_per_cycle_median": round(float(np.median(flat)), 4),
                                            "max_abs_trajectory_difference_to_the_default_form": worst}
        finally:
            pl.set_option("cycle_graph", 0)
            pl.set_option("path_qp_form", 0)
        return {"workload": "BASELINE configs[1]: one scene, 40x9 lattice, 8 obstacles, one synchronous call per cycle, inputs resident in HBM",
                "calls": calls, "ms_per_cycle_mean": round(float(lat.mean()), 4), "ms_per_cycle_median": round(float(np.median(lat)), 4),
                "ms_per_cycle_p95": round(float(lat[int(0.95 * (calls - 1))]), 4), "scene_status": int(r.status.cpu().numpy()[0]),
                "as_one_hipgraph": graph,
                "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}


def host_io_leg(pl, torch, cfg, scenes, steps, scene_kw=None):
    """The host path inside the default run: NumPy arrays in, NumPy arrays out, PCIe included (reference boundary: Python
    lists per request, test_9.py:92-96, 220, 390-395).  Ordinary (pageable) input arrays are copied into the page-locked
    ring (api.HostRing) with np.copyto, the cycle runs on the pipeline (three lanes) with its inputs on a copy stream and its outputs
    on a stream of their own, the results are read from the ring's page-locked output arrays.  Also the synchronous
    EMP_HOST path of rounds 1-4 (pageable arrays staged by the library, one call at a time), and one scene per call."""
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import dp_params_from_cfg, max_path_points, qp_params, smooth_params
    t_leg = time.perf_counter()
    ring = None
    try:
        batch = S.make_batch(range(scenes), cfg, **(scene_kw or {}))
        P = batch.ref.shape[1]
        c = np.ascontiguousarray
        host = dict(ref_line=c(batch.ref), n_ref=np.full(scenes, P, np.int32), origin_xy=c(batch.origin_xy), start_xy=c(batch.start_xy),
                    start_v=c(batch.start_v), start_a=c(batch.start_a), obs_xy=c(batch.obs_xy), n_obs=c(batch.n_obs))
        p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
        M = max_path_points(p)
        pl.set_timing(False)
        pl.set_pipeline(0)
        for _ in range(2):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **host)
        t0 = time.perf_counter()
        n_sync = max(3, steps // 4)
        for _ in range(n_sync):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **host)
        sync_ms = (time.perf_counter() - t0) / n_sync * 1e3
        in_bytes = sum(a.nbytes for a in host.values())
        pl.set_pipeline(LEG_PIPELINE)
        ring = pl.host_ring(p, scenes, P, cfg.n_obs, M)
        out_bytes = sum(a.nbytes for a in ring.slots[0].outputs.values())
        none8 = (None,) * 8

        def run(k, load):
            for _ in range(k):
                slot = ring.next()
                if load:
                    slot.load(**host)
                pl.plan_cycle(p, q, sp, *none8, max_pts=M, slot=slot)
            ring.wait_all()

        for s_ in ring.slots:
            s_.load(**host)
        run(120, True)                  # (the first hundred calls through fresh page-locked memory run at half speed)
        t0 = time.perf_counter()
        run(steps, True)
        ring_ms = (time.perf_counter() - t0) / steps * 1e3
        t0 = time.perf_counter()
        run(steps, False)
        inplace_ms = (time.perf_counter() - t0) / steps * 1e3
        ok = float(((ring.slots[0].outputs["status"] & ~1) == 0).mean())
        same = bool(np.array_equal(ring.slots[0].outputs["status"], r.status) and
                    np.array_equal(ring.slots[0].outputs["traj_len"], r.traj_len))
        ring.close()
        ring = None
        # one scene per call through the same path (a driver that plans for one vehicle and holds NumPy arrays)
        one = {k: v[:1] for k, v in host.items()}
        pl.set_pipeline(0)
        for _ in range(5):
            pl.plan_cycle(p, q, sp, max_pts=M, **one)
        t0 = time.perf_counter()
        for _ in range(30):
            pl.plan_cycle(p, q, sp, max_pts=M, **one)
        one_ms = (time.perf_counter() - t0) / 30 * 1e3
        return {"workload": f"{scenes} scenes, lattice col={cfg.col} x row={cfg.row}, {cfg.n_obs} obstacles; NumPy arrays in and out "
                            "(host memory at the boundary, PCIe included)",
                "steps": steps, "bytes_in_per_step": int(in_bytes), "bytes_out_per_step": int(out_bytes),
                "host_ring": {"ms_per_step": round(ring_ms, 4), "all_scenes_cycles_per_s": round(scenes / ring_ms * 1e3, 1),
                              "fully_planned_cycles_per_s": round(scenes / ring_ms * 1e3 * ok, 1),
                              "pcie_gbs_both_directions": round((in_bytes + out_bytes) / ring_ms / 1e6, 1),
                              "how": "pageable NumPy inputs -> np.copyto into a page-locked ring slot -> emp_plan_cycle(EMP_HOST_PINNED) "
                                     "three batches deep (one H2D copy on the copy stream, one D2H copy on its own stream) -> "
                                     "results read in place from the slot's page-locked arrays"},
                "host_ring_inputs_written_in_place": {"ms_per_step": round(inplace_ms, 4),
                                                      "all_scenes_cycles_per_s": round(scenes / inplace_ms * 1e3, 1)},
                "synchronous_pageable_path": {"ms_per_step": round(sync_ms, 4), "all_scenes_cycles_per_s": round(scenes / sync_ms * 1e3, 1),
                                              "how": "emp_plan_cycle(EMP_HOST): the library stages pageable arrays, one call at a time (rounds 1-4)"},
                "one_scene_host_latency_ms": round(one_ms, 4), "scenes_fully_planned_frac": round(ok, 4),
                "ring_outputs_equal_the_synchronous_path": same, "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    finally:
        try:
            if ring is not None:
                ring.close()
            pl.set_pipeline(0)
        except Exception:
            pass


def gather_path_leg(pl, torch, emp_dist, cfg, scenes, steps, device, scene_kw=None, records="full"):
    """The per-step code of an N > 1 rank on this one GPU (what `--force-gather-path` runs as a line of its own): the
    step + record packing on the result stream + the gather on a stream of its own (the identity without a process group).
    No 2/4/8-GPU node has been available to any round: this leg, the gloo step-loop tests and the shard == slice tests are
    what stands in for the scaling run."""
    from emplanner_carla_amd import _lib as L
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import dp_params_from_cfg, max_path_points, qp_params, smooth_params
    t_leg = time.perf_counter()
    try:
        inputs = _device_inputs(torch, S, cfg, range(scenes), device, scene_kw)
        p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
        M = max_path_points(p)
        pl.set_timing(False)
        pl.set_pipeline(LEG_PIPELINE)
        ts = pl.torch_stream()
        sg = emp_dist.StepGather(p.col, M, scenes, planner=pl, fields=records, device=device, dst=0, timing=True)

        def step(gather=True):
            with torch.cuda.stream(ts):
                res = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
            return sg.submit(res) if gather else res

        def fence():
            pl.synchronize()
            torch.cuda.synchronize()

        for _ in range(120):
            step()
        fence()
        sg.timed = []
        pl.set_timing(True, only="dp_sweep")
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        fence()
        el = time.perf_counter() - t0
        sweep_ms = pl.kernel_ms("dp_sweep")
        pl.set_timing(False)
        sg.drain()
        gms = sg.gather_ms()
        for _ in range(8):
            step(False)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(False)
        fence()
        nog = time.perf_counter() - t0
        pl.set_timing(True, only="pack_records")
        for _ in range(4):
            step()
        fence()
        pack_ms = pl.kernel_ms("pack_records")
        pl.set_timing(False)
        sg.drain()
        pl.set_pipeline(0)
        complete = bool(sg.unpack(out)["status"].shape[0] == scenes)
        bytes_dp = (8 * (cfg.row + (cfg.col - 1) * cfg.row ** 2) + 4 * cfg.row * cfg.col + 4 * cfg.col) * scenes
        return {"workload": f"{scenes} scenes on ONE GPU through the N > 1 per-step code: three lanes, {records} records packed on the "
                            "result stream, gather to rank 0 on its own stream (identity: one process)",
                "steps": steps, "ms_per_step": round(el / steps * 1e3, 4), "ms_per_step_without_pack_and_gather": round(nog / steps * 1e3, 4),
                "all_scenes_cycles_per_s": round(scenes * steps / el, 1),
                "sweep_mean_launch_us": round(sweep_ms * 1e3, 2), "sweep_frac": round(bytes_dp / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "pack_kernel_us": (round(pack_ms * 1e3, 2) if pack_ms >= 0 else None),
                "gather_us_on_its_stream": (round(gms[0] * 1e3, 2) if gms else None),
                "doubles_per_scene": sg.width, "bytes_sent_per_rank_and_step": sg.bytes_per_rank_and_step(scenes),
                "rank0_ingest_at_8_ranks": {"bytes_per_step": 7 * sg.bytes_per_rank_and_step(scenes),
                                            "per_xgmi_link_gbs": round(sg.bytes_per_rank_and_step(scenes) / (el / steps) / 1e9, 1),
                                            "note": "each peer reaches rank 0 over its own point-to-point xGMI link (~153 GB/s): the grouped "
                                                    "send / recv form of torch's gather uses the seven links side by side (DESIGN 7)"},
                "records_complete": complete, "no_scaling_curve_exists": "no 2/4/8-GPU node in any round (SCALE_r0x.json: skipped)",
                "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    except Exception as exc:
        try:
            pl.set_timing(False)
            pl.set_pipeline(0)
        except Exception:
            pass
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}


#: cycles a wave64 FP64 vector instruction holds a SIMD's vector pipe, measured (tools/fp64_pipe_bench.hip, four wavefronts per
#: SIMD: v_fma_f64 / v_add_f64 / v_mul_f64 / v_min_f64 4.31; the SQ counters charge an active VALU instruction 4) and the engine
#: clock the capacity is priced at (MI355X_MICROARCH.md; the sweep's in-kernel clock probe read 2.41-2.42 GHz in the step)
FP64_PIPE_CYCLES, SQ_CHARGED_CYCLES, ENGINE_CLOCK_HZ, SIMDS = 4.31, 4.0, 2.4e9, 1024


def roofline_step(cfg, count, scene_dist, ms_per_step, speed=False, measured_clock_mhz=None):
    """The whole step against the chip's vector-issue capacity (the path is FP64-issue bound everywhere but in the sweep): the
    VALU-busy quad-cycles of the step's six kernels, from the committed SQ counter pass of this workload (kernels run one at a
    time there: what they NEED, whatever overlaps what in the step), over what 1024 SIMDs offer in ms_per_step."""
    prof = committed_profile("step_valu_counters", config=cfg.name, scenes_per_gpu=count, scene_dist=scene_dist)
    if not prof:
        return None
    per_kernel = {n: int(k["valu_busy_quad_cycles"]) for n, k in prof["kernels"].items()}
    lanes = {n: k.get("lanes_active_frac") for n, k in prof["kernels"].items()}
    sources = [prof["source"]]
    if speed:          # configs[4]: the S-T speed DP of the same scenes belongs to the step (its own committed counter pass)
        sp_prof = committed_profile("speed_dp_counters", scenes_per_gpu=count, obstacle_slots=16)
        if not sp_prof:
            return None
        per_kernel["speed_dp"] = int(sp_prof["valu_busy_quad_cycles"])
        lanes["speed_dp"] = sp_prof.get("lanes_active_frac")
        sources.append(sp_prof["source"])
    busy = sum(per_kernel.values())
    capacity = SIMDS * ENGINE_CLOCK_HZ / 4.0 * ms_per_step * 1e-3
    frac = busy / capacity
    at_clock = {}
    if measured_clock_mhz:
        # the counters count SHADER-clock cycles; under this path's FP64 load the chip does not hold its 2.4 GHz peak (the edge-cost
        # kernel's own clock probe, read inside the timed schedule by bench.py): the capacity the step really had is smaller
        f_meas = frac * ENGINE_CLOCK_HZ / (measured_clock_mhz * 1e6)
        at_clock = {"engine_clock_hz_measured": round(measured_clock_mhz * 1e6),
                    "engine_clock_measured_by": "EMP_OPT_EDGE_CLOCK_PROBE: shader-clock over 100 MHz reference ticks of the edge-cost "
                                                "kernel's wavefronts, a few steps of the headline's own schedule after the timed region",
                    "frac_at_the_measured_clock": round(f_meas, 4),
                    "frac_at_the_measured_clock_with_the_measured_fp64_pipe_cost": round(f_meas * FP64_PIPE_CYCLES / SQ_CHARGED_CYCLES, 4)}
    return {"bound": "fp64_valu_issue", "unit": "fraction of the step's VALU issue capacity (1024 SIMDs) its kernels keep busy",
            "frac": round(frac, 4), "frac_is": "at the nominal 2.4 GHz peak clock; frac_at_the_measured_clock is what the chip offered",
            **at_clock,
            "frac_with_the_measured_fp64_pipe_cost": round(frac * FP64_PIPE_CYCLES / SQ_CHARGED_CYCLES, 4),
            "valu_busy_quad_cycles_per_step": int(busy), "capacity_quad_cycles_per_step": int(capacity),
            "per_kernel_valu_busy_quad_cycles": per_kernel, "per_kernel_active_lane_frac": lanes,
            "engine_clock_hz": ENGINE_CLOCK_HZ, "fp64_pipe_cycles_per_wave_instruction": FP64_PIPE_CYCLES,
            "counters_source": "; ".join(sources),
            "note": "SQ_ACTIVE_INST_VALU charges 4 cycles per wave64 vector instruction; an FP64 one holds the pipe 4.31 (v_rcp_f64: "
                    "16.2), measured by tools/fp64_pipe_bench.hip - the second figure scales by that.  What is left is ordering: the "
                    "front queue's kernels wait for each other (projection -> edge costs -> sweep), the path QP is one wavefront "
                    "per two SIMDs and as long as its slowest scene"}


_BENCH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench.py")     # the legs below run it in processes of their own
def rccl_gather_subprocess_leg(steps):
    """`python bench.py --force-gather-path --one-rank-rccl --no-legs --no-cpu-baseline` in a fresh process, as a rank of an
    N > 1 run is: the headline's step with its records packed and gathered to rank 0 by RCCL (one rank: the identity, through
    ProcessGroupNCCL) on the gather stream, every step."""
    import subprocess
    t_leg = time.perf_counter()
    cmd = [sys.executable, _BENCH, "--force-gather-path", "--one-rank-rccl", "--no-legs", "--no-cpu-baseline",
           "--steps", str(steps), "--warmup", "5"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(_BENCH))
        if out.returncode != 0:
            return {"error": f"exit {out.returncode}: {out.stderr[-400:]}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
        d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
        g = d["gather"]
        return {"workload": "the headline's step with every step's records packed on the result stream and gathered to rank 0 by RCCL "
                            "on the gather stream - a process group of ONE rank (all a one-GPU box allows), a process of its own",
                "backend": g.get("backend"), "world_size_seen_by_the_process_group": g.get("world_size_seen_by_the_process_group"),
                "steps": steps, "ms_per_step": d["ms_per_step"], "ms_per_step_without_pack_and_gather": g.get("ms_per_step_without_pack_and_gather"),
                "all_scenes_cycles_per_s": d["all_scenes_cycles_per_s"], "gather_ms_on_its_stream": g.get("gather_ms_on_its_stream"),
                "gather_hidden_behind_compute_frac": g.get("gather_hidden_behind_compute_frac"),
                "records_complete_on_rank0": g.get("records_complete_on_rank0"), "doubles_per_scene": g.get("doubles_per_scene"),
                "sweep_frac": d["roofline"]["frac"], "pipeline": d["config"]["pipeline"],
                "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}


def dropin_subprocess_leg(requests=200):
    """The reference's own call shape (one request, Python lists in, tuples out: test_9.py:92-96, 220, 390-395), timed by
    bench_dropin.py in a process of its own: (i) a request down a real multiprocessing.Pipe to a planning process running the package's
    motion_planning, (ii) the reference's planning-loop body as the explicit function sequence through the drop-in modules,
    (iii) service.plan_requests in-process."""
    import subprocess
    t_leg = time.perf_counter()
    cmd = [sys.executable, os.path.join(os.path.dirname(_BENCH), "bench_dropin.py"), "--requests", str(requests)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(_BENCH))
        if out.returncode != 0:
            return {"error": f"exit {out.returncode}: {out.stderr[-400:]}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
        d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
        d["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
        return d
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}


def staged_subprocess_leg(steps, options):
    """`python bench.py --pipeline staged --no-legs --no-cpu-baseline [--opt ...]` in a fresh process (this one waits, its GPU
    work fenced): the staged form's step time and its sweep's launches, as that command prints them."""
    import subprocess
    t_leg = time.perf_counter()
    cmd = [sys.executable, _BENCH, "--pipeline", "staged", "--no-legs", "--no-cpu-baseline", "--steps", str(steps),
           "--warmup", "5"]
    for k, v in options.items():
        cmd += ["--opt", f"{k}={v}"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(_BENCH))
        if out.returncode != 0:
            return {"error": f"exit {out.returncode}: {out.stderr[-400:]}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
        d = json.loads([ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")][-1])
        return {"pipeline": "staged (emp_set_pipeline(1)), a process of its own", "batches_in_flight": 2, "options": dict(options),
                "steps": steps, "ms_per_step": d["ms_per_step"], "all_scenes_cycles_per_s": d["all_scenes_cycles_per_s"],
                "fully_planned_cycles_per_s": d["value"],
                "sweep_mean_launch_us": d["roofline"]["mean_launch_us"], "sweep_frac": d["roofline"]["frac"],
                "sweep_frac_alone": d["roofline"]["frac_alone"],
                "note": ("the sweep of step k waits (stream-side) for the densification and path QP of step k-1 and overlaps only the "
                         "Cartesian tail: the HBM-bound kernel at its best" if options.get("sweep_exclusive") else
                         "the headline's form until round 5: longer steps, the HBM-bound sweep beside nothing but the previous batch's "
                         "Cartesian tail.  In the headline's three lanes the same sweep shares the chip with the edge-cost kernels of "
                         "two other batches and its launches last about twice as long: that is the schedule, not the kernel"),
                "leg_wall_s": round(time.perf_counter() - t_leg, 2)}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}", "leg_wall_s": round(time.perf_counter() - t_leg, 2)}

