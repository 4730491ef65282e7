"""Staged steps of the benchmark batch under emp_set_option variants, in one process (development A/B; the numbers behind
profiles/r04_sweep/README.md and DESIGN.md section 4).  For every variant: ms per step, the sweep's mean launch duration
(HIP events attached to its dispatch), the shader clock its wavefronts ran at and how long they were resident (the
in-kernel clock probe).  Usage: python tools/step_variants.py [--steps N] [--json out.json] name=value,name=value ...
("default" = no option; "off" = one batch in flight)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--scenes", type=int, default=4096)
ap.add_argument("--repeat", type=int, default=2)
ap.add_argument("--json", default="")
ap.add_argument("--no-probe", action="store_true", help="leave the in-kernel clock probe off")
ap.add_argument("variants", nargs="*", default=["default"])
a = ap.parse_args()
dev = torch.device("cuda", 0); cfg, B = S.CFG2, a.scenes
batch = S.make_batch(range(B), cfg); P = batch.ref.shape[1]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p); pl = Planner(0); ts = pl.torch_stream()
E_ = cfg.row + (cfg.col - 1) * cfg.row ** 2
bytes_dp = (8 * E_ + 4 * cfg.row * cfg.col + 4 * cfg.col) * B
def step():
    with torch.cuda.stream(ts):
        return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
def fence(): pl.synchronize(); torch.cuda.synchronize()
results = []
pl.set_pipeline(0)
ref = step(); fence()
ref_traj, ref_status = ref.traj.clone(), ref.status.clone()
for rep in range(a.repeat):
    for var in a.variants:
        for k in L.OPTIONS:                       # back to the defaults
            pl.set_option(k, L.OPTION_DEFAULTS.get(k, 0))
        mode = 1
        if var == "off": mode = 0
        elif var != "default":
            for kv in var.split(","):
                k, v = kv.split("="); pl.set_option(k, int(v))
        pl.set_option("sweep_clock_probe", 0 if a.no_probe else 1)
        pl.set_pipeline(mode)
        for _ in range(150): step()
        fence()
        pl.set_option("sweep_clock_probe", 0 if a.no_probe else 1)     # restart the probe's statistics
        pl.set_timing(True, only="dp_sweep")
        t0 = time.perf_counter()
        for _ in range(a.steps): last = step()
        fence()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        same = bool(torch.equal(last.status, ref_status)) and bool(torch.equal(torch.nan_to_num(last.traj), torch.nan_to_num(ref_traj)))
        sw = pl.kernel_ms("dp_sweep") * 1e3
        pl.set_timing(False)
        clk = pl.sweep_clock()
        spans = None if a.no_probe else pl.sweep_probe_spans()
        r = {"variant": var, "rep": rep, "ms_per_step": round(ms, 4), "sweep_us": round(sw, 2),
             "sweep_frac": round(bytes_dp / (sw * 1e-6) / 8e12, 4), "same_as_unpipelined": same,
             "sweep_clock_mhz": None if clk is None else round(clk[0], 1),
             "wave_resident_us_mean_max": None if clk is None else [round(clk[1], 2), round(clk[2], 2)],
             "wave_start_spread_us": None if spans is None else round(spans[0], 2),
             "first_start_to_last_end_us": None if spans is None else round(spans[1], 2)}
        results.append(r)
        print(json.dumps(r), flush=True)
if a.json:
    json.dump(results, open(a.json, "w"), indent=1)
pl.close()
