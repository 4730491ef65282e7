"""Per-launch durations of the DP sweep inside timed regions shaped like bench.py's (fence, set_timing, K staged steps, fence):
which launches of a region are the slow ones.  Usage: python tools/sweep_samples.py [--steps 20] [--regions 4] [--json out]
name=value,... variants as in tools/step_variants.py."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--regions", type=int, default=4)
ap.add_argument("--settle", type=int, default=150)
ap.add_argument("--json", default="")
ap.add_argument("variants", nargs="*", default=["default"])
a = ap.parse_args()
dev = torch.device("cuda", 0); cfg, B = S.CFG2, 4096
batch = S.make_batch(range(B), cfg); P = batch.ref.shape[1]
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p); pl = Planner(0); ts = pl.torch_stream()
def step():
    with torch.cuda.stream(ts):
        return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
def fence(): pl.synchronize(); torch.cuda.synchronize()
out = []
for var in a.variants:
    for k in L.OPTIONS:
        pl.set_option(k, L.OPTION_DEFAULTS.get(k, 0))
    mode = 1
    if var == "off": mode = 0
    elif var != "default":
        for kv in var.split(","):
            k, v = kv.split("="); pl.set_option(k, int(v))
    pl.set_pipeline(mode)
    for _ in range(a.settle): step()
    fence()
    for r in range(a.regions):
        pl.set_timing(True, only="dp_sweep")
        t0 = time.perf_counter()
        for _ in range(a.steps): step()
        fence()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        smp = pl.kernel_samples("dp_sweep") * 1e3
        pl.set_timing(False)
        rec = {"variant": var, "region": r, "ms_per_step": round(ms, 4), "mean_us": round(float(smp.mean()), 2),
               "median_us": round(float(np.median(smp)), 2), "samples_us": [round(float(v), 1) for v in smp]}
        out.append(rec); print(json.dumps(rec), flush=True)
if a.json: json.dump(out, open(a.json, "w"), indent=1)
pl.close()
