"""How much of the staged step is the path QP's instruction stream?  Runs the benchmark's step loop (4096 scenes, 40x9, staged
pipeline and three lanes) with the path QP's interior point capped at 0 / 1 / 2 / 4 / 8 iterations and uncapped (QpParams.reserved
= 10 + cap, a development switch: capped runs return unconverged paths, only their timing means anything).  The slope
ms-per-step over iterations is what removing QP instructions would buy.
NEEDS A DEV-HOOKS BUILD: the default library refuses QpParams.reserved != 0 (EMP_ERR_INVALID).  Build one first:
    EMP_EXTRA_FLAGS=-DEMP_DEV_HOOKS=1 python -m emplanner_carla_amd.build --force        (and rebuild without it afterwards)
Usage: python tools/qp_sensitivity_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
dev = torch.device("cuda", 0); cfg, B = S.CFG2, int(os.environ.get("SCENES", "4096"))
batch = S.make_batch(range(B), cfg); P = batch.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, sp = dp_params_from_cfg(cfg), smooth_params()
M = max_path_points(p); pl = Planner(0); ts = pl.torch_stream()
try:                                   # a default build refuses the development switch: say so instead of failing in the loop
    _q = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width); _q.reserved = 10
    pl.plan_cycle(p, _q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs); pl.synchronize()
except Exception as exc:
    sys.exit(f"this probe needs a library built with -DEMP_DEV_HOOKS=1 (EMP_EXTRA_FLAGS=-DEMP_DEV_HOOKS=1 python -m "
             f"emplanner_carla_amd.build --force): {exc}")
def fence(): pl.synchronize(); torch.cuda.synchronize()
for pmode in (1, 3):
    pl.set_pipeline(pmode)
    for cap in (None, 0, 1, 2, 4, 8):
        q = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width)
        if cap is not None: q.reserved = 10 + cap
        def step():
            with torch.cuda.stream(ts):
                return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
        for _ in range(150): step()
        fence()
        t0 = time.perf_counter()
        for _ in range(100): step()
        fence()
        el = (time.perf_counter() - t0) / 100 * 1e3
        print("pipeline %s  ipm iterations cap %s: %.4f ms per step" % ("staged" if pmode == 1 else "%d lanes" % pmode, cap, el), flush=True)
pl.close()
