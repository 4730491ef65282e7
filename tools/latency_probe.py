"""Development probe: end-to-end latency of one planning cycle at small batch sizes (host arrays in, host arrays out -
what the drop-in functions do - and device-resident tensors)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
pl = Planner(0); p = dp_params_from_cfg(cfg); q = qp_params(); sp = smooth_params(); M = max_path_points(p)
dev = torch.device("cuda:0")
for B in (1, 8, 64, 512):
    b = S.make_batch(range(B), cfg); P = b.ref.shape[1]
    host = dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v,
                start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
    devin = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in host.items()}
    for name, ins in (("host arrays", host), ("device tensors", devin)):
        for _ in range(5):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
        pl.synchronize()
        t0 = time.perf_counter()
        N = 50
        for _ in range(N):
            r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
            pl.synchronize()
        dt = (time.perf_counter() - t0) / N
        print(f"B={B:4d} {name:15s}: {dt * 1e6:8.1f} us per synchronous cycle call")
