"""PCIe-inclusive rate of the planning cycle: the bench workload (CFG2, 4096 scenes) with HOST buffers at the
boundary (numpy in, numpy out: emp_plan_cycle stages inputs to the device and copies every result back), beside
the HBM-resident rate bench.py reports.  Usage: python tools/host_rate_probe.py [scenes] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg = S.CFG2
batch = S.make_batch(range(B), cfg)
P = batch.ref.shape[1]
c = np.ascontiguousarray
inputs = dict(ref_line=c(batch.ref), n_ref=np.full(B, P, np.int32), origin_xy=c(batch.origin_xy),
              start_xy=c(batch.start_xy), start_v=c(batch.start_v), start_a=c(batch.start_a),
              obs_xy=c(batch.obs_xy), n_obs=c(batch.n_obs))
in_bytes = sum(a.nbytes for a in inputs.values())
pl = Planner(0)
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)
for _ in range(5):
    res = pl.plan_cycle(p, q, sp, max_pts=M, **inputs)
out_bytes = sum(v.nbytes for v in vars(res).values() if isinstance(v, np.ndarray)) if hasattr(res, "__dict__") else 0
t0 = time.perf_counter()
for _ in range(K):
    res = pl.plan_cycle(p, q, sp, max_pts=M, **inputs)
dt = (time.perf_counter() - t0) / K
print(f"host buffers: {B} scenes, {dt * 1e3:.3f} ms per step, {B / dt / 1e6:.3f} M planning cycles/s "
      f"(inputs {in_bytes / 1e6:.1f} MB, results {out_bytes / 1e6:.1f} MB per step)")
