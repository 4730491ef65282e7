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

# ---- the overlapped host path (api.HostRing: page-locked rings, copy streams, staged pipeline)
from concurrent.futures import ThreadPoolExecutor
pl.set_pipeline(1)
ring = pl.host_ring(p, B, P, cfg.n_obs, M)
for label, loader in (("inputs written in place (zero copy)", None), ("np.copyto from pageable arrays, 1 thread", "serial"),
                      ("np.copyto from pageable arrays, 8 threads", "pool"), ("inputs written in place (zero copy), again", None)):
    pool = ThreadPoolExecutor(8) if loader == "pool" else None
    for s_ in ring.slots:
        s_.load(**inputs)
    for _ in range(8):
        pl.plan_cycle(p, q, sp, None, None, None, None, None, None, None, None, max_pts=M, slot=ring.next())
    ring.wait_all()
    t0 = time.perf_counter()
    for _ in range(K):
        slot = ring.next()
        if loader:
            slot.load(pool=pool, **inputs)
        pl.plan_cycle(p, q, sp, None, None, None, None, None, None, None, None, max_pts=M, slot=slot)
    ring.wait_all()
    dt = (time.perf_counter() - t0) / K
    print(f"host ring, {label}: {dt * 1e3:.3f} ms per step, {B / dt / 1e6:.3f} M planning cycles/s, "
          f"{(in_bytes + out_bytes) / dt / 1e9:.1f} GB/s over PCIe (both directions)")
    if pool:
        pool.shutdown()
ok = float(((ring.slots[0].outputs["status"] & ~1) == 0).mean())
print("planned to the end:", ok)
