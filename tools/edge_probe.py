"""Are the edge-cost kernel's wavefronts SLOWER inside the staged step, or merely FEWER at a time?  
EMP_OPT_EDGE_CLOCK_PROBE stamps every wavefront's first and last instruction with the 100 MHz reference counter; this prints,
for the kernel alone (one batch in flight) and inside the staged step (beside the previous batch's densification, path QP and
Cartesian tail): mean wavefront residence, first-start-to-last-end span, mean wavefronts resident at once.
Usage: python tools/edge_probe.py [name=value options ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import numpy as np
import torch

from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

dev = torch.device("cuda", 0)
cfg, B = S.CFG2, 4096
batch = S.make_batch(range(B), cfg, start_ahead=S.BENCH_START_AHEAD)
P = batch.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)
pl = Planner(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    pl.set_option(k, int(v))
pl.set_option("edge_clock_probe", 1)
ts = pl.torch_stream()


def step():
    with torch.cuda.stream(ts):
        return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)


for mode, label in ((0, "alone (one batch in flight)"), (1, "inside the staged step")):
    pl.set_pipeline(mode)
    rows = []
    for rep in range(5):
        for _ in range(60):
            step()
        pl.synchronize()
        torch.cuda.synchronize()
        rows.append(pl.edge_probe())
    m = np.asarray(rows)
    print(f"{label:32s}: wavefront residence {m[:, 0].mean():7.2f} us, first start -> last end {m[:, 1].mean():7.2f} us, "
          f"{m[:, 2].mean():7.1f} wavefronts resident at once of {int(m[0, 3])} (5 launches; residence min {m[:, 0].min():.2f} max {m[:, 0].max():.2f})")
pl.close()
