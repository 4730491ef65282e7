"""The edge-cost kernel alone, a few launches (for rocprofv3 counter passes: tools/pmc_sq_cmd.sh TAG tools/edge_only.py ...).
Usage: python tools/edge_only.py [cfg2|cfg5] [scenes] [edge_form] [edge_block]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emplanner_carla_amd import _lib as L
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg

a = sys.argv[1:] + [None] * 5
cfg = {"cfg2": S.CFG2, "cfg5": S.CFG5}[a[0] or "cfg2"]
B = int(a[1] or 4096)
batch = S.make_batch(range(B), cfg, start_ahead=S.BENCH_START_AHEAD)
pl = Planner(0)
pl.set_option("edge_form", int(a[2] or 0))
pl.set_option("edge_block", int(a[3] or 0))
p = dp_params_from_cfg(cfg)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
obs_s, obs_l, n_obs, start = t(batch.sl_obs_s), t(batch.sl_obs_l), t(batch.n_obs), t(batch.sl_start)
pl.set_timing(True, only="dp_edge")
for _ in range(8):
    pl.dp_edge_costs(p, obs_s, obs_l, n_obs, start, layout=L.EMP_EDGE_TILED)
pl.synchronize()
print(f"{cfg.name} {B} scenes form {a[2] or 0}: dp_edge {pl.kernel_ms('dp_edge') * 1e3:.1f} us")
