"""Development probe: sweep kernel duration vs number of columns for one wavefront (latency chain)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd.api import Planner, dp_params
pl = Planner(0)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
for col in (2, 6, 11, 21, 40, 80):
    p = dp_params(row=9, col=col, sample_s=2.5, sample_l=1.5)
    rng = np.random.default_rng(0)
    obs_s = torch.from_numpy(rng.uniform(5, 80, (B, 8))).to(dev)
    obs_l = torch.from_numpy(rng.uniform(4.5, 7, (B, 8))).to(dev)
    n_obs = torch.full((B,), 8, dtype=torch.int32, device=dev)
    start = torch.from_numpy(np.tile([2.0, 0.1, 0.0, 0.0], (B, 1))).to(dev)
    for _ in range(3):
        pl.dp_plan(p, obs_s, obs_l, n_obs, start, mode=1)
    pl.set_timing(True)
    for _ in range(20):
        pl.dp_plan(p, obs_s, obs_l, n_obs, start, mode=1)
    pl.synchronize()
    print(f"col {col:3d}: sweep {pl.kernel_ms('dp_sweep')*1e3:7.2f} us  edge {pl.kernel_ms('dp_edge')*1e3:7.2f} us")
    pl.set_timing(False)
