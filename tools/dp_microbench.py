"""Development micro-benchmark of the DP kernels (not the judged bench.py)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from emplanner_carla_amd import scenes as S, _lib as L
from emplanner_carla_amd.api import Planner, dp_params_from_cfg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = S.CONFIGS.get(os.environ.get('EMP_CFG', ''), S.CFG2)
batch = S.make_batch(range(B), cfg)
pl = Planner(0)
p = dp_params_from_cfg(cfg)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
if len(sys.argv) > 2 and sys.argv[2] == 'noobs':
    batch.n_obs[:] = 0
obs_s, obs_l, n_obs, start = t(batch.sl_obs_s), t(batch.sl_obs_l), t(batch.n_obs), t(batch.sl_start)
pl.set_timing(True)
for mode in (0, 1):
    for it in range(5):
        rows, mc, st = pl.dp_plan(p, obs_s, obs_l, n_obs, start, mode=mode)
    pl.synchronize()
    t0 = time.perf_counter()
    N = 20
    for it in range(N):
        rows, mc, st = pl.dp_plan(p, obs_s, obs_l, n_obs, start, mode=mode)
    pl.synchronize()
    dt = (time.perf_counter() - t0) / N
    E = cfg.row + (cfg.col - 1) * cfg.row ** 2
    bytes_dp = (8 * E + 4 * cfg.row * cfg.col + 4 * cfg.col) * B
    print(f"mode {mode}: {dt*1e6:.1f} us/batch of {B} -> {B/dt/1e6:.2f} M scenes/s; edge {pl.kernel_ms('dp_edge')*1e3:.1f} us, "
          f"sweep {pl.kernel_ms('dp_sweep')*1e3:.1f} us -> sweep {bytes_dp/ (pl.kernel_ms('dp_sweep')*1e-3)/1e12:.2f} TB/s algorithmic")
print("infeasible", int((st.cpu().numpy() & 1).sum()))
