"""Development micro-benchmark of the S-T speed planning back end (emp_st_backend_kernels.h) behind the speed DP;
not the judged bench.py.  Usage: python tools/st_backend_bench.py [B]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, speed_dp_params, speed_qp_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
o = S.make_dynamic_batch(range(B), 16, None)
pl = Planner(0)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sets = pl.st_graph(*[t(a) for a in o[:4]])
v0 = t(o[4])
res = pl.speed_dp(speed_dp_params(), *sets, v0, tables=False)
pl.synchronize()
dp_s, dp_t = res.speed_s.clone(), res.speed_t.clone()
dp_s[:, 12:] = float("nan")          # speed_QP only accepts profiles with a NaN tail (speed_planning_test.py:435)
dp_t[:, 12:] = float("nan")
rng = np.random.default_rng(0)
P = 96
s_path = np.cumsum(rng.uniform(0.9, 1.1, (B, P)), axis=1) - 1.0
s_path[:, 0] = 0.0
kappa = 0.02 * np.sin(s_path / 15.0)
th = np.cumsum(kappa, axis=1)
x_init, y_init = np.cumsum(np.cos(th), axis=1), np.cumsum(np.sin(th), axis=1)
pad = lambda a: np.concatenate([a[:, :80], np.full((B, P - 80), np.nan)], axis=1)
path = dict(i2s=t(s_path), kappa=t(kappa), n=t(np.full(B, P, np.int32)), x=t(pad(x_init)), y=t(pad(y_init)), h=t(pad(th)),
            k=t(pad(kappa)), now=t(np.zeros(B)))
a0 = t(np.zeros(B))
qp = speed_qp_params()


def chain():
    cs = pl.speed_convex_space(dp_s, dp_t, path["i2s"], path["kappa"], path["n"], *sets)
    q = pl.speed_qp(qp, v0, a0, dp_s, dp_t, *cs[:4])
    d = pl.speed_increase_points(*q[:4])
    m = pl.path_speed_merge(*d[:4], path["now"], path["i2s"], path["x"], path["y"], path["h"], path["k"], path["n"])
    return cs, q, d, m


pl.set_timing(True)
for _ in range(3):
    out = chain()
pl.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    out = chain()
pl.synchronize()
dt = (time.perf_counter() - t0) / N
cs, q, d, m = out
st = [x[-1].cpu().numpy() if isinstance(x, tuple) else None for x in (cs, q, d)] + [m[1].cpu().numpy()]
print(f"B={B}: chain {dt * 1e3:.3f} ms -> {B / dt:.0f} speed profiles/s")
for name in ("speed_convex_space", "speed_qp", "speed_increase_points", "path_speed_merge"):
    print(f"  {name:24s} {pl.kernel_ms(name):.4f} ms")
print("  status ok fractions: convex", (st[0] == 0).mean(), "qp", (st[1] == 0).mean(), "dense", (st[2] == 0).mean(), "merge",
      (st[3] == 0).mean(), "| qp iterations mean/max", q[4].cpu().numpy()[st[1] == 0].mean(), q[4].cpu().numpy().max())
