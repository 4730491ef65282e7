"""Development micro-benchmark of the S-T speed DP kernel (BASELINE config 5 shape: 40 x 16 grid, 16 obstacle
slots); not the judged bench.py.  Usage: python tools/st_microbench.py [B] [n_present]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emplanner_carla_amd import _lib
if os.environ.get("EMP_DBG_LIB"):                     # a development build of the library (tools/dbg_*.hip)
    _lib.LIB_PATH = os.path.abspath(os.environ["EMP_DBG_LIB"])
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, speed_dp_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n_present = int(sys.argv[2]) if len(sys.argv) > 2 else None
o = S.make_dynamic_batch(range(B), 16, n_present)
pl = Planner(0)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sets = pl.st_graph(*[t(a) for a in o[:4]])
if os.environ.get("ST_SORT"):                         # heaviest scenes first (most S-T obstacles): what LPT scheduling would buy
    order = torch.argsort((~torch.isnan(sets[0])).sum(1), descending=os.environ["ST_SORT"] != "asc", stable=True)
    sets = tuple(a[order].contiguous() for a in sets)
    o = o[:4] + (o[4][order.cpu().numpy()],)
v0 = t(o[4])
live = float((~torch.isnan(sets[0])).sum().item()) / B
pl.set_timing(True)
p = speed_dp_params()
for tables in (True, False):
    for _ in range(3):
        res = pl.speed_dp(p, *sets, v0, tables=tables)
    pl.synchronize()
    N = 10
    t0 = time.perf_counter()
    for _ in range(N):
        res = pl.speed_dp(p, *sets, v0, tables=tables)
    pl.synchronize()
    dt = (time.perf_counter() - t0) / N
    edges = 40 + 15 * 1600
    print(f"tables={tables}: {dt * 1e3:.3f} ms per batch of {B} ({live:.1f} S-T obstacles/scene) -> {B / dt:.0f} speed DPs/s, "
          f"{B * edges / dt / 1e9:.2f} G edges/s, kernel {pl.kernel_ms('speed_dp'):.3f} ms")
end = res.end_node.cpu().numpy()
print("terminal column histogram", np.bincount(end[:, 1], minlength=16))
