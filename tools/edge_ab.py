"""Edge-cost kernel alone: work-ring form against the lockstep form, block sizes and columns per wavefront (development aid).
Usage: python tools/edge_ab.py [cfg2|cfg5] [scenes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emplanner_carla_amd import _lib as L
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg

cfg = {"cfg2": S.CFG2, "cfg5": S.CFG5}[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
batch = S.make_batch(range(B), cfg, start_ahead=S.BENCH_START_AHEAD)
pl = Planner(0)
p = dp_params_from_cfg(cfg)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
obs_s, obs_l, n_obs, start = t(batch.sl_obs_s), t(batch.sl_obs_l), t(batch.n_obs), t(batch.sl_start)
ref = None
for form in (1, 0):
    for block in ((0, 128, 192, 256, 320, 512) if cfg is S.CFG2 else (0, 512, 640, 1024)):
        for cpw in (0, 1, 2, 3, 4):
            pl.set_option("edge_form", form)
            pl.set_option("edge_block", block)
            # (edge_cols_per_wave was an option until ABI 11: the auto rule is the measured optimum)
            pl.set_timing(False)
            for _ in range(3):
                c0, e = pl.dp_edge_costs(p, obs_s, obs_l, n_obs, start, layout=L.EMP_EDGE_TILED)
            pl.synchronize()
            pl.set_timing(True, only="dp_edge")
            for _ in range(10):
                c0, e = pl.dp_edge_costs(p, obs_s, obs_l, n_obs, start, layout=L.EMP_EDGE_TILED)
            pl.synchronize()
            ms = pl.kernel_ms("dp_edge")
            same = ""
            if ref is None:
                ref = e.clone()
            else:
                same = "bit-identical" if torch.equal(torch.nan_to_num(e), torch.nan_to_num(ref)) else "DIFFERENT"
            print(f"form {form} ({'ring' if form == 0 else 'lockstep'}) block {block:4d} cols/wave {cpw:2d}: {ms * 1e3:8.1f} us  {same}", flush=True)
