"""The S-L DP on lattices of other widths: rows up to 32 take the tiled kernels (64 / row scenes per
wavefront), rows 33..256 the generic pair (one block per (scene, column), canonical tensor; emp_dp_kernels.h).  Per lattice:
edge-cost and sweep kernel time on B scenes and the cost per lattice edge, so that the wide path can be priced against the
21-row lattice of BASELINE configs[4].  Usage: python tools/wide_lattice_bench.py [scenes] [--json out.json]"""
import dataclasses, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
out_path = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else ""
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
pl = Planner(0)
res = []
for row, col, sample_l in ((9, 40, 1.5), (21, 40, 0.6), (32, 40, 0.4), (33, 40, 0.4), (48, 40, 0.27), (64, 40, 0.2), (128, 40, 0.1), (256, 40, 0.05)):
    cfg = dataclasses.replace(S.CFG2, name=f"{col}x{row}", row=row, col=col, sample_l=sample_l)
    batch = S.make_batch(range(B), cfg)
    p = dp_params_from_cfg(cfg)
    args = (t(batch.sl_obs_s), t(batch.sl_obs_l), t(batch.n_obs), t(batch.sl_start))
    for _ in range(2):
        pl.dp_plan(p, *args, mode=1)
    pl.synchronize()
    pl.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(5):
        rows, mc, st = pl.dp_plan(p, *args, mode=1)
    pl.synchronize()
    wall = (time.perf_counter() - t0) / 5
    e_ms, s_ms = pl.kernel_ms("dp_edge"), pl.kernel_ms("dp_sweep")
    pl.set_timing(False)
    E = row + (col - 1) * row * row
    r = {"lattice": f"col={col} x row={row}", "kernels": "tiled" if row <= 32 else "generic (wide)", "scenes": B, "edges_per_scene": E,
         "dp_edge_ms": round(e_ms, 4), "dp_sweep_ms": round(s_ms, 4), "ns_per_edge_edge_kernel": round(e_ms * 1e6 / (B * E), 3),
         "ns_per_edge_sweep": round(s_ms * 1e6 / (B * E), 4), "sweep_gbs_algorithmic": round(8 * E * B / (s_ms * 1e-3) / 1e9, 1),
         "wall_ms_per_dp_plan": round(wall * 1e3, 3), "infeasible": int((st.cpu().numpy() & 1).sum())}
    res.append(r); print(json.dumps(r), flush=True)
if out_path: json.dump(res, open(out_path, "w"), indent=1)
pl.close()
