"""Does running two half batches on two planner streams beat one full batch?  (experiment, not a test)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

cfg = S.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
batch = S.make_batch(range(B), cfg)
P = batch.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
full = dict(ref_line=batch.ref, n_ref=np.full(B, P, np.int32), origin_xy=batch.origin_xy, start_xy=batch.start_xy,
            start_v=batch.start_v, start_a=batch.start_a, obs_xy=batch.obs_xy, n_obs=batch.n_obs)
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)

def run(parts, steps=30, warm=5):
    pls = [Planner(0) for _ in parts]
    ins = [{k: t(v[a:b]) for k, v in full.items()} for a, b in parts]
    torch.cuda.synchronize()
    def step():
        for pl, i in zip(pls, ins):
            with torch.cuda.stream(pl.torch_stream()):
                pl.plan_cycle(p, q, sp, max_pts=M, **i)
    for _ in range(warm): step()
    for pl in pls: pl.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    for pl in pls: pl.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for pl in pls: pl.close()
    return dt

for lanes in (1, nl, 1, nl):
    S7 = 7 * 4
    cuts = [0] + [((B * k // lanes) // S7) * S7 for k in range(1, lanes)] + [B]
    parts = list(zip(cuts[:-1], cuts[1:]))
    dt = run(parts)
    print(f"lanes {lanes}: {dt*1e3:.4f} ms/step  {B/dt/1e6:.2f} M cycles/s")
