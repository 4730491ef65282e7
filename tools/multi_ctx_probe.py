"""Development probe: does a deeper pipeline help?  K planner contexts (each with its own batches in flight) take the
steps of the default bench in turn, so that up to 2K batches are in flight on 2K streams.
Usage: python tools/multi_ctx_probe.py [contexts] [steps] [batches in flight per context]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get('SETQ') == 'after_import':
    os.environ['GPU_MAX_HW_QUEUES'] = '8'
if os.environ.get('SETQ') == 'after_init':
    torch.zeros(4, device='cuda').sum().item()
    os.environ['GPU_MAX_HW_QUEUES'] = '8'
from emplanner_carla_amd import _lib as L, scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # batches in flight per context
dev = torch.device("cuda", 0)
cfg, B = S.CFG2, 4096
batch = S.make_batch(range(B), cfg)
P = batch.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)
pls = [Planner(0) for _ in range(K)]
for pl in pls:
    pl.set_pipeline(pipe)
tss = [pl.torch_stream() for pl in pls]
def step(i):
    pl = pls[i % K]
    with torch.cuda.stream(tss[i % K]):
        return pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
def fence():
    for pl in pls:
        pl.synchronize()
    torch.cuda.synchronize()
for i in range(10 * K):
    r = step(i)
fence()
t0 = time.perf_counter()
for i in range(steps):
    r = step(i)
fence()
el = time.perf_counter() - t0
print(f"contexts {K} in flight per context {pipe}: {el / steps * 1e3:.4f} ms per step, {B * steps / el / 1e6:.2f} M cycles/s")
