"""How long the host needs to ISSUE one pipelined plan_cycle call (no synchronisation inside the loop): if this is not
well below the GPU's time per step, the queues run dry between steps.  Development probe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import scenes as S, _lib as L
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
cfg = S.CFG2
B = 4096
batch = S.make_batch(range(B), cfg)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
P = batch.ref.shape[1]
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy), start_xy=t(batch.start_xy),
              start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
pl = Planner(0)
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=5, obs_width=5), smooth_params()
M = max_path_points(p)
pl.set_pipeline(True)
ts = pl.torch_stream()
sg = None
if "--gather" in sys.argv:        # the N > 1 per-step code on one GPU: pack on the result stream, gather stream, events
    from emplanner_carla_amd import dist as emp_dist
    sg = emp_dist.StepGather(p.col, M, B, planner=pl, fields="full", device=dev, dst=0, timing="--timing" in sys.argv)
for _ in range(10):
    with torch.cuda.stream(ts):
        r = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
pl.synchronize(); torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    with torch.cuda.stream(ts):
        r = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
    if sg is not None:
        sg.submit(r)
t1 = time.perf_counter()
pl.synchronize(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per call {1e6 * (t1 - t0) / N:.1f} us; until the GPU finished {1e6 * (t2 - t0) / N:.1f} us per step")
