"""Pipelined plan_cycle: throughput against the plain call, and bit-identity of the results (development probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emplanner_carla_amd import _lib
if os.environ.get("EMP_DBG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["EMP_DBG_LIB"])
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params, max_path_points
cfg = S.CFG2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = S.make_batch(range(B), cfg); P = b.ref.shape[1]
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ins = dict(ref_line=t(b.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(b.origin_xy), start_xy=t(b.start_xy), start_v=t(b.start_v),
           start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs))
pl = Planner(0); p = dp_params_from_cfg(cfg); q = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width); sp = smooth_params()
M = max_path_points(p)
ts = pl.torch_stream()
def run(n):
    outs = []
    with torch.cuda.stream(ts):
        for _ in range(n):
            outs.append(pl.plan_cycle(p, q, sp, max_pts=M, **ins))
            if len(outs) > 3: outs.pop(0)
    return outs[-1]
ref = None
for mode in (False, True, False, True):
    pl.set_pipeline(mode)
    run(5); pl.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = run(40); pl.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    pl.set_timing(True, only="dp_sweep"); run(10); pl.synchronize(); sw = pl.kernel_ms("dp_sweep"); pl.set_timing(False)
    got = {k: getattr(r, k).cpu().numpy() for k in ("dp_rows", "traj", "traj_len", "status", "path_l")}
    if ref is None: ref = got
    same = all(np.array_equal(ref[k], got[k], equal_nan=True) for k in ref)
    print("pipelined" if mode else "plain    ", "ms/step %.4f  %.2f M cycles/s  sweep %.1f us  identical %s" % (dt * 1e3, B / dt / 1e6, sw * 1e3, same))
