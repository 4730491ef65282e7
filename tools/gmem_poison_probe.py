"""Does the planning cycle depend on device memory it never wrote?  Fill free device memory with a pattern, release it,
create a fresh planner (its pools and the outputs then land on that memory), run the cycle plain and pipelined, and
compare every output bit for bit across the patterns.  Usage: python tools/gmem_poison_probe.py [scenes] [GiB]"""
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
gib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = S.CFG2
dev = torch.device("cuda:0")
batch = S.make_batch(range(B), cfg)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, batch.ref.shape[1], np.int32)), origin_xy=t(batch.origin_xy),
              start_xy=t(batch.start_xy), start_v=t(batch.start_v), start_a=t(batch.start_a), obs_xy=t(batch.obs_xy),
              n_obs=t(batch.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)


def poison(kind):
    n = gib * (1 << 30) // 8
    big = torch.empty(n, dtype=torch.float64, device=dev)
    if kind == "zero":
        big.zero_()
    elif kind == "nan":
        big.fill_(float("nan"))
    elif kind == "huge":
        big.fill_(1e300)
    elif kind == "ints":
        big.view(torch.int32).fill_(123456789)
    else:
        big.view(torch.int64).random_()
    torch.cuda.synchronize()
    del big
    torch.cuda.empty_cache()


fields = lambda r: {k: v.cpu().numpy() for k, v in vars(r).items() if isinstance(v, torch.Tensor)}
ref = None
for kind in ("zero", "nan", "huge", "ints", "random", "nan"):
    poison(kind)
    pl = Planner(0)
    outs = []
    r = pl.plan_cycle(p, q, sp, max_pts=M, **inputs)
    pl.synchronize()
    outs.append(("plain", fields(r)))
    pl.set_pipeline(True)
    with torch.cuda.stream(pl.torch_stream()):
        rs = [pl.plan_cycle(p, q, sp, max_pts=M, **inputs) for _ in range(3)]
    pl.synchronize()
    for i, r in enumerate(rs):
        outs.append((f"pipelined {i}", fields(r)))
    pl.set_pipeline(False)
    del pl
    if ref is None:
        ref = outs[0][1]
    for what, f in outs:
        for name in f:
            a, b = ref[name].reshape(B, -1), f[name].reshape(B, -1)
            if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                rows = np.nonzero(np.any(a.view(np.uint8) != b.view(np.uint8), axis=1))[0]
                ok = (ref["status"][rows] & ~1) == 0
                print(f"{kind:6s} {what:12s} {name:9s}: {rows.size} scenes differ ({int(ok.sum())} of them planned scenes), first {rows[:6]}")
    print(kind, "done")
