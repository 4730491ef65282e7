"""Does the planning cycle depend on LDS contents it never wrote?  Fill every CU's LDS with a pattern (zeros, NaNs,
huge values, ...), run the cycle, compare every output bit for bit across the patterns.
Usage: python tools/lds_poison_probe.py [scenes]"""
import ctypes
import os
import struct
import subprocess
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

so = os.path.join(root, "tools", "_build", "liblds_poison.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                           os.path.join(root, "tools", "lds_poison.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.lds_poison.argtypes = [ctypes.c_uint64]

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = S.CFG2
batch = S.make_batch(range(B), cfg)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
P = batch.ref.shape[1]
inputs = dict(ref_line=t(batch.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(batch.origin_xy),
              start_xy=t(batch.start_xy), start_v=t(batch.start_v), start_a=t(batch.start_a),
              obs_xy=t(batch.obs_xy), n_obs=t(batch.n_obs))
pl = Planner(0)
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)
bits = lambda x: struct.unpack("<Q", struct.pack("<d", x))[0]
patterns = {"zero": 0, "nan": bits(float("nan")), "huge": bits(1e300), "-huge": bits(-1e300), "one": bits(1.0),
            "ints": 0x0000000500000007, "inf": bits(float("inf"))}
first = None
for name, pat in patterns.items():
    assert lib.lds_poison(pat) == 0
    r = pl.plan_cycle(p, q, sp, max_pts=M, **inputs)
    pl.synchronize()
    out = {k: v.cpu().numpy() for k, v in vars(r).items() if isinstance(v, torch.Tensor)}
    if first is None:
        first = out
        print("fields:", sorted(out))
        continue
    for k in out:
        same = np.array_equal(first[k].view(np.uint8), out[k].view(np.uint8))
        if not same:
            a, b = first[k].reshape(B, -1), out[k].reshape(B, -1)
            rows = np.nonzero(np.any(a.view(np.uint8).reshape(B, -1) != b.view(np.uint8).reshape(B, -1), axis=1))[0]
            print(f"pattern {name}: {k} differs in {rows.size} scenes, first {rows[:8]}")
    print(f"pattern {name}: done")
