import sys
import numpy as np, torch
sys.path.insert(0, ".")
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner
from tests.test_gpu_fullsize import _host_inputs, _plan_resident, _masked, OUTPUTS
cfg = S.CFG2; B = 4096
pl = Planner(0)
host = _host_inputs(S.make_batch(range(B), cfg))
out = _masked(_plan_resident(pl, cfg, host))
perm = np.random.default_rng(7).permutation(B)
outp = _masked(_plan_resident(pl, cfg, host, perm))
ok = (out["status"] & ~1) == 0
for k in OUTPUTS:
    a, b = out[k][perm], outp[k]
    okp = ok[perm]
    d = (a != b) & (okp.reshape((-1,) + (1,) * (a.ndim - 1)))
    if d.any():
        idx = np.argwhere(d)
        print(k, "differs in", len(np.unique(idx[:, 0])), "scenes; first:", idx[:5].tolist())
        i = idx[0]
        print("  orig", a[tuple(i)], "perm", b[tuple(i)], "rel", abs(a[tuple(i)] - b[tuple(i)]) / max(abs(a[tuple(i)]), 1e-300))
        sc = i[0]
        print("  scene (perm pos)", sc, "orig index", perm[sc], "status", out["status"][perm[sc]], outp["status"][sc],
              "len", out["path_len"][perm[sc]], outp["path_len"][sc])
        if a.ndim == 2:
            print("  a", a[sc][:26]); print("  b", b[sc][:26])
print("---- neighbours of the scenes that differ")
for pos in (1311, 2727, 3700):
    nb = pos ^ 1
    print("pos", pos, "orig", perm[pos], "status orig/perm", out["status"][perm[pos]], outp["status"][pos],
          "| neighbour pos", nb, "orig", perm[nb], "status orig/perm", out["status"][perm[nb]], outp["status"][nb])
# alone and paired
import torch
def plan_idx(idx):
    o = _plan_resident(pl, cfg, host, np.array(idx))
    return o["status"].tolist()
for pos in (1311, 2727, 3700):
    a, b = perm[pos], perm[pos ^ 1]
    print("alone", plan_idx([a]), "pair(nb, me)", plan_idx([b, a]) if pos & 1 else plan_idx([a, b]), "pair swapped", plan_idx([a, b]) if pos & 1 else plan_idx([b, a]))
