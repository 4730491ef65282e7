"""Per-iteration trace (debug build of the library) of the feasible path QPs that need the most iterations
(cases saved by tools/qp_iters_probe.py).  Usage: python tools/qp_straggler_probe.py [min_iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emplanner_carla_amd import _lib
_lib.LIB_PATH = os.path.abspath("tools/_build/libemplanner_dbg.so")
from emplanner_carla_amd.api import Planner, qp_params
g = np.load("gpurun_out/qp_cases.npz")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 14
want = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pick = np.nonzero((g["status"] == want) & (g["iters"] >= thr))[0]
print("stragglers", pick.tolist(), g["iters"][pick].tolist(), flush=True)
pl = Planner(0)
for b in pick[:4]:
    print("==== scene", b, "iters", g["iters"][b], "n", g["n"][b], flush=True)
    print("lo", np.round(g["lo"][b, :g["n"][b]], 2), flush=True)
    print("hi", np.round(g["hi"][b, :g["n"][b]], 2), flush=True)
    sys.stdout.flush()
    pl.path_qp(qp_params(), g["lo"][b:b + 1], g["hi"][b:b + 1], g["n"][b:b + 1], g["start"][b:b + 1])
    pl.synchronize()
