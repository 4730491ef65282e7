"""How close is the process body on the reference's DEFAULT lattice (integer sample_s = 15, where int() truncation makes
the point count depend on the last bits of begin_s) to the reference driver run?  (development probe)"""
import sys
import numpy as np
sys.path.insert(0, ".")
from emplanner_carla_amd import service
from emplanner_carla_amd.api import Planner
from tests.conftest import load_golden
from tests.test_gpu_cycle import _driver_request

g = load_golden("driver.npz")
pl = Planner(0)
reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
out = service.plan_requests(pl, reqs)
for c, (reply, status) in enumerate(out):
    if reply is None:
        print(c, "kind", int(g["case"][c]), "refused status", status, "| reference qp_ok", int(g["qp_ok"][c]))
        continue
    traj, match, ps, plv = reply
    n, m = int(g["n_traj"][c]), int(g["n_path"][c])
    same = len(traj) == n
    dev = np.abs(np.array(traj)[:, :2] - g["traj"][c, :n, :2]).max() if same else float("nan")
    bs = ps[0] - g["path_s"][c, 0]
    print(c, "kind", int(g["case"][c]), "points", len(traj), "ref", n, "same" if same else "DIFF", "max dev %.2e" % dev,
          "begin_s diff %.3e" % bs, "ref begin_s %.17g" % g["path_s"][c, 0])
