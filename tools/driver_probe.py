"""Development probe: one driver-level request through service.plan_requests, printed next to the reference's reply."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from emplanner_carla_amd.api import Planner
from emplanner_carla_amd import service
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "driver.npz"))
def req(c):
    ns, nd = int(g["n_static"][c]), int(g["n_dynamic"][c])
    return ([tuple(r) for r in g["static"][c, :ns]], [tuple(r) for r in g["dynamic"][c, :nd]], tuple(g["veh"][c]),
            tuple(g["pred"][c]), tuple(g["v"][c]), tuple(g["a"][c]), [tuple(r) for r in g["path"][c]], [int(g["pre_match"][c])])
pl = Planner(0)
c = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for batch in ([req(c)], [req(k) for k in range(18)]):
    reps = service.plan_requests(pl, batch)
    r, st = reps[0] if len(batch) == 1 else reps[c]
    print("batch of", len(batch), "status", st)
    if r:
        traj, match, ps, pll = r
        print(" path_s", np.round(ps, 2)); print(" path_l", np.round(pll, 3))
m = int(g["n_path"][c])
print("ref path_l", np.round(g["path_l"][c, :m], 3))
