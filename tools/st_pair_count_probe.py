import sys, os
sys.path.insert(0, '/root/repo')
from emplanner_carla_amd import _lib
_lib.LIB_PATH = os.path.abspath('variants/lib_st5.so')
import numpy as np, torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, speed_dp_params
dyn = S.make_dynamic_batch(range(4096), 16)
dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in dyn[:5]]
pl = Planner(0)
sets = pl.st_graph(*dev[:4])
r = pl.speed_dp(speed_dp_params(), *sets, dev[4], tables=False)
t = np.asarray(r.speed_t.cpu() if hasattr(r.speed_t, 'cpu') else r.speed_t)
passes, pairs, rounds = t[:, 0].sum(), t[:, 1].sum(), t[:, 2].sum()
allp = 4096 * 16 * 25 - 4096 * 24   # wave-passes per scene: col0 1 + 15*25
print(f"wave-passes with pairs {passes:.0f} of {allp}; pairs {pairs:.0f} ({pairs/4096/24040:.2f} per edge); rounds {rounds:.0f}; pairs per round {pairs/rounds:.1f}; ideal rounds {pairs/64:.0f}; rounds per pass {rounds/passes:.2f}")
h = np.histogram(t[:,1]/np.maximum(t[:,0],1), bins=[0,32,64,96,128,192,256,512,4096])
print("scene mean pairs per pass histogram", h)
