import sys,time
sys.path.insert(0,'/root/repo')
import numpy as np
import bench_dropin as bd
from emplanner_carla_amd import service
from emplanner_carla_amd.api import Planner
pl=Planner(0)
one=service.RequestPlanner(pl)
reqs=[bd.make_request(1000+k) for k in range(100)]
for r in reqs[:20]: one.plan(r)
pl.set_timing(True)
t0=time.perf_counter()
for r in reqs: one.plan(r)
dt=(time.perf_counter()-t0)/len(reqs)*1e3
names=("reference_line","project","dp_edge","dp_sweep","dp_enrich","path_qp","to_cartesian")
k={n:round(pl.kernel_ms(n)*1e3,1) for n in names}
print("per request %.3f ms (with event timing on)"%dt, k, "sum", round(sum(k.values()),1))
pl.set_timing(False)
t0=time.perf_counter()
for r in reqs: one.plan(r)
print("per request %.3f ms"%((time.perf_counter()-t0)/len(reqs)*1e3))
# python-side share: pack only
import cProfile,pstats,io
pr=cProfile.Profile(); pr.enable()
for r in reqs: one.plan(r)
pr.disable(); s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(8); print(s.getvalue()[:1800])
