import cProfile,pstats,sys,io,contextlib,time
sys.path.insert(0,".")
import bench_dropin as bd
from emplanner_carla_amd import service
from emplanner_carla_amd.planner import path_planning as pp, planning_utils as pu, _runtime
reqs=[bd.make_request(1000+k) for k in range(100)]
def run_fn(r):
    try:
        with contextlib.redirect_stdout(io.StringIO()): bd.plan_by_functions(r,pu,pp)
    except (ValueError, IndexError): pass
for r in reqs[:10]: run_fn(r)
pr=cProfile.Profile(); pr.enable()
for r in reqs: run_fn(r)
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:5000])
pl=_runtime.planner()
for r in reqs[:10]: service.plan_requests(pl,[r])
pr=cProfile.Profile(); pr.enable()
for r in reqs: service.plan_requests(pl,[r])
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
