"""One scene per call (BASELINE configs[1]) under the options that matter for latency: plain, as one hipGraph
(EMP_OPT_CYCLE_GRAPH), and both with the two-scenes-per-wavefront path-QP kernel (EMP_OPT_PATH_QP_FORM = 1)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import torch
from emplanner_carla_amd.api import Planner
from emplanner_carla_amd import scenes as S
import bench_legs
pl = Planner(0)
for form in (0, 1):
    pl.set_option("path_qp_form", form)
    for cart in (0, 1):
        pl.set_option("cartesian_form", cart)
        r = bench_legs.latency_leg(pl, torch, torch.device("cuda", 0), calls=200, scene_kw=dict(start_ahead=S.BENCH_START_AHEAD))
        print("path_qp_form", form, "cartesian_form", cart, "plain", r["ms_per_cycle_median"], "graph", r["as_one_hipgraph"]["ms_per_cycle_median"], flush=True)
