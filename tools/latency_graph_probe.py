import sys; sys.path.insert(0,'.')
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(8)
import torch, json
from emplanner_carla_amd.api import Planner
from emplanner_carla_amd import scenes as S
import bench_legs
pl=Planner(0)
print(json.dumps(bench_legs.latency_leg(pl, torch, torch.device("cuda",0), calls=200, scene_kw=dict(start_ahead=S.BENCH_START_AHEAD))))
