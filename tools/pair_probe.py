import sys, os
import numpy as np
sys.path.insert(0, ".")
from emplanner_carla_amd import _lib
if os.environ.get("EMP_DBG") == "1":
    _lib.LIB_PATH = os.path.abspath("tools/_build/libemplanner_dbg.so")
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner
from tests.test_gpu_fullsize import _host_inputs, _plan_resident
cfg = S.CFG2
pl = Planner(0)
ids = [int(x) for x in sys.argv[1:]]
host = _host_inputs(S.make_batch(ids, cfg))
o = _plan_resident(pl, cfg, host)
pl.synchronize()
print("status", o["status"].tolist(), "path_len", o["path_len"].tolist())
