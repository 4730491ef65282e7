"""Drop-in for the reference's ``planner`` package (the modules on the EM-Planner hot path).

    from emplanner_carla_amd.planner import path_planning, planning_utils      # instead of `from planner import ...`

Same module-level function names, positional order, keyword names and defaults as reference
planner/path_planning.py, planner/planning_utils.py and (S-T speed DP only)
planner/speed_planning_test.py; every numeric result is computed by the HIP kernels behind the C-ABI
(batch size 1).  See INTEGRATION.md.
"""
from . import _runtime  # noqa: F401
from . import path_planning, planning_utils, speed_planning_test  # noqa: F401

__all__ = ["path_planning", "planning_utils", "speed_planning_test"]
