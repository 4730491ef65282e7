"""Drop-in for the lateral controllers of the reference's ``controller`` package (controller/controller.py:65-337 MPC,
:374-611 LQR).

    from emplanner_carla_amd.controller.controller import Lateral_MPC_controller, Lateral_LQR_controller

Same constructors and ``_control()`` as the reference classes; the vehicle object is duck-typed (anything with CARLA's
``get_location / get_transform / get_velocity / get_angular_velocity``), the controller arithmetic runs in the HIP
kernels behind ``emp_mpc_lateral`` / ``emp_lqr_lateral``.  The PID class, the feed-forward MPC variant and
``Vehicle_control`` are not provided.
"""
from . import controller  # noqa: F401
