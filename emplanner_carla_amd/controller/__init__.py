"""Drop-in for the lateral MPC of the reference's ``controller`` package (controller/controller.py:65-337).

    from emplanner_carla_amd.controller.controller import Lateral_MPC_controller

Same constructor and ``_control()`` as the reference class; the vehicle object is duck-typed (anything with CARLA's
``get_location / get_transform / get_velocity / get_angular_velocity``), the controller arithmetic runs in the HIP
kernel behind ``emp_mpc_lateral``.  The LQR / PID classes and ``Vehicle_control`` are not provided.
"""
from . import controller  # noqa: F401
