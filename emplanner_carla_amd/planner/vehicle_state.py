"""Planar state of a (duck-typed) CARLA vehicle, shared by the drop-in controllers and ``predict_block``.

The reference reads the same quantities from ``carla.Vehicle`` in three places (controller/controller.py:90-113 and
:405-422, planning_utils.py:591-614); here that is one helper.  The vehicle only needs ``get_location()``,
``get_transform().rotation.yaw`` (degrees), ``get_velocity()`` and ``get_angular_velocity().z`` (degrees per second).
"""
from __future__ import annotations

import math
from typing import NamedTuple


class PlanarState(NamedTuple):
    x: float
    y: float
    yaw: float          # rad
    v_long: float       # speed along the body's x axis
    v_lat: float        # speed along the body's y axis
    yaw_rate: float     # rad/s


def planar_state(vehicle) -> PlanarState:
    """Speed is the length of the full 3-D velocity (as the reference takes it); it is split by the slip angle between
    the course over ground and the heading."""
    where = vehicle.get_location()
    yaw = math.radians(vehicle.get_transform().rotation.yaw)
    vel = vehicle.get_velocity()
    speed = math.sqrt(vel.x * vel.x + vel.y * vel.y + vel.z * vel.z)
    slip = math.atan2(vel.y, vel.x) - yaw
    return PlanarState(where.x, where.y, yaw, speed * math.cos(slip), speed * math.sin(slip),
                       math.radians(vehicle.get_angular_velocity().z))
