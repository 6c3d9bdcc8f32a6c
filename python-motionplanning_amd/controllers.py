"""Host-side mirror of /root/reference/libs/controllers/stanley_controller.py executed on
MI355X: ``StanleyController`` and ``LongitudinalController`` keep the reference's
constructor arguments, method names, argument meaning and return values, and run the
controller kernel behind ``vdyn_controller_update_*`` (include/vdyn.h).  They are the
single-vehicle drop-ins for drive.py:107-110,129-133; batches go through
``VehicleModel.controller_update`` / ``VehicleModel.closed_loop``.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .vehicle_model import VehicleModel


def _gains(**kw):
    g = _lib.default_ctrl_gains()
    for k, v in kw.items():
        setattr(g, k, float(v))
    return g


class StanleyController:
    """stanley_controller.py:6-129."""

    def __init__(self, control_gain=2.5, softening_gain=1.0, yaw_rate_gain=0.0, steering_damp_gain=0.0,
                 max_steer=np.deg2rad(24), wheelbase=0.0, waypoints=None, device=0):
        self.k = control_gain
        self.k_soft = softening_gain
        self.k_yaw_rate = yaw_rate_gain
        self.k_damp_steer = steering_damp_gain
        self.max_steer = max_steer
        self.L = wheelbase
        self._waypoints = waypoints
        self._lookahead_distance = 5.0
        self.cross_track_deadband = 0.01
        self._vm = VehicleModel(wheelbase, max_steer, 1.0, device=device)

    def update_waypoints(self, new_waypoints):
        self._waypoints = new_waypoints

    def stanley_control(self, x, y, yaw, current_velocity):
        """-> (limited_steering_angle, target_index, crosstrack_error), :129."""
        wp = np.asarray(self._waypoints, dtype=np.float64)[:, :2]   # rows are [x, y, v] (local_planner.py:419)
        st = np.zeros((12, 1))
        st[0, 0], st[7, 0], st[8, 0], st[9, 0] = current_velocity, yaw, x, y
        g = _gains(k=self.k, k_soft=self.k_soft, max_steer=self.max_steer,
                   lookahead=self._lookahead_distance, deadband=self.cross_track_deadband)
        _, out = self._vm.controller_update(st, np.zeros((6, 1)), wp, gains=g)
        return out[0, 0], int(out[1, 0]), out[2, 0]


class LongitudinalController:
    """stanley_controller.py:138-159."""

    def __init__(self, p_gain=1, integral_gain=0, derivative_gain=0, device=0):
        self.kp = p_gain
        self.ki = integral_gain
        self.kd = derivative_gain
        self._vm = VehicleModel(1.0, 0.7, 1.0, device=device)

    def long_control(self, desired_velocity, current_velocity, prev_velocity, v_total_error, dt):
        """-> (v_total_error_new, [tau, tau, tau, tau]), :159."""
        st = np.zeros((12, 1))
        st[0, 0] = current_velocity
        cs = np.array([0.0, v_total_error, prev_velocity, desired_velocity, 0.0, 0.0])[:, None]
        g = _gains(kp=self.kp, ki=self.ki, kd=self.kd)
        cso, _ = self._vm.controller_update(st, cs, np.array([[0.0, 0.0], [1.0, 0.0]]), gains=g, dt=dt)
        tau = cso[5, 0]
        return cso[1, 0], [tau, tau, tau, tau]
