"""Host-side mirror of /root/reference/libs/motionplanner/collision_checker.py executed on
MI355X: ``CollisionChecker`` keeps the reference's constructor, method names, argument
meaning and return values; both methods run the kernel behind ``vdyn_select_best_path_*``
(include/vdyn.h).  Batches of egos go through ``VehicleModel.select_best_path`` /
``VehicleModel.select_best_rollout``.
"""
from __future__ import annotations

import numpy as np

from .vehicle_model import VehicleModel


class CollisionChecker:
    """collision_checker.py:16-203."""

    def __init__(self, circle_offsets, circle_radii, weight, device=0):
        self._circle_offsets = circle_offsets
        self._circle_radii = circle_radii
        self._weight = weight
        self._vm = VehicleModel(1.0, 0.7, 1.0, device=device)

    def collision_check(self, paths, obstacles):
        """ONE path ``[x_points, y_points, t_points]`` (the reference's `paths` argument is a
        single path, :63) -> True when it is collision free (:115-117)."""
        pa = np.asarray(paths, dtype=np.float64)[None, None]
        free, _, _ = self._vm.select_best_path(pa, np.asarray(obstacles, dtype=np.float64), np.zeros((2, 1)),
                                               self._circle_offsets, self._circle_radii, self._weight)
        return bool(free[0, 0])

    def select_best_path_index(self, paths, collision_check_array, goal_state):
        """-> best index or None (:134-203)."""
        pa = np.asarray(paths, dtype=np.float64)[None]
        goal = np.array([[goal_state[0]], [goal_state[1]]], dtype=np.float64)
        flags = np.asarray(collision_check_array, dtype=np.int32)[None]
        _, bi, _ = self._vm.select_best_path(pa, np.zeros((0, 2)), goal, self._circle_offsets,
                                             self._circle_radii, self._weight, collision_free=flags)
        return None if bi[0] < 0 else int(bi[0])
