"""Host-side mirror of /root/reference/libs/vehicle_model/drive.py executed on MI355X.

``Car`` keeps the reference's constructor, attributes and ``drive(frame)`` contract
(drive.py:40-154): one call advances the simulation by one frame = ``Veh_SIM_NUM`` sub-steps and
fills ``DataLog`` with the 45 columns plots.py expects.  Where the reference runs ~500 Python
calls per frame (planner, 10 controller updates, 100 RK4 steps), this class issues four
launches: ``plan_lattice`` -> ``select_best_path`` -> ``interpolate_waypoints`` -> ``closed_loop``
(with the DataLog columns written by the kernel).

Differences, all stated: the obstacle list is an argument (the reference reads the module
global ``world.obstacle_xy``, drive.py:119); the spiral optimiser is the device's projected
Levenberg-Marquardt, not SciPy's L-BFGS-B; ``os.system('clear')``
(drive.py:153) is not reproduced.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .vehicle_model import VehicleModel, VehicleParameters

# drive.py:17-35
Veh_SIM_NUM = 100
Control_SIM_NUM = Veh_SIM_NUM / 10
NUM_PATHS = 7
PATH_OFFSET = 2
CIRCLE_OFFSETS = [-1.0, 1.0, 3.0]
CIRCLE_RADII = [1.5, 1.5, 1.5]
PATH_SELECT_WEIGHT = 10
LOOKAHEAD = 30
INTERP_DISTANCE_RES = 0.01   # local_planner.py:19

p = VehicleParameters()       # drive.py:37


class Car:
    """drive.py:40-154."""

    def __init__(self, init_x, init_y, init_yaw, px, py, pyaw, dt, obstacles=None, device=0,
                 log_frames=40):
        self.DataLog = np.zeros((Veh_SIM_NUM * log_frames, 45))       # drive.py:44 (4000 frames there)
        init_vel = 25.0                                               # drive.py:46
        self.x, self.y, self.yaw = init_x, init_y, init_yaw
        self.prev_vel = self.v = init_vel
        self.target_vel = init_vel
        self.total_vel_error = 0
        self.delta = 0.0
        self.wheelbase = 2.906
        self.max_steer = np.deg2rad(30)
        self.dt = dt
        self.ax_prev = 0
        self.ay_prev = 0
        self.state = [init_vel, 0, 0, init_vel / p.rw, init_vel / p.rw, init_vel / p.rw, init_vel / p.rw,
                      init_yaw, init_x, init_y]                      # drive.py:64-65
        self.px, self.py, self.pyaw = np.asarray(px, float), np.asarray(py, float), pyaw
        self.k, self.ksoft, self.kyaw, self.ksteer = 100, 1.0, 0, 0   # drive.py:71-74
        self.crosstrack_error = None
        self.target_id = None
        self.x_del = [0]
        self.k_v, self.k_i, self.k_d = 1000, 100, 0                   # drive.py:83-85
        self.torque_vec = [0, 0, 0, 0]
        self.obstacles = np.zeros((0, 2)) if obstacles is None else np.asarray(obstacles, float)
        self._prev_best_path = None
        self.kbm = VehicleModel(self.wheelbase, self.max_steer, self.dt, device=device)
        g = _lib.default_ctrl_gains()
        g.k, g.k_soft, g.max_steer = float(self.k), float(self.ksoft), float(self.max_steer)
        g.kp, g.ki, g.kd = float(self.k_v), float(self.k_i), float(self.k_d)
        self._gains = g
        self._waypoints = None

    def drive(self, frame):
        """One frame (drive.py:112-154): returns ``paths, best_index, best_path`` like the
        reference (paths: list of [x_points, y_points, t_points] in the global frame)."""
        vm = self.kbm
        # ---- planner, once per frame (drive.py:115-122 -> local_planner.py:350-421)
        ego = np.array([[self.x], [self.y], [self.yaw]], dtype=np.float64)
        lat = vm.plan_lattice(self.px, self.py, ego, self.target_vel, LOOKAHEAD, NUM_PATHS, PATH_OFFSET)
        keep = np.flatnonzero(lat["validity"][0])                     # local_planner.py:317-321 drops the rest
        paths = lat["paths"][:, keep]
        gi = int(lat["goal_index"][0])
        goal = np.array([[self.px[gi]], [self.py[gi]]])
        # the device treats invalid spirals as absent (vdyn_select_best_path_*'s validity argument); its index
        # counts all NUM_PATHS paths, the reference's best_index the valid ones only
        _, best, _ = vm.select_best_path(lat["paths"], self.obstacles, goal, CIRCLE_OFFSETS, CIRCLE_RADII,
                                         PATH_SELECT_WEIGHT, validity=lat["validity"])
        best_index = None if best[0] < 0 else int(np.searchsorted(keep, int(best[0])))
        if best_index is None:
            best_path = self._prev_best_path                          # local_planner.py:380-381
        else:
            best_path = paths[0, best_index]
            self._prev_best_path = best_path
        if best_path is not None:
            wp, wc = vm.interpolate_waypoints(np.asarray(best_path)[None, None], np.zeros(1, np.int32),
                                              INTERP_DISTANCE_RES, 8192)
            self._waypoints = wp[0, :int(wc[0])]                      # local_planner.py:419
        # ---- controllers every 10th sub-step + RK4 every sub-step (drive.py:128-151), one launch
        s = np.concatenate([np.asarray(self.state, float), [self.ax_prev, self.ay_prev]])[:, None]
        c = np.array([self.x_del[-1], self.total_vel_error, self.prev_vel, self.target_vel, self.delta,
                      self.torque_vec[0]], dtype=np.float64)[:, None]
        s, c, log, dl = vm.closed_loop(s, c, self._waypoints, Veh_SIM_NUM, gains=self._gains,
                                       phase=frame * Veh_SIM_NUM, log=True, datalog=True)
        lo = frame * Veh_SIM_NUM
        if lo + Veh_SIM_NUM <= len(self.DataLog):
            self.DataLog[lo:lo + Veh_SIM_NUM] = dl[:, :, 0]           # drive.py:145-151
        self.state = s[:10, 0].copy()
        self.x, self.y, self.yaw, self.v = self.state[8], self.state[9], self.state[7], self.state[0]
        self.ax_prev, self.ay_prev = s[10, 0], s[11, 0]
        self.x_del.extend(log[::10, 12, 0])                           # the filter states of this frame
        self.total_vel_error, self.prev_vel = c[1, 0], c[2, 0]
        self.delta, tau = c[4, 0], c[5, 0]
        self.torque_vec = [tau, tau, tau, tau]
        self.target_id, self.crosstrack_error = int(log[-1, 14, 0]), log[-1, 15, 0]
        path_lists = [[list(pth[0]), list(pth[1]), list(pth[2])] for pth in paths[0]]
        return path_lists, best_index, (None if best_path is None else [list(r) for r in best_path])
