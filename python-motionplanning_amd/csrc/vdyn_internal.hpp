// vdyn_internal.hpp -- declarations shared by the kernel launchers
// (vdyn_kernels.hip) and the C ABI (vdyn_capi.hip).  Not installed.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "../../include/vdyn.h"

namespace vdyn {

template <typename T>
struct RolloutArgs {
    int64_t n = 0;
    int H = 0;
    const T *state0 = nullptr;   // [12][n]
    const T *ctrl = nullptr;     // per layout
    int k = 2;                   // 2 or 12
    int layout = VDYN_CTRL_PER_ROLLOUT;
    const int *path_id = nullptr;
    int P = 0;
    double dt = 0;
    const double *mu4 = nullptr; // host, nullable
    T *terminal = nullptr;       // [12][n]
    T *traj = nullptr;           // nullable
    int traj_stride = 0;
    T *state_dot = nullptr;      // nullable, last step's RK4-averaged derivative [10][n]
    T *outputs = nullptr;        // nullable, last step's RK4-averaged outputs   [18][n]
    int lanes_per_rollout = 1;   // 1: lane per rollout; 4: wheel-parallel (vdyn_quad.hpp)
    int state_rows = 12;         // 22: state0 / terminal carry the compensation terms (VDYN_OPT_STATE_ROWS; fp32 only)
    const T *fleet_tab = nullptr;      // fleet rollouts: device table [V][len] of per-class constants
    const int *vehicle_id = nullptr;   // fleet rollouts: device [n], class of every rollout
    int V = 0;
};

template <typename T>
hipError_t launch_rollout(const VdynParams &p, const RolloutArgs<T> &a, hipStream_t st);

// lattice-driven rollout: spiral [n][3] = (p1, p2, sf) per rollout (the `params` output of plan_lattice)
template <typename T>
hipError_t launch_rollout_spiral(const VdynParams &p, int64_t n, int H, const T *state0, const T *spiral,
                                 double wheelbase, double tan_max, double torque, double dt, const double *mu4,
                                 T *terminal, T *traj, int traj_stride, hipStream_t st);

template <typename T>
hipError_t launch_rollout_fleet(const RolloutArgs<T> &a, bool all_small, hipStream_t st);
template <typename T>
void build_fleet_table(const VdynParams *classes, int V, const double *mu4, T *out, bool *all_small);
template <typename T>
int fleet_table_len(int V);

template <typename T>
hipError_t launch_nonfinite_lanes(int rows, int64_t n, const T *x, int *status, unsigned long long *count, hipStream_t st);
// the handle's fp32 tire fit for one wheel (vdyn_kernels.hip, fp32 translation unit): coef [kTireFitDeg + 1]
bool tire_fit_coefficients(double C, float *coef);
bool tire_fit_coefficients64(double C, double *coef);   // [kTireFitDeg64 + 1]

template <typename T>
hipError_t launch_fastmath_eval(int fn, int64_t n, const T *x, double c, T *out0, T *out1, hipStream_t st);

template <typename T>
hipError_t launch_planar_model(const VdynParams &p, int64_t n, const T *state, const T *ctrl12,
                               const T *acc_prev, T *state_dot, T *aux, T *outputs, T *acc,
                               hipStream_t st);

template <typename T>
hipError_t launch_mpc_argmin(const VdynParams &p, int E, int C, int H, const T *ego, const T *cand,
                             const T *goal, double dt, double w_delta, T *best_cost, int *best_idx,
                             T *cost_all, void *scratch /* device, mpc_scratch_bytes() */, hipStream_t st);
template <typename T>
size_t mpc_scratch_bytes(int E, int C, int H);

template <typename T>
struct ClosedLoopArgs {
    int64_t n = 0;
    int H = 0, ctrl_every = 10, phase = 0;
    const T *state0 = nullptr;    // [12][n]
    const T *cstate0 = nullptr;   // [6][n]
    const T *wp = nullptr;        // [P][Wmax][2]
    int Wmax = 0;
    const int *wcount = nullptr;  // [P]
    const int *path_id = nullptr; // [n]
    int P = 0;
    double dt = 0;
    T *terminal = nullptr;        // [12][n]
    T *cstate = nullptr;          // [6][n]
    T *log = nullptr;             // nullable [H][16][n]
    T *datalog = nullptr;         // nullable [H][45][n]: the reference's DataLog columns
    T *ctrl_out = nullptr;        // controller_update only: [3][n]
    T *aux = nullptr;             // device scratch of closed_loop_aux_bytes() bytes (nullable: plain scan)
};

template <typename T>
hipError_t launch_closed_loop(const VdynParams &p, const VdynCtrlGains &g, const ClosedLoopArgs<T> &a,
                              hipStream_t st);
template <typename T>
hipError_t launch_controller_update(const VdynCtrlGains &g, const ClosedLoopArgs<T> &a, hipStream_t st);
template <typename T>
size_t closed_loop_aux_bytes(int P, int Wmax, bool update_only);

template <typename T>
struct SelectArgs {
    int E = 0, P = 0, L = 0;
    const T *x = nullptr, *y = nullptr, *yaw = nullptr;   // element (e,p,j) at e*ego + p*path + j*point
    int64_t ego_stride = 0, path_stride = 0, point_stride = 0;
    const T *obst = nullptr;                               // [M][2] (+ e*obst_ego_stride)
    int M = 0;
    int64_t obst_ego_stride = 0;
    const double *offsets = nullptr, *radii = nullptr;     // host, nc entries
    int nc = 0;
    const T *goal = nullptr;                               // [2][E]
    double weight = 0;
    const int *collision_in = nullptr;                     // nullable [E][P]: skip the check, use these flags
    const int *validity = nullptr;                         // nullable [E][P]: 0 = path absent (dropped by the planner)
    int *collision_free = nullptr;                         // [E][P]
    int *best_idx = nullptr;                               // [E]
    T *best_score = nullptr;                               // [E]
};

template <typename T>
hipError_t launch_select_best_path(const SelectArgs<T> &a, hipStream_t st);

template <typename T>
struct LatticeArgs {
    int E = 0, P = 0, nwp = 0;
    const T *px = nullptr, *py = nullptr;   // global path [nwp]
    const T *ego = nullptr;                 // [3][E]: x, y, yaw
    double goal_v = 0, lookahead = 0, path_offset = 0;
    const T *params_in = nullptr;           // nullable [E][P][3]: skip the optimiser
    int *closest_idx = nullptr, *goal_idx = nullptr;   // [E]
    T *closest_len = nullptr;               // [E]
    T *goal_set = nullptr;                  // [E][P][4]
    T *params = nullptr;                    // [E][P][3]
    T *paths = nullptr;                     // [E][P][3][49]
    int *validity = nullptr;                // [E][P]
    T *cost = nullptr;                      // [E][P]
};

template <typename T>
hipError_t launch_plan_lattice(const LatticeArgs<T> &a, hipStream_t st);
template <typename T>
hipError_t launch_interpolate_waypoints(int E, int P, int L, const T *paths, const int *best_idx, double res,
                                        int Wmax, T *wp_out, int *wcount, hipStream_t st);

}  // namespace vdyn
