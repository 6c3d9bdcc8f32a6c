/*
 * vdyn.h -- C ABI of libvdyn_hip.so: MI355X (gfx950) batched 7-DoF planar
 * vehicle model with Pacejka tires and its RK4 step.
 *
 * This is the drop-in boundary for ONE hot path of earasteh/Python-Motionplanning:
 *
 *   libs/vehicle_model/vehicle_model.py:220-425  VehicleModel.planar_model
 *   libs/vehicle_model/vehicle_model.py:427-445  VehicleModel.planar_model_RK4
 *   libs/vehicle_model/drive.py:141-143          the caller (zero-order-hold loop)
 *
 * The reference is pure Python, so "what its FFI would bind" is a ctypes
 * binding of these symbols (INTEGRATION.md shows the stub).  Plain pointers and
 * sizes only; no exceptions cross the ABI; every function returns VDYN_OK (0)
 * or a negative VDYN_ERR_* code, with text from vdyn_last_error().
 *
 * Data layout (all arrays contiguous, struct-of-arrays over rollouts):
 *   state   [10][N]  rows U, V, wz, wFL, wFR, wRL, wRR, yaw, x, y  (vehicle_model.py:224)
 *   state12 [12][N]  the 10 rows above + ax_prev, ay_prev: the two body
 *                    accelerations the caller carries step to step
 *                    (vehicle_model.py:255-258,442-443; drive.py:141)
 *   ctrl, k = 12     rows delta FL,FR,RL,RR | tire_torques FL..RR | mu_max FL..RR
 *                    (the three 4-vectors of vehicle_model.py:225-227)
 *   ctrl, k = 2      rows delta_front, torque_all: expands to delta=[d,d,0,0],
 *                    tire_torques=[t,t,t,t], mu_max = mu4 (NULL -> [1,1,1,1]),
 *                    i.e. exactly the call of drive.py:142-143
 *   outputs [18][N]  Fx x4, Fy x4, Fz x4, s x4, FxtFL, FytFL (vehicle_model.py:420-423)
 *
 * `_dev` entry points take DEVICE pointers, enqueue on `stream` (a hipStream_t
 * passed as void*; NULL = the default stream) and return without synchronising.
 * `_host` entry points take HOST pointers, stage through the handle's device
 * scratch and return after the results are in the caller's buffers.
 * The caller owns every buffer; the library never frees caller memory and never
 * writes an input.  A handle is not thread-safe; distinct handles are independent.
 * A handle also owns small device work buffers (fleet constants, the controllers' auxiliary
 * waypoint tables) that consecutive `_dev` calls reuse: enqueue the calls of one handle on ONE
 * stream at a time (or order the streams yourself); use one handle per concurrent stream.
 * Non-finite values propagate as in the reference (vx = 0 divides by zero,
 * vehicle_model.py:284-293); there is no clamping.
 */
#ifndef VDYN_H
#define VDYN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDYN_ABI_VERSION 2   /* 2: validity argument of vdyn_select_best_path_*, in/out tables of vdyn_interpolate_waypoints_* */

enum {
    VDYN_OK = 0,
    VDYN_ERR_ARG = -1,    /* bad argument (null pointer, bad size / k / layout) */
    VDYN_ERR_HIP = -2,    /* a HIP runtime call failed                          */
    VDYN_ERR_NODEV = -3,  /* no usable gfx950 device                            */
    VDYN_ERR_OOM = -4     /* device or host allocation failed                   */
};

enum {
    VDYN_CTRL_PER_ROLLOUT = 0, /* ctrl is [H][k][N], time-major                       */
    VDYN_CTRL_SHARED = 1       /* ctrl is a table [P][H][k] + path_id[N] (staged in LDS) */
};

/* Replaces: class VehicleParameters, vehicle_model.py:17-61 (the fields the path
 * reads) + g = 9.81 (vehicle_model.py:230).  Pacejka D is deliberately absent:
 * the reference overwrites it with mu_max on every call (vehicle_model.py:232-235),
 * so mu_max is an explicit input here and parameters are never mutated. */
typedef struct VdynParams {
    double m, a, b, Izz, Jw, hg, T, wL, wR, rw, g;
    double B[4], C[4]; /* Pacejka B, C for FL, FR, RL, RR (vehicle_model.py:41-54) */
} VdynParams;

/* Replaces: the gains StanleyController.__init__ / LongitudinalController.__init__ take
 * (stanley_controller.py:8-10,40-47,139-143; values used: drive.py:56,71-85) and the
 * steering-filter constant 1e-5 / (2 * 0.001) of drive.py:137. */
typedef struct VdynCtrlGains {
    double k, k_soft, max_steer, lookahead, deadband; /* Stanley */
    double kp, ki, kd;                                /* longitudinal PID */
    double filter_gain;                               /* x_del <- (1 - g) x_del + g delta */
} VdynCtrlGains;

typedef struct VdynHandle VdynHandle;

int vdyn_abi_version(void);
/* Identity of the code objects in this library: the first 16 hex digits of the SHA-256 of the sources and compiler
 * flags it was built from (python-motionplanning_amd/_build.py, source_hash()).  Measurement only:
 * profiles/summarize*.py store it beside the counters they condense, and bench.py reports a committed counter
 * summary as stale when it was taken on another build.  No reference counterpart. */
const char *vdyn_build_id(void);
int vdyn_device_count(void);

/* VehicleParameters() with its default arguments (vehicle_model.py:18-22). */
void vdyn_params_default(VdynParams *p);

/* Replaces: VehicleModel.__init__ (vehicle_model.py:69-95) as far as the path
 * needs it -- dt is per call here.  Copies *p.  device = HIP device ordinal. */
int vdyn_create(const VdynParams *p, int device, VdynHandle **out);
int vdyn_set_params(VdynHandle *h, const VdynParams *p);
void vdyn_destroy(VdynHandle *h);
/* Text of the last error on this handle (h == NULL: of the last failed vdyn_create). */
const char *vdyn_last_error(const VdynHandle *h);
/* Options.  VDYN_OPT_LANES_PER_ROLLOUT: how vdyn_rollout_* maps rollouts to lanes.
 *   1 (default)  one lane per rollout: results do not depend on the batch a rollout is in;
 *   4            wheel-parallel, four lanes per rollout: 2-3x shorter serial chain, meant for
 *                small N where the chip is mostly empty; agrees with 1 to rounding
 *                (different summation order of the four tire forces), not bit for bit;
 *   0            automatic: 4 when N <= 16384 (fp32) / 32768 (fp64), else 1.
 * VDYN_OPT_STATE_ROWS: rows of the state arrays of vdyn_rollout_f32_* (state0 and terminal).
 *   12 (default) U V wz wFL wFR wRL wRR yaw x y ax_prev ay_prev;
 *   22           the same + rows 12..21 = the compensation terms of rows 0..9: the fp32 state accumulation
 *                s <- s + h/6 (K1 + 2 K2 + 2 K3 + K4) (vehicle_model.py:438) is then a compensated (Kahan) sum --
 *                what rounding loses at one step is carried in row 12 + i and given back at the next.  BASELINE's
 *                second metric, the fp32 max-abs state error, is dominated by exactly that rounding at |x| ~ 100 m
 *                (4e-4 after 200 steps, 9e-4 after 1000); compensated it stays near 3e-5 for +3 % time.  Start with
 *                zeros in rows 12..21; a terminal state fed back as state0 continues the very same sum, so
 *                rollout(a) then rollout(b) equals rollout(a + b) bit for bit, as with 12 rows.  fp32,
 *                lane-per-rollout, fitted tire chain only (VDYN_ERR_ARG otherwise); trajectories keep 12 rows.   */
enum { VDYN_OPT_LANES_PER_ROLLOUT = 1, VDYN_OPT_STATE_ROWS = 2 };
int vdyn_set_option(VdynHandle *h, int option, int value);
/* hipStreamSynchronize(stream) for callers that have no HIP binding of their own. */
int vdyn_stream_synchronize(VdynHandle *h, void *stream);

/* ---- planar_model: one derivative evaluation, vehicle_model.py:220-425 -------------
 * state [10][n], ctrl12 [12][n], acc_prev [2][n] (ax_prev, ay_prev)
 * -> state_dot [10][n], aux [4][n] = vx, vy, ax, ay (:410-416; nullable),
 *    outputs [18][n] (nullable), acc [2][n] = axc, ayc (:413-414).                     */
int vdyn_planar_model_f64_dev(VdynHandle *h, int64_t n, const double *state, const double *ctrl12,
                              const double *acc_prev, double *state_dot, double *aux,
                              double *outputs, double *acc, void *stream);
int vdyn_planar_model_f32_dev(VdynHandle *h, int64_t n, const float *state, const float *ctrl12,
                              const float *acc_prev, float *state_dot, float *aux,
                              float *outputs, float *acc, void *stream);
int vdyn_planar_model_f64_host(VdynHandle *h, int64_t n, const double *state, const double *ctrl12,
                               const double *acc_prev, double *state_dot, double *aux,
                               double *outputs, double *acc);
int vdyn_planar_model_f32_host(VdynHandle *h, int64_t n, const float *state, const float *ctrl12,
                               const float *acc_prev, float *state_dot, float *aux,
                               float *outputs, float *acc);

/* ---- planar_model_RK4: one RK4 step, vehicle_model.py:427-445 ----------------------
 * state_in [12][n], ctrl [k][n] (k = 2 or 12), dt = VehicleModel.dt (:428)
 * -> state_out [12][n] (rows 10, 11 = RK4-averaged axc, ayc, :442-443; may alias
 *    state_in), state_dot [10][n] (:440; nullable), outputs [18][n] (:441; nullable).  */
int vdyn_step_f64_dev(VdynHandle *h, int64_t n, const double *state_in, const double *ctrl, int k,
                      double dt, const double *mu4, double *state_out, double *state_dot,
                      double *outputs, void *stream);
int vdyn_step_f32_dev(VdynHandle *h, int64_t n, const float *state_in, const float *ctrl, int k,
                      double dt, const double *mu4, float *state_out, float *state_dot,
                      float *outputs, void *stream);
int vdyn_step_f64_host(VdynHandle *h, int64_t n, const double *state_in, const double *ctrl, int k,
                       double dt, const double *mu4, double *state_out, double *state_dot,
                       double *outputs);
int vdyn_step_f32_host(VdynHandle *h, int64_t n, const float *state_in, const float *ctrl, int k,
                       double dt, const double *mu4, float *state_out, float *state_dot,
                       float *outputs);

/* ---- rollout: H RK4 steps in ONE launch, the loop of drive.py:114,141-143 -----------
 * state0 [12][n]; ctrl per `layout` (VDYN_CTRL_*): [H][k][n], or table [P][H][k]
 * with path_id [n] (values in [0,P)); mu4 = 4 host doubles used when k = 2 (NULL -> 1).
 * -> terminal [12][n]; traj (nullable) [H / traj_stride][12][n] = the state after
 *    steps traj_stride, 2*traj_stride, ...  One trajectory ROW (12 n values) may span at most
 *    2^31 bytes (fp32: n <= 44.7 M, fp64: n <= 22.3 M; VDYN_ERR_ARG beyond -- split the batch):
 *    the kernels address a row with a 64-bit base and 32-bit offsets.  The same holds for
 *    vdyn_rollout_spiral_*, vdyn_rollout_fleet_* and, with 16 / 45 values per vehicle, for the
 *    log / DataLog rows of vdyn_closed_loop_*.                                          */
int vdyn_rollout_f64_dev(VdynHandle *h, int64_t n, int32_t H, const double *state0,
                         const double *ctrl, int k, int layout, const int32_t *path_id, int32_t P,
                         double dt, const double *mu4, double *terminal, double *traj,
                         int32_t traj_stride, void *stream);
int vdyn_rollout_f32_dev(VdynHandle *h, int64_t n, int32_t H, const float *state0,
                         const float *ctrl, int k, int layout, const int32_t *path_id, int32_t P,
                         double dt, const double *mu4, float *terminal, float *traj,
                         int32_t traj_stride, void *stream);
int vdyn_rollout_f64_host(VdynHandle *h, int64_t n, int32_t H, const double *state0,
                          const double *ctrl, int k, int layout, const int32_t *path_id, int32_t P,
                          double dt, const double *mu4, double *terminal, double *traj,
                          int32_t traj_stride);
int vdyn_rollout_f32_host(VdynHandle *h, int64_t n, int32_t H, const float *state0,
                          const float *ctrl, int k, int layout, const int32_t *path_id, int32_t P,
                          double dt, const double *mu4, float *terminal, float *traj,
                          int32_t traj_stride);

/* ---- lattice-driven rollout: every rollout steers along its own cubic spiral -------------------
 * The conformal-lattice planner gives every (ego, lateral offset) a cubic spiral kappa(s) =
 * a + b s + c s^2 + d s^3 from PathOptimizer.optimize_spiral (path_optimizer.py:31-88); spiral [n][3]
 * = (p1, p2, sf) per rollout is exactly the `params` output of vdyn_plan_lattice_* ([E][P][3], rollout
 * r = ego r / P, path r % P) and is mapped to (a, b, c, d) by path_optimizer.py:149-154.  Step t
 * (0-based) applies  delta_t = clip(atan(wheelbase * kappa(min(U0 t dt, sf))), +-max_steer)  on both
 * front wheels (drive.py:143 pattern: [d, d, 0, 0], torques [torque x4], mu_max = mu4 or ones),
 * U0 = the rollout's initial speed state0[0]: the kinematic-bicycle steering angle of the planned
 * curvature at the arc length a vehicle holding its speed has covered.
 * The clip is stanley_controller.py:128's; max_steer >= pi/2 disables it.  No control array at all.
 * -> terminal [12][n], traj (nullable) [H / traj_stride][12][n] as vdyn_rollout_*.                  */
int vdyn_rollout_spiral_f64_dev(VdynHandle *h, int64_t n, int32_t H, const double *state0, const double *spiral,
                                double wheelbase, double max_steer, double torque, double dt, const double *mu4,
                                double *terminal, double *traj, int32_t traj_stride, void *stream);
int vdyn_rollout_spiral_f32_dev(VdynHandle *h, int64_t n, int32_t H, const float *state0, const float *spiral,
                                double wheelbase, double max_steer, double torque, double dt, const double *mu4,
                                float *terminal, float *traj, int32_t traj_stride, void *stream);
int vdyn_rollout_spiral_f64_host(VdynHandle *h, int64_t n, int32_t H, const double *state0, const double *spiral,
                                 double wheelbase, double max_steer, double torque, double dt, const double *mu4,
                                 double *terminal, double *traj, int32_t traj_stride);
int vdyn_rollout_spiral_f32_host(VdynHandle *h, int64_t n, int32_t H, const float *state0, const float *spiral,
                                 double wheelbase, double max_steer, double torque, double dt, const double *mu4,
                                 float *terminal, float *traj, int32_t traj_stride);

/* ---- rollout of a heterogeneous fleet ------------------------------------------------------
 * As vdyn_rollout_*, but every rollout has its own vehicle class: classes [V] (HOST array of
 * VdynParams, 1 <= V <= 256: different masses, geometry, Pacejka B / C ...), vehicle_id [n]
 * (values in [0, V)).  The classes' constants are staged through LDS and each lane keeps its
 * class's set in registers.  The handle's own parameters are not used.                        */
int vdyn_rollout_fleet_f64_dev(VdynHandle *h, int64_t n, int32_t H, const double *state0, const double *ctrl,
                               int k, int layout, const int32_t *path_id, int32_t P, const VdynParams *classes,
                               int32_t V, const int32_t *vehicle_id, double dt, const double *mu4,
                               double *terminal, double *traj, int32_t traj_stride, void *stream);
int vdyn_rollout_fleet_f32_dev(VdynHandle *h, int64_t n, int32_t H, const float *state0, const float *ctrl,
                               int k, int layout, const int32_t *path_id, int32_t P, const VdynParams *classes,
                               int32_t V, const int32_t *vehicle_id, double dt, const double *mu4,
                               float *terminal, float *traj, int32_t traj_stride, void *stream);
int vdyn_rollout_fleet_f64_host(VdynHandle *h, int64_t n, int32_t H, const double *state0, const double *ctrl,
                                int k, int layout, const int32_t *path_id, int32_t P, const VdynParams *classes,
                                int32_t V, const int32_t *vehicle_id, double dt, const double *mu4,
                                double *terminal, double *traj, int32_t traj_stride);
int vdyn_rollout_fleet_f32_host(VdynHandle *h, int64_t n, int32_t H, const float *state0, const float *ctrl,
                                int k, int layout, const int32_t *path_id, int32_t P, const VdynParams *classes,
                                int32_t V, const int32_t *vehicle_id, double dt, const double *mu4,
                                float *terminal, float *traj, int32_t traj_stride);

/* ---- MPC selection (BASELINE config 5): E egos x C shared candidates x H steps -------
 * Rollout (e, c) starts from ego e and applies candidate c's k = 2 controls.
 *   cost[e][c] = ||(x_T, y_T) - goal_e||_2 + w_delta * sum_t delta[t][c]^2
 * (terminal-distance score after collision_checker.py:175).  Candidates whose cost
 * is not finite are disqualified (collision_checker.py:190-191); the winner is the
 * lowest cost, lowest index on ties (strict '<' scan, :194-196); no finite
 * candidate -> best_idx = -1, best_cost = +inf (best_index = None, :163).
 * ego [12][E], cand [H][2][C], goal [2][E]
 * -> best_cost [E], best_idx [E], cost_all (nullable) [E][C].                           */
int vdyn_mpc_argmin_f32_dev(VdynHandle *h, int32_t E, int32_t C, int32_t H, const float *ego,
                            const float *cand, const float *goal, double dt, double w_delta,
                            float *best_cost, int32_t *best_idx, float *cost_all, void *stream);
int vdyn_mpc_argmin_f64_dev(VdynHandle *h, int32_t E, int32_t C, int32_t H, const double *ego,
                            const double *cand, const double *goal, double dt, double w_delta,
                            double *best_cost, int32_t *best_idx, double *cost_all, void *stream);
int vdyn_mpc_argmin_f32_host(VdynHandle *h, int32_t E, int32_t C, int32_t H, const float *ego,
                             const float *cand, const float *goal, double dt, double w_delta,
                             float *best_cost, int32_t *best_idx, float *cost_all);
int vdyn_mpc_argmin_f64_host(VdynHandle *h, int32_t E, int32_t C, int32_t H, const double *ego,
                             const double *cand, const double *goal, double dt, double w_delta,
                             double *best_cost, int32_t *best_idx, double *cost_all);

/* ==== "next" row: the controllers either side of the path ================================
 * cstate [6][n] rows: x_del (steering-filter state, drive.py:77,137), total_vel_error and
 * prev_vel (PID memory, drive.py:50,52,131-134), target_vel (drive.py:51), delta and torque
 * (the held commands, drive.py:138,131).  Waypoints: wp [P][Wmax][2] = (x, y) rows of the
 * lists LateralTrackerObj.update_waypoints receives (local_planner.py:419), wcount [P] valid
 * rows per table, path_id [n] the table each vehicle tracks.  fp32: the lookahead walk runs on the
 * cumulative arc length of a table; a table with a non-finite row (which poisons the sums behind
 * it) is walked segment by segment instead, as the reference -- and the fp64 entry points -- walk
 * every table: a non-finite row only matters when the walk crosses it.                        */

/* Gains of Car.__init__ (drive.py:56,71-85): k=100, k_soft=1, max_steer=30 deg, lookahead=5,
 * deadband=0.01 (stanley_controller.py:46-47), kp=1000, ki=100, kd=0, filter 1e-5/(2*0.001). */
void vdyn_ctrl_gains_default(VdynCtrlGains *g);

/* ---- one controller update for n vehicles -------------------------------------------------
 * Replaces: StanleyController.stanley_control (stanley_controller.py:78-129, incl.
 * get_lookahead_index :56-76), LongitudinalController.long_control (:138-159) and the filter
 * of drive.py:137-138, i.e. the `i % 10 == 0` block of drive.py:128-138.
 * state12 [12][n] (rows x, y, yaw, U are read), cstate_in [6][n]
 * -> cstate_out [6][n] (may alias cstate_in), out [3][n] = limited Stanley steering angle
 *    (before the filter), target index, crosstrack error (stanley_controller.py:129).        */
int vdyn_controller_update_f64_dev(VdynHandle *h, const VdynCtrlGains *g, int64_t n, const double *state12,
                                   const double *cstate_in, const double *wp, int32_t Wmax,
                                   const int32_t *wcount, const int32_t *path_id, int32_t P, double dt,
                                   double *cstate_out, double *out, void *stream);
int vdyn_controller_update_f32_dev(VdynHandle *h, const VdynCtrlGains *g, int64_t n, const float *state12,
                                   const float *cstate_in, const float *wp, int32_t Wmax,
                                   const int32_t *wcount, const int32_t *path_id, int32_t P, double dt,
                                   float *cstate_out, float *out, void *stream);
int vdyn_controller_update_f64_host(VdynHandle *h, const VdynCtrlGains *g, int64_t n, const double *state12,
                                    const double *cstate_in, const double *wp, int32_t Wmax,
                                    const int32_t *wcount, const int32_t *path_id, int32_t P, double dt,
                                    double *cstate_out, double *out);
int vdyn_controller_update_f32_host(VdynHandle *h, const VdynCtrlGains *g, int64_t n, const float *state12,
                                    const float *cstate_in, const float *wp, int32_t Wmax,
                                    const int32_t *wcount, const int32_t *path_id, int32_t P, double dt,
                                    float *cstate_out, float *out);

/* ---- closed-loop rollout: H sub-steps of Car.drive (drive.py:114-151) minus the planner -----
 * Controllers fire when (phase + t) % ctrl_every == 0 (drive.py:128: ctrl_every = 10) and hold
 * their commands in between; every sub-step is the RK4 step of vdyn_step_* with
 * delta=[d,d,0,0], torques=[t,t,t,t], mu_max=[1,1,1,1] (drive.py:141-143).
 * -> terminal [12][n], cstate_out [6][n], log (nullable) [H][16][n] with rows state12, delta,
 *    torque, target index, crosstrack error; datalog (nullable) [H][45][n] = the 45 columns
 *    the reference writes into Car.DataLog per sub-step (drive.py:145-151, names
 *    plots.py:19-27): t = (phase + step) * dt, state x10, state_dot x10, delta, torque x4,
 *    outputs x18, crosstrack error.
 * The target index and crosstrack error the log rows repeat between controller updates are not part of
 * cstate: a launch whose `phase` is not a multiple of ctrl_every logs index -1 / error 0 until its first
 * update (the trajectory itself is unaffected: commands ARE carried).  Chain launches at multiples of
 * ctrl_every (Car.drive's frames are: 100 = 10 x 10) when the log columns matter.                  */
int vdyn_closed_loop_f64_dev(VdynHandle *h, const VdynCtrlGains *g, int64_t n, int32_t H, int32_t ctrl_every,
                             int32_t phase, const double *state0, const double *cstate_in, const double *wp,
                             int32_t Wmax, const int32_t *wcount, const int32_t *path_id, int32_t P,
                             double dt, double *terminal, double *cstate_out, double *log, double *datalog,
                             void *stream);
int vdyn_closed_loop_f32_dev(VdynHandle *h, const VdynCtrlGains *g, int64_t n, int32_t H, int32_t ctrl_every,
                             int32_t phase, const float *state0, const float *cstate_in, const float *wp,
                             int32_t Wmax, const int32_t *wcount, const int32_t *path_id, int32_t P,
                             double dt, float *terminal, float *cstate_out, float *log, float *datalog,
                             void *stream);
int vdyn_closed_loop_f64_host(VdynHandle *h, const VdynCtrlGains *g, int64_t n, int32_t H, int32_t ctrl_every,
                              int32_t phase, const double *state0, const double *cstate_in, const double *wp,
                              int32_t Wmax, const int32_t *wcount, const int32_t *path_id, int32_t P,
                              double dt, double *terminal, double *cstate_out, double *log, double *datalog);
int vdyn_closed_loop_f32_host(VdynHandle *h, const VdynCtrlGains *g, int64_t n, int32_t H, int32_t ctrl_every,
                              int32_t phase, const float *state0, const float *cstate_in, const float *wp,
                              int32_t Wmax, const int32_t *wcount, const int32_t *path_id, int32_t P,
                              double dt, float *terminal, float *cstate_out, float *log, float *datalog);

/* ==== "next" row: collision check + best-path selection ===================================
 * Replaces: CollisionChecker.collision_check (collision_checker.py:32-117, one call per path;
 * the process pool of local_planner.py:369-374 becomes lanes) and
 * CollisionChecker.select_best_path_index (collision_checker.py:134-203), for E egos x P
 * candidate paths of L points each.
 * Circles of radius radii[c] sit at offsets[c] along the heading of every path point (:88-89;
 * drive.py:25-26: offsets -1, 1, 3 m, radii 1.5 m); a path is in collision when any obstacle
 * point lies strictly inside any circle (:104-106).  score = ||end - goal|| + weight *
 * sum over colliding paths j of ||end_i - end_j|| (:175,:183-186), inf for colliding paths
 * (:190-191); best = lowest score, lowest index on ties (:194-196); none free -> -1 (:163).
 *   _dev : x, y, yaw are separate pointers with strides (in elements): path point (e, p, j) is
 *          at [e*ego_stride + p*path_stride + j*point_stride], so the trajectory output of
 *          vdyn_rollout_* ([L][12][N]: x = traj + 8 N, y = traj + 9 N, yaw = traj + 7 N,
 *          point_stride = 12 N, path_stride = 1, ego_stride = P) is consumed in place.
 *   _host: paths [E][P][3][L] = the reference's path lists [x_points, y_points, t_points].
 *   obst [M][2] shared by all egos (obst_per_ego = 0) or [E][M][2]; circle_offsets / radii are
 *   HOST arrays of nc <= 8 doubles; P <= 64; goal [2][E].
 *   collision_in (nullable) [E][P]: flags the caller already has (1 = free); the check is then
 *   skipped and only select_best_path_index runs (its collision_check_array argument, :134).
 *   validity (nullable) [E][P]: the `validity` output of vdyn_plan_lattice_*.  The reference drops
 *   invalid spirals from its path list BEFORE the collision check and the selection
 *   (local_planner.py:312-321,367-378): a path with validity 0 is absent -- never selectable, no
 *   proximity penalty from it, collision_free = 0.  best_idx stays an index into all P paths (the
 *   reference's best_index counts valid paths only: its value is the number of valid paths before
 *   best_idx).
 * -> collision_free [E][P] (1 = free), best_idx [E] (-1: none, the reference's None), best_score [E]. */
int vdyn_select_best_path_f64_dev(VdynHandle *h, int32_t E, int32_t P, int32_t L, const double *x,
                                  const double *y, const double *yaw, int64_t ego_stride, int64_t path_stride,
                                  int64_t point_stride, const double *obst, int32_t M, int32_t obst_per_ego,
                                  const double *circle_offsets, const double *circle_radii, int32_t nc,
                                  const double *goal, double weight, const int32_t *collision_in,
                                  const int32_t *validity, int32_t *collision_free, int32_t *best_idx,
                                  double *best_score, void *stream);
int vdyn_select_best_path_f32_dev(VdynHandle *h, int32_t E, int32_t P, int32_t L, const float *x,
                                  const float *y, const float *yaw, int64_t ego_stride, int64_t path_stride,
                                  int64_t point_stride, const float *obst, int32_t M, int32_t obst_per_ego,
                                  const double *circle_offsets, const double *circle_radii, int32_t nc,
                                  const float *goal, double weight, const int32_t *collision_in,
                                  const int32_t *validity, int32_t *collision_free, int32_t *best_idx,
                                  float *best_score, void *stream);
int vdyn_select_best_path_f64_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, const double *paths,
                                   const double *obst, int32_t M, int32_t obst_per_ego,
                                   const double *circle_offsets, const double *circle_radii, int32_t nc,
                                   const double *goal, double weight, const int32_t *collision_in,
                                   const int32_t *validity, int32_t *collision_free, int32_t *best_idx,
                                   double *best_score);
int vdyn_select_best_path_f32_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, const float *paths,
                                   const float *obst, int32_t M, int32_t obst_per_ego,
                                   const double *circle_offsets, const double *circle_radii, int32_t nc,
                                   const float *goal, double weight, const int32_t *collision_in,
                                   const int32_t *validity, int32_t *collision_free, int32_t *best_idx,
                                   float *best_score);

/* ==== "next" row: lattice generation ========================================================
 * Replaces, for E egos at once, the planning cycle of LocalPlanner.MotionPlanner up to the
 * transformed lattice (local_planner.py:362-368):
 *   get_closest_index (:25-52), get_goal_index (:85-152), get_goal_state_set (:154-275),
 *   plan_paths (:277-323) = PathOptimizer.optimize_spiral (path_optimizer.py:31-88) +
 *   sample_spiral (:131-175) + the validity test (:317-321), transform_paths (:424-470).
 * px, py [nwp]: the global path (drive.py:117: self.px, self.py); ego [3][E] rows x, y, yaw;
 * goal_v = target_vel; lookahead, P (= num_paths), path_offset: drive.py:21,24,35.
 * params_in (nullable) [E][P][3] = (p1, p2, sf) per spiral: skips the optimiser.
 * The optimiser is NOT SciPy's L-BFGS-B (path_optimizer.py:84): the same objective
 * (path_optimizer.py:183-198) over the same bounds (:78) is minimised by a projected
 * Levenberg-Marquardt iteration; results agree with the reference to the optimiser's tolerance.
 * -> closest_idx [E], goal_idx [E], closest_len (nullable) [E], goal_set [E][P][4] (x, y, t, v in
 *    the vehicle frame), params [E][P][3], paths [E][P][3][49] (rows x, y, yaw, global frame:
 *    the reference's path lists), validity [E][P] (1 = kept, :317-321), cost [E][P] (objective). */
int vdyn_plan_lattice_f64_dev(VdynHandle *h, int32_t E, const double *px, const double *py, int32_t nwp,
                              const double *ego, double goal_v, double lookahead, int32_t P, double path_offset,
                              const double *params_in, int32_t *closest_idx, int32_t *goal_idx,
                              double *closest_len, double *goal_set, double *params, double *paths,
                              int32_t *validity, double *cost, void *stream);
int vdyn_plan_lattice_f32_dev(VdynHandle *h, int32_t E, const float *px, const float *py, int32_t nwp,
                              const float *ego, double goal_v, double lookahead, int32_t P, double path_offset,
                              const float *params_in, int32_t *closest_idx, int32_t *goal_idx,
                              float *closest_len, float *goal_set, float *params, float *paths,
                              int32_t *validity, float *cost, void *stream);
int vdyn_plan_lattice_f64_host(VdynHandle *h, int32_t E, const double *px, const double *py, int32_t nwp,
                               const double *ego, double goal_v, double lookahead, int32_t P, double path_offset,
                               const double *params_in, int32_t *closest_idx, int32_t *goal_idx,
                               double *closest_len, double *goal_set, double *params, double *paths,
                               int32_t *validity, double *cost);
int vdyn_plan_lattice_f32_host(VdynHandle *h, int32_t E, const float *px, const float *py, int32_t nwp,
                               const float *ego, double goal_v, double lookahead, int32_t P, double path_offset,
                               const float *params_in, int32_t *closest_idx, int32_t *goal_idx,
                               float *closest_len, float *goal_set, float *params, float *paths,
                               int32_t *validity, float *cost);

/* Replaces: the waypoint re-interpolation of local_planner.py:395-419 (INTERP_DISTANCE_RES =
 * 0.01, :19) that feeds StanleyController.update_waypoints: for every ego, path best_idx[e] of
 * paths [E][P][3][L] is resampled to `res` spacing.
 * wp_out [E][Wmax][2] (x, y) and wcount [E] are IN / OUT: an ego with best_idx[e] < 0 (no selectable
 * path: the reference's best_index None) keeps the table it had -- the reference goes on following
 * _prev_best_path (local_planner.py:380-384) -- so pass the previous cycle's buffers back in (or
 * zero-initialised ones on the first cycle).  wcount = 0 when Wmax is too small.  These are the
 * waypoint tables vdyn_closed_loop_* / vdyn_controller_update_* take.                          */
int vdyn_interpolate_waypoints_f64_dev(VdynHandle *h, int32_t E, int32_t P, int32_t L, const double *paths,
                                       const int32_t *best_idx, double res, int32_t Wmax, double *wp_out,
                                       int32_t *wcount, void *stream);
int vdyn_interpolate_waypoints_f32_dev(VdynHandle *h, int32_t E, int32_t P, int32_t L, const float *paths,
                                       const int32_t *best_idx, double res, int32_t Wmax, float *wp_out,
                                       int32_t *wcount, void *stream);
int vdyn_interpolate_waypoints_f64_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, const double *paths,
                                        const int32_t *best_idx, double res, int32_t Wmax, double *wp_out,
                                        int32_t *wcount);
int vdyn_interpolate_waypoints_f32_host(VdynHandle *h, int32_t E, int32_t P, int32_t L, const float *paths,
                                        const int32_t *best_idx, double res, int32_t Wmax, float *wp_out,
                                        int32_t *wcount);

/* ==== diagnostics =============================================================================
 * Per-lane non-finite status of a [rows][n] array (terminal states, trajectories ...): status[i] = 1 when
 * any row of column i is inf or NaN.  The reference propagates inf / NaN and NumPy raises a RuntimeWarning
 * (division by a zero wheel speed, vehicle_model.py:284-293); this is the batched counterpart of that
 * warning.  count (nullable, HOST int64): number of flagged lanes (the _dev form then synchronises).   */
int vdyn_nonfinite_lanes_f64_dev(VdynHandle *h, int32_t rows, int64_t n, const double *x, int32_t *status,
                                 int64_t *count, void *stream);
int vdyn_nonfinite_lanes_f32_dev(VdynHandle *h, int32_t rows, int64_t n, const float *x, int32_t *status,
                                 int64_t *count, void *stream);
int vdyn_nonfinite_lanes_f64_host(VdynHandle *h, int32_t rows, int64_t n, const double *x, int32_t *status,
                                  int64_t *count);
int vdyn_nonfinite_lanes_f32_host(VdynHandle *h, int32_t rows, int64_t n, const float *x, int32_t *status,
                                  int64_t *count);

/* The per-handle tire fit of the fp32 step (host arithmetic only: no device, no handle needed).
 *   sin(C atan x) / x = c W_C(c),  c = 1 / sqrt(1 + x^2),  W_C(c) = sin(C acos c) / sqrt(1 - c^2)
 * W_C -- the Chebyshev polynomial of the second kind U_{C-1} continued to non-integer C -- is analytic on (-1, 1],
 * so one degree-8 polynomial in c covers every slip.  coef [9] (HOST, out): its coefficients, highest degree
 * first, as the library computes them for a handle whose wheel has shape factor C (csrc/vdyn_kernels.hip,
 * fit_tire_wheel: interpolation at the Chebyshev nodes of [0, 1], rounded to float).  Returns VDYN_OK if the fp32
 * Horner evaluation passed the library's own check (|error of sin(C atan x)| <= 5e-7 for every x and relative
 * error of sin(C atan x) / x <= 5e-7 for x <= sqrt(3)) -- the fp32 lane kernels of a handle use the fit only if
 * all four wheels pass and every B >= 0, and keep the atan -> sine chain otherwise -- or VDYN_ERR_ARG
 * (coef is still filled in). */
int vdyn_tire_fit_f32(double C, float *coef);
/* The same for the fp64 step: coef [17], degree 16, checked to 5e-14 (2.3e-14 for the reference's C).  The fp64 kernels carry ONE set of
 * coefficients: a handle uses the fit only if its four wheels share C (the reference's parameters do,
 * vehicle_model.py:44-45). */
int vdyn_tire_fit_f64(double C, double *coef);

/* Device self-test of the bounded-range elementary functions the FAST step is built from
 * (csrc/vdyn_fastmath.hpp, csrc/vdyn_packed.hpp): evaluates function `fn` on x [n] (with the scalar
 * parameter `c` where one applies) -> out0 [n], out1 [n] (second result, or untouched).
 *   fn 0 atan_rcp(x, 1/x)          1 sin_0_pi(x)           2 sin_mid(x)        3 sincos_mid(x) -> (sin, cos)
 *      4 sincos_kernel(x) -> (sin, cos)        [scalar forms, fp32 and fp64]
 *      5 the tire chain of the step (csrc/vdyn_packed.hpp, pacejka_g2x2; fp64: tire_force in
 *        csrc/vdyn_device.hpp) with the fit of C = c, any x:
 *        -> (sin(c atan x), sin(c atan x) / x); an error if that C has no validated fit
 *   fp32 only, the packed step's own forms:
 *      6 sincos of an unwrapped yaw -> (sin, cos)
 *      7 small stage rotation -> (sin, cos)     8 steering sincos kernel -> (sin, cos)
 * tests/test_gpu_fastmath.py holds each to its stated accuracy against float64 libm.                  */
int vdyn_fastmath_eval_f32_dev(VdynHandle *h, int32_t fn, int64_t n, const float *x, double c, float *out0,
                               float *out1, void *stream);
int vdyn_fastmath_eval_f64_dev(VdynHandle *h, int32_t fn, int64_t n, const double *x, double c, double *out0,
                               double *out1, void *stream);
int vdyn_fastmath_eval_f32_host(VdynHandle *h, int32_t fn, int64_t n, const float *x, double c, float *out0,
                                float *out1);
int vdyn_fastmath_eval_f64_host(VdynHandle *h, int32_t fn, int64_t n, const double *x, double c, double *out0,
                                double *out1);

/* ==== multi-GPU exchange of terminal blocks without a collective kernel ====================
 * The path's only exchange step (north_star: "all-gather ... of final trajectories only") moves one
 * small block per rank ([12][n_local] fp32: 393 KB at 8192 rollouts, 3.1 MB at 65536).  RCCL does it
 * with a copy kernel that shares the CUs with the next rollout.  These entry points do it with
 * copies instead: every rank owns a slot buffer [world][block], exports it once (hipIpcGetMemHandle),
 * opens its peers' buffers (hipIpcOpenMemHandle) and, per step, pushes its block into slot `rank`
 * of every buffer with hipMemcpyAsync on the handle's own copy streams (one per destination, up to
 * eight: the copies to different peers travel at the same time, over different xGMI links), ordered
 * behind the compute stream by an event.  One process per GPU; the 64-byte handles travel through whatever side
 * channel the caller has (torch.distributed.all_gather_object in distributed.PeerExchange).
 * There is no counterpart in the reference (its only parallelism is the process pool of
 * local_planner.py:369-374).                                                                    */
typedef struct VdynIpcHandle { unsigned char bytes[64]; } VdynIpcHandle;

/* hipMalloc `bytes` on the handle's device and export it. */
int vdyn_xchg_alloc(VdynHandle *h, uint64_t bytes, void **dev_ptr, VdynIpcHandle *out);
int vdyn_xchg_free(VdynHandle *h, void *dev_ptr);
/* Map a peer process's buffer (not your own: use the pointer vdyn_xchg_alloc returned). */
int vdyn_xchg_open(VdynHandle *h, const VdynIpcHandle *peer, void **peer_ptr);
int vdyn_xchg_close(VdynHandle *h, void *peer_ptr);
/* Copy `bytes` from src (this device) to dst[i] + dst_offset for i < n_dst, on the handle's copy
 * streams, after everything enqueued so far on `after_stream` (the compute stream; NULL = default).
 * Returns without waiting.  At most 64 destinations.  Any number of pushes may be in flight: they
 * follow each other on the device (destination i always uses the same in-order copy stream, so of
 * two pushes into the same slot the later one lands last).  What the CALLER must keep is the source
 * block, until the push's copies have read it: vdyn_xchg_fence orders its reuse on the device.
 * On an error, copies already queued have been waited for before the call returns.             */
int vdyn_xchg_push(VdynHandle *h, void *const *dst, int32_t n_dst, uint64_t dst_offset, const void *src,
                   uint64_t bytes, void *after_stream);
/* Make `stream` wait -- on the device; the host does not block -- until every push issued so far has
 * finished.  After it, work enqueued on `stream` may overwrite (or the allocator may hand out) the
 * sources of those pushes.  Call it once per several pushes, while keeping the sources of the pushes
 * since the previous fence alive (distributed.PeerExchange: every fourth): a stream-wait is a
 * barrier between two kernels.  A no-op when nothing was pushed.                                  */
int vdyn_xchg_fence(VdynHandle *h, void *stream);
/* Block the host until all of this handle's pushes have landed (a no-op when none is in flight). */
int vdyn_xchg_wait(VdynHandle *h);

#ifdef __cplusplus
}
#endif
#endif /* VDYN_H */
