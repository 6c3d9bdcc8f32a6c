/*
 * ORACLE -- test infrastructure, NOT product code.
 *
 * CPU restatement of the reference's RK4 / Pacejka hot path
 * (/root/reference/libs/vehicle_model/vehicle_model.py:220-445), used only as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under python-motionplanning_amd/ may include, link or load it.
 *
 * Parity pinning: validated against golden vectors produced by importing the
 * reference itself in the build container (tests/golden/generate_golden.py,
 * tests/test_oracle_golden.py).  The reference has no tests of its own.
 */
#ifndef VDYN_ORACLE_H
#define VDYN_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same fields as VehicleParameters (vehicle_model.py:17-61) that the path reads,
 * plus g (vehicle_model.py:230).  Pacejka D is NOT here: the reference
 * overwrites it with mu_max on every call (vehicle_model.py:232-235). */
typedef struct OracleParams {
    double m, a, b, Izz, Jw, hg, T, wL, wR, rw, g;
    double B[4], C[4]; /* FL, FR, RL, RR */
} OracleParams;

/* Gains of the two controllers (stanley_controller.py:40-47,140-143; values at drive.py:71-85). */
typedef struct OracleCtrlParams {
    double k, k_soft, max_steer, lookahead, deadband; /* Stanley */
    double kp, ki, kd;                                /* longitudinal PID */
} OracleCtrlParams;

#define ORACLE_DECL(S, REAL)                                                                   \
    void oracle_planar_model_##S(const OracleParams *p, const REAL *state,                     \
                                 const REAL *tire_torques, const REAL *mu_max,                 \
                                 const REAL *delta, REAL ax_prev, REAL ay_prev,                \
                                 REAL *state_dot, REAL *aux, REAL *outputs, REAL *acc);        \
    void oracle_planar_model_RK4_##S(const OracleParams *p, double dt, const REAL *state,      \
                                     const REAL *tire_torques, const REAL *mu_max,             \
                                     const REAL *delta, REAL ax_prev, REAL ay_prev,            \
                                     REAL *state_update, REAL *state_dot, REAL *outputs,       \
                                     REAL *acc);                                               \
    int oracle_rollout_##S(const OracleParams *p, long n, int H, double dt,                    \
                           const REAL *state0, const REAL *ctrl, int k, int layout,            \
                           const int *path_id, int P, const double *mu4, REAL *terminal,       \
                           REAL *traj, int traj_stride, int nthreads);                         \
    int oracle_rollout_spiral_##S(const OracleParams *p, long n, int H, double dt, const REAL *state0,  \
                                  const REAL *spiral, double wheelbase, double max_steer,               \
                                  double torque_all, const double *mu4, REAL *terminal, REAL *traj,     \
                                  int traj_stride, REAL *delta_out, int nthreads);                      \
    void oracle_stanley_control_##S(const OracleCtrlParams *g, const REAL *wp, int W, int stride,       \
                                    REAL x, REAL y, REAL yaw, REAL v, REAL *out);                       \
    void oracle_long_control_##S(const OracleCtrlParams *g, REAL desired, REAL current, REAL prev,      \
                                 REAL total, REAL dt, REAL *out);                                       \
    int oracle_closed_loop_##S(const OracleParams *p, const OracleCtrlParams *g, long n, int H,         \
                               double dt, int ctrl_every, int phase, const REAL *state0,                \
                               const REAL *cstate0, const REAL *wp, int Wmax, const int *wcount,        \
                               const int *path_id, int P, REAL *terminal, REAL *cstate, REAL *log,      \
                               int nthreads);                                                           \
    int oracle_collision_check_##S(const REAL *x, const REAL *y, const REAL *yaw, int L,                \
                                   long point_stride, const REAL *obst, int M, const double *offsets,  \
                                   const double *radii, int nc);                                        \
    int oracle_select_best_path_##S(int E, int P, int L, const REAL *x, const REAL *y, const REAL *yaw, \
                                    long ego_stride, long path_stride, long point_stride,              \
                                    const REAL *obst, int M, long obst_ego_stride,                     \
                                    const double *offsets, const double *radii, int nc,                \
                                    const REAL *goal, double weight, const int *validity,              \
                                    int *collision_free, int *best_idx, REAL *best_score, int nthreads); \
    int oracle_closest_index_##S(const REAL *px, const REAL *py, int n, REAL ex, REAL ey, REAL *len);   \
    int oracle_goal_index_##S(const REAL *px, const REAL *py, int n, REAL lookahead, REAL closest_len,  \
                              int closest_index);                                                       \
    void oracle_goal_state_set_##S(const REAL *px, const REAL *py, int n, int goal_index, REAL goal_v,  \
                                   const REAL *ego, int P, REAL path_offset, REAL *out);                \
    REAL oracle_spiral_objective_##S(const REAL *p, REAL xf, REAL yf, REAL tf, REAL *grad);             \
    void oracle_sample_spiral_##S(const REAL *p, REAL *x, REAL *y, REAL *t);                            \
    void oracle_transform_path_##S(const REAL *x, const REAL *y, const REAL *t, int L, const REAL *ego, \
                                   REAL *gx, REAL *gy, REAL *gt);                                       \
    int oracle_interpolate_waypoints_##S(const REAL *x, const REAL *y, int L, REAL v, REAL res,         \
                                         REAL *out, int max_rows);                                      \
    int oracle_mpc_argmin_##S(const OracleParams *p, int E, int C, int H, double dt,           \
                              const REAL *ego, const REAL *cand, const REAL *goal,             \
                              REAL w_delta, REAL *best_cost, int *best_idx, REAL *cost_all,    \
                              int nthreads);

ORACLE_DECL(f64, double)
ORACLE_DECL(f32, float)

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
