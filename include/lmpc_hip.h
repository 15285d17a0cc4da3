/*
 * lmpc_hip.h -- C ABI of the MI355X-native batched LMPC solve path.
 *
 * Drop-in boundary for the per-step optimisation of MPC-Berkeley/Racing-LMPC-ROS2:
 *   lmpc::mpc::racing_mpc::RacingMPC::solve        src/mpc/racing_mpc/src/racing_mpc.cpp:209-372
 *   (problem definition                            src/mpc/racing_mpc/src/racing_mpc.cpp:31-202,442-543)
 *   SingleTrackPlanarModel::compile_dynamics       src/vehicle_dynamics_models/single_track_planar_model/src/single_track_planar_model.cpp:195-418
 *   SafeSetManager::query(SSQuery)                 src/vehicle_dynamics_models/racing_trajectory/src/safe_set.cpp:153-180
 *
 * Conventions
 *   - Plain C: pointers and sizes only.  No exception crosses this boundary: every
 *     entry point returns LMPC_OK (0) or a negative lmpc_status code; the message is
 *     kept on the handle (lmpc_last_error).
 *   - All `*_batch` array arguments are DEVICE pointers (HBM) unless a parameter is
 *     documented as host.  The caller owns every buffer; the library owns only the
 *     handle (constants, the safe-set copy, a linearisation workspace).  Launches go to
 *     the stream set by lmpc_set_stream.  After lmpc_reserve(max_batch) no `*_batch` call
 *     with batch <= max_batch allocates.
 *   - Batched layout is struct-of-arrays with the batch axis fastest:
 *         field[component][knot][batch]   ->   ((c * n_knots) + i) * batch + b
 *     so a wavefront's loads along the batch axis coalesce.
 *   - State  x = [s, e_y, e_psi, vx, vy, omega]  (base_vehicle_model.hpp:32-40),
 *     input  u = [u_lon, steer]                  (single_track_planar_model.hpp:62-66);
 *     N counts knot points: X is 6 x N, U and dU are 2 x (N-1) (racing_mpc.cpp:43-45).
 *   - A handle is not re-entrant (the reference holds traj_mutex_ across solve,
 *     racing_mpc_node.cpp:158,360); distinct handles (one per GPU) are independent.
 */
#ifndef LMPC_HIP_H_
#define LMPC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMPC_NX 6
#define LMPC_NU 2

/* return codes of the entry points */
typedef enum lmpc_status {
  LMPC_OK = 0,
  LMPC_ERR_ARGUMENT = -1,      /* null pointer, bad size, unsupported option          */
  LMPC_ERR_RUNTIME = -2,       /* HIP runtime error (message in lmpc_last_error)      */
  LMPC_ERR_UNSUPPORTED = -3    /* valid reference configuration not built yet         */
} lmpc_status;

/* per-problem status written by lmpc_solve_batch (replaces "X_optm absent from out",
 * racing_mpc.cpp:343-371 / racing_mpc_node.cpp:322-332) */
#define LMPC_SOLVE_OPTIMAL 0
#define LMPC_SOLVE_MAX_ITER 1   /* the iteration cap -- or (fp64) stopped short of the stated accuracy: the interior point reached its
                                   floor and the active-set polish was refused by a consistent held set whose multiplier steps do
                                   not settle, i.e. the problem's linear algebra is noisier than the 1e-6 contract (low speed x
                                   long horizon: the RK4 step map is unstable there), or (round 6) it stopped at its stall rule while its
                                   Newton step would still have moved the iterate by more than 1e-6 (scaled) and the polish did not
                                   verify the point; the iterate is returned, not vouched for */
#define LMPC_SOLVE_INFEASIBLE 2 /* x_ic outside [x_min, x_max] at knot 0, row residual stalls, NaN */
#define LMPC_SOLVE_UNVERIFIED 3 /* lmpc_solve_batch_mixed with lmpc_config.polish = 1 only: the fp32 iteration did not reach an
                                   answer it could verify -- polish refused, out of iterations, or infeasible by its
                                   single-precision residuals (the default two-pass solve re-solves these in fp64, which has
                                   the last word, and never reports 3) */

/* vehicle_model_factory.cpp:31-49 -- same selector names; only the first is built */
#define LMPC_MODEL_SINGLE_TRACK_PLANAR 0
#define LMPC_MODEL_KINEMATIC_BICYCLE 1
#define LMPC_MODEL_DOUBLE_TRACK_PLANAR 2
#define LMPC_INTEGRATOR_RK4 0
#define LMPC_INTEGRATOR_EULER 1

/* The ~25 scalars compile_dynamics/add_nlp_constraints read
 * (base_vehicle_model_config.hpp:30-154, single_track_planar_model.hpp:31-43). */
typedef struct lmpc_vehicle {
  int32_t model_id;      /* LMPC_MODEL_SINGLE_TRACK_PLANAR                                */
  int32_t integrator;    /* modeling.integrator_type: LMPC_INTEGRATOR_RK4 (0) | LMPC_INTEGRATOR_EULER (1)
                            (single_track_planar_model.cpp:357-368; lmpc_utils/src/utils.cpp:88-123)     */
  double m;              /* chassis.total_mass                                            */
  double Jzz;            /* chassis.moi                                                   */
  double l;              /* chassis.wheel_base                                            */
  double cg_ratio;       /* chassis.cg_ratio  (lr = cg_ratio * l)                         */
  double h;              /* chassis.cg_height                                             */
  double b;              /* chassis.b  (body width, boundary margin adds b/2)             */
  double fr;             /* chassis.fr (rolling resistance)                               */
  double kd;             /* powertrain.kd  (front drive share)                            */
  double kb;             /* front_brake.bias                                              */
  double cd;             /* aero.drag_coeff                                               */
  double Af;             /* aero.frontal_area                                             */
  double rho;            /* aero.air_density                                              */
  double cl_f;           /* aero.cl_f                                                     */
  double cl_r;           /* aero.cl_r                                                     */
  double mu;             /* single_track_planar.mu                                        */
  double Bf, Cf;         /* front_tyre.pacejka_b / pacejka_c                              */
  double Br, Cr;         /* rear_tyre.pacejka_b / pacejka_c                               */
  double Fd_max, Fb_max; /* single_track_planar.fd_max / fb_max                           */
  double Td, Tb;         /* single_track_planar.td / tb                                   */
  double max_steer;      /* steer.max_steer                                               */
  double max_steer_rate; /* steer.max_steer_rate                                          */
} lmpc_vehicle;

/* RacingMPCConfig (racing_mpc_config.hpp:37-82), numeric fields only. */
typedef struct lmpc_config {
  int32_t N;                  /* knot points                                                */
  int32_t learning;           /* 0 tracking MPC, 1 LMPC (safe-set terminal set + cost)      */
  int32_t num_ss_pts;         /* S                                                          */
  int32_t num_ss_pts_per_lap; /* K                                                          */
  int32_t max_lap_stored;
  int32_t max_iter;           /* interior-point iteration cap (<=0: default 60)             */
  int32_t polish;             /* active-set polish of the interior-point answer, the role of OSQP's polish = true
                                 (racing_mpc.cpp:90-95): 0 (default) on, < 0 off; 1: on, and lmpc_solve_batch_mixed
                                 leaves out its fp64 pass -- problems whose fp32 answer it could not verify keep
                                 status LMPC_SOLVE_UNVERIFIED (diagnostics)                    */
  int32_t reserved;           /* (keeps `tol` 8-byte aligned; set to 0)                      */
  double tol;                 /* complementarity tolerance (<=0: default 3e-14)             */
  double margin;
  double q_contour, q_heading, q_vel, q_vy, q_vyaw, q_boundary;
  double R[4];                /* row-major 2x2                                              */
  double R_d[4];
  double x_max[LMPC_NX], x_min[LMPC_NX]; /* +-INFINITY allowed                              */
  double u_max[LMPC_NU], u_min[LMPC_NU];
  double convex_hull_slack[LMPC_NX]; /* cost weights of the hull residual x_T - SS lambda (racing_mpc.cpp:493-499); a zero
                                        component leaves that component of the residual free; ALL zero is the hard
                                        equality x_T = SS lambda of :500-502, see LMPC_HARD_HULL_WEIGHT            */
  double max_vel_ref_diff;
} lmpc_config;

/* Hard convex-hull equality (all-zero convex_hull_slack, racing_mpc.cpp:500-502).  The terminal block eliminates the hull
 * residual eps = x_T - SS lambda through its weight (E^-1 in the Schur complement); the equality is the limit E^-1 -> 0,
 * taken numerically: the residual carries the weight LMPC_HARD_HULL_WEIGHT, at which the answer is within 1e-7 (scaled)
 * of the equality-constrained optimum (eps = multiplier / (2 weight); measured against the dense oracle with eps pinned
 * to zero, tests/test_gpu_mixed_lmpc.py) and the two-level elimination still has two decades of headroom (it breaks down
 * past 1e13).  A problem whose terminal state cannot reach the hull -- infeasible upstream -- is left with a residual the
 * weight does not close: scaled |eps|_inf > LMPC_HARD_HULL_RESIDUAL reports LMPC_SOLVE_INFEASIBLE.  fp64 only: the
 * single-precision and mixed entry points refuse it (LMPC_ERR_UNSUPPORTED). */
#define LMPC_HARD_HULL_WEIGHT 1e11
#define LMPC_HARD_HULL_RESIDUAL 1e-5

/* Closed track as uniform periodic tables over [0, L): sample j sits at s = j*L/M.
 * Lookup is periodic linear interpolation.  (The reference interpolates cubic
 * B-splines of a 17-column table, racing_trajectory.cpp:25-120 -- out of scope;
 * the tables are what RacingMPCNode samples at racing_mpc_node.cpp:261-266.) */
typedef struct lmpc_track {
  double L;
  int32_t M;
  int32_t reserved;
  const double* curvature;   /* device, M                                                  */
  const double* bound_left;  /* device, M  (signed lateral offset, > 0)                    */
  const double* bound_right; /* device, M  (signed lateral offset, < 0)                    */
  const double* vel;         /* device, M                                                  */
} lmpc_track;

typedef struct lmpc_handle lmpc_handle;

/* Builds the parametric problem once, as RacingMPC::RacingMPC does
 * (racing_mpc.cpp:31-202).  `device` is the HIP device ordinal. */
int lmpc_create(const lmpc_config* cfg, const lmpc_vehicle* veh, int device, lmpc_handle** out);
/* On failure *out may still be non-NULL so that lmpc_last_error(*out) can be read; destroy it. */
void lmpc_destroy(lmpc_handle* h);
const char* lmpc_last_error(const lmpc_handle* h);

/* Run subsequent launches on `hip_stream` (a hipStream_t).  A new handle uses the device's
 * default (null) stream; NULL selects it again. */
int lmpc_set_stream(lmpc_handle* h, void* hip_stream);
/* Block until everything queued on the handle's stream has finished. */
int lmpc_synchronize(lmpc_handle* h);

/* discrete_dynamics_jacobian for every stage of every problem
 * (single_track_planar_model.cpp:377-387, called at racing_mpc.cpp:173-182):
 *   X_ref [6][N][B], U_ref [2][N-1][B], T_ref [N-1][B], curvatures [N][B]  ->
 *   A [6][6][N-1][B] (row, col), Bm [6][2][N-1][B], g [6][N-1][B].                          */
int lmpc_linearize_batch(lmpc_handle* h, int32_t batch, const double* X_ref, const double* U_ref,
                         const double* T_ref, const double* curvatures, double* A, double* Bm,
                         double* g);

/* RacingMPC::solve for `batch` independent problems (racing_mpc.cpp:209-372).
 *   in : x_ic [6][B], u_ic [2][B], X_ref [6][N][B], U_ref [2][N-1][B], T_ref [N-1][B],
 *        bound_left/bound_right/curvatures/vel_ref [N][B], total_length (scalar, host value)
 *        ss_x [6][S][B], ss_j [S][B]  (learning only, already padded/truncated to S and
 *        with J - J[0] applied: racing_mpc.cpp:263-280; NULL for tracking)
 *   out: X_optm [6][N][B], U_optm [2][N-1][B], dU_optm [2][N-1][B],
 *        convex_combi_optm [S][B] (learning; may be NULL),
 *        status [B] (LMPC_SOLVE_*), iters [B] (the reference's stats["iter_count"]),
 *        kkt [4][B] optional: inf-norm of the last primal step, largest row residual,
 *        complementarity mu, boundary slack sigma.                                           */
int lmpc_solve_batch(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic,
                     const double* X_ref, const double* U_ref, const double* T_ref,
                     const double* bound_left, const double* bound_right, const double* curvatures,
                     const double* vel_ref, double total_length, const double* ss_x,
                     const double* ss_j, double* X_optm, double* U_optm, double* dU_optm,
                     double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt);

/* lmpc_solve_batch with the reference's warm-start inputs USED (racing_mpc.cpp:293-305: X_optm_ref, U_optm_ref, dU_optm_ref; in
 * the node they are the previous solution shifted by one knot, racing_mpc_node.cpp:245-254 -- the arrays lmpc_shift_batch writes).
 * X_optm_ref [6][N][B], U_optm_ref [2][N-1][B] (dU_optm_ref is implied: u_i = u_{i-1} + t_i dU_i is a row of the QP); they may alias
 * X_ref / U_ref, as they do in the node.  Before any interior point the kernel tries an ACTIVE-SET solve on the plan: the plan
 * made dynamically exact about this call's linearisation, its active bounds taken as the working set, the equality-constrained
 * QP solved and its KKT conditions verified (the polish's machinery; at most two repairs of the set).  Accepted -- 92 .. 96 % of
 * the periods of a closed loop -- the answer is the optimum the cold solve finds (same 1e-6 contract; measured 1e-11 apart) for
 * about two iterations' worth of work instead of seven; `iters` then counts the rounds (1 or 2).  Refused, the call IS
 * lmpc_solve_batch (the rounds spent are added to `iters`).  A bad plan costs time, never correctness: nothing is returned that
 * has not passed the KKT test of this problem.  fp64 tracking problem; learning handles: lmpc_solve_batch_warm_ss below.
 * With lmpc_config.polish < 0 (the polish switched off) there is no active-set machinery to try the plan with: the call is a cold
 * lmpc_solve_batch, silently -- lmpc_get_warm_accepted then reports 0 for every problem. */
int lmpc_solve_batch_warm(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                          const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                          const double* vel_ref, double total_length, const double* X_optm_ref, const double* U_optm_ref, double* X_optm,
                          double* U_optm, double* dU_optm, int32_t* status, int32_t* iters, double* kkt);

/* The same for the LEARNING problem (round 6; racing_mpc.cpp:281 `set_initial(convex_combi_, convex_combi_optm_ref)` and :293-305
 * apply to the learning controller too).  One more piece of the plan: convex_combi_optm_ref [S][B], the simplex weights of the plan's
 * terminal point ALIGNED WITH THIS CALL'S safe-set points (entry j weighs point j of ss_x / ss_idx).  Their support is the working
 * set of the simplex rows -- free where the weight is positive, held at zero elsewhere --, the weights start the multiplier steps, the
 * stage rows' working set is read off (X_optm_ref, U_optm_ref) as above; the candidate is verified against the KKT conditions of this
 * problem (simplex rows included) and repaired, or the cold solve runs.  The safe set as arrays (ss_x, ss_j; ss_idx NULL) or by
 * reference (ss_idx from lmpc_ss_query_idx_batch on this handle; ss_x, ss_j NULL).  The query returns its neighbours nearest first, so
 * from one control period to the next the POSITION of a point in the set changes: lmpc_shift_lambda_batch below carries the previous
 * solution's weights over by the identity of the points.  convex_combi_optm_ref NULL (or all zero): the cold solve.  fp64; horizons up
 * to N = 60 (longer: the cold solve).  lmpc_get_warm_accepted tells which problems took the short route. */
int lmpc_solve_batch_warm_ss(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                             const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                             const double* vel_ref, double total_length, const double* ss_x, const double* ss_j, const int32_t* ss_idx,
                             const double* X_optm_ref, const double* U_optm_ref, const double* convex_combi_optm_ref, double* X_optm,
                             double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt);

/* convex_combi_optm_ref for lmpc_solve_batch_warm_ss when the safe set goes by reference: lambda_prev [S][B] of the previous
 * solution on the codes ss_idx_prev [S][B] it was solved on -> lambda_ref [S][B] on this period's codes ss_idx.  From one period to
 * the next a support point of the optimum stays, moves one recorded sample along its lap, or two (measured in the LMPC experiment):
 * every support point (weight > 1e-9) is placed on the sample `advance` steps on when that is among the new neighbours, else on
 * the point itself, else `advance` + 1 steps on -- at most six positive entries per problem, what the solver's terminal block keeps
 * explicit; a support point none of whose candidates is in the new set drops out (the solver renormalises).  An active-set start
 * needs the support RIGHT: in the LMPC experiment this guess is (advance = 1; four repair rounds) for a quarter of the periods, and
 * the rest pay the refused rounds on top of their cold solve -- closed_loop.run_lmpc leaves it off by default.
 * No counterpart upstream: the reference hands the previous weights to OSQP by position.  All pointers DEVICE. */
int lmpc_shift_lambda_batch(lmpc_handle* h, int32_t batch, const int32_t* ss_idx_prev, const double* lambda_prev, const int32_t* ss_idx,
                            int32_t advance, double* lambda_ref);

/* accepted [batch] (DEVICE, int32): 1 where the most recent warm solve of this batch size on this handle (lmpc_solve_batch_warm,
 * lmpc_solve_batch_warm_ss) returned the active-set attempt's answer, 0 where the cold solve ran (refused attempt, or no warm kernel
 * for the configuration).  Written by the kernel itself (until round 6 callers inferred it from iters <= 4). */
int lmpc_get_warm_accepted(lmpc_handle* h, int32_t batch, int32_t* accepted);

/* Mixed precision (BASELINE configs[4]: "mixed fp32/fp64 KKT"): same arguments, layouts and fp64 arrays as
 * lmpc_solve_batch.  In fp64: the linearisation (discrete_dynamics_jacobian), the error-dynamics regression onto it when
 * lmpc_set_regression_laps is in effect, the centring of the abscissa on x_ic[0], the 2x2 pivots of the Riccati
 * recursion, the results and -- learning = 1 -- the simplex rows and the whole terminal elimination of the safe-set
 * block (racing_mpc.cpp:484-504).  In fp32: the stage records in LDS, the Riccati factor and sweeps and the stage rows --
 * half the LDS footprint, so twice the resident problems per CU where fp64 is capacity-bound.  Horizons: every N the
 * fp64 entry accepts for the tracking problem (iac_car_tracking_mpc.param.yaml ships N = 80); N <= 23 for the learning
 * problem.  A configuration without a reduced-precision kernel -- the learning problem at N >= 24 (barc_lmpc.param.yaml ships
 * N = 40: measured slower than fp64 in this layout), the hard hull equality -- is SOLVED IN FP64 by this entry (round 6;
 * it was LMPC_ERR_UNSUPPORTED): same results and statuses as lmpc_solve_batch, and lmpc_last_solve_precision() reports
 * LMPC_PRECISION_F64 for the call, so a caller iterating over horizons needs no special case and can still tell what ran
 * (lmpc_query_launch_for(h, LMPC_PRECISION_MIXED, ..) keeps answering LMPC_ERR_UNSUPPORTED for such a handle: "no mixed kernel").
 * Two passes when lmpc_config.polish = 0 (the default): the fp32 kernel ends with the active-set polish and a KKT test of
 * its answer (rows 1e-5, multipliers -1e-3, last step 1e-4, scaled); every problem whose answer did not pass -- polish
 * refused, out of iterations, infeasible by single-precision residuals -- is solved again by the fp64 kernel behind it,
 * which writes the fp64 entry's own answer and status over it (a percent of a batch).  Stated accuracy (scaled, against the fp64
 * answer; tests/tolerances.py): 1e-3 on every problem of the tracking configurations and of the learning problem on states near the
 * stored laps; on the learning workload of SURVEY.md 8(d) (random initial states) 1e-3 at the 99.99 % quantile and 5e-3 on every
 * problem -- a few problems per 32768 pass the single-precision KKT test with the weights of two or three nearly exchangeable
 * safe-set points off in the third digit (1.2 .. 3.4e-3 measured); away from the BASELINE shapes (every N, 96 / 160 points) 2e-3 on
 * every problem of 1024 per horizon, single problems up to 2.9e-3 at 4096 per horizon (profiles/r06_dispatch_sweep_4096.txt).
 * lmpc_solve_batch_f32 has no fp64 pass behind it: 1e-3 on the BASELINE shape (N = 40, every problem of 8192); over every horizon
 * from 3 to 81 at 4096 problems each (profiles/r06_dispatch_sweep_4096.txt) the worst single problems are 2.0 .. 2.2e-3 (N = 36, 75)
 * and at two horizons one problem of 4096 that the fp64 kernel solves is not solved.
 * Fit for well-scaled problems (IAC) and for the learning problem, NOT for the BARC
 * tracking problem at low speed, whose soft boundary needs complementarity below 1e-9.
 * polish < 0: one pass, the fp32 interior point's own answers (faster, a tail of problems up to 3e-2 away). */
int lmpc_solve_batch_mixed(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic,
                           const double* X_ref, const double* U_ref, const double* T_ref,
                           const double* bound_left, const double* bound_right, const double* curvatures,
                           const double* vel_ref, double total_length, const double* ss_x,
                           const double* ss_j, double* X_optm, double* U_optm, double* dU_optm,
                           double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt);

/* How many times a warm start (lmpc_solve_batch_warm, lmpc_solve_host_warm) may repair its working set before it is refused and
 * the cold start takes over: 1 .. 4 (the polish's own limit), or 0 for the default.  No counterpart upstream.  A round costs about
 * one interior-point iteration; a refused attempt has spent its rounds on top of the cold solve that follows, so it is the longest
 * job of its batch, and an accepted one the shortest.  In the closed loop at N = 20, 68 % of the attempts are accepted in the first
 * round, 95 % within two, 98 % within three, 99 % within four (N = 60: 22 / 73 / 82 / 89 %).  The default goes by the batch: 2
 * rounds while the batch is less than four times what the device holds at once (its duration is that of the longest jobs: refuse
 * early), 4 rounds beyond (throughput counts: every cold solve saved pays).  The answer does not depend on the setting -- an
 * accepted attempt is the optimum, a refused one falls back to the cold solve --; `iters` does: an accepted attempt reports its
 * rounds, a refused one its rounds + the cold solve's iterations. */
int lmpc_set_warm_rounds(lmpc_handle* h, int32_t rounds);

/* Single precision (BASELINE configs[3]: "IAC Putnam tracking MPC, N=40, ..., fp32"): the tracking problem with every
 * array in float and the interior point / Riccati recursion in fp32 (the linearisation is evaluated in fp64 and
 * rounded).  Same layouts and meaning as lmpc_solve_batch, no safe-set arguments; kkt [4][B] optional.  The abscissa
 * is carried relative to x_ic[0] inside the kernel (the QP is invariant to that shift), so a 2.8 km lap keeps its
 * resolution.  Stopping rule: complementarity <= max(tol, 2e-6), row residuals <= 1e-4, then the active-set polish in
 * fp32 (lmpc_config.polish >= 0).  There is no fp64 pass behind this entry (its arrays are float): an answer the polish
 * could not verify keeps status OPTIMAL at the interior point's accuracy.  Measured against fp64 on the IAC problem:
 * worst 4.6e-4 (scaled). */
int lmpc_solve_batch_f32(lmpc_handle* h, int32_t batch, const float* x_ic, const float* u_ic, const float* X_ref,
                         const float* U_ref, const float* T_ref, const float* bound_left, const float* bound_right,
                         const float* curvatures, const float* vel_ref, float* X_optm, float* U_optm, float* dU_optm,
                         int32_t* status, int32_t* iters, float* kkt);

/* RacingMPC::solve for ONE problem with HOST pointers in the reference's own (CasADi DM,
 * column-major) layout: X_ref is 6 x N with a knot's state contiguous, U_ref 2 x (N-1), the
 * per-knot rows are plain arrays.  Stages the data through buffers owned by the handle, runs
 * lmpc_solve_batch with batch = 1 and waits.  This is what the C++ facade (RacingMPC class,
 * racing-lmpc-ros2_amd/host/) calls.  ss_x is 6 x S column-major, ss_j has S entries
 * (learning only; NULL otherwise); convex_combi_optm may be NULL. */
int lmpc_solve_host(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref,
                    const double* U_ref, const double* T_ref, const double* bound_left,
                    const double* bound_right, const double* curvatures, const double* vel_ref,
                    double total_length, const double* ss_x, const double* ss_j, double* X_optm,
                    double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status,
                    int32_t* iters);

/* lmpc_solve_batch_warm for ONE problem with HOST pointers (layouts of lmpc_solve_host; X_optm_ref 6 x N, U_optm_ref 2 x (N-1)
 * column-major): what the facade's RacingMPC::solve calls when the warm-start keys are present and the controller has solved before. */
int lmpc_solve_host_warm(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                         const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                         const double* vel_ref, double total_length, const double* X_optm_ref, const double* U_optm_ref, double* X_optm,
                         double* U_optm, double* dU_optm, int32_t* status, int32_t* iters);

/* lmpc_solve_batch_warm_ss for ONE problem with HOST pointers (layouts of lmpc_solve_host; convex_combi_optm_ref S values): what the
 * facade's RacingMPC::solve calls for a learning controller when `convex_combi_optm_ref` is among the inputs (racing_mpc.cpp:281). */
int lmpc_solve_host_warm_ss(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                            const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                            const double* vel_ref, double total_length, const double* ss_x, const double* ss_j, const double* X_optm_ref,
                            const double* U_optm_ref, const double* convex_combi_optm_ref, double* X_optm, double* U_optm, double* dU_optm,
                            double* convex_combi_optm, int32_t* status, int32_t* iters);

/* RacingMPC(full_dynamics = true)::solve for a batch (racing_mpc.cpp:67-84: IPOPT on the problem whose dynamics rows are
 * x_{i+1} = f_d(x_i, u_i, k_i, t_i), :162-166, instead of their linearisation; the node uses it for its very first
 * solve, racing_mpc_node.cpp:299-314).  Sequential QPs over the batched kernels, globalised by a backtracking line search
 * on the l1 merit function cost + nu |dynamics defect|_1 (csrc/lmpc_sqp_kernel.hip): linearise about the iterate, solve
 * the QP, step, until the QP's own step falls below step_tol (scaled) or max_sqp QPs are spent.  A QP that is infeasible
 * about a new iterate (the step outran its linearisation) moves the iterate half way back and is linearised again, up to
 * 6 times in a row, before the problem stops with that status.  The iterate starts at
 * (X_ref, U_ref) -- the node hands its zero-input rollout, racing_mpc_node.cpp:210-235 -- with dU = 0, lambda = 0.
 * DEVICE pointers, layouts of lmpc_solve_batch.  Outputs: the iterate reached; status [B] = status of the last QP solved
 * for the problem; iters [B] = interior-point iterations summed over the QPs solved while the problem was still moving;
 * sqp_iters [B] = those QPs (steps taken + back-offs); sqp_move [B] = scaled |X_QP - X| of the last QP, the step it
 * proposed whatever part of it the line search took (converged iff <= step_tol); defect [B] =
 * |x_{i+1} - f_d(x_i, u_i)|_inf / scale_x of the iterate.  The first call for a batch size allocates a work area (like lmpc_reserve); synchronises the stream once per
 * QP (it has to know whether any problem is still moving). */
int lmpc_solve_full_dynamics_batch(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref,
                                   const double* U_ref, const double* T_ref, const double* bound_left,
                                   const double* bound_right, const double* curvatures, const double* vel_ref,
                                   double total_length, const double* ss_x, const double* ss_j, int32_t max_sqp,
                                   double step_tol, double* X_optm, double* U_optm, double* dU_optm,
                                   double* convex_combi_optm, int32_t* status, int32_t* iters, int32_t* sqp_iters,
                                   double* sqp_move, double* defect);

/* The same for ONE problem with HOST pointers (layouts of lmpc_solve_host): what the facade's
 * RacingMPC(config, model, full_dynamics = true) calls.  sqp_move / defect may be NULL. */
int lmpc_solve_full_dynamics_host(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref,
                                  const double* U_ref, const double* T_ref, const double* bound_left,
                                  const double* bound_right, const double* curvatures, const double* vel_ref,
                                  double total_length, const double* ss_x, const double* ss_j, int32_t max_sqp,
                                  double step_tol, double* X_optm, double* U_optm, double* dU_optm,
                                  double* convex_combi_optm, int32_t* status, int32_t* iters, int32_t* sqp_iters,
                                  double* sqp_move, double* defect);

/* Safe set store: SafeSetManager::add_lap / SSTrajectory::process_lap_data
 * (safe_set.cpp:116-151).  HOST pointers: laps oldest first, lap j has n_pts[j] samples,
 * x is the concatenation of the laps' [n_pts[j]][6] row-major state tables.  The library
 * builds the +-L unrolled copies and the cost-to-go J on the device. */
int lmpc_set_safe_set(lmpc_handle* h, int32_t n_laps, const int32_t* n_pts, const double* x,
                      double total_length);

/* One query, HOST pointers, for the C++ facade's SafeSetManager::query: query[2] = (s, e_y); ss_x column-major
 * 6 x S and ss_j [S] as lmpc_ss_query_batch produces them (padded, J - J[0]); *j0 (optional) = the J[0] that was
 * subtracted, so that SSResult::J = ss_j + j0 on the first *n_found points.  Synchronises the handle's stream. */
int lmpc_ss_query_host(lmpc_handle* h, const double* query, double* ss_x, double* ss_j, int32_t* n_found, double* j0);

/* Error-dynamics regression on the recorded laps: RegQuery / RegResult (safe_set.hpp:57-88),
 * SSTrajectory::query(RegQuery) (safe_set.cpp:56-114), SafeSetManager::query(RegQuery) (:182-245) -- BASELINE
 * config 5.  For every linearisation point a kernel-weighted ridge regression of the nominal model's one-step error
 * over the lap samples within dist_max of [X_ref[in_state]; U_ref[in_ctrl]] is added onto (A, B, g):
 *   K_j = 0.75/h (1 - (d_j/h)^2)^2,  M = [z_j' 1],  R_r = (M'KM + 1e-3 I)^-1 M'K y_r,
 *   A[r, in_state] += R_r[0:ns],  B[r, in_ctrl] += R_r[ns:ns+nc],  g[r] += R_r[-1],
 * with y_r,j = x_{j+1}[r] - f_d(x_j, u_j, k_j, t_{j+1} - t_j)[r], so that the corrected model's one-step error on the
 * recorded samples is the least-squares residual (it shrinks).  The reference never calls this query and two of its
 * expressions do not type-check as written; the reading taken (same index lists for every regressed row, nominal step
 * evaluated on the full recorded state, residual of the regressed row) is documented in oracle/regression.py.
 * AS WRITTEN upstream the step is dt_j = t_j - t_{j+1} < 0 (process_lap_data, safe_set.cpp:130-135: the nominal model is
 * integrated backwards) and b = -M'K y (:229-231): the correction then points away from the data -- which is consistent
 * with the query having no caller.  `as_written = 1` reproduces exactly that for comparison; the default (0) is the
 * physically meaningful regression, and only that one should be switched on in front of a solve.
 * Built for (n_in_state + n_in_ctrl, n_out) = (5, 3) -- rows vx, vy, yaw rate on (vx, vy, yaw rate; u) -- and (8, 6). */
typedef struct lmpc_regression_spec {
  int32_t n_out;       /* reg_out_state_idxs: one state index per regressed row */
  int32_t out[6];
  int32_t n_in_state;  /* reg_in_state_idxs */
  int32_t in_state[6];
  int32_t n_in_ctrl;   /* reg_in_control_idxs */
  int32_t in_ctrl[2];
  int32_t as_written;  /* 0: dt = t_{j+1} - t_j, b = +M'K y (default); 1: the reference's literal signs (see above) */
  double dist_max;     /* RegQuery::dist_max, the kernel bandwidth h */
} lmpc_regression_spec;

/* HOST pointers, laps concatenated as in lmpc_set_safe_set: x [n][6], u [n][2], k [n] (curvature), t [n] (time
 * stamps) -- the lap_.x/u/k/t of SSTrajectory.  n_laps = 0 or spec = NULL switches the regression off.  While it is
 * on, lmpc_solve_batch applies it between its linearisation and its QP kernel. */
int lmpc_set_regression_laps(lmpc_handle* h, int32_t n_laps, const int32_t* n_pts, const double* x, const double* u,
                             const double* k, const double* t, const lmpc_regression_spec* spec);

/* RegResult{A, B, C} for a batch: adds the regression onto A [6][6][N-1][B], Bm [6][2][N-1][B], g [6][N-1][B]
 * (the arrays of lmpc_linearize_batch; DEVICE pointers), linearisation points X_ref [6][N][B], U_ref [2][N-1][B]. */
int lmpc_regress_batch(lmpc_handle* h, int32_t batch, const double* X_ref, const double* U_ref, double* A, double* Bm,
                       double* g);

/* SafeSetManager::query(SSQuery) + the pad/truncate and J - J[0] of RacingMPC::solve
 * (safe_set.cpp:153-180, trajectory_kd_tree.cpp:53-63, racing_mpc.cpp:249-280):
 *   query [2][B] = (s, e_y) of X_ref[:, N-1] after abscissa alignment  ->
 *   ss_x [6][S][B], ss_j [S][B] (J - J[0]), n_found [B] (points before padding).           */
int lmpc_ss_query_batch(lmpc_handle* h, int32_t batch, const double* query, double* ss_x,
                        double* ss_j, int32_t* n_found);

/* The same query BY REFERENCE (round 5; no counterpart upstream, where the query returns copies, safe_set.cpp:153-180): instead
 * of the 7 S doubles per query of (ss_x, ss_j) the kernel leaves S int32 codes, ss_idx [S][B],
 *     code = (row of the point in the concatenated lap store of lmpc_set_safe_set) * 4 + rep,   rep = 0 / 1 / 2: the copy of the
 *            lap shifted by -L / 0 / +L (SSTrajectory::process_lap_data, :122-128);  -1: no point (nothing stored, NaN query)
 * padded with the last point's code (racing_mpc.cpp:263-272), and lmpc_solve_batch_ss_idx gathers the points from the store
 * itself: 640 B per query at S = 160 instead of 8960 B written here and read back by the QP kernel.  Same neighbours in the same
 * order as lmpc_ss_query_batch (one kernel, two output forms); the solve on them gives bit for bit the same result. */
int lmpc_ss_query_idx_batch(lmpc_handle* h, int32_t batch, const double* query, int32_t* ss_idx, int32_t* n_found);

/* lmpc_solve_batch (precision = LMPC_PRECISION_F64) or lmpc_solve_batch_mixed (LMPC_PRECISION_MIXED) for the learning problem with
 * the safe set by reference: ss_idx [S][B] from lmpc_ss_query_idx_batch ON THIS HANDLE, against the store lmpc_set_safe_set left
 * on it.  The store must be the one the query ran against: after another lmpc_set_safe_set (or without a query on this handle) the
 * call is refused with LMPC_ERR_ARGUMENT, and a code that names no row of the store is read as "no point" (-1), never out of bounds.
 * Everything else as lmpc_solve_batch. */
int lmpc_solve_batch_ss_idx(lmpc_handle* h, int32_t batch, int32_t precision, const double* x_ic, const double* u_ic, const double* X_ref,
                            const double* U_ref, const double* T_ref, const double* bound_left, const double* bound_right,
                            const double* curvatures, const double* vel_ref, double total_length, const int32_t* ss_idx, double* X_optm,
                            double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt);

/* Cold-start input preparation of RacingMPCNode::on_step_timer
 * (racing_mpc_node.cpp:210-235,261-292): U_ref = 1e-9, X_ref rolled out with the RK4
 * model and the track curvature at each knot, references sampled from the track tables,
 * vel_ref clamped to vx +- max_vel_ref_diff and the speed limit.
 *   in : x_ic [6][B], dt, speed_scale, speed_limit
 *   out: X_ref [6][N][B], U_ref [2][N-1][B], T_ref [N-1][B], bound_left, bound_right,
 *        curvatures, vel_ref [N][B]                                                         */
int lmpc_prepare_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const double* x_ic,
                       double dt, double speed_scale, double speed_limit, double* X_ref,
                       double* U_ref, double* T_ref, double* bound_left, double* bound_right,
                       double* curvatures, double* vel_ref);

/* The same cold start for the problems whose last solve failed only (status[b] != 0), in place: the arrays of the
 * others are left as they are.  What re-launching the node does for one car (racing_mpc_node.cpp:210-235), for a
 * closed-loop batch that keeps going without a host round trip: call it on the output of lmpc_shift_batch.         */
int lmpc_prepare_failed_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const double* x_ic,
                              const int32_t* status, double dt, double speed_scale, double speed_limit,
                              double* X_ref, double* U_ref, double* T_ref, double* bound_left,
                              double* bound_right, double* curvatures, double* vel_ref);

/* Warm-start shift of RacingMPCNode::on_step_timer (racing_mpc_node.cpp:245-254, 261-292): the previous
 * solution moves one knot forward, the last input is repeated, the last state is rolled out with the model
 * and the references are re-sampled at the shifted abscissa.  Per problem, `status` (may be NULL) selects the
 * previous solution (X_sol, U_sol; status 0) or the previous reference (X_old, U_old) when the solve failed
 * (racing_mpc_node.cpp:322-332).  Outputs must not alias inputs. */
int lmpc_shift_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const double* X_sol,
                     const double* U_sol, const double* X_old, const double* U_old, const int32_t* status,
                     double dt, double speed_scale, double speed_limit, double* X_ref, double* U_ref,
                     double* T_ref, double* bound_left, double* bound_right, double* curvatures,
                     double* vel_ref);

/* Plant step of RacingSimulator::step (racing_simulator.cpp:46-69,97-112): x [6][B] advanced in place by
 * `n_sub` RK4 sub-steps of dt_sim with u [2][B] held, curvature looked up at the current abscissa, abscissa
 * wrapped into [0, L). */
int lmpc_plant_step_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, double* x,
                          const double* u, double dt_sim, int32_t n_sub);

/* Everything between two solves of a closed loop in ONE launch: what RacingMPCNode::on_step_timer does with a solve's result and
 * what the simulator does with the node's command, for a batch of cars.  Per car:
 *   the input applied  -- the plan's first input U_optm[:, 0], or after a failed solve (status != 0) the first input of the plan the
 *                         solve started from, U_ref[:, 0] (racing_mpc_node.cpp:322-332); written to u_prev [2][B] (the next u_ic);
 *   the plant          -- lmpc_plant_step_batch on x [6][B] in place (racing_simulator.cpp:46-69,97-112);
 *   the next inputs    -- lmpc_shift_batch (racing_mpc_node.cpp:245-254) from the solution, or from the old plan after a failed
 *                         solve; with restart_failed != 0 a car whose solve failed is instead prepared from a cold start at its
 *                         new state (lmpc_prepare_failed_batch; racing_mpc_node.cpp:210-235) -- X_ref, U_ref, T_ref, bound_left,
 *                         bound_right, curvatures, vel_ref are updated IN PLACE (they must not alias X_optm / U_optm);
 *   bookkeeping        -- optional accumulators, any of them NULL: distance [B] += abscissa travelled (unwrapped), worst_excess [B]
 *                         = max(itself, excursion of the body beyond the track edge at the new state against the bounds of knot 0),
 *                         n_fail [B] += 1 after a failed solve, *n_accepted += number of cars whose warm start was accepted (the warm kernel's own flag; 0 after a cold solve)
 *                         (status 0 and iters <= 4; needs iters).
 * The same arithmetic as the three entry points it replaces -- bit for bit, except that the last knot of a shifted solution (one model
 * step from the knot before it) may differ by 1 - 2 ulp, the compiler contracting the inlined model differently in the two kernels
 * (tests/test_gpu_loop.py) --; a period of a closed loop is then
 * three launches -- linearisation, QP, this -- instead of ~45.  SOA result layout only.  All pointers DEVICE. */
int lmpc_loop_advance_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const int32_t* status, const int32_t* iters,
                            const double* X_optm, const double* U_optm, double* x, double* u_prev, double dt, double dt_sim,
                            int32_t n_sub, double speed_scale, double speed_limit, int32_t restart_failed, double* X_ref,
                            double* U_ref, double* T_ref, double* bound_left, double* bound_right, double* curvatures,
                            double* vel_ref, double* distance, double* worst_excess, int64_t* n_fail, uint64_t* n_accepted);

/* Layout of the RESULT arrays X_optm, U_optm, dU_optm of lmpc_solve_batch and lmpc_solve_batch_mixed AS THE CALLER INVOKES THEM
 * (inputs, status, iters, kkt and convex_combi_optm are not affected).  Nothing else follows the setting: lmpc_solve_batch_f32,
 * lmpc_solve_full_dynamics_batch (its inner QPs feed the line search and the next linearisation), the single-problem host
 * entry points (lmpc_solve_host, lmpc_solve_full_dynamics_host: column-major host arrays either way) and lmpc_shift_batch
 * (which READS a solution) always use the default layout.  LMPC_LAYOUT_SOA (default): [component][knot][batch], what batch-parallel
 * consumers (lmpc_shift_batch, lmpc_plant_step_batch, a result gather) read coalesced.  LMPC_LAYOUT_AOS: [batch][knot]
 * [component] -- per problem the reference's own DM layout (column-major 6 x N, 2 x (N-1); racing_mpc.cpp:347-349), for a
 * consumer that takes one problem's plan at a time; one wavefront's 196 results then leave as full cache lines instead of
 * eight-byte stores at stride B (measured HBM write traffic of the QP kernel: profiles/r03_*). */
#define LMPC_LAYOUT_SOA 0
#define LMPC_LAYOUT_AOS 1
int lmpc_set_output_layout(lmpc_handle* h, int32_t layout);

/* Grow the handle's device workspace (stage linearisations, 432 B per stage per problem; the list of a mixed solve's
 * unverified problems; the polish's save area, 10 N - 4 values per problem) so that no later *_batch call with
 * batch <= max_batch allocates.  lmpc_solve_batch_f32 needs none of the fp64 workspace: it grows its own fp32 workspace and
 * the save area on the first call for a batch size. */
int lmpc_reserve(lmpc_handle* h, int32_t max_batch);

/* Launch order of the QP kernel's workgroups (no counterpart upstream: a scheduling aid for closed-loop batches).  The
 * hardware starts workgroups in index order and a batch of 4096 fills the GPU twice, so the kernel's duration is that of
 * the problems that happen to start last.  With `order` -- DEVICE int32 [batch], a permutation of 0 .. batch-1 (not
 * checked: an entry outside the range skips that workgroup, a repeated one leaves another problem's outputs untouched) --
 * workgroup w solves problem order[w]; lmpc_launch_order_from_iters fills it from the iteration counts of the previous
 * solve of the same cars, longest first (runs on the handle's stream, graph-capturable).  Results per problem do not
 * depend on the order.  The order applies to lmpc_solve_batch* calls of exactly `batch` problems; calls with any other
 * batch size (and the single-problem host entry points) keep the default XCD-aware mapping.  order = NULL restores the
 * default for every size.  The pointer is kept, not copied: it must stay valid, and must not be rewritten while a solve
 * that reads it is in flight (one buffer per stream). */
int lmpc_set_launch_order(lmpc_handle* h, const int32_t* order, int32_t batch);
int lmpc_launch_order_from_iters(lmpc_handle* h, int32_t batch, const int32_t* iters, int32_t* order);

/* Library/kernel facts for harnesses: bytes of LDS one problem occupies, threads per problem. */
int lmpc_query_launch(const lmpc_handle* h, int32_t* lds_bytes_per_problem,
                      int32_t* threads_per_problem);

/* Occupancy of the QP kernel as the runtime reports it: resident problems (= wavefronts) per CU. */
int lmpc_query_residency(lmpc_handle* h, int32_t* problems_per_cu);

/* The same two facts for the kernel a given entry point launches: LMPC_PRECISION_F64 (lmpc_solve_batch),
 * LMPC_PRECISION_F32 (lmpc_solve_batch_f32) or LMPC_PRECISION_MIXED (the fp32 iteration of lmpc_solve_batch_mixed; its fp64
 * second pass is the F64 kernel).  LMPC_ERR_UNSUPPORTED when that entry point has no kernel for the handle's (N, num_ss_pts). */
#define LMPC_PRECISION_F64 0
#define LMPC_PRECISION_F32 1
#define LMPC_PRECISION_MIXED 2
int lmpc_query_launch_for(lmpc_handle* h, int32_t precision, int32_t* lds_bytes_per_problem, int32_t* problems_per_cu);

/* Wavefronts per problem of the cold fp64 tracking solve (lmpc_solve_batch; no counterpart upstream): 0 (default) the library's choice
 * by horizon, 1 one wavefront per problem (every configuration), 2 two wavefronts per problem on one LDS record -- the inequality
 * rows dealt to 128 lanes, the Riccati chains on the first wave (csrc/lmpc_solve_w2.hip.h): built for N >= 24; where it is not
 * built (N <= 23, the learning problem, reduced precision, the warm start) the setting is ignored.  Same algorithm, same
 * contract; the answers differ from the one-wave kernel's by the order of a few wave-wide sums (1e-12). */
int lmpc_set_waves_per_problem(lmpc_handle* h, int32_t waves);

/* The precision the most recent batched solve on this handle ran in: LMPC_PRECISION_MIXED after lmpc_solve_batch_mixed (or
 * lmpc_solve_batch_ss_idx with that precision) where a reduced-precision kernel exists for the handle's (N, num_ss_pts),
 * LMPC_PRECISION_F64 where the entry fell back to the fp64 kernels (above), LMPC_PRECISION_F32 after lmpc_solve_batch_f32. */
int lmpc_last_solve_precision(const lmpc_handle* h, int32_t* precision);

/* Per-kernel timing for benchmarks: when enabled, lmpc_solve_batch brackets its two launches
 * with HIP events on the handle's stream; lmpc_last_kernel_ms waits for them and returns the
 * durations of the linearisation kernel and of the QP kernel of the most recent call. */
int lmpc_enable_timing(lmpc_handle* h, int32_t on);
int lmpc_last_kernel_ms(lmpc_handle* h, float* linearize_ms, float* solve_ms);

#ifdef __cplusplus
}
#endif
#endif /* LMPC_HIP_H_ */
