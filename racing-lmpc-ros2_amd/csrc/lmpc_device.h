// lmpc_device.h -- parameter block shared by the host C-ABI layer and the gfx950 kernels.
#ifndef LMPC_DEVICE_H_
#define LMPC_DEVICE_H_

#include "lmpc_hip.h"

// Everything the kernels need from lmpc_config / lmpc_vehicle, pre-digested on the host
// (lmpc_create).  Passed by value as a kernel argument (lands in the kernarg segment / SGPRs).
struct lmpc_params {
  void* save;  // device [batch][10 N - 4] doubles: where a polish attempt puts the iterate aside (handle-owned, lmpc_reserve)
  const int* launch_order;  // device [batch] or null: workgroup w solves problem launch_order[w] (set per launch by the host
                            // layer from lmpc_set_launch_order, only when the batch size matches the registered length)
  const int* order_count;   // device scalar or null: only the first *order_count entries of launch_order are problems (the
                            // fp64 second pass of a mixed solve: the list of problems the fp32 pass could not verify)
  int N;          // knot points
  int has_sigma;  // q_boundary > 0: one shared boundary slack (racing_mpc.cpp:529-539)
  int learning;   // LMPC terminal set + cost (racing_mpc.cpp:479-522)
  int S;          // safe-set points
  int max_iter;
  int polish;     // lmpc_config.polish: >= 0 the active-set polish runs (racing_mpc.cpp:90-95, OSQP polish = true)
  double tol;      // complementarity tolerance of the interior-point iteration
  double Qd[6];    // 2*q stage weights on x      (racing_mpc.cpp:459-463)
  double Qt[6];    // 2*10*q terminal weights     (racing_mpc.cpp:474-476)
  double qv_stage; // -2 q_vel      (times vel_ref_i gives the linear term)
  double qv_term;  // -20 q_vel
  double Qu[4];    // R + R'
  double Sv[4];    // R_d + R_d'
  double qsig;     // 2 q_boundary
  double x_max[6], x_min[6];
  double u_hi[2], u_lo[2]; // MPC box intersected with the actuator box
  double v_hi[2], v_lo[2]; // rate box (single_track_planar_model.cpp:146-151)
  double marg;             // margin + chassis.b / 2 (racing_mpc.cpp:531)
  double chs2[6];          // 2 * convex_hull_slack
  double max_vel_ref_diff;
  // two-pass mixed precision (set per launch by the host layer): the fp32 iteration marks a problem whose answer it could not
  // verify (polish refused) with LMPC_SOLVE_UNVERIFIED instead of OPTIMAL; lmpc_cleanup_kernel behind it solves those in fp64
  int flag_unverified;
  int hard_hull;  // all-zero convex_hull_slack: chs2 = 2 LMPC_HARD_HULL_WEIGHT and the residual is checked at the exit
  int out_aos;  // lmpc_set_output_layout: results [batch][knot][component] instead of [component][knot][batch]
  int warm_rounds;  // lmpc_set_warm_rounds: repairs a warm start may spend before the cold start takes over (0: WARM_ROUNDS); read by the warm kernels only
  // the safe set by reference (lmpc_solve_batch_ss_idx): S codes per problem from lmpc_ss_query_idx_batch, [S][B], and the lap
  // store they point into (the handle's copy: lmpc_set_safe_set); ss_idx == null: the points arrive as arrays (ss_x, ss_j)
  const int* ss_idx;
  const double* ss_store;   // [rows][6]
  const int* ss_npts;       // [ss_laps]
  const int* ss_off;        // [ss_laps] first row of each lap
  int ss_laps;
  int ss_rows;              // rows of the store: a code naming a row past it (or a fourth copy) is treated as "no point"
  double ss_L;
  // warm start (lmpc_solve_batch_warm): the plan the active-set attempt starts from, [6][N][B] and [2][N-1][B]; null: a cold solve
  const double* warm_X;
  const double* warm_U;
  const double* warm_lam;  // learning: the plan's simplex weights [S][B] (convex_combi_optm_ref, racing_mpc.cpp:281), aligned with this call's points
  int* warm_flag;          // device [B] or null: 1 where the warm attempt was accepted, 0 where the cold solve ran (written by the warm kernels)
  lmpc_vehicle veh;
};

#define LMPC_WARM_ROUNDS_MAX 4  // lmpc_set_warm_rounds' upper limit = the polish's rounds (polish_limits::rounds); a cold solve reports more iterations

// LDS record sizes (in doubles) of the solve kernel; see DESIGN.md "data layout".
#define LMPC_STAGE_STRIDE 78
#define LMPC_KNOT_STRIDE 36
#define LMPC_TAIL_DOUBLES 320
// learning: the terminal region behind the records -- always fp64 cells, whatever the records' type: the terminal-block
// scratch (lmpc_solve_kernel.hip TL_*) and the (centred) safe-set points [6][64 KS], KS = 2 up to 128 points, 3 up to 192
#define LMPC_TERM_CELLS 272
#define LMPC_SS_STRIDE(S) ((S) + 1)  // safe-set points kept behind the terminal cells, 6 cells each: S of them + one zero point
#define LMPC_LIN_RECORD 54  // per stage in the linearisation workspace: ABt[8][6] | g[6]

// the lean layout (fp64, N > 40; lmpc_solve_kernel.hip): stage records without the stage model, which is streamed
// through two chunk buffers of LMPC_LEAN_CHUNK workspace records
#define LMPC_LEAN_STAGE_STRIDE 24
#define LMPC_LEAN_CHUNK 8
#ifndef LMPC_LEAN_MIN_KQ
#define LMPC_LEAN_MIN_KQ 11
#endif
// (the kernels' slot classes: 2 / 4 / 7 / 11 / 14 slots per lane for N <= 11 / 23 / 40 / 64 / 81; lean from LMPC_LEAN_MIN_KQ slots on)
static inline int lmpc_slot_class(int N) { return N <= 11 ? 2 : (N <= 23 ? 4 : (N <= 40 ? 7 : (N <= 64 ? 11 : 14))); }
static inline int lmpc_is_lean(int N, int real_bytes) { return real_bytes == 8 && lmpc_slot_class(N) >= LMPC_LEAN_MIN_KQ; }

// LDS bytes per problem; real_bytes = 8 (fp64 records) or 4 (fp32 records: single-precision and mixed solves)
static inline size_t lmpc_lds_bytes(int N, int learning, int S, int real_bytes) {
  const int ks = !learning ? 0 : (S <= 128 ? 2 : 3);
  const int lean = lmpc_is_lean(N, real_bytes);
  const size_t records = (size_t)((N - 1) * (lean ? LMPC_LEAN_STAGE_STRIDE : LMPC_STAGE_STRIDE) + N * LMPC_KNOT_STRIDE + LMPC_TAIL_DOUBLES) *
                         (size_t)real_bytes;
  (void)ks;
  return records + (learning ? (size_t)(LMPC_TERM_CELLS + 6 * LMPC_SS_STRIDE(S)) * 8 : 0) + (lean ? (size_t)2 * LMPC_LEAN_CHUNK * LMPC_LIN_RECORD * 8 : 0);
}

#endif
