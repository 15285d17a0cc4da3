// lmpc_prep_kernels.hip -- gfx950 kernels either side of the QP solve:
//   lmpc_linearize_kernel : discrete_dynamics_jacobian (A, B, g) of every stage of every problem
//                           (single_track_planar_model.cpp:377-387, used at racing_mpc.cpp:173-182)
//   lmpc_prepare_kernel   : the controller node's cold-start input preparation
//                           (racing_mpc_node.cpp:210-235, 261-292)
// Both are one thread per unit of work with the batch axis on the lanes, so every global access
// is a coalesced 512-byte wave transaction on the [field][knot][batch] arrays.
#include <hip/hip_runtime.h>

#include "lmpc_device.h"
#include "lmpc_dynamics.hip.h"

// One thread per (problem b, stage i).  blockIdx.y = stage, so a wave covers 64 consecutive
// problems of one stage.  Forward-mode chain rule through the four RK4 stages:
//   K_s = Fx(x_s) X_s + Fu(x_s) [0 I],   X_{s+1} = [I 0] + c_s dt K_s,   [A B] = [I 0] + dt/6 sum w_s K_s.
// The primal sweep stores the sparse partials of the four points; the eight tangent columns
// are then pushed through one at a time (keeps the live register set small).
// WS_LAYOUT = true : workspace for the solve kernel, record per (b, i): ABt[8][6] | g[6]
//                    (ABt[c][k] = [A B][k][c], i.e. columns contiguous)
// WS_LAYOUT = false: C-ABI arrays A [6][6][N-1][B], Bm [6][2][N-1][B], g [6][N-1][B]
// io = element type of the arrays (double, or float for the single-precision solve); the arithmetic is fp64.
// W = waves per SIMD the registers are sized for.  W = 1: 378 VGPRs + 122 AGPRs, no scratch -- the faster kernel on its own (43 against
// 49 us per 4096 x 19).  W = 2: 256 VGPRs and 612 B of spills, but a wave of it fits NEXT TO a resident wave of a two-waves-per-SIMD QP
// kernel (256 VGPRs each), so the next batch's linearisation runs in the residency tail of this batch's solve instead of waiting for
// whole SIMDs to drain: +5.5 % solves/s on the pipelined headline, +10 % on the fp32 IAC configuration; the host picks (lmpc_capi.hip).
template <bool WS_LAYOUT, typename io, int W>
__global__ __launch_bounds__(256, W) void lmpc_linearize_kernel(lmpc_params P, int B, const io* __restrict__ X_ref,
                                                             const io* __restrict__ U_ref, const io* __restrict__ T_ref,
                                                             const io* __restrict__ curv, io* __restrict__ outA,
                                                             io* __restrict__ outB, io* __restrict__ outg) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  const int N = P.N, NS = N - 1;
  if (b >= B) return;
  double x[6], u[2];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = X_ref[(size_t)(k * N + i) * B + b];
#pragma unroll
  for (int k = 0; k < 2; ++k) u[k] = U_ref[(size_t)(k * NS + i) * B + b];
  const double dt = T_ref[(size_t)i * B + b];
  const double kap = curv[(size_t)i * B + b];

  lmpc_uterms ut;
  lmpc_u_terms(P.veh, u[0], u[1], ut);
  lmpc_fjac J[4];
  double ks[4][6], xs[6];
  const double cs[4] = {0.0, 0.5, 0.5, 1.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int r = 0; r < 6; ++r) xs[r] = (s == 0) ? x[r] : x[r] + cs[s] * dt * ks[s - 1][r];
    lmpc_f<true>(P.veh, ut, xs, kap, ks[s], &J[s]);
  }
  // weights of the four slopes: dt/6 (1, 2, 2, 1), or -- Euler, utils.cpp:110-123 -- the first slope alone (the other
  // three evaluations are then wasted; no shipped file selects Euler)
  const bool euler = P.veh.integrator == LMPC_INTEGRATOR_EULER;
  const double wgt[4] = {euler ? dt : dt / 6, euler ? 0.0 : dt / 3, euler ? 0.0 : dt / 3, euler ? 0.0 : dt / 6};
  double xp[6], gacc[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    xp[r] = x[r] + (wgt[0] * ks[0][r] + wgt[1] * ks[1][r] + wgt[2] * ks[2][r] + wgt[3] * ks[3][r]);
    gacc[r] = xp[r];
  }
  // (ONE column at a time since round 6 -- `unroll 1`: unrolled, the compiler interleaved the eight columns' chains and the kernel
  //  held 378 + 122 registers, or 256 with 612 B of spills in the W = 2 build; now 256 + 66 / 256 with 268 B, same arithmetic per
  //  column, same bits.  A stage-major form -- one point's partials live at a time, seven columns travelling together -- was built
  //  and is worse: 728 B)
#pragma unroll 1
  for (int c = 0; c < 8; ++c) {
    double tu[2] = {c == 6 ? 1.0 : 0.0, c == 7 ? 1.0 : 0.0};
    double e[6], tx[6], kc[6], acc[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      e[r] = (r == c) ? 1.0 : 0.0;
      tx[r] = e[r];
      acc[r] = 0.0;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      lmpc_jvp(J[s], tx, tu, kc);
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        acc[r] += wgt[s] * kc[r];
        if (s < 3) tx[r] = e[r] + cs[s + 1] * dt * kc[r];
      }
    }
    double xu = x[0];  // (x[c] / u[c - 6] by selects: c is a run-time index now, and an indexed register array would live in scratch)
#pragma unroll
    for (int k = 1; k < 6; ++k) xu = (c == k) ? x[k] : xu;
    xu = (c == 6) ? u[0] : ((c == 7) ? u[1] : xu);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double d = e[r] + acc[r];  // [A B][r][c]
      gacc[r] -= d * xu;
      if (WS_LAYOUT)
        outA[((size_t)b * NS + i) * LMPC_LIN_RECORD + c * 6 + r] = (io)d;
      else if (c < 6)
        outA[((size_t)(r * 6 + c) * NS + i) * B + b] = (io)d;
      else
        outB[((size_t)(r * 2 + (c - 6)) * NS + i) * B + b] = (io)d;
    }
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    if (WS_LAYOUT)
      outA[((size_t)b * NS + i) * LMPC_LIN_RECORD + 48 + r] = (io)gacc[r];
    else
      outg[((size_t)r * NS + i) * B + b] = (io)gacc[r];
  }
}

template __global__ void lmpc_linearize_kernel<true, double, 1>(lmpc_params, int, const double*, const double*, const double*,
                                                             const double*, double*, double*, double*);
template __global__ void lmpc_linearize_kernel<true, float, 1>(lmpc_params, int, const float*, const float*, const float*, const float*,
                                                            float*, float*, float*);
template __global__ void lmpc_linearize_kernel<false, double, 1>(lmpc_params, int, const double*, const double*, const double*,
                                                              const double*, double*, double*, double*);
template __global__ void lmpc_linearize_kernel<true, double, 2>(lmpc_params, int, const double*, const double*, const double*,
                                                             const double*, double*, double*, double*);
template __global__ void lmpc_linearize_kernel<true, float, 2>(lmpc_params, int, const float*, const float*, const float*, const float*,
                                                            float*, float*, float*);

// periodic linear interpolation on a uniform table of M samples over [0, L)
__device__ __forceinline__ double track_lookup(const double* __restrict__ tab, int M, double L, double s) {
  double u = fmod(s, L);
  if (u < 0.0) u += L;
  u = u / (L / M);
  double fl = floor(u);
  const double fr = u - fl;
  int i0 = (int)fl;
  i0 = i0 % M;
  if (i0 < 0) i0 += M;
  const int i1 = (i0 + 1 == M) ? 0 : i0 + 1;
  return tab[i0] * (1.0 - fr) + tab[i1] * fr;
}

// Reference sampling at one knot (racing_mpc_node.cpp:261-292): bounds, curvature, clamped velocity reference.
__device__ __forceinline__ void sample_refs(const lmpc_track& trk, double s, double cur, double d, double speed_scale,
                                            double speed_limit, double& bl, double& br, double& kap, double& vr_out) {
  kap = track_lookup(trk.curvature, trk.M, trk.L, s);
  bl = track_lookup(trk.bound_left, trk.M, trk.L, s);
  br = track_lookup(trk.bound_right, trk.M, trk.L, s);
  const double vr = track_lookup(trk.vel, trk.M, trk.L, s) * speed_scale;
  const double lim = fmin(fmax(speed_limit, cur - d), cur + d);       // :273-275
  const double clipped = fmin(fmax(vr, cur - d), cur + d);
  vr_out = (vr > 0.0) ? fmin(clipped, lim) : lim;                     // :276-285
}

// Warm-start shift of RacingMPCNode::on_step_timer (racing_mpc_node.cpp:245-254): the previous solution moves
// one knot forward, the last input is repeated, the last state is rolled out with the model, and the references
// are re-sampled along the shifted abscissa.  `status` (may be NULL) selects, per problem, the previous SOLUTION
// (status 0) or the previous REFERENCE (solve failed: the node keeps driving on the shifted old plan, :322-332).
// One thread per problem; inputs and outputs must not alias.
__global__ __launch_bounds__(256) void lmpc_shift_kernel(lmpc_params P, int B, lmpc_track trk,
                                                         const double* __restrict__ X_sol, const double* __restrict__ U_sol,
                                                         const double* __restrict__ X_old, const double* __restrict__ U_old,
                                                         const int* __restrict__ status, double dt, double speed_scale,
                                                         double speed_limit, double* __restrict__ X_ref,
                                                         double* __restrict__ U_ref, double* __restrict__ T_ref,
                                                         double* __restrict__ bl, double* __restrict__ br,
                                                         double* __restrict__ curv, double* __restrict__ vref) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int N = P.N, NS = N - 1;
  const bool ok = !status || status[b] == 0;
  const double* Xs = ok ? X_sol : X_old;
  const double* Us = ok ? U_sol : U_old;
  const double d = P.max_vel_ref_diff;
  double x[6], u[2] = {0.0, 0.0}, xn[6];
  for (int i = 0; i < N; ++i) {
    if (i < NS) {
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = Xs[(size_t)(k * N + i + 1) * B + b];
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = xn[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) X_ref[(size_t)(k * N + i) * B + b] = x[k];
    double b_l, b_r, kap, vr;
    sample_refs(trk, x[0], x[3], d, speed_scale, speed_limit, b_l, b_r, kap, vr);
    bl[(size_t)i * B + b] = b_l;
    br[(size_t)i * B + b] = b_r;
    curv[(size_t)i * B + b] = kap;
    vref[(size_t)i * B + b] = vr;
    if (i < NS) {
      const int src = (i < NS - 1) ? i + 1 : NS - 1;  // last input repeated (:247)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        u[k] = Us[(size_t)(k * NS + src) * B + b];
        U_ref[(size_t)(k * NS + i) * B + b] = u[k];
      }
      T_ref[(size_t)i * B + b] = dt;
      if (i == NS - 1) lmpc_fd(P.veh, x, u, kap, dt, xn);  // :248-249
    }
  }
}

// Plant step of RacingSimulator::step (racing_simulator.cpp:46-69,97-112): RK4 with the track curvature at the
// current abscissa, abscissa wrapped into [0, L) by align_abscissa(s, L/2, L); nsub sub-steps of dt_sim with the
// input held.  One thread per car; x is updated in place.
__global__ __launch_bounds__(256) void lmpc_plant_kernel(lmpc_params P, int B, lmpc_track trk, double* __restrict__ x_io,
                                                         const double* __restrict__ u_in, double dt_sim, int nsub) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double x[6], xn[6];
  const double u[2] = {u_in[b], u_in[(size_t)B + b]};
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = x_io[(size_t)k * B + b];
  for (int j = 0; j < nsub; ++j) {
    if (fabs(x[3]) < 1e-6) x[3] = copysign(1e-6, x[3]);  // :99-102
    const double kap = track_lookup(trk.curvature, trk.M, trk.L, x[0]);
    lmpc_fd(P.veh, x, u, kap, dt_sim, xn);
    // align_abscissa(s, L/2, L): lmpc_utils/utils.hpp:35-41
    const double s1 = xn[0], s2 = trk.L / 2.0;
    const double kk = fabs(s2 - s1) + trk.L / 2.0;
    const double ll = kk - fmod(kk, trk.L);
    xn[0] = s1 + ll * ((s2 > s1) - (s2 < s1));
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = xn[k];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) x_io[(size_t)k * B + b] = x[k];
}

// One thread per problem: zero-input rollout of the reference, then reference sampling.  With `status` (may be
// NULL) only the problems whose last solve failed (status != 0) are prepared; the arrays of the others are not touched.
__global__ __launch_bounds__(256) void lmpc_prepare_kernel(lmpc_params P, int B, lmpc_track trk,
                                                           const double* __restrict__ x_ic,
                                                           const int* __restrict__ status, double dt,
                                                           double speed_scale, double speed_limit,
                                                           double* __restrict__ X_ref, double* __restrict__ U_ref,
                                                           double* __restrict__ T_ref, double* __restrict__ bl,
                                                           double* __restrict__ br, double* __restrict__ curv,
                                                           double* __restrict__ vref) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (status && status[b] == 0) return;
  const int N = P.N, NS = N - 1;
  double x[6], xn[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = x_ic[(size_t)k * B + b];
  const double u[2] = {1e-9, 1e-9};  // racing_mpc_node.cpp:212
  const double d = P.max_vel_ref_diff;
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < 6; ++k) X_ref[(size_t)(k * N + i) * B + b] = x[k];
    double b_l, b_r, kap, vr;
    sample_refs(trk, x[0], x[3], d, speed_scale, speed_limit, b_l, b_r, kap, vr);
    bl[(size_t)i * B + b] = b_l;
    br[(size_t)i * B + b] = b_r;
    curv[(size_t)i * B + b] = kap;
    vref[(size_t)i * B + b] = vr;
    if (i < NS) {
      U_ref[(size_t)(0 * NS + i) * B + b] = u[0];
      U_ref[(size_t)(1 * NS + i) * B + b] = u[1];
      T_ref[(size_t)i * B + b] = dt;
      lmpc_fd(P.veh, x, u, kap, dt, xn);  // :216-224, curvature at the knot's own abscissa (:72-76)
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = xn[k];
    }
  }
}

// What follows a solve in a closed loop, in one launch (lmpc_loop_advance_batch): the node's choice of the input to apply
// (racing_mpc_node.cpp:322-332: the plan's first input, or after a failed solve the first input of the plan it was linearised on),
// the plant step (racing_simulator.cpp:46-69,97-112), the bookkeeping a Monte-Carlo harness keeps per car, and the next
// period's inputs: the warm-start shift (:245-254) or, for a car whose solve failed and `restart_failed`, the cold-start
// preparation at its new state (:210-235, 261-292).  The same arithmetic as lmpc_plant_kernel, lmpc_shift_kernel and
// lmpc_prepare_kernel in that order (tests/test_gpu_loop.py: bit for bit, but for the last knot's one-step rollout, where the inlined
// model is contracted differently here and there: 1 - 2 ulp) without the ~40 small launches between them.
// A workgroup is 64 cars (the lanes) x LMPC_LOOP_WAVES waves.  Wave 0 applies the input, steps the plant and keeps the books;
// the shift of a car whose solve succeeded has no chain in it except the last knot's rollout, so the waves share its knots
// (wave w: knots w, w + W, ...; reads from the solution only, writes the references: no hazard, no barrier).  A car whose solve
// failed is wave 0's alone, knot after knot: the cold restart is a rollout, and the shift of the OLD plan is done in place (a
// thread reads knot i + 1 before it writes knot i + 1) -- which is why the reference arrays carry no __restrict__.
#ifndef LMPC_LOOP_WAVES  // (4 / 8 / 16 measured on the closed loop, profiles/r05_tail_ab.txt: 6.49 / 6.53 / 6.36 M car-steps/s at 4096 cars warm)
#define LMPC_LOOP_WAVES 8
#endif
__global__ __launch_bounds__(64 * LMPC_LOOP_WAVES) void lmpc_loop_advance_kernel(
    lmpc_params P, int B, lmpc_track trk, const int* __restrict__ status, const int* __restrict__ iters, const double* __restrict__ X_sol,
    const double* __restrict__ U_sol, double* __restrict__ x_io, double* __restrict__ u_prev, double dt, double dt_sim, int nsub,
    double speed_scale, double speed_limit, int restart_failed, double* X_ref, double* U_ref, double* T_ref, double* bl, double* br,
    double* curv, double* vref, double* __restrict__ distance, double* __restrict__ worst_excess, long long* __restrict__ n_fail,
    unsigned long long* __restrict__ n_accepted) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + lane;
  const int N = P.N, NS = N - 1;
  const bool live = b < B;
  const bool ok = live && status[b] == 0;
  if (n_accepted && w == 0) {
    // warm attempts accepted: the warm kernel's own flag (P.warm_flag, set by the host layer when the last solve of this batch was a
    // warm one).  Until round 6 this was inferred -- "at most LMPC_WARM_ROUNDS_MAX iterations: a cold solve takes more" -- which
    // miscounts any cold solve that finishes in four; kept only for a caller that solved cold (flag absent: count nothing).
    const unsigned long long m = __ballot(ok && P.warm_flag != nullptr && P.warm_flag[live ? b : 0] != 0);
    if (lane == 0 && m) atomicAdd(n_accepted, (unsigned long long)__popcll(m));
  }
  if (!live) return;
  const double d = P.max_vel_ref_diff;
  double x[6], xn[6], u[2];
  if (w == 0) {
    // ---- the input applied, the plant ----
#pragma unroll
    for (int k = 0; k < 2; ++k) u[k] = ok ? U_sol[(size_t)(k * NS) * B + b] : U_ref[(size_t)(k * NS) * B + b];
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = x_io[(size_t)k * B + b];
    const double s_before = x[0];
    for (int j = 0; j < nsub; ++j) {
      if (fabs(x[3]) < 1e-6) x[3] = copysign(1e-6, x[3]);
      const double kap = track_lookup(trk.curvature, trk.M, trk.L, x[0]);
      lmpc_fd(P.veh, x, u, kap, dt_sim, xn);
      const double s1 = xn[0], s2 = trk.L / 2.0;
      const double kk = fabs(s2 - s1) + trk.L / 2.0;
      const double ll = kk - fmod(kk, trk.L);
      xn[0] = s1 + ll * ((s2 > s1) - (s2 < s1));
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = xn[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) x_io[(size_t)k * B + b] = x[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) u_prev[(size_t)k * B + b] = u[k];
    // ---- bookkeeping: abscissa travelled (unwrapped), worst excursion of the body beyond the track edge, failed solves ----
    if (distance) {
      const double ds = x[0] - s_before;
      distance[b] += (ds < -trk.L / 2.0) ? ds + trk.L : ds;
    }
    if (worst_excess) {
      const double half_b = P.veh.b / 2.0;
      const double exc = fmax(x[1] + half_b - bl[b], br[b] - (x[1] - half_b));  // (knot 0 of the period that has just been driven)
      worst_excess[b] = fmax(worst_excess[b], exc);
    }
    if (n_fail && !ok) n_fail[b] += 1;
  }
  // ---- the next period's inputs ----
  if (!ok) {
    if (w != 0) return;
    if (restart_failed) {  // cold start at the new state (lmpc_prepare_kernel)
      const double u0[2] = {1e-9, 1e-9};
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < 6; ++k) X_ref[(size_t)(k * N + i) * B + b] = x[k];
        double b_l, b_r, kap, vr;
        sample_refs(trk, x[0], x[3], d, speed_scale, speed_limit, b_l, b_r, kap, vr);
        bl[(size_t)i * B + b] = b_l;
        br[(size_t)i * B + b] = b_r;
        curv[(size_t)i * B + b] = kap;
        vref[(size_t)i * B + b] = vr;
        if (i < NS) {
          U_ref[(size_t)(0 * NS + i) * B + b] = u0[0];
          U_ref[(size_t)(1 * NS + i) * B + b] = u0[1];
          T_ref[(size_t)i * B + b] = dt;
          lmpc_fd(P.veh, x, u0, kap, dt, xn);
#pragma unroll
          for (int k = 0; k < 6; ++k) x[k] = xn[k];
        }
      }
      return;
    }
    for (int i = 0; i < N; ++i) {  // the plan the failed solve started from, shifted in place (lmpc_shift_kernel with X_old = X_ref)
      if (i < NS) {
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = X_ref[(size_t)(k * N + i + 1) * B + b];
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = xn[k];
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) X_ref[(size_t)(k * N + i) * B + b] = x[k];
      double b_l, b_r, kap, vr;
      sample_refs(trk, x[0], x[3], d, speed_scale, speed_limit, b_l, b_r, kap, vr);
      bl[(size_t)i * B + b] = b_l;
      br[(size_t)i * B + b] = b_r;
      curv[(size_t)i * B + b] = kap;
      vref[(size_t)i * B + b] = vr;
      if (i < NS) {
        const int src = (i < NS - 1) ? i + 1 : NS - 1;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          u[k] = U_ref[(size_t)(k * NS + src) * B + b];
          U_ref[(size_t)(k * NS + i) * B + b] = u[k];
        }
        T_ref[(size_t)i * B + b] = dt;
        if (i == NS - 1) lmpc_fd(P.veh, x, u, kap, dt, xn);
      }
    }
    return;
  }
  // the solution, shifted: knot i of the new reference is knot i + 1 of the solution; the last one is rolled out from the
  // new knot N - 2 with the repeated last input (racing_mpc_node.cpp:247-249)
  for (int i = w; i < N; i += LMPC_LOOP_WAVES) {
    if (i < NS) {
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = X_sol[(size_t)(k * N + i + 1) * B + b];
    } else {
      double xm[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) xm[k] = X_sol[(size_t)(k * N + NS) * B + b];
#pragma unroll
      for (int k = 0; k < 2; ++k) u[k] = U_sol[(size_t)(k * NS + NS - 1) * B + b];
      const double kap_m = track_lookup(trk.curvature, trk.M, trk.L, xm[0]);
      lmpc_fd(P.veh, xm, u, kap_m, dt, x);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) X_ref[(size_t)(k * N + i) * B + b] = x[k];
    double b_l, b_r, kap, vr;
    sample_refs(trk, x[0], x[3], d, speed_scale, speed_limit, b_l, b_r, kap, vr);
    bl[(size_t)i * B + b] = b_l;
    br[(size_t)i * B + b] = b_r;
    curv[(size_t)i * B + b] = kap;
    vref[(size_t)i * B + b] = vr;
    if (i < NS) {
      const int src = (i < NS - 1) ? i + 1 : NS - 1;
#pragma unroll
      for (int k = 0; k < 2; ++k) U_ref[(size_t)(k * NS + i) * B + b] = U_sol[(size_t)(k * NS + src) * B + b];
      T_ref[(size_t)i * B + b] = dt;
    }
  }
}

// Launch order for the next solve of a closed-loop batch: problems sorted by the iteration count of their last solve,
// longest first (counting sort over 0 .. 63 iterations; one workgroup; the order inside a bucket is by problem index, so the
// result is reproducible).
__global__ __launch_bounds__(1024) void lmpc_launch_order_kernel(int B, const int* __restrict__ iters, int* __restrict__ order) {
  __shared__ int count[64], start[64];
  const int tid = threadIdx.x;
  if (tid < 64) count[tid] = 0;
  __syncthreads();
  for (int b = tid; b < B; b += blockDim.x) atomicAdd(&count[min(max(iters[b], 0), 63)], 1);
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = 63; k >= 0; --k) {  // descending: the longest first
      start[k] = acc;
      acc += count[k];
    }
  }
  __syncthreads();
  // rank inside the bucket = number of problems with the same count and a smaller index (keeps the order deterministic)
  for (int k = tid; k < 64; k += blockDim.x) count[k] = 0;
  __syncthreads();
  for (int base = 0; base < B; base += blockDim.x) {  // chunks in index order; inside a chunk ranks by a prefix over lanes
    const int b = base + tid;
    const int key = b < B ? min(max(iters[b], 0), 63) : -1;
    // serialise per key within the chunk: thread t counts equal keys among threads < t of this chunk
    __shared__ int keys[1024];
    keys[tid] = key;
    __syncthreads();
    int rank = 0;
    if (key >= 0)
      for (int t = 0; t < tid; ++t) rank += keys[t] == key ? 1 : 0;
    if (key >= 0) order[start[key] + count[key] + rank] = b;
    __syncthreads();
    if (key >= 0) atomicAdd(&count[key], 1);
    __syncthreads();
  }
}

// The problems a mixed-precision first pass has marked LMPC_SOLVE_UNVERIFIED: list [0 .. n) and n at list [B].  One workgroup;
// the order within the list is whatever the atomics produce (each problem is solved on its own: the results do not depend on it).
__global__ __launch_bounds__(1024) void lmpc_collect_unverified_kernel(int B, const int* __restrict__ status, int* __restrict__ list) {
  __shared__ int n;
  if (threadIdx.x == 0) n = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += 1024)
    if (status[b] == LMPC_SOLVE_UNVERIFIED) list[atomicAdd(&n, 1)] = b;
  __syncthreads();
  if (threadIdx.x == 0) list[B] = n;
}


// convex_combi_optm_ref for the learning problem's warm start when the safe set goes BY REFERENCE (lmpc_shift_lambda_batch): the
// previous solution's simplex weights carried onto this period's points by the IDENTITY of the points, not by their position in the
// set (the query returns its neighbours nearest first: positions are scrambled from one period to the next).  A point of the
// previous set with weight lam > 0 and code c is looked for in the new set as the sample `advance` steps further along the same
// lap copy (the plan's terminal state moves forward about one sample per control period: the optimum's support moves with it);
// where that sample is not among the new neighbours, the point itself.  One thread per problem: the support is <= 6 points.
__device__ __forceinline__ int lmpc_advance_code(int code, int adv, int n_laps, const int* __restrict__ npts, const int* __restrict__ off) {
  if (code < 0) return -1;
  int row = code >> 2, rep = code & 3, l = 0;
  for (int t = 1; t < n_laps; ++t) l = (row >= off[t]) ? t : l;
  int jj = row - off[l] + adv;
  const int n = npts[l];
  while (jj >= n) {  // past the end of a copy: the start of the next one (x + L e_0: SSTrajectory::process_lap_data, safe_set.cpp:123-125)
    jj -= n;
    ++rep;
  }
  return rep > 2 ? -1 : ((off[l] + jj) << 2) | rep;
}

__global__ __launch_bounds__(64) void lmpc_shift_lambda_kernel(int B, int S, int n_laps, const int* __restrict__ npts, const int* __restrict__ off,
                                                               const int* __restrict__ idx_prev, const double* __restrict__ lam_prev,
                                                               const int* __restrict__ idx, int advance, double* __restrict__ lam_ref) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  constexpr int MAXSUP = 8, MAXFREE = 6;  // (the terminal block keeps at most six weights explicit: MA_MAX of lmpc_solve_kernel.hip)
  int sup_code[MAXSUP], n = 0;
  double sup_lam[MAXSUP];
  for (int i = 0; i < S; ++i) {
    const double l = lam_prev[(size_t)i * B + b];
    if (l > 1e-9 && n < MAXSUP) {
      sup_code[n] = idx_prev[(size_t)i * B + b];
      sup_lam[n] = l;
      ++n;
    }
  }
  for (int j = 0; j < S; ++j) lam_ref[(size_t)j * B + b] = 0.0;
  // position of a code in the new set (its first occurrence: the padding of a short set repeats the last point), -1: not among the neighbours
  auto find = [&](int want) {
    if (want < 0) return -1;
    for (int j = 0; j < S; ++j)
      if (idx[(size_t)j * B + b] == want) return j;
    return -1;
  };
  // Where the support goes is not smooth (measured in the LMPC experiment, scratch/r6/lmpc_warm_probe.py, profiles/r06_lmpc_warm.txt):
  // from one period to the next a support point stays, moves one sample along its lap, or two; now and then a far point enters.  Each
  // support point takes ONE candidate -- `advance` samples on, else the point itself, else `advance` + 1 samples on.  (Proposing all
  // three as free weights -- a superset of the likely support, for the attempt's repair rounds to prune -- was built and measured:
  // acceptance fell from 12 % to 0.5 %.  Neighbouring samples of a lap are nearly collinear, the free weights' system C_A loses
  // rank, and the multiplier steps are noise.)
  int nfree = 0;
  for (int k = 0; k < n; ++k) {
    const int order[3] = {advance, 0, advance + 1};
    for (int t = 0; t < 3; ++t) {
      const int j = find(lmpc_advance_code(sup_code[k], order[t], n_laps, npts, off));
      if (j < 0) continue;
      double& cell = lam_ref[(size_t)j * B + b];
      if (cell == 0.0) {
        if (nfree >= MAXFREE) continue;
        ++nfree;
      }
      cell += sup_lam[k];
      break;
    }
  }
}
