// lmpc_solve_kernel.hip -- the batched QP solve of RacingMPC::solve on gfx950 (CDNA4), fp64.
//
// What it replaces: opti_.solve_limited() on the "conic"/OSQP problem built in
//   src/mpc/racing_mpc/src/racing_mpc.cpp:106-201 (constraints), :442-477 (tracking cost),
//   :524-543 (boundary slack); actuator boxes from single_track_planar_model.cpp:113-120,144-151.
//
// Mapping: ONE WAVEFRONT (64 lanes) PER PROBLEM, one wave per workgroup, everything the
// iteration touches resident in LDS (~20 KB at N = 20 -> 8 problems per CU).
//   * Riccati factorisation: the augmented state z = [x; u_prev] has 8 components, so the 8x8
//     cost-to-go matrix is exactly one wave: lane l owns element (r, c) = (l >> 3, l & 7).
//     Matrix products are 6-term dot products read from LDS (b128 row reads, padded rows);
//     no cross-lane shuffles are needed in any matrix or vector phase.
//   * Riccati vector solves: lane (s, r) = (l >> 3, l & 7) computes component r of right-hand
//     side s, so the predictor step and the boundary-slack Schur vector are solved in the same
//     instruction stream (two RHS for the price of one).
//   * Inequality rows: the 11 two-sided slots of each knot (6 state, 2 input, 2 input-rate,
//     1 track boundary) are dealt round-robin to lanes; slacks and multipliers never leave
//     registers.  The slot owner also owns the primal component the slot constrains: it writes
//     that component's barrier weight and gradient entry and applies its update.
//   * Wave-wide scalars (mu, step length, Schur dot products) use 6-step xor-shuffle reductions.
// Too small for MFMA (6..8-wide blocks); the kernel is FP64-VALU / LDS-latency bound.
//
// Algorithm (twin of oracle/c/lmpc_oracle.c, which documents the derivation): Mehrotra
// predictor-corrector interior point; Newton systems by Riccati recursion on (z, v = dU); the
// shared boundary slack sigma (one scalar coupling all knots) by a Schur complement.
#include <hip/hip_runtime.h>

#include "lmpc_device.h"

#define NSLOT 11
#define SL_U 6
#define SL_V 8
#define SL_EY 10

// ---- LDS layout (doubles) ---------------------------------------------------------------------
// stage record i (stride 80): ABt[8][6] @0 (ABt[c][k] = [A B][k][c]) | g[6] @48 | dt @54 | K[2][8] @56
//                             | Hinv (h00,h01,h11) @72 | kff[2 rhs][2] @76
// knot record i (stride 34):  z[8] v[2] @0 | rhs0: Th / q / d [10] @10 | rhs1: q / e [10] @20
//                             | csig @30 | eyT / eyD @31 | (free) @32 | qlin_vx @33
// tail: P[8][10] @0 | W[8][10] @80 | Y[8][10] @160 | pvec[2 buf][2 rhs][8] @240 | consts @272
#define ST_G 48
#define ST_DT 54
#define ST_K 56
#define ST_HI 72
#define ST_KFF 76
#define KN_R0 10
#define KN_R1 20
#define KN_CSIG 30
#define KN_EY 31
#define KN_QLIN 33
#define TL_P 0
#define TL_W 80
#define TL_Y 160
#define TL_PV 240
#define TL_CT 272
#define CT_QD 0
#define CT_QT 6
#define CT_QU 12
#define CT_SV 16
#define CT_HI 20
#define CT_LO 30
#define MROW 10  // padded row stride of the 8x8 work matrices (conflict-free b128 row reads)

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = fmax(x, __shfl_xor(x, m));
  return x;
}
__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = fmin(x, __shfl_xor(x, m));
  return x;
}

struct Lds {
  double* base;
  int N;
  __device__ __forceinline__ double* st(int i) const { return base + i * LMPC_STAGE_STRIDE; }
  __device__ __forceinline__ double* kn(int i) const { return base + (N - 1) * LMPC_STAGE_STRIDE + i * LMPC_KNOT_STRIDE; }
  __device__ __forceinline__ double* tail() const { return base + (N - 1) * LMPC_STAGE_STRIDE + N * LMPC_KNOT_STRIDE; }
};

// true cost Hessian entry on z_i (no barrier terms), racing_mpc.cpp:459-476
__device__ __forceinline__ double qz_entry(const double* ct, int N, int i, int r, int c) {
  if (r < 6 || c < 6) return (r == c) ? (i == N - 1 ? ct[CT_QT + r] : ct[CT_QD + r]) : 0.0;
  return (i >= 1) ? ct[CT_QU + (r - 6) * 2 + (c - 6)] : 0.0;
}

// Backward Riccati sweep for the barrier weights currently in the knots' rhs0 region
// (Thz @ +10..17, Thv @ +18,19, boundary weight @ KN_EY).  Leaves K, Hinv in the stage records.
__device__ void riccati_factor(const Lds& L, int lane) {
  const int N = L.N, r = lane >> 3, c = lane & 7;
  double* T = L.tail();
  double* MP = T + TL_P;
  double* MW = T + TL_W;
  double* MY = T + TL_Y;
  const double* ct = T + TL_CT;
  {
    const double* kn = L.kn(N - 1);
    double e = qz_entry(ct, N, N - 1, r, c);
    if (r == c) e += kn[KN_R0 + r] + (r == 1 ? kn[KN_EY] : 0.0);
    MP[r * MROW + c] = e;
  }
  __syncthreads();
  for (int i = N - 2; i >= 0; --i) {
    double* st = L.st(i);
    const double t = st[ST_DT];
    // W = Abar' P : W[r][c] = sum_k Abar[k][r] P[k][c]  (+ P[r][c] for the u rows); P[k][c] read as P[c][k]
    {
      double acc = (r >= 6) ? MP[r * MROW + c] : 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += st[r * 6 + k] * MP[c * MROW + k];
      MW[r * MROW + c] = acc;
    }
    __syncthreads();
    // Y = W Abar
    double y = (c >= 6) ? MW[r * MROW + c] : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) y += MW[r * MROW + k] * st[c * 6 + k];
    MY[r * MROW + c] = y;
    __syncthreads();
    // H = Sv + Thv + t^2 Y_uu, K = H^-1 t Y[6:8,:], P <- Qz + Thz + Y - t^2 Y[6:8,r]' H^-1 Y[6:8,c]
    const double* kn = L.kn(i);
    const double y6r = MY[6 * MROW + r], y7r = MY[7 * MROW + r];
    const double y6c = MY[6 * MROW + c], y7c = MY[7 * MROW + c];
    const double h00 = ct[CT_SV + 0] + kn[KN_R0 + 8] + t * t * MY[6 * MROW + 6];
    const double h01 = ct[CT_SV + 1] + t * t * MY[6 * MROW + 7];
    const double h11 = ct[CT_SV + 3] + kn[KN_R0 + 9] + t * t * MY[7 * MROW + 7];
    const double idet = 1.0 / (h00 * h11 - h01 * h01);
    const double hi00 = h11 * idet, hi01 = -h01 * idet, hi11 = h00 * idet;
    const double g0 = t * y6c, g1 = t * y7c;
    const double k0c = hi00 * g0 + hi01 * g1;
    const double k1c = hi01 * g0 + hi11 * g1;
    double pn = 0.0;
    if (i >= 1) {
      pn = qz_entry(ct, N, i, r, c) + y - t * (y6r * k0c + y7r * k1c);
      if (r == c) pn += kn[KN_R0 + r] + (r == 1 ? kn[KN_EY] : 0.0);
    }
    if (r == 0) {
      st[ST_K + c] = k0c;
      st[ST_K + 8 + c] = k1c;
    }
    if (lane == 8) {
      st[ST_HI + 0] = hi00;
      st[ST_HI + 1] = hi01;
      st[ST_HI + 2] = hi11;
    }
    if (i >= 1) MP[r * MROW + c] = pn;
    __syncthreads();
  }
}

// Riccati vector solve for nrhs (1 or 2) right-hand sides held in the knots' rhs regions
// (q_z @ +0..7, q_v @ +8,9 of region s); the step (dz, dv) overwrites them.  dz_0 = 0.
__device__ void riccati_solve(const Lds& L, int lane, int nrhs) {
  const int N = L.N, s = lane >> 3, r = lane & 7;
  const bool on = s < nrhs;
  const int reg = KN_R0 + 10 * s;
  double* pvb = L.tail() + TL_PV;
  int cur = 0;
  if (on) pvb[cur * 16 + s * 8 + r] = L.kn(N - 1)[reg + r];
  __syncthreads();
  for (int i = N - 2; i >= 0; --i) {
    double* st = L.st(i);
    if (on) {
      const double* pv = pvb + cur * 16 + s * 8;
      const double* kn = L.kn(i);
      const double t = st[ST_DT];
      double p[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) p[k] = pv[k];
      double wr = (r >= 6) ? pv[r] : 0.0, w6 = pv[6], w7 = pv[7];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        wr += st[r * 6 + k] * p[k];
        w6 += st[36 + k] * p[k];
        w7 += st[42 + k] * p[k];
      }
      const double hv0 = kn[reg + 8] + t * w6, hv1 = kn[reg + 9] + t * w7;
      if (i >= 1) pvb[(cur ^ 1) * 16 + s * 8 + r] = kn[reg + r] + wr - (st[ST_K + r] * hv0 + st[ST_K + 8 + r] * hv1);
      if (r == 0) {
        st[ST_KFF + 2 * s + 0] = st[ST_HI + 0] * hv0 + st[ST_HI + 1] * hv1;
        st[ST_KFF + 2 * s + 1] = st[ST_HI + 1] * hv0 + st[ST_HI + 2] * hv1;
      }
    }
    cur ^= 1;
    __syncthreads();
  }
  if (on) L.kn(0)[reg + r] = 0.0;
  __syncthreads();
  for (int i = 0; i < N - 1; ++i) {
    if (on) {
      const double* st = L.st(i);
      double* kn = L.kn(i);
      const double t = st[ST_DT];
      double d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = kn[reg + k];
      double dv0 = -st[ST_KFF + 2 * s], dv1 = -st[ST_KFF + 2 * s + 1];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        dv0 -= st[ST_K + k] * d[k];
        dv1 -= st[ST_K + 8 + k] * d[k];
      }
      const double du0 = d[6] + t * dv0, du1 = d[7] + t * dv1;
      double nx;
      if (r < 6) {
        nx = st[36 + r] * du0 + st[42 + r] * du1;
#pragma unroll
        for (int k = 0; k < 6; ++k) nx += st[k * 6 + r] * d[k];
      } else {
        nx = (r == 6) ? du0 : du1;
      }
      L.kn(i + 1)[reg + r] = nx;
      if (r == 0) {
        kn[reg + 8] = dv0;
        kn[reg + 9] = dv1;
      }
    }
    __syncthreads();
  }
}

// Closed-loop rollout z_{i+1} = Abar z_i + Bbar v_i + gbar, v_i = -K_i z_i (absolute variables).
// The linearised model can be open-loop unstable (|eig A| > 1 at low speed with dt = 25 ms), so
// the start trajectory is generated under the stabilising Riccati feedback.
__device__ void feedback_rollout(const Lds& L, int lane) {
  const int N = L.N, r = lane & 7;
  const bool on = lane < 8;
  for (int i = 0; i < N - 1; ++i) {
    if (on) {
      const double* st = L.st(i);
      double* kn = L.kn(i);
      const double t = st[ST_DT];
      double d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] = kn[k];
      double v0 = 0.0, v1 = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v0 -= st[ST_K + k] * d[k];
        v1 -= st[ST_K + 8 + k] * d[k];
      }
      const double u0 = d[6] + t * v0, u1 = d[7] + t * v1;
      double nx;
      if (r < 6) {
        nx = st[ST_G + r] + st[36 + r] * u0 + st[42 + r] * u1;
#pragma unroll
        for (int k = 0; k < 6; ++k) nx += st[k * 6 + r] * d[k];
      } else {
        nx = (r == 6) ? u0 : u1;
      }
      L.kn(i + 1)[r] = nx;
      if (r == 0) {
        kn[8] = v0;
        kn[9] = v1;
      }
    }
    __syncthreads();
  }
}

template <int KQ>
__global__ __launch_bounds__(64, (KQ <= 4 ? 2 : 1)) void lmpc_solve_kernel(lmpc_params P, int B, const double* __restrict__ ws_lin,
                                                        const double* __restrict__ x_ic, const double* __restrict__ u_ic,
                                                        const double* __restrict__ T_ref, const double* __restrict__ bl,
                                                        const double* __restrict__ br, const double* __restrict__ vref,
                                                        double* __restrict__ X_out, double* __restrict__ U_out,
                                                        double* __restrict__ dU_out, int* __restrict__ status_out,
                                                        int* __restrict__ iters_out, double* __restrict__ kkt_out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int N = P.N, NS = N - 1;
  Lds L{lds, N};
  double* T = L.tail();
  double* ct = T + TL_CT;

  // ---------------- load: linearisation records, per-knot data, constant tables ----------------
  {
    const double* wsb = ws_lin + (size_t)b * NS * LMPC_LIN_RECORD;
    for (int e = lane; e < NS * LMPC_LIN_RECORD; e += 64) {
      const int i = e / LMPC_LIN_RECORD, o = e - i * LMPC_LIN_RECORD;
      L.st(i)[o] = wsb[e];
    }
    for (int i = lane; i < NS; i += 64) L.st(i)[ST_DT] = T_ref[(size_t)i * B + b];
    for (int i = lane; i < N; i += 64) {
      double* kn = L.kn(i);
      kn[KN_QLIN] = P.learning ? 0.0 : (i == N - 1 ? P.qv_term : P.qv_stage) * vref[(size_t)i * B + b];
      kn[8] = 0.0;
      kn[9] = 0.0;
    }
    if (lane < 6) {
      L.kn(0)[lane] = x_ic[(size_t)lane * B + b];
      ct[CT_QD + lane] = P.learning ? 0.0 : P.Qd[lane];
      ct[CT_QT + lane] = P.learning ? 0.0 : P.Qt[lane];
      ct[CT_HI + lane] = P.x_max[lane];
      ct[CT_LO + lane] = P.x_min[lane];
    } else if (lane < 8) {
      L.kn(0)[lane] = u_ic[(size_t)(lane - 6) * B + b];
      ct[CT_HI + lane] = P.u_hi[lane - 6];
      ct[CT_LO + lane] = P.u_lo[lane - 6];
    } else if (lane < 10) {
      ct[CT_HI + lane] = P.v_hi[lane - 8];
      ct[CT_LO + lane] = P.v_lo[lane - 8];
    } else if (lane < 14) {
      ct[CT_QU + lane - 10] = P.Qu[lane - 10];
    } else if (lane < 18) {
      ct[CT_SV + lane - 14] = P.Sv[lane - 14];
    }
  }
  __syncthreads();

  // ---------------- slot ownership ----------------
  // slot j = lane + 64 q  ->  knot i = j / 11, kind sl = j % 11.  Kinds 0..9 address the primal
  // component (z[0..7], v[0..1]) at offset sl of the knot record; kind 10 is the track boundary
  // row pair on e_y (offset 1) which also carries the shared slack sigma.
  int s_i[KQ], s_sl[KQ];
  bool s_au[KQ], s_al[KQ];  // upper / lower row present
  double s_hi[KQ], s_lo[KQ], s_tu[KQ], s_tl[KQ], s_lu[KQ], s_ll[KQ], s_pu[KQ], s_pl[KQ];
  int m_rows = 0;
  const bool has_sigma = P.has_sigma != 0;
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int j = lane + 64 * q;
    const bool valid = j < NSLOT * N;
    const int i = valid ? j / NSLOT : 0;
    const int sl = valid ? j - i * NSLOT : 0;
    s_i[q] = i;
    s_sl[q] = valid ? sl : -1;
    double hi = INFINITY, lo = -INFINITY;
    bool on = false;
    if (valid) {
      if (sl < SL_EY) {
        hi = ct[CT_HI + sl];
        lo = ct[CT_LO + sl];
        on = (sl < SL_U) ? (i >= 1 && i <= N - 2) : (sl < SL_V ? (i >= 1) : (i <= N - 2));
      } else {
        hi = bl[(size_t)i * B + b] - P.marg;
        lo = br[(size_t)i * B + b] + P.marg;
        on = has_sigma || i >= 1;
      }
    }
    s_hi[q] = hi;
    s_lo[q] = lo;
    s_au[q] = on && (hi < INFINITY);
    s_al[q] = on && (lo > -INFINITY);
    m_rows += (s_au[q] ? 1 : 0) + (s_al[q] ? 1 : 0);
    s_tu[q] = s_tl[q] = 1.0;
    s_lu[q] = s_ll[q] = 0.0;
    s_pu[q] = s_pl[q] = 0.0;
  }
  const double m_tot = wave_sum((double)m_rows) + (has_sigma ? 1.0 : 0.0);

  // knot-0 feasibility: the state box applies to x_0 = x_ic (racing_mpc.cpp:147,201)
  bool feasible = true;
  {
    bool ok = true;
    if (lane < 6) {
      const double v = L.kn(0)[lane];
      ok = (v <= ct[CT_HI + lane]) && (v >= ct[CT_LO + lane]);
    }
    if (lane == 6 && !has_sigma) {
      const double ey = L.kn(0)[1];
      ok = (ey <= bl[b] - P.marg) && (ey >= br[b] + P.marg);
    }
    feasible = wave_min(ok ? 1.0 : 0.0) > 0.5;
  }

  double sigma = 0.0, ts = 0.1, lams = 0.0;

  // ---------------- start point: minimiser of the cost over the dynamics alone ----------------
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int sl = s_sl[q];
    if (sl < 0) continue;
    double* kn = L.kn(s_i[q]);
    if (sl < SL_EY) {
      kn[KN_R0 + sl] = 0.0;
    } else {
      kn[KN_EY] = 0.0;
      kn[KN_CSIG] = 0.0;
    }
  }
  __syncthreads();
  riccati_factor(L, lane);
  feedback_rollout(L, lane);

  const double tau = 0.995, mu0 = 1.0;
  int status = LMPC_SOLVE_MAX_ITER, it = 0;
  double mu = 0.0, rdmax = 0.0, last_step = 0.0, hsig = 0.0, ce = 0.0;
  const int max_iter = feasible ? P.max_iter : 0;

  // it == -1 is the start-point Newton step (all row weights zero, full step); it >= 0 the
  // interior-point iterations.
  for (it = -1; it <= max_iter; ++it) {
    const bool ipm = it >= 0;
    // ======== rows: complementarity, residual, barrier weights ========
    if (ipm) {
      double musum = 0.0, rdl = 0.0, eysum = 0.0;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int sl = s_sl[q];
        if (sl < 0) continue;
        double* kn = L.kn(s_i[q]);
        const double val = kn[sl < SL_EY ? sl : 1];
        const double sg = (sl == SL_EY && has_sigma) ? sigma : 0.0;
        double thu = 0.0, thd = 0.0;
        if (s_au[q]) {
          thu = s_lu[q] / s_tu[q];
          musum += s_lu[q] * s_tu[q];
          rdl = fmax(rdl, fabs(val - sg + s_tu[q] - s_hi[q]));
        }
        if (s_al[q]) {
          thd = s_ll[q] / s_tl[q];
          musum += s_ll[q] * s_tl[q];
          rdl = fmax(rdl, fabs(-val - sg + s_tl[q] + s_lo[q]));
        }
        if (sl < SL_EY) {
          kn[KN_R0 + sl] = thu + thd;
        } else {
          kn[KN_EY] = thu + thd;
          kn[KN_CSIG] = has_sigma ? thd - thu : 0.0;
          if (has_sigma) eysum += thu + thd;
        }
      }
      musum = wave_sum(musum);
      rdmax = wave_max(rdl);
      hsig = P.qsig + wave_sum(eysum);
      if (has_sigma) {
        musum += ts * lams;
        rdmax = fmax(rdmax, fabs(-sigma + ts));
        hsig += lams / ts;
      }
      mu = musum / m_tot;
      if (!(mu == mu) || !(rdmax == rdmax)) {
        status = LMPC_SOLVE_INFEASIBLE;
        break;
      }
      if (mu <= P.tol && rdmax <= 1e-9) {
        status = LMPC_SOLVE_OPTIMAL;
        break;
      }
      if (it == max_iter) break;
      __syncthreads();
      riccati_factor(L, lane);
    }

    double sigc = 0.0, alpha = 1.0, dsigma = 0.0, dts = 0.0, dlams = 0.0;
    double d_tu[KQ], d_tl[KQ], d_lu[KQ], d_ll[KQ], d_val[KQ];
    const int npass = ipm ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
      // ======== gradient: cost gradient + row coefficients, written by the component owner ========
      double sgsum = 0.0;  // sum of boundary-row coefficients entering the sigma gradient
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int sl = s_sl[q];
        if (sl < 0) continue;
        const int i = s_i[q];
        double* kn = L.kn(i);
        const double val = kn[sl < SL_EY ? sl : 1];
        const double sg = (sl == SL_EY && has_sigma) ? sigma : 0.0;
        double cu = 0.0, cd = 0.0;
        if (ipm && s_au[q]) {
          cu = (s_lu[q] / s_tu[q]) * (val - sg + s_tu[q] - s_hi[q]);
          if (pass == 1) cu += (sigc * mu - s_pu[q]) / s_tu[q];
        }
        if (ipm && s_al[q]) {
          cd = (s_ll[q] / s_tl[q]) * (-val - sg + s_tl[q] + s_lo[q]);
          if (pass == 1) cd += (sigc * mu - s_pl[q]) / s_tl[q];
        }
        if (sl < SL_EY) {
          double g;
          if (sl < SL_U) {
            g = (i == N - 1 ? ct[CT_QT + sl] : ct[CT_QD + sl]) * val + (sl == 3 ? kn[KN_QLIN] : 0.0);
          } else if (sl < SL_V) {
            g = (i >= 1) ? ct[CT_QU + (sl - SL_U) * 2] * kn[6] + ct[CT_QU + (sl - SL_U) * 2 + 1] * kn[7] : 0.0;
          } else {
            g = (i <= N - 2) ? ct[CT_SV + (sl - SL_V) * 2] * kn[8] + ct[CT_SV + (sl - SL_V) * 2 + 1] * kn[9] : 0.0;
          }
          kn[KN_R0 + sl] = g + cu - cd;
          if (pass == 0) kn[KN_R1 + sl] = 0.0;
        } else {
          kn[KN_EY] = cu - cd;
          if (has_sigma) sgsum += cu + cd;
        }
      }
      __syncthreads();
      for (int i = lane; i < N; i += 64) {
        double* kn = L.kn(i);
        kn[KN_R0 + 1] += kn[KN_EY];
        if (pass == 0) kn[KN_R1 + 1] = (i >= 1) ? kn[KN_CSIG] : 0.0;
      }
      __syncthreads();
      // ======== Newton step: predictor together with the Schur vector, then the corrector ========
      riccati_solve(L, lane, (pass == 0 && ipm && has_sigma) ? 2 : 1);
      // ======== boundary slack by Schur complement ========
      double cfs = 0.0;
      if (ipm && has_sigma) {
        double cep = 0.0, cap = 0.0;
#pragma unroll
        for (int q = 0; q < KQ; ++q)
          if (s_sl[q] == SL_EY && s_i[q] >= 1) {
            const double* kn = L.kn(s_i[q]);
            cap += kn[KN_CSIG] * kn[KN_R0 + 1];
            cep += kn[KN_CSIG] * kn[KN_R1 + 1];
          }
        const double ca = wave_sum(cap);
        if (pass == 0) ce = wave_sum(cep);
        cfs = (lams / ts) * (-sigma + ts);
        if (pass == 1) cfs += (sigc * mu - dts * dlams) / ts;
        const double qsg = P.qsig * sigma - wave_sum(sgsum) - cfs;
        dsigma = -(qsg + ca) / (hsig + ce);
      }
      // ======== row steps, largest feasible step ========
      double amax = 1.0;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int sl = s_sl[q];
        d_val[q] = 0.0;
        if (sl < 0) continue;
        const double* kn = L.kn(s_i[q]);
        const int off = sl < SL_EY ? sl : 1;
        const double dval = kn[KN_R0 + off] + ((ipm && has_sigma) ? dsigma * kn[KN_R1 + off] : 0.0);
        d_val[q] = dval;
        if (!ipm) continue;
        const double val = kn[off];
        const double sg = (sl == SL_EY && has_sigma) ? sigma : 0.0;
        const double dsg = (sl == SL_EY && has_sigma) ? dsigma : 0.0;
        if (s_au[q]) {
          const double t = s_tu[q], lam = s_lu[q], th = lam / t, rd = val - sg + t - s_hi[q];
          double cf = th * rd;
          if (pass == 1) cf += (sigc * mu - s_pu[q]) / t;
          const double dt_ = -rd - (dval - dsg);
          const double dl_ = -lam + cf - th * rd - th * dt_;
          d_tu[q] = dt_;
          d_lu[q] = dl_;
          if (dt_ < 0.0) amax = fmin(amax, -t / dt_);
          if (dl_ < 0.0) amax = fmin(amax, -lam / dl_);
        }
        if (s_al[q]) {
          const double t = s_tl[q], lam = s_ll[q], th = lam / t, rd = -val - sg + t + s_lo[q];
          double cf = th * rd;
          if (pass == 1) cf += (sigc * mu - s_pl[q]) / t;
          const double dt_ = -rd - (-dval - dsg);
          const double dl_ = -lam + cf - th * rd - th * dt_;
          d_tl[q] = dt_;
          d_ll[q] = dl_;
          if (dt_ < 0.0) amax = fmin(amax, -t / dt_);
          if (dl_ < 0.0) amax = fmin(amax, -lam / dl_);
        }
      }
      if (!ipm) break;
      amax = wave_min(amax);
      if (has_sigma) {
        const double th = lams / ts, rds = -sigma + ts;
        dts = -rds + dsigma;
        dlams = -lams + cfs - th * rds - th * dts;
        if (dts < 0.0) amax = fmin(amax, -ts / dts);
        if (dlams < 0.0) amax = fmin(amax, -lams / dlams);
      }
      if (pass == 0) {
        double sacc = 0.0;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          if (s_sl[q] < 0) continue;
          if (s_au[q]) {
            sacc += (s_tu[q] + amax * d_tu[q]) * (s_lu[q] + amax * d_lu[q]);
            s_pu[q] = d_tu[q] * d_lu[q];
          }
          if (s_al[q]) {
            sacc += (s_tl[q] + amax * d_tl[q]) * (s_ll[q] + amax * d_ll[q]);
            s_pl[q] = d_tl[q] * d_ll[q];
          }
        }
        sacc = wave_sum(sacc);
        if (has_sigma) sacc += (ts + amax * dts) * (lams + amax * dlams);
        const double ratio = (sacc / m_tot) / mu;
        sigc = ratio * ratio * ratio;
      } else {
        alpha = fmin(1.0, tau * amax);
      }
      __syncthreads();
    }

    // ======== update (component owners move the primal; rows stay in registers) ========
    double stepmax = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int sl = s_sl[q];
      if (sl < 0) continue;
      const int i = s_i[q];
      if (sl < SL_EY && (i >= 1 || sl >= SL_V)) {
        L.kn(i)[sl] += alpha * d_val[q];
        stepmax = fmax(stepmax, fabs(alpha * d_val[q]));
      }
      if (ipm) {
        if (s_au[q]) {
          s_tu[q] += alpha * d_tu[q];
          s_lu[q] += alpha * d_lu[q];
        }
        if (s_al[q]) {
          s_tl[q] += alpha * d_tl[q];
          s_ll[q] += alpha * d_ll[q];
        }
      }
    }
    __syncthreads();
    if (ipm) {
      last_step = wave_max(stepmax);
      if (has_sigma) {
        sigma += alpha * dsigma;
        ts += alpha * dts;
        lams += alpha * dlams;
      }
    } else {
      // ---- slacks and multipliers at the start point: t = max(slack, 0.1 range), lam = mu0 / t ----
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int sl = s_sl[q];
        if (sl < 0) continue;
        const double val = L.kn(s_i[q])[sl < SL_EY ? sl : 1];
        double range = (s_hi[q] < INFINITY && s_lo[q] > -INFINITY) ? (s_hi[q] - s_lo[q]) : 1.0;
        if (!(range > 1e-3)) range = 1e-3;
        const double thr = 0.1 * range;
        if (s_au[q]) {
          s_tu[q] = fmax(s_hi[q] - val, thr);
          s_lu[q] = mu0 / s_tu[q];
        }
        if (s_al[q]) {
          s_tl[q] = fmax(val - s_lo[q], thr);
          s_ll[q] = mu0 / s_tl[q];
        }
      }
      sigma = 0.0;
      ts = 0.1;
      lams = has_sigma ? mu0 / ts : 0.0;
    }
  }
  if (it < 0) it = 0;
  if (!feasible) status = LMPC_SOLVE_INFEASIBLE;

  // ---------------- write back: X [6][N][B], U, dU [2][N-1][B] ----------------
  __syncthreads();
  for (int e = lane; e < 6 * N; e += 64) {
    const int k = e / N, i = e - k * N;
    X_out[(size_t)(k * N + i) * B + b] = L.kn(i)[k];
  }
  for (int e = lane; e < 2 * NS; e += 64) {
    const int k = e / NS, i = e - k * NS;
    U_out[(size_t)(k * NS + i) * B + b] = L.kn(i + 1)[6 + k];
    dU_out[(size_t)(k * NS + i) * B + b] = L.kn(i)[8 + k];
  }
  if (lane == 0) {
    status_out[b] = status;
    iters_out[b] = it;
    if (kkt_out) {
      kkt_out[0 * (size_t)B + b] = last_step;
      kkt_out[1 * (size_t)B + b] = rdmax;
      kkt_out[2 * (size_t)B + b] = mu;
      kkt_out[3 * (size_t)B + b] = sigma;
    }
  }
}

template __global__ void lmpc_solve_kernel<2>(lmpc_params, int, const double*, const double*, const double*,
                                              const double*, const double*, const double*, const double*, double*,
                                              double*, double*, int*, int*, double*);
template __global__ void lmpc_solve_kernel<4>(lmpc_params, int, const double*, const double*, const double*,
                                              const double*, const double*, const double*, const double*, double*,
                                              double*, double*, int*, int*, double*);
template __global__ void lmpc_solve_kernel<7>(lmpc_params, int, const double*, const double*, const double*,
                                              const double*, const double*, const double*, const double*, double*,
                                              double*, double*, int*, int*, double*);
template __global__ void lmpc_solve_kernel<11>(lmpc_params, int, const double*, const double*, const double*,
                                               const double*, const double*, const double*, const double*, double*,
                                               double*, double*, int*, int*, double*);
template __global__ void lmpc_solve_kernel<14>(lmpc_params, int, const double*, const double*, const double*,
                                               const double*, const double*, const double*, const double*, double*,
                                               double*, double*, int*, int*, double*);
