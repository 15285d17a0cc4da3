// lmpc_reg_kernel.hip -- error-dynamics regression on the recorded laps (BASELINE config 5).
//
// What it replaces: SSTrajectory::query(RegQuery) (racing_trajectory/src/safe_set.cpp:56-114) and
// SafeSetManager::query(RegQuery) (:182-245): per linearisation point, a kernel-weighted ridge regression of the
// one-step model error over the lap samples within dist_max in feature space, added onto (A, B, g).
//   features  z_j = [x_j[in_state]; u_j[in_ctrl]]   (the last sample of a lap has no successor and is skipped, :68-76)
//   weights   K_j = 0.75 / h (1 - (d_j / h)^2)^2  for  d_j = ||z_j - q|| < h                      (:84-87, :224-225)
//   M = [z_j' 1],  Q = M' K M + 1e-3 I,  b_r = M' K y_r,  R_r = Q^-1 b_r      (upstream writes b_r = - M' K y_r, :229-231)
//   A[r, in_state] += R_r[0:ns],  B[r, in_ctrl] += R_r[ns:ns+nc],  g[r] += R_r[-1]               (:235-242)
// The reference has no caller and no test for this query, and two of its expressions do not type-check as written
// (the nominal model is handed the in_state rows instead of the state; the residual is taken on the in_state rows
// for every output).  The restatement (oracle/regression.py documents the same reading) evaluates the nominal RK4
// step on the full recorded state and regresses, for output row r, the residual of that row:
//   y_r,j = x_{j+1}[r] - f_d(x_j, u_j, k_j, dt_j)[r],   dt_j = t_{j+1} - t_j.
// As written upstream dt_j = t_j - t_{j+1} is negative (process_lap_data, :130-135: the model steps backwards in time, the
// "residual" is about twice the true step) and the right-hand side carries a minus sign, so the correction points away
// from the data; spec.as_written = 1 reproduces that literally, the default is the regression that reduces the one-step
// error (include/lmpc_hip.h).
//
// Three kernels.  Once per lap upload: lmpc_reg_residual_kernel (one thread per sample) and lmpc_reg_pack_kernel, which
// gathers the samples that have a successor into one dense table  tab[v] = [z (NF) | y (NOUT)]  (64 B per sample for
// (5, 3)), padded to a multiple of four with rows no query can reach.  Per solve: lmpc_regress_kernel with ONE LANE PER
// QUERY (problem, stage) -- stages of a problem on neighbouring lanes, they are neighbours in feature space too -- and
// the sample loop wave-uniform: a sample's row arrives by scalar loads (s_load, four samples in flight), every lane
// tests its own distance and, when any lane of the wave is within the bandwidth, adds w m m' and w y m' to its own
// 21 + 6 NOUT sums as FMAs with scalar operands.  No cross-lane reduction, no per-lane gather; each lane then factors
// its own (NF+1)^2 system.  The accumulation is the product W Phi (queries x samples times samples x 39) and would map
// onto v_mfma_f64_16x16x4 -- measured on the MI355X (scratch/mfma_f64_rate.hip): 47 TFLOP/s with two waves per SIMD and twelve
// independent accumulator tiles (105 cycles per instruction), 39 with five fp64 FMAs between MFMAs; the 64 M MFMAs of the
// 32768 x 19 x 2200 workload with 39 columns padded to 48 would take 2.8-3.3 ms, the vector kernel takes 2.5: not built.
// Per pair the vector loop spends 54 instructions: d^2 = (|q|^2 + |z|^2) - 2 z.q (one add + NF FMAs, |z|^2 from the table's
// tail), K / c0 = max(1 - d^2/h^2, 0)^2 (the bandwidth test is the max), NF products w z, 21 + 6 NOUT FMAs; c0 once at the end
// (round 4: 62 -> 54 instructions, 2.86 -> 2.52 ms; the group's scalar loads up front with one wait: 2.38 ms).
// (The first version ran one wavefront per query with the lanes striding the samples: three dependent gathers per
// iteration behind two branches and a 39-value shuffle reduction -- latency-bound at 13 ms per 32768 x 19 queries over
// 2200 samples; this one is FP64-issue-bound.)
#include <hip/hip_runtime.h>

#include "lmpc_device.h"
#include "lmpc_dynamics.hip.h"

__global__ void lmpc_reg_residual_kernel(lmpc_vehicle veh, int total, int as_written, const int* __restrict__ lap_end, const double* __restrict__ x,
                                         const double* __restrict__ u, const double* __restrict__ k,
                                         const double* __restrict__ t, double* __restrict__ y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  double r[6] = {0, 0, 0, 0, 0, 0};
  if (!lap_end[j]) {  // the last sample of a lap has no successor
    double xs[6], us[2], xp[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) xs[c] = x[(size_t)j * 6 + c];
    us[0] = u[(size_t)j * 2];
    us[1] = u[(size_t)j * 2 + 1];
    lmpc_fd(veh, xs, us, k[j], as_written ? t[j] - t[j + 1] : t[j + 1] - t[j], xp);
#pragma unroll
    for (int c = 0; c < 6; ++c) r[c] = x[(size_t)(j + 1) * 6 + c] - xp[c];
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) y[(size_t)j * 6 + c] = r[c];
}

// tab[v][NF + NOUT] for the valid samples (valid[v] = index of a sample that is not the last of its lap), v < nvalid;
// rows nvalid .. npad-1: features and residuals 0, |z|^2 = 1e30 (out of every bandwidth); zz[v] = |z_v|^2
__global__ void lmpc_reg_pack_kernel(lmpc_regression_spec spec, int nvalid, int npad, const int* __restrict__ valid,
                                     const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ y,
                                     double* __restrict__ tab, double* __restrict__ zz) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= npad) return;
  const int ns = spec.n_in_state, nf = ns + spec.n_in_ctrl, nrow = nf + spec.n_out;
  double* row = tab + (size_t)v * nrow;
  if (v >= nvalid) {  // (features 0 with |z|^2 = 1e30: d^2 = 1e30 for every query, and no 0 * inf in the sums)
    for (int f = 0; f < nf; ++f) row[f] = 0.0;
    for (int o = 0; o < spec.n_out; ++o) row[nf + o] = 0.0;
    zz[v] = 1e30;
    return;
  }
  const int j = valid[v];
  double s = 0.0;
  for (int f = 0; f < nf; ++f) {
    row[f] = f < ns ? x[(size_t)j * 6 + spec.in_state[f]] : u[(size_t)j * 2 + spec.in_ctrl[f - ns]];
    s = __builtin_fma(row[f], row[f], s);
  }
  zz[v] = s;
  for (int o = 0; o < spec.n_out; ++o) row[nf + o] = y[(size_t)j * 6 + spec.out[o]];
}

// WS_LAYOUT: update the handle's linearisation workspace [B][N-1][54]; otherwise the A/B/g arrays of lmpc_linearize_batch.
template <int NF, int NOUT, bool WS_LAYOUT>
__global__ __launch_bounds__(64) void lmpc_regress_kernel(int N, int B, lmpc_regression_spec spec, int npad,
                                                          const double* __restrict__ tab, const double* __restrict__ zz, const double* __restrict__ X_ref,
                                                          const double* __restrict__ U_ref, double* __restrict__ outA,
                                                          double* __restrict__ outB, double* __restrict__ outg) {
  constexpr int NM = NF + 1;
  constexpr int NQ = NM * (NM + 1) / 2;
  constexpr int NROW = NF + NOUT;
  constexpr int UNR = 4;
  const int NS = N - 1;
  const long long gq = (long long)blockIdx.x * 64 + threadIdx.x;
  const bool live = gq < (long long)B * NS;
  const long long gqc = live ? gq : 0;
  const int b = (int)(gqc / NS), i = (int)(gqc - (long long)b * NS);
  const int ns = spec.n_in_state;
  double q[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f)
    q[f] = f < ns ? X_ref[((size_t)spec.in_state[f] * N + i) * B + b] : U_ref[((size_t)spec.in_ctrl[f - ns] * NS + i) * B + b];
  const double h = spec.dist_max, h2 = h * h, nih2 = -1.0 / h2, c0 = 0.75 / h;
  // d^2 = (|q|^2 + |z|^2) - 2 z.q: one add and NF FMAs per pair instead of NF subtractions and NF FMAs (|z|^2 is formed on the
  // scalar side of the loop: it is the same for every lane).  A dead lane's query sits out of every bandwidth.
  double qm2[NF], qq = 0.0;
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (!live) q[f] = 1e30;
    qm2[f] = -2.0 * q[f];
    qq = __builtin_fma(q[f], q[f], qq);
  }
  double acc[NQ + NOUT * NM];
#pragma unroll
  for (int a = 0; a < NQ + NOUT * NM; ++a) acc[a] = 0.0;
  for (int j0 = 0; j0 < npad; j0 += UNR) {
    double row[UNR][NROW], sq[UNR];
#pragma unroll
    for (int t = 0; t < UNR; ++t)
#pragma unroll
      for (int c = 0; c < NROW; ++c) row[t][c] = tab[(size_t)(j0 + t) * NROW + c];  // wave-uniform address: scalar loads
    double zn[UNR];
#pragma unroll
    for (int t = 0; t < UNR; ++t) zn[t] = zz[j0 + t];
    // every row of the group is "used" here, in scalar registers: left alone, the compiler loads a sample's features, waits, tests
    // the distance, and only inside the hit branch loads its residuals and waits again -- two exposed scalar-cache round trips per
    // sample instead of one per group
    // (where the group fits the scalar registers: (8, 6) would need 120 of them)
    if constexpr (2 * (NROW + 1) * UNR <= 80) {
#pragma unroll
      for (int t = 0; t < UNR; ++t) {
#pragma unroll
        for (int c = 0; c < NROW; ++c) asm volatile("" : "+s"(row[t][c]));
        asm volatile("" : "+s"(zn[t]));
      }
    }
#pragma unroll
    for (int t = 0; t < UNR; ++t) {
      double s = qq + zn[t];
#pragma unroll
      for (int f = 0; f < NF; ++f) s = __builtin_fma(row[t][f], qm2[f], s);
      // K / c0 = (1 - (d/h)^2)^2 inside the bandwidth, 0 outside (safe_set.cpp:84-87): max(1 - d^2/h^2, 0)^2; c0 = 0.75/h
      // multiplies the sums once, after the loop
      sq[t] = fmax(__builtin_fma(s, nih2, 1.0), 0.0);
    }
#pragma unroll
    for (int t = 0; t < UNR; ++t) {
      if (!__any(sq[t] > 0.0)) continue;
      const double w = sq[t] * sq[t];
      double wm[NM];
#pragma unroll
      for (int r = 0; r < NF; ++r) wm[r] = w * row[t][r];
      wm[NF] = w;
      int a = 0;
#pragma unroll
      for (int r = 0; r < NM; ++r)
#pragma unroll
        for (int c = r; c < NM; ++c) {
          acc[a] += c < NF ? wm[r] * row[t][c] : wm[r];
          ++a;
        }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        const double yo = row[t][NF + o];
#pragma unroll
        for (int r = 0; r < NM; ++r) acc[a++] += wm[r] * yo;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NQ + NOUT * NM; ++a) acc[a] *= c0;
  // "if there are no points left, skip the regression" (safe_set.cpp:207-210): the weight sum is M'KM's last entry
  if (!live || !(acc[NQ - 1] > 0.0)) return;
  // Cholesky of Q = M'KM + 1e-3 I, this lane's own system
  double Lc[NM * NM];
  {
    double Q[NM * NM];
    int a = 0;
#pragma unroll
    for (int r = 0; r < NM; ++r)
#pragma unroll
      for (int c = r; c < NM; ++c) {
        Q[r * NM + c] = acc[a] + (r == c ? 1e-3 : 0.0);
        Q[c * NM + r] = Q[r * NM + c];
        ++a;
      }
#pragma unroll
    for (int jn = 0; jn < NM; ++jn) {
      double dd = Q[jn * NM + jn];
#pragma unroll
      for (int k = 0; k < jn; ++k) dd -= Lc[jn * NM + k] * Lc[jn * NM + k];
      const double id = 1.0 / sqrt(dd);
      Lc[jn * NM + jn] = id;  // reciprocal of the pivot
#pragma unroll
      for (int r = jn + 1; r < NM; ++r) {
        double tt = Q[r * NM + jn];
#pragma unroll
        for (int k = 0; k < jn; ++k) tt -= Lc[r * NM + k] * Lc[jn * NM + k];
        Lc[r * NM + jn] = tt * id;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    double yv[NM], R[NM];
#pragma unroll
    for (int r = 0; r < NM; ++r) {
      double tt = spec.as_written ? -acc[NQ + o * NM + r] : acc[NQ + o * NM + r];  // b = M'K y  (as written: -M'K y)
#pragma unroll
      for (int k = 0; k < r; ++k) tt -= Lc[r * NM + k] * yv[k];
      yv[r] = tt * Lc[r * NM + r];
    }
#pragma unroll
    for (int r = NM - 1; r >= 0; --r) {
      double tt = yv[r];
#pragma unroll
      for (int k = r + 1; k < NM; ++k) tt -= Lc[k * NM + r] * R[k];
      R[r] = tt * Lc[r * NM + r];
    }
    const int rowo = spec.out[o];
#pragma unroll
    for (int f = 0; f < NM; ++f) {
      if (f < NF) {
        const int col = f < ns ? spec.in_state[f] : 6 + spec.in_ctrl[f - ns];  // column of [A B]
        if (WS_LAYOUT)
          outA[((size_t)b * NS + i) * LMPC_LIN_RECORD + col * 6 + rowo] += R[f];
        else if (col < 6)
          outA[((size_t)(rowo * 6 + col) * NS + i) * B + b] += R[f];
        else
          outB[((size_t)(rowo * 2 + (col - 6)) * NS + i) * B + b] += R[f];
      } else {
        if (WS_LAYOUT)
          outA[((size_t)b * NS + i) * LMPC_LIN_RECORD + 48 + rowo] += R[f];
        else
          outg[((size_t)rowo * NS + i) * B + b] += R[f];
      }
    }
  }
}

#define LMPC_REG_INSTANTIATE(NF, NOUT, WS)                                                                                    \
  template __global__ void lmpc_regress_kernel<NF, NOUT, WS>(int, int, lmpc_regression_spec, int, const double*, const double*, const double*, \
                                                             const double*, double*, double*, double*);
LMPC_REG_INSTANTIATE(5, 3, true)
LMPC_REG_INSTANTIATE(5, 3, false)
LMPC_REG_INSTANTIATE(8, 6, true)
LMPC_REG_INSTANTIATE(8, 6, false)
