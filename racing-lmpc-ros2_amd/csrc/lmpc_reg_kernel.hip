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
// Two kernels: lmpc_reg_residual_kernel (once per lap upload: one thread per sample) and lmpc_regress_kernel (one
// wavefront per (problem, stage): lanes stride over the samples, 21 + 6 NOUT weighted sums per lane in registers,
// one batched wave reduction, the (ns+nc+1)^2 SPD solve done redundantly in registers).
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

// WS_LAYOUT: update the handle's linearisation workspace [B][N-1][54]; otherwise the A/B/g arrays of lmpc_linearize_batch.
template <int NF, int NOUT, bool WS_LAYOUT>
__global__ __launch_bounds__(64) void lmpc_regress_kernel(int N, int B, lmpc_regression_spec spec, int total,
                                                          const int* __restrict__ lap_end, const double* __restrict__ x,
                                                          const double* __restrict__ u, const double* __restrict__ yres,
                                                          const double* __restrict__ X_ref, const double* __restrict__ U_ref,
                                                          double* __restrict__ outA, double* __restrict__ outB,
                                                          double* __restrict__ outg) {
  constexpr int NM = NF + 1;
  constexpr int NQ = NM * (NM + 1) / 2;
  const int NS = N - 1;
  const int b = blockIdx.x / NS, i = blockIdx.x - b * NS;
  const int lane = threadIdx.x;
  const int ns = spec.n_in_state;
  double q[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f)
    q[f] = f < ns ? X_ref[((size_t)spec.in_state[f] * N + i) * B + b] : U_ref[((size_t)spec.in_ctrl[f - ns] * NS + i) * B + b];
  const double h = spec.dist_max, ih = 1.0 / h;
  double acc[NQ + NOUT * NM];
#pragma unroll
  for (int a = 0; a < NQ + NOUT * NM; ++a) acc[a] = 0.0;
  for (int j = lane; j < total; j += 64) {
    if (lap_end[j]) continue;
    double m[NM];
    double d2 = 0.0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      m[f] = f < ns ? x[(size_t)j * 6 + spec.in_state[f]] : u[(size_t)j * 2 + spec.in_ctrl[f - ns]];
      const double e = m[f] - q[f];
      d2 += e * e;
    }
    m[NF] = 1.0;
    const double d = sqrt(d2);
    if (!(d < h)) continue;
    const double s = 1.0 - (d * ih) * (d * ih);
    const double w = 0.75 * ih * s * s;
    int a = 0;
#pragma unroll
    for (int r = 0; r < NM; ++r)
#pragma unroll
      for (int c = r; c < NM; ++c) acc[a++] += w * m[r] * m[c];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const double wy = w * yres[(size_t)j * 6 + spec.out[o]];
#pragma unroll
      for (int r = 0; r < NM; ++r) acc[a++] += wy * m[r];
    }
  }
  // wave reduction of all sums in lock-step
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) {
#pragma unroll
    for (int a = 0; a < NQ + NOUT * NM; ++a) acc[a] += __shfl_xor(acc[a], msk);
  }
  // "if there are no points left, skip the regression" (safe_set.cpp:207-210): the weight sum is M'KM's last entry
  if (!(acc[NQ - 1] > 0.0)) return;
  // Cholesky of Q = M'KM + 1e-3 I (every lane, wave-uniform data)
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
    // lane f adds coefficient f of row spec.out[o]
    const int row = spec.out[o];
    double val = 0.0;
#pragma unroll
    for (int f = 0; f < NM; ++f)
      if (f == lane) val = R[f];
    if (lane < NM) {
      if (lane < NF) {
        const int col = lane < ns ? spec.in_state[lane] : 6 + spec.in_ctrl[lane - ns];  // column of [A B]
        if (WS_LAYOUT)
          outA[((size_t)b * NS + i) * LMPC_LIN_RECORD + col * 6 + row] += val;
        else if (col < 6)
          outA[((size_t)(row * 6 + col) * NS + i) * B + b] += val;
        else
          outB[((size_t)(row * 2 + (col - 6)) * NS + i) * B + b] += val;
      } else {
        if (WS_LAYOUT)
          outA[((size_t)b * NS + i) * LMPC_LIN_RECORD + 48 + row] += val;
        else
          outg[((size_t)row * NS + i) * B + b] += val;
      }
    }
  }
}

template __global__ void lmpc_regress_kernel<5, 3, true>(int, int, lmpc_regression_spec, int, const int*, const double*, const double*,
                                                         const double*, const double*, const double*, double*, double*, double*);
template __global__ void lmpc_regress_kernel<5, 3, false>(int, int, lmpc_regression_spec, int, const int*, const double*, const double*,
                                                          const double*, const double*, const double*, double*, double*, double*);
template __global__ void lmpc_regress_kernel<8, 6, true>(int, int, lmpc_regression_spec, int, const int*, const double*, const double*,
                                                         const double*, const double*, const double*, double*, double*, double*);
template __global__ void lmpc_regress_kernel<8, 6, false>(int, int, lmpc_regression_spec, int, const int*, const double*, const double*,
                                                          const double*, const double*, const double*, double*, double*, double*);
