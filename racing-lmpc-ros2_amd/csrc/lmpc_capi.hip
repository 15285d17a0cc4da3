// lmpc_capi.hip -- host side of the C ABI declared in include/lmpc_hip.h (gfx950 / ROCm).
//
// The handle owns: the digested parameter block (the reference builds its parametric problem
// once in RacingMPC::RacingMPC, racing_mpc.cpp:31-202), one HIP stream, the linearisation
// workspace and the device copy of the safe set.  Nothing here computes on the hot path; the
// product fails loudly (negative return code + message) rather than falling back to a CPU path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lmpc_device.h"

template <bool WS_LAYOUT, typename io, int W>
__global__ void lmpc_linearize_kernel(lmpc_params, int, const io*, const io*, const io*, const io*, io*, io*, io*);
__global__ void lmpc_prepare_kernel(lmpc_params, int, lmpc_track, const double*, const int*, double, double, double, double*,
                                    double*, double*, double*, double*, double*, double*);
__global__ void lmpc_shift_kernel(lmpc_params, int, lmpc_track, const double*, const double*, const double*,
                                  const double*, const int*, double, double, double, double*, double*, double*, double*,
                                  double*, double*, double*);
__global__ void lmpc_plant_kernel(lmpc_params, int, lmpc_track, double*, const double*, double, int);
template <typename real, int KQ, int KS, typename io>
__global__ void lmpc_solve_kernel(lmpc_params, int, const io*, const io*, const io*, const io*, const io*, const io*,
                                  const io*, const io*, const io*, io*, io*, io*, io*, int*, int*, io*);
template <int KQ>
__global__ void lmpc_solve_warm_kernel(lmpc_params, int, const double*, const double*, const double*, const double*, const double*, const double*,
                                       const double*, double*, double*, double*, int*, int*, double*);
__global__ void lmpc_ss_query_kernel(int, int, int, int, const int*, const int*, const double*, double, const double*,
                                     double*, double*, int*, double*, int*);
__global__ void lmpc_reg_residual_kernel(lmpc_vehicle, int, int, const int*, const double*, const double*, const double*,
                                         const double*, double*);
__global__ void lmpc_reg_pack_kernel(lmpc_regression_spec, int, int, const int*, const double*, const double*, const double*, double*, double*);
template <int NF, int NOUT, bool WS_LAYOUT>
__global__ void lmpc_regress_kernel(int, int, lmpc_regression_spec, int, const double*, const double*, const double*, const double*,
                                    double*, double*, double*);
__global__ void lmpc_launch_order_kernel(int, const int*, int*);
__global__ void lmpc_collect_unverified_kernel(int, const int*, int*);
template <typename real, int KQ, int KS, typename io>
__global__ void lmpc_cleanup_kernel(lmpc_params, int, const int*, const int*, const io*, const io*, const io*, const io*, const io*,
                                    const io*, const io*, const io*, const io*, io*, io*, io*, io*, int*, int*, io*);
struct lmpc_sqp_arrays;
__global__ void lmpc_sqp_linesearch_kernel(lmpc_params, int, lmpc_sqp_arrays, int, double);
__global__ void lmpc_sqp_accumulate_kernel(int, const int*, int*);

struct lmpc_handle {
  lmpc_params P;
  lmpc_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;  // nullptr = the device's default (null) stream
  bool out_aos = false;  // lmpc_set_output_layout: applies to lmpc_solve_batch / lmpc_solve_batch_mixed as the CALLER invokes them; every
                         // solve the library launches for itself (the SQP's QPs, the single-problem host path) writes the default layout
  const int* order = nullptr;  // lmpc_set_launch_order: device [order_n], applied to solves of that batch size only
  int order_n = 0;
  int warm_rounds = 0;    // lmpc_set_warm_rounds (0: by batch size, see launch_solve_warm)
  int warm_resident = 0;  // problems the device holds at once in the warm kernel (CUs x workgroups per CU); 0: no warm kernel for this handle
  double* ws = nullptr;  // [cap][N-1][LMPC_LIN_RECORD]
  int* unverified = nullptr;  // [cap + 1]: the problems a mixed first pass could not verify, and their number
  int* warm_flag = nullptr;   // [cap]: 1 where the last warm solve's attempt was accepted (written by the warm kernels)
  int warm_flag_n = 0;        // the batch of the solve that wrote it (0: no warm solve since)
  void* save = nullptr;       // [save_cap][10 N - 4] elements of save_elem bytes: the polish's save area (grown by reserve_save)
  size_t save_cap = 0, save_elem = 0;
  size_t ws_cap = 0;
  float* ws_f32 = nullptr;  // the same for the single-precision solve
  size_t ws_f32_cap = 0;
  // safe set (device): laps newest-first offsets
  int ss_laps = 0;
  int ss_total = 0;
  int* ss_npts = nullptr;
  int* ss_off = nullptr;
  double* ss_x = nullptr;  // [total][6]
  double ss_L = 0.0;
  int ss_nmax = 0;
  unsigned ss_gen = 0;      // bumped by every lmpc_set_safe_set: the store the codes of lmpc_ss_query_idx_batch point into
  unsigned ss_idx_gen = 0;  // the generation the last lmpc_ss_query_idx_batch ran against (0: none yet)
  int last_precision = LMPC_PRECISION_F64;  // what the last batched solve ran in (lmpc_last_solve_precision)
  int waves = 0;  // lmpc_set_waves_per_problem: 0 the library's choice, 1 / 2 forced (2: fp64 tracking problem, N >= 24, cold solves)
  // regression store (device): lap samples, one-step residuals of the nominal model, end-of-lap flags
  bool reg_on = false;
  int reg_total = 0;
  int* reg_end = nullptr;
  double* reg_x = nullptr;  // [total][6]
  double* reg_u = nullptr;  // [total][2]
  double* reg_y = nullptr;  // [total][6]
  double* reg_tab = nullptr;  // [reg_npad][NF + NOUT]: features and regressed residuals of the samples that have a successor,
                              // and behind them [reg_npad]: the features' squared norms
  int reg_npad = 0;           // their number, padded to a multiple of four with unreachable rows
  lmpc_regression_spec reg_spec{};
  // staging for the single-problem host entry points (lmpc_solve_host, lmpc_ss_query_host): device buffers and PINNED
  // host mirrors, all sized and allocated by lmpc_create -- the per-step path of one controller allocates nothing
  double* stage_dev = nullptr;
  double* stage_host = nullptr;
  size_t stage_doubles = 0;
  int* stage_int = nullptr;
  int* stage_int_host = nullptr;
  double* ssq_dev = nullptr;   // query [2] | ss_x [6][S] | ss_j [S] | j0
  double* ssq_host = nullptr;
  int* ssq_int = nullptr;
  int* ssq_int_host = nullptr;
  // work area of lmpc_solve_full_dynamics_batch (grown by its first call for a batch size, like the workspace)
  double* sqp_ws = nullptr;
  int* sqp_int = nullptr;
  size_t sqp_cap = 0;
  int* sqp_count_host = nullptr;  // pinned
  // timing
  bool timing = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  std::string err;
};

namespace {

// which cold fp64 tracking solves take the two-wave kernels: forced by lmpc_set_waves_per_problem, else by measurement (LMPC_W2_AUTO_KQ:
// the smallest one-wave slot count from which two waves are the default)
#ifndef LMPC_W2_AUTO_KQ
#define LMPC_W2_AUTO_KQ 11  // N >= 41 (with the fused factorisation in both): N = 80 9.70 -> 8.06 ms per 4096, N = 65 6.59 -> 5.32, N = 41 .. 64 -1 .. 4 %; N <= 40 +5 %: one wave (profiles/r06_fuse_ab.txt)
#endif
inline bool lmpc_use_two_waves(int waves, int kq) { return waves == 2 || (waves == 0 && kq >= LMPC_W2_AUTO_KQ); }

int fail(lmpc_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(h, LMPC_ERR_RUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

int kq_for(int N) {
  const int need = (11 * N + 63) / 64;
  const int opts[5] = {2, 4, 7, 11, 14};
  for (int o : opts)
    if (need <= o) return o;
  return -1;
}

int ks_for(int S) { return S <= 0 ? 0 : (S <= 128 ? 2 : (S <= 192 ? 3 : -1)); }

struct solve_args {
  int B;
  size_t lds_bytes;
  const double *x_ic, *u_ic, *T_ref, *bl, *br, *vref, *ss_x, *ss_j;
  double *lam, *X, *U, *dU;
  int *status, *iters;
  double* kkt;
  bool aos;  // results [batch][knot][component]
  const int* ss_idx;  // learning: the safe set by reference (lmpc_solve_batch_ss_idx) instead of ss_x / ss_j
  const double *warm_X, *warm_U;  // lmpc_solve_batch_warm: the plan of the active-set attempt (null: cold)
  const double* warm_lam;         // ... and, learning, the plan's simplex weights [S][B]
};

template <int KQ, int KS, typename real = double>
const void* solve_fn() {
  return reinterpret_cast<const void*>(&lmpc_solve_kernel<real, KQ, KS, double>);
}

// the (KQ, KS) instantiations that exist: KQ = ceil(11 N / 64) rounded up to {2,4,7,11,14}, KS = safe-set points / 64
const void* pick_solve_fn(int kq, int ks) {
  if (ks == 0) {
    switch (kq) {
      case 2: return solve_fn<2, 0>();
      case 4: return solve_fn<4, 0>();
      case 7: return solve_fn<7, 0>();
      case 11: return solve_fn<11, 0>();
      case 14: return solve_fn<14, 0>();
    }
  } else if (ks == 2) {
    if (kq <= 4) return solve_fn<4, 2>();
    if (kq == 7) return solve_fn<7, 2>();
    if (kq == 11) return solve_fn<11, 2>();
    if (kq == 14) return solve_fn<14, 2>();
  } else if (ks == 3) {
    if (kq <= 4) return solve_fn<4, 3>();
    if (kq == 7) return solve_fn<7, 3>();
    if (kq == 11) return solve_fn<11, 3>();
    if (kq == 14) return solve_fn<14, 3>();
  }
  return nullptr;
}

// fp32 interior-point iteration between fp64 arrays (lmpc_solve_batch_mixed): tracking problem
const void* pick_mixed_fn(int kq, int ks) {
  // the learning problem: N <= 23 (BASELINE configs[4]).  N = 40 stays fp64: built for the experiment (round 3, two-pass, the
  // fp32 kernel at one wave per SIMD) it ran 0.80 M solves/s against 0.70 M in fp64 with 13 % of the batch going to the
  // second pass, and one problem of 4096 passed its single-precision KKT test 3.6e-3 away from the fp64 answer -- outside
  // the 1e-3 this entry states, for a 1.14x gain
  // (the N = 40 / 60 instantiations of that experiment, -DLMPC_MIXED_LONG_LEARNING until round 5: scratch/r5/experiment_switches.patch)
  if (ks == 2) return kq <= 4 ? solve_fn<4, 2, float>() : nullptr;
  if (ks == 3) return kq <= 4 ? solve_fn<4, 3, float>() : nullptr;
  switch (kq) {
    case 2:
    case 4: return solve_fn<4, 0, float>();
    case 7: return solve_fn<7, 0, float>();
    case 11: return solve_fn<11, 0, float>();
    case 14: return solve_fn<14, 0, float>();
  }
  return nullptr;
}

// fp32 arrays and iteration (lmpc_solve_batch_f32): the tracking problem
const void* pick_f32_fn(int kq) {
  switch (kq) {
    case 2:
    case 4: return reinterpret_cast<const void*>(&lmpc_solve_kernel<float, 4, 0, float>);
    case 7: return reinterpret_cast<const void*>(&lmpc_solve_kernel<float, 7, 0, float>);
    case 11: return reinterpret_cast<const void*>(&lmpc_solve_kernel<float, 11, 0, float>);
    case 14: return reinterpret_cast<const void*>(&lmpc_solve_kernel<float, 14, 0, float>);
  }
  return nullptr;
}

// the fp64 second pass of a mixed solve, for the (KQ, KS) the mixed kernels exist for
const void* pick_cleanup_fn(int kq, int ks) {
  if (ks == 2) return kq <= 4 ? reinterpret_cast<const void*>(&lmpc_cleanup_kernel<double, 4, 2, double>) : nullptr;
  if (ks == 3) return kq <= 4 ? reinterpret_cast<const void*>(&lmpc_cleanup_kernel<double, 4, 3, double>) : nullptr;
  switch (kq) {
    case 2:
    case 4: return reinterpret_cast<const void*>(&lmpc_cleanup_kernel<double, 4, 0, double>);
    case 7: return reinterpret_cast<const void*>(&lmpc_cleanup_kernel<double, 7, 0, double>);
    case 11: return reinterpret_cast<const void*>(&lmpc_cleanup_kernel<double, 11, 0, double>);
    case 14: return reinterpret_cast<const void*>(&lmpc_cleanup_kernel<double, 14, 0, double>);
  }
  return nullptr;
}

// the safe set by reference: the codes of lmpc_ss_query_idx_batch and the handle's lap store they point into
void set_ss_reference(const lmpc_handle* h, lmpc_params& P, const solve_args& a) {
  P.ss_idx = a.ss_idx;
  P.ss_store = h->ss_x;
  P.ss_npts = h->ss_npts;
  P.ss_off = h->ss_off;
  P.ss_laps = h->ss_laps;
  P.ss_rows = h->ss_total;
  P.ss_L = h->ss_L;
  P.warm_X = a.warm_X;
  P.warm_U = a.warm_U;
  P.warm_lam = a.warm_lam;
  P.warm_flag = nullptr;
}

int launch_cleanup(lmpc_handle* h, const void* fn, const solve_args& a) {
  hipLaunchKernelGGL(lmpc_collect_unverified_kernel, dim3(1), dim3(1024), 0, h->stream, a.B, a.status, h->unverified);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds_bytes));
  lmpc_params P = h->P;
  P.launch_order = nullptr;
  P.flag_unverified = 0;
  P.out_aos = a.aos ? 1 : 0;
  set_ss_reference(h, P, a);
  int B = a.B;
  const double* ws = h->ws;
  const int* list = h->unverified;
  const int* count = h->unverified + a.B;
  void* args[] = {(void*)&P,      (void*)&B,    (void*)&list,  (void*)&count,  (void*)&ws,     (void*)&a.x_ic, (void*)&a.u_ic,
                  (void*)&a.T_ref, (void*)&a.bl, (void*)&a.br,  (void*)&a.vref, (void*)&a.ss_x, (void*)&a.ss_j, (void*)&a.lam,
                  (void*)&a.X,    (void*)&a.U,  (void*)&a.dU,  (void*)&a.status, (void*)&a.iters, (void*)&a.kkt};
#ifdef LMPC_DEBUG_HOOKS  // (measurement builds only, `make debug`: one workgroup per problem)
  static const bool wide = getenv("LMPC_DEBUG_CLEANUP_WIDE") != nullptr;
#else
  const bool wide = false;
#endif
  const int grid = (a.B < 1024 || wide) ? a.B : 1024;  // (a percent of a batch is marked: 1024 workgroups take them in one or two turns)
  HIP_TRY(h, hipLaunchKernel(fn, dim3(grid), dim3(64), args, a.lds_bytes, h->stream));
  return LMPC_OK;
}

// the warm-start kernels (fp64): one per KQ for the tracking problem; the learning problem up to N = 60 (KQ = 11)
const void* pick_warm_fn(int kq, int ks = 0) {
  if (ks == 2) {
    switch (kq) {
      case 2:
      case 4: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<4, 2>);
      case 7: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<7, 2>);
      case 11: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<11, 2>);
    }
    return nullptr;
  }
  if (ks == 3) {
    switch (kq) {
      case 2:
      case 4: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<4, 3>);
      case 7: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<7, 3>);
      case 11: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<11, 3>);
    }
    return nullptr;
  }
  switch (kq) {
    case 2: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<2, 0>);
    case 4: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<4, 0>);
    case 7: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<7, 0>);
    case 11: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<11, 0>);
    case 14: return reinterpret_cast<const void*>(&lmpc_solve_warm_kernel<14, 0>);
  }
  return nullptr;
}

int launch_solve_warm(lmpc_handle* h, const solve_args& a) {
  const void* fn = pick_warm_fn(kq_for(h->P.N), ks_for(h->P.S));
  if (!fn) return fail(h, LMPC_ERR_UNSUPPORTED, "no warm-start kernel for this (N, num_ss_pts)");
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds_bytes));
  lmpc_params P = h->P;
  P.out_aos = a.aos ? 1 : 0;
  set_ss_reference(h, P, a);
  // Repair rounds before the cold start takes over.  A refused attempt is the longest job of its batch (its rounds + a cold solve)
  // and an accepted one the shortest: while the batch fits the device a few times over, its duration is that of the longest jobs
  // and a refusal should come early (2 rounds: 95 % accepted at N = 20, 73 % at N = 60); once the batch is many times what the
  // device holds, throughput counts and every cold solve saved pays (4 rounds: 99 % / 89 %).  Measured in the closed loop
  // (profiles/r05_closed_loop_warm_rounds.txt): N = 20, 4096 cars (2 x resident) 5.50 / 5.36 / 5.17 M car-steps/s at 2 / 3 / 4
  // rounds, 16384 cars (8 x) 10.43 / 10.64 / 10.54; N = 60, 4096 cars (4 x) 1.00 / 1.02 / 1.05.
  P.warm_rounds = h->warm_rounds > 0 ? h->warm_rounds : ((h->warm_resident > 0 && a.B >= 4 * h->warm_resident) ? 4 : 2);
  P.launch_order = (h->order && a.B == h->order_n) ? h->order : nullptr;
  int B = a.B;
  const double* ws = h->ws;
  P.warm_flag = h->warm_flag;  // (reserved with the workspace: at least a.B entries)
  h->warm_flag_n = a.B;
  void* args[] = {(void*)&P, (void*)&B, (void*)&ws, (void*)&a.x_ic, (void*)&a.u_ic, (void*)&a.T_ref, (void*)&a.bl, (void*)&a.br,
                  (void*)&a.vref, (void*)&a.ss_x, (void*)&a.ss_j, (void*)&a.lam, (void*)&a.X, (void*)&a.U, (void*)&a.dU, (void*)&a.status,
                  (void*)&a.iters, (void*)&a.kkt};
  HIP_TRY(h, hipLaunchKernel(fn, dim3(8 * ((a.B + 7) / 8)), dim3(64), args, a.lds_bytes, h->stream));
  return LMPC_OK;
}

// two wavefronts per problem (csrc/lmpc_solve_w2.hip.h): the fp64 tracking problem from N = 24 on
const void* pick_w2_fn(int kq) {
  switch (kq) {
    case 7: return reinterpret_cast<const void*>(&lmpc_solve_kernel_w2<7>);
    case 11: return reinterpret_cast<const void*>(&lmpc_solve_kernel_w2<11>);
    case 14: return reinterpret_cast<const void*>(&lmpc_solve_kernel_w2<14>);
  }
  return nullptr;
}

int launch_solve_w2(lmpc_handle* h, const void* fn, const solve_args& a) {
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds_bytes));
  lmpc_params P = h->P;
  P.out_aos = a.aos ? 1 : 0;
  set_ss_reference(h, P, a);
  P.launch_order = (h->order && a.B == h->order_n) ? h->order : nullptr;
  int B = a.B;
  const double* ws = h->ws;
  void* args[] = {(void*)&P, (void*)&B, (void*)&ws, (void*)&a.x_ic, (void*)&a.u_ic, (void*)&a.T_ref, (void*)&a.bl, (void*)&a.br,
                  (void*)&a.vref, (void*)&a.X, (void*)&a.U, (void*)&a.dU, (void*)&a.status, (void*)&a.iters, (void*)&a.kkt};
  HIP_TRY(h, hipLaunchKernel(fn, dim3(8 * ((a.B + 7) / 8)), dim3(128), args, a.lds_bytes, h->stream));
  return LMPC_OK;
}

// pass: 0 a plain solve; 1 the fp32 iteration of a two-pass mixed solve (marks what it could not verify)
int launch_solve(lmpc_handle* h, const void* fn, const solve_args& a_in, int pass = 0) {
#ifdef LMPC_DEBUG_HOOKS  // LMPC_LDS_PAD (bytes; measurement builds only): over-allocate LDS per problem to cap the problems resident on a CU
  static const int lds_pad = [] { const char* e = getenv("LMPC_LDS_PAD"); return e ? atoi(e) : 0; }();
#else
  const int lds_pad = 0;
#endif
  solve_args a = a_in;
  a.lds_bytes += lds_pad;
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lds_bytes));
  lmpc_params P = h->P;
  P.flag_unverified = pass == 1;
  P.out_aos = a.aos ? 1 : 0;
  set_ss_reference(h, P, a);
  // the registered order applies to solves of exactly its own batch size; every other launch through this handle (the
  // single-problem host path, the SQP's QPs on another batch, ...) keeps the default mapping
  P.launch_order = (h->order && a.B == h->order_n) ? h->order : nullptr;
  int B = a.B;
  const double* ws = h->ws;
  void* args[] = {(void*)&P,        (void*)&B,      (void*)&ws,      (void*)&a.x_ic, (void*)&a.u_ic, (void*)&a.T_ref,
                  (void*)&a.bl,     (void*)&a.br,   (void*)&a.vref,  (void*)&a.ss_x, (void*)&a.ss_j, (void*)&a.lam,
                  (void*)&a.X,      (void*)&a.U,    (void*)&a.dU,    (void*)&a.status, (void*)&a.iters, (void*)&a.kkt};
  HIP_TRY(h, hipLaunchKernel(fn, dim3(8 * ((a.B + 7) / 8)), dim3(64), args, a.lds_bytes, h->stream));  // see the kernel: XCD-aware mapping
  return LMPC_OK;
}

}  // namespace


namespace {
// one lane per (problem, stage); WS: the handle's workspace, otherwise the caller's A/B/g arrays
template <bool WS>
int launch_regress(lmpc_handle* h, int batch, const double* X_ref, const double* U_ref, double* A, double* Bm, double* g) {
  const int nf = h->reg_spec.n_in_state + h->reg_spec.n_in_ctrl;
  const long long queries = (long long)batch * (h->P.N - 1);
  const dim3 grid((unsigned)((queries + 63) / 64)), block(64);
  const double* zz = h->reg_tab + (size_t)h->reg_npad * (size_t)(nf + h->reg_spec.n_out);
  if (nf == 5 && h->reg_spec.n_out == 3)
    hipLaunchKernelGGL((lmpc_regress_kernel<5, 3, WS>), grid, block, 0, h->stream, h->P.N, batch, h->reg_spec, h->reg_npad,
                       h->reg_tab, zz, X_ref, U_ref, A, Bm, g);
  else
    hipLaunchKernelGGL((lmpc_regress_kernel<8, 6, WS>), grid, block, 0, h->stream, h->P.N, batch, h->reg_spec, h->reg_npad,
                       h->reg_tab, zz, X_ref, U_ref, A, Bm, g);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}
}  // namespace

extern "C" {

namespace {
int reserve_sqp(lmpc_handle* h, size_t B);
// the polish's save area for `batch` problems of `elem`-byte values: the single-precision entry needs nothing else of
// lmpc_reserve (no fp64 workspace, no list), and half the bytes
int reserve_save(lmpc_handle* h, size_t batch, size_t elem) {
  if (batch * elem <= h->save_cap * h->save_elem && h->save) return LMPC_OK;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->save) HIP_TRY(h, hipFree(h->save));
  h->save = nullptr;
  h->P.save = nullptr;
  h->save_cap = h->save_elem = 0;
  HIP_TRY(h, hipMalloc(&h->save, batch * (size_t)(10 * h->P.N - 4) * elem));
  h->P.save = h->save;
  h->save_cap = batch;
  h->save_elem = elem;
  return LMPC_OK;
}
}

int lmpc_create(const lmpc_config* cfg, const lmpc_vehicle* veh, int device, lmpc_handle** out) {
  if (!cfg || !veh || !out) return LMPC_ERR_ARGUMENT;
  *out = nullptr;
  lmpc_handle* h = new (std::nothrow) lmpc_handle();
  if (!h) return LMPC_ERR_RUNTIME;
  *out = h;  // returned even on failure so that lmpc_last_error can be read; caller destroys it
  if (veh->model_id != LMPC_MODEL_SINGLE_TRACK_PLANAR)
    return fail(h, LMPC_ERR_UNSUPPORTED, "only single_track_planar_model is built (vehicle_model_factory.cpp:31-49)");
  if (cfg->N < 3 || kq_for(cfg->N) < 0) return fail(h, LMPC_ERR_ARGUMENT, "N must be in [3, 81]");
  if (cfg->learning && (cfg->num_ss_pts < 1 || cfg->num_ss_pts_per_lap < 1))
    return fail(h, LMPC_ERR_ARGUMENT, "learning needs num_ss_pts >= 1 and num_ss_pts_per_lap >= 1");
  if (cfg->learning && ks_for(cfg->num_ss_pts) < 0)
    return fail(h, LMPC_ERR_UNSUPPORTED, "LMPC kernel is built for num_ss_pts <= 192");
  bool hard_hull = false;
  if (cfg->learning) {
    bool any = false;
    for (int k = 0; k < 6; ++k) any = any || cfg->convex_hull_slack[k] > 0.0;
    for (int k = 0; k < 6; ++k)
      if (cfg->convex_hull_slack[k] < 0.0) return fail(h, LMPC_ERR_ARGUMENT, "convex_hull_slack must be non-negative");
    // a zero component is a free slack component (racing_mpc.cpp:497-499: its cost weight is zero); ALL zero turns the
    // hull row into an equality (:500-502): the penalty limit, see LMPC_HARD_HULL_WEIGHT
    hard_hull = !any;
  }
  h->cfg = *cfg;
  h->device = device;
  lmpc_params& P = h->P;
  std::memset(&P, 0, sizeof(P));
  P.N = cfg->N;
  P.has_sigma = cfg->q_boundary > 0.0 ? 1 : 0;
  P.learning = cfg->learning ? 1 : 0;
  P.S = cfg->learning ? cfg->num_ss_pts : 0;
  P.max_iter = cfg->max_iter > 0 ? cfg->max_iter : 60;  // (40 until round 6: one problem of the benched configs[4] share needs 46 -- e_y 0.19 m outside the boundary at x_0 --, the dense oracle solves it: tests/test_gpu_spec_workload.py)
  P.polish = cfg->polish;
  P.tol = cfg->tol > 0.0 ? cfg->tol : 3e-14;
  const double qd[6] = {0.0, cfg->q_contour, cfg->q_heading, cfg->q_vel, cfg->q_vy, cfg->q_vyaw};
  const double qt[6] = {0.0, cfg->q_contour, cfg->q_heading, cfg->q_vel, 0.0, 0.0};
  for (int k = 0; k < 6; ++k) {
    P.Qd[k] = 2.0 * qd[k];
    P.Qt[k] = 20.0 * qt[k];
    P.x_max[k] = cfg->x_max[k];
    P.x_min[k] = cfg->x_min[k];
    P.chs2[k] = hard_hull ? 2.0 * LMPC_HARD_HULL_WEIGHT : 2.0 * cfg->convex_hull_slack[k];
  }
  P.hard_hull = hard_hull ? 1 : 0;
  P.qv_stage = -2.0 * cfg->q_vel;
  P.qv_term = -20.0 * cfg->q_vel;
  for (int a = 0; a < 2; ++a)
    for (int c = 0; c < 2; ++c) {
      P.Qu[a * 2 + c] = cfg->R[a * 2 + c] + cfg->R[c * 2 + a];
      P.Sv[a * 2 + c] = cfg->R_d[a * 2 + c] + cfg->R_d[c * 2 + a];
    }
  P.qsig = 2.0 * cfg->q_boundary;
  P.u_lo[0] = std::fmax(cfg->u_min[0], veh->Fb_max / 1000.0);  // single_track_planar_model.cpp:114
  P.u_hi[0] = std::fmin(cfg->u_max[0], veh->Fd_max / 1000.0);
  P.u_lo[1] = std::fmax(cfg->u_min[1], -veh->max_steer);       // :120
  P.u_hi[1] = std::fmin(cfg->u_max[1], veh->max_steer);
  P.v_lo[0] = veh->Fb_max / 1000.0 / veh->Tb;                  // :146-151
  P.v_hi[0] = veh->Fd_max / 1000.0 / veh->Td;
  P.v_lo[1] = -veh->max_steer_rate;
  P.v_hi[1] = veh->max_steer_rate;
  P.marg = cfg->margin + veh->b / 2.0;                         // racing_mpc.cpp:531
  P.max_vel_ref_diff = cfg->max_vel_ref_diff;
  P.veh = *veh;
  if (!(P.Sv[0] > 0.0) || !(P.Sv[0] * P.Sv[3] - P.Sv[1] * P.Sv[2] > 0.0))
    return fail(h, LMPC_ERR_ARGUMENT, "R_d must be positive definite");
  HIP_TRY(h, hipSetDevice(device));
  for (auto& e : h->ev) HIP_TRY(h, hipEventCreate(&e));
  {  // staging of the single-problem host path, once
    const size_t N = (size_t)P.N, S = (size_t)P.S;
    h->stage_doubles = 8 + 6 * N + 2 * (N - 1) + (N - 1) + 4 * N + 7 * S + 6 * N + 4 * (N - 1) + S + 2 + 6 * N + 2 * (N - 1) + S;  // (+ a warm plan and its simplex weights)
    HIP_TRY(h, hipMalloc(&h->stage_dev, h->stage_doubles * sizeof(double)));
    HIP_TRY(h, hipHostMalloc(&h->stage_host, h->stage_doubles * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->stage_int, 3 * sizeof(int)));
    HIP_TRY(h, hipHostMalloc(&h->stage_int_host, 3 * sizeof(int)));
    if (cfg->num_ss_pts >= 1) {
      const size_t nd = 2 + 7 * (size_t)cfg->num_ss_pts + 1;
      HIP_TRY(h, hipMalloc(&h->ssq_dev, nd * sizeof(double)));
      HIP_TRY(h, hipHostMalloc(&h->ssq_host, nd * sizeof(double)));
      HIP_TRY(h, hipMalloc(&h->ssq_int, sizeof(int)));
      HIP_TRY(h, hipHostMalloc(&h->ssq_int_host, sizeof(int)));
    }
    int rc = lmpc_reserve(h, 1);
    if (rc == LMPC_OK) rc = reserve_sqp(h, 1);
    if (rc != LMPC_OK) return rc;
  }
  {  // what the device holds of the warm kernel at once (the default of lmpc_set_warm_rounds goes by it)
    if (const void* fn = pick_warm_fn(kq_for(P.N), ks_for(P.S))) {
      const size_t lds = lmpc_lds_bytes(P.N, P.learning, P.S, 8);
      int per_cu = 0, cus = 0;
      HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, lds));
      HIP_TRY(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
      h->warm_resident = per_cu * cus;
    }
  }
  return LMPC_OK;
}

void lmpc_destroy(lmpc_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->ws) (void)hipFree(h->ws);
  if (h->unverified) (void)hipFree(h->unverified);
  if (h->warm_flag) (void)hipFree(h->warm_flag);
  if (h->save) (void)hipFree(h->save);
  if (h->ws_f32) (void)hipFree(h->ws_f32);
  if (h->ss_npts) (void)hipFree(h->ss_npts);
  if (h->ss_off) (void)hipFree(h->ss_off);
  if (h->ss_x) (void)hipFree(h->ss_x);
  if (h->reg_end) (void)hipFree(h->reg_end);
  if (h->reg_x) (void)hipFree(h->reg_x);
  if (h->reg_u) (void)hipFree(h->reg_u);
  if (h->reg_y) (void)hipFree(h->reg_y);
  if (h->reg_tab) (void)hipFree(h->reg_tab);
  if (h->stage_dev) (void)hipFree(h->stage_dev);
  if (h->stage_int) (void)hipFree(h->stage_int);
  if (h->sqp_ws) (void)hipFree(h->sqp_ws);
  if (h->sqp_int) (void)hipFree(h->sqp_int);
  if (h->sqp_count_host) (void)hipHostFree(h->sqp_count_host);
  if (h->ssq_dev) (void)hipFree(h->ssq_dev);
  if (h->ssq_int) (void)hipFree(h->ssq_int);
  for (void* q : {(void*)h->stage_host, (void*)h->stage_int_host, (void*)h->ssq_host, (void*)h->ssq_int_host})
    if (q) (void)hipHostFree(q);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  delete h;
}

const char* lmpc_last_error(const lmpc_handle* h) { return h ? h->err.c_str() : "null handle"; }

int lmpc_set_stream(lmpc_handle* h, void* hip_stream) {
  if (!h) return LMPC_ERR_ARGUMENT;
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return LMPC_OK;
}

int lmpc_synchronize(lmpc_handle* h) {
  if (!h) return LMPC_ERR_ARGUMENT;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return LMPC_OK;
}

int lmpc_reserve(lmpc_handle* h, int32_t max_batch) {
  if (!h || max_batch < 0) return LMPC_ERR_ARGUMENT;
  if ((size_t)max_batch <= h->ws_cap) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->ws) HIP_TRY(h, hipFree(h->ws));
  h->ws = nullptr;
  h->ws_cap = 0;
  const size_t bytes = (size_t)max_batch * (h->P.N - 1) * LMPC_LIN_RECORD * sizeof(double);
  HIP_TRY(h, hipMalloc(&h->ws, bytes));
  if (h->unverified) HIP_TRY(h, hipFree(h->unverified));
  h->unverified = nullptr;
  HIP_TRY(h, hipMalloc(&h->unverified, ((size_t)max_batch + 1) * sizeof(int)));
  if (h->warm_flag) HIP_TRY(h, hipFree(h->warm_flag));
  h->warm_flag = nullptr;
  h->warm_flag_n = 0;
  HIP_TRY(h, hipMalloc(&h->warm_flag, (size_t)max_batch * sizeof(int)));
  // the polish's save area first: a batch counts as reserved only when everything a solve of that size touches exists
  // (with ws_cap set before a failed reserve_save a later solve skipped the reservation and wrote through a null save)
  const int rc = reserve_save(h, (size_t)max_batch, sizeof(double));
  if (rc != LMPC_OK) return rc;
  h->ws_cap = (size_t)max_batch;
  return LMPC_OK;
}

int lmpc_query_launch(const lmpc_handle* h, int32_t* lds_bytes_per_problem, int32_t* threads_per_problem) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (lds_bytes_per_problem) *lds_bytes_per_problem = (int32_t)lmpc_lds_bytes(h->P.N, h->P.learning, h->P.S, 8);
  if (threads_per_problem)  // (two wavefronts per problem where the cold fp64 tracking solve takes those kernels: lmpc_set_waves_per_problem)
    *threads_per_problem = (!h->P.learning && lmpc_use_two_waves(h->waves, kq_for(h->P.N)) && kq_for(h->P.N) >= 7) ? 128 : 64;
  return LMPC_OK;
}

int lmpc_query_residency(lmpc_handle* h, int32_t* problems_per_cu) {
  if (!h || !problems_per_cu) return LMPC_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t lds = lmpc_lds_bytes(h->P.N, h->P.learning, h->P.S, 8);
  int n = 0;
  const void* fn = pick_solve_fn(kq_for(h->P.N), ks_for(h->P.S));
  if (!fn) return fail(h, LMPC_ERR_UNSUPPORTED, "no kernel for this (N, num_ss_pts)");
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, lds));
  *problems_per_cu = n;
  return LMPC_OK;
}

int lmpc_query_launch_for(lmpc_handle* h, int32_t precision, int32_t* lds_bytes_per_problem, int32_t* problems_per_cu) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (precision != LMPC_PRECISION_F64 && precision != LMPC_PRECISION_F32 && precision != LMPC_PRECISION_MIXED)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_query_launch_for: unknown precision");
  HIP_TRY(h, hipSetDevice(h->device));
  const int kq = kq_for(h->P.N), ks = ks_for(h->P.S);
  const void* fn = nullptr;
  int threads = 64;
  if (precision == LMPC_PRECISION_F64) fn = pick_solve_fn(kq, ks);
  if (precision == LMPC_PRECISION_F64 && !h->P.learning && lmpc_use_two_waves(h->waves, kq) && pick_w2_fn(kq)) {
    fn = pick_w2_fn(kq);  // (what lmpc_solve_batch launches for this handle: two wavefronts per problem)
    threads = 128;
  }
  if (precision == LMPC_PRECISION_MIXED) fn = pick_mixed_fn(kq, ks);
  if (precision == LMPC_PRECISION_F32 && !h->P.learning) fn = pick_f32_fn(kq);
  if (!fn) return fail(h, LMPC_ERR_UNSUPPORTED, "no kernel for this (N, num_ss_pts) at this precision");
  const size_t lds = lmpc_lds_bytes(h->P.N, h->P.learning, h->P.S, precision == LMPC_PRECISION_F64 ? 8 : 4);
  if (lds_bytes_per_problem) *lds_bytes_per_problem = (int32_t)lds;
  if (problems_per_cu) {
    int n = 0;
    HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds));
    *problems_per_cu = n;
  }
  return LMPC_OK;
}

int lmpc_set_waves_per_problem(lmpc_handle* h, int32_t waves) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (waves < 0 || waves > 2) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_waves_per_problem: 0 (the library's choice), 1 or 2");
  h->waves = waves;
  return LMPC_OK;
}

int lmpc_last_solve_precision(const lmpc_handle* h, int32_t* precision) {
  if (!h || !precision) return LMPC_ERR_ARGUMENT;
  *precision = h->last_precision;
  return LMPC_OK;
}

int lmpc_enable_timing(lmpc_handle* h, int32_t on) {
  if (!h) return LMPC_ERR_ARGUMENT;
  h->timing = on != 0;
  return LMPC_OK;
}

int lmpc_last_kernel_ms(lmpc_handle* h, float* linearize_ms, float* solve_ms) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (!h->timing) return fail(h, LMPC_ERR_ARGUMENT, "timing not enabled");
  HIP_TRY(h, hipEventSynchronize(h->ev[2]));
  float a = 0.f, b = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
  HIP_TRY(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
  if (linearize_ms) *linearize_ms = a;
  if (solve_ms) *solve_ms = b;
  return LMPC_OK;
}

int lmpc_linearize_batch(lmpc_handle* h, int32_t batch, const double* X_ref, const double* U_ref, const double* T_ref,
                         const double* curvatures, double* A, double* Bm, double* g) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !X_ref || !U_ref || !T_ref || !curvatures || !A || !Bm || !g)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_linearize_batch: null pointer or negative batch");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  dim3 grid((batch + 255) / 256, h->P.N - 1);
  hipLaunchKernelGGL((lmpc_linearize_kernel<false, double, 1>), grid, dim3(256), 0, h->stream, h->P, batch, X_ref, U_ref, T_ref,
                     curvatures, A, Bm, g);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}

namespace {
int solve_batch_fp64_arrays(lmpc_handle* h, bool mixed, bool aos, int32_t batch, const double* x_ic, const double* u_ic,
                            const double* X_ref, const double* U_ref, const double* T_ref, const double* bound_left,
                            const double* bound_right, const double* curvatures, const double* vel_ref,
                            double total_length, const double* ss_x, const double* ss_j, double* X_optm, double* U_optm,
                            double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt,
                            const int32_t* ss_idx = nullptr, const double* warm_X = nullptr, const double* warm_U = nullptr,
                            const double* warm_lam = nullptr) {
  if (!h) return LMPC_ERR_ARGUMENT;
  (void)total_length;  // abscissa alignment (racing_mpc.cpp:219-223) shifts s only; the QP is invariant to it
  if (batch < 0 || !x_ic || !u_ic || !X_ref || !U_ref || !T_ref || !bound_left || !bound_right || !curvatures ||
      !vel_ref || !X_optm || !U_optm || !dU_optm || !status || !iters)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch: null pointer or negative batch");
  // A configuration the mixed entry has no kernel for (the learning problem at N >= 24, the hard hull equality) is solved in fp64
  // (round 6, VERDICT r5 item 7 / weak 10: it was LMPC_ERR_UNSUPPORTED, and a caller iterating over horizons had to special-case a
  // valid reference configuration); lmpc_last_solve_precision tells the caller which it was.
  if (mixed && (h->P.hard_hull || !pick_mixed_fn(kq_for(h->P.N), ks_for(h->P.S)))) mixed = false;
  h->last_precision = mixed ? LMPC_PRECISION_MIXED : LMPC_PRECISION_F64;
  if (h->P.learning && !ss_idx && (!ss_x || !ss_j)) return fail(h, LMPC_ERR_ARGUMENT, "learning=1 needs ss_x and ss_j");
  if (ss_idx && (!h->P.learning || !h->ss_x || h->ss_laps < 1))
    return fail(h, LMPC_ERR_ARGUMENT, "ss_idx needs learning=1 and a safe set stored on the handle (lmpc_set_safe_set)");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  if ((size_t)batch > h->ws_cap) {
    const int rc = lmpc_reserve(h, batch);
    if (rc != LMPC_OK) return rc;
  }
  const int N = h->P.N;
  if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
  dim3 grid((batch + 255) / 256, N - 1);
  // the 256-register build of the linearisation where a wave of it can sit next to a resident wave of the QP kernel that follows
  // (two waves per SIMD: the short-horizon tracking kernels), so that the next batch's linearisation fills this batch's residency tail
  if (!h->P.learning && lmpc_waves_per_simd(mixed ? 4 : 8, kq_for(N), 0) >= 2)
    hipLaunchKernelGGL((lmpc_linearize_kernel<true, double, 2>), grid, dim3(256), 0, h->stream, h->P, batch, X_ref, U_ref, T_ref,
                       curvatures, h->ws, (double*)nullptr, (double*)nullptr);
  else
    hipLaunchKernelGGL((lmpc_linearize_kernel<true, double, 1>), grid, dim3(256), 0, h->stream, h->P, batch, X_ref, U_ref, T_ref,
                       curvatures, h->ws, (double*)nullptr, (double*)nullptr);
  HIP_TRY(h, hipGetLastError());
  if (h->reg_on) {  // error-dynamics regression onto the workspace (safe_set.cpp:182-245)
    const int rc = launch_regress<true>(h, batch, X_ref, U_ref, h->ws, nullptr, nullptr);
    if (rc != LMPC_OK) return rc;
  }
  if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
  const void* fn = mixed ? pick_mixed_fn(kq_for(N), ks_for(h->P.S)) : pick_solve_fn(kq_for(N), ks_for(h->P.S));
  if (!fn) return fail(h, LMPC_ERR_UNSUPPORTED, "no kernel for this (N, num_ss_pts)");
  solve_args a{};
  a.B = batch;
  a.lds_bytes = lmpc_lds_bytes(N, h->P.learning, h->P.S, mixed ? 4 : 8);
  a.x_ic = x_ic; a.u_ic = u_ic; a.T_ref = T_ref; a.bl = bound_left; a.br = bound_right; a.vref = vel_ref;
  a.ss_x = (h->P.learning && !ss_idx) ? ss_x : nullptr; a.ss_j = (h->P.learning && !ss_idx) ? ss_j : nullptr;
  a.ss_idx = h->P.learning ? ss_idx : nullptr;
  // the warm start is built into fp64 kernels of its own -- the tracking problem at every horizon, the learning problem (with the
  // plan's simplex weights) up to N = 60; everywhere else the call is the cold solve it would fall back to
  const bool warm_built = !mixed && (!h->P.learning || warm_lam) && pick_warm_fn(kq_for(N), ks_for(h->P.S)) != nullptr;
  a.warm_X = warm_built ? warm_X : nullptr;
  a.warm_U = warm_built ? warm_U : nullptr;
  a.warm_lam = (warm_built && h->P.learning) ? warm_lam : nullptr;
  h->warm_flag_n = 0;  // (a cold solve: nothing was attempted)
  a.lam = h->P.learning ? convex_combi_optm : nullptr;
  a.X = X_optm; a.U = U_optm; a.dU = dU_optm; a.status = status; a.iters = iters; a.kkt = kkt;
  a.aos = aos;
  // Mixed precision is two launches when the polish is on: the fp32 iteration verifies its own answers (polish accepted =
  // KKT test passed) and marks the problems it could not verify -- a percent of a batch: active sets still ambiguous at
  // mu = 2e-6, or more than four free simplex weights -- and the fp64 kernel behind it solves exactly those.
  const bool two_pass = mixed && h->P.polish == 0;  // (polish = 1: the marks stay visible, no second pass -- diagnostics)
#ifdef LMPC_DEBUG_HOOKS  // LMPC_DEBUG_CLEANUP_ALL (the debug build, tests/second_pass_check.py): skip the fp32 pass and hand the
                         // whole batch to the fp64 second pass
  static const bool cleanup_all = getenv("LMPC_DEBUG_CLEANUP_ALL") != nullptr;
#else
  const bool cleanup_all = false;
#endif
  int rc = LMPC_OK;
  if (two_pass && cleanup_all)
    HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)status, LMPC_SOLVE_UNVERIFIED, (size_t)batch, h->stream));
  else if (a.warm_X && a.warm_U)
    rc = launch_solve_warm(h, a);
  else if (!mixed && !h->P.learning && lmpc_use_two_waves(h->waves, kq_for(N)) && pick_w2_fn(kq_for(N)))
    rc = launch_solve_w2(h, pick_w2_fn(kq_for(N)), a);
  else
    rc = launch_solve(h, fn, a, (mixed && h->P.polish >= 0) ? 1 : 0);
  if (rc != LMPC_OK) return rc;
  if (two_pass) {
    const void* fn64 = pick_cleanup_fn(kq_for(N), ks_for(h->P.S));
    if (!fn64) return fail(h, LMPC_ERR_UNSUPPORTED, "no fp64 second pass for this (N, num_ss_pts)");
    a.lds_bytes = lmpc_lds_bytes(N, h->P.learning, h->P.S, 8);
    rc = launch_cleanup(h, fn64, a);
    if (rc != LMPC_OK) return rc;
  }
  if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], h->stream));
  return LMPC_OK;
}
}  // namespace

int lmpc_solve_batch(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref,
                     const double* U_ref, const double* T_ref, const double* bound_left, const double* bound_right,
                     const double* curvatures, const double* vel_ref, double total_length, const double* ss_x,
                     const double* ss_j, double* X_optm, double* U_optm, double* dU_optm, double* convex_combi_optm,
                     int32_t* status, int32_t* iters, double* kkt) {
  return solve_batch_fp64_arrays(h, false, h && h->out_aos, batch, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures,
                                 vel_ref, total_length, ss_x, ss_j, X_optm, U_optm, dU_optm, convex_combi_optm, status,
                                 iters, kkt);
}

int lmpc_solve_batch_mixed(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref,
                           const double* U_ref, const double* T_ref, const double* bound_left, const double* bound_right,
                           const double* curvatures, const double* vel_ref, double total_length, const double* ss_x,
                           const double* ss_j, double* X_optm, double* U_optm, double* dU_optm,
                           double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt) {
  return solve_batch_fp64_arrays(h, true, h && h->out_aos, batch, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures,
                                 vel_ref, total_length, ss_x, ss_j, X_optm, U_optm, dU_optm, convex_combi_optm, status,
                                 iters, kkt);
}

int lmpc_solve_batch_warm(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                          const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                          const double* vel_ref, double total_length, const double* X_optm_ref, const double* U_optm_ref, double* X_optm,
                          double* U_optm, double* dU_optm, int32_t* status, int32_t* iters, double* kkt) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (!X_optm_ref || !U_optm_ref) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_warm: X_optm_ref / U_optm_ref is null");
  if (h->P.learning) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc_solve_batch_warm: the tracking problem only (learning handles: lmpc_solve_batch_warm_ss)");
  return solve_batch_fp64_arrays(h, false, h->out_aos, batch, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref,
                                 total_length, nullptr, nullptr, X_optm, U_optm, dU_optm, nullptr, status, iters, kkt, nullptr, X_optm_ref,
                                 U_optm_ref);
}

int lmpc_solve_batch_warm_ss(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                             const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                             const double* vel_ref, double total_length, const double* ss_x, const double* ss_j, const int32_t* ss_idx,
                             const double* X_optm_ref, const double* U_optm_ref, const double* convex_combi_optm_ref, double* X_optm,
                             double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (!X_optm_ref || !U_optm_ref) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_warm_ss: X_optm_ref / U_optm_ref is null");
  if (!h->P.learning) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_warm_ss: a learning handle (the tracking problem: lmpc_solve_batch_warm)");
  if (ss_idx && h->ss_idx_gen != h->ss_gen)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_warm_ss: the safe set was replaced (lmpc_set_safe_set) after the lmpc_ss_query_idx_batch these "
                                      "codes come from, or no such query ran on this handle");
  return solve_batch_fp64_arrays(h, false, h->out_aos, batch, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref,
                                 total_length, ss_x, ss_j, X_optm, U_optm, dU_optm, convex_combi_optm, status, iters, kkt, ss_idx, X_optm_ref,
                                 U_optm_ref, convex_combi_optm_ref);
}

int lmpc_get_warm_accepted(lmpc_handle* h, int32_t batch, int32_t* accepted) {
  if (!h || !accepted || batch < 0) return h ? fail(h, LMPC_ERR_ARGUMENT, "lmpc_get_warm_accepted: null pointer or negative batch") : LMPC_ERR_ARGUMENT;
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  if (h->warm_flag_n != batch) {  // the last solve of this size was not a warm one (or the kernel for it is not built): nothing was attempted
    HIP_TRY(h, hipMemsetAsync(accepted, 0, (size_t)batch * sizeof(int32_t), h->stream));
    return LMPC_OK;
  }
  HIP_TRY(h, hipMemcpyAsync(accepted, h->warm_flag, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  return LMPC_OK;
}

int lmpc_shift_lambda_batch(lmpc_handle* h, int32_t batch, const int32_t* ss_idx_prev, const double* lambda_prev, const int32_t* ss_idx,
                            int32_t advance, double* lambda_ref) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !ss_idx_prev || !lambda_prev || !ss_idx || !lambda_ref || advance < 0)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_shift_lambda_batch: null pointer, negative batch or negative advance");
  if (!h->P.learning || h->ss_laps < 1) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_shift_lambda_batch: a learning handle with a safe set stored");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(lmpc_shift_lambda_kernel, dim3((batch + 63) / 64), dim3(64), 0, h->stream, batch, h->P.S, h->ss_laps, h->ss_npts, h->ss_off,
                     ss_idx_prev, lambda_prev, ss_idx, advance, lambda_ref);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}

int lmpc_solve_batch_ss_idx(lmpc_handle* h, int32_t batch, int32_t precision, const double* x_ic, const double* u_ic, const double* X_ref,
                            const double* U_ref, const double* T_ref, const double* bound_left, const double* bound_right,
                            const double* curvatures, const double* vel_ref, double total_length, const int32_t* ss_idx, double* X_optm,
                            double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters, double* kkt) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (!ss_idx) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_ss_idx: ss_idx is null");
  if (precision != LMPC_PRECISION_F64 && precision != LMPC_PRECISION_MIXED)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_ss_idx: precision is LMPC_PRECISION_F64 or LMPC_PRECISION_MIXED");
  // the codes name rows of the store the query ran against: a store replaced since -- possibly by a smaller one -- would
  // be read out of bounds, or, in bounds, silently solve on other points with status OPTIMAL
  if (h->ss_idx_gen != h->ss_gen)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_ss_idx: the safe set was replaced (lmpc_set_safe_set) after the lmpc_ss_query_idx_batch these "
                                      "codes come from, or no such query ran on this handle");
  return solve_batch_fp64_arrays(h, precision == LMPC_PRECISION_MIXED, h->out_aos, batch, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right,
                                 curvatures, vel_ref, total_length, nullptr, nullptr, X_optm, U_optm, dU_optm, convex_combi_optm, status, iters, kkt,
                                 ss_idx);
}

int lmpc_solve_batch_f32(lmpc_handle* h, int32_t batch, const float* x_ic, const float* u_ic, const float* X_ref,
                         const float* U_ref, const float* T_ref, const float* bound_left, const float* bound_right,
                         const float* curvatures, const float* vel_ref, float* X_optm, float* U_optm, float* dU_optm,
                         int32_t* status, int32_t* iters, float* kkt) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !x_ic || !u_ic || !X_ref || !U_ref || !T_ref || !bound_left || !bound_right || !curvatures ||
      !vel_ref || !X_optm || !U_optm || !dU_optm || !status || !iters)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_batch_f32: null pointer or negative batch");
  if (h->P.learning) return fail(h, LMPC_ERR_UNSUPPORTED, "single precision is built for the tracking problem only");
  if (h->reg_on) return fail(h, LMPC_ERR_UNSUPPORTED, "the error-dynamics regression is applied in double precision only");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  const int N = h->P.N;
  const void* fn = pick_f32_fn(kq_for(N));
  if (!fn) return fail(h, LMPC_ERR_UNSUPPORTED, "no single-precision kernel for this N");
  h->last_precision = LMPC_PRECISION_F32;
  {
    const int rc = reserve_save(h, (size_t)batch, sizeof(float));
    if (rc != LMPC_OK) return rc;
  }
  if ((size_t)batch > h->ws_f32_cap) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->ws_f32) HIP_TRY(h, hipFree(h->ws_f32));
    h->ws_f32 = nullptr;
    h->ws_f32_cap = 0;
    HIP_TRY(h, hipMalloc(&h->ws_f32, (size_t)batch * (N - 1) * LMPC_LIN_RECORD * sizeof(float)));
    h->ws_f32_cap = (size_t)batch;
  }
  if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
  dim3 grid((batch + 255) / 256, N - 1);
  if (lmpc_waves_per_simd(4, kq_for(N), 0) >= 2)
    hipLaunchKernelGGL((lmpc_linearize_kernel<true, float, 2>), grid, dim3(256), 0, h->stream, h->P, batch, X_ref, U_ref, T_ref,
                       curvatures, h->ws_f32, (float*)nullptr, (float*)nullptr);
  else
    hipLaunchKernelGGL((lmpc_linearize_kernel<true, float, 1>), grid, dim3(256), 0, h->stream, h->P, batch, X_ref, U_ref, T_ref,
                       curvatures, h->ws_f32, (float*)nullptr, (float*)nullptr);
  HIP_TRY(h, hipGetLastError());
  if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
  const size_t lds = lmpc_lds_bytes(N, 0, 0, 4);
  HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  lmpc_params P = h->P;
  P.launch_order = (h->order && batch == h->order_n) ? h->order : nullptr;
  int B = batch;
  const float *ws = h->ws_f32, *nul = nullptr;
  float* nulo = nullptr;
  void* args[] = {&P, &B, &ws, &x_ic, &u_ic, &T_ref, &bound_left, &bound_right, &vel_ref, &nul, &nul, &nulo,
                  &X_optm, &U_optm, &dU_optm, &status, &iters, &kkt};
  HIP_TRY(h, hipLaunchKernel(fn, dim3(8 * ((batch + 7) / 8)), dim3(64), args, lds, h->stream));
  if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], h->stream));
  return LMPC_OK;
}

namespace {
// work area of the sequential-QP solve for up to B problems: QP solution + penalty weights, per-problem ints, a counter
int reserve_sqp(lmpc_handle* h, size_t B) {
  if (B <= h->sqp_cap) return LMPC_OK;
  const size_t N = (size_t)h->P.N, S = (size_t)h->P.S;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->sqp_ws) HIP_TRY(h, hipFree(h->sqp_ws));
  if (h->sqp_int) HIP_TRY(h, hipFree(h->sqp_int));
  h->sqp_ws = nullptr;
  h->sqp_int = nullptr;
  h->sqp_cap = 0;
  HIP_TRY(h, hipMalloc(&h->sqp_ws, (2 * (6 * N + 4 * (N - 1) + S) + 1) * B * sizeof(double)));
  HIP_TRY(h, hipMalloc(&h->sqp_int, (4 * B + 1) * sizeof(int)));
  if (!h->sqp_count_host) HIP_TRY(h, hipHostMalloc(&h->sqp_count_host, sizeof(int)));
  h->sqp_cap = B;
  return LMPC_OK;
}
}  // namespace

int lmpc_solve_full_dynamics_batch(lmpc_handle* h, int32_t batch, const double* x_ic, const double* u_ic, const double* X_ref,
                                   const double* U_ref, const double* T_ref, const double* bound_left,
                                   const double* bound_right, const double* curvatures, const double* vel_ref,
                                   double total_length, const double* ss_x, const double* ss_j, int32_t max_sqp,
                                   double step_tol, double* X_optm, double* U_optm, double* dU_optm,
                                   double* convex_combi_optm, int32_t* status, int32_t* iters, int32_t* sqp_iters,
                                   double* sqp_move, double* defect) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !X_ref || !U_ref || !X_optm || !U_optm || !dU_optm || !status || !iters || !sqp_iters || !sqp_move || !defect)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_full_dynamics_batch: null pointer or negative batch");
  if (max_sqp < 1) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_full_dynamics_batch: max_sqp < 1");
  if (h->P.learning && !convex_combi_optm) return fail(h, LMPC_ERR_ARGUMENT, "learning=1 needs convex_combi_optm");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t B = (size_t)batch, N = (size_t)h->P.N, NS = N - 1, S = (size_t)h->P.S;
  const size_t nX = 6 * N * B, nU = 2 * NS * B, nL = S * B;
  // work area: QP solution (Xq, Uq, dUq, lamq), the iterate before the last step (Xp, Up, dUp, lamp), penalty weights;
  // ints: status_q, iters_q, active, consecutive back-offs, counter
  {
    const int rc = reserve_sqp(h, B);
    if (rc != LMPC_OK) return rc;
  }
  double* Xq = h->sqp_ws;
  double* Uq = Xq + nX;
  double* dUq = Uq + nU;
  double* lamq = dUq + nU;
  double* Xp = lamq + nL;
  double* Up = Xp + nX;
  double* dUp = Up + nU;
  double* lamp = dUp + nU;
  double* nu = lamp + nL;
  int* status_q = h->sqp_int;
  int* iters_q = status_q + B;
  int* active = iters_q + B;
  int* backoffs = active + B;
  int* counter = backoffs + B;
  // the iterate lives in the caller's output arrays: start = the reference trajectory (the node's zero-input rollout,
  // racing_mpc_node.cpp:210-235), dU = 0, lambda = 0 (its convex_combi_optm_ref)
  HIP_TRY(h, hipMemcpyAsync(X_optm, X_ref, nX * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(U_optm, U_ref, nU * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(dU_optm, 0, nU * sizeof(double), h->stream));
  if (S) HIP_TRY(h, hipMemsetAsync(convex_combi_optm, 0, nL * sizeof(double), h->stream));
  HIP_TRY(h, hipMemsetAsync(nu, 0, B * sizeof(double), h->stream));
  HIP_TRY(h, hipMemsetAsync(backoffs, 0, B * sizeof(int), h->stream));
  HIP_TRY(h, hipMemsetAsync(iters, 0, B * sizeof(int), h->stream));
  HIP_TRY(h, hipMemsetAsync(sqp_iters, 0, B * sizeof(int), h->stream));
  HIP_TRY(h, hipMemsetAsync(defect, 0, B * sizeof(double), h->stream));
  {  // active = 1, move = inf
    std::vector<int> ones(B, 1);
    std::vector<double> inf(B, INFINITY);
    HIP_TRY(h, hipMemcpyAsync(active, ones.data(), B * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(sqp_move, inf.data(), B * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  lmpc_sqp_arrays A{};
  A.X = X_optm; A.U = U_optm; A.dU = dU_optm; A.lam = S ? convex_combi_optm : nullptr;
  A.Xq = Xq; A.Uq = Uq; A.dUq = dUq; A.lamq = S ? lamq : nullptr;
  A.Xp = Xp; A.Up = Up; A.dUp = dUp; A.lamp = S ? lamp : nullptr;
  A.backoffs = backoffs; A.iters_q = iters_q; A.iters = iters;
  A.status_q = status_q;
  A.T_ref = T_ref; A.curv = curvatures; A.bl = bound_left; A.br = bound_right; A.vref = vel_ref; A.ss_x = ss_x; A.ss_j = ss_j;
  A.nu = nu; A.active = active; A.status = status; A.sqp_iters = sqp_iters; A.move = sqp_move; A.defect = defect;
  A.n_active = counter;
  for (int it = 0; it < max_sqp; ++it) {
    // QP about the iterate (racing_mpc.cpp:169-186 with X_ref, U_ref := the iterate)
    // (the default layout whatever lmpc_set_output_layout says: the line search and the next linearisation index it)
    const int rc = solve_batch_fp64_arrays(h, false, false, batch, x_ic, u_ic, X_optm, U_optm, T_ref, bound_left, bound_right, curvatures,
                                           vel_ref, total_length, ss_x, ss_j, Xq, Uq, dUq, S ? lamq : nullptr, status_q, iters_q, nullptr);
    if (rc != LMPC_OK) return rc;
    HIP_TRY(h, hipMemsetAsync(counter, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(lmpc_sqp_linesearch_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, h->P, batch, A,
                       it == 0 ? 1 : 0, step_tol);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->sqp_count_host, counter, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (*h->sqp_count_host == 0) break;
  }
  return LMPC_OK;
}

namespace {
int solve_host_impl(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                    const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                    const double* vel_ref, double total_length, const double* ss_x, const double* ss_j, double* X_optm,
                    double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters,
                    int max_sqp, double step_tol, int32_t* sqp_iters, double* sqp_move, double* defect,
                    const double* X_warm = nullptr, const double* U_warm = nullptr, const double* lam_warm = nullptr) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (!x_ic || !u_ic || !X_ref || !U_ref || !T_ref || !bound_left || !bound_right || !curvatures || !vel_ref ||
      !X_optm || !U_optm || !dU_optm || !status || !iters)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_host: null pointer");
  const int N = h->P.N, NS = N - 1, S = h->P.S;
  // staging layout (doubles): inputs then outputs, each in the batch = 1 device layout
  const size_t o_x = 0, o_u = 6, o_X = 8, o_U = o_X + 6 * N, o_T = o_U + 2 * NS, o_bl = o_T + NS, o_br = o_bl + N,
               o_k = o_br + N, o_v = o_k + N, o_sx = o_v + N, o_sj = o_sx + 6 * (size_t)S, o_Xo = o_sj + S,
               o_Uo = o_Xo + 6 * N, o_dUo = o_Uo + 2 * NS, o_lam = o_dUo + 2 * NS, o_mv = o_lam + S, o_wX = o_mv + 2, o_wU = o_wX + 6 * N,
               o_wL = o_wU + 2 * NS, total = o_wL + S;
  HIP_TRY(h, hipSetDevice(h->device));
  if (total != h->stage_doubles || !h->stage_dev || !h->stage_host) return fail(h, LMPC_ERR_RUNTIME, "lmpc_solve_host: staging not allocated");
  double* const host = h->stage_host;  // pinned: the two copies below are asynchronous DMA transfers
  for (size_t e = o_sx; e < o_Xo; ++e) host[e] = 0.0;
  for (int k = 0; k < 6; ++k) host[o_x + k] = x_ic[k];
  for (int k = 0; k < 2; ++k) host[o_u + k] = u_ic[k];
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) host[o_X + (size_t)k * N + i] = X_ref[(size_t)i * 6 + k];  // column-major -> [6][N]
  for (int i = 0; i < NS; ++i) {
    for (int k = 0; k < 2; ++k) host[o_U + (size_t)k * NS + i] = U_ref[(size_t)i * 2 + k];
    host[o_T + i] = T_ref[i];
  }
  for (int i = 0; i < N; ++i) {
    host[o_bl + i] = bound_left[i];
    host[o_br + i] = bound_right[i];
    host[o_k + i] = curvatures[i];
    host[o_v + i] = vel_ref[i];
  }
  if (S && ss_x && ss_j)
    for (int j = 0; j < S; ++j) {
      for (int k = 0; k < 6; ++k) host[o_sx + (size_t)k * S + j] = ss_x[(size_t)j * 6 + k];
      host[o_sj + j] = ss_j[j];
    }
  double* d = h->stage_dev;
  HIP_TRY(h, hipMemcpyAsync(d, host, o_Xo * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const bool warm = X_warm && U_warm && max_sqp <= 0 && (!h->P.learning || lam_warm);
  if (warm) {  // the plan, column-major -> [6][N] / [2][N-1] (and, learning, its simplex weights)
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 6; ++k) host[o_wX + (size_t)k * N + i] = X_warm[(size_t)i * 6 + k];
    for (int i = 0; i < NS; ++i)
      for (int k = 0; k < 2; ++k) host[o_wU + (size_t)k * NS + i] = U_warm[(size_t)i * 2 + k];
    if (S && lam_warm)
      for (int j = 0; j < S; ++j) host[o_wL + j] = lam_warm[j];
    HIP_TRY(h, hipMemcpyAsync(d + o_wX, host + o_wX, (total - o_wX) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  const int rc = max_sqp > 0
      ? lmpc_solve_full_dynamics_batch(h, 1, d + o_x, d + o_u, d + o_X, d + o_U, d + o_T, d + o_bl, d + o_br, d + o_k, d + o_v,
                                       total_length, S ? d + o_sx : nullptr, S ? d + o_sj : nullptr, max_sqp, step_tol,
                                       d + o_Xo, d + o_Uo, d + o_dUo, S ? d + o_lam : nullptr, h->stage_int,
                                       h->stage_int + 1, h->stage_int + 2, d + o_mv, d + o_mv + 1)
      : solve_batch_fp64_arrays(h, false, false, 1, d + o_x, d + o_u, d + o_X, d + o_U, d + o_T, d + o_bl, d + o_br, d + o_k, d + o_v,
                                total_length, S ? d + o_sx : nullptr, S ? d + o_sj : nullptr, d + o_Xo, d + o_Uo,
                                d + o_dUo, S ? d + o_lam : nullptr, h->stage_int, h->stage_int + 1, nullptr,  // (the staging buffer is unpacked as [6][N])
                                nullptr, warm ? d + o_wX : nullptr, warm ? d + o_wU : nullptr, (warm && S) ? d + o_wL : nullptr);
  if (rc != LMPC_OK) return rc;
  int* const si = h->stage_int_host;
  HIP_TRY(h, hipMemcpyAsync(host + o_Xo, d + o_Xo, (o_wX - o_Xo) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(si, h->stage_int, 3 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) X_optm[(size_t)i * 6 + k] = host[o_Xo + (size_t)k * N + i];
  for (int i = 0; i < NS; ++i)
    for (int k = 0; k < 2; ++k) {
      U_optm[(size_t)i * 2 + k] = host[o_Uo + (size_t)k * NS + i];
      dU_optm[(size_t)i * 2 + k] = host[o_dUo + (size_t)k * NS + i];
    }
  if (S && convex_combi_optm)
    for (int j = 0; j < S; ++j) convex_combi_optm[j] = host[o_lam + j];
  *status = si[0];
  *iters = si[1];
  if (sqp_iters) *sqp_iters = max_sqp > 0 ? si[2] : 1;
  if (sqp_move) *sqp_move = max_sqp > 0 ? host[o_mv] : 0.0;
  if (defect) *defect = max_sqp > 0 ? host[o_mv + 1] : 0.0;
  return LMPC_OK;
}
}  // namespace

int lmpc_solve_host(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                    const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                    const double* vel_ref, double total_length, const double* ss_x, const double* ss_j, double* X_optm,
                    double* U_optm, double* dU_optm, double* convex_combi_optm, int32_t* status, int32_t* iters) {
  return solve_host_impl(h, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, total_length, ss_x,
                         ss_j, X_optm, U_optm, dU_optm, convex_combi_optm, status, iters, 0, 0.0, nullptr, nullptr, nullptr);
}

int lmpc_solve_host_warm(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                         const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                         const double* vel_ref, double total_length, const double* X_optm_ref, const double* U_optm_ref, double* X_optm,
                         double* U_optm, double* dU_optm, int32_t* status, int32_t* iters) {
  if (h && (!X_optm_ref || !U_optm_ref)) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_host_warm: X_optm_ref / U_optm_ref is null");
  if (h && h->P.learning) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc_solve_host_warm: the tracking problem only (learning handles: lmpc_solve_host_warm_ss)");
  return solve_host_impl(h, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, total_length, nullptr, nullptr,
                         X_optm, U_optm, dU_optm, nullptr, status, iters, 0, 0.0, nullptr, nullptr, nullptr, X_optm_ref, U_optm_ref);
}

int lmpc_solve_host_warm_ss(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref, const double* U_ref,
                            const double* T_ref, const double* bound_left, const double* bound_right, const double* curvatures,
                            const double* vel_ref, double total_length, const double* ss_x, const double* ss_j, const double* X_optm_ref,
                            const double* U_optm_ref, const double* convex_combi_optm_ref, double* X_optm, double* U_optm, double* dU_optm,
                            double* convex_combi_optm, int32_t* status, int32_t* iters) {
  if (h && (!X_optm_ref || !U_optm_ref || !convex_combi_optm_ref))
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_host_warm_ss: X_optm_ref / U_optm_ref / convex_combi_optm_ref is null");
  if (h && !h->P.learning) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_host_warm_ss: a learning handle (the tracking problem: lmpc_solve_host_warm)");
  return solve_host_impl(h, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, total_length, ss_x, ss_j,
                         X_optm, U_optm, dU_optm, convex_combi_optm, status, iters, 0, 0.0, nullptr, nullptr, nullptr, X_optm_ref, U_optm_ref,
                         convex_combi_optm_ref);
}

int lmpc_solve_full_dynamics_host(lmpc_handle* h, const double* x_ic, const double* u_ic, const double* X_ref,
                                  const double* U_ref, const double* T_ref, const double* bound_left,
                                  const double* bound_right, const double* curvatures, const double* vel_ref,
                                  double total_length, const double* ss_x, const double* ss_j, int32_t max_sqp,
                                  double step_tol, double* X_optm, double* U_optm, double* dU_optm,
                                  double* convex_combi_optm, int32_t* status, int32_t* iters, int32_t* sqp_iters,
                                  double* sqp_move, double* defect) {
  if (max_sqp < 1) return h ? fail(h, LMPC_ERR_ARGUMENT, "lmpc_solve_full_dynamics_host: max_sqp < 1") : LMPC_ERR_ARGUMENT;
  return solve_host_impl(h, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, total_length, ss_x,
                         ss_j, X_optm, U_optm, dU_optm, convex_combi_optm, status, iters, max_sqp, step_tol, sqp_iters,
                         sqp_move, defect);
}

namespace {
int prepare_impl(lmpc_handle* h, const char* who, int32_t batch, const lmpc_track* track, const double* x_ic,
                 const int32_t* status, double dt, double speed_scale, double speed_limit, double* X_ref, double* U_ref,
                 double* T_ref, double* bound_left, double* bound_right, double* curvatures, double* vel_ref) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !track || !x_ic || !X_ref || !U_ref || !T_ref || !bound_left || !bound_right || !curvatures ||
      !vel_ref || !track->curvature || !track->bound_left || !track->bound_right || !track->vel || track->M < 2 ||
      !(track->L > 0.0) || !(dt > 0.0))
    return fail(h, LMPC_ERR_ARGUMENT, std::string(who) + ": bad argument");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(lmpc_prepare_kernel, dim3((batch + 255) / 256), dim3(256), 0, h->stream, h->P, batch, *track, x_ic,
                     status, dt, speed_scale, speed_limit, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}
}  // namespace

int lmpc_prepare_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const double* x_ic, double dt,
                       double speed_scale, double speed_limit, double* X_ref, double* U_ref, double* T_ref,
                       double* bound_left, double* bound_right, double* curvatures, double* vel_ref) {
  return prepare_impl(h, "lmpc_prepare_batch", batch, track, x_ic, nullptr, dt, speed_scale, speed_limit, X_ref, U_ref,
                      T_ref, bound_left, bound_right, curvatures, vel_ref);
}

int lmpc_prepare_failed_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const double* x_ic,
                              const int32_t* status, double dt, double speed_scale, double speed_limit, double* X_ref,
                              double* U_ref, double* T_ref, double* bound_left, double* bound_right, double* curvatures,
                              double* vel_ref) {
  if (h && !status) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_prepare_failed_batch: status is NULL");
  return prepare_impl(h, "lmpc_prepare_failed_batch", batch, track, x_ic, status, dt, speed_scale, speed_limit, X_ref,
                      U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref);
}

static bool track_ok(const lmpc_track* t) {
  return t && t->curvature && t->bound_left && t->bound_right && t->vel && t->M >= 2 && t->L > 0.0;
}

int lmpc_shift_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const double* X_sol, const double* U_sol,
                     const double* X_old, const double* U_old, const int32_t* status, double dt, double speed_scale,
                     double speed_limit, double* X_ref, double* U_ref, double* T_ref, double* bound_left,
                     double* bound_right, double* curvatures, double* vel_ref) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !track_ok(track) || !X_sol || !U_sol || !X_ref || !U_ref || !T_ref || !bound_left || !bound_right ||
      !curvatures || !vel_ref || !(dt > 0.0) || (status && (!X_old || !U_old)))
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_shift_batch: bad argument");
  if (X_ref == X_sol || X_ref == X_old || U_ref == U_sol || U_ref == U_old)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_shift_batch: outputs must not alias inputs");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(lmpc_shift_kernel, dim3((batch + 255) / 256), dim3(256), 0, h->stream, h->P, batch, *track, X_sol,
                     U_sol, X_old ? X_old : X_sol, U_old ? U_old : U_sol, status, dt, speed_scale, speed_limit, X_ref,
                     U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}

int lmpc_plant_step_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, double* x, const double* u,
                          double dt_sim, int32_t n_sub) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !track_ok(track) || !x || !u || !(dt_sim > 0.0) || n_sub < 1)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_plant_step_batch: bad argument");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(lmpc_plant_kernel, dim3((batch + 255) / 256), dim3(256), 0, h->stream, h->P, batch, *track, x, u,
                     dt_sim, n_sub);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}

int lmpc_loop_advance_batch(lmpc_handle* h, int32_t batch, const lmpc_track* track, const int32_t* status, const int32_t* iters,
                            const double* X_optm, const double* U_optm, double* x, double* u_prev, double dt, double dt_sim, int32_t n_sub,
                            double speed_scale, double speed_limit, int32_t restart_failed, double* X_ref, double* U_ref, double* T_ref,
                            double* bound_left, double* bound_right, double* curvatures, double* vel_ref, double* distance,
                            double* worst_excess, int64_t* n_fail, uint64_t* n_accepted) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !track_ok(track) || !status || !X_optm || !U_optm || !x || !u_prev || !X_ref || !U_ref || !T_ref || !bound_left ||
      !bound_right || !curvatures || !vel_ref || !(dt > 0.0) || !(dt_sim > 0.0) || n_sub < 1 || (n_accepted && !iters))
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_loop_advance_batch: bad argument");
  if (X_ref == X_optm || U_ref == U_optm) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_loop_advance_batch: the references must not alias the solution");
  if (h->out_aos) return fail(h, LMPC_ERR_UNSUPPORTED, "lmpc_loop_advance_batch reads the solution in the [component][knot][batch] layout");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  static_assert(sizeof(long long) == sizeof(int64_t) && sizeof(unsigned long long) == sizeof(uint64_t), "counter types");
  lmpc_params PL = h->P;
  PL.warm_flag = (h->warm_flag && h->warm_flag_n == batch) ? h->warm_flag : nullptr;  // the last solve of this batch was a warm one
  hipLaunchKernelGGL(lmpc_loop_advance_kernel, dim3((batch + 63) / 64), dim3(64 * LMPC_LOOP_WAVES), 0, h->stream, PL, batch, *track, status, iters, X_optm,
                     U_optm, x, u_prev, dt, dt_sim, n_sub, speed_scale, speed_limit, restart_failed ? 1 : 0, X_ref, U_ref, T_ref, bound_left,
                     bound_right, curvatures, vel_ref, distance, worst_excess, reinterpret_cast<long long*>(n_fail),
                     reinterpret_cast<unsigned long long*>(n_accepted));
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}

int lmpc_set_safe_set(lmpc_handle* h, int32_t n_laps, const int32_t* n_pts, const double* x, double total_length) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (n_laps < 0 || (n_laps > 0 && (!n_pts || !x)) || !(total_length > 0.0))
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_safe_set: bad argument");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  if (h->ss_npts) HIP_TRY(h, hipFree(h->ss_npts));
  if (h->ss_off) HIP_TRY(h, hipFree(h->ss_off));
  if (h->ss_x) HIP_TRY(h, hipFree(h->ss_x));
  h->ss_npts = h->ss_off = nullptr;
  h->ss_x = nullptr;
  h->ss_laps = 0;
  h->ss_total = 0;
  h->ss_nmax = 0;
  h->ss_L = total_length;
  ++h->ss_gen;  // codes of an earlier lmpc_ss_query_idx_batch no longer name rows of this store (lmpc_solve_batch_ss_idx refuses them)
  // SafeSetManager keeps at most max_lap_stored laps (boost::circular_buffer, safe_set.cpp:139-151)
  int first = 0;
  if (h->cfg.max_lap_stored > 0 && n_laps > h->cfg.max_lap_stored) first = n_laps - h->cfg.max_lap_stored;
  std::vector<int> npts, off;
  size_t skip = 0, total = 0;
  for (int l = 0; l < n_laps; ++l) {
    if (n_pts[l] < 1) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_safe_set: empty lap");
    if (l < first) {
      skip += (size_t)n_pts[l];
      continue;
    }
    off.push_back((int)total);
    npts.push_back(n_pts[l]);
    if (n_pts[l] > h->ss_nmax) h->ss_nmax = n_pts[l];
    total += (size_t)n_pts[l];
  }
  if (npts.empty()) return LMPC_OK;
  HIP_TRY(h, hipMalloc(&h->ss_npts, npts.size() * sizeof(int)));
  HIP_TRY(h, hipMalloc(&h->ss_off, off.size() * sizeof(int)));
  HIP_TRY(h, hipMalloc(&h->ss_x, total * 6 * sizeof(double)));
  HIP_TRY(h, hipMemcpy(h->ss_npts, npts.data(), npts.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->ss_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->ss_x, x + skip * 6, total * 6 * sizeof(double), hipMemcpyHostToDevice));
  h->ss_laps = (int)npts.size();
  h->ss_total = (int)total;
  return LMPC_OK;
}

namespace {
int ss_query_launch(lmpc_handle* h, int32_t batch, const double* query, double* ss_x, double* ss_j, int32_t* n_found, int32_t* ss_idx);
}

int lmpc_ss_query_batch(lmpc_handle* h, int32_t batch, const double* query, double* ss_x, double* ss_j,
                        int32_t* n_found) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !query || !ss_x || !ss_j || !n_found)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_ss_query_batch: null pointer or negative batch");
  return ss_query_launch(h, batch, query, ss_x, ss_j, n_found, nullptr);
}

int lmpc_ss_query_idx_batch(lmpc_handle* h, int32_t batch, const double* query, int32_t* ss_idx, int32_t* n_found) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !query || !ss_idx || !n_found)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_ss_query_idx_batch: null pointer or negative batch");
  h->ss_idx_gen = h->ss_gen;
  return ss_query_launch(h, batch, query, nullptr, nullptr, n_found, ss_idx);
}

namespace {
int ss_query_launch(lmpc_handle* h, int32_t batch, const double* query, double* ss_x, double* ss_j, int32_t* n_found, int32_t* ss_idx) {
  if (h->cfg.num_ss_pts < 1 || h->cfg.num_ss_pts_per_lap < 1)
    return fail(h, LMPC_ERR_ARGUMENT, "num_ss_pts / num_ss_pts_per_lap not configured");
  if (h->cfg.num_ss_pts_per_lap > 64)
    return fail(h, LMPC_ERR_UNSUPPORTED, "num_ss_pts_per_lap > 64");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t lds = (size_t)3 * (h->ss_nmax > 0 ? h->ss_nmax : 1) * sizeof(double);
  if (lds > 160 * 1024) return fail(h, LMPC_ERR_UNSUPPORTED, "lap longer than 6826 samples");
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lmpc_ss_query_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(lmpc_ss_query_kernel, dim3(8 * ((batch + 7) / 8)), dim3(64), lds, h->stream, batch, h->ss_laps, h->cfg.num_ss_pts,
                     h->cfg.num_ss_pts_per_lap, h->ss_npts, h->ss_off, h->ss_x, h->ss_L, query, ss_x, ss_j, n_found,
                     (double*)nullptr, ss_idx);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}
}  // namespace

int lmpc_ss_query_host(lmpc_handle* h, const double* query, double* ss_x, double* ss_j, int32_t* n_found, double* j0) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (!query || !ss_x || !ss_j || !n_found) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_ss_query_host: null pointer");
  const int S = h->cfg.num_ss_pts;
  if (S < 1 || h->cfg.num_ss_pts_per_lap < 1) return fail(h, LMPC_ERR_ARGUMENT, "num_ss_pts / num_ss_pts_per_lap not configured");
  if (h->cfg.num_ss_pts_per_lap > 64) return fail(h, LMPC_ERR_UNSUPPORTED, "num_ss_pts_per_lap > 64");
  const size_t lds = (size_t)3 * (h->ss_nmax > 0 ? h->ss_nmax : 1) * sizeof(double);
  if (lds > 160 * 1024) return fail(h, LMPC_ERR_UNSUPPORTED, "lap longer than 6826 samples");
  HIP_TRY(h, hipSetDevice(h->device));
  const size_t nd = 2 + 7 * (size_t)S + 1;  // query | ss_x [6][S] | ss_j [S] | j0
  double* const d = h->ssq_dev;   // handle-owned staging (lmpc_create): nothing is allocated or freed per call
  int* const di = h->ssq_int;
  double* const host = h->ssq_host;
  if (!d || !di || !host) return fail(h, LMPC_ERR_RUNTIME, "lmpc_ss_query_host: staging not allocated");
  for (size_t e = 0; e < nd; ++e) host[e] = 0.0;
  host[0] = query[0];
  host[1] = query[1];
  *h->ssq_int_host = 0;
  HIP_TRY(h, hipMemcpyAsync(d, host, nd * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(di, h->ssq_int_host, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&lmpc_ss_query_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(lmpc_ss_query_kernel, dim3(8), dim3(64), lds, h->stream, 1, h->ss_laps, S, h->cfg.num_ss_pts_per_lap,
                     h->ss_npts, h->ss_off, h->ss_x, h->ss_L, d, d + 2, d + 2 + 6 * (size_t)S, di, d + 2 + 7 * (size_t)S, (int*)nullptr);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host, d, nd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->ssq_int_host, di, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const int nf = *h->ssq_int_host;
  for (int j = 0; j < S; ++j) {
    for (int k = 0; k < 6; ++k) ss_x[(size_t)j * 6 + k] = host[2 + (size_t)k * S + j];  // -> column-major 6 x S
    ss_j[j] = host[2 + 6 * (size_t)S + j];
  }
  *n_found = nf;
  if (j0) *j0 = host[2 + 7 * (size_t)S];
  return LMPC_OK;
}

int lmpc_set_regression_laps(lmpc_handle* h, int32_t n_laps, const int32_t* n_pts, const double* x, const double* u,
                             const double* k, const double* t, const lmpc_regression_spec* spec) {
  if (!h) return LMPC_ERR_ARGUMENT;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (void* p : {(void*)h->reg_end, (void*)h->reg_x, (void*)h->reg_u, (void*)h->reg_y, (void*)h->reg_tab})
    if (p) HIP_TRY(h, hipFree(p));
  h->reg_end = nullptr;
  h->reg_x = h->reg_u = h->reg_y = h->reg_tab = nullptr;
  h->reg_npad = 0;
  h->reg_on = false;
  h->reg_total = 0;
  if (n_laps == 0 || !spec) return LMPC_OK;
  if (n_laps < 0 || !n_pts || !x || !u || !k || !t) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_regression_laps: bad argument");
  const int nf = spec->n_in_state + spec->n_in_ctrl;
  if (!((nf == 5 && spec->n_out == 3) || (nf == 8 && spec->n_out == 6)) || spec->n_in_state < 1 || spec->n_in_state > 6 ||
      spec->n_in_ctrl < 0 || spec->n_in_ctrl > 2)
    return fail(h, LMPC_ERR_UNSUPPORTED, "regression built for (features, rows) = (5, 3) and (8, 6)");
  if (!(spec->dist_max > 0.0)) return fail(h, LMPC_ERR_ARGUMENT, "dist_max must be positive");
  for (int i = 0; i < spec->n_out; ++i)
    if (spec->out[i] < 0 || spec->out[i] > 5) return fail(h, LMPC_ERR_ARGUMENT, "regressed row out of range");
  for (int i = 0; i < spec->n_in_state; ++i)
    if (spec->in_state[i] < 0 || spec->in_state[i] > 5) return fail(h, LMPC_ERR_ARGUMENT, "feature state out of range");
  for (int i = 0; i < spec->n_in_ctrl; ++i)
    if (spec->in_ctrl[i] < 0 || spec->in_ctrl[i] > 1) return fail(h, LMPC_ERR_ARGUMENT, "feature control out of range");
  size_t total = 0;
  std::vector<int> end;
  for (int l = 0; l < n_laps; ++l) {
    if (n_pts[l] < 2) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_regression_laps: a lap needs two samples");
    total += (size_t)n_pts[l];
    end.resize(total, 0);
    end[total - 1] = 1;
  }
  // upload-time temporaries: freed on every way out of this function
  struct scratch_dev {
    void* p = nullptr;
    ~scratch_dev() {
      if (p) (void)hipFree(p);
    }
  } tk, tt, tv;
  double *dk = nullptr, *dt = nullptr;
  HIP_TRY(h, hipMalloc(&h->reg_end, total * sizeof(int)));
  HIP_TRY(h, hipMalloc(&h->reg_x, total * 6 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&h->reg_u, total * 2 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&h->reg_y, total * 6 * sizeof(double)));
  HIP_TRY(h, hipMalloc(&tk.p, total * sizeof(double)));
  HIP_TRY(h, hipMalloc(&tt.p, total * sizeof(double)));
  dk = static_cast<double*>(tk.p);
  dt = static_cast<double*>(tt.p);
  HIP_TRY(h, hipMemcpy(h->reg_end, end.data(), total * sizeof(int), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->reg_x, x, total * 6 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(h->reg_u, u, total * 2 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(dk, k, total * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(h, hipMemcpy(dt, t, total * sizeof(double), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(lmpc_reg_residual_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->P.veh,
                     (int)total, spec->as_written ? 1 : 0, h->reg_end, h->reg_x, h->reg_u, dk, dt, h->reg_y);
  HIP_TRY(h, hipGetLastError());
  {  // dense table of the samples that have a successor (what the per-solve kernel streams)
    std::vector<int> valid;
    valid.reserve(total);
    for (size_t j = 0; j < total; ++j)
      if (!end[j]) valid.push_back((int)j);
    const int nvalid = (int)valid.size(), npad = (nvalid + 3) / 4 * 4;
    HIP_TRY(h, hipMalloc(&tv.p, (size_t)nvalid * sizeof(int)));
    int* dvalid = static_cast<int*>(tv.p);
    HIP_TRY(h, hipMalloc(&h->reg_tab, (size_t)npad * (size_t)(nf + spec->n_out + 1) * sizeof(double)));
    HIP_TRY(h, hipMemcpy(dvalid, valid.data(), (size_t)nvalid * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(lmpc_reg_pack_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, *spec, nvalid, npad, dvalid,
                       h->reg_x, h->reg_u, h->reg_y, h->reg_tab, h->reg_tab + (size_t)npad * (size_t)(nf + spec->n_out));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->reg_npad = npad;
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->reg_total = (int)total;
  h->reg_spec = *spec;
  h->reg_on = true;
  return LMPC_OK;
}

int lmpc_set_output_layout(lmpc_handle* h, int32_t layout) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (layout != LMPC_LAYOUT_SOA && layout != LMPC_LAYOUT_AOS) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_output_layout: unknown layout");
  h->out_aos = layout == LMPC_LAYOUT_AOS;
  return LMPC_OK;
}

int lmpc_set_warm_rounds(lmpc_handle* h, int32_t rounds) {
  if (!h) return LMPC_ERR_ARGUMENT;
  static_assert(LMPC_WARM_ROUNDS_MAX == 4, "the message below");
  if (rounds < 0 || rounds > LMPC_WARM_ROUNDS_MAX) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_warm_rounds: 0 (the default) or 1 .. 4 rounds");
  h->warm_rounds = rounds;
  return LMPC_OK;
}

int lmpc_set_launch_order(lmpc_handle* h, const int32_t* order, int32_t batch) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (order && batch <= 0) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_set_launch_order: an order needs its length (batch > 0)");
  h->order = order;
  h->order_n = order ? batch : 0;
  return LMPC_OK;
}

int lmpc_launch_order_from_iters(lmpc_handle* h, int32_t batch, const int32_t* iters, int32_t* order) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !iters || !order) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_launch_order_from_iters: null pointer or negative batch");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(lmpc_launch_order_kernel, dim3(1), dim3(1024), 0, h->stream, batch, iters, order);
  HIP_TRY(h, hipGetLastError());
  return LMPC_OK;
}

int lmpc_regress_batch(lmpc_handle* h, int32_t batch, const double* X_ref, const double* U_ref, double* A, double* Bm,
                       double* g) {
  if (!h) return LMPC_ERR_ARGUMENT;
  if (batch < 0 || !X_ref || !U_ref || !A || !Bm || !g)
    return fail(h, LMPC_ERR_ARGUMENT, "lmpc_regress_batch: null pointer or negative batch");
  if (!h->reg_on) return fail(h, LMPC_ERR_ARGUMENT, "lmpc_regress_batch: no regression laps set");
  if (batch == 0) return LMPC_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  return launch_regress<false>(h, batch, X_ref, U_ref, A, Bm, g);
}

}  // extern "C"
