// lmpc_solve_kernel.hip -- the batched QP solve of RacingMPC::solve on gfx950 (CDNA4): fp64, and the same source in fp32.
//
// What it replaces: opti_.solve_limited() on the "conic"/OSQP problem built in
//   src/mpc/racing_mpc/src/racing_mpc.cpp:106-201 (constraints), :442-477 (tracking cost),
//   :524-543 (boundary slack); actuator boxes from single_track_planar_model.cpp:113-120,144-151.
//
// Mapping: ONE WAVEFRONT (64 lanes) PER PROBLEM, one wave per workgroup, everything the
// iteration touches resident in LDS (~20 KB at N = 20 -> 8 problems per CU).
//   * Riccati factorisation: the augmented state z = [x; u_prev] has 8 components, so the 8x8
//     cost-to-go matrix is exactly one wave: lane l owns element (r, c) = (l >> 3, l & 7).
//     Matrix products are 6-term dot products whose operands come as LDS row reads (b128; padded /
//     skewed rows keep them bank-conflict free); each lane keeps its own element in a register.
//   * Riccati vector solves: lane (s, r) = (l >> 3, l & 7) computes component r of right-hand
//     side s, so the predictor step and the boundary-slack Schur vector are solved in the same
//     instruction stream (two RHS for the price of one); the running vector makes one LDS trip per
//     stage, the stage's 2-vector is spread with ds_swizzle / v_readlane.
//   * Inequality rows: the 11 two-sided slots of each knot (6 state, 2 input, 2 input-rate,
//     1 track boundary) are dealt round-robin to lanes; slacks and multipliers never leave
//     registers.  The slot owner also owns the primal component the slot constrains: it writes
//     that component's barrier weight and gradient entry and applies its update.
//   * Wave-wide scalars (mu, step length, Schur dot products) are DPP reductions on the VALU and
//     live in SGPRs afterwards.
//   * The whole iteration is a chain of ~120 dependent stage steps: operands that do not depend on
//     the chain are fetched one stage ahead, exchanges through LDS use compiler-only fences (one
//     wave per workgroup: no s_barrier, no wait for the write to retire).
// Too small for MFMA (6..8-wide blocks); the kernel is bound by LDS instruction issue and FP64
// VALU issue, not by HBM.
//
// Algorithm (twin of oracle/c/lmpc_oracle.c, which documents the derivation): Mehrotra
// predictor-corrector interior point; Newton systems by Riccati recursion on (z, v = dU); the
// shared boundary slack sigma (one scalar coupling all knots) by a Schur complement.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "lmpc_device.h"

// waves per SIMD asked of the compiler for the single-precision N <= 23 kernel (its 10 KB records allow 16 per CU):
// measured 1.04 / 0.93 / 0.97 ms per 8192-batch at 2 / 3 / 4 -- the issue ceiling (LDS pipe, VALU) is ~12 % away
#ifndef OCCF
#define OCCF 3
#endif
#define NSLOT 11
// resident waves per SIMD the register allocation is sized for: the fp64 tracking kernels up to N = 23 and every fp32
// kernel up to N = 40 run two (three for fp32, N <= 23); the fp64 LMPC and long-horizon kernels need the full file
constexpr int lmpc_waves_per_simd(int real_bytes, int kq, int ks) {
  return real_bytes == 4 ? ((kq <= 4 && ks == 0) ? OCCF : (kq <= 7 ? 2 : 1)) : ((ks == 0 && kq <= 4) ? 2 : 1);
}

// Optional per-phase cycle accounting (make prof -> -DLMPC_PHASE_TIMING): one s_memtime read per
// phase boundary, per-wave totals written over kkt_out as [16][B] doubles (caller allocates 20 rows).
#ifdef LMPC_PHASE_TIMING
struct Prof {
  long long acc[16];
  long long t, w0;
};
#define PT_DECL Prof pf; { for (int k = 0; k < 16; ++k) pf.acc[k] = 0; pf.t = __builtin_readcyclecounter(); pf.w0 = wall_clock64(); }
#define PT_MARK(k) { const long long pt_n = __builtin_readcyclecounter(); pf.acc[k] += pt_n - pf.t; pf.t = pt_n; }
#else
struct Prof {};
#define PT_DECL Prof pf;
#define PT_MARK(k)
#endif
#define SL_U 6
#define SL_V 8
#define SL_EY 10

// ---- LDS layout (doubles) ---------------------------------------------------------------------
// stage record i (stride 78): M[8][8]: column c of the stage model with the feedback gain appended,
//                               M[c][k] = [A B][k][c] (k < 6), M[c][6 + j] = K_j[c].  Row c starts at
//                               ST_ROW(c) = 8 c + 2 (c >> 1): the two-cell skew after every second row puts
//                               the eight rows on eight different 4-bank groups, so the per-lane row reads
//                               (b128, lane = row) are conflict-free; column reads (lane = column) stay
//                               contiguous.  The six skew cells and the tail of the record hold
//                               Hinv (h00,h01) @16 | (h11, dt) @34 | kff rhs0 [2] @52 | kff rhs1 [2] @70 | g[6] @72
// knot record i (stride 36):  z[8] v[2] @0 | rhs0: Th / q / d [10] @10 | rhs1: q / e [10] @20
//                             | csig @30 | eyT / eyD @31 | boundary row bounds (hi, lo) @32 | qlin_vx @34
// tail: P[8][10] @0 | W[8][10] @80 | Y[8][10] @160 | pvec[2 buf][2 rhs][8] @240 | consts @272 (48)
#define ST_ROW(c) (8 * (c) + 2 * ((c) >> 1))
#define ST_HI 16     // h00, h01
#define ST_HI11 34   // h11
#define ST_DT 35
#define ST_KFF(s) ((s) ? 70 : 52)
#define ST_G 72
#define KN_R0 10
#define KN_R1 20
#define KN_CSIG 30
#define KN_EY 31
#define KN_BHL 32
#define KN_QLIN 34
#define TL_P 0
#define TL_W 80
#define TL_Y 160
#define TL_PV 240
#define TL_CT 272
#define CT_QD 0
#define CT_QT 6
#define CT_QU 12
#define CT_SV 16
#define CT_HL 20    // box bounds of the ten primal components, (hi, lo) interleaved
#define CT_ZERO 40  // a 0.0 entry: coefficient slot for "no term"
#define CT_E 41     // 2 * convex_hull_slack (LMPC)
// LMPC extension of the tail (only allocated when learning): terminal-block quantities (see term_factor_u)
// (offsets in `treal` cells from the start of the terminal region, which follows the real-typed records and tail)
#define TL_PT 0      // PT[6][6]: terminal cost-to-go contributed by the safe-set block
#define TL_TG 36     // terminal gradient contribution  E eps + pT
#define TL_EPS 42    // eps = (x_T - ss0) - (SS - ss0 1') lambda
#define TL_FB 48     // F_B^-1 [6][6], F_B = E^-1 + U_B Th_B^-1 U_B' (the points eliminated through 1/theta)
#define TL_WA 84     // W_A = F_B^-1 U_A, column a at +6a
#define TL_UA 120    // u of the explicit points, point a at +6a
#define TL_LC 156    // C_A^-1 [6][6] (full, symmetric), C_A = Theta_A + U_A'F_B^-1 U_A; C_A itself while it is being formed
#define TL_X1 192    // C_A^-1 (1_A - W_A'a_B)
#define TL_G 198     // g = E U M^-1 1
#define TL_AB 204    // a_B = U_B Th_B^-1 1
#define TL_RA 210    // right-hand side of the explicit points (written by their owner lanes)
#define TL_THA 216   // theta of the explicit points
#define TL_XA 222    // their step d lambda_A (read back by the owner lanes)
#define TL_S11 228   // s11 = 1'M^-1 1
#define TL_E 230     // E = 2 convex_hull_slack (exact, whatever `real` is)
#define TL_Z 236     // Z = C_A^-1 W_A' [6][6] (a product of term_factor_u)
#define TL_UL 272    // the (centred) safe-set points, point-major [S][6]
// Explicit points at most (the smallest theta below tau).  Four until round 5 ("supports of 1-3 points are what occurs"): with a
// FIVE-lap safe set the optimum blends one point per lap, a support of five, on ~0.1 % of the bench distribution at N = 27 .. 29
// and on most problems at N <= 5 (tests/dispatch_sweep.py found them).  The fifth point then went through 1 / theta with theta ->
// 1e-12: cond(F_B) 1e12, Newton steps with a stationarity residual of O(1), and an answer 1e-2 from the optimum reported OPTIMAL --
// by the kernel and its twin alike, so kernel-against-twin tests could not see it; the dense oracle did.  Six is what the terminal
// block can hold (C_A = Theta_A + U_A'F_B^-1 U_A has rank <= 6 as Theta_A -> 0) and what its LDS cells were laid out for.
#define MA_MAX 6
#define TAU_REL 1e-5   // tau = TAU_REL * max_j u_j'E u_j: cond(F_B) <= ~1e5 whatever the iteration does
#define STALL_MU 1e-9  // complementarity below which a step that does not lower it ends the solve
#define STALL_STEP 1e-6  // ... and the scaled size of that step above which the stalled iterate, unless the polish verifies it, is MAX_ITER
#define NBHD_GAMMA 1e-2  // once mu has risen: no complementarity product below this fraction of their mean after a step (1e-3
                        // does not stop the cycle the rule is there for; applied to every problem 3e-2 costs 13 % more iterations)
#define NBHD_TRIALS 3   // cuts of the step length by 0.6 at most (two are what the cycling problem needs; bounded so that a point already
                        // outside the neighbourhood cannot freeze the iteration)
#define F_UP 1
#define F_LO 2
#define F_SIG 4
#define F_QLIN 8
#define F_MOVE 16
#define F_EY 32
#define F_SCH 64
// Row r of the 8x8 work matrices P, W, Y starts at MROWS(r) = 8 r + 2 (r >> 1) -- the stage records' skew (ST_ROW): the eight rows
// sit on eight different 4-bank groups (conflict-free b128 row reads, as with the stride of 10 doubles used until round 5), AND the
// rows of a 16-lane store group (r = 2g, 2g + 1) are 16 banks apart, so the element stores of W, Y, P -- ds_write_b64: contiguous
// 16-lane groups, 32 banks -- are conflict-free too; with the stride of 10 rows 2g and 2g + 1 overlapped in four banks: every one of
// the four stores per factor stage took 8 LDS cycles instead of 4 (40 % of the headline kernel's SQ_LDS_BANK_CONFLICT, round 6).
#define MROWS(r) (8 * (r) + 2 * ((r) >> 1))

// The workgroup is a single wavefront and the LDS executes one wave's DS instructions in issue order, so cross-lane
// exchange through LDS needs no s_barrier and no wait for the write to retire: only the compiler must not move memory
// operations across the exchange point.  __builtin_amdgcn_wave_barrier alone does not say that -- it is declared as not
// touching memory, so around a store that only SOME lanes execute (`if (lane < 6) T[..] = ..`) the IR-level passes may
// still schedule the other lanes' later loads of those cells, on the not-taken path, ahead of the taken path's stores:
// the readers then see the previous content (seen once on the terminal block of the learning problem: results changed
// from process to process).  Release / acquire fences at WAVEFRONT scope around the barrier pin the order; at that scope
// they emit no cache action and no wait.
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void wave_sync() { wave_fence(); }

// A value that is the same in every lane, moved to scalar registers (v_readfirstlane): the solver's
// wave-wide scalars (mu, step lengths, sigma, ...) then cost no vector registers while they are carried
// across the Riccati sweeps, and feed the VALU as scalar operands.
__device__ __forceinline__ double uni(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

// Lane k's value of a wave-distributed number as a wave-uniform scalar (v_readlane_b32; the result lives in SGPRs
// and feeds the FMAs as a scalar operand).
__device__ __forceinline__ double lane_bcast(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }

// ds_swizzle bit mode: source lane = (lane & 0x18) | K, i.e. lane K of each group of 8 (PATTERN = 0x18 | K << 5)
template <int PATTERN>
__device__ __forceinline__ double group_bcast(double v) {
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), PATTERN), __builtin_amdgcn_ds_swizzle(__double2loint(v), PATTERN));
}
template <int PATTERN>
__device__ __forceinline__ float group_bcast(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), PATTERN)); }

// Wave reductions on the VALU (DPP), not through the LDS crossbar (ds_bpermute): the LDS pipeline is this kernel's
// tightest resource and a 6-step bpermute chain costs ~460 cycles of latency against ~130 here.  Four row_ror steps
// leave every lane of a 16-lane row with the row's total, row_bcast15 / row_bcast31 fold the rows into lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double x, double identity) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(x), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float x, float identity) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
#define DPP_ROW_ROR(n) (0x120 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
struct op_sum {
  template <typename real> static __device__ __forceinline__ real id() { return real(0); }
  template <typename real> static __device__ __forceinline__ real f(real a, real b) { return a + b; }
};
struct op_max {
  template <typename real> static __device__ __forceinline__ real id() { return -real(INFINITY); }
  template <typename real> static __device__ __forceinline__ real f(real a, real b) { return fmax(a, b); }
};
struct op_min {
  template <typename real> static __device__ __forceinline__ real id() { return real(INFINITY); }
  template <typename real> static __device__ __forceinline__ real f(real a, real b) { return fmin(a, b); }
};
// NV independent reductions in lock-step (their steps interleave); results as wave-uniform scalars
template <class OP, int NV, typename real>
__device__ __forceinline__ void wave_reduce_n(real (&v)[NV]) {
  const real id = OP::template id<real>();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = OP::f(v[k], dpp_move<DPP_ROW_ROR(8), 0xf>(v[k], id));
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = OP::f(v[k], dpp_move<DPP_ROW_ROR(4), 0xf>(v[k], id));
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = OP::f(v[k], dpp_move<DPP_ROW_ROR(2), 0xf>(v[k], id));
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = OP::f(v[k], dpp_move<DPP_ROW_ROR(1), 0xf>(v[k], id));
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = OP::f(v[k], dpp_move<DPP_ROW_BCAST15, 0xa>(v[k], id));
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = OP::f(v[k], dpp_move<DPP_ROW_BCAST31, 0xc>(v[k], id));
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = lane_bcast(v[k], 63);
}
template <typename real>
__device__ __forceinline__ real wave_sum(real x) {
  real v[1] = {x};
  wave_reduce_n<op_sum, 1>(v);
  return v[0];
}
template <typename real>
__device__ __forceinline__ real wave_max(real x) {
  real v[1] = {x};
  wave_reduce_n<op_max, 1>(v);
  return v[0];
}
template <typename real>
__device__ __forceinline__ real wave_min(real x) {
  real v[1] = {x};
  wave_reduce_n<op_min, 1>(v);
  return v[0];
}
template <int NV, typename real>
__device__ __forceinline__ void wave_sum_n(real (&v)[NV]) {
  wave_reduce_n<op_sum, NV>(v);
}

// Exchange with the partner lane that differs in bit BIT of the lane number (and, for bits 2 and 3, in the bits below:
// the row mirrors are the involutions DPP offers there) -- every pairing used by wave_sum_split below.
template <int BIT>
__device__ __forceinline__ double pair_exchange(double x) {
  if constexpr (BIT == 5) return __shfl_xor(x, 32, 64);
  if constexpr (BIT == 4)  // ds_swizzle, bit mode: and 0x1f, or 0, xor 0x10
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x401F), __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x401F));
  if constexpr (BIT == 3) return dpp_move<0x140, 0xf>(x, 0.0);  // row_mirror
  if constexpr (BIT == 2) return dpp_move<0x141, 0xf>(x, 0.0);  // row_half_mirror
  if constexpr (BIT == 1) return dpp_move<0x4E, 0xf>(x, 0.0);   // quad_perm [2,3,0,1]
  return dpp_move<0xB1, 0xf>(x, 0.0);                            // quad_perm [1,0,3,2]
}
template <int BIT>
__device__ __forceinline__ float pair_exchange(float x) { return (float)pair_exchange<BIT>((double)x); }

// Many sums at once, for NV up to 32 (the safe-set block reduces 35 per iteration): instead of NV full reductions, each
// step pairs the lanes across one bit of the lane number and SPLITS the values between the partners -- the lane with
// the bit clear keeps the lower half (adding its partner's contributions), the other the upper half -- so the work
// halves with every step: P/2 + P/4 + ... exchanges for P values instead of 6 P.  After log2 P steps lane l holds the
// partial total of value l >> (6 - log2 P) over its group; plain pairwise sums over the remaining bits finish it.
template <int NV, typename real>
__device__ __forceinline__ void wave_sum_split(real (&v)[NV], int lane) {
  constexpr int P = NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : 32;
  constexpr int LOGP = P == 2 ? 1 : P == 4 ? 2 : P == 8 ? 3 : P == 16 ? 4 : 5;
  static_assert(NV <= 32, "wave_sum_split handles up to 32 values");
  real a[P];
#pragma unroll
  for (int k = 0; k < P; ++k) a[k] = k < NV ? v[k] : real(0);
  auto split = [&](auto bit_c, auto half_c) {
    constexpr int BIT = decltype(bit_c)::value, H = decltype(half_c)::value;
    const bool up = (lane >> BIT) & 1;
#pragma unroll
    for (int k = 0; k < H; ++k) {
      const real keep = up ? a[k + H] : a[k];
      const real send = up ? a[k] : a[k + H];
      a[k] = keep + pair_exchange<BIT>(send);
    }
  };
  auto fold = [&](auto bit_c) {
    constexpr int BIT = decltype(bit_c)::value;
    a[0] = a[0] + pair_exchange<BIT>(a[0]);
  };
  using std::integral_constant;
  // bits 5, 4, 3, 2, 1 carry the splits while more than one value is left; the rest are plain sums
  if constexpr (LOGP >= 1) split(integral_constant<int, 5>{}, integral_constant<int, P / 2>{}); else fold(integral_constant<int, 5>{});
  if constexpr (LOGP >= 2) split(integral_constant<int, 4>{}, integral_constant<int, P / 4>{}); else fold(integral_constant<int, 4>{});
  if constexpr (LOGP >= 3) split(integral_constant<int, 3>{}, integral_constant<int, (P / 8 > 0 ? P / 8 : 1)>{}); else fold(integral_constant<int, 3>{});
  if constexpr (LOGP >= 4) split(integral_constant<int, 2>{}, integral_constant<int, (P / 16 > 0 ? P / 16 : 1)>{}); else fold(integral_constant<int, 2>{});
  if constexpr (LOGP >= 5) split(integral_constant<int, 1>{}, integral_constant<int, 1>{}); else fold(integral_constant<int, 1>{});
  fold(integral_constant<int, 0>{});
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = lane_bcast(a[0], k << (6 - LOGP));
}

// 1/x: hardware v_rcp seed + one Newton step (full accuracy for normal x); replaces the ~12-instruction IEEE
// division sequence in the per-row arithmetic.
__device__ __forceinline__ double frcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float frcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}

// 1/sqrt(x): hardware v_rsq seed + Newton steps y <- y + y (1 - x y^2)/2 (two in double: the seed carries ~26 bits);
// the IEEE sqrt + division pair it replaces is ~70 dependent instructions, ten times per terminal factorisation.
__device__ __forceinline__ double frsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
  return __builtin_fma(0.5 * y, __builtin_fma(-x * y, y, 1.0), y);
}
__device__ __forceinline__ float frsqrt(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  return __builtin_fmaf(0.5f * y, __builtin_fmaf(-x * y, y, 1.0f), y);
}

__device__ __forceinline__ double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Per-precision constants of the iteration.  fp32: the complementarity floor of a single-precision Riccati
// recursion is ~1e-6 (weights lam/t ~ 1e5 already cancel four digits in P), rows are feasible to ~1e-4.
template <typename real> struct ipm_limits;
template <> struct ipm_limits<double> {
  static __device__ __forceinline__ double tol(double cfg) { return cfg; }
  static constexpr double rd_ok = 1e-9, rd_infeasible = 1e-6, tiny = 1e-300;
  static constexpr double rd_distress = 1e-6;  // a rise of mu counts as distress only in the (nearly) feasible end game
};
template <> struct ipm_limits<float> {
  static __device__ __forceinline__ float tol(double cfg) { return fmaxf((float)cfg, 2e-6f); }
  static constexpr float rd_ok = 1e-4f, rd_infeasible = 1e-2f, tiny = 1e-30f;
  static constexpr float rd_distress = 1e-4f;
};
// Active-set polish (what OSQP's polish = true is to the reference, racing_mpc.cpp:90-95; derivation and measurements in
// oracle/c/lmpc_oracle.c, which runs the same rounds): rows with lam > t are HELD -- weight theta on them, none on the
// others, one stabilised factorisation -- then `steps` multiplier steps on that factor (gradient y + theta * residual on
// the held rows, full Newton step, y <- y + theta * (residual + the row's own increment)), a KKT test (held rows met to
// `feas` with y >= -dual, the others satisfied to `feas`, the last step below step_tol in the reference's scaled units),
// and up to `rounds` repairs of the held set.  Double precision
// tries it once as soon as mu <= mu_early with rows feasible to rd_early -- about two iterations before the interior
// point's own tolerance, and the stabilised factorisations of those iterations are the ones it saves -- and again at
// convergence if refused; single precision polishes at its convergence (mu ~ 2e-6), where it turns "within sqrt(mu) of the
// optimum" into "the optimum to the accuracy of an fp32 solve".
// The acceptance test of the single-precision polish (-D overrides: the tail measurements behind profiles/r04_f32_acceptance.md).
// Until round 4: rows to 1e-5, multipliers to -1e-3.  At the batch sizes one GPU runs, that let three answers of 65536 learning
// problems through that the fp32 KKT test verified and the fp64 kernel contradicts: 2.4e-3 away with the regression on (a held
// row's multiplier between -1e-3 and -3e-4), 1.2e-3 and 1.0e-3 without it (a multiplier at -7e-5; a row violated by 3.8e-6).
// With rows to 3e-6 and multipliers to -3e-5 the worst of four 32768-batches is 9.0e-4 (one problem; every other below 2.5e-5)
// and the worst IAC problem 8.5e-5; the fp64 pass gets 4 more problems of 32768 and the solve takes the same time.
#ifndef LMPC_F32_POL_FEAS
#define LMPC_F32_POL_FEAS 3e-6f
#endif
#ifndef LMPC_F32_POL_DUAL
#define LMPC_F32_POL_DUAL 3e-5f
#endif
#ifndef LMPC_F32_POL_STEP_TOL
#define LMPC_F32_POL_STEP_TOL 1e-4f
#endif
#ifndef LMPC_F32_POL_STEPS
#define LMPC_F32_POL_STEPS 3
#endif
template <typename real> struct polish_limits;
template <> struct polish_limits<double> {
  static constexpr bool early = true;
  // Round 5: up to FOUR multiplier steps (the loop stops after the second when that one moved the iterate by <= step_ok), a last
  // step of at most 1e-6 (1e-5 until then) and four rounds (three).  A held set that a repair has extended -- the new rows start
  // from a zero multiplier -- or two boundary rows coupled through sigma converge like 0.1 per step, not at once: the second
  // step was still 1.5e-5 .. 1.7e-4, the attempt was refused, and what stood was the interior point's own answer, 9e-6 (tracking,
  // N = 80) and 2e-5 (learning, N = 60) from the dense optimum; an attempt accepted at 1e-5 with that rate is itself 1e-6 off.
  // Measured on the serial twin against the dense optimum over the bench distributions (scratch/r5/cmp_cache.py): worst 2e-7
  // at every horizon, mean iterations -0.3 %, the slowest problem of the N = 20 batch 18 -> 14 iterations.
  // Round 6: up to SIX steps.  The fused factorisation (riccati_factor<.., FUSE>) changes the last bits of every sweep, and one problem of
  // tests/dispatch_sweep.py's 323 584 (BARC tracking, N = 51) fell on the other side of the limit: its exit attempt's steps go 2.1e-5,
  // 5.9e-6, 1.9e-6, 2.2e-7 on the twin (accepted) and ended just above step_tol in the kernel -- refused, and the interior point's own
  // iterate (mu 5e-12, a degenerate problem: 3.4e-6 from the twin in dU) stood with status OPTIMAL.  A consistent set whose steps are
  // still CONVERGING is not a reason to give the optimum up: two more steps cost two sweeps on the few problems that need them (the
  // loop leaves after any step <= step_ok) and nothing on the others.
  // mu_early stays 1e-8.  1e-7 was measured in round 6 (scratch/r6/twin_mu_early.py on the twin, then on the GPU): mean iterations
  // 8.92 -> 8.63 (BARC N = 20), 9.41 -> 9.17 (N = 40), 12.5 -> 12.2 (learning), every answer still 1e-9 from the dense optimum, the
  // pipelined rate +1.3 % -- and the KERNEL slower: 0.813 -> 0.861 ms (N = 20), 2.35 -> 2.48 (N = 40), 8.11 -> 8.65 (N = 80) per 4096:
  // more early attempts are refused, those problems pay the attempt and a second one, and a launch lasts as long as its slowest waves.
  static constexpr double theta = 1e8, feas = 1e-9, dual = 1e-7, mu_early = 1e-8, rd_early = 1e-6, step_ok = 1e-7, step_tol = 1e-6;
  // dual_l: the same test on the SIMPLEX rows' multipliers, a decade tighter (round 6).  The learning problem is LP-like along blends of
  // nearly exchangeable safe-set points: a wrong vertex whose pinned weights have multipliers of -2e-8 .. -1e-7 passed at -1e-7 and sat
  // 1.4e-3 from the dense optimum in X at an objective gap below 1e-9 (the twin, one problem of the 3.9 M of the large dispatch sweeps:
  // N = 71, 160 points; at -1e-8 it is repaired to the optimum).  No problem of the learning fixtures has a multiplier in between: same
  // iterations, same answers (scratch/r6/twin_mu_early.py with TWIN_MACRO=POLISH_DUAL_L).
  static constexpr double dual_l = 1e-8;
  static constexpr int rounds = 4, steps = 6;
};
template <> struct polish_limits<float> {
  // theta: 1e7 needs the stabilised factor and the fp64 2x2 pivot; with 1e5 chains of held input rows (u_i = u_{i-1} + t v_i,
  // stiffness R_d / t^2 per link) converge like 0.7 per step.  Up to three steps: the third removes what the rounding of the
  // first two has left, and is only taken when the second still moved the iterate by more than step_ok (scaled units).
  static constexpr bool early = false;  // (an early attempt at mu ~ 1e-4 was measured on the serial twin: the iterations it saves are fewer than the rounds it adds)
  static constexpr float theta = 1e7f, feas = LMPC_F32_POL_FEAS, dual = LMPC_F32_POL_DUAL, mu_early = 0.0f, rd_early = 0.0f, step_ok = 3e-6f,
                         step_tol = LMPC_F32_POL_STEP_TOL;
  static constexpr float dual_l = LMPC_F32_POL_DUAL;  // (the simplex rows' multipliers: the rows' own limit in single precision)
  static constexpr int rounds = 4, steps = LMPC_F32_POL_STEPS;
};
// 1 / scale of the quantity a slot constrains: the reference's scale vectors (racing_mpc.cpp:36-37, hard-coded there for every
// vehicle) -- used only to measure a polish step
__device__ __forceinline__ float slot_inv_scale(int sl) {
  return sl == 0 ? 5e-4f : (sl == 1 || sl == 10) ? 0.1f : sl == 2 ? 10.0f : sl == 3 ? 0.0125f : (sl == 4 || sl == 5) ? 0.5f : (sl == 6 || sl == 8) ? 0.1f : (1.0f / 0.3f);
}
#define POLISH_THETA_L 1e8  // the simplex rows (always fp64)
#define POLISH_STRONG 1e3   // a row with lam >= POLISH_STRONG t is one the interior point holds firmly
#define POLISH_EXIT 256         // flag in PolishArgs::max_rounds: the attempt at the interior point's exit
#ifndef POLISH_EXIT_GAMMA
#define POLISH_EXIT_GAMMA 1e-2
#endif
// ... holds a row from lam > 1e-2 t on (early and warm attempts: lam > t)
#define WARM_ROUNDS 2       // repairs a warm start may spend before the cold start takes over
static_assert(polish_limits<double>::rounds == LMPC_WARM_ROUNDS_MAX, "lmpc_set_warm_rounds' upper limit is the polish's");
#define WARM_ACT 1e-9       // a box row of the plan counts as active within this slack (a polished plan sits on its bounds to ~1e-16)
#define WARM_ACT_EY 1e-3    // boundary rows: their bounds move with the shift (the track's half-width over one knot's travel)
template <typename real> struct vec2;
template <> struct vec2<double> { typedef double2 type; };
template <> struct vec2<float> { typedef float2 type; };

// Inverse of a symmetric positive definite 6x6 (row-major, full storage) by Cholesky; every index is
// a compile-time constant after unrolling, so the factor lives in registers.  Executed redundantly by
// all lanes on wave-uniform data.
template <typename real>
__device__ __forceinline__ void spd_inv6(const real (&F)[36], real (&Fi)[36]) {
  real Lc[36];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    real d = F[j * 6 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= Lc[j * 6 + k] * Lc[j * 6 + k];
    d = sqrt(d);
    const real id = 1.0 / d;
    Lc[j * 6 + j] = id;  // store the reciprocal of the pivot
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      real t = F[i * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= Lc[i * 6 + k] * Lc[j * 6 + k];
      Lc[i * 6 + j] = t * id;
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    real y[6], x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      real t = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) t -= Lc[i * 6 + k] * y[k];
      y[i] = t * Lc[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      real t = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) t -= Lc[k * 6 + i] * x[k];
      x[i] = t * Lc[i * 6 + i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) Fi[i * 6 + c] = x[i];
  }
}

// LMPC simplex row j: gradient of the (eps-eliminated) terminal cost wrt lambda_j including the row's
// barrier coefficient, bl_j = ss_j - cf_j - u_j'E eps; also returns 1/theta_j.
// (ee = E eps of this iteration, wave-uniform)
template <typename real>
__device__ __forceinline__ real simplex_bl(real lm, real t, real l, real pprod, real ssj, const real (&u)[6],
                                             real smu, real pm, const real (&ee)[6], real& itf) {
  const real it_ = frcp(t);
  const real th = l * it_;
  itf = frcp(th);
  const real cf = th * (-lm + t) + (smu - pm * pprod) * it_;
  real ue = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) ue += u[k] * ee[k];
  return ssj - cf - ue;
}

// The same row in a polish round: barrier coefficient y + theta (-lambda_j) if the row lambda_j >= 0 is held (weight theta),
// none if lambda_j is free (it is then one of the explicit unknowns: 1/theta is not used).
template <typename real>
__device__ __forceinline__ real simplex_bl_polish(real lm, real y, bool held, real ssj, const real (&u)[6], const real (&ee)[6],
                                                    real& itf) {
  itf = held ? real(1.0 / POLISH_THETA_L) : real(0);
  const real cf = held ? y - real(POLISH_THETA_L) * lm : real(0);
  real ue = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) ue += u[k] * ee[k];
  return ssj - cf - ue;
}

// ---- terminal block: two-level elimination of the simplex weights (oracle/c/lmpc_oracle.c documents the derivation) ----
// Points with theta >= tau (B) are eliminated through 1/theta and enter as wave sums (T_B, a_B, s_B); the few points
// whose lambda stays positive have theta -> 0 and are kept as explicit unknowns (A, at most MA_MAX): nothing is ever
// divided by a small theta, and cond(F_B) stays below ~1/TAU_REL.  Everything here is wave-uniform arithmetic on values
// every lane holds; results go to the LDS tail (lane 0 writes), the per-right-hand-side solves read them back as
// broadcast reads.  Unused explicit slots (a >= m) hold u = 0, theta = 1, so they drop out without a branch.
// x <- C_A^-1 x.  Lane `la` (< MA_MAX) brings component la of x in `v` and holds row la of C_A^-1 in `ci`; every lane gets all
// of the result.  (Until round 5 every lane carried the whole Cholesky factor and ran both substitutions itself: with six
// explicit points that is 21 live values and two dependent chains of 21 operations per right-hand side.)
template <typename real>
__device__ __forceinline__ void cinv_apply(const real (&ci)[MA_MAX], real v, real (&x)[MA_MAX]) {
  real s = 0.0;
#pragma unroll
  for (int b = 0; b < MA_MAX; ++b) s += ci[b] * lane_bcast(v, b);
#pragma unroll
  for (int a = 0; a < MA_MAX; ++a) x[a] = lane_bcast(s, a);
}

// F = E^-1 + T_B (full 6x6), a_B, s_B, m explicit points (their u, theta already in T[TL_UA], T[TL_THA]).
// Writes F_B^-1, W_A, C_A^-1, x1, g, a_B, s11 and PT = F^-1 + g g'/s11 to the LDS tail.
// The two Cholesky factors are wave-uniform arithmetic in registers (every lane holds the sums they start from); the
// products in between run one OUTPUT per lane -- a column of F_B^-1 or C_A^-1, an element of W_A, C_A, Z, PT -- on operands
// fetched from LDS in one batch per stage, results to LDS (each cell has one writer), a fence, next stage.  An earlier form
// computed everything in every lane with lane 0 storing: ~300 dependent LDS round trips per call, 26 k cycles per
// iteration at one wave per SIMD.
template <typename real>
__device__ __forceinline__ void term_factor_u(real* T, int lane, const real (&F)[36], const real (&aB)[6], real sB, int m) {
  {  // F_B^-1 by Cholesky: lane c < 6 solves for column c
    real Lf[36];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      real d = F[j * 6 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) d -= Lf[j * 6 + k] * Lf[j * 6 + k];
      const real id = frsqrt(d);
      Lf[j * 6 + j] = id;  // reciprocal pivot
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        real t = F[i * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) t -= Lf[i * 6 + k] * Lf[j * 6 + k];
        Lf[i * 6 + j] = t * id;
      }
    }
    const int c = lane < 6 ? lane : 0;
    real y[6], x[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      real t = (i == c) ? real(1) : real(0);
#pragma unroll
      for (int k = 0; k < i; ++k) t -= Lf[i * 6 + k] * y[k];
      y[i] = t * Lf[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
      real t = y[i];
#pragma unroll
      for (int k = i + 1; k < 6; ++k) t -= Lf[k * 6 + i] * x[k];
      x[i] = t * Lf[i * 6 + i];
    }
    if (lane < 6) {
#pragma unroll
      for (int i = 0; i < 6; ++i) T[TL_FB + i * 6 + lane] = x[i];
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) T[TL_AB + k] = aB[k];
    }
  }
  wave_fence();
  {  // W[a][r] = sum_c F_B^-1[r][c] u_a[c]: lane 6a + r
    const int l = lane < 6 * MA_MAX ? lane : 0, a = (l * 43) >> 8, r = l - 6 * a;
    real fb[6], ua[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      fb[c] = T[TL_FB + r * 6 + c];
      ua[c] = T[TL_UA + a * 6 + c];
    }
    real v = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) v += fb[c] * ua[c];
    if (lane < 6 * MA_MAX) T[TL_WA + lane] = v;
  }
  wave_fence();
  {  // C_A = Theta_A + U_A'W_A, staged through the cells of its inverse: lane 6a + b
    static_assert(MA_MAX == 6, "lane mapping of C_A, stride of its cells");
    const int l = lane < 36 ? lane : 0, a = (l * 43) >> 8, bq = l - 6 * a;
    real ua[6], wb[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      ua[r] = T[TL_UA + a * 6 + r];
      wb[r] = T[TL_WA + bq * 6 + r];
    }
    real v = (a == bq) ? T[TL_THA + a] : real(0);
#pragma unroll
    for (int r = 0; r < 6; ++r) v += ua[r] * wb[r];
    if (lane < 36) T[TL_LC + a * 6 + bq] = v;
  }
  wave_fence();
  {  // C_A^-1 by Cholesky, like F_B^-1: the factor in registers (wave-uniform), lane c < MA_MAX solves for column c.  A jitter for
     // identical points (the padding repeats the last point of the set: C_A is then singular as theta -> 0).  An unused slot
     // (a >= m) has u = 0, theta = 1: its row and column of C_A are those of the identity, and so are its inverse's.
    real Lc[36];
    real jit = 0.0;
#pragma unroll
    for (int a = 0; a < MA_MAX; ++a) {
#pragma unroll
      for (int bq = 0; bq <= a; ++bq) Lc[a * 6 + bq] = T[TL_LC + a * 6 + bq];
      jit += Lc[a * 6 + a];
    }
    jit *= real(sizeof(real) == 4 ? 1e-6 : 1e-13);
#pragma unroll
    for (int a = 0; a < MA_MAX; ++a) {
#pragma unroll
      for (int bq = 0; bq <= a; ++bq) {
        real v = Lc[a * 6 + bq] + (a == bq ? jit : real(0));
#pragma unroll
        for (int k = 0; k < bq; ++k) v -= Lc[a * 6 + k] * Lc[bq * 6 + k];
        Lc[a * 6 + bq] = (a == bq) ? frsqrt(v) : v * Lc[bq * 6 + bq];
      }
    }
    const int c = lane < MA_MAX ? lane : 0;
    real y[MA_MAX], x[MA_MAX];
#pragma unroll
    for (int i = 0; i < MA_MAX; ++i) {
      real t = (i == c) ? real(1) : real(0);
#pragma unroll
      for (int k = 0; k < i; ++k) t -= Lc[i * 6 + k] * y[k];
      y[i] = t * Lc[i * 6 + i];
    }
#pragma unroll
    for (int i = MA_MAX - 1; i >= 0; --i) {
      real t = y[i];
#pragma unroll
      for (int k = i + 1; k < MA_MAX; ++k) t -= Lc[k * 6 + i] * x[k];
      x[i] = t * Lc[i * 6 + i];
    }
    wave_fence();  // (every lane has read C_A before its cells take the inverse)
    if (lane < MA_MAX) {
#pragma unroll
      for (int i = 0; i < MA_MAX; ++i) T[TL_LC + i * 6 + lane] = x[i];
    }
  }
  wave_fence();
  // x1 = C_A^-1 (1_A - W_A'a_B), z1 = a_B + U_A x1, g = F_B^-1 z1, s11 = 1_A'x1 + s_B - a_B'g: a row per lane, the
  // vectors from one product to the next through v_readlane
  real s11 = sB;
  {
    const int la = lane < MA_MAX ? lane : 0, lr = lane < 6 ? lane : 0;
    real wa[6], fb[6], ua[MA_MAX], ci[MA_MAX];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      wa[r] = T[TL_WA + la * 6 + r];
      fb[r] = T[TL_FB + lr * 6 + r];
    }
#pragma unroll
    for (int a = 0; a < MA_MAX; ++a) {
      ua[a] = T[TL_UA + a * 6 + lr];
      ci[a] = T[TL_LC + la * 6 + a];
    }
    real x1[MA_MAX], g[6];
    {
      real v = la < m ? real(1) : real(0);
#pragma unroll
      for (int r = 0; r < 6; ++r) v -= wa[r] * aB[r];
      cinv_apply(ci, v, x1);
    }
    real z1u[6];
    {
      real z = aB[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) z = (lr == k) ? aB[k] : z;
#pragma unroll
      for (int a = 0; a < MA_MAX; ++a) {
        z += ua[a] * x1[a];
        s11 += a < m ? x1[a] : real(0);
      }
#pragma unroll
      for (int c = 0; c < 6; ++c) z1u[c] = lane_bcast(z, c);
    }
    {
      real v = 0.0;
#pragma unroll
      for (int c = 0; c < 6; ++c) v += fb[c] * z1u[c];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        g[r] = lane_bcast(v, r);
        s11 -= aB[r] * g[r];
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) T[TL_G + k] = g[k];
#pragma unroll
      for (int a = 0; a < MA_MAX; ++a) T[TL_X1 + a] = x1[a];
      T[TL_S11] = s11;
    }
  }
  {  // Z = C_A^-1 W_A': lane 6a + c
    const int l = lane < 36 ? lane : 0, a = (l * 43) >> 8, c = l - 6 * a;
    real v = 0.0;
#pragma unroll
    for (int bq = 0; bq < MA_MAX; ++bq) v += T[TL_LC + a * 6 + bq] * T[TL_WA + bq * 6 + c];
    if (lane < 36) T[TL_Z + a * 6 + c] = v;
  }
  wave_fence();
  {  // PT = F_B^-1 - W_A C_A^-1 W_A' + g g'/s11: lane 6r + c
    const real is11 = frcp(s11);
    const int l = lane < 36 ? lane : 0, r = (l * 43) >> 8, c = l - 6 * r;
    real v = T[TL_FB + r * 6 + c] + T[TL_G + r] * T[TL_G + c] * is11;
#pragma unroll
    for (int a = 0; a < MA_MAX; ++a) v -= T[TL_WA + a * 6 + r] * T[TL_Z + a * 6 + c];
    if (lane < 36) T[TL_PT + lane] = v;
  }
  wave_fence();
}

// One right-hand side: beta = U_B Th_B^-1 r_B, sig = 1'Th_B^-1 r_B (wave sums over B), r_A in T[TL_RA], simplex
// residual r1.  Returns h = E U dlambda and nu; writes the explicit points' step to T[TL_XA] (lane 0).
// The operands (rows of W_A, U_A, F_B^-1, C_A^-1, a_B, g) do not depend on the right-hand side: every lane
// fetches the row it works on in ONE batch of LDS reads, the four short products run one output per lane, and what the
// next product needs of the previous one travels through v_readlane (scalar registers), not through LDS -- at one wave
// per SIMD every dependent LDS round trip is ~100 idle cycles, and the all-lanes-compute-everything form of this
// routine had ~100 of them.
template <typename real>
__device__ __forceinline__ void term_solve_u(real* T, int lane, int m, const real (&beta)[6], real sig, real r1, real (&h)[6],
                                             real& nu) {
  const int la = lane < MA_MAX ? lane : 0, lr = lane < 6 ? lane : 0;
  real wa[6], fb[6], ua[MA_MAX], ci[MA_MAX], ab[6], gg[6], x1[MA_MAX];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    wa[r] = T[TL_WA + la * 6 + r];
    fb[r] = T[TL_FB + lr * 6 + r];
    ab[r] = T[TL_AB + r];
    gg[r] = T[TL_G + r];
  }
  const real ra = T[TL_RA + la], s11 = T[TL_S11];
#pragma unroll
  for (int a = 0; a < MA_MAX; ++a) {
    ua[a] = T[TL_UA + a * 6 + lr];
    x1[a] = T[TL_X1 + a];
    ci[a] = T[TL_LC + la * 6 + a];
  }
  real xa[MA_MAX];
  {
    real v = ra;
#pragma unroll
    for (int r = 0; r < 6; ++r) v -= wa[r] * beta[r];
    cinv_apply(ci, v, xa);
  }
  real num = sig - r1;
  real zu[6];
  {
    real z = beta[0];
#pragma unroll
    for (int k = 1; k < 6; ++k) z = (lr == k) ? beta[k] : z;
#pragma unroll
    for (int a = 0; a < MA_MAX; ++a) {
      z += ua[a] * xa[a];
      num += a < m ? xa[a] : real(0);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) zu[c] = lane_bcast(z, c);
  }
  {
    real v = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) v += fb[c] * zu[c];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      h[r] = lane_bcast(v, r);
      num -= ab[r] * h[r];
    }
  }
  nu = num * frcp(s11);
#pragma unroll
  for (int r = 0; r < 6; ++r) h[r] = h[r] - nu * gg[r];
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < MA_MAX; ++a) T[TL_XA + a] = xa[a] - nu * x1[a];
  }
  wave_fence();
}

// Register-resident state of the LMPC simplex rows lambda_j >= 0 (KS safe-set points per lane); empty for
// the tracking kernel so that it costs it nothing.
template <typename real, int KS>
struct SimplexRows {
  bool on[KS];
  int aidx[KS];  // slot of the point among the explicit ones of this iteration, -1: eliminated through 1/theta
  real lm[KS], t[KS], l[KS], p[KS], j[KS], dl[KS];
  real sv[KS];  // lambda at the start of a polish (restored when it is refused)
  const real* ul;  // the (centred) points in LDS, point-major: component k of point j at ul[6 j + k], S points -- read-only
                   // after the load, 36 registers (KS = 3) the iteration's row state needs more.  A point is 48 bytes = three
                   // 16-byte reads; 16 consecutive lanes at a 48-byte stride cover all 64 banks once, so the reads are
                   // conflict-free.  A lane slot past S reads the zero point stored behind the last one (uz[q] = its index).
  int uz[KS];      // 6 * (index of the point this lane's slot q reads)
  __device__ __forceinline__ void load_u(int q, int lane, real (&u)[6]) const {
    const real* p = ul + uz[q];
#pragma unroll
    for (int k = 0; k < 6; ++k) u[k] = p[k];
  }
  real ss0[6];
  real r1;   // 1 - 1'lambda
  real tau;  // theta below which a point is kept explicit
  int m;     // explicit points of this iteration
};
template <typename real>
struct SimplexRows<real, 0> {};

template <typename real>
struct Lds {
  real* base;
  int N;
  int stride;  // of a stage record: LMPC_STAGE_STRIDE, or LMPC_LEAN_STAGE_STRIDE in the lean layout (below)
  bool fresh;  // FRESH_LANE in the sweeps of this instantiation (a compile-time constant where the sweeps are inlined)
  bool chain_prio;  // CHAIN_PRIO around the serial stage chains (likewise)
  __device__ __forceinline__ real* st(int i) const { return base + i * stride; }
  __device__ __forceinline__ real* kn(int i) const { return base + (N - 1) * stride + i * LMPC_KNOT_STRIDE; }
  __device__ __forceinline__ real* tail() const { return base + (N - 1) * stride + N * LMPC_KNOT_STRIDE; }
};

// ---- the LEAN layout (fp64, N > 40) -----------------------------------------------------------------------------------
// At long horizons the stage records are what limits residency: 78 doubles per stage, 48 of them the stage model [A B],
// which the iteration only reads.  The lean layout keeps the model OUT of the records: every sweep streams it from the
// linearisation workspace (L2 / MALL resident, written by lmpc_linearize_kernel just before) through two chunk buffers of
// LN_CHUNK stages each, filled by asynchronous global -> LDS copies (global_load_lds_dwordx4: no registers, no ds_write)
// one chunk ahead of the sweep.  A stage record shrinks to what the factorisation produces:
//     K_j[c] @ 2 c + j (c < 8) | Hinv (h00, h01) @16, h11 @18 | dt @19 | kff rhs0 [2] @20 | kff rhs1 [2] @22        (24)
// and a chunk slot holds the workspace record as it is: ABt[8][6] (row c of it = column c of [A B]: the rows the backward
// sweeps read are contiguous 48-byte runs on distinct banks, the columns the forward sweep reads are stride-6) | g [6].
// N = 60: 57 KB -> 38 KB per problem, 2 -> 4 resident problems per CU (the register file allows no more); N = 80: 2 -> 3.
#define LN_CHUNK 8
#define LN_REC LMPC_LIN_RECORD
#define LN_HI 16
#define LN_HI11 18
#define LN_DT 19
#define LN_KFF(s) ((s) ? 22 : 20)
#ifndef LMPC_LEAN_MIN_KQ  // (A/B switch: 99 builds every kernel on the fat layout; lmpc_device.h's lmpc_is_lean follows it)
#define LMPC_LEAN_MIN_KQ 11
#endif
constexpr bool lmpc_lean(int real_bytes, int kq) { return real_bytes == 8 && kq >= LMPC_LEAN_MIN_KQ; }
template <typename real>
struct ModelStream {
  const real* ws;  // this problem's [N - 1][LN_REC] in the workspace (HBM / L2)
  real* buf;       // LDS: 2 chunks of LN_CHUNK records
  int NS, lane;
  int have0 = -1, have1 = -1;  // the chunk each buffer holds (or is being filled with): a sweep that finds its first chunks
                               // resident -- the forward sweep after a backward one, the factorisation after a forward sweep --
                               // starts without a fetch
  __device__ __forceinline__ void ensure(int ch) {
    if ((ch & 1 ? have1 : have0) != ch) fetch(ch);
  }
  // chunk ch -> buffer (ch & 1), asynchronously: 16 bytes per lane per instruction, the wave's lanes in address order.
  // ONE per-lane address and one LDS base for the whole chunk, the instruction's immediate offset (applied to both sides)
  // steps through it: with an address per instruction the compiler kept seven 64-bit addresses per call site alive,
  // spilled them, and every copy then waited (vmcnt counts the reload and the copies alike) for the one before it.
  __device__ __forceinline__ void fetch(int ch) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (no LDS read of the buffer about to be overwritten is still in flight)
    if (ch & 1) have1 = ch; else have0 = ch;
    const int lo = ch * LN_CHUNK * LN_REC;
    const int n = min(LN_CHUNK * LN_REC, NS * LN_REC - lo);
    real* const dst = buf + (ch & 1) * LN_CHUNK * LN_REC;
    // (the per-lane source address from a lane number the optimiser cannot see through: hoisted out of the sweeps as the loop
    //  invariant it is, the 64-bit address was spilled, and each of the four copies below then waited -- vmcnt(0): for its own
    //  reload AND the copy before it -- four memory round trips in a row per chunk)
    int fl = lane;
    asm volatile("" : "+v"(fl));
    const real* const src = ws + lo + 2 * fl;
    const int e = 2 * fl;
    static_assert(sizeof(real) == 8 && (LN_CHUNK * LN_REC + 127) / 128 == 4, "four 1 KB slices per chunk");
#define LMPC_GLDS(J)                                                                                                   \
  if (e + 128 * J < n)                                                                                                 \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,                               \
                                     (__attribute__((address_space(3))) void*)dst, 16, 1024 * J, 0);
    LMPC_GLDS(0) LMPC_GLDS(1) LMPC_GLDS(2) LMPC_GLDS(3)
#undef LMPC_GLDS
  }
  // every copy issued so far has landed (vmcnt counts them), and no later LDS read moves ahead of this point
#ifdef LMPC_PHASE_TIMING
  mutable long long waited = 0;  // cycles spent in wait() (profiling build: reported as phase 5)
#endif
  __device__ __forceinline__ void wait() const {
#ifdef LMPC_PHASE_TIMING
    const long long t0 = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wave_fence();
#ifdef LMPC_PHASE_TIMING
    waited += __builtin_readcyclecounter() - t0;
#endif
  }
  __device__ __forceinline__ const real* stage(int i) const {  // (i is wave-uniform: keep the slot arithmetic on the scalar unit)
    const int off = __builtin_amdgcn_readfirstlane(((i / LN_CHUNK) & 1) * LN_CHUNK * LN_REC + (i % LN_CHUNK) * LN_REC);
    return buf + off;
  }
};

// true cost Hessian entry on z_i (no barrier terms), racing_mpc.cpp:459-476; terminal knot or a knot 1 <= i <= N-2
template <typename real>
__device__ __forceinline__ real qz_entry(const real* ct, bool terminal, int r, int c) {
  const int rx = r < 6 ? r : 0, ru = r >= 6 ? r - 6 : 0, cu = c >= 6 ? c - 6 : 0;
  const real dx = terminal ? ct[CT_QT + rx] : ct[CT_QD + rx];
  const real uu = ct[CT_QU + ru * 2 + cu];
  return (r < 6 || c < 6) ? ((r == c) ? dx : 0.0) : uu;
}

// Pin the issue order of LDS traffic: one wave's DS instructions return in issue order, so the read a
// serial chain waits for must be queued ahead of the operand prefetch of the following stage.
#define ISSUE_ORDER() __builtin_amdgcn_sched_barrier(0)
// The serial stage chains (factorisation, sweeps) at a higher issue priority than the row phases of the wave they share a SIMD
// with (s_setprio): the chain's next instruction is the one a solve waits for, the row phases are throughput work that fills in.
// Two-waves-per-SIMD kernels only (alone on its SIMD a wave has nobody to yield to: +0.5 %): headline kernel -1.5 %, pipelined
// +1.4 %, one batch at a time -1.9 %, the mixed learning kernel -1.1 %; same bits (profiles/r04_row_phases.md).
#define LMPC_CHAIN_PRIO 3
// (the learning problem's terminal elimination is another serial chain; raising its priority the same way was measured in round 4 and
//  bought nothing: profiles/r04_row_phases.md)
#define CHAIN_PRIO_ENTER() do { if (LMPC_CHAIN_PRIO && L.chain_prio) __builtin_amdgcn_s_setprio(LMPC_CHAIN_PRIO); } while (0)
#define CHAIN_PRIO_LEAVE() do { if (LMPC_CHAIN_PRIO && L.chain_prio) __builtin_amdgcn_s_setprio(0); } while (0)
// ... and the other way round: value x is complete before any later memory operation is issued (an
// empty asm that consumes x and clobbers memory), used to keep a prefetch behind the last use of the
// registers it overwrites.
#define AFTER_VALUE(x) asm volatile("" : "+v"(x) : : "memory")
// The lane number as a value the optimiser cannot see through: everything a sweep derives from it (row / column indices,
// LDS addresses, 0/1 multipliers, predicates) is then computed where the sweep starts -- a dozen VALU instructions -- instead
// of once at the top of the kernel and kept alive over the whole iteration, which at the register limit means spilled and
// reloaded from scratch (or from VGPR lanes, for the predicates) inside the sweep's preamble, one wait per reload.  Per
// instantiation (Lds::fresh, lmpc_fresh_lane below): what it does to the register allocation of a 3000-line kernel is not
// monotone, and it is kept only where it was measured to pay.  (`site` numbers the sweep functions: round 4 bisected a miscompute
// with a per-site mask, profiles/r04_d70_bisect.md.)
#define FRESH_LANE(l, site) do { if (L.fresh) asm volatile("" : "+v"(l)); } while (0)

// Lane K of every 16-lane row to the whole row (DPP row_newbcast, gfx90a+; v_mov_b64_dpp for doubles): a register-to-
// register broadcast.  bound_ctrl:1 with full row / bank masks tells the compiler that the tied "old" operand is never
// read, so it is not materialised (with bound_ctrl:0 every broadcast costs a v_mov of a constant first -- a fifth of the
// sweeps' VALU instructions).  (An inline-asm form of the same instructions measured the same speed and was NOT safe: in
// the most register-starved instantiation, KQ = 14 with KS = 3, it gave wrong and run-to-run different results that wider
// wait states did not cure, while this builtin form is bitwise reproducible there -- the compiler has to see DPP.)
template <int K>
__device__ __forceinline__ double row_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true); }
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, true));
}
template <typename real>
__device__ __forceinline__ void row_bcast6(real v, real (&o)[6]) {
  o[0] = row_bcast<0>(v); o[1] = row_bcast<1>(v); o[2] = row_bcast<2>(v);
  o[3] = row_bcast<3>(v); o[4] = row_bcast<4>(v); o[5] = row_bcast<5>(v);
}
template <typename real>
__device__ __forceinline__ void row_bcast8(real v, real (&o)[8]) {
  o[0] = row_bcast<0>(v); o[1] = row_bcast<1>(v); o[2] = row_bcast<2>(v); o[3] = row_bcast<3>(v);
  o[4] = row_bcast<4>(v); o[5] = row_bcast<5>(v); o[6] = row_bcast<6>(v); o[7] = row_bcast<7>(v);
}
template <typename real>
__device__ __forceinline__ void row_bcast67(real v, real& a, real& b) {  // lanes 6 and 7
  a = row_bcast<6>(v);
  b = row_bcast<7>(v);
}

// Backward Riccati sweep for the barrier weights currently in the knots' rhs0 region
// (Thz @ +10..17, Thv @ +18,19, boundary weight @ KN_EY).  Leaves K (columns 6,7 of M) and Hinv in
// the stage records.
//
// The sweep is one dependent chain over the knots, so it is written for latency: lane (r, c) keeps its
// own element of P / W in a register, everything that does not depend on the chain (model rows, barrier
// weights) is fetched one stage ahead, and the three exchanges per stage (P, W, Y rows through LDS) are
// the only waits; the 2x2 inverse starts from v_readlane copies of Y_uu while the Y rows are in flight.
//
// JOSEPH selects the stabilised form of the cost-to-go update, P <- Qz + Thz + Phi' P Phi + K' (Sv + Thv) K with
// Phi = Abar - Bbar K: the same matrix in exact arithmetic, but a sum of positive semidefinite products.  The plain form
// Y - G'K subtracts two numbers of the size of the largest barrier weight (1e10 .. 1e13 late in the iteration) to leave
// one of the size of the cost, and the Newton directions lose those digits (the former "mu ~ 1e-11 floor"); here the
// cancellation happens inside Phi, before the multiplication by P.  It costs a second pair of 8x8 products (8-term,
// Phi has no identity block), so the iteration uses it only once mu <= JOSEPH_MU: about two factorisations per solve.
#define JOSEPH_MU 1e-8
//
// FUSE (round 6): the factorisation carries the BACKWARD sweep of the predictor's two right-hand sides along.  Both recursions run
// from the last knot to the first, the sweep's stage i needs nothing but what the factor's stage i holds in registers at its end --
// K_i[:, c], H_i^-1, dt_i, column c of [A B]_i (lane (r, c) of the factor IS lane (s, c) of the sweep: s = DPP row parity) -- and the
// right-hand side, and its ~30 instructions sit in the shadows of the factor's LDS exchanges: an iteration is four dependent chains
// over the horizon instead of five.  The right-hand side must then exist BEFORE the factorisation: the iteration assembles the
// predictor's gradient first (it depends on the iterate only) into the rhs0 cells, and the barrier weights the factor reads move
// to the rhs1 cells (`th_off` = KN_R1, the boundary row's weight to `tey_off` = KN_TEY) -- the Schur vector that used to sit there
// is one number per knot (c_sigma on e_y: KN_CSIG), generated on the fly.  The arithmetic of either recursion is the unfused one's,
// operation for operation.  riccati_solve<2, true> then runs the forward half only.
#define KN_TEY 35  // (the knot record's spare cell)
template <bool HAS_PT, bool JOSEPH, bool FUSE = false, typename real, typename ptreal>
__device__ __forceinline__ void riccati_factor(const Lds<real>& L, int lane, const ptreal* PT, const int th_off = KN_R0, const int tey_off = KN_EY) {
  FRESH_LANE(lane, 0);
  CHAIN_PRIO_ENTER();
  const int N = L.N, r = lane >> 3, c = lane & 7;
  real* T = L.tail();
  real* MP = T + TL_P;
  real* MW = T + TL_W;
  real* MY = T + TL_Y;
  const real* ct = T + TL_CT;
  const bool diag = r == c;
  // per-lane 0/1 multipliers instead of selects (exact: the products are the operand or zero)
  const real m_diag = diag ? real(1) : real(0), m_r1 = r == 1 ? real(1) : real(0);
  const real m_r6 = r >= 6 ? real(1) : real(0), m_c6 = c >= 6 ? real(1) : real(0);
  const real qmid = qz_entry(ct, false, r, c);
  real pown;
  {
    const real* kn = L.kn(N - 1);
    real e = qz_entry(ct, true, r, c);
    const real th = kn[th_off + r] + (r == 1 ? kn[tey_off] : real(0));
    if (diag) e += th;
    // LMPC: safe-set block condensed onto x_T; its upper triangle is the block (see the symmetry note in the loop)
    if (HAS_PT && r < 6 && c < 6) e += real(PT[r <= c ? r * 6 + c : c * 6 + r]);
    pown = e;
    MP[MROWS(r) + c] = e;
  }
  // fused backward sweep: lane (s, c), s = parity of the DPP row; rhs0 = the knots' rhs0 cells, rhs1 = c_sigma e_y (knots >= 1)
  const int fs = (lane >> 4) & 1;
  const bool f_own = FUSE && (lane & 8) == 0 && lane < 32 && c < 2;
  const int f_oq = fs == 0 ? KN_R0 + c : KN_CSIG;                       // where the sweep's q_z comes from ...
  const real f_mq = (fs == 0 || c == 1) ? real(1) : real(0);            // ... and whether it counts
  const real f_m6 = (c >= 6) ? real(1) : real(0);
  real fp = 0.0, fws = 0.0, fw6 = 0.0, fw7 = 0.0, fqz = 0.0, fqv0 = 0.0, fqv1 = 0.0;
  if constexpr (FUSE) fp = f_mq * L.kn(N - 1)[f_oq];
  // The upper triangle (lanes r <= c) is the cost-to-go; those lanes store their element at (r, c) AND (c, r), the others
  // store to a dead cell.  Why: the two computed halves differ by rounding, and that antisymmetric part is not contracted by
  // the recursion -- it is multiplied by Abar'(.)Abar, the OPEN-loop dynamics, whose RK4 map has |eig| up to ~15-25 below
  // 1 m/s, so within twenty stages it reaches 1e17 and the Newton directions are noise (every low-speed cold start at
  // N >= 40 was lost to this).  An exactly symmetric P only carries symmetric error, which the closed loop damps.
  const bool upper = r <= c;
  real* const p_dst0 = upper ? MP + MROWS(r) + c : MW + MROWS(r) + c;
  real* const p_dst1 = upper ? MP + MROWS(c) + r : MW + MROWS(r) + c;
  // where this lane's share of the stage results goes: lanes 0..15 K_j[c] (j = r), 16..18 Hinv, the
  // rest to their own (dead) W cell
  const int res_off = lane < 16 ? ST_ROW(c) + 6 + r : (lane == 18 ? ST_HI11 : ST_HI + (lane - 16));
  const bool res_on = lane < 19;
  real* const res_junk = MW + MROWS(r) + c;
  // phase-1/2 operands of stage N-2 (later stages: fetched during phase 3 of the stage before)
  real ar[6], ac[6];
  {
    const real* st = L.st(N - 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      ar[k] = st[ST_ROW(r) + k];
      ac[k] = st[ST_ROW(c) + k];
    }
  }
  // the input-rate weights are the same at every stage: scalar registers, not an LDS read per stage
  const real sv00 = uni(ct[CT_SV + 0]), sv01 = uni(ct[CT_SV + 1]), sv11 = uni(ct[CT_SV + 3]);
  wave_sync();
  real pr[6];
  if constexpr (FUSE) {  // (fused: the P row of a stage is requested at the end of the stage before, ahead of the sweep's share of it)
#pragma unroll
    for (int k = 0; k < 6; ++k) pr[k] = MP[MROWS(c) + k];
    pown = MP[MROWS(r) + c];
    ISSUE_ORDER();
  }
  for (int i = N - 2; i >= 0; --i) {
    real* st = L.st(i);
    const real* kn = L.kn(i);
    // W = Abar' P : W[r][c] = sum_k Abar[k][r] P[k][c]  (+ P[r][c] for the u rows); P[k][c] read as P[c][k]
    if constexpr (!FUSE) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pr[k] = MP[MROWS(c) + k];
      pown = MP[MROWS(r) + c];  // the symmetrised element (lanes below the diagonal did not compute it)
      ISSUE_ORDER();
    }
    // phase-3 operands of this stage, queued behind the P row
    const real t = st[ST_DT], thr = kn[th_off + r], ey = kn[tey_off], thv0 = kn[th_off + 8], thv1 = kn[th_off + 9];
    if constexpr (FUSE) {
      fqz = kn[f_oq];
      fqv0 = kn[KN_R0 + 8];
      fqv1 = kn[KN_R0 + 9];
    }
    ISSUE_ORDER();
    real w = m_r6 * pown;
#pragma unroll
    for (int k = 0; k < 6; ++k) w += ar[k] * pr[k];
    MW[MROWS(r) + c] = w;
    wave_sync();
    // Y = W Abar
    real wr[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) wr[k] = MW[MROWS(r) + k];
    if constexpr (FUSE) {  // the sweep's w = Abar' p (+ p_u on the input rows), while the W row is on its way; ac = [A B](:, c)
      ISSUE_ORDER();
      real pb[6];
      row_bcast6(fp, pb);
      fws = f_m6 * fp;
#pragma unroll
      for (int k = 0; k < 6; ++k) fws = rfma(ac[k], pb[k], fws);
      row_bcast67(fws, fw6, fw7);
    }
    real y = m_c6 * w;
#pragma unroll
    for (int k = 0; k < 6; ++k) y += wr[k] * ac[k];
    MY[MROWS(r) + c] = y;
    wave_sync();
    // H = Sv + Thv + t^2 Y_uu, K = H^-1 t Y[6:8,:], P <- Qz + Thz + Y - t^2 Y[6:8,r]' H^-1 Y[6:8,c]
    const real y6r = MY[MROWS(6) + r], y7r = MY[MROWS(7) + r];
    const real y6c = MY[MROWS(6) + c], y7c = MY[MROWS(7) + c];
    AFTER_VALUE(y);
    if constexpr (!JOSEPH) {  // phase-1/2 operands of the next stage (their registers are dead by now), queued behind the Y rows
      const real* stn = L.st(i > 0 ? i - 1 : 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        ar[k] = stn[ST_ROW(r) + k];
        ac[k] = stn[ST_ROW(c) + k];
      }
    }
    ISSUE_ORDER();
    const real y66 = lane_bcast(y, 54), y67 = lane_bcast(y, 55), y77 = lane_bcast(y, 63);
    const real tt = t * t;
    const real h00 = sv00 + thv0 + tt * y66;
    const real h01 = sv01 + tt * y67;
    const real h11 = sv11 + thv1 + tt * y77;
    real idet;
    if constexpr (sizeof(real) == 4) {
      // H = R + t^2 Y_uu is rank-one dominated once a state row is stiff: its determinant is the difference of two
      // products that agree to six digits, which single precision cannot form -- a handful of fp64 operations per stage
      const double det = (double)h00 * (double)h11 - (double)h01 * (double)h01;
      idet = (real)(1.0 / det);
    } else {
      idet = frcp(h00 * h11 - h01 * h01);
    }
    const real hi00 = h11 * idet, hi01 = -h01 * idet, hi11 = h00 * idet;
    const real g0 = t * y6c, g1 = t * y7c;
    const real k0c = hi00 * g0 + hi01 * g1;
    const real k1c = hi01 * g0 + hi11 * g1;
    real pn;
    if constexpr (JOSEPH) {
      // the gain at this lane's ROW index as well: K[:, r] from Y[6:8, r], the same expression lane (., r) evaluates
      const real g0r = t * y6r, g1r = t * y7r;
      const real k0r = hi00 * g0r + hi01 * g1r;
      const real k1r = hi01 * g0r + hi11 * g1r;
      // columns r and c of Phi (overwriting the columns of Abar they are made from); B = Abar[:, 6:8] as broadcast reads
      {
        real b0[6], b1[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          b0[k] = st[ST_ROW(6) + k];
          b1[k] = st[ST_ROW(7) + k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          ar[k] -= t * (b0[k] * k0r + b1[k] * k1r);
          ac[k] -= t * (b0[k] * k0c + b1[k] * k1c);
        }
      }
      const real fr6 = (r == 6 ? real(1) : real(0)) - t * k0r, fr7 = (r == 7 ? real(1) : real(0)) - t * k1r;
      const real fc6 = (c == 6 ? real(1) : real(0)) - t * k0c, fc7 = (c == 7 ? real(1) : real(0)) - t * k1c;
      // W2 = Phi' P (P symmetric: P[k][c] read as P[c][k]), exchanged through the W matrix (its phase-2 reads are done)
      real w2 = 0.0;
      {
        real pc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pc[k] = MP[MROWS(c) + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) w2 += ar[k] * pc[k];
        w2 += fr6 * pc[6] + fr7 * pc[7];
      }
      MW[MROWS(r) + c] = w2;
      wave_sync();
      real y2 = 0.0;
      {
        real w2r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w2r[k] = MW[MROWS(r) + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) y2 += w2r[k] * ac[k];
        y2 += w2r[6] * fc6 + w2r[7] * fc7;
      }
      {  // operands of the next stage
        const real* stn = L.st(i > 0 ? i - 1 : 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          ar[k] = stn[ST_ROW(r) + k];
          ac[k] = stn[ST_ROW(c) + k];
        }
      }
      const real v00 = sv00 + thv0, v11 = sv11 + thv1;
      pn = qmid + y2 + k0r * (v00 * k0c + sv01 * k1c) + k1r * (sv01 * k0c + v11 * k1c);
    } else {
      pn = qmid + y - t * (y6r * k0c + y7r * k1c);
    }
    const real th = rfma(m_r1, ey, thr);
    pn = rfma(m_diag, th, pn);
    *p_dst0 = pn;  // (not used after stage 0)
    *p_dst1 = pn;
    const real res = lane < 8 ? k0c : (lane < 16 ? k1c : (lane == 16 ? hi00 : (lane == 17 ? hi01 : hi11)));
    *(res_on ? st + res_off : res_junk) = res;
    wave_sync();
    if constexpr (FUSE) {
      if (i > 0) {  // the next stage's P row, behind the stores above
#pragma unroll
        for (int k = 0; k < 6; ++k) pr[k] = MP[MROWS(c) + k];
        pown = MP[MROWS(r) + c];
      }
      ISSUE_ORDER();
      // the sweep's stage i: p_i = q_i + Abar' p_{i+1} - K' (q_v + Bbar' p_{i+1}),  kff_i = H^-1 (q_v + Bbar' p_{i+1})
      const real qv0 = fs == 0 ? fqv0 : real(0), qv1 = fs == 0 ? fqv1 : real(0);
      const real hv0 = rfma(t, fw6, qv0);
      const real hv1 = rfma(t, fw7, qv1);
      fp = f_mq * fqz + fws - (k0c * hv0 + k1c * hv1);
      const real ha = c == 0 ? hi00 : hi01, hb = c == 0 ? hi01 : hi11;
      const real kff = ha * hv0 + hb * hv1;
      *(f_own ? st + ST_KFF(fs) + c : res_junk) = kff;
    }
  }
  if constexpr (FUSE) wave_sync();
  CHAIN_PRIO_LEAVE();
}

// Riccati vector solve for NRHS (1 or 2) right-hand sides held in the knots' rhs regions
// (q_z @ +0..7, q_v @ +8,9 of region s); the step (dz, dv) overwrites them.  dz_0 = 0.
//
// Both sweeps are strictly serial over the knots: one dependent chain of N - 1 stages each, and the chain's length, not
// the instruction count, is what a solve waits for.  Lane (s, r) = ((lane >> 4) % NRHS, lane & 7): each right-hand side
// has a 16-lane DPP row of its own (lanes 8..15 of a row, and the rows above NRHS, mirror lanes below and write to dead
// cells), the running vector lives in ONE register per lane, and a stage gets the other components by row_newbcast --
// nothing on the chain goes through LDS (it did until round 2: one write + broadcast reads per stage, and a ds_swizzle
// for the condensed 2-vector; ~400 cycles per stage under load).  Stage operands are fetched one stage ahead; the
// results a stage leaves behind (kff, dz, dv) are stored off the chain.
template <int NRHS, bool FWD_ONLY = false, typename real>  // (FWD_ONLY: the backward half ran inside the factorisation, riccati_factor<.., FUSE>)
__device__ __forceinline__ void riccati_solve(const Lds<real>& L, int lane, Prof& pf) {
  FRESH_LANE(lane, 2);
  CHAIN_PRIO_ENTER();
  const int N = L.N;
  const int r = lane & 7, s = (lane >> 4) & (NRHS - 1);
  const bool own = (lane & 8) == 0 && lane < 16 * NRHS;
  const int reg = KN_R0 + 10 * s;
  real* T = L.tail();
  real* const junk0 = T + TL_W + lane;  // 64 + 64 dead cells: W (80) and Y (80) are contiguous
  real* const junk1 = T + TL_W + 80 + lane;
  const real m6 = (r >= 6) ? real(1) : real(0);  // rows 6, 7 (the input rows of z) take an extra term: a multiplier, not a select
  // kff = H^-1 hv: lane r = 0 takes row (h00, h01), the others row (h01, h11) -- picked by the address, not by a select
  const int o_ha = r == 0 ? ST_HI : ST_HI + 1, o_hb = r == 0 ? ST_HI + 1 : ST_HI11;
  // ---- backward: p_i = q_i + Abar' p_{i+1} - K' (q_v + Bbar' p_{i+1});  kff_i = H^-1 (q_v + Bbar' p_{i+1})
  real p = FWD_ONLY ? real(0) : L.kn(N - 1)[reg + r];
  real row[6];  // [A B](:, r), fetched one stage ahead
  if constexpr (!FWD_ONLY) {
    const real* st = L.st(N - 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) row[k] = st[ST_ROW(r) + k];
  }
  for (int i = FWD_ONLY ? -1 : N - 2; i >= 0; --i) {
    real* st = L.st(i);
    const real* kn = L.kn(i);
    const real k0r = st[ST_ROW(r) + 6], k1r = st[ST_ROW(r) + 7], t = st[ST_DT];
    const real qz = kn[reg + r], qv0 = kn[reg + 8], qv1 = kn[reg + 9];
    const real ha = st[o_ha], hb = st[o_hb];
    real pb[6];
    row_bcast6(p, pb);
    real w = m6 * p;  // w = Abar' p: rows 6,7 also take p_u
#pragma unroll
    for (int k = 0; k < 6; ++k) w = rfma(row[k], pb[k], w);
    real w6, w7;
    row_bcast67(w, w6, w7);
    {
      const real* stn = L.st(i > 0 ? i - 1 : 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) row[k] = stn[ST_ROW(r) + k];
    }
    const real hv0 = rfma(t, w6, qv0);
    const real hv1 = rfma(t, w7, qv1);
    p = qz + w - (k0r * hv0 + k1r * hv1);  // (not used after stage 0)
    const real kff = ha * hv0 + hb * hv1;
    *((own && r < 2) ? st + ST_KFF(s) + r : junk1) = kff;
  }
  wave_sync();
  PT_MARK(8 + NRHS - 1)
  // ---- forward: dv_i = -kff_i - K dz_i,  dz_{i+1} = Abar dz_i + Bbar dv_i
  // lanes r < 6 take a state row, lanes 6, 7 the two rows of K: column k of M is [A B](:, k) | K(:, k)
  *(own ? L.kn(0) + reg + r : junk0) = 0.0;
  real d = 0.0;     // component r of dz_i
  real col[8], a0;  // M(r, :) and the feed-forward term, fetched one stage ahead
  {
    const real* st = L.st(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) col[k] = st[ST_ROW(k) + r];
    a0 = st[ST_KFF(s) + (r & 1)];
  }
  for (int i = 0; i < N - 1; ++i) {
    const real* st = L.st(i);
    real* kn = L.kn(i);
    const real b0 = st[ST_ROW(6) + r], b1 = st[ST_ROW(7) + r], t = st[ST_DT];
    real dz[8];
    row_bcast8(d, dz);
    real acc = m6 * a0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc = rfma(col[k], dz[k], acc);
    const real ax = acc;  // state rows: A dz_x
    acc = rfma(col[6], dz[6], acc);
    acc = rfma(col[7], dz[7], acc);
    const real dv = -acc;  // lanes 6, 7
    const real du = rfma(t, dv, d);  // (lanes 6, 7: their own component of dz is u_{i-1})
    real du0, du1;
    row_bcast67(du, du0, du1);
    {
      const real* stn = L.st(i < N - 2 ? i + 1 : i);
#pragma unroll
      for (int k = 0; k < 8; ++k) col[k] = stn[ST_ROW(k) + r];
      a0 = stn[ST_KFF(s) + (r & 1)];
    }
    const real nx = rfma(b1, du1, rfma(b0, du0, ax));
    d = (r < 6) ? nx : du;
    *(own ? kn + LMPC_KNOT_STRIDE + reg + r : junk0) = d;
    *((own && r >= 6) ? kn + reg + 2 + r : junk1) = dv;
  }
  wave_sync();
  CHAIN_PRIO_LEAVE();
  PT_MARK(10 + NRHS - 1)
}

// Closed-loop rollout z_{i+1} = Abar z_i + Bbar v_i + gbar, v_i = -K_i z_i (absolute variables).
// The linearised model can be open-loop unstable (|eig A| > 1 at low speed with dt = 25 ms), so
// the start trajectory is generated under the stabilising Riccati feedback.  Same lane roles as the
// forward sweep of riccati_solve.
// WARM (lmpc_solve_batch_warm): the rollout TRACKS a plan -- v_i = v_i^plan - K_i (z_i - z_i^plan) -- whose knots sit in the rhs0
// cells of the knot records ([z^plan (8) | v^plan (2)] at KN_R0: free between the factorisation that has read its weights
// there and the first gradient): the plan made dynamically exact about this linearisation, a few 1e-3 from where it was.
template <bool WARM = false, typename real>
__device__ __forceinline__ void feedback_rollout(const Lds<real>& L, int lane) {
  FRESH_LANE(lane, 4);
  const int N = L.N, r = lane & 7;
  const bool own = lane < 8;
  real* T = L.tail();
  real* const junk0 = T + TL_W + lane;
  real* const junk1 = T + TL_W + 80 + lane;
  for (int i = 0; i < N - 1; ++i) {
    const real* st = L.st(i);
    real* kn = L.kn(i);
    real dz[8], col[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dz[k] = kn[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) col[k] = st[ST_ROW(k) + r];
    const real t = st[ST_DT];
    const real g = st[ST_G + (r < 6 ? r : 0)];
    real acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc = rfma(col[k], dz[k], acc);
    const real ax = acc;
    acc = rfma(col[6], dz[6], acc);
    acc = rfma(col[7], dz[7], acc);
    real v = -acc;
    if constexpr (WARM) {  // (lanes 6, 7: col = a row of K)  v = v^plan + K z^plan - K z
      real kz = kn[KN_R0 + 8 + (r & 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k) kz = rfma(col[k], kn[KN_R0 + k], kz);
      v += kz;
    }
    const real u = rfma(t, v, (r == 6) ? dz[6] : dz[7]);
    const real u0 = lane_bcast(u, 6), u1 = lane_bcast(u, 7);
    const real nx = g + rfma(col[7], u1, rfma(col[6], u0, ax));
    *(own ? kn + LMPC_KNOT_STRIDE + r : junk0) = (r < 6) ? nx : u;
    *((own && r >= 6) ? kn + 2 + r : junk1) = v;
    wave_sync();
  }
}

// ---- the sweeps of the lean layout: the same arithmetic, stage by stage, with the stage model read from the chunk slots
// ---- of a ModelStream and K / Hinv / kff in the 24-cell records -------------------------------------------------------
// Streaming discipline of a sweep that walks the stages downwards (factor, backward vector sweep): the chunk it starts in
// is fetched and waited for, the one below it is fetched at once; the stage that reads ahead into the next chunk waits for
// it first and, the chunk just left being dead by then, fetches the one after into its buffer.  Upwards (forward sweep,
// rollout) the mirror image.  One fetch is in flight at a time; a chunk is 8 stages of work ahead of its first use.
template <bool HAS_PT, bool JOSEPH, bool FUSE = false, typename real, typename ptreal>
__device__ __forceinline__ void riccati_factor_lean(const Lds<real>& L, ModelStream<real>& M, int lane, const ptreal* PT, const int th_off = KN_R0,
                                                    const int tey_off = KN_EY) {
  FRESH_LANE(lane, 1);
  const int N = L.N, r = lane >> 3, c = lane & 7;
  real* T = L.tail();
  real* MP = T + TL_P;
  real* MW = T + TL_W;
  real* MY = T + TL_Y;
  const real* ct = T + TL_CT;
  const bool diag = r == c;
  const real m_diag = diag ? real(1) : real(0), m_r1 = r == 1 ? real(1) : real(0);
  const real m_r6 = r >= 6 ? real(1) : real(0), m_c6 = c >= 6 ? real(1) : real(0);
  const real qmid = qz_entry(ct, false, r, c);
  real pown;
  {
    const int ch = (N - 2) / LN_CHUNK;
    M.ensure(ch);
    const real* kn = L.kn(N - 1);
    real e = qz_entry(ct, true, r, c);
    const real th = kn[th_off + r] + (r == 1 ? kn[tey_off] : real(0));
    if (diag) e += th;
    if (HAS_PT && r < 6 && c < 6) e += real(PT[r <= c ? r * 6 + c : c * 6 + r]);
    pown = e;
    MP[MROWS(r) + c] = e;
    M.wait();
    if (ch > 0) M.ensure(ch - 1);
  }
  // fused backward sweep of the predictor's right-hand sides (see riccati_factor)
  const int fs = (lane >> 4) & 1;
  const bool f_own = FUSE && (lane & 8) == 0 && lane < 32 && c < 2;
  const int f_oq = fs == 0 ? KN_R0 + c : KN_CSIG;
  const real f_mq = (fs == 0 || c == 1) ? real(1) : real(0);
  const real f_m6 = (c >= 6) ? real(1) : real(0);
  real fp = 0.0, fws = 0.0, fw6 = 0.0, fw7 = 0.0, fqz = 0.0, fqv0 = 0.0, fqv1 = 0.0;
  if constexpr (FUSE) fp = f_mq * L.kn(N - 1)[f_oq];
  const bool upper = r <= c;
  real* const p_dst0 = upper ? MP + MROWS(r) + c : MW + MROWS(r) + c;
  real* const p_dst1 = upper ? MP + MROWS(c) + r : MW + MROWS(r) + c;
  const int res_off = lane < 16 ? 2 * c + r : (lane == 18 ? LN_HI11 : LN_HI + (lane - 16));
  const bool res_on = lane < 19;
  real* const res_junk = MW + MROWS(r) + c;
  real ar[6], ac[6];
  {
    const real* ab = M.stage(N - 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      ar[k] = ab[6 * r + k];
      ac[k] = ab[6 * c + k];
    }
  }
  const real sv00 = uni(ct[CT_SV + 0]), sv01 = uni(ct[CT_SV + 1]), sv11 = uni(ct[CT_SV + 3]);
  wave_sync();
  real pr[6];
  if constexpr (FUSE) {
#pragma unroll
    for (int k = 0; k < 6; ++k) pr[k] = MP[MROWS(c) + k];
    pown = MP[MROWS(r) + c];
    ISSUE_ORDER();
  }
  for (int i = N - 2; i >= 0; --i) {
    real* st = L.st(i);
    const real* kn = L.kn(i);
    const real* ab = M.stage(i);
    const bool cross = (i % LN_CHUNK) == 0 && i > 0;  // the read-ahead of this stage is the first read of the chunk below
    if constexpr (!FUSE) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pr[k] = MP[MROWS(c) + k];
      pown = MP[MROWS(r) + c];
      ISSUE_ORDER();
    }
    const real t = st[LN_DT], thr = kn[th_off + r], ey = kn[tey_off], thv0 = kn[th_off + 8], thv1 = kn[th_off + 9];
    if constexpr (FUSE) {
      fqz = kn[f_oq];
      fqv0 = kn[KN_R0 + 8];
      fqv1 = kn[KN_R0 + 9];
    }
    ISSUE_ORDER();
    real w = m_r6 * pown;
#pragma unroll
    for (int k = 0; k < 6; ++k) w += ar[k] * pr[k];
    MW[MROWS(r) + c] = w;
    wave_sync();
    real wr[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) wr[k] = MW[MROWS(r) + k];
    if constexpr (FUSE) {
      ISSUE_ORDER();
      real pb[6];
      row_bcast6(fp, pb);
      fws = f_m6 * fp;
#pragma unroll
      for (int k = 0; k < 6; ++k) fws = rfma(ac[k], pb[k], fws);
      row_bcast67(fws, fw6, fw7);
    }
    real y = m_c6 * w;
#pragma unroll
    for (int k = 0; k < 6; ++k) y += wr[k] * ac[k];
    MY[MROWS(r) + c] = y;
    wave_sync();
    const real y6r = MY[MROWS(6) + r], y7r = MY[MROWS(7) + r];
    const real y6c = MY[MROWS(6) + c], y7c = MY[MROWS(7) + c];
    AFTER_VALUE(y);
    if constexpr (!JOSEPH) {
      if (cross) M.wait();
      const real* abn = M.stage(i > 0 ? i - 1 : 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        ar[k] = abn[6 * r + k];
        ac[k] = abn[6 * c + k];
      }
      if (cross && i / LN_CHUNK >= 2) {
        wave_fence();
        M.fetch(i / LN_CHUNK - 2);
      }
    }
    ISSUE_ORDER();
    const real y66 = lane_bcast(y, 54), y67 = lane_bcast(y, 55), y77 = lane_bcast(y, 63);
    const real tt = t * t;
    const real h00 = sv00 + thv0 + tt * y66;
    const real h01 = sv01 + tt * y67;
    const real h11 = sv11 + thv1 + tt * y77;
    const real idet = frcp(h00 * h11 - h01 * h01);
    const real hi00 = h11 * idet, hi01 = -h01 * idet, hi11 = h00 * idet;
    const real g0 = t * y6c, g1 = t * y7c;
    const real k0c = hi00 * g0 + hi01 * g1;
    const real k1c = hi01 * g0 + hi11 * g1;
    real pn;
    if constexpr (JOSEPH) {
      const real g0r = t * y6r, g1r = t * y7r;
      const real k0r = hi00 * g0r + hi01 * g1r;
      const real k1r = hi01 * g0r + hi11 * g1r;
      {
        real b0[6], b1[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          b0[k] = ab[36 + k];
          b1[k] = ab[42 + k];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          ar[k] -= t * (b0[k] * k0r + b1[k] * k1r);
          ac[k] -= t * (b0[k] * k0c + b1[k] * k1c);
        }
      }
      const real fr6 = (r == 6 ? real(1) : real(0)) - t * k0r, fr7 = (r == 7 ? real(1) : real(0)) - t * k1r;
      const real fc6 = (c == 6 ? real(1) : real(0)) - t * k0c, fc7 = (c == 7 ? real(1) : real(0)) - t * k1c;
      real w2 = 0.0;
      {
        real pc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pc[k] = MP[MROWS(c) + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) w2 += ar[k] * pc[k];
        w2 += fr6 * pc[6] + fr7 * pc[7];
      }
      MW[MROWS(r) + c] = w2;
      wave_sync();
      real y2 = 0.0;
      {
        real w2r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w2r[k] = MW[MROWS(r) + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) y2 += w2r[k] * ac[k];
        y2 += w2r[6] * fc6 + w2r[7] * fc7;
      }
      {
        if (cross) M.wait();
        const real* abn = M.stage(i > 0 ? i - 1 : 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          ar[k] = abn[6 * r + k];
          ac[k] = abn[6 * c + k];
        }
        if (cross && i / LN_CHUNK >= 2) {
          wave_fence();
          M.fetch(i / LN_CHUNK - 2);
        }
      }
      const real v00 = sv00 + thv0, v11 = sv11 + thv1;
      pn = qmid + y2 + k0r * (v00 * k0c + sv01 * k1c) + k1r * (sv01 * k0c + v11 * k1c);
    } else {
      pn = qmid + y - t * (y6r * k0c + y7r * k1c);
    }
    const real th = rfma(m_r1, ey, thr);
    pn = rfma(m_diag, th, pn);
    *p_dst0 = pn;
    *p_dst1 = pn;
    const real res = lane < 8 ? k0c : (lane < 16 ? k1c : (lane == 16 ? hi00 : (lane == 17 ? hi01 : hi11)));
    *(res_on ? st + res_off : res_junk) = res;
    wave_sync();
    if constexpr (FUSE) {
      if (i > 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pr[k] = MP[MROWS(c) + k];
        pown = MP[MROWS(r) + c];
      }
      ISSUE_ORDER();
      const real qv0 = fs == 0 ? fqv0 : real(0), qv1 = fs == 0 ? fqv1 : real(0);
      const real hv0 = rfma(t, fw6, qv0);
      const real hv1 = rfma(t, fw7, qv1);
      fp = f_mq * fqz + fws - (k0c * hv0 + k1c * hv1);
      const real ha = c == 0 ? hi00 : hi01, hb = c == 0 ? hi01 : hi11;
      const real kff = ha * hv0 + hb * hv1;
      *(f_own ? st + LN_KFF(fs) + c : res_junk) = kff;
    }
  }
  if constexpr (FUSE) {
    wave_sync();  // (what follows is the forward half of the solve: chunks 0 and 1 are what the factor ended on)
  } else if ((N - 2) / LN_CHUNK >= 2) {
    // what follows is a vector solve, whose backward sweep starts at the top again: its first chunks are on their way while
    // the gradient is assembled (both buffers are free: the factor is done with them)
    M.fetch((N - 2) / LN_CHUNK);
    M.fetch((N - 2) / LN_CHUNK - 1);
  }
}

// The lean vector solve with the running vector in registers (DPP row broadcasts, as riccati_solve) instead of one LDS round
// trip per stage: at one wave per SIMD a sweep stage costs what its instruction count costs (~4.5 cycles each, nothing to
// overlap with), and the exchange through LDS was a third of the lean stage's instructions (the LDS-exchange form of rounds 2-3,
// bit for bit the same results, is scratch/r5/experiment_switches.patch); lane (s, r) = ((lane >> 4) % NRHS, lane & 7), lanes 8..15 of a row mirror 0..7.
template <int NRHS, bool FWD_ONLY = false, typename real>
__device__ __forceinline__ void riccati_solve_lean_dpp(const Lds<real>& L, ModelStream<real>& M, int lane, Prof& pf) {
  FRESH_LANE(lane, 5);
  const int N = L.N;
  const int r = lane & 7, s = (lane >> 4) & (NRHS - 1);
  const bool own = (lane & 8) == 0 && lane < 16 * NRHS;
  const int reg = KN_R0 + 10 * s;
  real* T = L.tail();
  real* const junk0 = T + TL_W + lane;
  real* const junk1 = T + TL_W + 80 + lane;
  const real m6 = (r >= 6) ? real(1) : real(0);
  const int o_ha = r == 0 ? LN_HI : LN_HI + 1, o_hb = r == 0 ? LN_HI + 1 : LN_HI11;
  // ---- backward
  if constexpr (!FWD_ONLY) {
    const int ch = (N - 2) / LN_CHUNK;
    M.ensure(ch);
    M.wait();
    if (ch > 0) M.ensure(ch - 1);
  }
  real p = FWD_ONLY ? real(0) : L.kn(N - 1)[reg + r];
  real row[6];
  if constexpr (!FWD_ONLY) {
    const real* ab = M.stage(N - 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) row[k] = ab[6 * r + k];
  }
  for (int i = FWD_ONLY ? -1 : N - 2; i >= 0; --i) {
    real* st = L.st(i);
    const real* kn = L.kn(i);
    if ((i % LN_CHUNK) == 0 && i > 0) {  // (chunk boundary: the next chunk must have landed)
      M.wait();
      if (i / LN_CHUNK >= 2) M.fetch(i / LN_CHUNK - 2);
    }
    const real k0r = st[2 * r], k1r = st[2 * r + 1], t = st[LN_DT];
    const real qz = kn[reg + r], qv0 = kn[reg + 8], qv1 = kn[reg + 9];
    const real ha = st[o_ha], hb = st[o_hb];
    real pb[6];
    row_bcast6(p, pb);
    real w = m6 * p;
#pragma unroll
    for (int k = 0; k < 6; ++k) w = rfma(row[k], pb[k], w);
    real w6, w7;
    row_bcast67(w, w6, w7);
    {
      const real* abn = M.stage(i > 0 ? i - 1 : 0);
#pragma unroll
      for (int k = 0; k < 6; ++k) row[k] = abn[6 * r + k];
    }
    const real hv0 = rfma(t, w6, qv0);
    const real hv1 = rfma(t, w7, qv1);
    p = qz + w - (k0r * hv0 + k1r * hv1);
    const real kff = ha * hv0 + hb * hv1;
    *((own && r < 2) ? st + LN_KFF(s) + r : junk1) = kff;
  }
  wave_sync();
  PT_MARK(8 + NRHS - 1)
  // ---- forward
  const int nch = (N - 2) / LN_CHUNK + 1;
  M.ensure(0);
  *(own ? L.kn(0) + reg + r : junk0) = 0.0;
  M.wait();
  if (nch > 1) M.ensure(1);
  const int cb = r < 6 ? r : 0;
  const int cstride = r < 6 ? 6 : 2;
  real d = 0.0;
  real col[8], a0, b0n, b1n;
  auto load_stage = [&](int i) {
    const real* ab = M.stage(i);
    const real* st = L.st(i);
    {
      const real* const cbase = r < 6 ? ab + cb : st + (r - 6);
#pragma unroll
      for (int k = 0; k < 8; ++k) col[k] = cbase[k * cstride];
    }
    a0 = st[LN_KFF(s) + (r & 1)];
    b0n = ab[36 + cb];
    b1n = ab[42 + cb];
  };
  load_stage(0);
  wave_sync();
  for (int i = 0; i < N - 1; ++i) {
    const real* st = L.st(i);
    real* kn = L.kn(i);
    if ((i % LN_CHUNK) == LN_CHUNK - 1 && i < N - 2) {
      M.wait();
      if (i / LN_CHUNK + 2 < nch) M.fetch(i / LN_CHUNK + 2);
    }
    const real b0 = b0n, b1 = b1n, t = st[LN_DT];
    real dz[8];
    row_bcast8(d, dz);
    real acc = m6 * a0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc = rfma(col[k], dz[k], acc);
    const real ax = acc;
    acc = rfma(col[6], dz[6], acc);
    acc = rfma(col[7], dz[7], acc);
    const real dv = -acc;
    const real du = rfma(t, dv, d);
    real du0, du1;
    row_bcast67(du, du0, du1);
    load_stage(i < N - 2 ? i + 1 : i);
    const real nx = rfma(b1, du1, rfma(b0, du0, ax));
    d = (r < 6) ? nx : du;
    *(own ? kn + LMPC_KNOT_STRIDE + reg + r : junk0) = d;
    *((own && r >= 6) ? kn + reg + 2 + r : junk1) = dv;
  }
  wave_sync();
  PT_MARK(10 + NRHS - 1)
}

template <bool WARM = false, typename real>
__device__ __forceinline__ void feedback_rollout_lean(const Lds<real>& L, ModelStream<real>& M, int lane) {
  FRESH_LANE(lane, 6);
  const int N = L.N, r = lane & 7;
  const bool own = lane < 8;
  real* T = L.tail();
  real* const junk0 = T + TL_W + lane;
  real* const junk1 = T + TL_W + 80 + lane;
  const int nch = (N - 2) / LN_CHUNK + 1;
  const int cb = r < 6 ? r : 0;
  M.ensure(0);
  M.wait();
  if (nch > 1) M.ensure(1);
  for (int i = 0; i < N - 1; ++i) {
    const real* st = L.st(i);
    const real* ab = M.stage(i);
    real* kn = L.kn(i);
    real dz[8], col[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dz[k] = kn[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) col[k] = r < 6 ? ab[6 * k + cb] : st[2 * k + (r - 6)];
    const real t = st[LN_DT];
    const real g = ab[48 + cb];
    real acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc = rfma(col[k], dz[k], acc);
    const real ax = acc;
    acc = rfma(col[6], dz[6], acc);
    acc = rfma(col[7], dz[7], acc);
    real v = -acc;
    if constexpr (WARM) {  // (lanes 6, 7: col = a row of K)  v = v^plan + K z^plan - K z
      real kz = kn[KN_R0 + 8 + (r & 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k) kz = rfma(col[k], kn[KN_R0 + k], kz);
      v += kz;
    }
    const real u = rfma(t, v, (r == 6) ? dz[6] : dz[7]);
    const real u0 = lane_bcast(u, 6), u1 = lane_bcast(u, 7);
    const real nx = g + rfma(col[7], u1, rfma(col[6], u0, ax));
    *(own ? kn + LMPC_KNOT_STRIDE + r : junk0) = (r < 6) ? nx : u;
    *((own && r >= 6) ? kn + 2 + r : junk1) = v;
    wave_sync();
    if ((i % LN_CHUNK) == LN_CHUNK - 1 && i < N - 2) {  // the next stage is the first of the chunk above
      M.wait();
      if (i / LN_CHUNK + 2 < nch) M.fetch(i / LN_CHUNK + 2);
    }
  }
}

// ======== the active-set polish (polish_limits above; the twin's polish() runs the same rounds) ========
// A CALL, not part of the interior point's body (round 4; until then a lambda inlined into the solve): the polish keeps
// ~190 B more state per lane alive than an iteration does, and inlined -- even in an outer loop of its own -- it weighed on
// the register allocation of the iteration: 292 B of scratch per lane in the N = 20 fp64 kernel (824 B in the mixed learning
// kernel), 88 of the 116 scratch reloads of an iteration, +10 % per iteration.  Behind a call boundary the two are allocated
// separately.  What crosses it is passed BY VALUE (the iteration's arrays must not escape: an array whose address is taken
// lives in scratch for good): the row state t, lambda of the interior point (read only), the slot tables, the simplex rows,
// the wave-wide scalars.  The iterate itself is in LDS, which the callee names through the workgroup's dynamic LDS symbol
// (its accesses stay DS instructions).  Slacks and multipliers of the interior point are not touched; the iterate is put
// aside in the handle's save area and comes back unless the attempt is accepted.  Wave-uniform values arrive in vector
// registers (the calling convention has no scalar arguments) and go back to scalar registers first thing.
// Every instantiation runs the DPP form of the vector solves (riccati_solve, riccati_solve_lean_dpp) since round 4: bit for bit the
// LDS-exchange forms of rounds 2-3, -3 % at N = 40 tracking / IAC, -8 % at N = 60, -4 % at N = 80, -4 / -12 % for the learning
// problem at N = 60 / 80.  (Rounds 2-3 kept DPP out of the one-wave-per-SIMD kernels after a non-reproducible KQ = 14 / KS = 3
// build; with the polish behind a call that family has been bit-stable in every build since.  The LDS-exchange functions and
// their -D switches were removed in round 5: scratch/r5/experiment_switches.patch.)

template <typename real, int KQ, int KS>
struct PolishArgs {
  void* keep;      // this problem's block of the save area
  const void* ws;  // lean layout: this problem's records in the linearisation workspace (ModelStream::ws)
  int N, S, has_sigma, have0, have1, pol_rounds;
  int max_rounds;  // repairs this attempt may spend (polish_limits::rounds; fewer for a warm start)
  real qsig, inv_m, sigma;
  int o_val[KQ], o_hl[KQ], s_gf[KQ];
  real s_tu[KQ], s_tl[KQ], s_lu[KQ], s_ll[KQ];
  SimplexRows<double, KS> sx;
};
template <typename real, int KS>
struct PolishResult {
  int accepted, pol_rounds, have0, have1;
  int noise;  // refused, and the last round failed on nothing but the multiplier steps / the held rows' residual (see the exit of lmpc_solve_problem)
  real sigma, mu, rdmax, last_step;
  double lm[KS > 0 ? KS : 1];  // the simplex weights of an accepted polish
};
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <typename T>
__device__ __forceinline__ T* uni_ptr(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// Where the polish is a call, and where the sweeps take a fresh lane value (FRESH_LANE above) -- measured per instantiation in
// round 4 (profiles/r04_polish_forms.md: {inline, call} x {fresh, not}, every (KQ, KS), checksums against the round-3 build):
//   * fp64, one wave per SIMD (KQ >= 7 or the learning problem: the instantiations that live in VGPRs + AGPRs): a CALL.  The call
//     form computes bit for bit what the round-3 kernel computed, at every horizon and for both problems; the inlined function
//     does not: with FRESH_LANE in the factorisation AND the vector solve the <double, 7, 0> instance (N = 24 .. 40) returns
//     different -- wrong -- answers (72 of 8192 IAC problems "infeasible"), the same wrong answers whatever the post-RA schedule
//     or the wait counts, and the right ones again with SGPR spills sent to memory instead of VGPR lanes, or at -O2 (DESIGN.md
//     section 4, "the register-starved instantiations").  The call costs the callee's prologue / epilogue and the spills around
//     the call site, ~1 KB of scratch traffic per lane and call, and buys -5 .. -16 % of the kernel time from N = 60 on and for
//     the learning problem from N = 40 on (nothing either way at N = 40 tracking).
//   * fp64, two waves per SIMD (tracking up to N = 23: the headline): INLINED.  Same time as the call form (0.872 against 0.868 ms
//     per 4096 at N = 20), bit for bit the round-3 answers, and 102 MB of HBM traffic per launch against 250 MB (round 3: 174 MB):
//     at 256 VGPRs there are no AGPR copies for a call boundary to save, and the call's own spills are the larger traffic.
//   * fp32 (single precision and the fp32 pass of the mixed entry): INLINED -- the call form is 10 % slower on the mixed learning
//     kernel and changes single-precision roundings enough to lose four solves of 4096 at N = 80.
//   * FRESH_LANE everywhere.
// (The A/B switches behind these measurements -- never / always a call, never / always fresh -- went to scratch/r5/experiment_switches.patch.)
#ifndef LMPC_POLISH_CALL_2W  // (A/B switch: 1 puts the polish behind a call in the two-waves-per-SIMD fp64 kernels too)
#define LMPC_POLISH_CALL_2W 1
#endif
constexpr bool lmpc_polish_is_call(int real_bytes, int kq, int ks) { return real_bytes == 8 && (LMPC_POLISH_CALL_2W || lmpc_waves_per_simd(real_bytes, kq, ks) < 2); }
constexpr bool lmpc_fresh_lane(int real_bytes, int kq, int ks) { return true; }

template <typename real, int KQ, int KS, typename io, bool SECOND = false>  // (SECOND: see lmpc_solve_problem)
__device__ __forceinline__ PolishResult<real, KS> lmpc_polish(const PolishArgs<real, KQ, KS>& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  real* const lds = reinterpret_cast<real*>(lds_raw);
  typedef typename vec2<real>::type real2;
  typedef double treal;
  typedef polish_limits<real> pol;
  const int lane = threadIdx.x;
  const int N = uni(a.N), NS = N - 1;
  constexpr bool LEAN = lmpc_lean(sizeof(real), KQ);
  Lds<real> L{lds, N, LEAN ? LMPC_LEAN_STAGE_STRIDE : LMPC_STAGE_STRIDE, lmpc_fresh_lane(sizeof(real), KQ, KS),
              !SECOND && lmpc_waves_per_simd(sizeof(real), KQ, KS) >= 2};
  real* const T = L.tail();
  treal* const TT = reinterpret_cast<treal*>(T + LMPC_TAIL_DOUBLES);
  ModelStream<real> MS{nullptr, nullptr, NS, lane, uni(a.have0), uni(a.have1)};
  if constexpr (LEAN) {
    MS.ws = uni_ptr(reinterpret_cast<const real*>(a.ws));
    MS.buf = reinterpret_cast<real*>(reinterpret_cast<unsigned char*>(TT) +
                                     (KS > 0 ? (LMPC_TERM_CELLS + 6 * LMPC_SS_STRIDE(uni(a.S))) * sizeof(treal) : 0));
  }
  const real inf = real(INFINITY);
  const treal tinf = treal(INFINITY);
  const bool has_sigma = uni(a.has_sigma) != 0;
  const real qsig = uni(a.qsig), inv_m = uni(a.inv_m);
  real sigma = uni(a.sigma);
  real hsig = 0.0, ce = 0.0, mu = 0.0, rdmax = 0.0, last_step = 0.0;
  int pol_rounds = uni(a.pol_rounds);
  Prof pf;
  (void)pf;
  (void)tinf;
  const int KNB = NS * L.stride;
  const int JB = KNB + N * LMPC_KNOT_STRIDE + TL_W;
  const int CTB = KNB + N * LMPC_KNOT_STRIDE + TL_CT;
  int o_val[KQ], o_hl[KQ], s_gf[KQ];
  real s_tu[KQ], s_tl[KQ], s_lu[KQ], s_ll[KQ], s_pu[KQ], s_pl[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    o_val[q] = a.o_val[q];
    o_hl[q] = a.o_hl[q];
    s_gf[q] = a.s_gf[q];
    s_tu[q] = a.s_tu[q];
    s_tl[q] = a.s_tl[q];
    s_lu[q] = a.s_lu[q];
    s_ll[q] = a.s_ll[q];
    s_pu[q] = s_pl[q] = 0.0;
  }
  SimplexRows<treal, KS> sx = a.sx;
  if constexpr (KS > 0) {
    sx.ul = TT + TL_UL;
    sx.tau = uni(sx.tau);
#pragma unroll
    for (int k = 0; k < 6; ++k) sx.ss0[k] = uni(sx.ss0[k]);
  }
  auto flags = [&](int q) { return s_gf[q] >> 20; };
  auto o_w = [&](int q) { return o_val[q] + ((flags(q) & F_EY) ? KN_EY - 1 : KN_R0); };
  auto o_csig = [&](int q) { return (flags(q) & F_EY) ? o_val[q] + (KN_CSIG - 1) : JB + KN_CSIG; };
  auto bounds = [&](int q) { return *reinterpret_cast<const real2*>(&lds[o_hl[q]]); };
  // The iterate is put aside in the handle's save area -- one contiguous block of 10 N - 4 values per problem (full-line
  // writes), values as they stand in LDS, so that they read back exactly -- and taken back from there at the start of every
  // round and when the attempt is refused: every lane reads back what it wrote itself.
  io* const keep = uni_ptr(reinterpret_cast<io*>(a.keep));
  auto put_keep = [&]() {
    for (int e = lane; e < 10 * N - 4; e += 64) {  // knot i: z [8] (x_i, u_{i-1}) at 10 i - 2 .. (knot 0: x_0 only is not kept), v_i [2] behind it
      const int i = (e + 2) / 10, o = e + 2 - 10 * i;
      keep[e] = io(L.kn(i)[o]);
    }
  };
  auto get_primal = [&]() {
    for (int e = lane; e < 10 * N - 4; e += 64) {
      const int i = (e + 2) / 10, o = e + 2 - 10 * i;
      if (i >= 1 || o >= 8) L.kn(i)[o] = real(keep[e]);
    }
  };
  // (bit 8 of max_rounds: the attempt at the interior point's EXIT, fp64: in doubt -- lam ~ t, a weakly active row -- a row is held,
  //  from lam > 1e-2 t on; held wrongly it comes out with a negative multiplier and the next round releases it, free wrongly it is
  //  violated, drags its neighbours' multipliers below zero and the repairs release THEM.  oracle/c/lmpc_oracle.c, POLISH_EXIT, has
  //  the problem this was found on.)
  const real gam = (sizeof(real) == 8 && (uni(a.max_rounds) & POLISH_EXIT)) ? real(POLISH_EXIT_GAMMA) : real(1);
  int held = 0;  // (per lane) bit 2q / 2q + 1: upper / lower row of slot q held; bit 28 + q: simplex row q held
#pragma unroll
  for (int q = 0; q < KQ; ++q) held |= (s_lu[q] > gam * s_tu[q] ? 1 << (2 * q) : 0) | (s_ll[q] > gam * s_tl[q] ? 2 << (2 * q) : 0);
  if constexpr (KS > 0) {
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      held |= (sx.on[q] && sx.l[q] > sx.t[q]) ? 1 << (28 + q) : 0;
      sx.sv[q] = sx.lm[q];
    }
  }
  const real sigma_keep = sigma;
  put_keep();
  bool accepted = false, noise = false;
  const int max_rounds = min((int)pol::rounds, uni(a.max_rounds) & ~POLISH_EXIT);
  for (int round = 0; round < max_rounds; ++round) {
    if (round > 0) {  // every round starts from the interior point's iterate
      wave_sync();
      get_primal();
      sigma = sigma_keep;
      if constexpr (KS > 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) sx.lm[q] = sx.sv[q];
      }
      wave_sync();
    }
    // ---- weights: theta on the held rows, nothing on the others; multipliers start from the interior point's on the
    // rows it held itself and from zero on rows a repair has added ----
    real eysum = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int f = flags(q);
      const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
      const real thu = hu ? real(pol::theta) : real(0), thd = hd ? real(pol::theta) : real(0);
      lds[o_w(q)] = thu + thd;
      lds[o_csig(q)] = (f & F_SIG) ? (thd - thu) : real(0);
      eysum += (f & F_SIG) ? (thu + thd) : real(0);
      s_pu[q] = (hu && s_lu[q] > s_tu[q]) ? s_lu[q] : real(0);
      s_pl[q] = (hd && s_ll[q] > s_tl[q]) ? s_ll[q] : real(0);
    }
    if constexpr (KS > 0) {
      // the free simplex weights have theta = 0 and must ALL be explicit unknowns of the terminal block: more than MA_MAX
      // of them and the round cannot be set up
      treal thq[KS];
      int nfree = 0;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const bool hq = (held >> (28 + q)) & 1;
        thq[q] = !sx.on[q] ? tinf : (hq ? treal(POLISH_THETA_L) : treal(0));
        sx.aidx[q] = -1;
        nfree += (sx.on[q] && !hq) ? 1 : 0;
        sx.p[q] = (hq && sx.l[q] > sx.t[q]) ? sx.l[q] : treal(0);
      }
      if (__popcll(__ballot(nfree > 0)) + __popcll(__ballot(nfree > 1)) + __popcll(__ballot(nfree > 2)) > MA_MAX) break;
      if (lane < 6 * MA_MAX) TT[TL_UA + lane] = 0.0;
      if (lane < MA_MAX) {
        TT[TL_THA + lane] = 1.0;
        TT[TL_RA + lane] = 0.0;
      }
      wave_fence();
      int m = 0;
      for (int a = 0; a < MA_MAX; ++a) {
        treal cand = tinf;
        int cq = 0;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const bool better = sx.aidx[q] < 0 && thq[q] < sx.tau && thq[q] < cand;
          cand = better ? thq[q] : cand;
          cq = better ? q : cq;
        }
        const treal best = wave_min(cand);
        if (!(best < tinf)) break;
        const int owner = __ffsll((long long)__ballot(cand == best)) - 1;
        if (lane == owner) {
#pragma unroll
          for (int q = 0; q < KS; ++q)
            if (q == cq) {
              sx.aidx[q] = a;
              treal uq[6];
              sx.load_u(q, lane, uq);
#pragma unroll
              for (int k = 0; k < 6; ++k) TT[TL_UA + a * 6 + k] = uq[k];
              TT[TL_THA + a] = thq[q];
            }
        }
        wave_fence();
        m = a + 1;
      }
      sx.m = m;
      wave_fence();
      treal tt[21], av[14];
#pragma unroll
      for (int k = 0; k < 21; ++k) tt[k] = 0.0;
#pragma unroll
      for (int k = 0; k < 14; ++k) av[k] = 0.0;
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const treal itf = (sx.on[q] && sx.aidx[q] < 0) ? frcp(thq[q]) : treal(0);
        int n = 0;
        treal uq[6];
        sx.load_u(q, lane, uq);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = r; c < 6; ++c) tt[n++] += uq[r] * uq[c] * itf;
          av[r] += uq[r] * itf;
          av[7 + r] += uq[r] * sx.lm[q];
        }
        av[6] += itf;
        av[13] += sx.lm[q];
      }
      {
        treal red[32], red3[3] = {av[11], av[12], av[13]};
#pragma unroll
        for (int k = 0; k < 21; ++k) red[k] = tt[k];
#pragma unroll
        for (int k = 0; k < 11; ++k) red[21 + k] = av[k];
        wave_sum_split<32>(red, lane);
        wave_sum_split<3>(red3, lane);
#pragma unroll
        for (int k = 0; k < 21; ++k) tt[k] = red[k];
#pragma unroll
        for (int k = 0; k < 11; ++k) av[k] = red[21 + k];
        av[11] = red3[0];
        av[12] = red3[1];
        av[13] = red3[2];
      }
      sx.r1 = 1.0 - av[13];
      {
        treal F[36], aB[6];
        int n = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = r; c < 6; ++c) {
            F[r * 6 + c] = tt[n];
            F[c * 6 + r] = tt[n];
            ++n;
          }
          aB[r] = av[r];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) F[k * 6 + k] += 1.0 / fmax(TT[TL_E + k], treal(1e-30));
        term_factor_u(TT, lane, F, aB, av[6], m);
      }
      if (lane < 6) {
        treal e = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (k == lane) e = (treal(L.kn(N - 1)[k]) - sx.ss0[k]) - av[7 + k];
        TT[TL_EPS + lane] = e;
      }
      wave_fence();
    }
    hsig = uni(qsig + wave_sum(eysum));
    ++pol_rounds;
    wave_sync();
    if constexpr (LEAN)
      riccati_factor_lean<(KS > 0), true>(L, MS, lane, TT + TL_PT);
    else
      riccati_factor<(KS > 0), true>(L, lane, TT + TL_PT);
    bool nan_step = false;
    // ---- pol::steps multiplier steps on that factor, each from the point the one before has reached ----
    for (int k = 0; k < pol::steps; ++k) {
      treal eeps[6] = {0, 0, 0, 0, 0, 0};
      if constexpr (KS > 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) eeps[c] = uni(TT[TL_E + c] * TT[TL_EPS + c]);
        treal bs[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal itf, uq[6];
          sx.load_u(q, lane, uq);
          const treal rj = sx.on[q] ? -simplex_bl_polish(sx.lm[q], sx.p[q], ((held >> (28 + q)) & 1) != 0, sx.j[q], uq, eeps, itf) : treal(0);
          if (sx.aidx[q] >= 0) TT[TL_RA + sx.aidx[q]] = rj;
          const treal w = (sx.on[q] && sx.aidx[q] < 0) ? rj * itf : treal(0);
          bs[6] += w;
#pragma unroll
          for (int c = 0; c < 6; ++c) bs[c] += uq[c] * w;
        }
        wave_sum_split<7>(bs, lane);
        wave_fence();
        treal beta[6], h[6], nu;
#pragma unroll
        for (int c = 0; c < 6; ++c) beta[c] = bs[c];
        term_solve_u(TT, lane, sx.m, beta, bs[6], sx.r1, h, nu);
        if (lane < 6) {
          treal hs = h[0];
#pragma unroll
          for (int c = 1; c < 6; ++c) hs = (lane == c) ? h[c] : hs;
          TT[TL_TG + lane] = TT[TL_E + lane] * TT[TL_EPS + lane] - hs;
        }
        wave_fence();
      }
      real sgsum = 0.0;
      {  // gradient: cost gradient + (y + theta * residual) on the held rows
        real val[KQ], par[KQ], ca[KQ], cb[KQ], ql[KQ];
        real2 hl[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int gr = s_gf[q];
          val[q] = lds[o_val[q]];
          hl[q] = bounds(q);
          par[q] = lds[o_val[q] + ((gr >> 16) & 3) - 1];
          ca[q] = lds[CTB + (gr & 0xff)];
          cb[q] = lds[CTB + ((gr >> 8) & 0xff)];
          ql[q] = lds[o_val[q] + (KN_QLIN - 3)];
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int f = flags(q);
          const real sg = (f & F_SIG) ? sigma : 0.0;
          const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
          const real cu = hu ? rfma(real(pol::theta), val[q] - sg - hl[q].x, s_pu[q]) : real(0);
          const real cd = hd ? rfma(real(pol::theta), -val[q] - sg + hl[q].y, s_pl[q]) : real(0);
          const real g = ca[q] * val[q] + cb[q] * par[q] + ((f & F_QLIN) ? ql[q] : real(0));
          lds[o_w(q)] = g + cu - cd;
          if (k == 0) lds[(f & F_EY) ? JB + KN_EY : o_w(q) + 10] = 0.0;
          sgsum += (f & F_SIG) ? (cu + cd) : real(0);
        }
      }
      wave_sync();
      for (int i = lane; i < N; i += 64) {
        real* kn = L.kn(i);
        kn[KN_R0 + 1] += kn[KN_EY];
        if (k == 0) kn[KN_R1 + 1] = (i >= 1) ? kn[KN_CSIG] : 0.0;
      }
      if constexpr (KS > 0) {
        wave_sync();
        if (lane < 6) L.kn(N - 1)[KN_R0 + lane] += real(TT[TL_TG + lane]);
      }
      wave_sync();
      if constexpr (LEAN) {
        if (k == 0 && has_sigma)
          riccati_solve_lean_dpp<2>(L, MS, lane, pf);
        else
          riccati_solve_lean_dpp<1>(L, MS, lane, pf);
      } else {
        if (k == 0 && has_sigma)
          riccati_solve<2>(L, lane, pf);
        else
          riccati_solve<1>(L, lane, pf);
      }
      real dz0[KQ], dz1[KQ], val[KQ];
      real2 hl[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        dz0[q] = lds[o_val[q] + 10];
        dz1[q] = lds[o_val[q] + 20];
        val[q] = lds[o_val[q]];
        hl[q] = bounds(q);
      }
      real dsigma = 0.0;
      if (has_sigma) {
        real red[3] = {0.0, 0.0, sgsum};
        {
          real cs[KQ];
#pragma unroll
          for (int q = 0; q < KQ; ++q) cs[q] = lds[o_csig(q)];
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            const bool sch = (flags(q) & F_SCH) != 0;
            red[0] += sch ? cs[q] * dz0[q] : real(0);
            red[1] += sch ? cs[q] * dz1[q] : real(0);
          }
        }
        wave_sum_n<3>(red);
        if (k == 0) ce = red[1];
        const real qsg = qsig * sigma - red[2];
        dsigma = uni(-(qsg + red[0]) / (hsig + ce));
      }
      if constexpr (KS > 0) {
        const real* knT = L.kn(N - 1);
        treal e[6], gs[7] = {0, 0, 0, 0, 0, 0, 0}, rj[KS], itfq[KS];
#pragma unroll
        for (int c = 0; c < 6; ++c) e[c] = TT[TL_E + c] * treal(knT[KN_R0 + c] + dsigma * knT[KN_R1 + c]);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal itf, uq[6];
          sx.load_u(q, lane, uq);
          treal r = -simplex_bl_polish(sx.lm[q], sx.p[q], ((held >> (28 + q)) & 1) != 0, sx.j[q], uq, eeps, itf);
#pragma unroll
          for (int c = 0; c < 6; ++c) r += uq[c] * e[c];
          r = sx.on[q] ? r : 0.0;
          rj[q] = r;
          itfq[q] = itf;
          if (sx.aidx[q] >= 0) TT[TL_RA + sx.aidx[q]] = r;
          const treal w = (sx.on[q] && sx.aidx[q] < 0) ? r * itf : treal(0);
          gs[6] += w;
#pragma unroll
          for (int c = 0; c < 6; ++c) gs[c] += uq[c] * w;
        }
        wave_sum_split<7>(gs, lane);
        wave_fence();
        treal beta[6], h[6], nu;
#pragma unroll
        for (int c = 0; c < 6; ++c) beta[c] = gs[c];
        term_solve_u(TT, lane, sx.m, beta, gs[6], sx.r1, h, nu);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal uh = 0.0, uq[6];
          sx.load_u(q, lane, uq);
#pragma unroll
          for (int c = 0; c < 6; ++c) uh += uq[c] * h[c];
          const treal dB = (rj[q] - nu - uh) * itfq[q];
          const treal dA = TT[TL_XA + (sx.aidx[q] >= 0 ? sx.aidx[q] : 0)];
          sx.dl[q] = sx.on[q] ? (sx.aidx[q] >= 0 ? dA : dB) : treal(0);
        }
      }
      // the full step, taken at once; multipliers from the residual BEFORE the step plus the row's own increment (the
      // stored value is rounded after the update, the increment is not: in single precision the re-read residual of a
      // row that has landed on its bound is exactly zero and says nothing)
      bool finite_step = dsigma == dsigma;
      real stepmax = 0.0;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int f = flags(q);
        const real dval = dz0[q] + dsigma * dz1[q];
        finite_step = finite_step && (fabs(dval) < inf);
        stepmax = fmax(stepmax, (f & F_MOVE) ? fabs(dval) * real(slot_inv_scale((s_gf[q] >> 27) & 15)) : real(0));
        const real sg = (f & F_SIG) ? sigma : 0.0, dsg = (f & F_SIG) ? dsigma : 0.0;
        const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
        s_pu[q] = hu ? rfma(real(pol::theta), (val[q] - sg - hl[q].x) + (dval - dsg), s_pu[q]) : real(0);
        s_pl[q] = hd ? rfma(real(pol::theta), (-val[q] - sg + hl[q].y) + (-dval - dsg), s_pl[q]) : real(0);
        const bool mv = (f & F_MOVE) != 0;
        lds[mv ? o_val[q] : JB + q] += mv ? dval : real(0);
      }
      if (has_sigma) sigma = uni(sigma + dsigma);
      if constexpr (KS > 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const bool hq = (held >> (28 + q)) & 1;
          sx.p[q] = hq ? sx.p[q] + treal(POLISH_THETA_L) * (-sx.lm[q] - sx.dl[q]) : treal(0);
          sx.lm[q] += sx.dl[q];
          finite_step = finite_step && (fabs(sx.dl[q]) < tinf);
        }
      }
      nan_step = nan_step || (__ballot(!finite_step) != 0);
      last_step = wave_max(stepmax);  // (scaled: what kkt[0] reports after an accepted polish)
      wave_sync();
      if constexpr (KS > 0) {  // the hull residual and the simplex residual at the new point (the next step's gradient)
        treal ul[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal uq[6];
          sx.load_u(q, lane, uq);
#pragma unroll
          for (int c = 0; c < 6; ++c) ul[c] += uq[c] * sx.lm[q];
          ul[6] += sx.lm[q];
        }
        wave_sum_split<7>(ul, lane);
        sx.r1 = 1.0 - ul[6];
        if (lane < 6) {
          treal e = 0.0;
#pragma unroll
          for (int c = 0; c < 6; ++c)
            if (c == lane) e = (treal(L.kn(N - 1)[c]) - sx.ss0[c]) - ul[c];
          TT[TL_EPS + lane] = e;
        }
        wave_fence();
      }
#ifdef LMPC_POLISH_TRACE
      if (blockIdx.x == 0 && lane == 0) printf("      step %d: %.3e\n", k, (double)last_step);
#endif
      if (k >= 1 && last_step <= real(pol::step_ok)) break;  // (converged: the remaining steps would move nothing)
    }
    // ---- KKT test of the point reached; repair of the held set ----
    real val[KQ];
    real2 hl[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      val[q] = lds[o_val[q]];
      hl[q] = bounds(q);
    }
    bool bad = false, neg = false, weakneg = false, viol = false;
    real ymin = 0.0, comp = 0.0, worst = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int f = flags(q);
      const real sg = (f & F_SIG) ? sigma : 0.0;
      const real ru = val[q] - sg - hl[q].x, rl = -val[q] - sg + hl[q].y;
      const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
      const bool nu_ = hu && s_pu[q] < -real(pol::dual), nd_ = hd && s_pl[q] < -real(pol::dual);
      bad = bad || (hu && !(fabs(ru) <= real(pol::feas))) || (hd && !(fabs(rl) <= real(pol::feas)));
      neg = neg || nu_ || nd_;
      weakneg = weakneg || (nu_ && s_lu[q] < real(POLISH_STRONG) * s_tu[q]) || (nd_ && s_ll[q] < real(POLISH_STRONG) * s_tl[q]);
      viol = viol || (!hu && (f & F_UP) && !(ru <= real(pol::feas))) || (!hd && (f & F_LO) && !(rl <= real(pol::feas)));
      ymin = fmin(ymin, fmin(hu ? s_pu[q] : real(0), hd ? s_pl[q] : real(0)));
      comp += (hu ? fabs(s_pu[q] * ru) : real(0)) + (hd ? fabs(s_pl[q] * rl) : real(0));
      worst = fmax(worst, fmax((f & F_UP) ? ru : real(0), (f & F_LO) ? rl : real(0)));
    }
    if constexpr (KS > 0) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const bool hq = (held >> (28 + q)) & 1, fr = sx.on[q] && !hq;
        const bool nq = hq && sx.p[q] < -treal(pol::dual_l);
        bad = bad || (hq && !(fabs(sx.lm[q]) <= treal(pol::feas)));
        neg = neg || nq;
        weakneg = weakneg || (nq && sx.l[q] < treal(POLISH_STRONG) * sx.t[q]);
        viol = viol || (fr && !(-sx.lm[q] <= treal(pol::feas)));
        ymin = fmin(ymin, hq ? real(sx.p[q]) : real(0));
        comp += hq ? real(fabs(sx.p[q] * sx.lm[q])) : real(0);
      }
    }
    // (a last multiplier step that still moved the iterate: the steps have not converged -- held rows that are stiff meet
    // their bounds long before the point is stationary, so the row tests alone would pass)
    const bool anybad = nan_step || !(last_step <= real(pol::step_tol)) || __ballot(bad) != 0;
    const bool anyneg = __ballot(neg) != 0, anyweak = __ballot(weakneg) != 0;
    const bool anyviol = __ballot(viol) != 0;
#ifdef LMPC_POLISH_TRACE  // (diagnostics: workgroup 0 prints what each round's KKT test saw)
    {
      const real tr_held = wave_sum(real(__popc(held & 0x0fffffff))), tr_ymin = wave_min(ymin), tr_worst = wave_max(worst);
      if (blockIdx.x == 0 && lane == 0)
        printf("polish round %d: held %d last_step %.3e bad %d (rows %d nan %d) neg %d weak %d viol %d ymin %.3e worst %.3e\n", pol_rounds, (int)tr_held,
               (double)last_step, (int)anybad, (int)(__ballot(bad) != 0), (int)nan_step, (int)anyneg, (int)anyweak, (int)anyviol, (double)tr_ymin, (double)tr_worst);
    }
#endif
    if (!anybad && !anyneg && !anyviol) {  // the optimum for the held set, and the held set passes the KKT test
      mu = uni(wave_sum(comp) * inv_m);
      rdmax = wave_max(worst);
      accepted = true;
      break;
    }
    noise = anybad && !anyneg && !anyviol;  // a consistent held set whose multiplier steps did not converge
    // repair: release rows with a negative multiplier -- those the interior point did not hold firmly if there are such,
    // otherwise the most negative ones (a wrong row drags its neighbours' multipliers below zero) -- and only when no
    // multiplier is negative, hold the rows the new point violates
    const real ycut = real(0.5) * wave_min(ymin);
    const int before = held;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int f = flags(q);
      const real sg = (f & F_SIG) ? sigma : 0.0;
      const real ru = val[q] - sg - hl[q].x, rl = -val[q] - sg + hl[q].y;
      const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
      const bool du = hu && s_pu[q] < -real(pol::dual) && (anyweak ? s_lu[q] < real(POLISH_STRONG) * s_tu[q] : s_pu[q] <= ycut);
      const bool dd = hd && s_pl[q] < -real(pol::dual) && (anyweak ? s_ll[q] < real(POLISH_STRONG) * s_tl[q] : s_pl[q] <= ycut);
      const bool au = !anyneg && !hu && (f & F_UP) && !(ru <= real(pol::feas));
      const bool ad = !anyneg && !hd && (f & F_LO) && !(rl <= real(pol::feas));
      held = (held & ~((du ? 1 : 0) << (2 * q)) & ~((dd ? 2 : 0) << (2 * q))) | ((au ? 1 : 0) << (2 * q)) | ((ad ? 2 : 0) << (2 * q));
    }
    if constexpr (KS > 0) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        const bool hq = (held >> (28 + q)) & 1, fr = sx.on[q] && !hq;
        const bool dq = hq && sx.p[q] < -treal(pol::dual_l) && (anyweak ? sx.l[q] < treal(POLISH_STRONG) * sx.t[q] : real(sx.p[q]) <= ycut);
        const bool aq = !anyneg && fr && !(-sx.lm[q] <= treal(pol::feas));
        held = (held & ~((dq ? 1 : 0) << (28 + q))) | ((aq ? 1 : 0) << (28 + q));
      }
    }
    if (__ballot(held != before) == 0) break;  // (the multiplier steps did not converge on a consistent set: nothing to repair)
  }
  if (!accepted) {  // refused: iterate, slack variable and simplex weights as the interior point left them
    wave_sync();
    get_primal();
    sigma = sigma_keep;
    if constexpr (KS > 0) {
#pragma unroll
      for (int q = 0; q < KS; ++q) sx.lm[q] = sx.sv[q];
    }
    wave_sync();
  }
  PolishResult<real, KS> res;
  res.accepted = accepted ? 1 : 0;
  res.noise = (!accepted && noise) ? 1 : 0;
  res.pol_rounds = pol_rounds;
  res.have0 = MS.have0;
  res.have1 = MS.have1;
  res.sigma = sigma;
  res.mu = mu;
  res.rdmax = rdmax;
  res.last_step = last_step;
  if constexpr (KS > 0) {
#pragma unroll
    for (int q = 0; q < KS; ++q) res.lm[q] = sx.lm[q];
  }
  return res;
}

template <typename real, int KQ, int KS, typename io, bool SECOND = false>
__device__ __attribute__((noinline)) PolishResult<real, KS> lmpc_polish_call(const PolishArgs<real, KQ, KS> a) {
  return lmpc_polish<real, KQ, KS, io, SECOND>(a);
}

// `real` is the arithmetic and LDS type, `io` the type of the arrays in HBM: <double, double> is the reference's
// precision, <float, float> the single-precision path, <float, double> the mixed path (fp64 linearisation, regression,
// safe-set centring and results around an fp32 interior-point iteration).
// One problem, solved by the calling wavefront in the LDS block it is given (the body of both kernels below).
// Row-phase policies, per instantiation by measurement (profiles/r04_row_phases.md):
//   slots per chunk -- the long-horizon fp64 tracking kernels take their 11 / 14 slots half at a time (load, compute, store);
//   opaque slot tables (SlotRef below) -- the long-horizon fp64 kernels and every learning kernel.
// (fp32 at KQ >= 11 has no spills to begin with and loses 2-3 % to either; KQ <= 7 tracking loses 1-3 % to the opaque tables;
//  the learning kernels at KQ >= 11 lose 15 % to the chunks.)
// (bit mask of the four flag-select address sites recomputed per use in the fp64 tracking kernels with KQ <= 4: all four, -1.6..2.2 %
//  at N = 20, bit-identical; +1 % at KQ = 7 and in fp32, which keep the hoisted form: profiles/r04_row_phases.md)
__host__ __device__ constexpr int lmpc_row_chunk(int real_bytes, int kq, int ks) {
  return (real_bytes == 8 && kq >= 11 && ks == 0) ? (kq + 1) / 2 : kq;
}
__host__ __device__ constexpr bool lmpc_opaque_slots(int real_bytes, int kq, int ks) { return (real_bytes == 8 && kq >= 11) || ks > 0; }
__host__ __device__ constexpr int lmpc_opaque_sites(int real_bytes, int kq, int ks) { return (real_bytes == 8 && kq <= 4 && ks == 0) ? 15 : 0; }
// the predictor's backward sweep fused into the factorisation (riccati_factor<.., FUSE>): per instantiation, by measurement
// (MI355X, 4096 problems, kernel ms five chains -> four; profiles/r06_fuse_ab.txt):
//   fp64 tracking  N = 20 0.859 -> 0.835, N = 24 1.639 -> 1.565, N = 40 2.461 -> 2.351, N = 60 5.368 -> 4.957, IAC N = 40 4.322 -> 4.052
//   fp64 learning  N = 20 / 160 points 2.072 -> 2.028 (on); N = 40 5.52 -> 5.81, N = 60 14.54 -> 14.21 (off: KQ >= 7 with KS > 0 is the most register-starved family)
//   fp32 / mixed   IAC N = 40 3.275 -> 3.203 / 6.091 -> 5.850 (on); learning N = 20 mixed 3.385 -> 3.239, but OFF: which ill-conditioned blends
//                  of safe-set points pass the fp32 KKT test 1e-3 .. 5e-3 from the fp64 answer is decided by the last bits of the fp32
//                  sweeps -- 3 of configs[4]'s 32768 before, 5 fused (tests/test_gpu_spec_workload.py holds the 99.99 % quantile to 1e-3);
//                  tracking N <= 23 (KQ <= 4, three waves per SIMD): OFF -- the <float, 4, 0, double> instance of the fused build
//                  took a memory access fault in the mixed entry (the polish's flat store of the iterate to the save area with a
//                  clobbered address register; the fp32-array instance of the same source is fine): another of the
//                  compiler-sensitive corners of DESIGN.md section 4, found by tests/dispatch_sweep.py on its first run.
//   fp64 tracking N <= 23 (KQ <= 4, two waves per SIMD; the headline): ON, WITH THE POLISH BEHIND A CALL (LMPC_POLISH_CALL_2W).  Fused with the
//                  polish inlined it gained 2.7 % and passed every test -- until an unrelated edit of the polish (a multiplier in its
//                  classification) changed the register allocation: <double, 4, 0, double> then left the iteration after its first pass
//                  (status MAX_ITER, 0 iterations; the loop's control variables read back correct; the same source with a printf, or with
//                  one more integer assigned before each break, is correct: CHANGELOG.md, round 6).  Behind a call the polish's live state
//                  (the spill source of this kernel since round 3) is out of the iteration's register allocation -- the form every
//                  one-wave kernel has always had, all of them fused without incident: 0.819 -> 0.794 ms per 4096, every GPU test green
//                  (the call form alone, unfused: 0.817).
#ifndef LMPC_FUSE_MASK
#define LMPC_FUSE_MASK 0x57
#endif
__host__ __device__ constexpr bool lmpc_fuse_bwd(int real_bytes, int kq, int ks) {
  // bit 0: fp64 tracking KQ <= 4 (two waves per SIMD), 1: fp64 tracking KQ = 7, 2: fp64 tracking lean (KQ >= 11), 3: fp64 learning KQ >= 7,
  // 4: fp32 / mixed tracking KQ >= 7, 5: fp32 / mixed tracking KQ <= 4, 6: fp64 learning KQ <= 4, 7: fp32 / mixed learning
  return real_bytes == 8 ? (ks == 0 ? (kq <= 4 ? (LMPC_FUSE_MASK & 1) : (kq <= 7 ? (LMPC_FUSE_MASK & 2) : (LMPC_FUSE_MASK & 4))) : (kq <= 4 ? (LMPC_FUSE_MASK & 64) : (LMPC_FUSE_MASK & 8))) != 0
                         : (ks == 0 ? (kq <= 4 ? (LMPC_FUSE_MASK & 32) : (LMPC_FUSE_MASK & 16)) : (LMPC_FUSE_MASK & 128)) != 0;
}

// SECOND: the fp64 second pass of a mixed solve -- a handful of problems a whole batch waits for, sharing the chip with the next batch's
// first pass: its waves run at the top issue priority throughout (and do not drop it between chains).
template <typename real, int KQ, int KS, typename io, bool SECOND = false, bool WARMK = false>  // (WARMK: the warm-start kernels, below)
__device__ __forceinline__ void lmpc_solve_problem(
    const lmpc_params& P, const int B, const int b, unsigned char* lds_raw, const io* __restrict__ ws_lin,
    const io* __restrict__ x_ic, const io* __restrict__ u_ic, const io* __restrict__ T_ref, const io* __restrict__ bl,
    const io* __restrict__ br, const io* __restrict__ vref, const io* __restrict__ ss_x,
    const io* __restrict__ ss_j, io* __restrict__ lam_out, io* __restrict__ X_out,
    io* __restrict__ U_out, io* __restrict__ dU_out, int* __restrict__ status_out,
    int* __restrict__ iters_out, io* __restrict__ kkt_out) {
  real* const lds = reinterpret_cast<real*>(lds_raw);
  const int lane = threadIdx.x;
  const int N = P.N, NS = N - 1;
  typedef typename vec2<real>::type real2;
  typedef double treal;  // the safe-set block (simplex rows, terminal elimination) is always carried in fp64
  typedef ipm_limits<real> lim;
  const real inf = real(INFINITY), marg = real(P.marg), qsig = real(P.qsig), tol = lim::tol(P.tol);
  // single precision carries the abscissa relative to x_ic[0] (the QP is invariant to the shift: A(:, s) = e_s)
  const io s_shift = sizeof(real) == 4 ? x_ic[b] : io(0);
  constexpr bool LEAN = lmpc_lean(sizeof(real), KQ);
  // slots a row phase loads, computes and stores in one go: all of them where the registers hold the temporaries of all KQ at
  // once; LMPC_ROW_CHUNK at a time in the long-horizon kernels (KQ >= 11: 11-14 slots x 7 operands on top of 12 doubles of row
  // state per slot do not fit 512 registers, and what the allocator spills is the row state)
  constexpr int QC = lmpc_row_chunk(sizeof(real), KQ, KS);
  constexpr int SITES = lmpc_opaque_sites(sizeof(real), KQ, KS);
  // the predictor's backward sweep inside the factorisation (riccati_factor<.., FUSE>): the barrier weights then go to the rhs1 cells
  // and the predictor's gradient is assembled ahead of the factorisation.  Problems without the shared slack have one right-hand
  // side and keep the five-chain iteration (no shipped configuration: q_boundary > 0 everywhere).
  // (the warm-start kernels too: their cold path is this iteration -- except at two waves per SIMD, where the fused iteration is only
  //  trusted with the polish behind a call, lmpc_fuse_bwd's comment, and the warm kernels keep theirs inline)
  constexpr bool FUSEK = lmpc_fuse_bwd(sizeof(real), KQ, KS) && !(WARMK && lmpc_waves_per_simd(sizeof(real), KQ, KS) >= 2);
  const bool fuse = FUSEK && P.has_sigma != 0;
  Lds<real> L{lds, N, LEAN ? LMPC_LEAN_STAGE_STRIDE : LMPC_STAGE_STRIDE, lmpc_fresh_lane(sizeof(real), KQ, KS),
              !SECOND && lmpc_waves_per_simd(sizeof(real), KQ, KS) >= 2};
  if constexpr (SECOND && LMPC_CHAIN_PRIO) __builtin_amdgcn_s_setprio(3);
  real* T = L.tail();
  real* ct = T + TL_CT;
  real* KN0 = L.kn(0);
  // terminal region (learning only): treal cells behind the records; 16-byte aligned because every record size is even
  treal* const TT = reinterpret_cast<treal*>(T + LMPC_TAIL_DOUBLES);
  // lean layout: the chunk buffers sit behind everything else (records, tail, terminal region)
  ModelStream<real> MS{nullptr, nullptr, NS, lane, -1, -1};
  if constexpr (LEAN) {
    static_assert(!LEAN || (std::is_same<real, double>::value && std::is_same<io, double>::value), "the lean layout streams the fp64 workspace as it is");
    MS.ws = reinterpret_cast<const real*>(ws_lin) + (size_t)b * NS * LMPC_LIN_RECORD;
    MS.buf = reinterpret_cast<real*>(reinterpret_cast<unsigned char*>(TT) + (KS > 0 ? (LMPC_TERM_CELLS + 6 * LMPC_SS_STRIDE(P.S)) * sizeof(treal) : 0));
  }
  const treal tinf = treal(INFINITY);
  PT_DECL

  // ---------------- load: linearisation records, per-knot data, constant tables ----------------
  {
    if constexpr (!LEAN) {  // (the lean layout leaves the model in the workspace and streams it, sweep by sweep)
      const io* wsb = ws_lin + (size_t)b * NS * LMPC_LIN_RECORD;
      for (int e = lane; e < NS * LMPC_LIN_RECORD; e += 64) {
        const int i = e / LMPC_LIN_RECORD, o = e - i * LMPC_LIN_RECORD;
        const int c = o / 6;
        L.st(i)[o < 48 ? ST_ROW(c) + (o - c * 6) : ST_G + (o - 48)] = real(wsb[e]);
      }
    }
    for (int i = lane; i < NS; i += 64) L.st(i)[LEAN ? LN_DT : ST_DT] = real(T_ref[(size_t)i * B + b]);
    for (int i = lane; i < N; i += 64) {
      real* kn = L.kn(i);
      kn[KN_QLIN] = P.learning ? real(0) : real(i == N - 1 ? P.qv_term : P.qv_stage) * real(vref[(size_t)i * B + b]);
      kn[8] = 0.0;
      kn[9] = 0.0;
      kn[KN_BHL] = real(bl[(size_t)i * B + b]) - marg;
      kn[KN_BHL + 1] = real(br[(size_t)i * B + b]) + marg;
    }
    if (lane < 6) {
      KN0[lane] = real((lane == 0) ? x_ic[b] - s_shift : x_ic[(size_t)lane * B + b]);
      ct[CT_QD + lane] = P.learning ? 0.0 : P.Qd[lane];
      ct[CT_QT + lane] = P.learning ? 0.0 : P.Qt[lane];
      // (the abscissa box moves with the abscissa: single precision carries s relative to x_ic[0])
      ct[CT_HL + 2 * lane] = lane == 0 ? real(io(P.x_max[0]) - s_shift) : real(P.x_max[lane]);
      ct[CT_HL + 2 * lane + 1] = lane == 0 ? real(io(P.x_min[0]) - s_shift) : real(P.x_min[lane]);
    } else if (lane < 8) {
      KN0[lane] = real(u_ic[(size_t)(lane - 6) * B + b]);
      ct[CT_HL + 2 * lane] = P.u_hi[lane - 6];
      ct[CT_HL + 2 * lane + 1] = P.u_lo[lane - 6];
    } else if (lane < 10) {
      ct[CT_HL + 2 * lane] = P.v_hi[lane - 8];
      ct[CT_HL + 2 * lane + 1] = P.v_lo[lane - 8];
    } else if (lane < 14) {
      ct[CT_QU + lane - 10] = P.Qu[lane - 10];
    } else if (lane < 18) {
      ct[CT_SV + lane - 14] = P.Sv[lane - 14];
    } else if (lane == 18) {
      ct[CT_ZERO] = 0.0;
    } else if (lane < 25) {
      ct[CT_E + lane - 19] = P.chs2[lane - 19];
    }
  }
  wave_sync();
  PT_MARK(0)

  // ---------------- slot ownership ----------------
  // slot j = lane + 64 q  ->  knot i = j / 11, kind sl = j % 11.  Kinds 0..9 constrain the primal
  // component (z[0..7], v[0..1]) at offset sl of the knot record; kind 10 is the track boundary
  // row pair on e_y (offset 1) which also carries the shared slack sigma.  Everything a slot needs
  // is reduced to LDS offsets and flag bits here so that the per-iteration row code is branch-free
  // (its loads batch and the slots of a lane interleave): an absent row keeps t = 1, lam = 0 and a
  // clear flag; a lane's surplus slots (j >= 11 N) point at a dead record inside the factor's work
  // matrices, so their loads and stores are harmless.
  const bool has_sigma = P.has_sigma != 0;
  const int KNB = NS * L.stride;                     // knot 0, offset from the LDS base (doubles)
  const int JB = KNB + N * LMPC_KNOT_STRIDE + TL_W;  // dead record (34 cells of W)
  const int CTB = KNB + N * LMPC_KNOT_STRIDE + TL_CT;
  int o_val[KQ];  // the constrained value; its Newton step sits at +10 (rhs0) and +20 (rhs1)
  int o_hl[KQ];   // the row pair's bounds (hi, lo): box table for kinds 0..9, knot record for the boundary
  int s_gf[KQ];   // gradient recipe: ct index of the coefficient on the value | on its partner << 8 | (partner offset + 1) << 16,
                  // and flags << 20: F_UP / F_LO row exists, F_SIG boundary slot carrying sigma, F_QLIN takes the linear
                  // vx cost, F_MOVE owns a moving primal component, F_EY boundary slot, F_SCH boundary slot of a knot >= 1
  real s_tu[KQ], s_tl[KQ], s_lu[KQ], s_ll[KQ], s_pu[KQ], s_pl[KQ];
  real m_rows = 0.0;
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    const int j = lane + 64 * q;
    const bool valid = j < NSLOT * N;
    const int i = valid ? j / NSLOT : 0;
    const int sl = valid ? j - i * NSLOT : 0;
    const int kb = valid ? KNB + i * LMPC_KNOT_STRIDE : JB;
    o_val[q] = kb + (sl < SL_EY ? sl : 1);
    o_hl[q] = (valid && sl < SL_EY) ? CTB + CT_HL + 2 * sl : kb + KN_BHL;
    real hi = inf, lo = -inf;
    bool on = false;
    int ca = CT_ZERO, cb = CT_ZERO, pd = 0;
    if (valid) {
      if (sl < SL_EY) {
        hi = ct[CT_HL + 2 * sl];
        lo = ct[CT_HL + 2 * sl + 1];
        if (sl < SL_U) {
          on = i >= 1 && i <= N - 2;
          ca = (i == N - 1 ? CT_QT : CT_QD) + sl;
        } else {
          const bool uslot = sl < SL_V;
          const int k = uslot ? sl - SL_U : sl - SL_V;  // row of the 2x2 block (Qu on u, Sv on v)
          on = uslot ? i >= 1 : i <= N - 2;
          if (on) {
            ca = (uslot ? CT_QU : CT_SV) + 3 * k;        // diagonal entry
            cb = (uslot ? CT_QU : CT_SV) + 2 * k + 1 - k;  // [k][1-k]
          }
          pd = k == 0 ? 1 : -1;
        }
      } else {
        hi = real(bl[(size_t)i * B + b]) - marg;
        lo = real(br[(size_t)i * B + b]) + marg;
        on = has_sigma || i >= 1;
      }
    }
    const bool au = on && (hi < inf), al = on && (lo > -inf);
    const bool eys = valid && sl == SL_EY;
    const int fl = (au ? F_UP : 0) | (al ? F_LO : 0) | ((eys && has_sigma) ? F_SIG : 0) | ((valid && sl == 3) ? F_QLIN : 0) |
             ((valid && sl < SL_EY && (i >= 1 || sl >= SL_V)) ? F_MOVE : 0) | (eys ? F_EY : 0) | ((eys && i >= 1) ? F_SCH : 0);
    s_gf[q] = ca | (cb << 8) | ((pd + 1) << 16) | (fl << 20) | (sl << 27);  // (bits 27..30: the slot kind, for the polish's scaled step)
    m_rows += (au ? 1.0 : 0.0) + (al ? 1.0 : 0.0);
    s_tu[q] = s_tl[q] = 1.0;
    s_lu[q] = s_ll[q] = 0.0;
    s_pu[q] = s_pl[q] = 0.0;
  }
  // cell next to a boundary slot's weight that takes the sigma coupling coefficient (dead cell otherwise)
  auto flags = [&](int q) { return s_gf[q] >> 20; };
  // where the slot writes its barrier weight / gradient entry: the rhs0 cell of its component, or the boundary cell
  auto o_w = [&](int q) { return o_val[q] + ((flags(q) & F_EY) ? KN_EY - 1 : KN_R0); };
  auto o_csig = [&](int q) { return (flags(q) & F_EY) ? o_val[q] + (KN_CSIG - 1) : JB + KN_CSIG; };
  auto bounds = [&](int q) { return *reinterpret_cast<const real2*>(&lds[o_hl[q]]); };
  // The slot's three table entries as the row phases of the iteration read them.  Where lmpc_opaque_slots says so they pass
  // through an empty asm first: every LDS address of a slot is loop-invariant, so the optimiser hoists all of them out of the
  // iteration -- eight addresses per slot where the tables hold three integers --, the allocator spills them, and each use then
  // waits on its own scratch reload (`scratch_load_dword; s_waitcnt vmcnt(0); ds_read`: a gradient evaluation at KQ = 11 was ~60 of
  // those in a row).  Behind the asm the addresses are recomputed from the tables with a handful of integer instructions.
  struct SlotRef { int ov, oh, gf; };
  auto slot = [&](int q) {
    SlotRef r{o_val[q], o_hl[q], s_gf[q]};
    if constexpr (lmpc_opaque_slots(sizeof(real), KQ, KS)) asm volatile("" : "+v"(r.ov), "+v"(r.oh), "+v"(r.gf));
    return r;
  };
  // (the short-horizon tracking kernels keep the hoisted addresses -- the opaque tables cost them 1-3 % -- except at the sites of
  //  lmpc_opaque_sites, where the address is a select on the slot's flags: bit 0 the coupling cell in the row phase, 1 the rhs1 cell
  //  the gradient clears, 2 the coupling cell in the Schur sums, 3 the primal update)
  auto slot_at = [&](int q, int site) {
    SlotRef r{o_val[q], o_hl[q], s_gf[q]};
    if constexpr (lmpc_opaque_slots(sizeof(real), KQ, KS)) {
      asm volatile("" : "+v"(r.ov), "+v"(r.oh), "+v"(r.gf));
    } else if ((SITES >> site) & 1) {
      asm volatile("" : "+v"(r.gf));
    }
    return r;
  };
  auto r_flags = [](const SlotRef& r) { return r.gf >> 20; };
  auto r_w = [&](const SlotRef& r) { return r.ov + ((r_flags(r) & F_EY) ? KN_EY - 1 : KN_R0); };
  auto r_csig = [&](const SlotRef& r) { return (r_flags(r) & F_EY) ? r.ov + (KN_CSIG - 1) : JB + KN_CSIG; };
  auto r_bounds = [&](const SlotRef& r) { return *reinterpret_cast<const real2*>(&lds[r.oh]); };
  // ---------------- LMPC: simplex rows lambda_j >= 0, one safe-set point per lane and k < KS ----------------
  // (racing_mpc.cpp:484-504).  The points are centred on the first one (valid because 1'lambda = 1):
  // all sums below then run over O(1) offsets instead of absolute abscissae.
  const int S = P.S;
  SimplexRows<treal, KS> sx;
  treal lm_cold = 0.0;  // the cold start's simplex weight, 1 / (points kept)
  if constexpr (KS > 0) {
    treal* const UL = TT + TL_UL;
    sx.ul = UL;
    if (lane < 6) TT[TL_E + lane] = treal(P.chs2[lane]);
    // The safe set of this problem: as arrays (ss_x [6][S][B], ss_j [S][B]: lmpc_ss_query_batch wrote them), or BY REFERENCE
    // (P.ss_idx, round 5): S codes from lmpc_ss_query_idx_batch, each naming a row of the lap store and which of its three
    // unrolled copies; the point is read from the store (48 contiguous bytes, L2-resident: the whole store is < 0.2 MB) and its
    // cost-to-go recomputed from the row -- the same expressions as lmpc_ss_query_kernel uses, so both routes give the same bits.
    const bool by_ref = P.ss_idx != nullptr;
    auto ss_point = [&](int j, io (&pt)[6], io& jv) {  // point j of this problem's safe set (storage precision) and its J (not yet relative)
      const int code = P.ss_idx[(size_t)j * B + b];
      if (code < 0 || (code >> 2) >= P.ss_rows || (code & 3) == 3) {  // (no point; or not a code of this store: never read out of bounds)
#pragma unroll
        for (int k = 0; k < 6; ++k) pt[k] = io(0);
        jv = io(0);
        return;
      }
      const int row = code >> 2, rep = code & 3;
      int l = 0;
      for (int t = 1; t < P.ss_laps; ++t) l = (row >= P.ss_off[t]) ? t : l;
      const int n = P.ss_npts[l], jj = row - P.ss_off[l];
      const double* src = P.ss_store + (size_t)row * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) pt[k] = io(src[k] + (k == 0 ? (rep - 1) * P.ss_L : 0.0));
      jv = io((double)(n - 1 - jj) + (1 - rep) * (double)(n - 1));
    };
    io c0[6], j0v = io(0);  // the differences are formed in the storage precision, the abscissa relative to the shift
    if (by_ref) {
      ss_point(0, c0, j0v);
    } else {
#pragma unroll
      for (int k = 0; k < 6; ++k) c0[k] = ss_x[((size_t)k * S) * B + b];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) sx.ss0[k] = treal(k == 0 ? c0[k] - s_shift : c0[k]);
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int j = lane + 64 * q;
      sx.on[q] = j < S;
      sx.uz[q] = 6 * (sx.on[q] ? j : S);
      if (by_ref) {
        io pt[6], jv = io(0);
        if (sx.on[q]) {
          ss_point(j, pt, jv);
#pragma unroll
          for (int k = 0; k < 6; ++k) UL[6 * j + k] = treal(pt[k] - c0[k]);
        }
        sx.j[q] = sx.on[q] ? treal(jv - j0v) : treal(0);
      } else {
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (sx.on[q]) UL[6 * j + k] = treal(ss_x[((size_t)k * S + j) * B + b] - c0[k]);
        sx.j[q] = sx.on[q] ? treal(ss_j[(size_t)j * B + b]) : treal(0);
      }
      if (q == 0 && lane < 6) UL[6 * S + lane] = treal(0);  // the zero point
    }
    wave_fence();
    // A point that repeats the one before it is dropped (round 5): it adds nothing to the hull, and two free copies of one point
    // make the explicit points' system C_A singular -- the interior point's steps and the polish's multiplier steps then carry
    // noise of 1e-4.  The padding of a set with fewer than S points is such a run (racing_mpc.cpp:263-272 repeats the last point
    // S - n_found times): with 64 points found and 96 copies the answers were 5e-3 from the dense optimum, status OPTIMAL
    // (scratch/r5/pad_check.py; the twin has the same rule).  X, U, dU are those of the full set; a run's weight sits on its first
    // point, the copies report lambda = 0.
    real n_on = 0.0;
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      const int j = lane + 64 * q;
      bool dup = false;
      if (sx.on[q] && j > 0) {
        if (by_ref) {
          dup = P.ss_idx[(size_t)j * B + b] == P.ss_idx[(size_t)(j - 1) * B + b];
        } else {
          dup = ss_j[(size_t)j * B + b] == ss_j[(size_t)(j - 1) * B + b];
#pragma unroll
          for (int k = 0; k < 6; ++k) dup = dup && UL[6 * j + k] == UL[6 * (j - 1) + k];
        }
      }
      sx.on[q] = sx.on[q] && !dup;
      sx.uz[q] = 6 * (sx.on[q] ? j : S);
      sx.j[q] = sx.on[q] ? sx.j[q] : treal(0);
      n_on += sx.on[q] ? real(1) : real(0);
    }
    const treal inv_on = treal(1) / treal(uni(wave_sum(n_on)));
    lm_cold = inv_on;
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      sx.lm[q] = sx.on[q] ? inv_on : treal(0);
      sx.t[q] = sx.on[q] ? inv_on : treal(1);
      sx.l[q] = 0.0;
      sx.p[q] = 0.0;
      sx.dl[q] = 0.0;
      sx.aidx[q] = -1;
      m_rows += sx.on[q] ? 1.0 : 0.0;
    }
    treal umax = 0.0;  // largest u_j'E u_j of the (centred) points
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      treal uq[6], v = 0.0;
      sx.load_u(q, lane, uq);
#pragma unroll
      for (int k = 0; k < 6; ++k) v += treal(P.chs2[k]) * uq[k] * uq[k];
      umax = fmax(umax, v);
    }
    sx.tau = uni(treal(TAU_REL) * wave_max(umax));
    sx.m = 0;
  }
  const real m_tot = wave_sum(m_rows);
  const real inv_m = uni(real(1) / m_tot);

  // knot-0 feasibility: the state box applies to x_0 = x_ic (racing_mpc.cpp:147,201)
  bool feasible = true;
  {
    bool ok = true;
    if (lane < 6) {
      const real v = KN0[lane];
      ok = (v <= ct[CT_HL + 2 * lane]) && (v >= ct[CT_HL + 2 * lane + 1]);
    }
    if (lane == 6 && !has_sigma) {
      const real ey = KN0[1];
      ok = (ey <= bl[b] - marg) && (ey >= br[b] + marg);
    }
    feasible = wave_min(ok ? 1.0 : 0.0) > 0.5;
  }

  // The row sigma >= 0 (racing_mpc.cpp:536) is redundant and not carried: replacing a negative sigma by 0 loosens every
  // boundary row and lowers q_boundary sigma^2, so the QP without the row has the same optimum.  Carried, it is a
  // degenerate row whenever the boundary is inactive (sigma* = 0, multiplier 0) and costs two to three iterations.
  real sigma = 0.0;

  const real tau = 0.995, mu0 = 0.1, thr_frac = 0.5;
  int status = LMPC_SOLVE_MAX_ITER, it = 0;
  real mu = 0.0, rdmax = 0.0, rd_check = 0.0, last_step = 0.0, hsig = 0.0, ce = 0.0, mu_prev = inf;
  bool distress = false;
  const int max_iter = feasible ? P.max_iter : 0;
  typedef polish_limits<real> pol;
  const bool polish_on = P.polish >= 0;
  bool polished = false, pol_early_done = false, reentry = false, stall_moving = false;
  int pol_rounds = 0;
  // results: layout by strides (lmpc_set_output_layout): [component][knot][batch] by default -- what batch-parallel consumers
  // read coalesced -- or [batch][knot][component], one problem's plan contiguous (the reference's DM layout).  One code
  // path: element (k, i) of problem b at k sk + i si + b sb.
  auto put_primal = [&]() {
    const size_t xk = P.out_aos ? 1 : (size_t)N * B, xi = P.out_aos ? 6 : (size_t)B, xb = P.out_aos ? (size_t)6 * N : 1;
    const size_t uk = P.out_aos ? 1 : (size_t)NS * B, ui = P.out_aos ? 2 : (size_t)B, ub = P.out_aos ? (size_t)2 * NS : 1;
    for (int e = lane; e < 6 * N; e += 64) {
      const int k = e / N, i = e - k * N;
      X_out[k * xk + i * xi + b * xb] = io(L.kn(i)[k]) + (k == 0 ? s_shift : io(0));
    }
    for (int e = lane; e < 2 * NS; e += 64) {
      const int k = e / NS, i = e - k * NS;
      U_out[k * uk + i * ui + b * ub] = io(L.kn(i + 1)[6 + k]);
      dU_out[k * uk + i * ui + b * ub] = io(L.kn(i)[8 + k]);
    }
  };
  // The active-set polish is a call (lmpc_polish_call above): the interior point's row state goes in by value, an accepted
  // attempt returns the wave-wide scalars and the simplex weights of the polished point (the point itself is in LDS), a
  // refused one has put everything back.
  bool pol_noise = false;
  auto polish_attempt = [&](int max_rounds) -> bool {
    PolishArgs<real, KQ, KS> pa;
    pa.max_rounds = max_rounds;
    pa.keep = reinterpret_cast<io*>(P.save) + (size_t)b * (10 * N - 4);
    pa.ws = MS.ws;
    pa.N = N;
    pa.S = P.S;
    pa.has_sigma = P.has_sigma;
    pa.have0 = MS.have0;
    pa.have1 = MS.have1;
    pa.pol_rounds = pol_rounds;
    pa.qsig = qsig;
    pa.inv_m = inv_m;
    pa.sigma = sigma;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      pa.o_val[q] = o_val[q];
      pa.o_hl[q] = o_hl[q];
      pa.s_gf[q] = s_gf[q];
      pa.s_tu[q] = s_tu[q];
      pa.s_tl[q] = s_tl[q];
      pa.s_lu[q] = s_lu[q];
      pa.s_ll[q] = s_ll[q];
    }
    pa.sx = sx;
    PolishResult<real, KS> pr;
    // (the warm-start kernels at two waves per SIMD keep the polish inline: their accepted attempts ARE the polish, and behind a call the
    //  closed loop loses 3 .. 5 %: 6.65 -> 6.43 M car-steps/s at 4096 cars, 12.5 -> 11.9 M at 16384)
    if constexpr (lmpc_polish_is_call(sizeof(real), KQ, KS) && !(WARMK && lmpc_waves_per_simd(sizeof(real), KQ, KS) >= 2))
      pr = lmpc_polish_call<real, KQ, KS, io, SECOND>(pa);
    else
      pr = lmpc_polish<real, KQ, KS, io, SECOND>(pa);
    pol_rounds = uni(pr.pol_rounds);
    // what the interior point recomputes before it reads it again (predictor products -- multiplied by zero in the next
    // predictor pass --, the explicit-point bookkeeping, the Schur scalars) does not live across the polish
#pragma unroll
    for (int q = 0; q < KQ; ++q) s_pu[q] = s_pl[q] = 0.0;
    hsig = 0.0;
    ce = 0.0;
    if constexpr (KS > 0) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        sx.p[q] = 0.0;
        sx.dl[q] = 0.0;
        sx.aidx[q] = -1;
      }
      sx.m = 0;
      sx.r1 = 0.0;
    }
    if constexpr (LEAN) {
      MS.have0 = uni(pr.have0);
      MS.have1 = uni(pr.have1);
    }
    const bool accepted = uni(pr.accepted) != 0;
    pol_noise = uni(pr.noise) != 0;
    if (accepted) {
      sigma = uni(pr.sigma);
      mu = uni(pr.mu);
      rdmax = uni(pr.rdmax);
      last_step = uni(pr.last_step);
      if constexpr (KS > 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) sx.lm[q] = pr.lm[q];
      }
    }
    return accepted;
  };

  // ---------------- start point: minimiser of the cost over the dynamics alone ----------------
  // ... preceded, in a warm solve (lmpc_solve_batch_warm; fp64 tracking kernels), by an ACTIVE-SET attempt on the plan the caller
  // hands in -- the reference's X_optm_ref / U_optm_ref (racing_mpc.cpp:293-305), in the node the previous solution shifted by
  // one knot (racing_mpc_node.cpp:245-254).  Attempt 0: (1) the same zero-weight factorisation as the cold start; (2) the plan is
  // written into the rhs0 cells and rolled out under that factorisation's feedback, v = v^plan - K (z - z^plan): dynamically exact
  // about THIS linearisation, a few 1e-3 from the plan (which misses the new dynamics by the linearisation's change and x_0 by the
  // plant's step); (3) the working set is read off the PLAN: a polished optimum sits on its active bounds to rounding and the
  // boxes do not move with the shift (WARM_ACT); the boundary rows' bounds do, by the track's change over one knot (WARM_ACT_EY);
  // (4) the polish solves on that set, verifies the KKT conditions of this problem and repairs the set, WARM_ROUNDS times at
  // most.  Accepted: the optimum for about two iterations' worth of sweeps -- 92 .. 96 % of the periods of a closed loop at
  // N = 20 .. 60 on the serial twin (scratch/r5/warm_loop.py), iterations 6.5 -> 1.5 in the mean, answers those of the cold
  // solve to 1e-11.  Refused: attempt 1, the cold start, as if nothing had happened (the rounds spent are counted in `iters`).
  // The learning problem (round 6) takes the route with one more piece of the plan: the simplex weights of its terminal point --
  // the reference's convex_combi_optm_ref (set_initial(convex_combi_, ...), racing_mpc.cpp:281), P.warm_lam [S][B], aligned with
  // THIS call's safe-set points.  Their support is the working set of the simplex rows -- free where the plan's weight is positive,
  // held at zero elsewhere --, the weights themselves start the multiplier steps.  Without them there is nothing to start from
  // (all S weights free is more than the terminal block keeps explicit): the solve is cold.
  constexpr bool WARM_BUILT = WARMK && sizeof(real) == 8 && sizeof(io) == 8;
  const bool warm = WARM_BUILT && P.warm_X != nullptr && polish_on && feasible && (KS == 0 || P.warm_lam != nullptr);
  bool warm_done = false;
  for (int attempt = warm ? 0 : 1; attempt < 2 && !warm_done; ++attempt) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      lds[o_w(q)] = 0.0;
      lds[o_csig(q)] = 0.0;
    }
    if constexpr (KS > 0) {  // lambda frozen at 1/S: terminal cost eps'D eps only
      if (lane < 36) TT[TL_PT + lane] = (lane % 7 == 0) ? TT[TL_E + lane / 7] : treal(0);
    }
    wave_sync();
    if constexpr (LEAN)
      riccati_factor_lean<(KS > 0), false>(L, MS, lane, TT + TL_PT);
    else
      riccati_factor<(KS > 0), false>(L, lane, TT + TL_PT);
    if constexpr (WARM_BUILT) {
      if (attempt == 0) {
        // the plan in the solver's variables into the rhs0 cells: z_i = [x_i; u_{i-1}] (knot 0: the measured state and the applied
        // input), v_i = (u_i - u_{i-1}) / t_i
        const double* const WX = P.warm_X;
        const double* const WU = P.warm_U;
        for (int e = lane; e < 10 * N; e += 64) {
          const int i = e / 10, o = e - 10 * i;
          real val = 0.0;
          if (o < 6) {
            val = i == 0 ? KN0[o] : real(WX[((size_t)o * N + i) * B + b]);
          } else if (o < 8) {
            val = i == 0 ? KN0[o] : real(WU[((size_t)(o - 6) * NS + (i - 1)) * B + b]);
          } else if (i < NS) {
            const real up = i == 0 ? KN0[o - 2] : real(WU[((size_t)(o - 8) * NS + (i - 1)) * B + b]);
            val = (real(WU[((size_t)(o - 8) * NS + i) * B + b]) - up) / real(T_ref[(size_t)i * B + b]);
          }
          L.kn(i)[KN_R0 + o] = val;
        }
        wave_sync();
        if constexpr (LEAN)
          feedback_rollout_lean<true>(L, MS, lane);
        else
          feedback_rollout<true>(L, lane);
        // the plan's boundary slack, then its working set (the interior point's row state is what the polish classifies by:
        // a held row gets lam = 1 > t = 0, every other row lam = 0 <= t = its slack)
        real vw[KQ];
        real2 hlw[KQ];
        real sgmax = 0.0;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          vw[q] = lds[o_val[q] + KN_R0];
          hlw[q] = bounds(q);
          const int f = flags(q);
          if (f & F_SIG) sgmax = fmax(sgmax, fmax((f & F_UP) ? vw[q] - hlw[q].x : real(0), (f & F_LO) ? hlw[q].y - vw[q] : real(0)));
        }
        sigma = uni(wave_max(sgmax));
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int f = flags(q);
          const real sg = (f & F_SIG) ? sigma : real(0), tolw = (f & F_EY) ? real(WARM_ACT_EY) : real(WARM_ACT);
          const real su = hlw[q].x + sg - vw[q], sd = vw[q] + sg - hlw[q].y;
          const bool hu = (f & F_UP) && su <= tolw, hd = (f & F_LO) && sd <= tolw;
          s_tu[q] = hu ? real(0) : fmax(su, real(0));
          s_tl[q] = hd ? real(0) : fmax(sd, real(0));
          s_lu[q] = hu ? real(1) : real(0);
          s_ll[q] = hd ? real(1) : real(0);
          s_tu[q] = (f & F_UP) ? s_tu[q] : real(1);
          s_tl[q] = (f & F_LO) ? s_tl[q] : real(1);
        }
        bool lam_ok = true;
        if constexpr (KS > 0) {  // the simplex rows' working set and start from the plan's weights (copies of a dropped run: ignored)
          treal lw[KS], lsum = 0.0;
#pragma unroll
          for (int q = 0; q < KS; ++q) {
            const int j = lane + 64 * q;
            lw[q] = sx.on[q] ? fmax(treal(P.warm_lam[(size_t)(j < S ? j : 0) * B + b]), treal(0)) : treal(0);
            lsum += lw[q];
          }
          lsum = uni(wave_sum(lsum));
          lam_ok = lsum > treal(0);
          const treal inv = lam_ok ? treal(1) / lsum : treal(0);
#pragma unroll
          for (int q = 0; q < KS; ++q) {
            const treal lm = lw[q] * inv;
            const bool fr = sx.on[q] && lm > treal(WARM_ACT);
            sx.lm[q] = lm;
            sx.t[q] = sx.on[q] ? (fr ? lm : treal(0)) : treal(1);  // (the polish classifies by l > t: held)
            sx.l[q] = sx.on[q] ? (fr ? treal(0) : treal(1)) : treal(0);
            sx.p[q] = 0.0;
            sx.dl[q] = 0.0;
            sx.aidx[q] = -1;
          }
        }
        wave_sync();
        if (lam_ok && polish_attempt(P.warm_rounds > 0 ? P.warm_rounds : WARM_ROUNDS)) {
          polished = true;
          status = LMPC_SOLVE_OPTIMAL;
          warm_done = true;
        } else {  // refused: the cold start sets everything up again (row state: at the end of its Newton step)
          sigma = 0.0;
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            s_tu[q] = s_tl[q] = 1.0;
            s_lu[q] = s_ll[q] = 0.0;
          }
          if constexpr (KS > 0) {
#pragma unroll
            for (int q = 0; q < KS; ++q) {
              sx.lm[q] = sx.on[q] ? lm_cold : treal(0);
              sx.t[q] = sx.on[q] ? lm_cold : treal(1);
              sx.l[q] = 0.0;
              sx.p[q] = 0.0;
              sx.dl[q] = 0.0;
              sx.aidx[q] = -1;
            }
            sx.m = 0;
          }
        }
        continue;
      }
    }
    if constexpr (LEAN)
      feedback_rollout_lean(L, MS, lane);
    else
      feedback_rollout(L, lane);
  }
  if constexpr (KS > 0) {
    treal ul[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < KS; ++q) {
      treal uq[6];
      sx.load_u(q, lane, uq);
#pragma unroll
      for (int k = 0; k < 6; ++k) ul[k] += uq[k] * sx.lm[q];
    }
    wave_sum_split<6>(ul, lane);
    if (lane < 6) {
      treal e = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (k == lane) e = (treal(L.kn(N - 1)[k]) - sx.ss0[k]) - ul[k];
      TT[TL_EPS + lane] = e;
      TT[TL_TG + lane] = TT[TL_E + lane] * e;
    }
    wave_sync();
  }
  PT_MARK(1)

  // it == -1 is the start-point Newton step (all row weights zero, full step); it >= 0 the
  // interior-point iterations.  The iterations are the INNER loop; a polish attempt (a call) sits between two runs of it
  // (outer loop: at most twice round).
  it = -1;
  for (; !warm_done;) {
  int hand_over = 0;  // why the interior point stopped: 0 for good (status says why), 1 the early polish attempt, 2 the one at its exit,
                      // 3 out of iterations but close (single precision): a last attempt
  for (; it <= max_iter; ++it) {
    const bool ipm = it >= 0;
    // ======== gradient: cost gradient + row coefficients, written by the component owner (into the rhs0 cells) ========
    // (a function of the iterate and of (sigma_c mu, pm) only: with the fused factorisation the predictor's is assembled BEFORE the
    //  factorisation, which consumes it; the rhs1 cells then hold the barrier weights and are not cleared here)
    treal eeps[6] = {0, 0, 0, 0, 0, 0};  // E eps of this iterate (safe-set block), in scalar registers
    auto load_eeps = [&]() {
      if constexpr (KS > 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) eeps[k] = uni(TT[TL_E + k] * TT[TL_EPS + k]);
      }
    };
    real sgsum0 = 0.0;
    auto gradient = [&](const int pass, const real smu, const real pm) -> real {
      const bool fused = FUSEK && fuse && ipm;
      real sgsum = 0.0;
      if constexpr (KS > 0) {
        if (ipm) {
          PT_MARK(12)
          // right-hand side r_j = -bl_j (dx = 0): sums over the eliminated points, the explicit ones through LDS
          treal bs[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < KS; ++q) {
            treal itf, uq[6];
            sx.load_u(q, lane, uq);
            const treal rj = sx.on[q] ? -simplex_bl(sx.lm[q], sx.t[q], sx.l[q], sx.p[q], sx.j[q], uq, treal(smu), treal(pm), eeps, itf) : treal(0);
            if (sx.aidx[q] >= 0) TT[TL_RA + sx.aidx[q]] = rj;
            const treal w = (sx.on[q] && sx.aidx[q] < 0) ? rj * itf : treal(0);
            bs[6] += w;
#pragma unroll
            for (int k = 0; k < 6; ++k) bs[k] += uq[k] * w;
          }
          wave_sum_split<7>(bs, lane);
          wave_fence();
          PT_MARK(13)
          treal beta[6], h[6], nu;
#pragma unroll
          for (int k = 0; k < 6; ++k) beta[k] = bs[k];
          term_solve_u(TT, lane, sx.m, beta, bs[6], sx.r1, h, nu);
          PT_MARK(14)
          if (lane < 6) {  // terminal gradient onto x_T: E eps + pT, pT = -h
            treal hs = h[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) hs = (lane == k) ? h[k] : hs;
            TT[TL_TG + lane] = TT[TL_E + lane] * TT[TL_EPS + lane] - hs;
          }
          wave_fence();
          PT_MARK(15)
        }
      }
#pragma unroll
      for (int q0 = 0; q0 < KQ; q0 += QC) {
        real val[QC], par[QC], ca[QC], cb[QC], ql[QC];
        real2 hl[QC];
        SlotRef sr[QC];
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
          const int q = q0 + qq;
          if (q >= KQ) continue;
          sr[qq] = slot(q);
          const int gr = sr[qq].gf, ov = sr[qq].ov;
          val[qq] = lds[ov];
          hl[qq] = r_bounds(sr[qq]);
          par[qq] = lds[ov + ((gr >> 16) & 3) - 1];
          ca[qq] = lds[CTB + (gr & 0xff)];
          cb[qq] = lds[CTB + ((gr >> 8) & 0xff)];
          ql[qq] = lds[ov + (KN_QLIN - 3)];
        }
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
          const int q = q0 + qq;
          if (q >= KQ) continue;
          const int f = r_flags(sr[qq]);
          const real sg = (f & F_SIG) ? sigma : 0.0;
          const real itu = frcp(s_tu[q]), itl = frcp(s_tl[q]);
          real cu = s_lu[q] * itu * (val[qq] - sg + s_tu[q] - hl[qq].x) + (smu - pm * s_pu[q]) * itu;
          real cd = s_ll[q] * itl * (-val[qq] - sg + s_tl[q] + hl[qq].y) + (smu - pm * s_pl[q]) * itl;
          cu = (ipm && (f & F_UP)) ? cu : 0.0;
          cd = (ipm && (f & F_LO)) ? cd : 0.0;
          const real g = ca[qq] * val[qq] + cb[qq] * par[qq] + ((f & F_QLIN) ? ql[qq] : real(0));  // (zero coefficients on a boundary slot)
          lds[r_w(sr[qq])] = g + cu - cd;
          if (pass == 0 && !fused) {
            const SlotRef sz = ((SITES & 2) && !lmpc_opaque_slots(sizeof(real), KQ, KS)) ? slot_at(q, 1) : sr[qq];
            lds[(r_flags(sz) & F_EY) ? JB + KN_EY : r_w(sz) + 10] = 0.0;
          }
          sgsum += (f & F_SIG) ? (cu + cd) : real(0);
        }
        if constexpr (QC < KQ) ISSUE_ORDER();
      }
      wave_sync();
      PT_MARK(12)
      for (int i = lane; i < N; i += 64) {
        real* kn = L.kn(i);
        kn[KN_R0 + 1] += kn[KN_EY];
        if (pass == 0 && !fused) kn[KN_R1 + 1] = (i >= 1) ? kn[KN_CSIG] : 0.0;
      }
      if constexpr (KS > 0) {  // terminal gradient of the safe-set block onto x_T (lanes 0..5 != EY lanes' cells)
        wave_sync();
        if (lane < 6) L.kn(N - 1)[KN_R0 + lane] += real(TT[TL_TG + lane]);
      }
      wave_sync();
      return sgsum;
    };
    // ======== rows: complementarity, residual, barrier weights ========
    if (ipm) {
      real musum = 0.0, rdl = 0.0, eysum = 0.0;
#pragma unroll
      for (int q0 = 0; q0 < KQ; q0 += QC) {  // (QC slots at a time: the whole row in one go where the registers allow it)
        real val[QC];
        real2 hl[QC];
        SlotRef sr[QC];
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
          const int q = q0 + qq;
          if (q >= KQ) continue;
          sr[qq] = slot(q);
          val[qq] = lds[sr[qq].ov];
          hl[qq] = r_bounds(sr[qq]);
        }
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
          const int q = q0 + qq;
          if (q >= KQ) continue;
          const int f = r_flags(sr[qq]);
          const real sg = (f & F_SIG) ? sigma : 0.0;
          const real thu = s_lu[q] * frcp(s_tu[q]), thd = s_ll[q] * frcp(s_tl[q]);
          musum += s_lu[q] * s_tu[q] + s_ll[q] * s_tl[q];
          rdl = fmax(rdl, (f & F_UP) ? fabs(val[qq] - sg + s_tu[q] - hl[qq].x) : real(0));
          rdl = fmax(rdl, (f & F_LO) ? fabs(-val[qq] - sg + s_tl[q] + hl[qq].y) : real(0));
          lds[r_w(sr[qq]) + (fuse ? ((f & F_EY) ? KN_TEY - KN_EY : KN_R1 - KN_R0) : 0)] = thu + thd;
          lds[r_csig(((SITES & 1) && !lmpc_opaque_slots(sizeof(real), KQ, KS)) ? slot_at(q, 0) : sr[qq])] = (f & F_SIG) ? (thd - thu) : 0.0;
          eysum += (f & F_SIG) ? (thu + thd) : real(0);
        }
        if constexpr (QC < KQ) ISSUE_ORDER();
      }
      if constexpr (KS > 0) {
        PT_MARK(2)
        // ---- which points stay explicit this iteration: the (at most MA_MAX) smallest theta below tau ----
        treal thq[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          thq[q] = sx.on[q] ? sx.l[q] * frcp(sx.t[q]) : tinf;
          sx.aidx[q] = -1;
        }
        if (lane < 6 * MA_MAX) TT[TL_UA + lane] = 0.0;            // unused slots: u = 0, theta = 1, rhs = 0
        if (lane < MA_MAX) {
          TT[TL_THA + lane] = 1.0;
          TT[TL_RA + lane] = 0.0;
        }
        wave_fence();
        int m = 0;
        for (int a = 0; a < MA_MAX; ++a) {
          treal cand = tinf;
          int cq = 0;
#pragma unroll
          for (int q = 0; q < KS; ++q) {
            const bool better = sx.aidx[q] < 0 && thq[q] < sx.tau && thq[q] < cand;
            cand = better ? thq[q] : cand;
            cq = better ? q : cq;
          }
          const treal best = wave_min(cand);
          if (!(best < tinf)) break;
          const int owner = __ffsll((long long)__ballot(cand == best)) - 1;  // the lowest lane holding the minimum
          if (lane == owner) {
#pragma unroll
            for (int q = 0; q < KS; ++q)
              if (q == cq) {
                sx.aidx[q] = a;
                treal uq[6];
                sx.load_u(q, lane, uq);
#pragma unroll
                for (int k = 0; k < 6; ++k) TT[TL_UA + a * 6 + k] = uq[k];
                TT[TL_THA + a] = thq[q];
              }
          }
          wave_fence();
          m = a + 1;
        }
        sx.m = m;
        wave_fence();
        // ---- sums over the eliminated points: T_B (21), a_B (6), s_B; and over all points: U lambda (6), sum lambda ----
        treal tt[21], av[14];
#pragma unroll
        for (int k = 0; k < 21; ++k) tt[k] = 0.0;
#pragma unroll
        for (int k = 0; k < 14; ++k) av[k] = 0.0;
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          const treal on = sx.on[q] ? 1.0 : 0.0;
          const treal itf = (sx.on[q] && sx.aidx[q] < 0) ? frcp(thq[q]) : treal(0);
          musum += real(on * sx.l[q] * sx.t[q]);
          rdl = fmax(rdl, real(on * fabs(-sx.lm[q] + sx.t[q])));
          int n = 0;
          treal uq[6];
          sx.load_u(q, lane, uq);
#pragma unroll
          for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = r; c < 6; ++c) tt[n++] += uq[r] * uq[c] * itf;
            av[r] += uq[r] * itf;
            av[7 + r] += uq[r] * sx.lm[q];
          }
          av[6] += itf;
          av[13] += sx.lm[q];
        }
        {  // 35 sums: 32 through the splitting butterfly, the last three on their own
          treal red[32], red3[3] = {av[11], av[12], av[13]};
#pragma unroll
          for (int k = 0; k < 21; ++k) red[k] = tt[k];
#pragma unroll
          for (int k = 0; k < 11; ++k) red[21 + k] = av[k];
          wave_sum_split<32>(red, lane);
          wave_sum_split<3>(red3, lane);
#pragma unroll
          for (int k = 0; k < 21; ++k) tt[k] = red[k];
#pragma unroll
          for (int k = 0; k < 11; ++k) av[k] = red[21 + k];
          av[11] = red3[0];
          av[12] = red3[1];
          av[13] = red3[2];
        }
        sx.r1 = 1.0 - av[13];
        {
          treal F[36], aB[6];
          int n = 0;
#pragma unroll
          for (int r = 0; r < 6; ++r) {
#pragma unroll
            for (int c = r; c < 6; ++c) {
              F[r * 6 + c] = tt[n];
              F[c * 6 + r] = tt[n];
              ++n;
            }
            aB[r] = av[r];
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) F[k * 6 + k] += 1.0 / fmax(TT[TL_E + k], treal(1e-30));  // (a zero weight: that component of eps is free)
          term_factor_u(TT, lane, F, aB, av[6], m);
        }
        if (lane < 6) {
          treal e = 0.0;
#pragma unroll
          for (int k = 0; k < 6; ++k)
            if (k == lane) e = (treal(L.kn(N - 1)[k]) - sx.ss0[k]) - av[7 + k];
          TT[TL_EPS + lane] = e;
        }
        wave_fence();
      }
      {
        real red[2] = {musum, eysum};
        wave_sum_n<2>(red);
        musum = red[0];
        hsig = qsig + red[1];
      }
      rdmax = wave_max(rdl);
      hsig = uni(hsig);
      mu = uni(musum * inv_m);
      // (see the step-length rule; far from feasibility mu may rise legitimately: IAC at 60 m/s into a corner)
      // (reentry: the same iterate a second time -- after a refused polish, or on the way to the final one: none of the
      // bookkeeping repeats)
      if (!reentry && it >= 1 && mu >= mu_prev && rdmax <= lim::rd_distress) distress = true;
      mu_prev = mu;
      const bool again_here = reentry;
      reentry = false;
      if (!(mu == mu) || !(rdmax == rdmax)) {
        status = LMPC_SOLVE_INFEASIBLE;
        break;
      }
      if (mu <= tol && rdmax <= lim::rd_ok) {
        status = LMPC_SOLVE_OPTIMAL;
        hand_over = polish_on ? 2 : 0;
        break;
      }
      // primal infeasibility: on a feasible problem the row residual contracts by (1 - alpha) per iteration; not
      // losing a tenth over five iterations while still large (step lengths stuck below ~2 %) ends the solve (this
      // also bounds the straggler that would otherwise hold its CU slot for max_iter iterations).  "Not halved" is
      // too tight: feasible problems with a slow start (IAC at 60 m/s into a corner) contract by 0.6-0.8 per five.
      if (!again_here && it % 5 == 0) {
        if (it >= 10 && rdmax > lim::rd_infeasible && rdmax > real(0.9) * rd_check) {
          status = LMPC_SOLVE_INFEASIBLE;
          break;
        }
        rd_check = rdmax;
      }
      if (it == max_iter) {
        // single precision, out of iterations with the rows (nearly) feasible: the complementarity of an fp32 recursion can
        // hover just above its floor for good (seen on one IAC problem of 8192 that the fp64 kernel solves in 9 iterations).
        // The polish does not care how the interior point stopped -- it verifies what it returns -- so it gets the iterate;
        // refused, the status stays MAX_ITER (and the two-pass entries hand the problem to the fp64 kernel).
        if (sizeof(real) == 4 && polish_on && max_iter > 0 && mu <= real(1e-3) && rdmax <= real(10) * lim::rd_ok) hand_over = 3;
        break;
      }
      // the early attempt: the active set is usually settled two iterations before the interior point's own tolerance
      if constexpr (pol::early) {
        if (polish_on && !pol_early_done && mu <= real(pol::mu_early) && rdmax <= real(pol::rd_early)) {
          pol_early_done = true;
          hand_over = 1;
          break;
        }
      }
      wave_sync();
      PT_MARK(2)
      if constexpr (FUSEK) {
        if (fuse) {
          load_eeps();
          sgsum0 = gradient(0, real(0), real(0));
          PT_MARK(4)
        }
      }
      if constexpr (LEAN) {
        if (mu <= real(JOSEPH_MU)) {
          if (FUSEK && fuse)
            riccati_factor_lean<(KS > 0), true, FUSEK>(L, MS, lane, TT + TL_PT, KN_R1, KN_TEY);
          else
            riccati_factor_lean<(KS > 0), true>(L, MS, lane, TT + TL_PT);
        } else {
          if (FUSEK && fuse)
            riccati_factor_lean<(KS > 0), false, FUSEK>(L, MS, lane, TT + TL_PT, KN_R1, KN_TEY);
          else
            riccati_factor_lean<(KS > 0), false>(L, MS, lane, TT + TL_PT);
        }
      } else if (sizeof(real) == 8 && mu <= real(JOSEPH_MU)) {  // (single precision stops at mu ~ 2e-6)
        if (FUSEK && fuse)
          riccati_factor<(KS > 0), (sizeof(real) == 8), FUSEK>(L, lane, TT + TL_PT, KN_R1, KN_TEY);
        else
          riccati_factor<(KS > 0), (sizeof(real) == 8)>(L, lane, TT + TL_PT);
      } else {
        if (FUSEK && fuse)
          riccati_factor<(KS > 0), false, FUSEK>(L, lane, TT + TL_PT, KN_R1, KN_TEY);
        else
          riccati_factor<(KS > 0), false>(L, lane, TT + TL_PT);
      }
      PT_MARK(3)
    }

    real sigc = 0.0, alpha = 1.0, dsigma = 0.0;
    bool numerics_failed = false, stalled = false;
    if (!(FUSEK && fuse && ipm)) load_eeps();
    real d_val[KQ];
    const int npass = ipm ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
      const real smu = (pass == 1) ? sigc * mu : 0.0, pm = (pass == 1) ? 1.0 : 0.0;
      real sgsum = sgsum0;  // sum of boundary-row coefficients entering the sigma gradient
      if (!(FUSEK && fuse && ipm && pass == 0)) {
        sgsum = gradient(pass, smu, pm);
        PT_MARK(4)
      }
      // ======== Newton step: predictor together with the Schur vector, then the corrector ========
      if constexpr (LEAN) {
        if (pass == 0 && ipm && has_sigma) {
          if (FUSEK && fuse)
            riccati_solve_lean_dpp<2, FUSEK>(L, MS, lane, pf);
          else
            riccati_solve_lean_dpp<2>(L, MS, lane, pf);
        } else
          riccati_solve_lean_dpp<1>(L, MS, lane, pf);
      } else {
        if (pass == 0 && ipm && has_sigma) {
          if (FUSEK && fuse)
            riccati_solve<2, FUSEK>(L, lane, pf);
          else
            riccati_solve<2>(L, lane, pf);
        } else
          riccati_solve<1>(L, lane, pf);
      }
      PT_MARK(5)
      // ======== step of every constrained value; boundary slack by Schur complement ========
      // (QC == KQ: the steps and values are loaded once, here, and stay in registers across the reduction; otherwise chunk-wise,
      //  once for the Schur sums and again for the row steps)
      real dz0[QC == KQ ? KQ : 1], dz1[QC == KQ ? KQ : 1], val[QC == KQ ? KQ : 1];
      real2 hl[QC == KQ ? KQ : 1];
      if constexpr (QC == KQ) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          dz0[q] = lds[o_val[q] + 10];
          dz1[q] = lds[o_val[q] + 20];
          val[q] = lds[o_val[q]];
          hl[q] = bounds(q);
        }
      }
      if (!ipm) {
#pragma unroll
        for (int q = 0; q < KQ; ++q) d_val[q] = QC == KQ ? dz0[QC == KQ ? q : 0] : lds[slot(q).ov + 10];
        break;
      }
      if (has_sigma) {
        real red[3] = {0.0, 0.0, sgsum};  // c'dz (this rhs), c'e (Schur vector), sum of boundary coefficients
        if constexpr (QC == KQ) {
          real cs[KQ];
#pragma unroll
          for (int q = 0; q < KQ; ++q) cs[q] = lds[r_csig(slot_at(q, 2))];
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            const bool sch = (flags(q) & F_SCH) != 0;
            red[0] += sch ? cs[q] * dz0[q] : real(0);
            red[1] += sch ? cs[q] * dz1[q] : real(0);
          }
        } else {
#pragma unroll
          for (int q0 = 0; q0 < KQ; q0 += QC) {
            real cs[QC], e0[QC], e1[QC];
            SlotRef sr[QC];
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
              const int q = q0 + qq;
              if (q >= KQ) continue;
              sr[qq] = slot(q);
              cs[qq] = lds[r_csig(sr[qq])];
              e0[qq] = lds[sr[qq].ov + 10];
              e1[qq] = lds[sr[qq].ov + 20];
            }
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
              const int q = q0 + qq;
              if (q >= KQ) continue;
              const bool sch = (r_flags(sr[qq]) & F_SCH) != 0;
              red[0] += sch ? cs[qq] * e0[qq] : real(0);
              red[1] += sch ? cs[qq] * e1[qq] : real(0);
            }
            ISSUE_ORDER();
          }
        }
        wave_sum_n<3>(red);
        if (pass == 0) ce = red[1];
        const real qsg = qsig * sigma - red[2];
        dsigma = uni(-(qsg + red[0]) / (hsig + ce));
      }
      if constexpr (KS > 0) {
        // r_j = u_j'E dx_T - bl_j;  d lambda_j = (r_j - nu - u_j'h)/theta_j for the eliminated points, the explicit
        // ones from the small dense solve
        const real* knT = L.kn(N - 1);
        treal e[6], gs[7] = {0, 0, 0, 0, 0, 0, 0}, rj[KS], itfq[KS];
#pragma unroll
        for (int k = 0; k < 6; ++k) e[k] = TT[TL_E + k] * treal(knT[KN_R0 + k] + dsigma * knT[KN_R1 + k]);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal itf, uq[6];
          sx.load_u(q, lane, uq);
          treal r = -simplex_bl(sx.lm[q], sx.t[q], sx.l[q], sx.p[q], sx.j[q], uq, treal(smu), treal(pm), eeps, itf);
#pragma unroll
          for (int k = 0; k < 6; ++k) r += uq[k] * e[k];
          r = sx.on[q] ? r : 0.0;
          rj[q] = r;
          itfq[q] = itf;
          if (sx.aidx[q] >= 0) TT[TL_RA + sx.aidx[q]] = r;
          const treal w = (sx.on[q] && sx.aidx[q] < 0) ? r * itf : treal(0);
          gs[6] += w;
#pragma unroll
          for (int k = 0; k < 6; ++k) gs[k] += uq[k] * w;
        }
        wave_sum_split<7>(gs, lane);
        wave_fence();
        treal beta[6], h[6], nu;
#pragma unroll
        for (int k = 0; k < 6; ++k) beta[k] = gs[k];
        term_solve_u(TT, lane, sx.m, beta, gs[6], sx.r1, h, nu);
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal uh = 0.0, uq[6];
          sx.load_u(q, lane, uq);
#pragma unroll
          for (int k = 0; k < 6; ++k) uh += uq[k] * h[k];
          const treal dB = (rj[q] - nu - uh) * itfq[q];
          const treal dA = TT[TL_XA + (sx.aidx[q] >= 0 ? sx.aidx[q] : 0)];
          sx.dl[q] = sx.on[q] ? (sx.aidx[q] >= 0 ? dA : dB) : treal(0);
        }
      }
      // ======== row steps; largest feasible step as 1 / max(1, max -dt/t, max -dlam/lam) ========
      auto row_step = [&](bool on, treal t, treal lam, treal pprod, treal rd, treal cdy, treal& dt_, treal& dl_,
                          treal& it_) {  // (the simplex rows: always fp64)
        it_ = frcp(t);
        dt_ = on ? (-rd - cdy) : 0.0;
        dl_ = on ? (-lam + (treal(smu) - treal(pm) * pprod) * it_ - lam * it_ * dt_) : 0.0;
      };
      real dtu[KQ], dlu[KQ], dtl[KQ], dll[KQ];
      real rmax = 1.0;
      bool finite_step = true;
#pragma unroll
      for (int q0 = 0; q0 < KQ; q0 += QC) {
        real e0[QC], e1[QC], ev[QC];
        real2 eh[QC];
        SlotRef sr[QC];
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
          const int q = q0 + qq;
          if (q >= KQ) continue;
          sr[qq] = slot(q);
          if constexpr (QC == KQ) {
            e0[qq] = dz0[q], e1[qq] = dz1[q], ev[qq] = val[q], eh[qq] = hl[q];
          } else {
            e0[qq] = lds[sr[qq].ov + 10];
            e1[qq] = lds[sr[qq].ov + 20];
            ev[qq] = lds[sr[qq].ov];
            eh[qq] = r_bounds(sr[qq]);
          }
        }
#pragma unroll
        for (int qq = 0; qq < QC; ++qq) {
          const int q = q0 + qq;
          if (q >= KQ) continue;
          const int f = r_flags(sr[qq]);
          const real dval = e0[qq] + dsigma * e1[qq];
          d_val[q] = dval;
          finite_step = finite_step && (fabs(dval) < inf);
          const real sg = (f & F_SIG) ? sigma : 0.0, dsg = (f & F_SIG) ? dsigma : 0.0;
          const real itu = frcp(s_tu[q]), itl = frcp(s_tl[q]);
          const real a = (f & F_UP) ? -(ev[qq] - sg + s_tu[q] - eh[qq].x) - (dval - dsg) : 0.0;
          const real bq = (f & F_UP) ? -s_lu[q] + (smu - pm * s_pu[q]) * itu - s_lu[q] * itu * a : 0.0;
          const real c = (f & F_LO) ? -(-ev[qq] - sg + s_tl[q] + eh[qq].y) - (-dval - dsg) : 0.0;
          const real d = (f & F_LO) ? -s_ll[q] + (smu - pm * s_pl[q]) * itl - s_ll[q] * itl * c : 0.0;
          dtu[q] = a;
          dlu[q] = bq;
          dtl[q] = c;
          dll[q] = d;
          rmax = fmax(rmax, fmax(-a * itu, -bq * frcp(fmax(s_lu[q], lim::tiny))));
          rmax = fmax(rmax, fmax(-c * itl, -d * frcp(fmax(s_ll[q], lim::tiny))));
        }
        if constexpr (QC < KQ) ISSUE_ORDER();
      }
      if constexpr (KS > 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal dt_, dl_, it_;
          row_step(sx.on[q], sx.t[q], sx.l[q], sx.p[q], -sx.lm[q] + sx.t[q], -sx.dl[q], dt_, dl_, it_);
          rmax = fmax(rmax, real(fmax(-dt_ * it_, -dl_ * frcp(fmax(sx.l[q], treal(1e-300))))));
        }
      }
      rmax = wave_max(finite_step ? rmax : inf);
      if (!(rmax < inf) || !(dsigma == dsigma)) {
        // a Newton step that is not a number (the Schur complement of sigma or the 2x2 H cancelled completely -- in
        // practice single precision on its last iteration): keep the iterate, report it by what it has reached
        numerics_failed = true;
        break;
      }
      const real amax = uni(real(1) / rmax);
      if (pass == 1) {
        alpha = uni(fmin(real(1), tau * amax));
        if (sizeof(real) == 8 && KQ <= 7 && distress) {  // (fp64 arithmetic, N <= 40: see below)
          // A problem whose complementarity has gone UP once gets the wide-neighbourhood rule from then on: the step is
          // cut back until no complementarity product falls below NBHD_GAMMA times their mean.  Mehrotra's iteration can
          // otherwise leave the neighbourhood of the central path and cycle -- seen on a learning problem whose safe set
          // offers two nearly exchangeable points: products at 0.01 and 300 times mu, mu bouncing between 6e-6 and 2e-5
          // up to the iteration cap while the dense solver finds the optimum (19 iterations with the rule).  Problems
          // whose mu falls monotonically (all but a few per thousand) never enter this branch.  The single-precision
          // instantiations do without it: their mu is noisy at its floor, and the branch costs the mixed learning kernel
          // 280 B of scratch per lane at two waves per SIMD (3.13 -> 2.89 M solves/s).  Nor do the N > 40 instantiations,
          // which already spill their row state: the branch costs them another 260 B per lane (N = 60: +9 % time).
          for (int trial = 0; trial < NBHD_TRIALS; ++trial) {
            real sl = 0.0, pmin = inf;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
              const int f = flags(q);
              const real pu = (s_tu[q] + alpha * dtu[q]) * (s_lu[q] + alpha * dlu[q]);
              const real pl = (s_tl[q] + alpha * dtl[q]) * (s_ll[q] + alpha * dll[q]);
              sl += pu + pl;  // (an absent row has lam = 0, d lam = 0: no contribution)
              pmin = fmin(pmin, fmin((f & F_UP) ? pu : inf, (f & F_LO) ? pl : inf));
            }
            if constexpr (KS > 0) {
#pragma unroll
              for (int q = 0; q < KS; ++q) {
                treal dt_, dl_, it_;
                row_step(sx.on[q], sx.t[q], sx.l[q], sx.p[q], -sx.lm[q] + sx.t[q], -sx.dl[q], dt_, dl_, it_);
                const real pr = real((sx.t[q] + treal(alpha) * dt_) * (sx.l[q] + treal(alpha) * dl_));
                sl += sx.on[q] ? pr : real(0);
                pmin = fmin(pmin, sx.on[q] ? pr : inf);
              }
            }
            if (wave_min(pmin) >= real(NBHD_GAMMA) * wave_sum(sl) * inv_m) break;
            alpha = uni(alpha * real(0.6));
          }
        }
      }
      real sacc = 0.0;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        if (pass == 0) {
          sacc += (s_tu[q] + amax * dtu[q]) * (s_lu[q] + amax * dlu[q]) + (s_tl[q] + amax * dtl[q]) * (s_ll[q] + amax * dll[q]);
          s_pu[q] = dtu[q] * dlu[q];
          s_pl[q] = dtl[q] * dll[q];
        } else {
          s_tu[q] += alpha * dtu[q];
          s_lu[q] += alpha * dlu[q];
          s_tl[q] += alpha * dtl[q];
          s_ll[q] += alpha * dll[q];
          sacc += s_tu[q] * s_lu[q] + s_tl[q] * s_ll[q];
        }
      }
      if constexpr (KS > 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) {
          treal dt_, dl_, it_;
          row_step(sx.on[q], sx.t[q], sx.l[q], sx.p[q], -sx.lm[q] + sx.t[q], -sx.dl[q], dt_, dl_, it_);
          if (pass == 0) {
            sacc += sx.on[q] ? real((sx.t[q] + treal(amax) * dt_) * (sx.l[q] + treal(amax) * dl_)) : real(0);
            sx.p[q] = dt_ * dl_;
          } else {
            sx.t[q] += treal(alpha) * dt_;
            sx.l[q] += treal(alpha) * dl_;
            sx.lm[q] += treal(alpha) * sx.dl[q];
            sacc += sx.on[q] ? real(sx.t[q] * sx.l[q]) : real(0);
          }
        }
      }
      if (pass == 1) {
        // no further progress: rows feasible, complementarity already small, and the corrector step would not lower it
        // (the Newton direction has reached the accuracy of the factorisation): keep the current primal iterate
        sacc = wave_sum(sacc);
        if (rdmax <= lim::rd_ok && mu <= real(STALL_MU) && sacc * inv_m >= mu) stalled = true;
      }
      if (pass == 0) {
        sacc = wave_sum(sacc);
        const real ratio = (sacc * inv_m) / mu;
        sigc = uni(ratio * ratio * ratio);
        wave_sync();
      }
    }

    if (numerics_failed) {
      status = (mu <= real(10) * tol && rdmax <= lim::rd_ok) ? LMPC_SOLVE_OPTIMAL : LMPC_SOLVE_MAX_ITER;
      // (what the interior point has reached is polished like a converged iterate; single precision short of its tolerance
      // but close gets the last attempt of the out-of-iterations exit above: refused, it stays MAX_ITER)
      hand_over = (status == LMPC_SOLVE_OPTIMAL && polish_on) ? 2 : 0;
      if (sizeof(real) == 4 && status != LMPC_SOLVE_OPTIMAL && polish_on && mu <= real(1e-3) && rdmax <= real(10) * lim::rd_ok) hand_over = 3;
      break;
    }
    if (stalled) {
      // (round 6: a stall is not convergence when the Newton step it declines would still move the iterate -- the point is kept and handed to the
      //  polish as before, but unless the polish verifies it the status is MAX_ITER: oracle/c/lmpc_oracle.c, STALL_STEP, has the problem)
      real cand = 0.0;  // the step the stall declines to take, in the reference's scaled units (like the polish's steps)
#pragma unroll
      for (int q = 0; q < KQ; ++q) cand = fmax(cand, (flags(q) & F_MOVE) ? fabs(alpha * d_val[q]) * real(slot_inv_scale((s_gf[q] >> 27) & 15)) : real(0));
      stall_moving = sizeof(real) == 8 && wave_max(cand) > real(STALL_STEP);
      status = (stall_moving && !polish_on) ? LMPC_SOLVE_MAX_ITER : LMPC_SOLVE_OPTIMAL;
      hand_over = polish_on ? 2 : 0;
      break;
    }
    // ======== primal update by the component owners ========
    PT_MARK(6)
    real stepmax = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const SlotRef sr = slot_at(q, 3);
      const bool mv = (r_flags(sr) & F_MOVE) != 0;
      const real dz = mv ? alpha * d_val[q] : 0.0;
      lds[mv ? sr.ov : JB + q] += dz;
      stepmax = fmax(stepmax, fabs(dz));
    }
    wave_sync();
    if (ipm) {
      last_step = wave_max(stepmax);
      if (has_sigma) sigma = uni(sigma + alpha * dsigma);
    } else {
      // ---- slacks and multipliers at the start point: t = max(slack, 0.5 range), lam = mu0 / t ----
      real val[KQ];
      real2 hl[KQ];
      int fl[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const SlotRef sr = slot(q);
        val[q] = lds[sr.ov];
        hl[q] = r_bounds(sr);
        fl[q] = r_flags(sr);
      }
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        real range = ((fl[q] & (F_UP | F_LO)) == (F_UP | F_LO)) ? (hl[q].x - hl[q].y) : 1.0;
        if (!(range > real(1e-3))) range = real(1e-3);
        const real thr = thr_frac * range;
        if (fl[q] & F_UP) {
          s_tu[q] = fmax(hl[q].x - val[q], thr);
          s_lu[q] = mu0 / s_tu[q];
        }
        if (fl[q] & F_LO) {
          s_tl[q] = fmax(val[q] - hl[q].y, thr);
          s_ll[q] = mu0 / s_tl[q];
        }
      }
      sigma = 0.0;
      if constexpr (KS > 0) {
#pragma unroll
        for (int q = 0; q < KS; ++q) sx.l[q] = sx.on[q] ? treal(mu0) / sx.t[q] : treal(0);
      }
    }
  }
  if (hand_over == 0) break;
  wave_sync();
  if (polish_attempt(pol::rounds | (hand_over >= 2 ? POLISH_EXIT : 0))) {
    polished = true;
    status = LMPC_SOLVE_OPTIMAL;
    break;
  }
  if (hand_over == 3) break;  // refused after the last iteration: out of iterations it is
  if (hand_over == 2 || !pol::early) {  // refused at the exit: the interior point's own answer stands
    status = LMPC_SOLVE_OPTIMAL;
    // ... unless the refusal itself says that answer cannot be trusted to the contract (fp64, round 5).  A held set that is
    // CONSISTENT -- no negative multiplier, no violated row -- whose multiplier steps nevertheless do not settle (last step
    // above step_tol, or a held row not met to `feas`, after four steps) is a problem whose linear algebra is noisier than
    // 1e-6: seen at vx < 0.6 m/s with N >= 36, where the RK4 step map's spectral radius is ~25 per stage (DESIGN.md section
    // 3, "where it has to converge") and the steps bounce at 1e-4.  The interior point's iterate, computed with the same
    // sweeps, was then 1e-6 .. 2e-3 from the dense optimum with status OPTIMAL (tests/dispatch_sweep.py, one problem of
    // 1024 at eighteen horizons).  It now says LMPC_SOLVE_MAX_ITER: stopped short of the stated accuracy.  No problem of
    // the bench distributions takes this branch (twin, 4096 / 1024 problems per family).
    // (round 6: likewise a STALLED iterate whose declined step would still have moved it by more than STALL_STEP)
    if (sizeof(real) == 8 && (pol_noise || stall_moving)) status = LMPC_SOLVE_MAX_ITER;
    break;
  }
  reentry = true;  // refused early: the interior point goes on from the same iterate (same `it`; the rows phase puts its
                   // barrier weights back into the records)
  }
  PT_MARK(7)
  if (it < 0) it = 0;
  it += pol_rounds;  // (a polish round costs about what an iteration does and is counted as one)
  if (!feasible) status = LMPC_SOLVE_INFEASIBLE;
  if constexpr (KS > 0) {
    // hard hull equality (the penalty limit, LMPC_HARD_HULL_WEIGHT): a terminal state the hull cannot reach leaves a
    // residual the weight does not close -- the reference's QP is infeasible there
    if (P.hard_hull && status == LMPC_SOLVE_OPTIMAL) {
      wave_sync();
      treal worst = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) worst = fmax(worst, fabs(TT[TL_EPS + k]) * treal(slot_inv_scale(k)));
      if (worst > treal(LMPC_HARD_HULL_RESIDUAL)) status = LMPC_SOLVE_INFEASIBLE;
    }
  }
  // two-pass mixed precision: an answer the fp32 iteration could not verify -- or did not reach: out of iterations, or
  // "infeasible" by single-precision residuals -- is marked for the fp64 kernel behind it, which has the last word
  if (P.flag_unverified && (status != LMPC_SOLVE_OPTIMAL || (polish_on && !polished))) status = LMPC_SOLVE_UNVERIFIED;

  // ---------------- write back: X [6][N][B], U, dU [2][N-1][B] ----------------
  wave_sync();
  put_primal();
  if constexpr (KS > 0) {
    if (lam_out) {
#pragma unroll
      for (int q = 0; q < KS; ++q)
        if (lane + 64 * q < S) lam_out[(size_t)(lane + 64 * q) * B + b] = sx.on[q] ? io(sx.lm[q]) : io(0);  // (a dropped copy: 0)
    }
  }
  if (lane == 0) {
    status_out[b] = status;
    iters_out[b] = it;
    if constexpr (WARMK) {
      if (P.warm_flag) P.warm_flag[b] = warm_done ? 1 : 0;  // (explicit: until round 6 callers inferred it from iters <= 4)
    }
#ifdef LMPC_PHASE_TIMING
    if (kkt_out) {
      if constexpr (LEAN) pf.acc[5] = MS.waited;  // (cycles inside ModelStream::wait of the interior point's own sweeps -- also counted in their phases)
      for (int k = 0; k < 16; ++k) kkt_out[k * (size_t)B + b] = (io)pf.acc[k];
      kkt_out[16 * (size_t)B + b] = (io)pf.w0;               // 100 MHz wall clock at start
      kkt_out[17 * (size_t)B + b] = (io)wall_clock64();      // ... at end
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      kkt_out[18 * (size_t)B + b] = (io)hwid;
      kkt_out[19 * (size_t)B + b] = (io)(xcc & 0xf);
    }
#else
    if (kkt_out) {
      kkt_out[0 * (size_t)B + b] = io(last_step);
      kkt_out[1 * (size_t)B + b] = io(rdmax);
      kkt_out[2 * (size_t)B + b] = io(mu);
      kkt_out[3 * (size_t)B + b] = io(sigma);
    }
#endif
  }
}

template <typename real, int KQ, int KS, typename io>
__global__ __launch_bounds__(64, lmpc_waves_per_simd(sizeof(real), KQ, KS)) void lmpc_solve_kernel(
    lmpc_params P, int B, const io* __restrict__ ws_lin, const io* __restrict__ x_ic,
    const io* __restrict__ u_ic, const io* __restrict__ T_ref, const io* __restrict__ bl,
    const io* __restrict__ br, const io* __restrict__ vref, const io* __restrict__ ss_x,
    const io* __restrict__ ss_j, io* __restrict__ lam_out, io* __restrict__ X_out,
    io* __restrict__ U_out, io* __restrict__ dU_out, int* __restrict__ status_out,
    int* __restrict__ iters_out, io* __restrict__ kkt_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];  // one symbol for every instantiation
  // XCD-aware problem assignment: consecutive workgroups go round-robin to the 8 XCDs (each with its own L2), while
  // consecutive problems share 64-byte lines in the [field][knot][batch] arrays.  Workgroup w takes problem
  // (w mod 8) * ceil(B / 8) + w / 8, so each XCD owns a contiguous eighth of the batch and the 8-byte strided
  // accesses of neighbouring problems merge in that XCD's L2 instead of reaching HBM as partial lines.
  // With a launch order (longest job first from the previous solve's iteration counts, lmpc_set_launch_order) workgroup
  // w takes problem launch_order[w]: the hardware starts workgroups in index order, so the long problems go first and the
  // short ones fill the tail of the second residency round.
  int b = (int)(blockIdx.x & 7) * ((B + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (P.launch_order) {
    const int n = P.order_count ? *P.order_count : B;  // (a list: the workgroups past its end have nothing to do)
    b = (int)blockIdx.x < n ? P.launch_order[blockIdx.x] : B;
  }
  if (b >= B) return;
  lmpc_solve_problem<real, KQ, KS, io>(P, B, b, lds_raw, ws_lin, x_ic, u_ic, T_ref, bl, br, vref, ss_x, ss_j, lam_out, X_out, U_out,
                                       dU_out, status_out, iters_out, kkt_out);
}

// Second pass of a mixed-precision solve (lmpc_solve_batch_mixed): the problems the fp32 iteration marked
// LMPC_SOLVE_UNVERIFIED, solved in fp64.  `list` [count] holds them (lmpc_collect_unverified_kernel); a small fixed grid
// of workgroups takes list entries in turn -- launching one workgroup per problem of the batch to have 99 % of them
// return at once costs more than the solves (38 KB of LDS and 500 registers to allocate per workgroup: 1.3 ms per 8192).
// The solve is a CALL here, not inlined into the loop: with the 3000-line body inlined under a loop the <double, 7, 3>
// instance computed garbage (nondeterministically; the same body without the loop, or called once per workgroup, is bit
// for bit the direct kernel -- scratch/r3_cleanup_dbg.py); behind a call boundary every instance is.  Round 4 met the same
// family deterministically in <double, 7, 0> and bisected it ("the register-starved instantiations":
// the answers of these 460-to-512-register functions depend on how the compiler parks spilled scalars in VGPR lanes; wait
// counts, post-RA scheduling and the machine verifier are ruled out): the noinline works around exactly that -- a call
// boundary is the one thing that has kept every build of this family right.  The callee names the workgroup's dynamic
// LDS block itself, so its accesses stay in the LDS address space.
template <typename real, int KQ, int KS, typename io>
__device__ __attribute__((noinline)) void lmpc_solve_problem_call(
    const lmpc_params& P, const int B, const int b, const io* __restrict__ ws_lin, const io* __restrict__ x_ic,
    const io* __restrict__ u_ic, const io* __restrict__ T_ref, const io* __restrict__ bl, const io* __restrict__ br,
    const io* __restrict__ vref, const io* __restrict__ ss_x, const io* __restrict__ ss_j, io* __restrict__ lam_out,
    io* __restrict__ X_out, io* __restrict__ U_out, io* __restrict__ dU_out, int* __restrict__ status_out,
    int* __restrict__ iters_out, io* __restrict__ kkt_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  lmpc_solve_problem<real, KQ, KS, io, true>(P, B, b, lds_raw, ws_lin, x_ic, u_ic, T_ref, bl, br, vref, ss_x, ss_j, lam_out, X_out, U_out,
                                             dU_out, status_out, iters_out, kkt_out);
}

template <typename real, int KQ, int KS, typename io>
__global__ __launch_bounds__(64, lmpc_waves_per_simd(sizeof(real), KQ, KS)) void lmpc_cleanup_kernel(
    lmpc_params P, int B, const int* __restrict__ list, const int* __restrict__ count, const io* __restrict__ ws_lin,
    const io* __restrict__ x_ic, const io* __restrict__ u_ic, const io* __restrict__ T_ref, const io* __restrict__ bl,
    const io* __restrict__ br, const io* __restrict__ vref, const io* __restrict__ ss_x,
    const io* __restrict__ ss_j, io* __restrict__ lam_out, io* __restrict__ X_out,
    io* __restrict__ U_out, io* __restrict__ dU_out, int* __restrict__ status_out,
    int* __restrict__ iters_out, io* __restrict__ kkt_out) {
  const int n = __builtin_amdgcn_readfirstlane(*count);
  for (int w = blockIdx.x; w < n; w += gridDim.x) {
    const int b = __builtin_amdgcn_readfirstlane(list[w]);  // (wave-uniform: keep the problem index in a scalar register)
    lmpc_solve_problem_call<real, KQ, KS, io>(P, B, b, ws_lin, x_ic, u_ic, T_ref, bl, br, vref, ss_x, ss_j, lam_out, X_out, U_out, dU_out,
                                              status_out, iters_out, kkt_out);
    wave_fence();
  }
}

// The warm-start solve (lmpc_solve_batch_warm) is a kernel of its own: compiled into lmpc_solve_kernel, the attempt's code changed
// the register allocation of the cold path -- 992 -> 1296 B of scratch per lane at KQ = 11, 1796 -> 2128 B at KQ = 14, 132 more
// spilled scalars in the headline kernel -- for callers that never pass a plan.  Same body, WARMK = true.
template <int KQ, int KS>
__global__ __launch_bounds__(64, lmpc_waves_per_simd(8, KQ, KS)) void lmpc_solve_warm_kernel(
    lmpc_params P, int B, const double* __restrict__ ws_lin, const double* __restrict__ x_ic, const double* __restrict__ u_ic,
    const double* __restrict__ T_ref, const double* __restrict__ bl, const double* __restrict__ br, const double* __restrict__ vref,
    const double* __restrict__ ss_x, const double* __restrict__ ss_j, double* __restrict__ lam_out,
    double* __restrict__ X_out, double* __restrict__ U_out, double* __restrict__ dU_out, int* __restrict__ status_out,
    int* __restrict__ iters_out, double* __restrict__ kkt_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  int b = (int)(blockIdx.x & 7) * ((B + 7) >> 3) + (int)(blockIdx.x >> 3);  // (as lmpc_solve_kernel)
  if (P.launch_order) {
    const int n = P.order_count ? *P.order_count : B;
    b = (int)blockIdx.x < n ? P.launch_order[blockIdx.x] : B;
  }
  if (b >= B) return;
  lmpc_solve_problem<double, KQ, KS, double, false, true>(P, B, b, lds_raw, ws_lin, x_ic, u_ic, T_ref, bl, br, vref, ss_x, ss_j, lam_out, X_out,
                                                          U_out, dU_out, status_out, iters_out, kkt_out);
}

#include "lmpc_solve_w2.hip.h"  // two wavefronts per problem: the fp64 tracking kernels for N >= 24 (round 6)
#define LMPC_W2_SIGNATURE(KQ)                                                                                            \
  __global__ void lmpc_solve_kernel_w2<KQ>(lmpc_params, int, const double*, const double*, const double*, const double*, \
                                           const double*, const double*, const double*, double*, double*, double*, int*, int*, double*);

#define LMPC_INSTANTIATE(REAL, KQ, KS, IO)                                                                              \
  template __global__ void lmpc_solve_kernel<REAL, KQ, KS, IO>(lmpc_params, int, const IO*, const IO*, const IO*,        \
                                                                const IO*, const IO*, const IO*, const IO*, const IO*,    \
                                                                const IO*, IO*, IO*, IO*, IO*, int*, int*, IO*);
#ifdef LMPC_SINGLE_INSTANCE  // (ISA inspection: hipcc -S -DLMPC_SINGLE_INSTANCE="double, 4, 3, double")
#define LMPC_INSTANTIATE_X(...) LMPC_INSTANTIATE(__VA_ARGS__)
LMPC_INSTANTIATE_X(LMPC_SINGLE_INSTANCE)
#elif defined(LMPC_MINREG_TU)  // lmpc_lib_minreg.hip: the two kernels that translation unit exists for
LMPC_INSTANTIATE(float, 4, 2, double)
LMPC_INSTANTIATE(float, 4, 3, double)
#elif defined(LMPC_W2_TU)  // lmpc_lib_w2.hip: the two-wave kernels (a translation unit of their own: they rebuild in a minute)
template LMPC_W2_SIGNATURE(7)
template LMPC_W2_SIGNATURE(11)
template LMPC_W2_SIGNATURE(14)
#else
extern template LMPC_W2_SIGNATURE(7)
extern template LMPC_W2_SIGNATURE(11)
extern template LMPC_W2_SIGNATURE(14)
#define LMPC_INSTANTIATE_WARM(KQ, KS)                                                                                    \
  template __global__ void lmpc_solve_warm_kernel<KQ, KS>(lmpc_params, int, const double*, const double*, const double*, const double*,   \
                                                          const double*, const double*, const double*, const double*, const double*, double*, \
                                                          double*, double*, double*, int*, int*, double*);
LMPC_INSTANTIATE_WARM(2, 0)
LMPC_INSTANTIATE_WARM(4, 0)
LMPC_INSTANTIATE_WARM(7, 0)
LMPC_INSTANTIATE_WARM(11, 0)
LMPC_INSTANTIATE_WARM(14, 0)
// the learning problem's warm start (round 6): the horizons the reference ships for it (barc_lmpc N = 40, iac_car_lmpc N = 60) and
// BASELINE's N = 20, with 96 (KS = 2) and 160 (KS = 3) points
LMPC_INSTANTIATE_WARM(4, 2)
LMPC_INSTANTIATE_WARM(4, 3)
LMPC_INSTANTIATE_WARM(7, 2)
LMPC_INSTANTIATE_WARM(7, 3)
LMPC_INSTANTIATE_WARM(11, 2)
LMPC_INSTANTIATE_WARM(11, 3)
LMPC_INSTANTIATE(double, 2, 0, double)
LMPC_INSTANTIATE(double, 4, 0, double)
LMPC_INSTANTIATE(double, 7, 0, double)
LMPC_INSTANTIATE(double, 11, 0, double)
LMPC_INSTANTIATE(double, 14, 0, double)
LMPC_INSTANTIATE(double, 4, 2, double)
LMPC_INSTANTIATE(double, 4, 3, double)
LMPC_INSTANTIATE(double, 7, 2, double)
LMPC_INSTANTIATE(double, 7, 3, double)
LMPC_INSTANTIATE(double, 11, 2, double)  // iac_car_lmpc.param.yaml ships N = 60
LMPC_INSTANTIATE(double, 11, 3, double)
LMPC_INSTANTIATE(double, 14, 2, double)
LMPC_INSTANTIATE(double, 14, 3, double)
LMPC_INSTANTIATE(float, 4, 0, float)
LMPC_INSTANTIATE(float, 7, 0, float)
LMPC_INSTANTIATE(float, 11, 0, float)
LMPC_INSTANTIATE(float, 14, 0, float)
// mixed: fp32 interior-point iteration between fp64 arrays (io = double); the third kernel type, `treal` -- the simplex rows
// and the terminal elimination of the learning problem -- is double in every instantiation: F = D^-1 + U Theta^-1 U' has a
// condition number ~1e8 late in the iteration, which fp32 cannot carry
LMPC_INSTANTIATE(float, 4, 0, double)
LMPC_INSTANTIATE(float, 7, 0, double)
LMPC_INSTANTIATE(float, 11, 0, double)  // iac_car_tracking_mpc.param.yaml ships N = 80
LMPC_INSTANTIATE(float, 14, 0, double)
// the learning problem, N <= 23 (BASELINE configs[4]): instantiated in a translation unit of their own (lmpc_lib_minreg.hip),
// which is compiled with -mllvm -amdgpu-sched-strategy=iterative-minreg -- the most spill-bound kernel of the library is the one
// place where the minimum-register scheduler pays (131 against 184 spilled VGPRs, 9.93 against 10.65 ms per 32768, same bits;
// every other kernel is 3-20 % slower with it: profiles/r04_sched_strategies.md).  Here: declarations only.
extern template __global__ void lmpc_solve_kernel<float, 4, 2, double>(lmpc_params, int, const double*, const double*, const double*, const double*,
    const double*, const double*, const double*, const double*, const double*, double*, double*, double*, double*, int*, int*, double*);
extern template __global__ void lmpc_solve_kernel<float, 4, 3, double>(lmpc_params, int, const double*, const double*, const double*, const double*,
    const double*, const double*, const double*, const double*, const double*, double*, double*, double*, double*, int*, int*, double*);
// the fp64 second pass behind each of the mixed kernels above
#define LMPC_INSTANTIATE_CLEANUP(KQ, KS)                                                                                  \
  template __global__ void lmpc_cleanup_kernel<double, KQ, KS, double>(lmpc_params, int, const int*, const int*,          \
      const double*, const double*, const double*, const double*, const double*, const double*, const double*,           \
      const double*, const double*, double*, double*, double*, double*, int*, int*, double*);
LMPC_INSTANTIATE_CLEANUP(4, 0)
LMPC_INSTANTIATE_CLEANUP(7, 0)
LMPC_INSTANTIATE_CLEANUP(11, 0)
LMPC_INSTANTIATE_CLEANUP(14, 0)
LMPC_INSTANTIATE_CLEANUP(4, 2)
LMPC_INSTANTIATE_CLEANUP(4, 3)
// (the learning problem at N = 40 in mixed precision was built and measured: 1.16 M solves/s against 0.70 M in fp64, but
//  median 1.2e-3 / 99th percentile 1.5e-2 from the fp64 answers -- outside the 1e-3 the mixed entry states; not shipped)
#endif
