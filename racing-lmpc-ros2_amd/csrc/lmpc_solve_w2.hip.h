// lmpc_solve_w2.hip.h -- TWO wavefronts per problem on one LDS record: the fp64 tracking kernels for N >= 24 (round 6).
// (Included by lmpc_solve_kernel.hip: every device function of that file -- Riccati factorisation and sweeps, the lean model stream,
//  the reductions -- is used as it is.  A translation unit of its own, lmpc_lib_w2.hip: it rebuilds in twenty seconds.)
//
// Why.  With one wave per problem the slots of the inequality rows are dealt to 64 lanes: 7 per lane at N = 40, 11 at N = 60, 14 at
// N = 80.  A slot is 6 doubles of row state (slacks and multipliers of both sides, the predictor's products) + three table integers
// + the temporaries of whichever row phase is running, so those kernels take the whole register file of a SIMD (512 registers: one
// wave per SIMD) and from N = 41 on spill what does not fit: 992 B of scratch per lane at N = 60, 1792 B at N = 80 -- 28x / 113x the
// algorithmic HBM traffic (profiles/r05_pmc_n60.json).  Here the workgroup is 128 threads: the slots are dealt to 128 lanes (4 / 6 /
// 7 per lane), the kernel is compiled for 256 registers -- two waves per SIMD, i.e. the SAME number of problems per CU as before (LDS
// decides that: 4 at N <= 64, 3 beyond) with twice the waves to run their row phases.
//
// What the second wave does NOT do: the Riccati factorisation and the four vector sweeps of an iteration are one dependent chain
// per problem; they run on ONE wave as in the one-wave kernel while the other waits at the workgroup barrier.  So the chain phases --
// 62 % of the one-wave kernel at N = 60 -- cost what they did (measured, phase clocks of the chain wave: factor 47.0 k cycles per
// iteration against 44.1 k, the four sweeps 118 k against 116 k); what can change is the row phases' share.
//
// Measured (MI355X, 4096 problems, kernel ms; profiles/r06_w2_variants.txt, profiles/r06_fuse_ab.txt):
//                                        N = 24    40     60     80    IAC 40   IAC 80     scratch B / lane (KQ = 7 / 11 / 14)
//   one wave per problem (round 5)        1.63    2.47   5.37   11.44   4.28    13.53      424 /  992 / 1792
//   two waves, five chains per iteration  1.70    2.55   5.37    9.55   4.24    11.38      316 /  740 /  996
//   ... slots dealt unevenly (chain wave 3 / 4 / 4, the other 4 / 7 / 10)
//                                         1.69    2.50   5.67   11.32   4.19    13.27      284 /  896 / 1632
//   ... without the opaque slot tables                   6.23   10.64                      676 / 1276 / 1364
//   ... the chain role by hardware placement (HW_ID) instead of wave 0: no difference (6.48 / 11.10)
//   one wave, fused factorisation         1.56    2.35   5.00    9.70   4.05    11.41      424 / 1008 / 1792
//   two waves, fused (LMPC_W2_FUSE, shipped) 1.65  2.47   4.85    8.06   4.15     9.54      316 /  644 /  820
// (fused: the predictor's backward sweep inside the factorisation, riccati_factor<.., FUSE> -- four dependent chains per iteration
//  instead of five.)  The library takes these kernels from N = 41 on -- every lean-layout horizon: -1 .. 4 % up to N = 64, -19 % at N = 65
//  (6.59 -> 5.32 ms: where the one-wave kernel goes from 11 to 14 slots per lane), -17 % at N = 80 -- lmpc_capi.hip: LMPC_W2_AUTO_KQ;
// lmpc_set_waves_per_problem forces either.  Answers: the one-wave kernel's to 1e-11, same statuses, same iteration counts on every
// problem of the six batches.
// Why not more: the chain wave still carries its row state across a Riccati stage that wants ~200 registers of its own, and the
// chains are 75 % of the kernel.  VERDICT r5's targets (N = 60 <= 3.2 ms, N = 80 <= 5.5 ms) need the chain itself shortened further
// (DESIGN.md section 8 has the arithmetic of what was considered).
//
// Exchange between the waves: LDS + s_barrier (wg_sync).  Every wave-wide scalar of the one-wave kernel (mu, the step lengths, the
// Schur sums, the polish's votes) becomes a two-step reduction -- DPP inside each wave, then both partial results through two LDS
// cells, combined in wave order by BOTH waves, so that the two waves hold bit-identical scalars and take every branch together.
#ifndef LMPC_SOLVE_W2_HIP_H_
#define LMPC_SOLVE_W2_HIP_H_

#define W2_THREADS 128
#ifndef LMPC_W2_FUSE  // 1 (shipped): four dependent chains per iteration instead of five (N = 80: 9.55 -> ms, profiles/r06_fuse_ab.txt)
#define LMPC_W2_FUSE 1
#endif
// slots per lane with 128 lanes: KQ is the one-wave kernel's (7: N <= 40, 11: N <= 64, 14: N <= 81)
#ifndef LMPC_W2_SPLIT  // 1 (shipped): the slots dealt evenly, 4 / 6 / 7 per lane; 0: unevenly (the chain wave 3 / 4 / 4, the other 4 / 7 / 10)
#define LMPC_W2_SPLIT 1
#endif
#ifndef LMPC_W2_ROLE  // 0 (shipped): the chain phases on wave 0 of every workgroup; 1: on the wave the hardware placed on the SIMD whose parity is its wave slot's
#define LMPC_W2_ROLE 0
#endif
constexpr int lmpc_w2_slots0(int kq) { return LMPC_W2_SPLIT ? (kq <= 7 ? 4 : (kq <= 11 ? 6 : 7)) : (kq <= 7 ? 3 : 4); }  // the chain wave
constexpr int lmpc_w2_slots1(int kq) {  // the other wave: 64 (KL0 + KL1) >= 11 N at the class's largest N
  return LMPC_W2_SPLIT ? (kq <= 7 ? 4 : (kq <= 11 ? 6 : 7)) : (kq <= 7 ? 4 : (kq <= 11 ? 7 : 10));
}
static_assert(64 * (lmpc_w2_slots0(7) + lmpc_w2_slots1(7)) >= 11 * 40 && 64 * (lmpc_w2_slots0(11) + lmpc_w2_slots1(11)) >= 11 * 64 &&
              64 * (lmpc_w2_slots0(14) + lmpc_w2_slots1(14)) >= 11 * 81, "every slot has an owner");

struct Wg2 {
  double* red;  // LDS: [2 parities][2 waves][8 values] -- cells of the factorisation's Y matrix, idle outside the chain phases
  int wv;       // this wave: 0 / 1
  int par;      // parity of the next exchange (alternating buffers: one barrier per reduction)
};
__device__ __forceinline__ void wg_sync() { __syncthreads(); }

template <class OP, int NV>
__device__ __forceinline__ void wg_reduce_n(Wg2& g, double (&v)[NV]) {
  static_assert(NV <= 8, "eight values per exchange");
  wave_reduce_n<OP, NV>(v);  // (wave-uniform)
  double* cell = g.red + g.par * 16;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) cell[g.wv * 8 + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = uni(OP::f(cell[k], cell[8 + k]));  // wave 0's share first, in both waves: the same bits
  g.par ^= 1;
}
__device__ __forceinline__ double wg_sum(Wg2& g, double x) {
  double v[1] = {x};
  wg_reduce_n<op_sum, 1>(g, v);
  return v[0];
}
__device__ __forceinline__ double wg_max(Wg2& g, double x) {
  double v[1] = {x};
  wg_reduce_n<op_max, 1>(g, v);
  return v[0];
}
__device__ __forceinline__ double wg_min(Wg2& g, double x) {
  double v[1] = {x};
  wg_reduce_n<op_min, 1>(g, v);
  return v[0];
}
__device__ __forceinline__ bool wg_any(Wg2& g, bool p) { return wg_max(g, __ballot(p) != 0 ? 1.0 : 0.0) > 0.5; }

// ---- the active-set polish on two waves: lmpc_polish (lmpc_solve_kernel.hip) for real = io = double, KS = 0, rows on 128 lanes ----
template <int KQ, int KL, bool CHAIN>
__device__ __attribute__((noinline)) PolishResult<double, 0> lmpc_polish_w2(const PolishArgs<double, KL, 0> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  typedef double real;
  typedef double2 real2;
  typedef polish_limits<double> pol;
  real* const lds = reinterpret_cast<real*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = uni(a.N), NS = N - 1;
  constexpr bool LEAN = lmpc_lean(8, KQ);
  Lds<real> L{lds, N, LEAN ? LMPC_LEAN_STAGE_STRIDE : LMPC_STAGE_STRIDE, true, true};
  real* const T = L.tail();
  Wg2 g{T + TL_Y, wv, 0};
  ModelStream<real> MS{nullptr, nullptr, NS, lane, uni(a.have0), uni(a.have1)};
  if constexpr (LEAN) {
    MS.ws = uni_ptr(reinterpret_cast<const real*>(a.ws));
    MS.buf = T + LMPC_TAIL_DOUBLES;
  }
  const real inf = real(INFINITY);
  const bool has_sigma = uni(a.has_sigma) != 0;
  const real qsig = uni(a.qsig), inv_m = uni(a.inv_m);
  real sigma = uni(a.sigma);
  real hsig = 0.0, ce = 0.0, mu = 0.0, rdmax = 0.0, last_step = 0.0;
  int pol_rounds = uni(a.pol_rounds);
  Prof pf;
  (void)pf;
  const int KNB = NS * L.stride;
  const int JB = KNB + N * LMPC_KNOT_STRIDE + TL_W;
  const int CTB = KNB + N * LMPC_KNOT_STRIDE + TL_CT;
  int o_val[KL], o_hl[KL], s_gf[KL];
  real s_tu[KL], s_tl[KL], s_lu[KL], s_ll[KL], s_pu[KL], s_pl[KL];
#pragma unroll
  for (int q = 0; q < KL; ++q) {
    o_val[q] = a.o_val[q];
    o_hl[q] = a.o_hl[q];
    s_gf[q] = a.s_gf[q];
    s_tu[q] = a.s_tu[q];
    s_tl[q] = a.s_tl[q];
    s_lu[q] = a.s_lu[q];
    s_ll[q] = a.s_ll[q];
    s_pu[q] = s_pl[q] = 0.0;
  }
  auto ovq = [&](int q) { int v = o_val[q]; asm volatile("" : "+v"(v)); return v; };  // (opaque table reads: see lmpc_solve_problem_w2)
  auto ohq = [&](int q) { int v = o_hl[q]; asm volatile("" : "+v"(v)); return v; };
  auto gfq = [&](int q) { int v = s_gf[q]; asm volatile("" : "+v"(v)); return v; };
  auto flags = [&](int q) { return gfq(q) >> 20; };
  auto o_w = [&](int q) { return ovq(q) + ((flags(q) & F_EY) ? KN_EY - 1 : KN_R0); };
  auto o_csig = [&](int q) { return (flags(q) & F_EY) ? ovq(q) + (KN_CSIG - 1) : JB + KN_CSIG; };
  auto bounds = [&](int q) { return *reinterpret_cast<const real2*>(&lds[ohq(q)]); };
  double* const keep = uni_ptr(reinterpret_cast<double*>(a.keep));
  auto put_keep = [&]() {
    for (int e = tid; e < 10 * N - 4; e += W2_THREADS) {
      const int i = (e + 2) / 10, o = e + 2 - 10 * i;
      keep[e] = L.kn(i)[o];
    }
  };
  auto get_primal = [&]() {
    for (int e = tid; e < 10 * N - 4; e += W2_THREADS) {
      const int i = (e + 2) / 10, o = e + 2 - 10 * i;
      if (i >= 1 || o >= 8) L.kn(i)[o] = keep[e];
    }
  };
  const real gam = (uni(a.max_rounds) & POLISH_EXIT) ? real(POLISH_EXIT_GAMMA) : real(1);  // (lmpc_polish: in doubt a row is held at the exit)
  int held = 0;
#pragma unroll
  for (int q = 0; q < KL; ++q) held |= (s_lu[q] > gam * s_tu[q] ? 1 << (2 * q) : 0) | (s_ll[q] > gam * s_tl[q] ? 2 << (2 * q) : 0);
  const real sigma_keep = sigma;
  put_keep();
  bool accepted = false, noise = false;
  const int max_rounds = min((int)pol::rounds, uni(a.max_rounds) & ~POLISH_EXIT);
  for (int round = 0; round < max_rounds; ++round) {
    if (round > 0) {
      wg_sync();
      get_primal();
      sigma = sigma_keep;
      wg_sync();
    }
    real eysum = 0.0;
#pragma unroll
    for (int q = 0; q < KL; ++q) {
      const int f = flags(q);
      const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
      const real thu = hu ? real(pol::theta) : real(0), thd = hd ? real(pol::theta) : real(0);
      lds[o_w(q)] = thu + thd;
      lds[o_csig(q)] = (f & F_SIG) ? (thd - thu) : real(0);
      eysum += (f & F_SIG) ? (thu + thd) : real(0);
      s_pu[q] = (hu && s_lu[q] > s_tu[q]) ? s_lu[q] : real(0);
      s_pl[q] = (hd && s_ll[q] > s_tl[q]) ? s_ll[q] : real(0);
    }
    hsig = qsig + wg_sum(g, eysum);  // (its barrier also publishes the weights)
    ++pol_rounds;
    if constexpr (CHAIN) {
      if constexpr (LEAN)
        riccati_factor_lean<false, true>(L, MS, lane, (const double*)nullptr);
      else
        riccati_factor<false, true>(L, lane, (const double*)nullptr);
    }
    wg_sync();
    bool nan_step = false;
    for (int k = 0; k < pol::steps; ++k) {
      real sgsum = 0.0;
      {
        real val[KL], par[KL], ca[KL], cb[KL], ql[KL];
        real2 hl[KL];
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          const int gr = gfq(q);
          val[q] = lds[ovq(q)];
          hl[q] = bounds(q);
          par[q] = lds[ovq(q) + ((gr >> 16) & 3) - 1];
          ca[q] = lds[CTB + (gr & 0xff)];
          cb[q] = lds[CTB + ((gr >> 8) & 0xff)];
          ql[q] = lds[ovq(q) + (KN_QLIN - 3)];
        }
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          const int f = flags(q);
          const real sg = (f & F_SIG) ? sigma : 0.0;
          const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
          const real cu = hu ? rfma(real(pol::theta), val[q] - sg - hl[q].x, s_pu[q]) : real(0);
          const real cd = hd ? rfma(real(pol::theta), -val[q] - sg + hl[q].y, s_pl[q]) : real(0);
          const real gq = ca[q] * val[q] + cb[q] * par[q] + ((f & F_QLIN) ? ql[q] : real(0));
          lds[o_w(q)] = gq + cu - cd;
          if (k == 0) lds[(f & F_EY) ? JB + KN_EY : o_w(q) + 10] = 0.0;
          sgsum += (f & F_SIG) ? (cu + cd) : real(0);
        }
      }
      wg_sync();
      for (int i = tid; i < N; i += W2_THREADS) {
        real* kn = L.kn(i);
        kn[KN_R0 + 1] += kn[KN_EY];
        if (k == 0) kn[KN_R1 + 1] = (i >= 1) ? kn[KN_CSIG] : 0.0;
      }
      wg_sync();
      if constexpr (CHAIN) {
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
      }
      wg_sync();
      real dz0[KL], dz1[KL], val[KL];
      real2 hl[KL];
#pragma unroll
      for (int q = 0; q < KL; ++q) {
        dz0[q] = lds[ovq(q) + 10];
        dz1[q] = lds[ovq(q) + 20];
        val[q] = lds[ovq(q)];
        hl[q] = bounds(q);
      }
      real dsigma = 0.0;
      if (has_sigma) {
        real red[3] = {0.0, 0.0, sgsum};
        {
          real cs[KL];
#pragma unroll
          for (int q = 0; q < KL; ++q) cs[q] = lds[o_csig(q)];
#pragma unroll
          for (int q = 0; q < KL; ++q) {
            const bool sch = (flags(q) & F_SCH) != 0;
            red[0] += sch ? cs[q] * dz0[q] : real(0);
            red[1] += sch ? cs[q] * dz1[q] : real(0);
          }
        }
        wg_reduce_n<op_sum, 3>(g, red);
        if (k == 0) ce = red[1];
        const real qsg = qsig * sigma - red[2];
        dsigma = uni(-(qsg + red[0]) / (hsig + ce));
      }
      bool finite_step = dsigma == dsigma;
      real stepmax = 0.0;
#pragma unroll
      for (int q = 0; q < KL; ++q) {
        const int f = flags(q);
        const real dval = dz0[q] + dsigma * dz1[q];
        finite_step = finite_step && (fabs(dval) < inf);
        stepmax = fmax(stepmax, (f & F_MOVE) ? fabs(dval) * real(slot_inv_scale((gfq(q) >> 27) & 15)) : real(0));
        const real sg = (f & F_SIG) ? sigma : 0.0, dsg = (f & F_SIG) ? dsigma : 0.0;
        const bool hu = (held >> (2 * q)) & 1, hd = (held >> (2 * q + 1)) & 1;
        s_pu[q] = hu ? rfma(real(pol::theta), (val[q] - sg - hl[q].x) + (dval - dsg), s_pu[q]) : real(0);
        s_pl[q] = hd ? rfma(real(pol::theta), (-val[q] - sg + hl[q].y) + (-dval - dsg), s_pl[q]) : real(0);
        const bool mv = (f & F_MOVE) != 0;
        lds[mv ? ovq(q) : JB + q] += mv ? dval : real(0);
      }
      if (has_sigma) sigma = uni(sigma + dsigma);
      {
        real red[2] = {stepmax, finite_step ? 0.0 : 1.0};
        wg_reduce_n<op_max, 2>(g, red);  // (its barrier also publishes the update)
        last_step = red[0];
        nan_step = nan_step || red[1] > 0.5;
      }
      if (k >= 1 && last_step <= real(pol::step_ok)) break;
    }
    // ---- KKT test of the point reached; repair of the held set ----
    real val[KL];
    real2 hl[KL];
#pragma unroll
    for (int q = 0; q < KL; ++q) {
      val[q] = lds[ovq(q)];
      hl[q] = bounds(q);
    }
    bool bad = false, neg = false, weakneg = false, viol = false;
    real ymin = 0.0, comp = 0.0, worst = 0.0;
#pragma unroll
    for (int q = 0; q < KL; ++q) {
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
    // the four votes, the most negative multiplier (as a maximum of its negative) and the worst row in one exchange
    real vote[6] = {__ballot(bad) != 0 ? 1.0 : 0.0, __ballot(neg) != 0 ? 1.0 : 0.0, __ballot(weakneg) != 0 ? 1.0 : 0.0,
                    __ballot(viol) != 0 ? 1.0 : 0.0, -ymin, worst};
    wg_reduce_n<op_max, 6>(g, vote);
    const bool anybad = nan_step || !(last_step <= real(pol::step_tol)) || vote[0] > 0.5;
    const bool anyneg = vote[1] > 0.5, anyweak = vote[2] > 0.5, anyviol = vote[3] > 0.5;
    if (!anybad && !anyneg && !anyviol) {
      mu = wg_sum(g, comp) * inv_m;
      rdmax = vote[5];
      accepted = true;
      break;
    }
    noise = anybad && !anyneg && !anyviol;
    const real ycut = real(0.5) * (-vote[4]);
    const int before = held;
#pragma unroll
    for (int q = 0; q < KL; ++q) {
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
    if (!wg_any(g, held != before)) break;
  }
  if (!accepted) {
    wg_sync();
    get_primal();
    sigma = sigma_keep;
    wg_sync();
  }
  PolishResult<double, 0> res;
  res.accepted = accepted ? 1 : 0;
  res.noise = (!accepted && noise) ? 1 : 0;
  res.pol_rounds = pol_rounds;
  res.have0 = MS.have0;
  res.have1 = MS.have1;
  res.sigma = sigma;
  res.mu = mu;
  res.rdmax = rdmax;
  res.last_step = last_step;
  return res;
}

// ---- one problem on two waves: lmpc_solve_problem (lmpc_solve_kernel.hip) for real = io = double, KS = 0, cold start ----
template <int KQ, int KL, int SLOT0, bool CHAIN>
__device__ __forceinline__ void lmpc_solve_problem_w2(const lmpc_params& P, const int B, const int b, unsigned char* lds_raw,
                                                      const double* __restrict__ ws_lin, const double* __restrict__ x_ic,
                                                      const double* __restrict__ u_ic, const double* __restrict__ T_ref,
                                                      const double* __restrict__ bl, const double* __restrict__ br,
                                                      const double* __restrict__ vref, double* __restrict__ X_out, double* __restrict__ U_out,
                                                      double* __restrict__ dU_out, int* __restrict__ status_out, int* __restrict__ iters_out,
                                                      double* __restrict__ kkt_out) {
  typedef double real;
  typedef double2 real2;
  typedef ipm_limits<double> lim;
  typedef polish_limits<double> pol;
  real* const lds = reinterpret_cast<real*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int N = P.N, NS = N - 1;
  const real inf = real(INFINITY), marg = real(P.marg), qsig = real(P.qsig), tol = lim::tol(P.tol);
  constexpr bool LEAN = lmpc_lean(8, KQ);
  Lds<real> L{lds, N, LEAN ? LMPC_LEAN_STAGE_STRIDE : LMPC_STAGE_STRIDE, true, true};
  real* T = L.tail();
  real* ct = T + TL_CT;
  real* KN0 = L.kn(0);
  Wg2 g{T + TL_Y, wv, 0};
  ModelStream<real> MS{nullptr, nullptr, NS, lane, -1, -1};
  if constexpr (LEAN) {
    MS.ws = ws_lin + (size_t)b * NS * LMPC_LIN_RECORD;
    MS.buf = T + LMPC_TAIL_DOUBLES;
  }
  PT_DECL
  (void)pf;

  // ---------------- load ----------------
  {
    if constexpr (!LEAN) {
      const double* wsb = ws_lin + (size_t)b * NS * LMPC_LIN_RECORD;
      for (int e = tid; e < NS * LMPC_LIN_RECORD; e += W2_THREADS) {
        const int i = e / LMPC_LIN_RECORD, o = e - i * LMPC_LIN_RECORD;
        const int c = o / 6;
        L.st(i)[o < 48 ? ST_ROW(c) + (o - c * 6) : ST_G + (o - 48)] = wsb[e];
      }
    }
    for (int i = tid; i < NS; i += W2_THREADS) L.st(i)[LEAN ? LN_DT : ST_DT] = T_ref[(size_t)i * B + b];
    for (int i = tid; i < N; i += W2_THREADS) {
      real* kn = L.kn(i);
      kn[KN_QLIN] = real(i == N - 1 ? P.qv_term : P.qv_stage) * vref[(size_t)i * B + b];
      kn[8] = 0.0;
      kn[9] = 0.0;
      kn[KN_BHL] = bl[(size_t)i * B + b] - marg;
      kn[KN_BHL + 1] = br[(size_t)i * B + b] + marg;
    }
    if (tid < 6) {
      KN0[tid] = x_ic[(size_t)tid * B + b];
      ct[CT_QD + tid] = P.Qd[tid];
      ct[CT_QT + tid] = P.Qt[tid];
      ct[CT_HL + 2 * tid] = P.x_max[tid];
      ct[CT_HL + 2 * tid + 1] = P.x_min[tid];
    } else if (tid < 8) {
      KN0[tid] = u_ic[(size_t)(tid - 6) * B + b];
      ct[CT_HL + 2 * tid] = P.u_hi[tid - 6];
      ct[CT_HL + 2 * tid + 1] = P.u_lo[tid - 6];
    } else if (tid < 10) {
      ct[CT_HL + 2 * tid] = P.v_hi[tid - 8];
      ct[CT_HL + 2 * tid + 1] = P.v_lo[tid - 8];
    } else if (tid < 14) {
      ct[CT_QU + tid - 10] = P.Qu[tid - 10];
    } else if (tid < 18) {
      ct[CT_SV + tid - 14] = P.Sv[tid - 14];
    } else if (tid == 18) {
      ct[CT_ZERO] = 0.0;
    } else if (tid < 25) {
      ct[CT_E + tid - 19] = P.chs2[tid - 19];
    }
  }
  wg_sync();
  PT_MARK(0)

  // ---------------- slot ownership: slot j = SLOT0 + lane + 64 q ----------------
  const bool has_sigma = P.has_sigma != 0;
  const bool fuse = LMPC_W2_FUSE && has_sigma;  // the predictor's backward sweep inside the factorisation (riccati_factor<.., FUSE>)
  const int KNB = NS * L.stride;
  const int JB = KNB + N * LMPC_KNOT_STRIDE + TL_W;
  const int CTB = KNB + N * LMPC_KNOT_STRIDE + TL_CT;
  int o_val[KL], o_hl[KL], s_gf[KL];
  real s_tu[KL], s_tl[KL], s_lu[KL], s_ll[KL], s_pu[KL], s_pl[KL];
  real m_rows = 0.0;
#pragma unroll
  for (int q = 0; q < KL; ++q) {
    const int j = SLOT0 + lane + 64 * q;  // (wave 0: slots 0 .. 64 KL0 - 1; wave 1: the rest)
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
          const int k = uslot ? sl - SL_U : sl - SL_V;
          on = uslot ? i >= 1 : i <= N - 2;
          if (on) {
            ca = (uslot ? CT_QU : CT_SV) + 3 * k;
            cb = (uslot ? CT_QU : CT_SV) + 2 * k + 1 - k;
          }
          pd = k == 0 ? 1 : -1;
        }
      } else {
        hi = bl[(size_t)i * B + b] - marg;
        lo = br[(size_t)i * B + b] + marg;
        on = has_sigma || i >= 1;
      }
    }
    const bool au = on && (hi < inf), al = on && (lo > -inf);
    const bool eys = valid && sl == SL_EY;
    const int fl = (au ? F_UP : 0) | (al ? F_LO : 0) | ((eys && has_sigma) ? F_SIG : 0) | ((valid && sl == 3) ? F_QLIN : 0) |
                   ((valid && sl < SL_EY && (i >= 1 || sl >= SL_V)) ? F_MOVE : 0) | (eys ? F_EY : 0) | ((eys && i >= 1) ? F_SCH : 0);
    s_gf[q] = ca | (cb << 8) | ((pd + 1) << 16) | (fl << 20) | (sl << 27);
    m_rows += (au ? 1.0 : 0.0) + (al ? 1.0 : 0.0);
    s_tu[q] = s_tl[q] = 1.0;
    s_lu[q] = s_ll[q] = 0.0;
    s_pu[q] = s_pl[q] = 0.0;
  }
  // The slot's table integers as the row phases read them: through an empty asm, so that the LDS addresses derived from them are
  // recomputed at each use (a handful of integer instructions) instead of hoisted out of the iteration -- eight addresses per slot --,
  // spilled, and reloaded one s_waitcnt at a time (lmpc_opaque_slots of the one-wave kernels, profiles/r04_row_phases.md; without
  // it this kernel's gradient took 53 k cycles per iteration at N = 60 against 23 k in the one-wave kernel)
  auto ovq = [&](int q) { int v = o_val[q]; asm volatile("" : "+v"(v)); return v; };
  auto ohq = [&](int q) { int v = o_hl[q]; asm volatile("" : "+v"(v)); return v; };
  auto gfq = [&](int q) { int v = s_gf[q]; asm volatile("" : "+v"(v)); return v; };
  auto flags = [&](int q) { return gfq(q) >> 20; };
  auto o_w = [&](int q) { return ovq(q) + ((flags(q) & F_EY) ? KN_EY - 1 : KN_R0); };
  auto o_csig = [&](int q) { return (flags(q) & F_EY) ? ovq(q) + (KN_CSIG - 1) : JB + KN_CSIG; };
  auto bounds = [&](int q) { return *reinterpret_cast<const real2*>(&lds[ohq(q)]); };
  const real inv_m = real(1) / wg_sum(g, m_rows);

  // knot-0 feasibility
  bool feasible = true;
  {
    bool ok = true;
    if (tid < 6) {
      const real v = KN0[tid];
      ok = (v <= ct[CT_HL + 2 * tid]) && (v >= ct[CT_HL + 2 * tid + 1]);
    }
    if (tid == 6 && !has_sigma) {
      const real ey = KN0[1];
      ok = (ey <= bl[b] - marg) && (ey >= br[b] + marg);
    }
    feasible = wg_min(g, ok ? 1.0 : 0.0) > 0.5;
  }
  real sigma = 0.0;
  const real tau = 0.995, mu0 = 0.1, thr_frac = 0.5;
  int status = LMPC_SOLVE_MAX_ITER, it = 0;
  real mu = 0.0, rdmax = 0.0, rd_check = 0.0, last_step = 0.0, hsig = 0.0, ce = 0.0, mu_prev = inf;
  const int max_iter = feasible ? P.max_iter : 0;
  const bool polish_on = P.polish >= 0;
  bool polished = false, pol_early_done = false, reentry = false, pol_noise = false, distress = false, stall_moving = false;
  int pol_rounds = 0;
  auto put_primal = [&]() {
    const size_t xk = P.out_aos ? 1 : (size_t)N * B, xi = P.out_aos ? 6 : (size_t)B, xb = P.out_aos ? (size_t)6 * N : 1;
    const size_t uk = P.out_aos ? 1 : (size_t)NS * B, ui = P.out_aos ? 2 : (size_t)B, ub = P.out_aos ? (size_t)2 * NS : 1;
    for (int e = tid; e < 6 * N; e += W2_THREADS) {
      const int k = e / N, i = e - k * N;
      X_out[k * xk + i * xi + b * xb] = L.kn(i)[k];
    }
    for (int e = tid; e < 2 * NS; e += W2_THREADS) {
      const int k = e / NS, i = e - k * NS;
      U_out[k * uk + i * ui + b * ub] = L.kn(i + 1)[6 + k];
      dU_out[k * uk + i * ui + b * ub] = L.kn(i)[8 + k];
    }
  };
  auto polish_attempt = [&](int max_rounds) -> bool {
    PolishArgs<real, KL, 0> pa;
    pa.max_rounds = max_rounds;
    pa.keep = reinterpret_cast<double*>(P.save) + (size_t)b * (10 * N - 4);
    pa.ws = MS.ws;
    pa.N = N;
    pa.S = 0;
    pa.has_sigma = P.has_sigma;
    pa.have0 = MS.have0;
    pa.have1 = MS.have1;
    pa.pol_rounds = pol_rounds;
    pa.qsig = qsig;
    pa.inv_m = inv_m;
    pa.sigma = sigma;
#pragma unroll
    for (int q = 0; q < KL; ++q) {
      pa.o_val[q] = o_val[q];
      pa.o_hl[q] = o_hl[q];
      pa.s_gf[q] = s_gf[q];
      pa.s_tu[q] = s_tu[q];
      pa.s_tl[q] = s_tl[q];
      pa.s_lu[q] = s_lu[q];
      pa.s_ll[q] = s_ll[q];
    }
    const PolishResult<real, 0> pr = lmpc_polish_w2<KQ, KL, CHAIN>(pa);
    g.par = 0;  // (the polish leaves its exchange parity behind; both waves restart from the same one)
    wg_sync();
    pol_rounds = uni(pr.pol_rounds);
#pragma unroll
    for (int q = 0; q < KL; ++q) s_pu[q] = s_pl[q] = 0.0;
    hsig = 0.0;
    ce = 0.0;
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
    }
    return accepted;
  };

  // ---------------- start point: minimiser of the cost over the dynamics alone ----------------
#pragma unroll
  for (int q = 0; q < KL; ++q) {
    lds[o_w(q)] = 0.0;
    lds[o_csig(q)] = 0.0;
  }
  wg_sync();
  if constexpr (CHAIN) {
    if constexpr (LEAN) {
      riccati_factor_lean<false, false>(L, MS, lane, (const double*)nullptr);
      feedback_rollout_lean(L, MS, lane);
    } else {
      riccati_factor<false, false>(L, lane, (const double*)nullptr);
      feedback_rollout(L, lane);
    }
  }
  wg_sync();
  PT_MARK(1)

  it = -1;
  for (;;) {
    int hand_over = 0;
    for (; it <= max_iter; ++it) {
      const bool ipm = it >= 0;
      // ======== gradient (both waves; with the fused factorisation the predictor's runs AHEAD of the factorisation: lmpc_solve_problem) ========
      real sgsum0 = 0.0;
      auto gradient = [&](const int pass, const real smu, const real pm) -> real {
        const bool fused = fuse && ipm;
        real sgsum = 0.0;
        {
          real val[KL], par[KL], ca[KL], cb[KL], ql[KL];
          real2 hl[KL];
#pragma unroll
          for (int q = 0; q < KL; ++q) {
            const int gr = gfq(q), ov = ovq(q);
            val[q] = lds[ov];
            hl[q] = bounds(q);
            par[q] = lds[ov + ((gr >> 16) & 3) - 1];
            ca[q] = lds[CTB + (gr & 0xff)];
            cb[q] = lds[CTB + ((gr >> 8) & 0xff)];
            ql[q] = lds[ov + (KN_QLIN - 3)];
          }
#pragma unroll
          for (int q = 0; q < KL; ++q) {
            const int f = flags(q);
            const real sg = (f & F_SIG) ? sigma : 0.0;
            const real itu = frcp(s_tu[q]), itl = frcp(s_tl[q]);
            real cu = s_lu[q] * itu * (val[q] - sg + s_tu[q] - hl[q].x) + (smu - pm * s_pu[q]) * itu;
            real cd = s_ll[q] * itl * (-val[q] - sg + s_tl[q] + hl[q].y) + (smu - pm * s_pl[q]) * itl;
            cu = (ipm && (f & F_UP)) ? cu : 0.0;
            cd = (ipm && (f & F_LO)) ? cd : 0.0;
            const real gq = ca[q] * val[q] + cb[q] * par[q] + ((f & F_QLIN) ? ql[q] : real(0));
            lds[o_w(q)] = gq + cu - cd;
            if (pass == 0 && !fused) lds[(f & F_EY) ? JB + KN_EY : o_w(q) + 10] = 0.0;
            sgsum += (f & F_SIG) ? (cu + cd) : real(0);
          }
        }
        wg_sync();
        for (int i = tid; i < N; i += W2_THREADS) {
          real* kn = L.kn(i);
          kn[KN_R0 + 1] += kn[KN_EY];
          if (pass == 0 && !fused) kn[KN_R1 + 1] = (i >= 1) ? kn[KN_CSIG] : 0.0;
        }
        wg_sync();
        return sgsum;
      };
      // ======== rows ========
      if (ipm) {
        real musum = 0.0, rdl = 0.0, eysum = 0.0;
        {
          real val[KL];
          real2 hl[KL];
#pragma unroll
          for (int q = 0; q < KL; ++q) {
            val[q] = lds[ovq(q)];
            hl[q] = bounds(q);
          }
#pragma unroll
          for (int q = 0; q < KL; ++q) {
            const int f = flags(q);
            const real sg = (f & F_SIG) ? sigma : 0.0;
            const real thu = s_lu[q] * frcp(s_tu[q]), thd = s_ll[q] * frcp(s_tl[q]);
            musum += s_lu[q] * s_tu[q] + s_ll[q] * s_tl[q];
            rdl = fmax(rdl, (f & F_UP) ? fabs(val[q] - sg + s_tu[q] - hl[q].x) : real(0));
            rdl = fmax(rdl, (f & F_LO) ? fabs(-val[q] - sg + s_tl[q] + hl[q].y) : real(0));
            lds[o_w(q) + (fuse ? ((f & F_EY) ? KN_TEY - KN_EY : KN_R1 - KN_R0) : 0)] = thu + thd;  // (fused factorisation: the weights in the rhs1 cells)
            lds[o_csig(q)] = (f & F_SIG) ? (thd - thu) : 0.0;
            eysum += (f & F_SIG) ? (thu + thd) : real(0);
          }
        }
        {
          real red[2] = {musum, eysum};
          wg_reduce_n<op_sum, 2>(g, red);
          musum = red[0];
          hsig = qsig + red[1];
        }
        rdmax = wg_max(g, rdl);  // (the two barriers above also publish the barrier weights for the factorisation)
        mu = musum * inv_m;
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
        if (!again_here && it % 5 == 0) {
          if (it >= 10 && rdmax > lim::rd_infeasible && rdmax > real(0.9) * rd_check) {
            status = LMPC_SOLVE_INFEASIBLE;
            break;
          }
          rd_check = rdmax;
        }
        if (it == max_iter) break;
        if (polish_on && !pol_early_done && mu <= real(pol::mu_early) && rdmax <= real(pol::rd_early)) {
          pol_early_done = true;
          hand_over = 1;
          break;
        }
        PT_MARK(2)
        if (fuse) {
          sgsum0 = gradient(0, real(0), real(0));
          PT_MARK(12)
        }
        if constexpr (CHAIN) {
          if constexpr (LEAN) {
            if (mu <= real(JOSEPH_MU)) {
              if (fuse)
                riccati_factor_lean<false, true, true>(L, MS, lane, (const double*)nullptr, KN_R1, KN_TEY);
              else
                riccati_factor_lean<false, true>(L, MS, lane, (const double*)nullptr);
            } else {
              if (fuse)
                riccati_factor_lean<false, false, true>(L, MS, lane, (const double*)nullptr, KN_R1, KN_TEY);
              else
                riccati_factor_lean<false, false>(L, MS, lane, (const double*)nullptr);
            }
          } else if (mu <= real(JOSEPH_MU)) {
            if (fuse)
              riccati_factor<false, true, true>(L, lane, (const double*)nullptr, KN_R1, KN_TEY);
            else
              riccati_factor<false, true>(L, lane, (const double*)nullptr);
          } else {
            if (fuse)
              riccati_factor<false, false, true>(L, lane, (const double*)nullptr, KN_R1, KN_TEY);
            else
              riccati_factor<false, false>(L, lane, (const double*)nullptr);
          }
        }
        wg_sync();
        PT_MARK(3)
      }

      real sigc = 0.0, alpha = 1.0, dsigma = 0.0;
      bool numerics_failed = false, stalled = false;
      real d_val[KL];
      const int npass = ipm ? 2 : 1;
      for (int pass = 0; pass < npass; ++pass) {
        const real smu = (pass == 1) ? sigc * mu : 0.0, pm = (pass == 1) ? 1.0 : 0.0;
        real sgsum = sgsum0;
        if (!(fuse && ipm && pass == 0)) sgsum = gradient(pass, smu, pm);
        PT_MARK(12)
        // ======== Newton step (wave 0) ========
        if constexpr (CHAIN) {
          if constexpr (LEAN) {
            if (pass == 0 && ipm && has_sigma) {
              if (fuse)
                riccati_solve_lean_dpp<2, true>(L, MS, lane, pf);
              else
                riccati_solve_lean_dpp<2>(L, MS, lane, pf);
            } else
              riccati_solve_lean_dpp<1>(L, MS, lane, pf);
          } else {
            if (pass == 0 && ipm && has_sigma) {
              if (fuse)
                riccati_solve<2, true>(L, lane, pf);
              else
                riccati_solve<2>(L, lane, pf);
            } else
              riccati_solve<1>(L, lane, pf);
          }
        }
        wg_sync();
        PT_MARK(4)
        // ======== steps ========
        real dz0[KL], dz1[KL], val[KL];
        real2 hl[KL];
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          dz0[q] = lds[ovq(q) + 10];
          dz1[q] = lds[ovq(q) + 20];
          val[q] = lds[ovq(q)];
          hl[q] = bounds(q);
        }
        if (!ipm) {
#pragma unroll
          for (int q = 0; q < KL; ++q) d_val[q] = dz0[q];
          break;
        }
        if (has_sigma) {
          real red[3] = {0.0, 0.0, sgsum};
          real cs[KL];
#pragma unroll
          for (int q = 0; q < KL; ++q) cs[q] = lds[o_csig(q)];
#pragma unroll
          for (int q = 0; q < KL; ++q) {
            const bool sch = (flags(q) & F_SCH) != 0;
            red[0] += sch ? cs[q] * dz0[q] : real(0);
            red[1] += sch ? cs[q] * dz1[q] : real(0);
          }
          wg_reduce_n<op_sum, 3>(g, red);
          if (pass == 0) ce = red[1];
          const real qsg = qsig * sigma - red[2];
          dsigma = uni(-(qsg + red[0]) / (hsig + ce));
        }
        real dtu[KL], dlu[KL], dtl[KL], dll[KL];
        real rmax = 1.0;
        bool finite_step = true;
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          const int f = flags(q);
          const real dval = dz0[q] + dsigma * dz1[q];
          d_val[q] = dval;
          finite_step = finite_step && (fabs(dval) < inf);
          const real sg = (f & F_SIG) ? sigma : 0.0, dsg = (f & F_SIG) ? dsigma : 0.0;
          const real itu = frcp(s_tu[q]), itl = frcp(s_tl[q]);
          const real a_ = (f & F_UP) ? -(val[q] - sg + s_tu[q] - hl[q].x) - (dval - dsg) : 0.0;
          const real b_ = (f & F_UP) ? -s_lu[q] + (smu - pm * s_pu[q]) * itu - s_lu[q] * itu * a_ : 0.0;
          const real c_ = (f & F_LO) ? -(-val[q] - sg + s_tl[q] + hl[q].y) - (-dval - dsg) : 0.0;
          const real d_ = (f & F_LO) ? -s_ll[q] + (smu - pm * s_pl[q]) * itl - s_ll[q] * itl * c_ : 0.0;
          dtu[q] = a_;
          dlu[q] = b_;
          dtl[q] = c_;
          dll[q] = d_;
          rmax = fmax(rmax, fmax(-a_ * itu, -b_ * frcp(fmax(s_lu[q], lim::tiny))));
          rmax = fmax(rmax, fmax(-c_ * itl, -d_ * frcp(fmax(s_ll[q], lim::tiny))));
        }
        rmax = wg_max(g, finite_step ? rmax : inf);
        if (!(rmax < inf) || !(dsigma == dsigma)) {
          numerics_failed = true;
          break;
        }
        const real amax = real(1) / rmax;
        if (pass == 1) {
          alpha = fmin(real(1), tau * amax);
          if (KQ <= 7 && distress) {  // the wide-neighbourhood rule of the N <= 40 kernels (lmpc_solve_problem; mirrored by the twin)
            for (int trial = 0; trial < NBHD_TRIALS; ++trial) {
              real sl = 0.0, pmin = inf;
#pragma unroll
              for (int q = 0; q < KL; ++q) {
                const int f = flags(q);
                const real pu = (s_tu[q] + alpha * dtu[q]) * (s_lu[q] + alpha * dlu[q]);
                const real pl = (s_tl[q] + alpha * dtl[q]) * (s_ll[q] + alpha * dll[q]);
                sl += pu + pl;
                pmin = fmin(pmin, fmin((f & F_UP) ? pu : inf, (f & F_LO) ? pl : inf));
              }
              pmin = wg_min(g, pmin);
              if (pmin >= real(NBHD_GAMMA) * wg_sum(g, sl) * inv_m) break;
              alpha = alpha * real(0.6);
            }
          }
        }
        real sacc = 0.0;
#pragma unroll
        for (int q = 0; q < KL; ++q) {
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
        sacc = wg_sum(g, sacc);
        if (pass == 1) {
          if (rdmax <= lim::rd_ok && mu <= real(STALL_MU) && sacc * inv_m >= mu) stalled = true;
        } else {
          const real ratio = (sacc * inv_m) / mu;
          sigc = ratio * ratio * ratio;
        }
      }

      if (numerics_failed) {
        status = (mu <= real(10) * tol && rdmax <= lim::rd_ok) ? LMPC_SOLVE_OPTIMAL : LMPC_SOLVE_MAX_ITER;
        hand_over = (status == LMPC_SOLVE_OPTIMAL && polish_on) ? 2 : 0;
        break;
      }
      if (stalled) {
        // (round 6: a stall is not convergence when the Newton step it declines would still move the iterate -- the point is kept and handed to the
        //  polish as before, but unless the polish verifies it the status is MAX_ITER: oracle/c/lmpc_oracle.c, STALL_STEP, has the problem)
        real cand = 0.0;
#pragma unroll
        for (int q = 0; q < KL; ++q) cand = fmax(cand, (flags(q) & F_MOVE) ? fabs(alpha * d_val[q]) * real(slot_inv_scale((gfq(q) >> 27) & 15)) : real(0));
        stall_moving = wg_max(g, cand) > real(STALL_STEP);
        status = (stall_moving && !polish_on) ? LMPC_SOLVE_MAX_ITER : LMPC_SOLVE_OPTIMAL;
        hand_over = polish_on ? 2 : 0;
        break;
      }
      // ======== primal update ========
      PT_MARK(6)
      real stepmax = 0.0;
#pragma unroll
      for (int q = 0; q < KL; ++q) {
        const bool mv = (flags(q) & F_MOVE) != 0;
        const real dz = mv ? alpha * d_val[q] : 0.0;
        lds[mv ? ovq(q) : JB + q] += dz;
        stepmax = fmax(stepmax, fabs(dz));
      }
      if (ipm) {
        last_step = wg_max(g, stepmax);  // (its barrier also publishes the update)
        if (has_sigma) sigma = sigma + alpha * dsigma;
      } else {
        wg_sync();
        real val[KL];
        real2 hl[KL];
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          val[q] = lds[ovq(q)];
          hl[q] = bounds(q);
        }
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          const int fl = flags(q);
          real range = ((fl & (F_UP | F_LO)) == (F_UP | F_LO)) ? (hl[q].x - hl[q].y) : 1.0;
          if (!(range > real(1e-3))) range = real(1e-3);
          const real thr = thr_frac * range;
          if (fl & F_UP) {
            s_tu[q] = fmax(hl[q].x - val[q], thr);
            s_lu[q] = mu0 / s_tu[q];
          }
          if (fl & F_LO) {
            s_tl[q] = fmax(val[q] - hl[q].y, thr);
            s_ll[q] = mu0 / s_tl[q];
          }
        }
        sigma = 0.0;
      }
    }
    if (hand_over == 0) break;
    wg_sync();
    if (polish_attempt(pol::rounds | (hand_over >= 2 ? POLISH_EXIT : 0))) {
      polished = true;
      status = LMPC_SOLVE_OPTIMAL;
      break;
    }
    if (hand_over == 2) {
      status = LMPC_SOLVE_OPTIMAL;
      if (pol_noise || stall_moving) status = LMPC_SOLVE_MAX_ITER;  // (lmpc_solve_problem: the noise rule, and round 6's stall rule)
      break;
    }
    reentry = true;
  }
  (void)polished;
  PT_MARK(7)
  if (it < 0) it = 0;
  it += pol_rounds;
  if (!feasible) status = LMPC_SOLVE_INFEASIBLE;
  wg_sync();
  put_primal();
  if (CHAIN && lane == 0) {
    status_out[b] = status;
    iters_out[b] = it;
#ifdef LMPC_PHASE_TIMING
    if (kkt_out) {  // (scratch/phase_timing.py: the chain wave's clock)
      for (int k = 0; k < 16; ++k) kkt_out[k * (size_t)B + b] = (double)pf.acc[k];
      kkt_out[16 * (size_t)B + b] = (double)pf.w0;
      kkt_out[17 * (size_t)B + b] = (double)wall_clock64();
      unsigned hwid, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      kkt_out[18 * (size_t)B + b] = (double)hwid;
      kkt_out[19 * (size_t)B + b] = (double)(xcc & 0xf);
    }
#else
    if (kkt_out) {
      kkt_out[0 * (size_t)B + b] = last_step;
      kkt_out[1 * (size_t)B + b] = rdmax;
      kkt_out[2 * (size_t)B + b] = mu;
      kkt_out[3 * (size_t)B + b] = sigma;
    }
#endif
  }
}

// (two waves per SIMD: eight waves = four problems per CU at N <= 64, six = three beyond -- what LDS allows either way)
template <int KQ>
__global__ __launch_bounds__(W2_THREADS, 2) void lmpc_solve_kernel_w2(lmpc_params P, int B, const double* __restrict__ ws_lin,
                                                                       const double* __restrict__ x_ic, const double* __restrict__ u_ic,
                                                                       const double* __restrict__ T_ref, const double* __restrict__ bl,
                                                                       const double* __restrict__ br, const double* __restrict__ vref,
                                                                       double* __restrict__ X_out, double* __restrict__ U_out,
                                                                       double* __restrict__ dU_out, int* __restrict__ status_out,
                                                                       int* __restrict__ iters_out, double* __restrict__ kkt_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  int b = (int)(blockIdx.x & 7) * ((B + 7) >> 3) + (int)(blockIdx.x >> 3);  // (XCD-aware, as lmpc_solve_kernel)
  if (P.launch_order) {
    const int n = P.order_count ? *P.order_count : B;
    b = (int)blockIdx.x < n ? P.launch_order[blockIdx.x] : B;
  }
  if (b >= B) return;
  constexpr int KL0 = lmpc_w2_slots0(KQ), KL1 = lmpc_w2_slots1(KQ);
  // Which of the two waves runs the chain phases: wave 0.  (LMPC_W2_ROLE = 1, an experiment kept for the record: if the hardware
  // dealt a workgroup's waves to neighbouring SIMDs and filled a CU pair by pair, the chain waves of two problems would share SIMD 0
  // and two more SIMD 2; giving the role to the wave whose SIMD parity equals its wave slot's parity would spread them.  Measured: no
  // difference -- the chain phases' cycle counts are the one-wave kernel's either way.)
  int chain_wave = 0;
  if (LMPC_W2_ROLE) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int claim = ((((hw >> 4) & 3) ^ (hw & 15)) & 1) == 0;
    int* const vote = reinterpret_cast<int*>(lds_raw);  // (the first cells of the records: not loaded yet)
    if ((threadIdx.x & 63) == 0) vote[threadIdx.x >> 6] = claim;
    __syncthreads();
    const int v0 = vote[0], v1 = vote[1];
    __syncthreads();
    chain_wave = __builtin_amdgcn_readfirstlane((v0 != v1) ? (v0 ? 0 : 1) : 0);
  }
  if ((int)(threadIdx.x >> 6) == chain_wave)
    lmpc_solve_problem_w2<KQ, KL0, 0, true>(P, B, b, lds_raw, ws_lin, x_ic, u_ic, T_ref, bl, br, vref, X_out, U_out, dU_out, status_out, iters_out,
                                            kkt_out);
  else
    lmpc_solve_problem_w2<KQ, KL1, 64 * KL0, false>(P, B, b, lds_raw, ws_lin, x_ic, u_ic, T_ref, bl, br, vref, X_out, U_out, dU_out, status_out,
                                                    iters_out, kkt_out);
}

#endif
