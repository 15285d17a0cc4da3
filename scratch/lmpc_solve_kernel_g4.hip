// Grouped variant of lmpc_solve_kernel for the tracking problem in fp64 (KS = 0): FOUR problems per workgroup, one wave
// each for everything that is organised per problem (rows, factorisation: lane (r, c) of the 8x8 cost-to-go), and ONE wave
// for the Riccati vector sweeps of all four -- lane (p, s, r) = (problem, right-hand side, component): 4 x 2 x 8 = 64
// lanes in the predictor solve, 4 x 8 in the corrector.  In the one-wave-per-problem kernel the sweeps are 47 % of the
// cycles with 16 (8) of 64 lanes at work; here their DS / VALU instructions issue once per four problems.  The price:
// the four problems iterate in lock step (two s_barrier pairs per iteration around the solves, one for the exit vote) and a
// group holds its LDS until its slowest problem is done.  Per problem the arithmetic is that of lmpc_solve_kernel,
// operation for operation.  The wave that runs the merged sweep rotates (solve counter mod 4) so the four SIMDs of a CU
// share that work.
template <int KQ>
__global__ __launch_bounds__(256, 2) void lmpc_solve_kernel_g4(
    lmpc_params P, int B, const double* __restrict__ ws_lin, const double* __restrict__ x_ic,
    const double* __restrict__ u_ic, const double* __restrict__ T_ref, const double* __restrict__ bl,
    const double* __restrict__ br, const double* __restrict__ vref, double* __restrict__ X_out,
    double* __restrict__ U_out, double* __restrict__ dU_out, int* __restrict__ status_out,
    int* __restrict__ iters_out, double* __restrict__ kkt_out) {
  typedef double real;
  typedef double io;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  // XCD-aware assignment of GROUPS (see lmpc_solve_kernel): each XCD owns a contiguous eighth of the batch
  const int G = (B + 3) >> 2;
  const int g = (int)(blockIdx.x & 7) * ((G + 7) >> 3) + (int)(blockIdx.x >> 3);
  if (g >= G) return;
  const bool live = 4 * g + wv < B;
  const int b = live ? 4 * g + wv : B - 1;  // a surplus wave of the last group repeats the last problem and writes nothing
  const int N = P.N, NS = N - 1;
  typedef typename vec2<real>::type real2;
  typedef double treal;
  typedef ipm_limits<real> lim;
  const real inf = real(INFINITY), marg = real(P.marg), qsig = real(P.qsig), tol = lim::tol(P.tol);
  const io s_shift = io(0);
  const int PBD = NS * LMPC_STAGE_STRIDE + N * LMPC_KNOT_STRIDE + LMPC_TAIL_DOUBLES;  // one problem's records (doubles)
  real* const lds_group = reinterpret_cast<real*>(lds_raw);
  real* const lds = lds_group + wv * PBD;
  int* const vote = reinterpret_cast<int*>(lds_group + 4 * PBD);  // [4]: this wave's problem is finished
  Lds<real> L{lds, N};
  real* T = L.tail();
  real* ct = T + TL_CT;
  real* KN0 = L.kn(0);
  treal* const TT = nullptr;  // (no terminal region: tracking problem)
  PT_DECL

  // ---------------- load: linearisation records, per-knot data, constant tables ----------------
  {
    const io* wsb = ws_lin + (size_t)b * NS * LMPC_LIN_RECORD;
    for (int e = lane; e < NS * LMPC_LIN_RECORD; e += 64) {
      const int i = e / LMPC_LIN_RECORD, o = e - i * LMPC_LIN_RECORD;
      const int c = o / 6;
      L.st(i)[o < 48 ? ST_ROW(c) + (o - c * 6) : ST_G + (o - 48)] = real(wsb[e]);
    }
    for (int i = lane; i < NS; i += 64) L.st(i)[ST_DT] = real(T_ref[(size_t)i * B + b]);
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
  const int KNB = NS * LMPC_STAGE_STRIDE;            // knot 0, offset from the LDS base (doubles)
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
    s_gf[q] = ca | (cb << 8) | ((pd + 1) << 16) | (fl << 20);
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

  // ---------------- start point: minimiser of the cost over the dynamics alone ----------------
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    lds[o_w(q)] = 0.0;
    lds[o_csig(q)] = 0.0;
  }
  wave_sync();
  riccati_factor<false, false>(L, lane, TT);
  feedback_rollout(L, lane);
  PT_MARK(1)

  const real tau = 0.995, mu0 = 0.1, thr_frac = 0.5;
  int status = LMPC_SOLVE_MAX_ITER, it = 0, it_done = 0, solves = 0;
  bool done = false;  // this wave's problem has left the iteration (status and it_done are final)
  real mu = 0.0, rdmax = 0.0, rd_check = 0.0, last_step = 0.0, hsig = 0.0, ce = 0.0, mu_prev = inf;
  bool distress = false;
  const int max_iter = feasible ? P.max_iter : 0;

  // it == -1 is the start-point Newton step (all row weights zero, full step); it >= 0 the
  // interior-point iterations.
  for (it = -1;; ++it) {
    const bool ipm = it >= 0;  // (the four problems of the group are at the same `it`)
    // ======== rows: complementarity, residual, barrier weights ========
    if (ipm && !done) {
      real musum = 0.0, rdl = 0.0, eysum = 0.0;
      {
        real val[KQ];
        real2 hl[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          val[q] = lds[o_val[q]];
          hl[q] = bounds(q);
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
          const int f = flags(q);
          const real sg = (f & F_SIG) ? sigma : 0.0;
          const real thu = s_lu[q] * frcp(s_tu[q]), thd = s_ll[q] * frcp(s_tl[q]);
          musum += s_lu[q] * s_tu[q] + s_ll[q] * s_tl[q];
          rdl = fmax(rdl, (f & F_UP) ? fabs(val[q] - sg + s_tu[q] - hl[q].x) : real(0));
          rdl = fmax(rdl, (f & F_LO) ? fabs(-val[q] - sg + s_tl[q] + hl[q].y) : real(0));
          lds[o_w(q)] = thu + thd;
          lds[o_csig(q)] = (f & F_SIG) ? (thd - thu) : 0.0;
          eysum += (f & F_SIG) ? (thu + thd) : real(0);
        }
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
      if (it >= 1 && mu >= mu_prev && rdmax <= lim::rd_distress) distress = true;
      mu_prev = mu;
      if (!(mu == mu) || !(rdmax == rdmax)) {
        status = LMPC_SOLVE_INFEASIBLE;
        done = true;
      } else if (mu <= tol && rdmax <= lim::rd_ok) {
        status = LMPC_SOLVE_OPTIMAL;
        done = true;
      } else {
        // primal infeasibility: see lmpc_solve_kernel
        if (it % 5 == 0) {
          if (it >= 10 && rdmax > lim::rd_infeasible && rdmax > real(0.9) * rd_check) {
            status = LMPC_SOLVE_INFEASIBLE;
            done = true;
          }
          rd_check = rdmax;
        }
        if (it >= max_iter) done = true;
        if (!done) {
          wave_sync();
          PT_MARK(2)
          if (mu <= real(JOSEPH_MU))
            riccati_factor<false, true>(L, lane, TT);
          else
            riccati_factor<false, false>(L, lane, TT);
          PT_MARK(3)
        }
      }
      if (done) it_done = it;
    }

    real sigc = 0.0, alpha = 1.0, dsigma = 0.0;
    bool stalled = false;
    real d_val[KQ];
    const int npass = ipm ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
      const real smu = (pass == 1) ? sigc * mu : 0.0, pm = (pass == 1) ? 1.0 : 0.0;
      // ======== gradient: cost gradient + row coefficients, written by the component owner ========
      real sgsum = 0.0;  // sum of boundary-row coefficients entering the sigma gradient
      if (!done) {
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
          const real itu = frcp(s_tu[q]), itl = frcp(s_tl[q]);
          real cu = s_lu[q] * itu * (val[q] - sg + s_tu[q] - hl[q].x) + (smu - pm * s_pu[q]) * itu;
          real cd = s_ll[q] * itl * (-val[q] - sg + s_tl[q] + hl[q].y) + (smu - pm * s_pl[q]) * itl;
          cu = (ipm && (f & F_UP)) ? cu : 0.0;
          cd = (ipm && (f & F_LO)) ? cd : 0.0;
          const real g = ca[q] * val[q] + cb[q] * par[q] + ((f & F_QLIN) ? ql[q] : real(0));  // (zero coefficients on a boundary slot)
          lds[o_w(q)] = g + cu - cd;
          if (pass == 0) lds[(f & F_EY) ? JB + KN_EY : o_w(q) + 10] = 0.0;
          sgsum += (f & F_SIG) ? (cu + cd) : real(0);
        }
      }
      if (!done) {
        wave_sync();
        PT_MARK(12)
        for (int i = lane; i < N; i += 64) {
          real* kn = L.kn(i);
          kn[KN_R0 + 1] += kn[KN_EY];
          if (pass == 0) kn[KN_R1 + 1] = (i >= 1) ? kn[KN_CSIG] : 0.0;
        }
      }
      // ======== Newton step: predictor together with the Schur vector, then the corrector ========
      // The right-hand sides of the four problems are in their knots; one wave runs the sweeps of all four (a finished
      // problem's lanes run along on its records: only the rhs regions are written, the solution is not touched).
      PT_MARK(4)
      __syncthreads();
      if (wv == (solves & 3)) {
        if (pass == 0 && ipm && has_sigma) {
          const Lds<real> Lp{lds_group + (lane >> 4) * PBD, N};
          riccati_solve_lds<2, 4>(Lp, lane, pf);
        } else {
          const Lds<real> Lp{lds_group + ((lane >> 3) & 3) * PBD, N};
          riccati_solve_lds<1, 4>(Lp, lane, pf);
        }
      }
      ++solves;
      __syncthreads();
      PT_MARK(5)
      if (done) continue;
      // ======== step of every constrained value; boundary slack by Schur complement ========
      real dz0[KQ], dz1[KQ], val[KQ];
      real2 hl[KQ];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        dz0[q] = lds[o_val[q] + 10];
        dz1[q] = lds[o_val[q] + 20];
        val[q] = lds[o_val[q]];
        hl[q] = bounds(q);
      }
      if (!ipm) {  // (start point: one pass, full step)
#pragma unroll
        for (int q = 0; q < KQ; ++q) d_val[q] = dz0[q];
        continue;
      }
      if (has_sigma) {
        real red[3] = {0.0, 0.0, sgsum};  // c'dz (this rhs), c'e (Schur vector), sum of boundary coefficients
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
        if (pass == 0) ce = red[1];
        const real qsg = qsig * sigma - red[2];
        dsigma = uni(-(qsg + red[0]) / (hsig + ce));
      }
      // ======== row steps; largest feasible step as 1 / max(1, max -dt/t, max -dlam/lam) ========
      real dtu[KQ], dlu[KQ], dtl[KQ], dll[KQ];
      real rmax = 1.0;
      bool finite_step = true;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        const int f = flags(q);
        const real dval = dz0[q] + dsigma * dz1[q];
        d_val[q] = dval;
        finite_step = finite_step && (fabs(dval) < inf);
        const real sg = (f & F_SIG) ? sigma : 0.0, dsg = (f & F_SIG) ? dsigma : 0.0;
        const real itu = frcp(s_tu[q]), itl = frcp(s_tl[q]);
        const real a = (f & F_UP) ? -(val[q] - sg + s_tu[q] - hl[q].x) - (dval - dsg) : 0.0;
        const real bq = (f & F_UP) ? -s_lu[q] + (smu - pm * s_pu[q]) * itu - s_lu[q] * itu * a : 0.0;
        const real c = (f & F_LO) ? -(-val[q] - sg + s_tl[q] + hl[q].y) - (-dval - dsg) : 0.0;
        const real d = (f & F_LO) ? -s_ll[q] + (smu - pm * s_pl[q]) * itl - s_ll[q] * itl * c : 0.0;
        dtu[q] = a;
        dlu[q] = bq;
        dtl[q] = c;
        dll[q] = d;
        rmax = fmax(rmax, fmax(-a * itu, -bq * frcp(fmax(s_lu[q], lim::tiny))));
        rmax = fmax(rmax, fmax(-c * itl, -d * frcp(fmax(s_ll[q], lim::tiny))));
      }
      rmax = wave_max(finite_step ? rmax : inf);
      if (!(rmax < inf) || !(dsigma == dsigma)) {
        // a Newton step that is not a number (the Schur complement of sigma or the 2x2 H cancelled completely -- in
        // practice single precision on its last iteration): keep the iterate, report it by what it has reached
        status = (mu <= real(10) * tol && rdmax <= lim::rd_ok) ? LMPC_SOLVE_OPTIMAL : LMPC_SOLVE_MAX_ITER;
        done = true;
        it_done = it;
        continue;
      }
      const real amax = uni(real(1) / rmax);
      if (pass == 1) {
        alpha = uni(fmin(real(1), tau * amax));
        if (sizeof(real) == 8 && distress) {  // (fp64 arithmetic only)
          // A problem whose complementarity has gone UP once gets the wide-neighbourhood rule from then on: the step is
          // cut back until no complementarity product falls below NBHD_GAMMA times their mean.  Mehrotra's iteration can
          // otherwise leave the neighbourhood of the central path and cycle -- seen on a learning problem whose safe set
          // offers two nearly exchangeable points: products at 0.01 and 300 times mu, mu bouncing between 6e-6 and 2e-5
          // up to the iteration cap while the dense solver finds the optimum (19 iterations with the rule).  Problems
          // whose mu falls monotonically (all but a few per thousand) never enter this branch.
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

    if (!done && stalled) {
      status = LMPC_SOLVE_OPTIMAL;
      done = true;
      it_done = it;
    }
    // ======== exit vote: the group leaves when its four problems have ========
    if (lane == 0) vote[wv] = done ? 1 : 0;
    __syncthreads();
    if ((vote[0] & vote[1] & vote[2] & vote[3]) != 0) break;
    if (done) continue;
    // ======== primal update by the component owners ========
    PT_MARK(6)
    real stepmax = 0.0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const bool mv = (flags(q) & F_MOVE) != 0;
      const real dz = mv ? alpha * d_val[q] : 0.0;
      lds[mv ? o_val[q] : JB + q] += dz;
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
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        val[q] = lds[o_val[q]];
        hl[q] = bounds(q);
      }
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        real range = ((flags(q) & (F_UP | F_LO)) == (F_UP | F_LO)) ? (hl[q].x - hl[q].y) : 1.0;
        if (!(range > real(1e-3))) range = real(1e-3);
        const real thr = thr_frac * range;
        if (flags(q) & F_UP) {
          s_tu[q] = fmax(hl[q].x - val[q], thr);
          s_lu[q] = mu0 / s_tu[q];
        }
        if (flags(q) & F_LO) {
          s_tl[q] = fmax(val[q] - hl[q].y, thr);
          s_ll[q] = mu0 / s_tl[q];
        }
      }
      sigma = 0.0;
    }
  }
  PT_MARK(7)
  it = it_done < 0 ? 0 : it_done;
  if (!feasible) status = LMPC_SOLVE_INFEASIBLE;
  if (!live) return;

  // ---------------- write back: X [6][N][B], U, dU [2][N-1][B] ----------------
  wave_sync();
  for (int e = lane; e < 6 * N; e += 64) {
    const int k = e / N, i = e - k * N;
    X_out[(size_t)(k * N + i) * B + b] = io(L.kn(i)[k]) + (k == 0 ? s_shift : io(0));
  }
  for (int e = lane; e < 2 * NS; e += 64) {
    const int k = e / NS, i = e - k * NS;
    U_out[(size_t)(k * NS + i) * B + b] = io(L.kn(i + 1)[6 + k]);
    dU_out[(size_t)(k * NS + i) * B + b] = io(L.kn(i)[8 + k]);
  }
  if (lane == 0) {
    status_out[b] = status;
    iters_out[b] = it;
#ifdef LMPC_PHASE_TIMING
    if (kkt_out) {
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

template __global__ void lmpc_solve_kernel_g4<2>(lmpc_params, int, const double*, const double*, const double*, const double*,
                                                 const double*, const double*, const double*, double*, double*, double*, int*,
                                                 int*, double*);
template __global__ void lmpc_solve_kernel_g4<4>(lmpc_params, int, const double*, const double*, const double*, const double*,
                                                 const double*, const double*, const double*, double*, double*, double*, int*,
                                                 int*, double*);
