/*
 * lmpc_oracle.c -- CPU restatement (plain C, fp64) of the batched LMPC solve path.
 *
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (racing-lmpc-ros2_amd/) never
 * links or calls it.
 *
 * PARITY UNPINNED: the reference (MPC-Berkeley/Racing-LMPC-ROS2) evaluates this path
 * inside CasADi (>= 3.6.3, unpinned) + OSQP (conic plugin) + CGAL, none of which is
 * available, and its own tests assert no numbers.  This file follows
 * the reference's problem definition line by line and solves the same QP to its
 * (unique) optimum; tests pin it against the dense numpy restatement in oracle/qp.py
 * and its solver-independent KKT certificate.
 *
 * What is restated, and from where:
 *   dynamics f, partials          single_track_planar_model.cpp:195-332
 *   RK4                           lmpc_utils/src/utils.cpp:88-108
 *   A, B, g of the RK4 map        single_track_planar_model.cpp:377-387
 *   QP (cost, constraints)        racing_mpc.cpp:106-201, 442-543
 *   actuator boxes                single_track_planar_model.cpp:113-120,144-151
 *   safe-set unrolling + J        safe_set.cpp:116-137
 *   safe-set k-NN query           safe_set.cpp:42-54,153-180; trajectory_kd_tree.cpp:53-63
 *   pad / truncate / J - J[0]     racing_mpc.cpp:263-280
 *   node cold start               racing_mpc_node.cpp:210-235,261-292
 *
 * The QP is solved by a Mehrotra predictor-corrector interior-point method whose
 * Newton systems are solved by a Riccati recursion on the augmented state
 * z_i = [x_i; u_{i-1}] (8), input v_i = dU_i (2); the shared boundary slack (one
 * scalar coupling all knots, racing_mpc.cpp:533) is eliminated by a Schur complement.
 * The HIP kernel implements the same algorithm; this file is its serial twin.
 *
 * The row  sigma >= 0  (racing_mpc.cpp:536) is NOT carried by the iteration: it is redundant.  For any
 * feasible point with sigma < 0, replacing sigma by 0 keeps every boundary row  +-e_y - sigma <= b  satisfied
 * (they only loosen) and lowers the cost q_boundary sigma^2, so the optimum of the QP without the row has
 * sigma >= 0 and is the optimum of the reference's QP (oracle/qp.py keeps the row; tests compare against it).
 * Carried, the row is degenerate whenever the track boundary is inactive (sigma* = 0 with multiplier 0), which
 * is the usual case, and an interior-point iterate then approaches like sqrt(mu): two to three extra iterations.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lmpc_hip.h"

#define NMAX 128
#define NSLOT 11 /* two-sided slots per knot: x0..x5, u0,u1, v0,v1, e_y boundary */
#define SL_U 6
#define SL_V 8
#define SL_EY 10
#define SMAX 512 /* safe-set points */
#define GRAV 9.8 /* single_track_planar_model.cpp:18 */

/* ------------------------------------------------------------------------------------------ */
/* dynamics                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* value and partials of x_dot = f(x, u, k); Fx row-major 6x6, Fu row-major 6x2 */
static void f_partials(const lmpc_vehicle* v, const double* x, const double* u, double k, double* f,
                       double* Fx, double* Fu) {
  const double ey = x[1], phi = x[2], vx = x[3], vy = x[4], om = x[5];
  const double ul = u[0], de = u[1];
  const double m = v->m, l = v->l, lr = v->cg_ratio * l, lf = l - lr;
  const double th = tanh(ul), sech2 = 1.0 - th * th;
  const double fd = 1000.0 * ul * (0.5 * th + 0.5);
  const double fb = 1000.0 * ul * (0.5 - 0.5 * th);
  const double dfd = 1000.0 * ((0.5 * th + 0.5) + ul * 0.5 * sech2);
  const double dfb = 1000.0 * ((0.5 - 0.5 * th) - ul * 0.5 * sech2);
  const double Fxf = 0.5 * v->kd * fd + 0.5 * v->kb * fb - 0.5 * v->fr * m * GRAV * lr / l;
  const double Fxr = 0.5 * (1 - v->kd) * fd + 0.5 * (1 - v->kb) * fb - 0.5 * v->fr * m * GRAV * lf / l;
  const double dFxf = 0.5 * v->kd * dfd + 0.5 * v->kb * dfb;
  const double dFxr = 0.5 * (1 - v->kd) * dfd + 0.5 * (1 - v->kb) * dfb;
  const double vsq = vx * vx;
  const double ax = (fd + fb - 0.5 * v->cd * v->Af * vsq - v->fr * m * GRAV) / m;
  const double dax_vx = -v->cd * v->Af * vx / m;
  const double dax_ul = (dfd + dfb) / m;
  const double hl = v->h / l;
  const double Fzf = 0.5 * m * GRAV * lr / l - 0.5 * hl * m * ax + 0.25 * v->cl_f * v->rho * v->Af * vsq;
  const double Fzr = 0.5 * m * GRAV * lf / l + 0.5 * hl * m * ax + 0.25 * v->cl_r * v->rho * v->Af * vsq;
  const double dFzf_vx = -0.5 * hl * m * dax_vx + 0.5 * v->cl_f * v->rho * v->Af * vx;
  const double dFzr_vx = 0.5 * hl * m * dax_vx + 0.5 * v->cl_r * v->rho * v->Af * vx;
  const double dFzf_ul = -0.5 * hl * m * dax_ul, dFzr_ul = 0.5 * hl * m * dax_ul;
  const double den = vx + 1e-3;
  const double rf = (lf * om + vy) / den, rr = (lr * om - vy) / den;
  const double wf = 1.0 / ((1.0 + rf * rf) * den), wr = 1.0 / ((1.0 + rr * rr) * den);
  const double af = de - atan(rf), ar = atan(rr);
  const double daf_vx = rf * wf, daf_vy = -wf, daf_om = -lf * wf;
  const double dar_vx = -rr * wr, dar_vy = -wr, dar_om = lr * wr;
  const double tf = atan(v->Bf * af), tr = atan(v->Br * ar);
  const double Sf = sin(v->Cf * tf), Sr = sin(v->Cr * tr);
  const double Df = cos(v->Cf * tf) * v->Cf * v->Bf / (1.0 + v->Bf * af * v->Bf * af);
  const double Dr = cos(v->Cr * tr) * v->Cr * v->Br / (1.0 + v->Br * ar * v->Br * ar);
  const double Fyf = v->mu * Fzf * Sf, Fyr = v->mu * Fzr * Sr;
  /* dFy / d(vx, vy, om, ul, de) */
  const double dFyf[5] = {v->mu * (dFzf_vx * Sf + Fzf * Df * daf_vx), v->mu * Fzf * Df * daf_vy,
                          v->mu * Fzf * Df * daf_om, v->mu * dFzf_ul * Sf, v->mu * Fzf * Df};
  const double dFyr[5] = {v->mu * (dFzr_vx * Sr + Fzr * Dr * dar_vx), v->mu * Fzr * Dr * dar_vy,
                          v->mu * Fzr * Dr * dar_om, v->mu * dFzr_ul * Sr, 0.0};
  const double cd_ = cos(de), sd_ = sin(de), cph = cos(phi), sph = sin(phi);
  const double q = 1.0 / (1.0 - ey * k);
  const double num = vx * cph - vy * sph;
  const double drag = 0.5 * v->cd * v->rho * v->Af;
  f[0] = num * q;
  f[1] = vx * sph + vy * cph;
  f[2] = om - k * f[0];
  f[3] = (2 * Fxr + 2 * Fxf * cd_ - 2 * Fyf * sd_ - drag * vsq) / m + om * vy;
  f[4] = (2 * Fyr + 2 * Fyf * cd_ + 2 * Fxf * sd_) / m - om * vx;
  f[5] = (-2 * Fyr * lr + (2 * Fyf * cd_ + 2 * Fxf * sd_) * lf) / v->Jzz;
  memset(Fx, 0, 36 * sizeof(double));
  memset(Fu, 0, 12 * sizeof(double));
  Fx[0 * 6 + 1] = num * q * q * k;
  Fx[0 * 6 + 2] = (-vx * sph - vy * cph) * q;
  Fx[0 * 6 + 3] = cph * q;
  Fx[0 * 6 + 4] = -sph * q;
  Fx[1 * 6 + 2] = num;
  Fx[1 * 6 + 3] = sph;
  Fx[1 * 6 + 4] = cph;
  for (int c = 0; c < 6; ++c) Fx[2 * 6 + c] = -k * Fx[0 * 6 + c];
  Fx[2 * 6 + 5] += 1.0;
  for (int j = 0; j < 3; ++j) { /* vx, vy, om */
    const int c = 3 + j;
    Fx[3 * 6 + c] = (-2 * dFyf[j] * sd_) / m;
    Fx[4 * 6 + c] = (2 * dFyr[j] + 2 * dFyf[j] * cd_) / m;
    Fx[5 * 6 + c] = (-2 * dFyr[j] * lr + 2 * dFyf[j] * cd_ * lf) / v->Jzz;
  }
  Fx[3 * 6 + 3] += -2 * drag * vx / m;
  Fx[3 * 6 + 4] += om;
  Fx[3 * 6 + 5] += vy;
  Fx[4 * 6 + 3] += -om;
  Fx[4 * 6 + 5] += -vx;
  Fu[3 * 2 + 0] = (2 * dFxr + 2 * dFxf * cd_ - 2 * dFyf[3] * sd_) / m;
  Fu[4 * 2 + 0] = (2 * dFyr[3] + 2 * dFyf[3] * cd_ + 2 * dFxf * sd_) / m;
  Fu[5 * 2 + 0] = (-2 * dFyr[3] * lr + (2 * dFyf[3] * cd_ + 2 * dFxf * sd_) * lf) / v->Jzz;
  const double dvy_de = 2 * dFyf[4] * cd_ - 2 * Fyf * sd_ + 2 * Fxf * cd_;
  Fu[3 * 2 + 1] = (-2 * Fxf * sd_ - 2 * dFyf[4] * sd_ - 2 * Fyf * cd_) / m;
  Fu[4 * 2 + 1] = dvy_de / m;
  Fu[5 * 2 + 1] = dvy_de * lf / v->Jzz;
}

static void f_only(const lmpc_vehicle* v, const double* x, const double* u, double k, double* f) {
  double Fx[36], Fu[12];
  f_partials(v, x, u, k, f, Fx, Fu);
}

/* x+ = rk4(x, u, k, dt) */
void lmpc_oracle_rk4(const lmpc_vehicle* v, const double* x, const double* u, double k, double dt,
                     double* xp) {
  double k1[6], k2[6], k3[6], k4[6], xs[6];
  f_only(v, x, u, k, k1);
  if (v->integrator == LMPC_INTEGRATOR_EULER) { /* utils.cpp:110-123 */
    for (int r = 0; r < 6; ++r) xp[r] = x[r] + dt * k1[r];
    return;
  }
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt / 2.0 * k1[r];
  f_only(v, xs, u, k, k2);
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt / 2.0 * k2[r];
  f_only(v, xs, u, k, k3);
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt * k3[r];
  f_only(v, xs, u, k, k4);
  for (int r = 0; r < 6; ++r) xp[r] = x[r] + dt / 6 * (k1[r] + 2 * k2[r] + 2 * k3[r] + k4[r]);
}

/* A (6x6 row-major), B (6x2 row-major), g, xp: forward-mode chain rule through the RK4 stages */
void lmpc_oracle_linearize(const lmpc_vehicle* v, const double* x, const double* u, double k,
                           double dt, double* A, double* B, double* g, double* xp) {
  double ks[4][6], Ks[4][48]; /* Ks[s]: d k_s / d(x,u), row-major 6x8 */
  double X[48];               /* d x_s / d(x,u) */
  double xs[6], Fx[36], Fu[12];
  const double cs[4] = {0.0, 0.5, 0.5, 1.0};
  for (int s = 0; s < 4; ++s) {
    for (int r = 0; r < 6; ++r) {
      xs[r] = x[r] + (s ? cs[s] * dt * ks[s - 1][r] : 0.0);
      for (int c = 0; c < 8; ++c)
        X[r * 8 + c] = (r == c ? 1.0 : 0.0) + (s ? cs[s] * dt * Ks[s - 1][r * 8 + c] : 0.0);
    }
    f_partials(v, xs, u, k, ks[s], Fx, Fu);
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 8; ++c) {
        double acc = (c >= 6) ? Fu[r * 2 + (c - 6)] : 0.0;
        for (int j = 0; j < 6; ++j) acc += Fx[r * 6 + j] * X[j * 8 + c];
        Ks[s][r * 8 + c] = acc;
      }
  }
  const int euler = v->integrator == LMPC_INTEGRATOR_EULER; /* x+ = x + dt f(x, u): the first slope alone */
  const double w0 = euler ? dt : dt / 6, w12 = euler ? 0.0 : dt / 3, w3 = euler ? 0.0 : dt / 6;
  for (int r = 0; r < 6; ++r) {
    xp[r] = x[r] + (w0 * ks[0][r] + w12 * ks[1][r] + w12 * ks[2][r] + w3 * ks[3][r]);
    for (int c = 0; c < 8; ++c) {
      const double d = (r == c ? 1.0 : 0.0) +
                       (w0 * Ks[0][r * 8 + c] + w12 * Ks[1][r * 8 + c] + w12 * Ks[2][r * 8 + c] + w3 * Ks[3][r * 8 + c]);
      if (c < 6)
        A[r * 6 + c] = d;
      else
        B[r * 2 + (c - 6)] = d;
    }
  }
  for (int r = 0; r < 6; ++r) {
    double acc = xp[r];
    for (int c = 0; c < 6; ++c) acc -= A[r * 6 + c] * x[c];
    for (int c = 0; c < 2; ++c) acc -= B[r * 2 + c] * u[c];
    g[r] = acc;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* QP solve                                                                                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  int N, has_sigma, S;
  int max_iter;
  double tol;
  /* dynamics */
  double A[NMAX][36], B[NMAX][12], g[NMAX][6], dt[NMAX];
  /* cost: diag state Hessian + linear term per knot, 2x2 blocks, slack weight */
  double Qx[NMAX][6], qx[NMAX][6], Qu[4], Sv[4], qsig;
  /* LMPC terminal block */
  double ssx[6][SMAX], ssj[SMAX], chs2[6]; /* safe-set points CENTRED on ss0 = first point; 2*convex_hull_slack */
  double ss0[6];
  int warm;                       /* lmpc_oracle_solve_range_warm: (X_ref, U_ref) is the previous optimal plan, shifted */
  double zw[NMAX][8], vw[NMAX][2]; /* that plan in the solver's variables */
  int has_lamw;                    /* learning: the plan's simplex weights were handed in (convex_combi_optm_ref, racing_mpc.cpp:281) */
  double lamw[SMAX];               /* ... per KEPT point (the copies of a run report 0 and are ignored) */
  int S_in, ss_map[SMAX]; /* points as handed in, and which of them each kept point is (runs of identical points are one point) */
  int hard_hull; /* all-zero convex_hull_slack: the hull row is an equality, realised as the penalty limit */
  /* bounds per slot */
  double hi[NMAX][NSLOT], lo[NMAX][NSLOT];
  int act[NMAX][NSLOT][2]; /* [upper, lower] */
  /* iterate */
  double z[NMAX][8], v[NMAX][2], sigma;
  double t[NMAX][NSLOT][2], lam[NMAX][NSLOT][2];
  double lmb[SMAX], tl[SMAX], ll[SMAX]; /* LMPC: simplex weights, their slacks and multipliers */
  /* Newton step */
  double dz[NMAX][8], dv[NMAX][2], dsigma;
  double dtt[NMAX][NSLOT][2], dlam[NMAX][NSLOT][2];
  double dlmb[SMAX], dtl[SMAX], dll[SMAX];
  /* assembled Newton data */
  double Thz[NMAX][8], Thv[NMAX][2], csig[NMAX], hsig;
  /* Riccati */
  double K[NMAX][16], Hinv[NMAX][4];
  double ez[NMAX][8], ev[NMAX][2]; /* Schur vector e = R(c) */
  /* LMPC terminal elimination */
  double PT[36], Minv_cache[49];
  double tau; /* theta below which a safe-set point is kept as an explicit unknown of the terminal block */
} prob_t;

static inline double abar(const prob_t* p, int i, int k, int r) { /* Abar[k][r], k < 6 */
  return r < 6 ? p->A[i][k * 6 + r] : p->B[i][k * 2 + (r - 6)];
}

/* true cost Hessian on z_i (without barrier terms) */
static inline double Qz_entry(const prob_t* p, int i, int r, int c) {
  if (r < 6 || c < 6) return (r == c) ? p->Qx[i][r] : 0.0;
  return (i >= 1) ? p->Qu[(r - 6) * 2 + (c - 6)] : 0.0;
}

/* Riccati factorisation for the current Thz/Thv; PT = extra terminal x-block (LMPC) or NULL.
 * Backward sweep on z = [x; u_prev] (8), v = dU (2):
 *   Y = Abar' P Abar,  H = Sv + Thv + t^2 Y_uu,  G = t Y[6:8,:],  K = H^-1 G,
 *   P <- Qz + Thz + Y - G' K.        (stage 0 only needs H^-1: dz_0 = 0)                      */
static void riccati_factor(prob_t* p, const double* PT, int joseph) {
  const int N = p->N;
  double P[64], W[64], Y[64];
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 8; ++c) {
      double e = Qz_entry(p, N - 1, r, c) + (r == c ? p->Thz[N - 1][r] : 0.0);
      if (PT && r < 6 && c < 6) e += PT[r <= c ? r * 6 + c : c * 6 + r]; /* upper triangle, mirrored */
      P[r * 8 + c] = e;
    }
  for (int i = N - 2; i >= 0; --i) {
    const double t = p->dt[i];
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 8; ++c) {
        double acc = (r >= 6) ? P[r * 8 + c] : 0.0;
        for (int k = 0; k < 6; ++k) acc += abar(p, i, k, r) * P[c * 8 + k]; /* P[k][c] read as P[c][k] (kernel's row read) */
        W[r * 8 + c] = acc;
      }
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < 8; ++c) {
        double acc = (c >= 6) ? W[r * 8 + c] : 0.0;
        for (int k = 0; k < 6; ++k) acc += W[r * 8 + k] * abar(p, i, k, c);
        Y[r * 8 + c] = acc;
      }
    const double h00 = p->Sv[0] + p->Thv[i][0] + t * t * Y[6 * 8 + 6];
    const double h01 = p->Sv[1] + t * t * Y[6 * 8 + 7];
    const double h11 = p->Sv[3] + p->Thv[i][1] + t * t * Y[7 * 8 + 7];
    const double idet = 1.0 / (h00 * h11 - h01 * h01);
    double* Hi = p->Hinv[i];
    Hi[0] = h11 * idet;
    Hi[1] = -h01 * idet;
    Hi[2] = -h01 * idet;
    Hi[3] = h00 * idet;
    for (int c = 0; c < 8; ++c) {
      const double g0 = t * Y[6 * 8 + c], g1 = t * Y[7 * 8 + c];
      p->K[i][0 * 8 + c] = Hi[0] * g0 + Hi[1] * g1;
      p->K[i][1 * 8 + c] = Hi[2] * g0 + Hi[3] * g1;
    }
    if (i >= 1 && joseph) {
      /* Stabilised ("Joseph") form  P <- Qz + Thz + Phi' P Phi + K' (Sv + Thv) K,  Phi = Abar - Bbar K: the same matrix
       * as below in exact arithmetic, but a sum of positive semidefinite products.  Y - G'K subtracts two numbers of the
       * size of the largest barrier weight (1e10..1e13 late in the iteration) to leave one of the size of the cost, and
       * the Newton directions lose those digits; here the cancellation happens inside Phi, BEFORE the multiplication by
       * P.  Costs a second pair of 8x8 products, so it is used only for the last iterations (mu <= JOSEPH_MU). */
      double Phi[64], W2[64];
      for (int k = 0; k < 8; ++k)
        for (int c = 0; c < 8; ++c) {
          if (k < 6)
            Phi[k * 8 + c] = abar(p, i, k, c) - t * (p->B[i][k * 2] * p->K[i][c] + p->B[i][k * 2 + 1] * p->K[i][8 + c]);
          else
            Phi[k * 8 + c] = (k == c ? 1.0 : 0.0) - t * p->K[i][(k - 6) * 8 + c];
        }
      for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 8; ++k) acc += Phi[k * 8 + r] * P[c * 8 + k]; /* P symmetric: P[k][c] read as P[c][k] */
          W2[r * 8 + c] = acc;
        }
      const double v00 = p->Sv[0] + p->Thv[i][0], v01 = p->Sv[1], v11 = p->Sv[3] + p->Thv[i][1];
      for (int r = 0; r < 8; ++r)
        for (int c = r; c < 8; ++c) {
          double acc = 0.0;
          for (int k = 0; k < 8; ++k) acc += W2[r * 8 + k] * Phi[k * 8 + c];
          const double k0r = p->K[i][r], k1r = p->K[i][8 + r], k0c = p->K[i][c], k1c = p->K[i][8 + c];
          Y[r * 8 + c] = Qz_entry(p, i, r, c) + (r == c ? p->Thz[i][r] : 0.0) + acc + k0r * (v00 * k0c + v01 * k1c) +
                         k1r * (v01 * k0c + v11 * k1c);
        }
      for (int r = 0; r < 8; ++r)
        for (int c = r; c < 8; ++c) P[c * 8 + r] = P[r * 8 + c] = Y[r * 8 + c];
    } else if (i >= 1) {
      for (int r = 0; r < 8; ++r)
        for (int c = r; c < 8; ++c)
          P[r * 8 + c] = Qz_entry(p, i, r, c) + (r == c ? p->Thz[i][r] : 0.0) + Y[r * 8 + c] -
                         t * (Y[6 * 8 + r] * p->K[i][0 * 8 + c] + Y[7 * 8 + r] * p->K[i][1 * 8 + c]);
      /* The upper triangle is the cost-to-go; the lower one mirrors it.  Rounding makes the two computed halves differ,
       * and that antisymmetric part is NOT contracted by the recursion: it is multiplied by Abar'(.)Abar, i.e. by the
       * OPEN-loop dynamics, whose RK4 map has |eig| up to ~15-25 below 1 m/s -- left alone it reaches 1e17 within
       * twenty stages and the Newton directions are noise (N >= 40 cold starts at low speed).  Exact symmetry removes it:
       * the symmetric error is damped by the closed loop like the cost-to-go itself. */
      for (int r = 0; r < 8; ++r)
        for (int c = r + 1; c < 8; ++c) P[c * 8 + r] = P[r * 8 + c];
    }
  }
}

/* Solve min 1/2 d'H~d + q'd s.t. linearised dynamics, dz_0 = 0.  (qz, qv) -> (dz, dv) */
static void riccati_solve(const prob_t* p, double (*qz)[8], double (*qv)[2], double (*dz)[8],
                          double (*dv)[2]) {
  const int N = p->N;
  double pv[8], w[8], kk[NMAX][2];
  memcpy(pv, qz[N - 1], sizeof(pv));
  for (int i = N - 2; i >= 0; --i) {
    const double t = p->dt[i];
    for (int r = 0; r < 8; ++r) {
      double acc = (r >= 6) ? pv[r] : 0.0;
      for (int k = 0; k < 6; ++k) acc += abar(p, i, k, r) * pv[k];
      w[r] = acc;
    }
    const double h0 = qv[i][0] + t * w[6], h1 = qv[i][1] + t * w[7];
    kk[i][0] = p->Hinv[i][0] * h0 + p->Hinv[i][1] * h1;
    kk[i][1] = p->Hinv[i][2] * h0 + p->Hinv[i][3] * h1;
    if (i >= 1)
      for (int r = 0; r < 8; ++r) pv[r] = qz[i][r] + w[r] - (p->K[i][r] * h0 + p->K[i][8 + r] * h1);
  }
  memset(dz[0], 0, sizeof(double) * 8);
  for (int i = 0; i < N - 1; ++i) {
    const double t = p->dt[i];
    for (int a = 0; a < 2; ++a) {
      double acc = -kk[i][a];
      for (int c = 0; c < 8; ++c) acc -= p->K[i][a * 8 + c] * dz[i][c];
      dv[i][a] = acc;
    }
    const double du0 = dz[i][6] + t * dv[i][0], du1 = dz[i][7] + t * dv[i][1];
    for (int r = 0; r < 6; ++r) {
      double acc = p->B[i][r * 2] * du0 + p->B[i][r * 2 + 1] * du1;
      for (int c = 0; c < 6; ++c) acc += p->A[i][r * 6 + c] * dz[i][c];
      dz[i + 1][r] = acc;
    }
    dz[i + 1][6] = du0;
    dz[i + 1][7] = du1;
  }
}

/* value of the quantity slot (i, sl) constrains (without the sigma term) */
static inline double slot_val(double (*z)[8], double (*v)[2], int i, int sl) {
  if (sl < SL_U) return z[i][sl];
  if (sl < SL_V) return z[i][6 + (sl - SL_U)];
  if (sl < SL_EY) return v[i][sl - SL_V];
  return z[i][1];
}

/* LMPC terminal block -------------------------------------------------------------------------
 * Terminal variables lambda (S) and eps = x_T - U lambda (eliminated), U = SS (6 x S):
 *   cost  ss_j' lambda + eps' D eps,   1' lambda = 1,   lambda >= 0        (racing_mpc.cpp:484-504)
 * With E = 2D and barrier weights Th = diag(theta_l), the Newton step in lambda for a given
 * terminal-state step dx is
 *     dl = Z (U'E dx - bl) + M^-1 1 r1 / s11,   M = Th + U'EU,   Z = M^-1 - M^-1 1 1'M^-1 / s11,
 * (bl: gradient wrt lambda, r1 = 1 - 1'lambda, s11 = 1'M^-1 1) and eliminating it leaves a quadratic
 * in dx:  1/2 dx' PT dx + pT' dx  with  PT = E - E U Z U' E.
 * Woodbury (M^-1 = Th^-1 - Th^-1 U' F^-1 U Th^-1, F = E^-1 + T, T = U Th^-1 U') reduces everything
 * to 6x6 algebra on three kinds of sums over the S points:
 *     T = U Th^-1 U' (21 sums),  a = U Th^-1 1 (6),  sth = sum 1/theta (1)      -- per factorisation
 *     beta = U Th^-1 bl (6),  sbl = sum bl/theta (1)                             -- per right-hand side
 * which is what the GPU kernel reduces across the wave:
 *     U M^-1 U' = T - T F^-1 T =: G,   U M^-1 1 = a - T F^-1 a =: g,   s11 = sth - a'F^-1 a,
 *     PT = E - E (G - g g'/s11) E,
 *     U dl0 = -(beta - T F^-1 beta) + g (sbl - a'F^-1 beta + r1)/s11,   pT = -E U dl0,
 *     dl_j = (r_j - u_j'F^-1 gamma)/theta_j - Mi1_j (oMr - r1)/s11,  r_j = u_j'E dx - bl_j,
 *            gamma = T E dx - beta,  oMr = a'E dx - sbl - a'F^-1 gamma,  Mi1_j = (1 - u_j'F^-1 a)/theta_j. */
/* Two-level elimination -- no division by a small theta.  The points split into
 *     B: theta_j >= tau      eliminated through Theta_B^-1 (Woodbury as above, on the sums over B only):
 *                            T_B = U_B Th_B^-1 U_B', a_B = U_B Th_B^-1 1, s_B = sum_B 1/theta, F_B = E^-1 + T_B
 *     A: theta_j <  tau      the (few) points whose lambda stays positive: theta = l/t -> 0 there.  They are kept as
 *                            explicit unknowns of a small dense system, at most MA_MAX of them (the smallest theta).
 * With W_A = F_B^-1 U_A (6 x m) and C_A = Theta_A + U_A' F_B^-1 U_A (m x m, the Schur complement of M_BB in M):
 *     F^-1      = F_B^-1 - W_A C_A^-1 W_A'                       (F over all points)
 *     g = E U M^-1 1 = F_B^-1 z1,  z1 = a_B + U_A x1,  x1 = C_A^-1 (1_A - W_A' a_B)
 *     s11       = 1_A' x1 + s_B - a_B' F_B^-1 z1
 *     PT        = F^-1 + g g'/s11
 * and for a right-hand side r (r_j = u_j'E dx - bl_j) with simplex residual r1:
 *     beta = U_B Th_B^-1 r_B, sig = 1'Th_B^-1 r_B,  x_A = C_A^-1 (r_A - W_A' beta),  z = beta + U_A x_A,
 *     nu = (1_A' x_A + sig - a_B' F_B^-1 z - r1)/s11,   h = E U dl = F_B^-1 z - nu g,
 *     dl_A = x_A - nu x1,   dl_j = (r_j - nu - u_j' h)/theta_j  (j in B).
 * cond(F_B) <= 1 + E |u|^2 / tau whatever the iteration does, and theta_A enters only as an addend on the diagonal of
 * C_A.  (Round 1 floored theta inside the Newton matrix instead -- a proximal term on d lambda: it kept F conditioned but
 * damped the step of exactly the points that matter, so problems with two or three supporting points crawled or stalled,
 * and at IAC scale the iterate stopped 1e-3 .. 1e-2 from the optimum with mu -> 0.) */
/* complementarity below which the factorisation switches to the stabilised form (riccati_factor) */
#define NBHD_GAMMA 1e-2
#define NBHD_TRIALS 3
#ifndef JOSEPH_MU
#define JOSEPH_MU 1e-8
#endif
/* complementarity below which a step that does not lower it ends the solve (ipm_solve) */
#define STALL_MU 1e-9
#ifndef STALL_STEP
#define STALL_STEP 1e-6 /* scaled size of the last primal step above which a stalled, unpolished iterate is not vouched for (ipm_solve's exit) */
#endif
#define MA_MAX 6 /* four until round 5: a five-lap safe set has optima that blend one point per lap, and the fifth point then went
                  * through 1 / theta -> 1e12 (cond(F_B) with it): answers 1e-2 off, reported OPTIMAL (csrc/lmpc_solve_kernel.hip) */
#define TAU_REL 1e-5 /* tau = TAU_REL * max_j u_j'E u_j: cond(F_B) <= ~1e5 */
typedef struct {
  double Fi[36];       /* F^-1 over all points */
  double FBi[36];      /* F_B^-1 */
  double aB[6], sB;    /* sums over B */
  double g[6], s11;
  int m, idx[MA_MAX];  /* the explicit points */
  double W[6][MA_MAX]; /* W_A = F_B^-1 U_A */
  double Lc[MA_MAX][MA_MAX]; /* Cholesky factor of C_A */
  double Ci[MA_MAX][MA_MAX]; /* C_A^-1 */
  double x1[MA_MAX];
  unsigned char inA[SMAX];
} term_t;

static void sym_inv6(const double* F, double* Fi) { /* Cholesky inverse of SPD 6x6 */
  double Lc[36] = {0};
  for (int j = 0; j < 6; ++j) {
    double d = F[j * 6 + j];
    for (int k = 0; k < j; ++k) d -= Lc[j * 6 + k] * Lc[j * 6 + k];
    d = sqrt(d);
    Lc[j * 6 + j] = d;
    for (int i = j + 1; i < 6; ++i) {
      double s = F[i * 6 + j];
      for (int k = 0; k < j; ++k) s -= Lc[i * 6 + k] * Lc[j * 6 + k];
      Lc[i * 6 + j] = s / d;
    }
  }
  for (int c = 0; c < 6; ++c) { /* solve L L' x = e_c */
    double y[6], xv[6];
    for (int i = 0; i < 6; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= Lc[i * 6 + k] * y[k];
      y[i] = s / Lc[i * 6 + i];
    }
    for (int i = 5; i >= 0; --i) {
      double s = y[i];
      for (int k = i + 1; k < 6; ++k) s -= Lc[k * 6 + i] * xv[k];
      xv[i] = s / Lc[i * 6 + i];
    }
    for (int i = 0; i < 6; ++i) Fi[i * 6 + c] = xv[i];
  }
}

static void mv6(const double* M, const double* x, double* y) {
  for (int r = 0; r < 6; ++r) {
    double s = 0;
    for (int c = 0; c < 6; ++c) s += M[r * 6 + c] * x[c];
    y[r] = s;
  }
}

/* x <- C_A^-1 x with the explicit inverse (term_factor forms it column by column from the Cholesky factor, as the kernel does:
 * there one lane per row applies it, which is why it is not a pair of substitutions) */
static void solve_CA(const term_t* tm, double* x) {
  const int m = tm->m;
  double y[MA_MAX];
  for (int a = 0; a < m; ++a) {
    double s = 0.0;
    for (int b = 0; b < m; ++b) s += tm->Ci[a][b] * x[b];
    y[a] = s;
  }
  for (int a = 0; a < m; ++a) x[a] = y[a];
}

static void term_factor(prob_t* p, term_t* tm, const double* thl, double* PT) {
  const int S = p->S;
  /* the explicit set: the (at most MA_MAX) smallest theta below tau, ties to the lower index */
  memset(tm->inA, 0, sizeof(tm->inA));
  tm->m = 0;
  for (int q = 0; q < MA_MAX; ++q) {
    int best = -1;
    for (int j = 0; j < S; ++j)
      if (!tm->inA[j] && thl[j] < p->tau && (best < 0 || thl[j] < thl[best])) best = j;
    if (best < 0) break;
    tm->inA[best] = 1;
    tm->idx[tm->m++] = best;
  }
  const int m = tm->m;
  double F[36], T[36] = {0};
  tm->sB = 0.0;
  for (int r = 0; r < 6; ++r) tm->aB[r] = 0.0;
  for (int j = 0; j < S; ++j) {
    if (tm->inA[j]) continue;
    const double it = 1.0 / thl[j];
    tm->sB += it;
    for (int r = 0; r < 6; ++r) {
      tm->aB[r] += p->ssx[r][j] * it;
      for (int c = 0; c < 6; ++c) T[r * 6 + c] += p->ssx[r][j] * p->ssx[c][j] * it;
    }
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) /* a zero hull-slack weight (racing_mpc.cpp:497: that component of eps is free): E_k^-1 -> "infinite" */
      F[r * 6 + c] = T[r * 6 + c] + (r == c ? 1.0 / fmax(p->chs2[r], 1e-30) : 0.0);
  sym_inv6(F, tm->FBi);
  double C[MA_MAX][MA_MAX];
  for (int a = 0; a < m; ++a)
    for (int r = 0; r < 6; ++r) {
      double s = 0.0;
      for (int c = 0; c < 6; ++c) s += tm->FBi[r * 6 + c] * p->ssx[c][tm->idx[a]];
      tm->W[r][a] = s;
    }
  for (int a = 0; a < m; ++a)
    for (int b = 0; b <= a; ++b) {
      double s = (a == b) ? thl[tm->idx[a]] : 0.0;
      for (int r = 0; r < 6; ++r) s += p->ssx[r][tm->idx[a]] * tm->W[r][b];
      C[a][b] = s;
    }
  double jit = 0.0; /* identical points (the padding repeats the last one) make C_A singular as theta -> 0 */
  for (int a = 0; a < m; ++a) jit += C[a][a];
  jit *= 1e-13;
  for (int a = 0; a < m; ++a) {
    for (int b = 0; b <= a; ++b) {
      double s = C[a][b] + (a == b ? jit : 0.0);
      for (int k = 0; k < b; ++k) s -= tm->Lc[a][k] * tm->Lc[b][k];
      tm->Lc[a][b] = (a == b) ? sqrt(s) : s / tm->Lc[b][b];
    }
  }
  for (int c = 0; c < m; ++c) { /* C_A^-1, column c: L L' x = e_c */
    double y[MA_MAX], x[MA_MAX];
    for (int i = 0; i < m; ++i) {
      double t = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) t -= tm->Lc[i][k] * y[k];
      y[i] = t / tm->Lc[i][i];
    }
    for (int i = m - 1; i >= 0; --i) {
      double t = y[i];
      for (int k = i + 1; k < m; ++k) t -= tm->Lc[k][i] * x[k];
      x[i] = t / tm->Lc[i][i];
    }
    for (int i = 0; i < m; ++i) tm->Ci[i][c] = x[i];
  }
  /* F^-1 = F_B^-1 - W C^-1 W' (column by column) */
  for (int c = 0; c < 6; ++c) {
    double x[MA_MAX];
    for (int a = 0; a < m; ++a) x[a] = tm->W[c][a];
    solve_CA(tm, x);
    for (int r = 0; r < 6; ++r) {
      double s = tm->FBi[r * 6 + c];
      for (int a = 0; a < m; ++a) s -= tm->W[r][a] * x[a];
      tm->Fi[r * 6 + c] = s;
    }
  }
  /* x1, z1, g, s11 */
  double z1[6];
  for (int a = 0; a < m; ++a) {
    double s = 1.0;
    for (int r = 0; r < 6; ++r) s -= tm->W[r][a] * tm->aB[r];
    tm->x1[a] = s;
  }
  solve_CA(tm, tm->x1);
  for (int r = 0; r < 6; ++r) {
    z1[r] = tm->aB[r];
    for (int a = 0; a < m; ++a) z1[r] += p->ssx[r][tm->idx[a]] * tm->x1[a];
  }
  mv6(tm->FBi, z1, tm->g);
  tm->s11 = tm->sB;
  for (int a = 0; a < m; ++a) tm->s11 += tm->x1[a];
  for (int r = 0; r < 6; ++r) tm->s11 -= tm->aB[r] * tm->g[r];
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) PT[r * 6 + c] = tm->Fi[r * 6 + c] + tm->g[r] * tm->g[c] / tm->s11;
}

/* the solve for a right-hand side r_j = (e'u_j) - bl_j (e = E dx, or 0) and simplex residual r1: fills dl (all points)
 * and h = E U dl */
static void term_solve(const prob_t* p, const term_t* tm, const double* thl, const double* bl, double r1, const double* e,
                       double* dl, double* h) {
  const int S = p->S, m = tm->m;
  double beta[6] = {0}, sig = 0.0, xA[MA_MAX], z[6], Fz[6];
  for (int j = 0; j < S; ++j) {
    if (tm->inA[j]) continue;
    double rj = -bl[j];
    if (e)
      for (int k = 0; k < 6; ++k) rj += p->ssx[k][j] * e[k];
    const double w = rj / thl[j];
    sig += w;
    for (int k = 0; k < 6; ++k) beta[k] += p->ssx[k][j] * w;
  }
  for (int a = 0; a < m; ++a) {
    const int j = tm->idx[a];
    double s = -bl[j];
    if (e)
      for (int k = 0; k < 6; ++k) s += p->ssx[k][j] * e[k];
    for (int r = 0; r < 6; ++r) s -= tm->W[r][a] * beta[r];
    xA[a] = s;
  }
  solve_CA(tm, xA);
  double num = sig - r1;
  for (int r = 0; r < 6; ++r) {
    z[r] = beta[r];
    for (int a = 0; a < m; ++a) z[r] += p->ssx[r][tm->idx[a]] * xA[a];
  }
  mv6(tm->FBi, z, Fz);
  for (int a = 0; a < m; ++a) num += xA[a];
  for (int r = 0; r < 6; ++r) num -= tm->aB[r] * Fz[r];
  const double nu = num / tm->s11;
  for (int r = 0; r < 6; ++r) h[r] = Fz[r] - nu * tm->g[r];
  if (!dl) return;
  for (int j = 0; j < S; ++j) {
    if (tm->inA[j]) continue;
    double rj = -bl[j] - nu;
    for (int k = 0; k < 6; ++k) rj += p->ssx[k][j] * ((e ? e[k] : 0.0) - h[k]);
    dl[j] = rj / thl[j];
  }
  for (int a = 0; a < m; ++a) dl[tm->idx[a]] = xA[a] - nu * tm->x1[a];
}

/* per right-hand side: the terminal-gradient contribution pT = -E U dl0 (dl0: the step at dx = 0) */
static void term_rhs(const prob_t* p, term_t* tm, const double* thl, const double* bl, double r1, double* pT) {
  double h[6];
  term_solve(p, tm, thl, bl, r1, NULL, NULL, h);
  for (int r = 0; r < 6; ++r) pT[r] = -h[r];
}

static void term_dl_of_dx(const prob_t* p, const term_t* tm, const double* thl, const double* bl, double r1,
                          const double* dx, double* dl) {
  double e[6], h[6];
  for (int k = 0; k < 6; ++k) e[k] = p->chs2[k] * dx[k];
  term_solve(p, tm, thl, bl, r1, e, dl, h);
}

/* ---- shared Newton machinery -------------------------------------------------------------
 * Rows (one-sided inequality constraints) are addressed as [knot][slot][side]; side 0 is the
 * upper row  (+val - sigma <= hi), side 1 the lower row (-val - sigma <= -lo); the sigma term
 * exists only for the e_y boundary slot.  A Newton system is defined by per-row weights `th`
 * (added as th * c c' to the Hessian) and per-row coefficients `cf` (added as cf * c to the
 * gradient); the interior-point phase and the active-set polish differ only in how they
 * choose (th, cf).                                                                           */
typedef double rows_t[NMAX][NSLOT][2];

typedef struct {
  rows_t th, cf, rd;
  double thl[SMAX], cfl[SMAX], rdl[SMAX], r1, eps_T[6]; /* LMPC terminal rows                  */
  double gz[NMAX][8], gv[NMAX][2], gsig; /* gradient of the cost at the iterate                */
  double az[NMAX][8], av[NMAX][2];
  double ce;
  term_t tm;
  double bl[SMAX];
  int frozen_lambda; /* start point: lambda held fixed, terminal cost = eps'D eps only */
} work_t;

/* gradient of the objective at the current iterate (no multipliers) */
static void cost_gradient(const prob_t* p, work_t* w) {
  const int N = p->N, S = p->S;
  w->gsig = p->qsig * p->sigma;
  for (int i = 0; i < N; ++i) {
    for (int r = 0; r < 6; ++r) w->gz[i][r] = p->Qx[i][r] * p->z[i][r] + p->qx[i][r];
    for (int a = 0; a < 2; ++a)
      w->gz[i][6 + a] = (i >= 1) ? p->Qu[a * 2] * p->z[i][6] + p->Qu[a * 2 + 1] * p->z[i][7] : 0.0;
    for (int a = 0; a < 2; ++a)
      w->gv[i][a] = (i < N - 1) ? p->Sv[a * 2] * p->v[i][0] + p->Sv[a * 2 + 1] * p->v[i][1] : 0.0;
  }
  if (S) {
    double sl_ = 0;
    for (int j = 0; j < S; ++j) sl_ += p->lmb[j];
    w->r1 = 1.0 - sl_;
    for (int k = 0; k < 6; ++k) {
      double s = p->z[N - 1][k] - p->ss0[k]; /* centred: x_T - ss0 - (SS - ss0 1') lambda, valid as 1'lambda = 1 */
      for (int j = 0; j < S; ++j) s -= p->ssx[k][j] * p->lmb[j];
      w->eps_T[k] = s;
      w->gz[N - 1][k] += p->chs2[k] * s;
    }
  }
}

/* constraint value c'y - d of a row (without slack) */
static inline double row_res(const prob_t* p, int i, int sl, int sd) {
  const double val = slot_val((double (*)[8])p->z, (double (*)[2])p->v, i, sl);
  const double sg = (sl == SL_EY && p->has_sigma) ? p->sigma : 0.0;
  return sd == 0 ? (val - sg - p->hi[i][sl]) : (-val - sg + p->lo[i][sl]);
}

/* reduced-gradient stationarity residual for multipliers lam (adjoint sweep) */
static double stationarity(const prob_t* p, const work_t* w, const rows_t lam, const double* ll) {
  const int N = p->N, S = p->S;
  double pi[8] = {0}, wv[8], rg = 0.0, gs = w->gsig;
  for (int i = N - 1; i >= 0; --i) {
    double gzl[8], gvl[2];
    memcpy(gzl, w->gz[i], sizeof(gzl));
    gvl[0] = w->gv[i][0];
    gvl[1] = w->gv[i][1];
    for (int sl = 0; sl < NSLOT; ++sl) {
      const double lu = p->act[i][sl][0] ? lam[i][sl][0] : 0.0, ld = p->act[i][sl][1] ? lam[i][sl][1] : 0.0;
      const double dl_ = lu - ld;
      if (sl < SL_U)
        gzl[sl] += dl_;
      else if (sl < SL_V)
        gzl[6 + sl - SL_U] += dl_;
      else if (sl < SL_EY)
        gvl[sl - SL_V] += dl_;
      else {
        gzl[1] += dl_;
        if (p->has_sigma) gs -= lu + ld;
      }
    }
    if (i == N - 1) {
      memcpy(pi, gzl, sizeof(pi));
    } else {
      for (int r = 0; r < 8; ++r) {
        double acc = (r >= 6) ? pi[r] : 0.0;
        for (int k = 0; k < 6; ++k) acc += abar(p, i, k, r) * pi[k];
        wv[r] = acc;
      }
      for (int a = 0; a < 2; ++a) {
        const double rv = gvl[a] + p->dt[i] * wv[6 + a];
        if (fabs(rv) > rg) rg = fabs(rv);
      }
      for (int r = 0; r < 8; ++r) pi[r] = gzl[r] + wv[r];
    }
  }
  if (p->has_sigma && fabs(gs) > rg) rg = fabs(gs);
  if (S) {
    double rl[SMAX], nu = 0;
    for (int j = 0; j < S; ++j) {
      double s = p->ssj[j] - ll[j];
      for (int k = 0; k < 6; ++k) s -= p->ssx[k][j] * p->chs2[k] * w->eps_T[k];
      rl[j] = s;
      nu -= s;
    }
    nu /= S;
    for (int j = 0; j < S; ++j)
      if (fabs(rl[j] + nu) > rg) rg = fabs(rl[j] + nu);
  }
  return rg;
}

/* Hessian side: weights -> diagonal additions, sigma coupling; factorise. */
static void newton_factor(prob_t* p, work_t* w, int joseph) {
  const int N = p->N, S = p->S;
  p->hsig = p->qsig;
  for (int i = 0; i < N; ++i) {
    for (int r = 0; r < 8; ++r) p->Thz[i][r] = 0.0;
    p->Thv[i][0] = p->Thv[i][1] = 0.0;
    p->csig[i] = 0.0;
    for (int sl = 0; sl < NSLOT; ++sl) {
      const double thu = p->act[i][sl][0] ? w->th[i][sl][0] : 0.0;
      const double thd = p->act[i][sl][1] ? w->th[i][sl][1] : 0.0;
      if (sl < SL_U)
        p->Thz[i][sl] += thu + thd;
      else if (sl < SL_V)
        p->Thz[i][6 + sl - SL_U] += thu + thd;
      else if (sl < SL_EY)
        p->Thv[i][sl - SL_V] += thu + thd;
      else {
        p->Thz[i][1] += thu + thd;
        if (p->has_sigma) {
          p->csig[i] = thd - thu;
          p->hsig += thu + thd;
        }
      }
    }
  }
  double PT[36] = {0};
  if (S && !w->frozen_lambda) term_factor(p, &w->tm, w->thl, PT);
  if (S && w->frozen_lambda)
    for (int k = 0; k < 6; ++k) PT[k * 6 + k] = p->chs2[k];
  riccati_factor(p, S ? PT : NULL, joseph);
  w->ce = 0.0;
  if (p->has_sigma) {
    for (int i = 0; i < N; ++i) {
      memset(w->az[i], 0, sizeof(double) * 8);
      w->az[i][1] = (i >= 1) ? p->csig[i] : 0.0;
      w->av[i][0] = w->av[i][1] = 0.0;
    }
    riccati_solve(p, w->az, w->av, p->ez, p->ev);
    for (int i = 1; i < N; ++i) w->ce += p->csig[i] * p->ez[i][1];
  }
}

/* Gradient side: cost gradient + cf * c, solve for (dz, dv, dsigma[, dlmb]). */
static void newton_solve(prob_t* p, work_t* w) {
  const int N = p->N, S = p->S;
  double qsg = w->gsig;
  for (int i = 0; i < N; ++i) {
    memcpy(w->az[i], w->gz[i], sizeof(double) * 8);
    w->av[i][0] = w->gv[i][0];
    w->av[i][1] = w->gv[i][1];
    for (int sl = 0; sl < NSLOT; ++sl) {
      const double cu = p->act[i][sl][0] ? w->cf[i][sl][0] : 0.0;
      const double cd = p->act[i][sl][1] ? w->cf[i][sl][1] : 0.0;
      const double dlt = cu - cd;
      if (sl < SL_U)
        w->az[i][sl] += dlt;
      else if (sl < SL_V)
        w->az[i][6 + sl - SL_U] += dlt;
      else if (sl < SL_EY)
        w->av[i][sl - SL_V] += dlt;
      else {
        w->az[i][1] += dlt;
        if (p->has_sigma) qsg -= cu + cd;
      }
    }
  }
  if (S && !w->frozen_lambda) {
    double pT[6];
    for (int j = 0; j < S; ++j) {
      double sgr = p->ssj[j] - w->cfl[j];
      for (int k = 0; k < 6; ++k) sgr -= p->ssx[k][j] * p->chs2[k] * w->eps_T[k];
      w->bl[j] = sgr;
    }
    term_rhs(p, &w->tm, w->thl, w->bl, w->r1, pT);
    for (int k = 0; k < 6; ++k) w->az[N - 1][k] += pT[k];
  }
  riccati_solve(p, w->az, w->av, p->dz, p->dv);
  if (p->has_sigma) {
    double ca = 0.0;
    for (int i = 1; i < N; ++i) ca += p->csig[i] * p->dz[i][1];
    p->dsigma = -(qsg + ca) / (p->hsig + w->ce);
    for (int i = 0; i < N; ++i) {
      for (int r = 0; r < 8; ++r) p->dz[i][r] += p->dsigma * p->ez[i][r];
      if (i < N - 1) {
        p->dv[i][0] += p->dsigma * p->ev[i][0];
        p->dv[i][1] += p->dsigma * p->ev[i][1];
      }
    }
  } else {
    p->dsigma = 0.0;
  }
  if (S && !w->frozen_lambda) term_dl_of_dx(p, &w->tm, w->thl, w->bl, w->r1, p->dz[N - 1], p->dlmb);
  if (S && w->frozen_lambda)
    for (int j = 0; j < S; ++j) p->dlmb[j] = 0.0;
}

static void primal_update(prob_t* p, double alpha) {
  const int N = p->N;
  for (int i = 0; i < N; ++i) {
    if (i >= 1)
      for (int r = 0; r < 8; ++r) p->z[i][r] += alpha * p->dz[i][r];
    if (i < N - 1) {
      p->v[i][0] += alpha * p->dv[i][0];
      p->v[i][1] += alpha * p->dv[i][1];
    }
  }
  if (p->has_sigma) p->sigma += alpha * p->dsigma;
  for (int j = 0; j < p->S; ++j) p->lmb[j] += alpha * p->dlmb[j];
}
/* the size of the step alpha (dz, dv) in the reference's scaled units (racing_mpc.cpp:36-37), as the kernel's slots measure it */
static double scaled_step(const prob_t* p, double alpha) {
  static const double isx[10] = {5e-4, 0.1, 10.0, 0.0125, 0.5, 0.5, 0.1, 1.0 / 0.3, 0.1, 1.0 / 0.3};
  double step = 0.0;
  for (int i = 0; i < p->N; ++i) {
    if (i >= 1)
      for (int r = 0; r < 8; ++r) step = fmax(step, fabs(alpha * p->dz[i][r]) * isx[r]);
    if (i < p->N - 1) step = fmax(step, fmax(fabs(alpha * p->dv[i][0]) * isx[8], fabs(alpha * p->dv[i][1]) * isx[9]));
  }
  return step;
}

/* ---- active-set polish -------------------------------------------------------------------------
 * What OSQP's polish = true does for the reference (racing_mpc.cpp:90-95): guess the active rows from the interior
 * point's slacks and multipliers, solve the equality-constrained QP they define, keep the answer if it passes the KKT
 * test.  Here, on the machinery of the iteration:
 *   held rows  = rows with lam > t (for the simplex rows lambda_j >= 0: ll > tl, i.e. lambda_j pinned at 0);
 *   each round = one factorisation with weight POLISH_THETA on the held rows and 0 on the others (stabilised form),
 *                then POLISH_STEPS multiplier (augmented-Lagrangian) steps on that factor -- gradient
 *                cf = y + THETA * residual on the held rows, FULL Newton step, y <- y + THETA * (residual + c'dz),
 *                the residual taken before the step and the row's own increment added (not re-read after the update:
 *                in single precision the stored value is rounded, its increment is not) -- which converge like
 *                (curvature / THETA)^k, i.e. at once;
 *   verify     : held rows met to POLISH_FEAS with multiplier >= -POLISH_DUAL, every other row satisfied to POLISH_FEAS,
 *                the last step below POLISH_STEP_TOL (scaled units);
 *   repair     : if a held row has a negative multiplier, release rows -- only those the interior point did not hold
 *                firmly (lam < POLISH_STRONG t) if there are such, otherwise the most negative ones (a wrong row drags
 *                its neighbours' multipliers below zero: releasing everything negative wrecks the set) -- and only when
 *                no multiplier is negative, hold the rows the new point violates.  Every round restarts from the
 *                interior-point iterate (held rows from its lam, added rows from 0).
 * The interior point calls it once when mu <= POLISH_MU with rows feasible to POLISH_RD (about two iterations before its
 * own tolerance) and again at its own exit if that attempt was refused.  A refused polish leaves the iterate and (t, lam)
 * untouched.  An accepted one is the optimum to the accuracy of the linear solves (1e-9 .. 1e-12 scaled on the bench
 * distributions, degenerate problems included: a weakly active row may sit on either side of the guess, the solution is
 * the same).  The terminal block can hold at most MA_MAX free simplex weights explicitly; with more the polish is not
 * attempted. */
#define POLISH_THETA 1e8
#ifndef POLISH_MU /* (scratch/r5/twin_variant.py builds variants with -D) */
#define POLISH_MU 1e-8 /* (1e-7 was measured in round 6 and lost: polish_limits<double>::mu_early of the kernel has the numbers) */
#endif
#ifndef POLISH_RD
#define POLISH_RD 1e-6
#endif
#define POLISH_ROUNDS 4
#define POLISH_EXIT 256        /* flag in polish_rounds' max_rounds: the attempt at the interior point's exit */
#define POLISH_EXIT_GAMMA 1e-2 /* ... holds a row from lam > 1e-2 t on (the early and the warm attempts: lam > t) */
#define POLISH_STEPS 6 /* at most (round 6: 4 until then; polish_limits<double>::steps of the kernel says why); the loop stops after the second when that one moved the iterate by <= POLISH_STEP_OK */
#define POLISH_STEP_OK 1e-7
#ifndef WARM_ROUNDS
#define WARM_ROUNDS 2       /* repairs a warm start may spend before the cold start takes over */
#endif
#define WARM_ACT 1e-9       /* a box row of the plan counts as active within this slack (physical units; a polished plan: ~1e-16) */
#define WARM_ACT_EY 1e-3    /* boundary rows: their bounds move with the shift (the track half-width over one knot's travel) */
#define POLISH_FEAS 1e-9
#define POLISH_DUAL 1e-7
#ifndef POLISH_DUAL_L
#define POLISH_DUAL_L 1e-8 /* the same test on the simplex rows' multipliers, a decade tighter (round 6; polish_limits<double>::dual_l of the kernel says why) */
#endif
#define POLISH_STRONG 1e3
#define POLISH_STEP_TOL 1e-6 /* the last multiplier step, in the reference's scaled units (racing_mpc.cpp:36-37): converged
                              * steps are 1e-7 .. 1e-10, the ones this is there to catch 1e-3 .. 1e-1.  Round 5: until then exactly
                              * two steps and 1e-5.  A set repaired with rows that start from a zero multiplier, or two sigma-coupled
                              * boundary rows, converges like 0.1 per step: the second step was still 1.5e-5 .. 1.7e-4 and the
                              * attempt was refused -- at N = 80 / learning N = 60 that left the interior point's own answer, 9e-6 /
                              * 2e-5 from the dense optimum (profiles/r04_fullsize_parity.txt, 1.1e-6 / 1.7e-5 kernel against twin) --
                              * while an attempt accepted at 1e-5 with that rate is 1e-6 off.  Now: up to four steps, stop at 1e-7,
                              * accept at 1e-6, four rounds.  On the bench distributions against the dense optimum (scratch/r5/
                              * cmp_cache.py): worst 2e-7 at every horizon, mean iterations -0.3 %, maximum 18 -> 14 at N = 20. */
typedef struct {
  double z[NMAX][8], v[NMAX][2], sigma, lmb[SMAX];
  rows_t y;
  double yl[SMAX], resl[SMAX];
  rows_t res;
  unsigned char held[NMAX][NSLOT][2], heldl[SMAX];
  int noise;
} polish_t;

/* (no auto-vectorisation: gcc 11 at -O3 -march=x86-64-v3 miscompiles the mixed int / double classification loop) */
__attribute__((optimize("no-tree-vectorize"))) static int polish_rounds(prob_t* p, work_t* w, polish_t* q, int m_rows, int* rounds_out,
                                                                         double* mu_out, int max_rounds) {
  const int N = p->N, S = p->S;
  /* POLISH_EXIT in max_rounds marks the attempt at the interior point's EXIT (round 6).  There a row with lam ~ t is a weakly
   * active one (lam t = mu <= 1e-9: both ~ 1e-5 .. 1e-7, where an inactive row has lam / t = mu / t^2 <= 1e-3 unless its slack is
   * itself below 1e-3), and which side of lam = t it falls on is rounding: the kernel with the fused factorisation put one such row of
   * one problem of tests/dispatch_sweep.py (BARC tracking, N = 51) on the free side where this twin held it -- the free row was then
   * violated by the equality-constrained solve, its neighbours' multipliers went negative, the repairs released THEM, and the
   * interior point's own iterate (3e-6 from the dense optimum) stood as OPTIMAL.  In doubt a row is HELD: a weakly active row that is
   * held costs nothing (multiplier ~ 0 >= -POLISH_DUAL), one that should be free comes out with a negative multiplier and is released
   * by the next round.  The early attempt (mu <= 1e-8) keeps lam > t: there the ratio of an inactive row is still 1e-2 .. 1. */
  const double gam = (max_rounds & POLISH_EXIT) ? POLISH_EXIT_GAMMA : 1.0;
  max_rounds &= ~POLISH_EXIT;
  memcpy(q->z, p->z, sizeof(q->z));
  memcpy(q->v, p->v, sizeof(q->v));
  memcpy(q->lmb, p->lmb, sizeof(q->lmb));
  q->sigma = p->sigma;
  for (int i = 0; i < N; ++i)
    for (int sl = 0; sl < NSLOT; ++sl)
      for (int sd = 0; sd < 2; ++sd) q->held[i][sl][sd] = p->act[i][sl][sd] && p->lam[i][sl][sd] > gam * p->t[i][sl][sd];
  for (int j = 0; j < S; ++j) q->heldl[j] = p->ll[j] > p->tl[j];
  if (getenv("LMPC_ORACLE_POLISH_TRACE")) {
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd)
          if (p->act[i][sl][sd] && p->lam[i][sl][sd] > 1e-3 * p->t[i][sl][sd] && p->lam[i][sl][sd] < 1e3 * p->t[i][sl][sd])
            fprintf(stderr, "   borderline row (%d,%d,%d): lam %.3e t %.3e held %d\n", i, sl, sd, p->lam[i][sl][sd], p->t[i][sl][sd], q->held[i][sl][sd]);
  }
  int ok = 0;
  q->noise = 0;
  for (int round = 0; round < max_rounds && !ok; ++round) {
    int nfree = 0;
    for (int j = 0; j < S; ++j) nfree += !q->heldl[j];
    if (S && nfree > MA_MAX) break;
    ++*rounds_out;
    memcpy(p->z, q->z, sizeof(q->z));
    memcpy(p->v, q->v, sizeof(q->v));
    memcpy(p->lmb, q->lmb, sizeof(q->lmb));
    p->sigma = q->sigma;
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd) {
          const int h = q->held[i][sl][sd];
          w->th[i][sl][sd] = h ? POLISH_THETA : 0.0;
          q->y[i][sl][sd] = (h && p->lam[i][sl][sd] > p->t[i][sl][sd]) ? p->lam[i][sl][sd] : 0.0;
        }
    for (int j = 0; j < S; ++j) {
      w->thl[j] = q->heldl[j] ? POLISH_THETA : 0.0;
      q->yl[j] = (q->heldl[j] && p->ll[j] > p->tl[j]) ? p->ll[j] : 0.0;
    }
    cost_gradient(p, w);
    newton_factor(p, w, 1);
    double last_step = 0.0;
    for (int k = 0; k < POLISH_STEPS; ++k) {
      cost_gradient(p, w);
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl)
          for (int sd = 0; sd < 2; ++sd) {
            const int h = q->held[i][sl][sd];
            q->res[i][sl][sd] = h ? row_res(p, i, sl, sd) : 0.0;
            w->cf[i][sl][sd] = h ? q->y[i][sl][sd] + POLISH_THETA * q->res[i][sl][sd] : 0.0;
          }
      for (int j = 0; j < S; ++j) {
        q->resl[j] = q->heldl[j] ? -p->lmb[j] : 0.0;
        w->cfl[j] = q->heldl[j] ? q->yl[j] + POLISH_THETA * q->resl[j] : 0.0;
      }
      newton_solve(p, w);
      primal_update(p, 1.0);
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl) {
          const double dval = slot_val(p->dz, p->dv, i, sl);
          const double dsg = (sl == SL_EY && p->has_sigma) ? p->dsigma : 0.0;
          for (int sd = 0; sd < 2; ++sd)
            if (q->held[i][sl][sd]) q->y[i][sl][sd] += POLISH_THETA * (q->res[i][sl][sd] + ((sd == 0 ? dval : -dval) - dsg));
        }
      for (int j = 0; j < S; ++j)
        if (q->heldl[j]) q->yl[j] += POLISH_THETA * (q->resl[j] - p->dlmb[j]);
      last_step = 0.0;
      for (int i = 0; i < N; ++i) {
        static const double isx[8] = {1.0 / 2000, 1.0 / 10, 1.0 / 0.1, 1.0 / 80, 1.0 / 2, 1.0 / 2, 1.0 / 10, 1.0 / 0.3};
        if (i >= 1)
          for (int r = 0; r < 8; ++r) last_step = fmax(last_step, fabs(p->dz[i][r]) * isx[r]);
        if (i < N - 1)
          for (int r = 0; r < 2; ++r) last_step = fmax(last_step, fabs(p->dv[i][r]) * isx[6 + r]);
      }
      if (getenv("LMPC_ORACLE_POLISH_TRACE")) fprintf(stderr, "      step %d: %.3e\n", k, last_step);
      if (k >= 1 && last_step <= POLISH_STEP_OK) break; /* (converged: the remaining steps would move nothing) */
    }
    /* ---- verify ---- */
    int bad = !(last_step <= POLISH_STEP_TOL), anyneg = 0, anyweak = 0, anyviol = 0; /* (a last step that still moved the
                                                                                       * iterate: the steps have not converged) */
    double ymin = 0.0, comp = 0.0;
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd) {
          if (!p->act[i][sl][sd]) continue;
          const double r = row_res(p, i, sl, sd);
          if (q->held[i][sl][sd]) {
            const double y = q->y[i][sl][sd];
            if (!(fabs(r) <= POLISH_FEAS)) bad = 1; /* (also a NaN) */
            if (y < -POLISH_DUAL) {
              anyneg = 1;
              if (p->lam[i][sl][sd] < POLISH_STRONG * p->t[i][sl][sd]) anyweak = 1;
            }
            if (y < ymin) ymin = y;
            comp += fabs(y * r);
          } else if (!(r <= POLISH_FEAS)) {
            anyviol = 1;
          }
        }
    for (int j = 0; j < S; ++j) {
      if (q->heldl[j]) {
        if (!(fabs(p->lmb[j]) <= POLISH_FEAS)) bad = 1;
        if (q->yl[j] < -POLISH_DUAL_L) {
          anyneg = 1;
          if (p->ll[j] < POLISH_STRONG * p->tl[j]) anyweak = 1;
        }
        if (q->yl[j] < ymin) ymin = q->yl[j];
        comp += fabs(q->yl[j] * p->lmb[j]);
      } else if (!(-p->lmb[j] <= POLISH_FEAS)) {
        anyviol = 1;
      }
    }
    if (getenv("LMPC_ORACLE_POLISH_TRACE")) {
      int nheld = 0, nbadfeas = 0, nneg = 0, nviol = 0; double worst_feas = 0.0, worst_viol = 0.0;
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl)
          for (int sd = 0; sd < 2; ++sd) {
            if (!p->act[i][sl][sd]) continue;
            const double r = row_res(p, i, sl, sd);
            if (q->held[i][sl][sd]) { ++nheld; if (!(fabs(r) <= POLISH_FEAS)) { ++nbadfeas; fprintf(stderr, "   held row (%d,%d,%d) res %.3e y %.3e ipm lam %.3e t %.3e\n", i, sl, sd, r, q->y[i][sl][sd], p->lam[i][sl][sd], p->t[i][sl][sd]); } if (fabs(r) > worst_feas) worst_feas = fabs(r); if (q->y[i][sl][sd] < -POLISH_DUAL) { ++nneg; fprintf(stderr, "   neg mult (%d,%d,%d) y %.3e ipm lam %.3e t %.3e\n", i, sl, sd, q->y[i][sl][sd], p->lam[i][sl][sd], p->t[i][sl][sd]); } }
            else if (!(r <= POLISH_FEAS)) { ++nviol; if (r > worst_viol) worst_viol = r; fprintf(stderr, "   violated (%d,%d,%d) res %.3e ipm lam %.3e t %.3e\n", i, sl, sd, r, p->lam[i][sl][sd], p->t[i][sl][sd]); }
          }
      fprintf(stderr, "polish round %d: held %d last_step %.3e bad %d (held rows off %d, worst %.3e) neg %d (weak %d, ymin %.3e) viol %d (worst %.3e)\n",
              round, nheld, last_step, bad, nbadfeas, worst_feas, nneg, anyweak, ymin, nviol, worst_viol);
    }
    if (!bad && !anyneg && !anyviol) {
      ok = 1;
      *mu_out = comp / m_rows;
      break;
    }
    q->noise = bad && !anyneg && !anyviol; /* a consistent held set whose multiplier steps did not converge (see ipm_solve's exit) */
    /* ---- repair the working set ---- */
    int changed = 0;
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd) {
          if (!p->act[i][sl][sd]) continue;
          if (q->held[i][sl][sd]) {
            const double y = q->y[i][sl][sd];
            const int weak = p->lam[i][sl][sd] < POLISH_STRONG * p->t[i][sl][sd];
            if (y < -POLISH_DUAL && (anyweak ? weak : y <= 0.5 * ymin)) {
              q->held[i][sl][sd] = 0;
              changed = 1;
            }
          } else if (!anyneg && !(row_res(p, i, sl, sd) <= POLISH_FEAS)) {
            q->held[i][sl][sd] = 1;
            changed = 1;
          }
        }
    for (int j = 0; j < S; ++j) {
      if (q->heldl[j]) {
        const int weak = p->ll[j] < POLISH_STRONG * p->tl[j];
        if (q->yl[j] < -POLISH_DUAL_L && (anyweak ? weak : q->yl[j] <= 0.5 * ymin)) {
          q->heldl[j] = 0;
          changed = 1;
        }
      } else if (!anyneg && !(-p->lmb[j] <= POLISH_FEAS)) {
        q->heldl[j] = 1;
        changed = 1;
      }
    }
    if (!changed) break; /* (the multiplier steps did not converge on a consistent set: nothing to repair) */
  }
  if (!ok) {
    memcpy(p->z, q->z, sizeof(q->z));
    memcpy(p->v, q->v, sizeof(q->v));
    memcpy(p->lmb, q->lmb, sizeof(q->lmb));
    p->sigma = q->sigma;
  }
  return ok;
}

static int polish(prob_t* p, work_t* w, polish_t* q, int m_rows, int* rounds_out, double* mu_out) {
  return polish_rounds(p, w, q, m_rows, rounds_out, mu_out, POLISH_ROUNDS | POLISH_EXIT);
}

/* Solve the QP with a Mehrotra predictor-corrector interior-point method.  The iteration
 * stops when the average complementarity mu <= tol (default 3e-14) and every row residual is
 * below 1e-9.  Accuracy (DESIGN.md "numerics"): the cost-to-go is kept exactly symmetric and the last
 * iterations (mu <= JOSEPH_MU) factorise in the stabilised form, so the Newton directions stay accurate
 * down to mu ~ 1e-14; against the dense optimum the returned point is within 1e-6 (scaled) wherever strict
 * complementarity holds with a margin >= 1e-4 (oracle/qp.py strict_complementarity) and within ~1e-5 on
 * degenerate problems, where any interior point is O(sqrt(mu)) away.                                 */
static int g_warm_rounds = 0; /* lmpc_set_warm_rounds (0: WARM_ROUNDS); a process-wide setting of this test library */
void lmpc_oracle_set_warm_rounds(int rounds) { g_warm_rounds = rounds > POLISH_ROUNDS ? POLISH_ROUNDS : rounds; } /* (1 .. 4 as lmpc_set_warm_rounds) */

static int ipm_solve(prob_t* p, work_t* w, polish_t* pq, int* iters_out, double* kkt_out) {
  const int N = p->N, S = p->S;
#ifndef IPM_TAU /* (scratch/r5/twin_variant.py builds variants with -D) */
#define IPM_TAU 0.995
#define IPM_MU0 0.1
#define IPM_THR 0.5
#endif
  const double tau = IPM_TAU, mu0 = IPM_MU0, thr_frac = IPM_THR;
  int warm_spent = 0;
  memset(w, 0, sizeof(work_t)); /* (the caller owns the allocation: one per range, not one mmap per solve) */
  /* ---- initial point: the minimiser of the cost over the dynamics alone (no inequality
   * rows, sigma = 0).  The linearised model can be open-loop unstable (|eig A| > 1 at low speed
   * with dt = 25 ms), so the trajectory is first rolled out under the stabilising Riccati
   * feedback v = -K z (all row weights zero), then one Newton step from that dynamics-feasible
   * point lands on the minimiser exactly (the cost is quadratic).                              */
  memset(w->th, 0, sizeof(w->th));
  memset(w->cf, 0, sizeof(w->cf));
  for (int j = 0; j < S; ++j) {
    p->lmb[j] = 1.0 / S;
    w->thl[j] = 1.0;
    w->cfl[j] = 0.0;
  }
  p->sigma = 0.0;
  w->frozen_lambda = 1;
  newton_factor(p, w, 0);
  /* ---- warm start (lmpc_solve_batch_warm; the reference's X_optm_ref / U_optm_ref / dU_optm_ref, racing_mpc.cpp:293-305: in the
   * node the reference IS the previous solution shifted, racing_mpc_node.cpp:245-254).  An active-set solve before any interior
   * point: (1) the plan is made dynamically exact about the new linearisation by rolling it out under the Riccati feedback of
   * the factorisation above, v_i = v_i^plan - K_i (z_i - z_i^plan) (the shifted plan misses the new dynamics by the
   * linearisation's change and x_0 by the plant's step: a few 1e-3); (2) the working set is read off the PLAN -- a polished
   * optimum sits on its active bounds to rounding, and the boxes do not move with the shift; the boundary rows' bounds do, by
   * the track's change over one knot, so they are taken with a tolerance --; (3) the polish solves on that set, verifies the KKT
   * conditions of THIS problem and repairs the set, WARM_ROUNDS times at most.  Accepted: the optimum, for the price of about two
   * iterations.  Refused: the cold start below, as if nothing had happened (the attempt has cost about two more). */
  /* The learning problem (round 6) takes the same route with one more piece of the plan: the simplex weights of its terminal point,
   * the reference's convex_combi_optm_ref (set_initial(convex_combi_, ...), racing_mpc.cpp:281).  Their support -- the safe-set
   * points that carried weight -- is the working set of the simplex rows: free where the plan's weight is positive, held at zero
   * elsewhere (a polished plan's held weights are zero to rounding); the weights themselves start the multiplier steps.  Without
   * them there is no working set to start from (all S weights free is more than the terminal block keeps explicit) and the solve is
   * cold. */
  int lam_ok = !S;
  if (S && p->has_lamw) {
    double sum = 0.0;
    for (int j = 0; j < S; ++j) sum += p->lamw[j] > 0.0 ? p->lamw[j] : 0.0;
    lam_ok = sum > 0.0;
  }
  if (p->warm && pq && lam_ok) {
    for (int i = 0; i < N - 1; ++i) {
      for (int a = 0; a < 2; ++a) {
        double acc = p->vw[i][a];
        for (int c = 0; c < 8; ++c) acc -= p->K[i][a * 8 + c] * (p->z[i][c] - p->zw[i][c]);
        p->v[i][a] = acc;
      }
      const double u0 = p->z[i][6] + p->dt[i] * p->v[i][0], u1 = p->z[i][7] + p->dt[i] * p->v[i][1];
      for (int r = 0; r < 6; ++r) {
        double acc = p->g[i][r] + p->B[i][r * 2] * u0 + p->B[i][r * 2 + 1] * u1;
        for (int c = 0; c < 6; ++c) acc += p->A[i][r * 6 + c] * p->z[i][c];
        p->z[i + 1][r] = acc;
      }
      p->z[i + 1][6] = u0;
      p->z[i + 1][7] = u1;
    }
    /* the plan's boundary slack, then its working set */
    double sgw = 0.0;
    if (p->has_sigma)
      for (int i = 0; i < N; ++i) {
        if (p->act[i][SL_EY][0] && p->zw[i][1] - p->hi[i][SL_EY] > sgw) sgw = p->zw[i][1] - p->hi[i][SL_EY];
        if (p->act[i][SL_EY][1] && p->lo[i][SL_EY] - p->zw[i][1] > sgw) sgw = p->lo[i][SL_EY] - p->zw[i][1];
      }
    p->sigma = sgw;
    int mw = 0;
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl) {
        const double val = slot_val(p->zw, p->vw, i, sl), sg = (sl == SL_EY && p->has_sigma) ? sgw : 0.0;
        const double tol = sl == SL_EY ? WARM_ACT_EY : WARM_ACT;
        for (int sd = 0; sd < 2; ++sd) {
          if (!p->act[i][sl][sd]) continue;
          const double slack = sd == 0 ? (p->hi[i][sl] + sg - val) : (val + sg - p->lo[i][sl]);
          const int held = slack <= tol;
          p->t[i][sl][sd] = held ? 0.0 : (slack > 0.0 ? slack : 0.0);
          p->lam[i][sl][sd] = held ? 1.0 : 0.0;
          ++mw;
        }
      }
    if (S) {
      double sum = 0.0;
      for (int j = 0; j < S; ++j) sum += p->lamw[j] > 0.0 ? p->lamw[j] : 0.0;
      for (int j = 0; j < S; ++j) {
        const double lm = (p->lamw[j] > 0.0 ? p->lamw[j] : 0.0) / sum;
        const int fr = lm > WARM_ACT;
        p->lmb[j] = lm;
        p->tl[j] = fr ? lm : 0.0; /* (the polish classifies by ll > tl: held) */
        p->ll[j] = fr ? 0.0 : 1.0;
        ++mw;
      }
    }
    w->frozen_lambda = 0;
    int wr = 0;
    double wmu = 0.0;
    if (polish_rounds(p, w, pq, mw, &wr, &wmu, g_warm_rounds > 0 ? g_warm_rounds : WARM_ROUNDS)) {
      *iters_out = wr;
      if (kkt_out) {
        kkt_out[0] = 0.0;
        kkt_out[1] = 0.0;
        kkt_out[2] = wmu;
        kkt_out[3] = p->sigma;
      }
      return LMPC_SOLVE_OPTIMAL;
    }
    warm_spent = wr; /* the rounds a refused attempt took are counted with the solve's iterations */
    /* refused: everything the attempt touched is set up again by the cold start */
    memset(w->th, 0, sizeof(w->th));
    memset(w->cf, 0, sizeof(w->cf));
    memset(p->t, 0, sizeof(p->t));
    memset(p->lam, 0, sizeof(p->lam));
    for (int i = 1; i < N; ++i) memset(p->z[i], 0, sizeof(p->z[i]));
    memset(p->v, 0, sizeof(p->v));
    p->sigma = 0.0;
    for (int j = 0; j < S; ++j) {
      p->lmb[j] = 1.0 / S;
      p->tl[j] = p->ll[j] = 0.0;
      w->thl[j] = 1.0;
      w->cfl[j] = 0.0;
    }
    w->frozen_lambda = 1;
    newton_factor(p, w, 0);
  }
  for (int i = 0; i < N - 1; ++i) {
    for (int a = 0; a < 2; ++a) {
      double acc = 0.0;
      for (int c = 0; c < 8; ++c) acc -= p->K[i][a * 8 + c] * p->z[i][c];
      p->v[i][a] = acc;
    }
    const double u0 = p->z[i][6] + p->dt[i] * p->v[i][0], u1 = p->z[i][7] + p->dt[i] * p->v[i][1];
    for (int r = 0; r < 6; ++r) {
      double acc = p->g[i][r] + p->B[i][r * 2] * u0 + p->B[i][r * 2 + 1] * u1;
      for (int c = 0; c < 6; ++c) acc += p->A[i][r * 6 + c] * p->z[i][c];
      p->z[i + 1][r] = acc;
    }
    p->z[i + 1][6] = u0;
    p->z[i + 1][7] = u1;
  }
  cost_gradient(p, w);
  newton_solve(p, w);
  p->dsigma = 0.0;
  primal_update(p, 1.0);
  w->frozen_lambda = 0;
  int m = 0;
  for (int i = 0; i < N; ++i)
    for (int sl = 0; sl < NSLOT; ++sl) {
      const double hi = p->hi[i][sl], lo = p->lo[i][sl];
      double range = (isfinite(hi) && isfinite(lo)) ? (hi - lo) : 1.0;
      if (!(range > 1e-3)) range = 1e-3;
      const double thr = thr_frac * range;
      for (int sd = 0; sd < 2; ++sd) {
        if (!p->act[i][sl][sd]) continue;
        const double r = -row_res(p, i, sl, sd);
        const double t = r > thr ? r : thr;
        p->t[i][sl][sd] = t;
        p->lam[i][sl][sd] = mu0 / t;
        ++m;
      }
    }
  for (int j = 0; j < S; ++j) {
    p->tl[j] = p->lmb[j];
    p->ll[j] = mu0 / p->tl[j];
    ++m;
  }
  int status = LMPC_SOLVE_MAX_ITER, it = 0;
  double mu = 0.0, rdmax = 0.0, rd_check = 0.0;

  /* ================= phase 1: interior point ================= */
  int distress = 0; /* the complementarity has gone up once: the wide-neighbourhood rule applies from then on */
  int pol_tried = 0, pol_done = 0, pol_rounds = 0; /* the early polish attempt; an accepted polish; rounds spent */
  int stall_moving = 0;
  double mu_prev = INFINITY;
  for (it = 0; it <= p->max_iter; ++it) {
    double musum = 0.0;
    rdmax = 0.0;
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd) {
          if (!p->act[i][sl][sd]) continue;
          const double r = row_res(p, i, sl, sd) + p->t[i][sl][sd];
          w->rd[i][sl][sd] = r;
          if (fabs(r) > rdmax) rdmax = fabs(r);
          musum += p->t[i][sl][sd] * p->lam[i][sl][sd];
          w->th[i][sl][sd] = p->lam[i][sl][sd] / p->t[i][sl][sd];
        }
    for (int j = 0; j < S; ++j) {
      w->rdl[j] = -p->lmb[j] + p->tl[j];
      if (fabs(w->rdl[j]) > rdmax) rdmax = fabs(w->rdl[j]);
      musum += p->tl[j] * p->ll[j];
      w->thl[j] = p->ll[j] / p->tl[j];
    }
    mu = musum / m;
    /* (see the step-length rule; far from feasibility mu may rise legitimately) */
    if (it >= 1 && mu >= mu_prev && rdmax <= 1e-6 && N <= 40) distress = 1; /* (N <= 40: the kernel's instantiations
                                                                              * for longer horizons do without the rule) */
    mu_prev = mu;
    if (!(mu == mu) || !(rdmax == rdmax)) {
      status = LMPC_SOLVE_INFEASIBLE;
      break;
    }
    if (mu <= p->tol && rdmax <= 1e-9) {
      status = LMPC_SOLVE_OPTIMAL;
      break;
    }
    /* primal infeasibility: the row residual contracts by (1 - alpha) per iteration on a feasible problem; if it
     * has not lost a tenth over five iterations while still large (step lengths stuck below ~2 %), give up.  (A
     * tighter test -- "not halved" -- rejects feasible problems with a slow start: IAC at 60 m/s into a corner
     * needs 20-30 iterations and contracts by 0.6-0.8 per five early on.) */
    if (it % 5 == 0) {
      if (it >= 10 && rdmax > 1e-6 && rdmax > 0.9 * rd_check) {
        status = LMPC_SOLVE_INFEASIBLE;
        break;
      }
      rd_check = rdmax;
    }
    if (it == p->max_iter) break;
    if (pq && !pol_tried && mu <= POLISH_MU && rdmax <= POLISH_RD) { /* the active set is usually settled by now */
      pol_tried = 1;
#ifndef EARLY_ROUNDS /* (scratch/r5/twin_variant.py) */
#define EARLY_ROUNDS POLISH_ROUNDS
#endif
      if (polish_rounds(p, w, pq, m, &pol_rounds, &mu, EARLY_ROUNDS)) {
        status = LMPC_SOLVE_OPTIMAL;
        pol_done = 1;
        break;
      }
      /* refused: the iterate is untouched; the weights the polish overwrote are the iteration's own again */
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl)
          for (int sd = 0; sd < 2; ++sd)
            if (p->act[i][sl][sd]) w->th[i][sl][sd] = p->lam[i][sl][sd] / p->t[i][sl][sd];
      for (int j = 0; j < S; ++j) w->thl[j] = p->ll[j] / p->tl[j];
    }
    cost_gradient(p, w);
    newton_factor(p, w, mu <= JOSEPH_MU);
    double sigc = 0.0, alpha = 1.0;
    int numerics_failed = 0, stalled = 0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl)
          for (int sd = 0; sd < 2; ++sd) {
            if (!p->act[i][sl][sd]) continue;
            double c = w->th[i][sl][sd] * w->rd[i][sl][sd];
            if (pass == 1) c += (sigc * mu - p->dtt[i][sl][sd] * p->dlam[i][sl][sd]) / p->t[i][sl][sd];
            w->cf[i][sl][sd] = c;
          }
      for (int j = 0; j < S; ++j) {
        w->cfl[j] = w->thl[j] * w->rdl[j];
        if (pass == 1) w->cfl[j] += (sigc * mu - p->dtl[j] * p->dll[j]) / p->tl[j];
      }
      newton_solve(p, w);
      /* a Newton step that is not a number (complete cancellation in the Schur complement of sigma or in H): keep
       * the iterate and report it by what it has reached -- same rule as the kernel */
      {
        int finite = isfinite(p->dsigma);
        for (int i = 0; i < N && finite; ++i)
          for (int sl = 0; sl < NSLOT; ++sl)
            if (!isfinite(slot_val(p->dz, p->dv, i, sl))) finite = 0;
        if (!finite) {
          numerics_failed = 1;
          break;
        }
      }
      /* row steps and the largest step keeping t, lam > 0 */
      double amax = 1.0;
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl) {
          const double dval = slot_val(p->dz, p->dv, i, sl);
          const double dsg = (sl == SL_EY && p->has_sigma) ? p->dsigma : 0.0;
          for (int sd = 0; sd < 2; ++sd) {
            if (!p->act[i][sl][sd]) continue;
            const double cdy = (sd == 0 ? dval : -dval) - dsg;
            const double t = p->t[i][sl][sd], lam = p->lam[i][sl][sd], th = w->th[i][sl][sd];
            const double dt_ = -w->rd[i][sl][sd] - cdy;
            const double dl_ = -lam + w->cf[i][sl][sd] - th * w->rd[i][sl][sd] - th * dt_;
            p->dtt[i][sl][sd] = dt_;
            p->dlam[i][sl][sd] = dl_;
            if (dt_ < 0 && -t / dt_ < amax) amax = -t / dt_;
            if (dl_ < 0 && -lam / dl_ < amax) amax = -lam / dl_;
          }
        }
      for (int j = 0; j < S; ++j) {
        p->dtl[j] = -w->rdl[j] + p->dlmb[j];
        p->dll[j] = -p->ll[j] + w->cfl[j] - w->thl[j] * w->rdl[j] - w->thl[j] * p->dtl[j];
        if (p->dtl[j] < 0 && -p->tl[j] / p->dtl[j] < amax) amax = -p->tl[j] / p->dtl[j];
        if (p->dll[j] < 0 && -p->ll[j] / p->dll[j] < amax) amax = -p->ll[j] / p->dll[j];
      }
      if (pass == 0) {
        double s = 0.0;
        for (int i = 0; i < N; ++i)
          for (int sl = 0; sl < NSLOT; ++sl)
            for (int sd = 0; sd < 2; ++sd)
              if (p->act[i][sl][sd])
                s += (p->t[i][sl][sd] + amax * p->dtt[i][sl][sd]) * (p->lam[i][sl][sd] + amax * p->dlam[i][sl][sd]);
        for (int j = 0; j < S; ++j) s += (p->tl[j] + amax * p->dtl[j]) * (p->ll[j] + amax * p->dll[j]);
        const double ratio = (s / m) / mu;
        sigc = ratio * ratio * ratio;
      } else {
        alpha = tau * amax;
        if (alpha > 1.0) alpha = 1.0;
        /* For a problem whose mu has risen once (distress): cut the step back until no complementarity product falls
         * below NBHD_GAMMA times their mean (the wide neighbourhood of the central path).  Without it Mehrotra's iteration can leave the neighbourhood and cycle:
         * seen on a learning problem whose safe set offers two nearly exchangeable points (products at 0.01 and 300
         * times mu, mu bouncing between 6e-6 and 2e-5 up to the iteration cap while the dense solver finds the
         * optimum; 19 iterations with the rule).  1e-3 does not stop that cycle.  Problems whose mu falls monotonically
         * never take the cut.  The same rule as the kernel. */
        double s = 0.0;
        for (int trial = 0;; ++trial) {
          double pmin = INFINITY;
          s = 0.0;
          for (int i = 0; i < N; ++i)
            for (int sl = 0; sl < NSLOT; ++sl)
              for (int sd = 0; sd < 2; ++sd)
                if (p->act[i][sl][sd]) {
                  const double pr = (p->t[i][sl][sd] + alpha * p->dtt[i][sl][sd]) * (p->lam[i][sl][sd] + alpha * p->dlam[i][sl][sd]);
                  s += pr;
                  if (pr < pmin) pmin = pr;
                }
          for (int j = 0; j < S; ++j) {
            const double pr = (p->tl[j] + alpha * p->dtl[j]) * (p->ll[j] + alpha * p->dll[j]);
            s += pr;
            if (pr < pmin) pmin = pr;
          }
          if (!distress || trial == NBHD_TRIALS || pmin >= NBHD_GAMMA * s / m) break;
          alpha *= 0.6;
        }
        /* no further progress: with the rows feasible and the complementarity already small, a corrector step that
         * would not lower it (the Newton direction has reached the accuracy of the factorisation; seen on learning
         * problems whose speed rides its bound over most of the horizon) ends the solve at the current iterate
         * instead of letting mu wander upwards until the iteration cap */
        if (rdmax <= 1e-9 && mu <= STALL_MU && s / m >= mu) stalled = 1;
      }
    }
    if (stalled) { /* the primal iterate is kept; slacks and multipliers have taken the corrector step (as in the kernel, which
                    * updates them before it knows): they are what the polish classifies the rows by */
      for (int i = 0; i < N; ++i)
        for (int sl = 0; sl < NSLOT; ++sl)
          for (int sd = 0; sd < 2; ++sd)
            if (p->act[i][sl][sd]) {
              p->t[i][sl][sd] += alpha * p->dtt[i][sl][sd];
              p->lam[i][sl][sd] += alpha * p->dlam[i][sl][sd];
            }
      for (int j = 0; j < S; ++j) { /* (the kernel's simplex rows carry lambda with them: it moves too) */
        p->tl[j] += alpha * p->dtl[j];
        p->ll[j] += alpha * p->dll[j];
        p->lmb[j] += alpha * p->dlmb[j];
      }
      status = LMPC_SOLVE_OPTIMAL;
      stall_moving = scaled_step(p, alpha) > STALL_STEP; /* (round 6, below: the step the stall declines would still move the iterate) */
      break;
    }
    if (numerics_failed) {
      status = (mu <= 10.0 * p->tol && rdmax <= 1e-9) ? LMPC_SOLVE_OPTIMAL : LMPC_SOLVE_MAX_ITER;
      break;
    }
    primal_update(p, alpha);
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd)
          if (p->act[i][sl][sd]) {
            p->t[i][sl][sd] += alpha * p->dtt[i][sl][sd];
            p->lam[i][sl][sd] += alpha * p->dlam[i][sl][sd];
          }
    for (int j = 0; j < S; ++j) {
      p->tl[j] += alpha * p->dtl[j];
      p->ll[j] += alpha * p->dll[j];
    }
  }

  /* the interior point has converged (or stopped at its floor with status OPTIMAL): polish what it reached */
  if (pq && status == LMPC_SOLVE_OPTIMAL && !pol_done) {
    pol_done = polish(p, w, pq, m, &pol_rounds, &mu);
    /* refused, by a consistent held set whose multiplier steps did not settle: the sweeps are noisier than the contract on
     * this problem, and the interior point's iterate came from the same sweeps (csrc/lmpc_solve_kernel.hip, same place) */
    if (!pol_done && pq->noise) status = LMPC_SOLVE_MAX_ITER;
  }
  /* Round 6: a STALL is not convergence when the Newton step it declines would still move the iterate.  The stall rule keeps the
   * current point once the corrector would no longer lower mu (<= 1e-9, rows feasible) -- the direction has reached the accuracy of
   * the factorisation -- and reported it OPTIMAL whatever the polish then said.  On one problem of tests/dispatch_sweep.py at 4096
   * problems per case (learning, 160 points, N = 57) the iterate was still moving by 7.5e-4 (scaled) per step, the polish refused,
   * and the point returned was 3.4e-4 from the optimum with status OPTIMAL.  A stalled iterate whose declined step exceeds
   * STALL_STEP and that no polish has verified is returned as it is, with LMPC_SOLVE_MAX_ITER: stopped short of the stated
   * accuracy (the status the noise rule above uses). */
  if (status == LMPC_SOLVE_OPTIMAL && !pol_done && stall_moving) status = LMPC_SOLVE_MAX_ITER;
  /* ---- exit: report the reduced-gradient stationarity for the multipliers reached ---- */
  double rg = 0.0, viol = rdmax;
  if (status != LMPC_SOLVE_INFEASIBLE) {
    cost_gradient(p, w);
    rg = pol_done ? stationarity(p, w, pq->y, pq->yl) : stationarity(p, w, p->lam, p->ll);
  }
  if (pol_done) {
    viol = 0.0;
    for (int i = 0; i < N; ++i)
      for (int sl = 0; sl < NSLOT; ++sl)
        for (int sd = 0; sd < 2; ++sd)
          if (p->act[i][sl][sd]) {
            const double r = row_res(p, i, sl, sd);
            if (r > viol) viol = r;
          }
  }
  if (kkt_out) {
    kkt_out[0] = rg;
    kkt_out[1] = viol;
    kkt_out[2] = mu;
    kkt_out[3] = p->sigma;
  }
  /* hard hull equality (penalty limit): a terminal state the hull cannot reach leaves a residual the weight does not close */
  if (p->S && p->hard_hull && status == LMPC_SOLVE_OPTIMAL) {
    static const double isc[6] = {1.0 / 2000.0, 1.0 / 10.0, 1.0 / 0.1, 1.0 / 80.0, 1.0 / 2.0, 1.0 / 2.0};
    /* (eps_T is that of the final point: cost_gradient above) */
    for (int k = 0; k < 6; ++k)
      if (fabs(w->eps_T[k]) * isc[k] > LMPC_HARD_HULL_RESIDUAL) status = LMPC_SOLVE_INFEASIBLE;
  }
  *iters_out = it + pol_rounds + warm_spent; /* a polish round costs about what an iteration does and is counted as one */
  return status;
}

/* ------------------------------------------------------------------------------------------ */
/* problem set-up from the C-ABI style inputs (batch axis fastest)                             */
/* ------------------------------------------------------------------------------------------ */

/* the caller's plan in the solver's variables (used by the warm start only): z_i = [x_i; u_{i-1}], v_i = (u_i - u_{i-1}) / t_i.
   lmpc_solve_batch_warm's (X_plan, U_plan); when the plan is the linearisation trajectory itself, (X_ref, U_ref). */
static void set_plan(prob_t* p, int B, int b, const double* X_plan, const double* U_plan) {
  const int N = p->N;
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 6; ++k) p->zw[i][k] = i == 0 ? p->z[0][k] : X_plan[(size_t)(k * N + i) * B + b];
    for (int k = 0; k < 2; ++k) p->zw[i][6 + k] = i == 0 ? p->z[0][6 + k] : U_plan[(size_t)(k * (N - 1) + i - 1) * B + b];
  }
  for (int i = 0; i < N - 1; ++i)
    for (int k = 0; k < 2; ++k) p->vw[i][k] = (U_plan[(size_t)(k * (N - 1) + i) * B + b] - p->zw[i][6 + k]) / p->dt[i];
}

static void setup_problem(prob_t* p, const lmpc_config* cfg, const lmpc_vehicle* veh, int B, int b,
                          const double* x_ic, const double* u_ic, const double* X_ref,
                          const double* U_ref, const double* T_ref, const double* bl,
                          const double* br, const double* curv, const double* vref,
                          const double* ss_x, const double* ss_j) {
  const int N = cfg->N;
  memset(p, 0, sizeof(*p));
  p->N = N;
  p->has_sigma = cfg->q_boundary > 0.0;
  p->S = cfg->learning ? cfg->num_ss_pts : 0;
  p->max_iter = cfg->max_iter > 0 ? cfg->max_iter : 60; /* (as lmpc_create) */
  p->tol = cfg->tol > 0 ? cfg->tol : 3e-14;
  for (int i = 0; i < N - 1; ++i) {
    double x[6], u[2], xp[6];
    for (int k = 0; k < 6; ++k) x[k] = X_ref[(size_t)(k * N + i) * B + b];
    for (int k = 0; k < 2; ++k) u[k] = U_ref[(size_t)(k * (N - 1) + i) * B + b];
    p->dt[i] = T_ref[(size_t)i * B + b];
    lmpc_oracle_linearize(veh, x, u, curv[(size_t)i * B + b], p->dt[i], p->A[i], p->B[i], p->g[i], xp);
  }
  for (int k = 0; k < 6; ++k) p->z[0][k] = x_ic[(size_t)k * B + b];
  for (int k = 0; k < 2; ++k) p->z[0][6 + k] = u_ic[(size_t)k * B + b];
  /* cost */
  const double qd[6] = {0.0, cfg->q_contour, cfg->q_heading, cfg->q_vel, cfg->q_vy, cfg->q_vyaw};
  for (int i = 0; i < N; ++i)
    for (int k = 0; k < 6; ++k) {
      p->Qx[i][k] = 0.0;
      p->qx[i][k] = 0.0;
    }
  if (!cfg->learning) {
    for (int i = 0; i < N - 1; ++i) {
      for (int k = 0; k < 6; ++k) p->Qx[i][k] = 2.0 * qd[k];
      p->qx[i][3] = -2.0 * cfg->q_vel * vref[(size_t)i * B + b];
    }
    p->Qx[N - 1][1] = 20.0 * cfg->q_contour;
    p->Qx[N - 1][2] = 20.0 * cfg->q_heading;
    p->Qx[N - 1][3] = 20.0 * cfg->q_vel;
    p->qx[N - 1][3] = -20.0 * cfg->q_vel * vref[(size_t)(N - 1) * B + b];
  }
  for (int a = 0; a < 2; ++a)
    for (int c = 0; c < 2; ++c) {
      p->Qu[a * 2 + c] = cfg->R[a * 2 + c] + cfg->R[c * 2 + a];
      p->Sv[a * 2 + c] = cfg->R_d[a * 2 + c] + cfg->R_d[c * 2 + a];
    }
  p->qsig = 2.0 * cfg->q_boundary;
  if (p->S) {
    int any_slack = 0;
    for (int k = 0; k < 6; ++k) any_slack |= cfg->convex_hull_slack[k] > 0.0;
    p->hard_hull = !any_slack; /* racing_mpc.cpp:500-502: x_T = SS lambda, no slack variable */
    for (int k = 0; k < 6; ++k) {
      p->chs2[k] = any_slack ? 2.0 * cfg->convex_hull_slack[k] : 2.0 * LMPC_HARD_HULL_WEIGHT;
      p->ss0[k] = ss_x[(size_t)(k * p->S) * B + b];
    }
    /* A point that repeats the one before it adds nothing to the hull and makes the free weights' system singular (C_A, the
     * polish's multiplier steps): the padding of a set with fewer than S points repeats the last point S - n_found times
     * (racing_mpc.cpp:263-272).  Runs of identical points are kept once -- X, U, dU are those of the full set, the weight of a
     * run sits on its first point.  The kernel does the same (csrc/lmpc_solve_kernel.hip, the learning prologue). */
    const int S_in = p->S;
    int kept = 0;
    for (int j = 0; j < S_in; ++j) {
      int dup = j > 0 && ss_j[(size_t)j * B + b] == ss_j[(size_t)(j - 1) * B + b];
      for (int k = 0; k < 6 && dup; ++k) dup = ss_x[(size_t)(k * S_in + j) * B + b] == ss_x[(size_t)(k * S_in + j - 1) * B + b];
      if (dup) continue;
      for (int k = 0; k < 6; ++k) p->ssx[k][kept] = ss_x[(size_t)(k * S_in + j) * B + b] - p->ss0[k];
      p->ssj[kept] = ss_j[(size_t)j * B + b];
      p->ss_map[kept++] = j;
    }
    p->S_in = S_in;
    p->S = kept;
    double umax = 0.0; /* largest u_j' E u_j over the (centred) points */
    for (int j = 0; j < p->S; ++j) {
      double q = 0.0;
      for (int k = 0; k < 6; ++k) q += p->chs2[k] * p->ssx[k][j] * p->ssx[k][j];
      if (q > umax) umax = q;
    }
    p->tau = TAU_REL * umax;
  }
  /* bounds */
  const double u_lo[2] = {fmax(cfg->u_min[0], veh->Fb_max / 1000.0), fmax(cfg->u_min[1], -veh->max_steer)};
  const double u_hi[2] = {fmin(cfg->u_max[0], veh->Fd_max / 1000.0), fmin(cfg->u_max[1], veh->max_steer)};
  const double v_lo[2] = {veh->Fb_max / 1000.0 / veh->Tb, -veh->max_steer_rate};
  const double v_hi[2] = {veh->Fd_max / 1000.0 / veh->Td, veh->max_steer_rate};
  const double marg = cfg->margin + veh->b / 2.0;
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 6; ++k) {
      p->hi[i][k] = cfg->x_max[k];
      p->lo[i][k] = cfg->x_min[k];
      const int on = (i >= 1 && i <= N - 2);
      p->act[i][k][0] = on && isfinite(cfg->x_max[k]);
      p->act[i][k][1] = on && isfinite(cfg->x_min[k]);
    }
    for (int k = 0; k < 2; ++k) {
      p->hi[i][SL_U + k] = u_hi[k];
      p->lo[i][SL_U + k] = u_lo[k];
      p->act[i][SL_U + k][0] = p->act[i][SL_U + k][1] = (i >= 1);
      p->hi[i][SL_V + k] = v_hi[k];
      p->lo[i][SL_V + k] = v_lo[k];
      p->act[i][SL_V + k][0] = p->act[i][SL_V + k][1] = (i <= N - 2);
    }
    p->hi[i][SL_EY] = bl[(size_t)i * B + b] - marg;
    p->lo[i][SL_EY] = br[(size_t)i * B + b] + marg;
    p->act[i][SL_EY][0] = p->act[i][SL_EY][1] = (p->has_sigma || i >= 1);
  }
}

/* x_ic inside the state box at knot 0?  (racing_mpc.cpp:147 applies the box to x_0 = x_ic) */
static int knot0_feasible(const prob_t* p, const lmpc_config* cfg) {
  for (int k = 0; k < 6; ++k)
    if (p->z[0][k] > cfg->x_max[k] || p->z[0][k] < cfg->x_min[k]) return 0;
  if (!p->has_sigma && (p->z[0][1] > p->hi[0][SL_EY] || p->z[0][1] < p->lo[0][SL_EY])) return 0;
  for (int k = 0; k < 6; ++k)
    if (!(p->z[0][k] == p->z[0][k])) return 0;
  return 1;
}

static int solve_range_impl(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch, int32_t b0,
                            int32_t b1, const double* x_ic, const double* u_ic, const double* X_ref,
                            const double* U_ref, const double* T_ref, const double* bound_left,
                            const double* bound_right, const double* curvatures,
                            const double* vel_ref, const double* ss_x, const double* ss_j,
                            double* X_optm, double* U_optm, double* dU_optm, double* lambda_out,
                            int32_t* status, int32_t* iters, double* kkt, int warm, const double* X_plan, const double* U_plan,
                            const double* lam_plan);

/* Same signature family as lmpc_solve_batch (host pointers); b0..b1 is the slice solved. */
int lmpc_oracle_solve_range(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch, int32_t b0,
                            int32_t b1, const double* x_ic, const double* u_ic, const double* X_ref,
                            const double* U_ref, const double* T_ref, const double* bound_left,
                            const double* bound_right, const double* curvatures,
                            const double* vel_ref, const double* ss_x, const double* ss_j,
                            double* X_optm, double* U_optm, double* dU_optm, double* lambda_out,
                            int32_t* status, int32_t* iters, double* kkt) {
  return solve_range_impl(cfg, veh, batch, b0, b1, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, ss_x, ss_j,
                          X_optm, U_optm, dU_optm, lambda_out, status, iters, kkt, 0, NULL, NULL, NULL);
}

/* lmpc_solve_batch_warm: (X_ref, U_ref) is the previous optimal plan, shifted (see ipm_solve) */
int lmpc_oracle_solve_range_warm(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch, int32_t b0,
                                 int32_t b1, const double* x_ic, const double* u_ic, const double* X_ref,
                                 const double* U_ref, const double* T_ref, const double* bound_left,
                                 const double* bound_right, const double* curvatures,
                                 const double* vel_ref, const double* ss_x, const double* ss_j,
                                 double* X_optm, double* U_optm, double* dU_optm, double* lambda_out,
                                 int32_t* status, int32_t* iters, double* kkt) {
  return solve_range_impl(cfg, veh, batch, b0, b1, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, ss_x, ss_j,
                          X_optm, U_optm, dU_optm, lambda_out, status, iters, kkt, 1, X_ref, U_ref, NULL);
}

/* lmpc_solve_batch_warm with its own plan arguments: linearised along (X_ref, U_ref), started from (X_plan, U_plan) */
int lmpc_oracle_solve_range_warm_plan(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch, int32_t b0,
                                      int32_t b1, const double* x_ic, const double* u_ic, const double* X_ref,
                                      const double* U_ref, const double* T_ref, const double* bound_left,
                                      const double* bound_right, const double* curvatures,
                                      const double* vel_ref, const double* ss_x, const double* ss_j,
                                      const double* X_plan, const double* U_plan,
                                      double* X_optm, double* U_optm, double* dU_optm, double* lambda_out,
                                      int32_t* status, int32_t* iters, double* kkt) {
  if (!X_plan || !U_plan) return LMPC_ERR_ARGUMENT;
  return solve_range_impl(cfg, veh, batch, b0, b1, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, ss_x, ss_j,
                          X_optm, U_optm, dU_optm, lambda_out, status, iters, kkt, 1, X_plan, U_plan, NULL);
}

/* lmpc_solve_batch_warm_ss: the learning problem's warm start -- the plan AND the simplex weights of its terminal point,
 * lam_plan [S][B] (the reference's convex_combi_optm_ref, racing_mpc.cpp:281), aligned with THIS call's safe-set points */
int lmpc_oracle_solve_range_warm_lam(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch, int32_t b0,
                                     int32_t b1, const double* x_ic, const double* u_ic, const double* X_ref,
                                     const double* U_ref, const double* T_ref, const double* bound_left,
                                     const double* bound_right, const double* curvatures,
                                     const double* vel_ref, const double* ss_x, const double* ss_j,
                                     const double* X_plan, const double* U_plan, const double* lam_plan,
                                     double* X_optm, double* U_optm, double* dU_optm, double* lambda_out,
                                     int32_t* status, int32_t* iters, double* kkt) {
  if (!X_plan || !U_plan) return LMPC_ERR_ARGUMENT;
  return solve_range_impl(cfg, veh, batch, b0, b1, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right, curvatures, vel_ref, ss_x, ss_j,
                          X_optm, U_optm, dU_optm, lambda_out, status, iters, kkt, 1, X_plan, U_plan, lam_plan);
}

static int solve_range_impl(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch, int32_t b0,
                            int32_t b1, const double* x_ic, const double* u_ic, const double* X_ref,
                            const double* U_ref, const double* T_ref, const double* bound_left,
                            const double* bound_right, const double* curvatures,
                            const double* vel_ref, const double* ss_x, const double* ss_j,
                            double* X_optm, double* U_optm, double* dU_optm, double* lambda_out,
                            int32_t* status, int32_t* iters, double* kkt, int warm, const double* X_plan, const double* U_plan,
                            const double* lam_plan) {
  const int N = cfg->N, B = batch;
  if (N < 3 || N > NMAX) return LMPC_ERR_ARGUMENT;
  if (cfg->learning && (cfg->num_ss_pts < 1 || cfg->num_ss_pts > SMAX)) return LMPC_ERR_ARGUMENT;
  prob_t* p = malloc(sizeof(prob_t));
  work_t* w = malloc(sizeof(work_t));
  polish_t* pq = cfg->polish >= 0 ? malloc(sizeof(polish_t)) : NULL; /* lmpc_config.polish: 0 (default) on, < 0 off */
  if (!p || !w || (cfg->polish >= 0 && !pq)) {
    free(p);
    free(w);
    free(pq);
    return LMPC_ERR_RUNTIME;
  }
  for (int b = b0; b < b1; ++b) {
    setup_problem(p, cfg, veh, B, b, x_ic, u_ic, X_ref, U_ref, T_ref, bound_left, bound_right,
                  curvatures, vel_ref, ss_x, ss_j);
    p->warm = warm;
    if (warm) set_plan(p, B, b, X_plan, U_plan);
    if (warm && lam_plan && p->S) {
      p->has_lamw = 1;
      for (int j = 0; j < p->S; ++j) p->lamw[j] = lam_plan[(size_t)p->ss_map[j] * B + b];
    }
    int it = 0, st;
    double kk[4] = {0, 0, 0, 0};
    if (!knot0_feasible(p, cfg)) {
      st = LMPC_SOLVE_INFEASIBLE;
      /* still report the rollout so the buffers are defined */
      p->max_iter = 0;
      ipm_solve(p, w, NULL, &it, kk);
    } else {
      st = ipm_solve(p, w, pq, &it, kk);
    }
    for (int i = 0; i < N; ++i)
      for (int k = 0; k < 6; ++k) X_optm[(size_t)(k * N + i) * B + b] = p->z[i][k];
    for (int i = 0; i < N - 1; ++i)
      for (int k = 0; k < 2; ++k) {
        U_optm[(size_t)(k * (N - 1) + i) * B + b] = p->z[i + 1][6 + k];
        dU_optm[(size_t)(k * (N - 1) + i) * B + b] = p->v[i][k];
      }
    if (lambda_out && p->S) {
      for (int j = 0; j < p->S_in; ++j) lambda_out[(size_t)j * B + b] = 0.0;
      for (int j = 0; j < p->S; ++j) lambda_out[(size_t)p->ss_map[j] * B + b] = p->lmb[j];
    }
    status[b] = st;
    iters[b] = it;
    if (kkt)
      for (int k = 0; k < 4; ++k) kkt[(size_t)k * B + b] = kk[k];
  }
  free(p);
  free(w);
  free(pq);
  return LMPC_OK;
}

/* stage-wise linearisation in the C-ABI layout (host pointers) */
int lmpc_oracle_linearize_batch(const lmpc_config* cfg, const lmpc_vehicle* veh, int32_t batch,
                                const double* X_ref, const double* U_ref, const double* T_ref,
                                const double* curvatures, double* A, double* Bm, double* g) {
  const int N = cfg->N, B = batch;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < N - 1; ++i) {
      double x[6], u[2], xp[6], Al[36], Bl[12], gl[6];
      for (int k = 0; k < 6; ++k) x[k] = X_ref[(size_t)(k * N + i) * B + b];
      for (int k = 0; k < 2; ++k) u[k] = U_ref[(size_t)(k * (N - 1) + i) * B + b];
      lmpc_oracle_linearize(veh, x, u, curvatures[(size_t)i * B + b], T_ref[(size_t)i * B + b], Al, Bl, gl, xp);
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) A[((size_t)(r * 6 + c) * (N - 1) + i) * B + b] = Al[r * 6 + c];
        for (int c = 0; c < 2; ++c) Bm[((size_t)(r * 2 + c) * (N - 1) + i) * B + b] = Bl[r * 2 + c];
        g[((size_t)r * (N - 1) + i) * B + b] = gl[r];
      }
    }
  return LMPC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* safe set                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* Brute-force restatement of SafeSetManager::query(SSQuery) + RacingMPC::solve's padding:
 * laps newest first; per lap the K nearest of the 3n unrolled points in (s, e_y), nearest
 * first, ties by lower unrolled index; concatenate, truncate to S, pad with the last column,
 * subtract J[0].  laps: oldest first, x rows [n][6].  Host pointers, batch axis fastest.   */
typedef struct {
  double d;
  int idx;
} cand_t;
static int cand_cmp(const void* a, const void* b) {
  const cand_t *x = a, *y = b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return x->idx - y->idx;
}

int lmpc_oracle_ss_query_batch(int32_t n_laps, const int32_t* n_pts, const double* x, double L,
                               int32_t S, int32_t K, int32_t batch, const double* query, double* ss_x,
                               double* ss_j, int32_t* n_found) {
  const int B = batch;
  int nmax = 0;
  for (int l = 0; l < n_laps; ++l)
    if (n_pts[l] > nmax) nmax = n_pts[l];
  cand_t* cand = malloc(sizeof(cand_t) * 3 * (size_t)(nmax > 0 ? nmax : 1));
  for (int b = 0; b < B; ++b) {
    const double qs = query[b], qe = query[(size_t)B + b];
    int tot = 0;
    for (int l = n_laps - 1; l >= 0 && tot < S; --l) {
      const int n = n_pts[l];
      const double* xl = x;
      for (int j = 0; j < l; ++j) xl += (size_t)n_pts[j] * 6;
      for (int c = 0; c < 3 * n; ++c) {
        const int rep = c / n, j = c % n;
        const double s = xl[(size_t)j * 6] + (rep - 1) * L;
        const double ds = s - qs, de = xl[(size_t)j * 6 + 1] - qe;
        cand[c].d = ds * ds + de * de;
        cand[c].idx = c;
      }
      qsort(cand, 3 * (size_t)n, sizeof(cand_t), cand_cmp);
      const int take = K < 3 * n ? K : 3 * n;
      for (int q = 0; q < take && tot < S; ++q, ++tot) {
        const int c = cand[q].idx, rep = c / n, j = c % n;
        for (int k = 0; k < 6; ++k)
          ss_x[(size_t)(k * S + tot) * B + b] = xl[(size_t)j * 6 + k] + (k == 0 ? (rep - 1) * L : 0.0);
        /* J = steps to finish, unrolled: [J+(n-1), J, J-(n-1)]  (safe_set.cpp:122,128) */
        ss_j[(size_t)tot * B + b] = (double)(n - 1 - j) + (1 - rep) * (double)(n - 1);
      }
    }
    n_found[b] = tot;
    if (tot > 0) {
      for (int q = tot; q < S; ++q) {
        for (int k = 0; k < 6; ++k) ss_x[(size_t)(k * S + q) * B + b] = ss_x[(size_t)(k * S + tot - 1) * B + b];
        ss_j[(size_t)q * B + b] = ss_j[(size_t)(tot - 1) * B + b];
      }
      const double j0 = ss_j[b];
      for (int q = 0; q < S; ++q) ss_j[(size_t)q * B + b] -= j0;
    }
  }
  free(cand);
  return LMPC_OK;
}
