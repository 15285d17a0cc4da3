// lmpc_dynamics.hip.h -- single-track planar model on the device (gfx950, fp64).
//
// Restates SingleTrackPlanarModel::compile_dynamics
//   (src/vehicle_dynamics_models/single_track_planar_model/src/single_track_planar_model.cpp:195-332,
//    simplify_lon_control = true, use_frenet = true) with hand-derived partial derivatives in place
// of CasADi's symbolic Jacobian (:344-345, :377-378), and lmpc::utils::rk4_function
//   (src/tools/lmpc_utils/src/utils.cpp:88-108).
// f does not depend on s, so column 0 of df/dx is zero; the Jacobian's sparsity
//   row 0 (s_dot)   : e_y, e_psi, vx, vy        row 3..5 (vx_dot, vy_dot, om_dot): vx, vy, om, u_lon, steer
//   row 1 (e_y_dot) : e_psi, vx, vy             row 2 (e_psi_dot) = e_5 - k * row 0
// is hard-coded in jvp() below.
#ifndef LMPC_DYNAMICS_HIP_H_
#define LMPC_DYNAMICS_HIP_H_

#include <hip/hip_runtime.h>

#include "lmpc_device.h"

#define LMPC_GRAVITY 9.8  // single_track_planar_model.cpp:18

// Quantities that depend on u only (u is held over the RK4 step): computed once per stage.
struct lmpc_uterms {
  double Fxf, Fxr, dFxf, dFxr;  // longitudinal tyre forces and d/du_lon
  double fsum;                  // fd + fb
  double dfsum;                 // d(fd + fb)/du_lon
  double cd_, sd_;              // cos/sin(steer)
  double de;                    // steer
};

__device__ __forceinline__ void lmpc_u_terms(const lmpc_vehicle& v, double ul, double de, lmpc_uterms& t) {
  const double th = tanh(ul), sech2 = 1.0 - th * th;
  const double fd = 1000.0 * ul * (0.5 * th + 0.5);         // :215
  const double fb = 1000.0 * ul * (0.5 - 0.5 * th);         // :216 (tanh(-u) = -tanh(u))
  const double dfd = 1000.0 * ((0.5 * th + 0.5) + ul * 0.5 * sech2);
  const double dfb = 1000.0 * ((0.5 - 0.5 * th) - ul * 0.5 * sech2);
  const double lr = v.cg_ratio * v.l, lf = v.l - lr;
  t.Fxf = 0.5 * v.kd * fd + 0.5 * v.kb * fb - 0.5 * v.fr * v.m * LMPC_GRAVITY * lr / v.l;              // :258
  t.Fxr = 0.5 * (1 - v.kd) * fd + 0.5 * (1 - v.kb) * fb - 0.5 * v.fr * v.m * LMPC_GRAVITY * lf / v.l;  // :261
  t.dFxf = 0.5 * v.kd * dfd + 0.5 * v.kb * dfb;
  t.dFxr = 0.5 * (1 - v.kd) * dfd + 0.5 * (1 - v.kb) * dfb;
  t.fsum = fd + fb;
  t.dfsum = dfd + dfb;
  sincos(de, &t.sd_, &t.cd_);
  t.de = de;
}

// Non-zero partials of f at one point.
struct lmpc_fjac {
  double a01, a02, a03, a04;  // d s_dot / d(e_y, e_psi, vx, vy)
  double a12, a13, a14;       // d e_y_dot / d(e_psi, vx, vy)
  double a3[3], a4[3], a5[3]; // d (vx_dot, vy_dot, om_dot) / d(vx, vy, om)
  double b3[2], b4[2], b5[2]; // ... / d(u_lon, steer)
  double k;                   // curvature (row 2 = e_5 - k * row 0)
};

template <bool WITH_JAC>
__device__ __forceinline__ void lmpc_f(const lmpc_vehicle& v, const lmpc_uterms& ut, const double* x, double k,
                                       double* f, lmpc_fjac* J) {
  const double ey = x[1], phi = x[2], vx = x[3], vy = x[4], om = x[5];
  const double m = v.m, l = v.l, lr = v.cg_ratio * l, lf = l - lr;
  const double vsq = vx * vx;                                                        // :210
  const double ax = (ut.fsum - 0.5 * v.cd * v.Af * vsq - v.fr * m * LMPC_GRAVITY) / m;  // :267
  const double hl = v.h / l;
  const double Fzf = 0.5 * m * LMPC_GRAVITY * lr / l - 0.5 * hl * m * ax + 0.25 * v.cl_f * v.rho * v.Af * vsq;  // :270
  const double Fzr = 0.5 * m * LMPC_GRAVITY * lf / l + 0.5 * hl * m * ax + 0.25 * v.cl_r * v.rho * v.Af * vsq;  // :274
  const double den = vx + 1e-3;
  const double rf = (lf * om + vy) / den, rr = (lr * om - vy) / den;
  const double af = ut.de - atan(rf), ar = atan(rr);                                 // :280-283
  const double tf = atan(v.Bf * af), tr = atan(v.Br * ar);
  double Sf, Cf_, Sr, Cr_;
  sincos(v.Cf * tf, &Sf, &Cf_);
  sincos(v.Cr * tr, &Sr, &Cr_);
  const double Fyf = v.mu * Fzf * Sf, Fyr = v.mu * Fzr * Sr;                         // :299-300
  double sph, cph;
  sincos(phi, &sph, &cph);
  const double q = 1.0 / (1.0 - ey * k);
  const double num = vx * cph - vy * sph;
  const double drag = 0.5 * v.cd * v.rho * v.Af;
  const double cd_ = ut.cd_, sd_ = ut.sd_;
  f[0] = num * q;                                                                    // :322,328
  f[1] = vx * sph + vy * cph;                                                        // :323
  f[2] = om - k * f[0];                                                              // :329
  f[3] = (2 * ut.Fxr + 2 * ut.Fxf * cd_ - 2 * Fyf * sd_ - drag * vsq) / m + om * vy;    // :314-316
  f[4] = (2 * Fyr + 2 * Fyf * cd_ + 2 * ut.Fxf * sd_) / m - om * vx;                 // :317-319
  f[5] = (-2 * Fyr * lr + (2 * Fyf * cd_ + 2 * ut.Fxf * sd_) * lf) / v.Jzz;          // :309-310
  if (WITH_JAC) {
    const double dax_vx = -v.cd * v.Af * vx / m;
    const double dax_ul = ut.dfsum / m;
    const double dFzf_vx = -0.5 * hl * m * dax_vx + 0.5 * v.cl_f * v.rho * v.Af * vx;
    const double dFzr_vx = 0.5 * hl * m * dax_vx + 0.5 * v.cl_r * v.rho * v.Af * vx;
    const double dFzf_ul = -0.5 * hl * m * dax_ul, dFzr_ul = 0.5 * hl * m * dax_ul;
    const double wf = 1.0 / ((1.0 + rf * rf) * den), wr = 1.0 / ((1.0 + rr * rr) * den);
    const double Df = Cf_ * v.Cf * v.Bf / (1.0 + v.Bf * af * v.Bf * af);
    const double Dr = Cr_ * v.Cr * v.Br / (1.0 + v.Br * ar * v.Br * ar);
    // d Fy / d(vx, vy, om, u_lon, steer); d alpha_f = (rf wf, -wf, -lf wf), d alpha_r = (-rr wr, -wr, lr wr)
    const double gF = v.mu * Fzf * Df, gR = v.mu * Fzr * Dr;
    const double dFyf[5] = {v.mu * dFzf_vx * Sf + gF * rf * wf, -gF * wf, -gF * lf * wf, v.mu * dFzf_ul * Sf, gF};
    const double dFyr[5] = {v.mu * dFzr_vx * Sr - gR * rr * wr, -gR * wr, gR * lr * wr, v.mu * dFzr_ul * Sr, 0.0};
    J->k = k;
    J->a01 = num * q * q * k;
    J->a02 = (-vx * sph - vy * cph) * q;
    J->a03 = cph * q;
    J->a04 = -sph * q;
    J->a12 = num;
    J->a13 = sph;
    J->a14 = cph;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      J->a3[j] = (-2 * dFyf[j] * sd_) / m;
      J->a4[j] = (2 * dFyr[j] + 2 * dFyf[j] * cd_) / m;
      J->a5[j] = (-2 * dFyr[j] * lr + 2 * dFyf[j] * cd_ * lf) / v.Jzz;
    }
    J->a3[0] += -2 * drag * vx / m;
    J->a3[1] += om;
    J->a3[2] += vy;
    J->a4[0] += -om;
    J->a4[2] += -vx;
    J->b3[0] = (2 * ut.dFxr + 2 * ut.dFxf * cd_ - 2 * dFyf[3] * sd_) / m;
    J->b4[0] = (2 * dFyr[3] + 2 * dFyf[3] * cd_ + 2 * ut.dFxf * sd_) / m;
    J->b5[0] = (-2 * dFyr[3] * lr + (2 * dFyf[3] * cd_ + 2 * ut.dFxf * sd_) * lf) / v.Jzz;
    const double dvy_de = 2 * dFyf[4] * cd_ - 2 * Fyf * sd_ + 2 * ut.Fxf * cd_;
    J->b3[1] = (-2 * ut.Fxf * sd_ - 2 * dFyf[4] * sd_ - 2 * Fyf * cd_) / m;
    J->b4[1] = dvy_de / m;
    J->b5[1] = dvy_de * lf / v.Jzz;
  }
}

// out = (df/dx) tx + (df/du) tu  for one tangent direction
__device__ __forceinline__ void lmpc_jvp(const lmpc_fjac& J, const double* tx, const double* tu, double* o) {
  o[0] = J.a01 * tx[1] + J.a02 * tx[2] + J.a03 * tx[3] + J.a04 * tx[4];
  o[1] = J.a12 * tx[2] + J.a13 * tx[3] + J.a14 * tx[4];
  o[2] = tx[5] - J.k * o[0];
  o[3] = J.a3[0] * tx[3] + J.a3[1] * tx[4] + J.a3[2] * tx[5] + J.b3[0] * tu[0] + J.b3[1] * tu[1];
  o[4] = J.a4[0] * tx[3] + J.a4[1] * tx[4] + J.a4[2] * tx[5] + J.b4[0] * tu[0] + J.b4[1] * tu[1];
  o[5] = J.a5[0] * tx[3] + J.a5[1] * tx[4] + J.a5[2] * tx[5] + J.b5[0] * tu[0] + J.b5[1] * tu[1];
}

// x+ = f_d(x, u, k, dt): the model's discrete dynamics -- classic RK4 with u, k held over the step (utils.cpp:88-108) or,
// with modeling.integrator_type = "euler", x + dt f(x, u, k) (utils.cpp:110-123)
__device__ __forceinline__ void lmpc_fd(const lmpc_vehicle& v, const double* x, const double* u, double k,
                                        double dt, double* xp) {
  lmpc_uterms ut;
  lmpc_u_terms(v, u[0], u[1], ut);
  double k1[6], k2[6], k3[6], k4[6], xs[6];
  lmpc_f<false>(v, ut, x, k, k1, nullptr);
  if (v.integrator == LMPC_INTEGRATOR_EULER) {
#pragma unroll
    for (int r = 0; r < 6; ++r) xp[r] = x[r] + dt * k1[r];
    return;
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt / 2.0 * k1[r];
  lmpc_f<false>(v, ut, xs, k, k2, nullptr);
#pragma unroll
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt / 2.0 * k2[r];
  lmpc_f<false>(v, ut, xs, k, k3, nullptr);
#pragma unroll
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt * k3[r];
  lmpc_f<false>(v, ut, xs, k, k4, nullptr);
#pragma unroll
  for (int r = 0; r < 6; ++r) xp[r] = x[r] + dt / 6 * (k1[r] + 2 * k2[r] + 2 * k3[r] + k4[r]);
}

#endif
