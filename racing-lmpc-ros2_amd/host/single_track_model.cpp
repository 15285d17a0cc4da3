// single_track_model.cpp -- see single_track_model.hpp.  Plain C++17, no device code.
#include "single_track_model.hpp"

#include <cmath>

namespace lmpc {
namespace vehicle_model {
namespace single_track_planar_model {

void continuous_dynamics(const lmpc_vehicle& v, const double* x, const double* u, double k, double* f) {
  const double g = 9.8;  // single_track_planar_model.cpp:18
  const double ey = x[1], phi = x[2], vx = x[3], vy = x[4], om = x[5], ul = u[0], de = u[1];
  const double m = v.m, l = v.l, lr = v.cg_ratio * l, lf = l - lr;
  const double th = std::tanh(ul);
  const double fd = 1000.0 * ul * (0.5 * th + 0.5), fb = 1000.0 * ul * (0.5 - 0.5 * th);      // :215-216
  const double Fxf = 0.5 * v.kd * fd + 0.5 * v.kb * fb - 0.5 * v.fr * m * g * lr / l;         // :258
  const double Fxr = 0.5 * (1 - v.kd) * fd + 0.5 * (1 - v.kb) * fb - 0.5 * v.fr * m * g * lf / l;
  const double ax = (fd + fb - 0.5 * v.cd * v.Af * vx * vx - v.fr * m * g) / m;               // :267 (no rho)
  const double Fzf = 0.5 * m * g * lr / l - 0.5 * (v.h / l) * m * ax + 0.25 * v.cl_f * v.rho * v.Af * vx * vx;
  const double Fzr = 0.5 * m * g * lf / l + 0.5 * (v.h / l) * m * ax + 0.25 * v.cl_r * v.rho * v.Af * vx * vx;
  const double af = de - std::atan((lf * om + vy) / (vx + 1e-3));                             // :280-283
  const double ar = std::atan((lr * om - vy) / (vx + 1e-3));
  const double Fyf = v.mu * Fzf * std::sin(v.Cf * std::atan(v.Bf * af));                      // :299-300
  const double Fyr = v.mu * Fzr * std::sin(v.Cr * std::atan(v.Br * ar));
  const double cd_ = std::cos(de), sd_ = std::sin(de);
  const double sdot = (vx * std::cos(phi) - vy * std::sin(phi)) / (1.0 - ey * k);             // :322-330
  f[0] = sdot;
  f[1] = vx * std::sin(phi) + vy * std::cos(phi);
  f[2] = om - k * sdot;
  f[3] = (2 * Fxr + 2 * Fxf * cd_ - 2 * Fyf * sd_ - 0.5 * v.cd * v.rho * v.Af * vx * vx) / m + om * vy;  // :314-316 (with rho)
  f[4] = (2 * Fyr + 2 * Fyf * cd_ + 2 * Fxf * sd_) / m - om * vx;
  f[5] = (-2 * Fyr * lr + (2 * Fyf * cd_ + 2 * Fxf * sd_) * lf) / v.Jzz;
}

void discrete_dynamics(const lmpc_vehicle& v, const double* x, const double* u, double k, double dt, double* xn) {
  double k1[6], k2[6], k3[6], k4[6], xs[6];
  continuous_dynamics(v, x, u, k, k1);
  if (v.integrator == LMPC_INTEGRATOR_EULER) {  // utils.cpp:110-123
    for (int r = 0; r < 6; ++r) xn[r] = x[r] + dt * k1[r];
    return;
  }
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt / 2.0 * k1[r];
  continuous_dynamics(v, xs, u, k, k2);
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt / 2.0 * k2[r];
  continuous_dynamics(v, xs, u, k, k3);
  for (int r = 0; r < 6; ++r) xs[r] = x[r] + dt * k3[r];
  continuous_dynamics(v, xs, u, k, k4);
  for (int r = 0; r < 6; ++r) xn[r] = x[r] + dt / 6 * (k1[r] + 2 * k2[r] + 2 * k3[r] + k4[r]);  // utils.cpp:88-108
}

void from_base_control(const double* ub, double* u) {
  u[0] = std::fabs(ub[0]) > std::fabs(ub[1]) ? ub[0] : ub[1];
  u[1] = ub[2];
}

void to_base_control(const double* u, double* ub) {
  ub[0] = u[0] / (1.0 + std::exp(-u[0]));
  ub[1] = u[0] / (1.0 + std::exp(u[0]));
  ub[2] = u[1];
}

}  // namespace single_track_planar_model
}  // namespace vehicle_model
}  // namespace lmpc

// plain-C handle on the host model (tests compare it with the device kernels)
extern "C" void lmpc_host_discrete_dynamics(const lmpc_vehicle* v, const double* x, const double* u, double k, double dt,
                                            double* x_next) {
  lmpc::vehicle_model::single_track_planar_model::discrete_dynamics(*v, x, u, k, dt, x_next);
}
