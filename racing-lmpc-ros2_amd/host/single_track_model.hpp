// single_track_model.hpp -- host-side evaluation of the vehicle model for ONE car: what the controller node needs
// outside the solve (the reference evaluates the model's CasADi functions on the host for exactly these steps):
//   discrete_dynamics      SingleTrackPlanarModel::compile_dynamics + rk4 / euler
//                          (single_track_planar_model.cpp:195-368; lmpc_utils/src/utils.cpp:88-123)
//   from_base_control      [FD, FB, STEER] -> [LON, STEER], LON = |FD| > |FB| ? FD : FB          (:401-407)
//   to_base_control        [LON, STEER] -> [LON s(LON), LON s(-LON), STEER], s = logistic        (:395-400)
// The batched solve never calls this (its model lives in csrc/lmpc_dynamics.hip.h); tests hold the two to 1e-12.
#ifndef LMPC_HOST_SINGLE_TRACK_MODEL_HPP_
#define LMPC_HOST_SINGLE_TRACK_MODEL_HPP_

#include "lmpc_hip.h"

namespace lmpc {
namespace vehicle_model {
namespace single_track_planar_model {

// x_dot = f(x, u, k): x = [s, e_y, e_psi, vx, vy, w], u = [u_lon, steer]
void continuous_dynamics(const lmpc_vehicle& v, const double* x, const double* u, double k, double* x_dot);
// x+ = f_d(x, u, k, dt) with the vehicle's integrator (lmpc_vehicle.integrator)
void discrete_dynamics(const lmpc_vehicle& v, const double* x, const double* u, double k, double dt, double* x_next);
void from_base_control(const double* u_base3, double* u2);
void to_base_control(const double* u2, double* u_base3);

}  // namespace single_track_planar_model
}  // namespace vehicle_model
}  // namespace lmpc
#endif
