// racing_mpc_node_core.cpp -- see racing_mpc_node_core.hpp.  Plain C++17; links the facade only.
#include "racing_mpc_node_core.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <iostream>
#include <stdexcept>
#include <string>

#include "single_track_model.hpp"

namespace lmpc {
namespace mpc {
namespace racing_mpc {

namespace stm = lmpc::vehicle_model::single_track_planar_model;

RacingMPCNodeCore::RacingMPCNodeCore(RacingMPC::SharedPtr mpc, RacingMPC::SharedPtr mpc_full,
                                     lmpc::vehicle_model::racing_trajectory::RacingTrajectory::SharedPtr track, double dt,
                                     RacingMPCStepMode step_mode, int delay_step, bool jit)
    : mpc_(mpc), mpc_full_(mpc_full), track_(track), dt_(dt), step_mode_(step_mode), delay_step_(delay_step), jitted_(!jit),
      speed_limit_(mpc->get_config().c.x_max[3]) {  // racing_mpc_node.hpp:69
  if (!mpc_ || !mpc_full_ || !track_) throw std::invalid_argument("RacingMPCNodeCore: null controller or track");
  const std::size_t N = static_cast<std::size_t>(mpc_->get_config().c.N);
  DM T(1, N - 1);
  for (auto& v : T.data) v = dt_;
  sol_in_["T_ref"] = T;                               // racing_mpc_node.cpp:65
  sol_in_["total_length"] = DM(track_->total_length());
}

void RacingMPCNodeCore::discrete_dynamics(const double* x, const double* u, double* xn) const {
  stm::discrete_dynamics(mpc_->get_model().v, x, u, track_->curvature_interpolation(x[0]), dt_, xn);  // :68-76
}

DiagnosticStatus CycleWindow::status(const std::string& name, const std::string& message, double warn_threshold) const {
  double mx = 0.0, mn = 0.0, mean = 0.0;  // an empty window reports zeros (cycle_profiler.hpp:108-113)
  if (n_ > 0) {
    mx = mn = buf_[0];
    for (std::size_t i = 0; i < n_; ++i) {
      mx = std::max(mx, buf_[i]);
      mn = std::min(mn, buf_[i]);
      mean += buf_[i];
    }
    mean /= static_cast<double>(n_);
  }
  DiagnosticStatus s;
  s.values = {{"max", std::to_string(mx)}, {"mean", std::to_string(mean)}, {"min", std::to_string(mn)}};
  s.level = mx > warn_threshold ? DiagnosticStatus::WARN : DiagnosticStatus::OK;
  s.name = name;
  s.message = message;
  return s;
}

bool RacingMPCNodeCore::take_diagnostics(DiagnosticArray& diagnostics) {
  if (!diagnostics_ready_) return false;
  diagnostics = diagnostics_;
  diagnostics_ready_ = false;
  return true;
}

void RacingMPCNodeCore::change_trajectory(lmpc::vehicle_model::racing_trajectory::RacingTrajectory::SharedPtr new_track) {
  using lmpc::FrenetPose2D;
  using lmpc::Pose2D;
  if (!new_track || new_track == track_) return;
  if (mpc_->solved() && last_x_.cols > 0) {  // :540-551: the previous solution into the new coordinate system
    for (std::size_t i = 0; i < last_x_.cols; ++i) {
      FrenetPose2D old_fp, new_fp;
      old_fp.position.s = last_x_(0, i);
      old_fp.position.t = last_x_(1, i);
      old_fp.yaw = last_x_(2, i);
      Pose2D gp;
      track_->frenet_to_global(old_fp, gp);
      new_track->global_to_frenet(gp, new_fp);
      last_x_(0, i) = new_fp.position.s;
      last_x_(1, i) = new_fp.position.t;
      last_x_(2, i) = new_fp.yaw;
    }
  }
  track_ = new_track;  // discrete_dynamics() reads the curvature from track_ (:558-565)
  sol_in_["total_length"] = DM(track_->total_length());
}

void RacingMPCNodeCore::set_speed_limit(const double& speed_limit) { speed_limit_ = speed_limit; }

void RacingMPCNodeCore::set_speed_scale(const double& speed_scale) {
  speed_scale_ = (speed_scale > 1.0 || speed_scale <= 0.0) ? 0.2 : speed_scale;
}

RacingMPCNodeCore::Result RacingMPCNodeCore::step(const VehicleState& st, VehicleActuation& act, MPCTelemetry& tel) {
  using lmpc::FrenetPose2D;
  using lmpc::Pose2D;
  const auto t0 = std::chrono::system_clock::now();
  const auto& cfg = mpc_->get_config();
  const std::size_t N = static_cast<std::size_t>(cfg.c.N);
  tel = MPCTelemetry();

  // state in the Frenet frame of the track (:181-185); from_base_state is the identity for this model
  Pose2D gp;
  gp.position.x = st.x;
  gp.position.y = st.y;
  gp.yaw = st.psi;
  FrenetPose2D fp;
  track_->global_to_frenet(gp, fp);
  DM x_ic(6, 1);
  x_ic(0, 0) = fp.position.s;
  x_ic(1, 0) = fp.position.t;
  x_ic(2, 0) = fp.yaw;
  x_ic(3, 0) = st.v_long;
  x_ic(4, 0) = st.v_tran;
  x_ic(5, 0) = st.w_psi;
  const double u_base[3] = {act.u_a > 0.0 ? act.u_a : 0.0, act.u_a < 0.0 ? act.u_a : 0.0, act.u_steer};  // :191-195
  DM u_ic(2, 1);
  stm::from_base_control(u_base, u_ic.data.data());
  sol_in_["u_ic"] = u_ic;
  sol_in_["t_ic"] = DM(st.t);

  const bool first = !mpc_full_->solved();
  if (first) {  // :210-235
    last_x_ = DM(6, N);
    last_u_ = DM(2, N - 1);
    for (auto& v : last_u_.data) v = 1e-9;
    last_du_ = DM(2, N - 1);
    if (cfg.c.learning) last_convex_combi_ = DM(static_cast<std::size_t>(cfg.c.num_ss_pts), 1);
    for (int r = 0; r < 6; ++r) last_x_(r, 0) = x_ic(r, 0);
    for (std::size_t i = 1; i < N; ++i) discrete_dynamics(&last_x_(0, i - 1), &last_u_(0, i - 1), &last_x_(0, i));
    sol_in_["X_optm_ref"] = last_x_;
    sol_in_["U_optm_ref"] = last_u_;
    sol_in_["dU_optm_ref"] = last_du_;
    if (cfg.c.learning) sol_in_["convex_combi_optm_ref"] = last_convex_combi_;
    sol_in_["T_optm_ref"] = sol_in_.at("T_ref");
    sol_in_["X_ref"] = last_x_;
    sol_in_["U_ref"] = last_u_;
    sol_in_["x_ic"] = x_ic;
  } else {      // :236-259
    if (step_mode_ == RacingMPCStepMode::CONTINUOUS) {
      DM x_next(6, 1);
      discrete_dynamics(x_ic.data.data(), &last_u_(0, 0), x_next.data.data());
      sol_in_["x_ic"] = x_next;
    } else {
      sol_in_["x_ic"] = x_ic;
    }
    DM nx(6, N), nu(2, N - 1), ndu(2, N - 1);
    for (std::size_t i = 0; i + 1 < N; ++i)
      for (int r = 0; r < 6; ++r) nx(r, i) = last_x_(r, i + 1);
    for (std::size_t i = 0; i + 2 < N; ++i)
      for (int r = 0; r < 2; ++r) {
        nu(r, i) = last_u_(r, i + 1);
        ndu(r, i) = last_du_(r, i + 1);
      }
    for (int r = 0; r < 2; ++r) nu(r, N - 2) = last_u_(r, N - 2);  // the last input is repeated, its rate is zero
    discrete_dynamics(&nx(0, N - 2), &nu(0, N - 2), &nx(0, N - 1));
    last_x_ = nx;
    last_u_ = nu;
    last_du_ = ndu;
    sol_in_["X_ref"] = last_x_;
    sol_in_["U_ref"] = last_u_;
    sol_in_["X_optm_ref"] = last_x_;
    sol_in_["U_optm_ref"] = last_u_;
    sol_in_["dU_optm_ref"] = last_du_;
    if (cfg.c.learning) sol_in_["convex_combi_ref"] = last_convex_combi_;
  }

  // references at the plan's abscissae (:261-292)
  DM left(1, N), right(1, N), curv(1, N), vref(1, N);
  for (std::size_t i = 0; i < N; ++i) {
    const double s = last_x_(0, i);
    left(0, i) = track_->left_boundary_interpolation(s);
    right(0, i) = track_->right_boundary_interpolation(s);
    curv(0, i) = track_->curvature_interpolation(s);
    const double current_speed = last_x_(3, i);
    const double ref_speed = track_->velocity_interpolation(s) * speed_scale_;
    const double d = cfg.c.max_vel_ref_diff;
    const double limit_clipped = std::clamp(speed_limit_, current_speed - d, current_speed + d);
    // a valid profile is positive; a negative one means "use the speed limit"
    vref(0, i) = ref_speed > 0.0 ? std::min(std::clamp(ref_speed, current_speed - d, current_speed + d), limit_clipped) : limit_clipped;
  }
  sol_in_["bound_left"] = left;
  sol_in_["bound_right"] = right;
  sol_in_["curvatures"] = curv;
  sol_in_["vel_ref"] = vref;

  DMDict sol_out;
  Dict stats;
  if (first) {  // :299-314: the full-dynamics controller, once
    mpc_full_->solve(sol_in_, sol_out, stats);
    if (!mpc_full_->solved()) return Result::INITIAL_SOLVE_FAILED;  // (upstream logs FATAL and reads the missing keys)
    last_x_ = sol_out.at("X_optm");
    last_u_ = sol_out.at("U_optm");
    last_du_ = sol_out.at("dU_optm");
    if (cfg.c.learning) last_convex_combi_ = sol_out.at("convex_combi_optm");
    return Result::INITIAL_SOLVE;
  }
  mpc_->solve(sol_in_, sol_out, stats);
  if (sol_out.count("X_optm")) {  // :322-332: on failure the shifted plan stays
    last_x_ = sol_out["X_optm"];
    last_u_ = sol_out["U_optm"];
    last_du_ = sol_out["dU_optm"];
    if (cfg.c.learning && sol_out.count("convex_combi_optm")) last_convex_combi_ = sol_out["convex_combi_optm"];
    tel.solved = true;
  } else {
    std::cerr << "MPC could not be solved." << '\n';
    tel.solved = false;
  }
  tel.state = last_x_.data;
  tel.control = last_u_.data;
  if (!jitted_) {  // :337-342
    jitted_ = true;
    return Result::JIT_DISCARDED;
  }
  tel.solve_time = std::chrono::duration<double, std::milli>(std::chrono::system_clock::now() - t0).count();
  // :351-384: the two windows, and every `capacity` published steps the diagnostics array
  profiler_.add(tel.solve_time);
  if (stats.count("iter_count")) profiler_iter_count_.add(stats.at("iter_count"));
  if (++profile_step_count_ % profiler_.capacity() == 0) {
    diagnostics_ = DiagnosticArray();
    diagnostics_.stamp = st.t;
    diagnostics_.status.push_back(profiler_.status("Racing MPC Solve Time", "(ms)", dt_ * 1e3));
    diagnostics_.status.push_back(profiler_iter_count_.status("Racing MPC Iteration Count", "Number of Solver Iterations", 50));
    diagnostics_ready_ = true;
    profile_step_count_ = 0;
  }
  // actuation from column delay_step of the plan (:395-413)
  const std::size_t col = std::min(static_cast<std::size_t>(std::max(delay_step_, 0)), N - 2);
  double ub[3];
  stm::to_base_control(&last_u_(0, col), ub);
  act.u_a = std::fabs(ub[0]) > std::fabs(ub[1]) ? ub[0] : ub[1];
  act.u_steer = ub[2];
  return Result::PUBLISHED;
}

}  // namespace racing_mpc
}  // namespace mpc
}  // namespace lmpc
