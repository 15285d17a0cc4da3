// racing_lmpc.cpp -- see racing_lmpc.hpp.  Plain C++17, links liblmpc_hip.so only.
#include "racing_lmpc.hpp"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>

namespace lmpc {
namespace mpc {
namespace racing_lmpc {

namespace {
// lmpc_utils/utils.hpp:35-41 (sign(0) = 0 as in CasADi)
double align_abscissa(double s1, double s2, double s_total) {
  const double k = std::fabs(s2 - s1) + s_total / 2.0;
  const double l = k - std::fmod(k, s_total);
  const double d = s2 - s1;
  return s1 + l * ((d > 0) - (d < 0));
}
double scalar(const DM& m, const char* what) {
  if (m.data.size() != 1) throw std::invalid_argument(std::string("RacingLMPC: ") + what + " must be 1 x 1");
  return m.data[0];
}
constexpr double kNewtonPerUnit = 1000.0;  // single_track_planar_model.cpp:215-216: u_lon is in kN
constexpr double kRateRegularisation = 0.1;
}  // namespace

RacingLMPC::RacingLMPC(RacingLMPCConfig::SharedPtr mpc_config, VehicleModel::SharedPtr model, int device)
    : config_(mpc_config), model_(model), three_controls_(false), solved_(false), have_sol_(false), h_(nullptr) {
  if (!config_ || !model_) throw std::invalid_argument("RacingLMPC: null config or model");
  if (model_->name != "single_track_planar_model")
    throw std::runtime_error("RacingLMPC: vehicle model '" + model_->name + "' is not built");
  const RacingLMPCConfig& cf = *config_;
  const std::size_t nu = cf.u_max.data.size();
  if (nu != 2 && nu != 3) throw std::invalid_argument("RacingLMPC: u_max must have 2 or 3 entries");
  three_controls_ = nu == 3;
  if (cf.u_min.data.size() != nu || cf.R.rows != nu || cf.R.cols != nu)
    throw std::invalid_argument("RacingLMPC: u_min / R do not match u_max's control layout");
  if (cf.x_max.data.size() != 6 || cf.x_min.data.size() != 6) throw std::invalid_argument("RacingLMPC: x_max / x_min must have 6 entries");
  lmpc_config& c = c_;
  c = lmpc_config{};
  c.N = static_cast<int32_t>(cf.N);
  c.learning = 0;
  c.num_ss_pts = c.num_ss_pts_per_lap = c.max_lap_stored = 1;
  c.margin = cf.margin;
  c.q_contour = scalar(cf.q_contour, "q_contour");
  c.q_heading = scalar(cf.q_heading, "q_heading");
  c.q_vel = scalar(cf.q_vel, "q_vel");
  c.q_vy = c.q_vyaw = 0.0;
  c.q_boundary = scalar(cf.q_boundary, "q_boundary");
  if (three_controls_) {
    // [f_drive, f_brake, steer] in newtons onto [u_lon (kN), steer]: one force weight, no coupling with the steering row
    const DM& R = cf.R;
    if (R(0, 0) != R(1, 1) || R(0, 1) != 0.0 || R(1, 0) != 0.0 || R(0, 2) != 0.0 || R(2, 0) != 0.0 || R(1, 2) != 0.0 || R(2, 1) != 0.0)
      throw std::invalid_argument(
          "RacingLMPC: a 3 x 3 R folds onto the single longitudinal input only with R(0,0) == R(1,1) and no off-diagonal terms");
    c.R[0] = R(0, 0) * kNewtonPerUnit * kNewtonPerUnit;
    c.R[3] = R(2, 2);
    c.u_max[0] = cf.u_max.data[0] / kNewtonPerUnit;  // largest drive force
    c.u_min[0] = cf.u_min.data[1] / kNewtonPerUnit;  // largest brake force (negative)
    c.u_max[1] = cf.u_max.data[2];
    c.u_min[1] = cf.u_min.data[2];
  } else {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) c.R[a * 2 + b] = cf.R(a, b);
    for (int a = 0; a < 2; ++a) {
      c.u_max[a] = cf.u_max.data[a];
      c.u_min[a] = cf.u_min.data[a];
    }
  }
  // The reference has no rate cost; the solver needs a positive definite one, and the sequential-QP loop needs it for more than
  // that: its QPs carry the cost's Hessian only, so with a vanishing weight on the input increments successive QPs zig-zag
  // (measured with the dense SQP of oracle/nlp.py on tests/cpp/test_racing_lmpc.cpp's first solve: R_d = 1e-6 -> no convergence
  // in 60 QPs, 1e-4 -> 30, 1e-3 -> 9).  One tenth of the input weight, per component.
  c.R_d[0] = kRateRegularisation * c.R[0];
  c.R_d[3] = kRateRegularisation * c.R[3];
  if (!(c.R_d[0] > 0.0) || !(c.R_d[3] > 0.0)) throw std::invalid_argument("RacingLMPC: R must have a positive diagonal");
  for (int k = 0; k < 6; ++k) {
    c.x_max[k] = cf.x_max.data[k];
    c.x_min[k] = cf.x_min.data[k];
  }
  c.max_vel_ref_diff = std::numeric_limits<double>::infinity();  // (a node-side clamp; this class never prepares inputs)
  model_->v.model_id = LMPC_MODEL_SINGLE_TRACK_PLANAR;
  const int rc = lmpc_create(&c, &model_->v, device, &h_);
  if (rc != LMPC_OK) {
    const std::string msg = h_ ? lmpc_last_error(h_) : "allocation failed";
    if (h_) lmpc_destroy(h_);
    h_ = nullptr;
    throw std::runtime_error("RacingLMPC: lmpc_create failed: " + msg);
  }
}

RacingLMPC::~RacingLMPC() { lmpc_destroy(h_); }

const RacingLMPCConfig& RacingLMPC::get_config() const { return *config_; }
VehicleModel& RacingLMPC::get_model() { return *model_; }
const bool& RacingLMPC::solved() const { return solved_; }

void RacingLMPC::solve(const DMDict& in, DMDict& out, Dict& stats) {
  const std::size_t N = config_->N;
  const std::size_t nu = three_controls_ ? 3 : 2;
  const double total_length = static_cast<double>(in.at("total_length"));
  const DM& x_ic = in.at("x_ic");
  const DM& u_ic_in = in.at("u_ic");
  DM X_ref = in.at("X_ref");
  if (X_ref.rows != 6 || X_ref.cols != N || x_ic.data.size() != 6)
    throw std::length_error("RacingLMPC::solve: input dimension does not match MPC dimension");
  for (std::size_t i = 0; i < N; ++i) X_ref(0, i) = align_abscissa(X_ref(0, i), x_ic(0, 0), total_length);  // :194-198
  (void)in.at("U_ref");  // a parameter upstream that no row or cost term reads (racing_lmpc.cpp:43,244)
  const DM& bound_left = in.at("bound_left");
  const DM& bound_right = in.at("bound_right");
  const DM& curvatures = in.at("curvatures");
  const DM& vel_ref = in.at("vel_ref");

  auto fold = [&](const DM& U3) {  // [f_drive, f_brake, steer] (N) -> [u_lon (kN), steer]; the larger force wins (from_base_control)
    if (!three_controls_) return U3;
    DM U2(2, U3.cols);
    for (std::size_t i = 0; i < U3.cols; ++i) {
      const double fd = U3(0, i), fb = U3(1, i);
      U2(0, i) = (std::fabs(fd) > std::fabs(fb) ? fd : fb) / kNewtonPerUnit;
      U2(1, i) = U3(2, i);
    }
    return U2;
  };
  auto unfold = [&](const DM& U2) {
    if (!three_controls_) return U2;
    DM U3(3, U2.cols);
    for (std::size_t i = 0; i < U2.cols; ++i) {
      U3(0, i) = std::max(U2(0, i), 0.0) * kNewtonPerUnit;
      U3(1, i) = std::min(U2(0, i), 0.0) * kNewtonPerUnit;
      U3(2, i) = U2(1, i);
    }
    return U3;
  };
  if (u_ic_in.data.size() != nu) throw std::length_error("RacingLMPC::solve: u_ic does not match the control layout");
  DM u_ic_m(nu, 1);
  u_ic_m.data = u_ic_in.data;
  const DM u_ic = fold(u_ic_m);

  // start iterate (racing_lmpc.cpp:208-243): the optimal reference when given, else the previous solution with its abscissa
  // re-aligned to this call's reference
  DM X0, U0;
  const DM* T = nullptr;
  if (in.count("X_optm_ref")) {
    X0 = in.at("X_optm_ref");
    const DM& U_optm_ref = in.at("U_optm_ref");
    T = &in.at("T_optm_ref");
    if (X0.rows != 6 || X0.cols != N || U_optm_ref.rows != nu || U_optm_ref.cols != N - 1)
      throw std::length_error("RacingLMPC::solve: warm start dimension does not match MPC dimension");
    for (std::size_t i = 0; i < N; ++i) X0(0, i) = align_abscissa(X0(0, i), x_ic(0, 0), total_length);
    U0 = fold(U_optm_ref);
  } else {
    if (!have_sol_) throw std::runtime_error("No warm start given and no previous solution found.");
    T = &in.at("T_ref");
    X0 = sol_X_;
    U0 = sol_U_;
    for (std::size_t i = 0; i < N; ++i) X0(0, i) = align_abscissa(X0(0, i), X_ref(0, i), total_length);
  }
  if (T->data.size() != N - 1 || bound_left.data.size() != N || bound_right.data.size() != N || curvatures.data.size() != N ||
      vel_ref.data.size() != N)
    throw std::length_error("RacingLMPC::solve: input dimension does not match MPC dimension");
  // the iterate's first knot is the measured state (the row x_0 = x_ic, :174)
  for (int r = 0; r < 6; ++r) X0(r, 0) = x_ic.data[r];

  DM X(6, N), U(2, N - 1), dU(2, N - 1);
  int32_t status = 0, iters = 0, sqp_iters = 0;
  double move = 0.0, defect = 0.0;
  const int32_t max_sqp = static_cast<int32_t>(std::min<int64_t>(std::max<int64_t>(config_->max_iter, 1), 200));
  const int rc = lmpc_solve_full_dynamics_host(h_, x_ic.data.data(), u_ic.data.data(), X0.data.data(), U0.data.data(), T->data.data(),
                                               bound_left.data.data(), bound_right.data.data(), curvatures.data.data(),
                                               vel_ref.data.data(), total_length, nullptr, nullptr, max_sqp, 1e-8, X.data.data(),
                                               U.data.data(), dU.data.data(), nullptr, &status, &iters, &sqp_iters, &move, &defect);
  stats["iter_count"] = static_cast<double>(iters);
  stats["sqp_iter_count"] = static_cast<double>(sqp_iters);
  stats["dynamics_defect"] = defect;
  if (rc != LMPC_OK) {
    std::cerr << "RacingLMPC::solve: " << lmpc_last_error(h_) << '\n';
    stats["success"] = 0.0;
    out["X_optm"] = X0;  // upstream: the solver's debug values (:261-262)
    out["U_optm"] = unfold(U0);
    return;
  }
  out["X_optm"] = X;
  out["U_optm"] = unfold(U);
  // solve_limited() (:254) accepts a run that stops on its iteration or time limit: the loop running out of QPs with every
  // QP solved (`move` still above the step tolerance) is that case.  A QP that is infeasible about the iterate, or not solved,
  // is the failure (IPOPT: restoration failed / infeasible problem detected -> the catch branch, :258-265)
  const bool ok = status == LMPC_SOLVE_OPTIMAL;
  stats["success"] = ok ? 1.0 : 0.0;
  stats["converged"] = (ok && move <= 1e-8) ? 1.0 : 0.0;
  if (!ok) {
    std::cerr << "RacingLMPC::solve: the QP about the iterate " << (status == LMPC_SOLVE_INFEASIBLE ? "is infeasible" : "hit the iteration cap")
              << '\n';
    return;
  }
  solved_ = true;
  have_sol_ = true;
  sol_X_ = X;
  sol_U_ = U;
}

void RacingLMPC::create_warm_start(const DMDict& in, DMDict& out) {
  const std::size_t N = config_->N;
  const DM& P0 = in.at("P0");
  const DM& Yaws = in.at("Yaws");
  const DM& Radii = in.at("Radii");
  const double current_vel = static_cast<double>(in.at("current_vel"));
  const double target_vel = static_cast<double>(in.at("target_vel"));
  if (P0.size2() != N) throw std::length_error("create_warm_start: P0 dimension does not match MPC dimension.");
  if (Yaws.size2() != N) throw std::length_error("create_warm_start: Yaws dimension does not match MPC dimension.");
  if (current_vel <= 0.0) throw std::range_error("Current velocity cannot be smaller than or equal to zero.");
  if (target_vel <= 0.0) throw std::range_error("Target velocity cannot be smaller than or equal to zero.");
  DM X_ref(6, N), U_ref(three_controls_ ? 3 : 2, N - 1);
  for (std::size_t i = 0; i < N; ++i) {
    X_ref(0, i) = P0(0, i);
    X_ref(1, i) = P0(1, i);
    X_ref(2, i) = Yaws.data[i];
    X_ref(3, i) = current_vel + (target_vel - current_vel) * (N > 1 ? double(i) / double(N - 1) : 0.0);
    X_ref(5, i) = X_ref(3, i) / Radii.data[i];
  }
  for (std::size_t i = 0; i + 1 < N; ++i) {  // force from Newton's second law, steering from pure pursuit (:303-326)
    const double v0 = X_ref(3, i), v1 = X_ref(3, i + 1);
    const double d = std::hypot(P0(0, i) - P0(0, i + 1), P0(1, i) - P0(1, i + 1));
    const double f = model_->v.m * (v1 * v1 - v0 * v0) / (2 * d);
    const double steer = std::atan(model_->v.l / Radii.data[i]);
    if (three_controls_) {
      U_ref(f > 0.0 ? 0 : 1, i) = f;
      U_ref(2, i) = steer;
    } else {
      U_ref(0, i) = f / kNewtonPerUnit;
      U_ref(1, i) = steer;
    }
  }
  out["X_ref"] = X_ref;
  out["U_ref"] = U_ref;
}

}  // namespace racing_lmpc
}  // namespace mpc
}  // namespace lmpc
