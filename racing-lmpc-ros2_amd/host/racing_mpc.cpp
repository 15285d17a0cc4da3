// racing_mpc.cpp -- see racing_mpc.hpp.  Plain C++17, links liblmpc_hip.so only.
#include "racing_mpc.hpp"

#include <algorithm>
#include <cmath>
#include <iostream>
#include <stdexcept>

namespace lmpc {
namespace mpc {
namespace racing_mpc {

namespace {
// lmpc_utils/utils.hpp:35-41 (sign(0) = 0 as in CasADi)
double align_abscissa(double s1, double s2, double s_total) {
  const double k = std::fabs(s2 - s1) + s_total / 2.0;
  const double l = k - std::fmod(k, s_total);
  const double d = s2 - s1;
  return s1 + l * ((d > 0) - (d < 0));
}
}  // namespace

RacingMPC::RacingMPC(RacingMPCConfig::SharedPtr mpc_config, VehicleModel::SharedPtr model, const bool& full_dynamics,
                     int device)
    : config_(mpc_config), model_(model), full_dynamics_(full_dynamics), solved_(false), ran_(false), h_(nullptr) {
  if (!config_ || !model_) throw std::invalid_argument("RacingMPC: null config or model");
  if (model_->name != "single_track_planar_model")
    throw std::runtime_error("RacingMPC: vehicle model '" + model_->name + "' is not built");
  model_->v.model_id = LMPC_MODEL_SINGLE_TRACK_PLANAR;
  const int rc = lmpc_create(&config_->c, &model_->v, device, &h_);
  if (rc != LMPC_OK) {
    const std::string msg = h_ ? lmpc_last_error(h_) : "allocation failed";
    if (h_) lmpc_destroy(h_);
    h_ = nullptr;
    throw std::runtime_error("RacingMPC: lmpc_create failed: " + msg);
  }
  // racing_mpc.cpp:58-61: manager sized by max_lap_stored and recorder writing under path_prefix exist for EVERY
  // controller, learning or not -- a tracking controller with record = true is how the laps an LMPC run loads get
  // written (hawaii_kart_tracking_mpc.param.yaml: learning false, record true, load true)
  namespace rt = lmpc::vehicle_model::racing_trajectory;
  ss_manager_.reset(new rt::SafeSetManager(h_, static_cast<std::size_t>(config_->c.max_lap_stored)));
  ss_recorder_.reset(new rt::SafeSetRecorder(*ss_manager_, config_->record, config_->path_prefix));
  if (config_->c.learning) {
    ss_x_.assign(6 * static_cast<std::size_t>(config_->c.num_ss_pts), 0.0);
    ss_j_.assign(static_cast<std::size_t>(config_->c.num_ss_pts), 0.0);
  }
}

RacingMPC::~RacingMPC() { lmpc_destroy(h_); }

const RacingMPCConfig& RacingMPC::get_config() const { return *config_; }
VehicleModel& RacingMPC::get_model() { return *model_; }
const bool& RacingMPC::solved() const { return solved_; }

void RacingMPC::solve(const DMDict& in, DMDict& out, Dict& stats) {
  const std::size_t N = static_cast<std::size_t>(config_->c.N);
  const double total_length = static_cast<double>(in.at("total_length"));
  const DM& x_ic = in.at("x_ic");
  const DM& u_ic = in.at("u_ic");
  (void)in.at("t_ic");  // consumed by the lap recorder upstream (racing_mpc.cpp:218,246)
  DM X_ref = in.at("X_ref");
  for (std::size_t i = 0; i < N; ++i)  // racing_mpc.cpp:219-223
    X_ref(0, i) = align_abscissa(X_ref(0, i), x_ic(0, 0), total_length);
  DM U_ref = in.at("U_ref");
  const DM& bound_left = in.at("bound_left");
  const DM& bound_right = in.at("bound_right");
  const DM& curvatures = in.at("curvatures");
  const DM& vel_ref = in.at("vel_ref");
  // Warm start (racing_mpc.cpp:287-327).  With the keys: all four are required (upstream reads them with at()), and
  // T_optm_ref replaces T_ref.  Without them upstream restarts from its own previous solution -- and throws when there is
  // none.  Since round 5 the plan is USED by the tracking QP (lmpc_solve_host_warm: an active-set solve on it before any
  // interior point; refused, the cold solve -- the same optimum either way); since round 6 by the learning QP too, when
  // `convex_combi_optm_ref` is among the inputs (racing_mpc.cpp:281: lmpc_solve_host_warm_ss); the sequential-QP solve keeps the
  // contract only: the keys are checked, and a controller that has never run its solver refuses a call without them.
  const bool warm = in.count("X_optm_ref") > 0;
  const DM* X_warm = nullptr;
  const DM* U_warm = nullptr;
  if (warm) {
    X_warm = &in.at("X_optm_ref");
    U_warm = &in.at("U_optm_ref");
    (void)in.at("dU_optm_ref");  // (implied by U_optm_ref and u_ic: u_i = u_{i-1} + t_i dU_i is a row of the QP)
  } else if (!ran_) {
    throw std::runtime_error("No warm start given and no previous solution found.");
  }
  const DM& T = warm ? in.at("T_optm_ref") : in.at("T_ref");
  if (X_ref.rows != 6 || X_ref.cols != N || U_ref.rows != 2 || U_ref.cols != N - 1 || T.data.size() != N - 1 ||
      bound_left.data.size() != N || bound_right.data.size() != N || curvatures.data.size() != N ||
      vel_ref.data.size() != N || x_ic.data.size() != 6 || u_ic.data.size() != 2)
    throw std::length_error("RacingMPC::solve: input dimension does not match MPC dimension");
  const std::size_t S = config_->c.learning ? static_cast<std::size_t>(config_->c.num_ss_pts) : 0;
  {  // racing_mpc.cpp:240-281: load, record and query unconditionally; only the solve's parameters depend on learning
    namespace rt = lmpc::vehicle_model::racing_trajectory;
    if (!ss_loaded_ && config_->load) {
      ss_recorder_->load(config_->load_path, total_length);
      ss_loaded_ = true;
    }
    DM k0(1, 1);
    k0(0, 0) = curvatures.data[0];
    ss_recorder_->step(x_ic, u_ic, k0, in.at("t_ic"), total_length);
    rt::SSQuery q;
    q.x = DM(6, 1);
    for (int r = 0; r < 6; ++r) q.x(r, 0) = X_ref(r, N - 1);
    q.max_num_total = static_cast<std::size_t>(config_->c.num_ss_pts);
    q.max_num_per_lap = static_cast<std::size_t>(config_->c.num_ss_pts_per_lap);
    const rt::SSResult ss = ss_manager_->query(q);
    out["ss_x"] = ss.x;
    out["ss_j"] = ss.J;
    if (S && ss.x.size2() > 0) {  // pad with the last point or truncate; costs relative to the first (racing_mpc.cpp:263-280)
      const std::size_t n = ss.x.size2();
      for (std::size_t j = 0; j < S; ++j) {
        const std::size_t src = j < n ? j : n - 1;
        for (int r = 0; r < 6; ++r) ss_x_[j * 6 + r] = ss.x(r, src);
        ss_j_[j] = ss.J(0, src) - ss.J(0, 0);
      }
    }  // else: "No safe set found, using previous safe set." (racing_mpc.cpp:259-261)
  }
  DM X(6, N), U(2, N - 1), dU(2, N - 1), lam(S, 1);
  int32_t status = 0, iters = 0, total_iters = 0;
  if (full_dynamics_) {
    // The nonlinear-dynamics problem (racing_mpc.cpp:162-166; IPOPT upstream, :67-84): sequential QPs over the same
    // kernels with a line search on the l1 merit function, from the reference trajectory handed in (the node's
    // zero-input rollout, racing_mpc_node.cpp:210-235).  Like IPOPT with error_on_fail, a run that has not converged
    // within its budget is a failure.
    int32_t sqp_iters = 0;
    double move = 0.0, defect = 0.0;
    const int rc = lmpc_solve_full_dynamics_host(h_, x_ic.data.data(), u_ic.data.data(), X_ref.data.data(), U_ref.data.data(),
                                                 T.data.data(), bound_left.data.data(), bound_right.data.data(),
                                                 curvatures.data.data(), vel_ref.data.data(), total_length,
                                                 S ? ss_x_.data() : nullptr, S ? ss_j_.data() : nullptr, 40, 1e-8,
                                                 X.data.data(), U.data.data(), dU.data.data(), S ? lam.data.data() : nullptr,
                                                 &status, &iters, &sqp_iters, &move, &defect);
    if (rc != LMPC_OK) {
      std::cerr << "RacingMPC::solve: " << lmpc_last_error(h_) << '\n';
      return;
    }
    total_iters = iters;
    stats["sqp_iter_count"] = static_cast<double>(sqp_iters);
    stats["dynamics_defect"] = defect;
    if (status == LMPC_SOLVE_OPTIMAL && !(move <= 1e-8)) status = LMPC_SOLVE_MAX_ITER;
  } else {
    // a plan is worth trying once the controller has a solution behind it (the node's first call hands the zero-input rollout)
    // (the learning problem also needs the plan's simplex weights and the set they were solved on)
    const bool have_lam = S && in.count("convex_combi_optm_ref") > 0 && in.at("convex_combi_optm_ref").data.size() == S && ss_x_prev_.size() == 6 * S;
    const bool use_plan = warm && ran_ && (!S || have_lam) && X_warm->rows == 6 && X_warm->cols == N && U_warm->rows == 2 && U_warm->cols == N - 1;
    DM Xw;
    if (use_plan) {  // the same abscissa alignment as X_ref (racing_mpc.cpp:297-301)
      Xw = *X_warm;
      for (std::size_t i = 0; i < N; ++i) Xw(0, i) = align_abscissa(Xw(0, i), x_ic(0, 0), total_length);
    }
    std::vector<double> lam_ref;
    if (use_plan && S) {
      // Upstream hands convex_combi_optm_ref to OSQP by POSITION (racing_mpc.cpp:281), but the query returns its neighbours nearest
      // first, so the same safe-set point sits at another position from one call to the next.  The weights are carried over by the
      // identity of the points: entry j of this call's set takes the weight of the previous call's point with the same coordinates
      // (its first occurrence: the padding repeats the last point); a point that has left the set drops its weight.
      const DM& lp = in.at("convex_combi_optm_ref");
      lam_ref.assign(S, 0.0);
      std::vector<char> used(S, 0);
      for (std::size_t j = 0; j < S; ++j) {
        if (j > 0 && std::equal(ss_x_.begin() + 6 * j, ss_x_.begin() + 6 * j + 6, ss_x_.begin() + 6 * (j - 1))) continue;
        for (std::size_t i = 0; i < S; ++i)
          if (!used[i] && lp.data[i] > 0.0 && std::equal(ss_x_.begin() + 6 * j, ss_x_.begin() + 6 * j + 6, ss_x_prev_.begin() + 6 * i)) {
            lam_ref[j] += lp.data[i];
            used[i] = 1;
          }
      }
    }
    const int rc = (use_plan && S)
        ? lmpc_solve_host_warm_ss(h_, x_ic.data.data(), u_ic.data.data(), X_ref.data.data(), U_ref.data.data(), T.data.data(),
                                  bound_left.data.data(), bound_right.data.data(), curvatures.data.data(), vel_ref.data.data(), total_length,
                                  ss_x_.data(), ss_j_.data(), Xw.data.data(), U_warm->data.data(), lam_ref.data(), X.data.data(), U.data.data(),
                                  dU.data.data(), lam.data.data(), &status, &iters)
        : use_plan
        ? lmpc_solve_host_warm(h_, x_ic.data.data(), u_ic.data.data(), X_ref.data.data(), U_ref.data.data(), T.data.data(),
                               bound_left.data.data(), bound_right.data.data(), curvatures.data.data(), vel_ref.data.data(), total_length,
                               Xw.data.data(), U_warm->data.data(), X.data.data(), U.data.data(), dU.data.data(), &status, &iters)
        : lmpc_solve_host(h_, x_ic.data.data(), u_ic.data.data(), X_ref.data.data(), U_ref.data.data(),
                          T.data.data(), bound_left.data.data(), bound_right.data.data(),
                          curvatures.data.data(), vel_ref.data.data(), total_length, S ? ss_x_.data() : nullptr,
                          S ? ss_j_.data() : nullptr, X.data.data(), U.data.data(), dU.data.data(),
                          S ? lam.data.data() : nullptr, &status, &iters);
    stats["warm_start"] = use_plan ? 1.0 : 0.0;
    if (rc != LMPC_OK) {
      std::cerr << "RacingMPC::solve: " << lmpc_last_error(h_) << '\n';
      return;
    }
    total_iters = iters;
  }
  stats["iter_count"] = static_cast<double>(total_iters);
  if (status != LMPC_SOLVE_OPTIMAL) {
    std::cerr << "RacingMPC::solve: QP " << (status == LMPC_SOLVE_INFEASIBLE ? "infeasible" : "hit the iteration cap")
              << '\n';
    return;  // out lacks X_optm, solved_ unchanged (racing_mpc.cpp:358-371)
  }
  solved_ = true;
  // upstream assigns sol_ from solve_limited() (racing_mpc.cpp:343-345), and its solver is built with error_on_fail = true
  // (:86-103): a failed solve throws before the assignment, so only a successful one leaves a "previous solution" behind
  ran_ = true;
  out["X_optm"] = X;
  out["U_optm"] = U;
  out["dU_optm"] = dU;
  if (S) {
    out["convex_combi_optm"] = lam;
    ss_x_prev_ = ss_x_;  // the set these weights belong to (the next call's convex_combi_optm_ref is matched against it)
  }
}

void RacingMPC::create_warm_start(const DMDict& in, DMDict& out) {
  const std::size_t N = static_cast<std::size_t>(config_->c.N);
  const DM& P0 = in.at("P0");
  const DM& Yaws = in.at("Yaws");
  const DM& Radii = in.at("Radii");
  const double current_vel = static_cast<double>(in.at("current_vel"));
  const double target_vel = static_cast<double>(in.at("target_vel"));
  if (P0.size2() != N) throw std::length_error("create_warm_start: P0 dimension does not match MPC dimension.");
  if (Yaws.size2() != N) throw std::length_error("create_warm_start: Yaws dimension does not match MPC dimension.");
  if (current_vel <= 0.0) throw std::range_error("Current velocity cannot be smaller than or equal to zero.");
  if (target_vel <= 0.0) throw std::range_error("Target velocity cannot be smaller than or equal to zero.");
  DM X_ref(6, N), U_ref(2, N - 1);
  for (std::size_t i = 0; i < N; ++i) {
    X_ref(0, i) = P0(0, i);
    X_ref(1, i) = P0(1, i);
    X_ref(2, i) = Yaws.data[i];
    X_ref(3, i) = current_vel + (target_vel - current_vel) * (N > 1 ? double(i) / double(N - 1) : 0.0);
    X_ref(5, i) = X_ref(3, i) / Radii.data[i];
  }
  for (std::size_t i = 0; i + 1 < N; ++i) {
    const double v0 = X_ref(3, i), v1 = X_ref(3, i + 1);
    const double d = std::hypot(P0(0, i) - P0(0, i + 1), P0(1, i) - P0(1, i + 1));
    const double a = (v1 * v1 - v0 * v0) / (2 * d);
    // upstream writes a force into the 3-control layout (racing_mpc.cpp:413-418); with the simplified
    // longitudinal control one unit of u_lon is 1000 N (single_track_planar_model.cpp:215-216)
    U_ref(0, i) = model_->v.m * a / 1000.0;
    U_ref(1, i) = std::atan(model_->v.l / Radii.data[i]);
  }
  out["X_ref"] = X_ref;
  out["U_ref"] = U_ref;
}

}  // namespace racing_mpc
}  // namespace mpc
}  // namespace lmpc
