// racing_mpc_node_core.hpp -- the controller node's per-step logic without ROS 2.
//
// Mirrors RacingMPCNode::on_step_timer (src/mpc/racing_mpc/src/racing_mpc_node.cpp:150-477) for one car, message in /
// message out, over the RacingMPC facade (racing_mpc.hpp) and RacingTrajectory (racing_trajectory.hpp):
//   state message  -> global-to-Frenet projection (:181-185), from_base_control of the last actuation (:191-202)
//   first call     -> zero-input rollout U = 1e-9 with the model's discrete dynamics and the track curvature at each knot
//                     (:210-235), the full-dynamics controller solves it (:299-314), nothing is published
//   later calls    -> x_ic (CONTINUOUS: one model step ahead with the input about to be applied, :238-240; STEP: as
//                     measured), shift of the last plan with the last input repeated and the last state rolled out
//                     (:245-249), references at the plan's abscissae and the velocity-reference clamp (:261-292), the
//                     QP solve (:320), keep the old plan on failure (:322-332), discard the first QP solve when
//                     `jit` (:337-342), to_base_control of column delay_step -> actuation (:395-413), telemetry (:333-336)
//   every 10th published step -> the diagnostics array (:351-384); change_trajectory (:509-571) re-expresses the plan on
//                     another reference line
// What a rclcpp wrapper adds is subscriptions, publishers, the timer and the visualisation topics (no ROS 2 in this image).
#ifndef LMPC_HOST_RACING_MPC_NODE_CORE_HPP_
#define LMPC_HOST_RACING_MPC_NODE_CORE_HPP_

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "racing_mpc.hpp"
#include "racing_trajectory.hpp"

namespace lmpc {
namespace mpc {
namespace racing_mpc {

enum class RacingMPCStepMode { STEP, CONTINUOUS };  // racing_mpc_config.hpp

// mpclab_msgs/VehicleStateMsg: the fields on_step_timer reads
struct VehicleState {
  double t = 0.0;
  double x = 0.0, y = 0.0, psi = 0.0;              // global pose (x.x, x.y, e.psi)
  double v_long = 0.0, v_tran = 0.0, w_psi = 0.0;  // body velocities
};
// mpclab_msgs/VehicleActuationMsg
struct VehicleActuation {
  double u_a = 0.0, u_steer = 0.0;
};
// lmpc_msgs/MPCTelemetry.msg
struct MPCTelemetry {
  int trajectory_index = 0;
  bool solved = false;
  double cost = 0.0, cost_trajectory = 0.0;  // (never filled upstream either)
  std::vector<double> state, control;        // column-major predictions
  double solve_time = 0.0;                   // ms
};

// diagnostic_msgs/DiagnosticStatus as the node fills it (lmpc_utils/cycle_profiler.hpp:37-67): max / mean / min of a window
// as strings, WARN when the window's maximum exceeds the threshold
struct DiagnosticStatus {
  enum Level { OK = 0, WARN = 1 };
  int level = OK;
  std::string name, message;
  std::vector<std::pair<std::string, std::string>> values;  // ("max", ...), ("mean", ...), ("min", ...)
};
// diagnostic_msgs/DiagnosticArray: what the node publishes every `window` published steps (racing_mpc_node.cpp:370-384)
struct DiagnosticArray {
  double stamp = 0.0;  // the state message's time (the node stamps with its clock)
  std::vector<DiagnosticStatus> status;
};

// lmpc::utils::CycleProfiler<double> (lmpc_utils/cycle_profiler.hpp:70-133): the last `window` samples, their max / mean / min
class CycleWindow {
 public:
  explicit CycleWindow(std::size_t window) : buf_(window), n_(0), next_(0) {}
  void add(double v) {
    if (buf_.empty()) return;
    buf_[next_] = v;
    next_ = (next_ + 1) % buf_.size();
    if (n_ < buf_.size()) ++n_;
  }
  std::size_t capacity() const { return buf_.size(); }
  DiagnosticStatus status(const std::string& name, const std::string& message, double warn_threshold) const;

 private:
  std::vector<double> buf_;
  std::size_t n_, next_;
};

class RacingMPCNodeCore {
 public:
  enum class Result { INITIAL_SOLVE, INITIAL_SOLVE_FAILED, JIT_DISCARDED, PUBLISHED };

  // mpc: the QP controller; mpc_full: RacingMPC(config, model, full_dynamics = true) (racing_mpc_node.cpp:52-56)
  RacingMPCNodeCore(RacingMPC::SharedPtr mpc, RacingMPC::SharedPtr mpc_full,
                    lmpc::vehicle_model::racing_trajectory::RacingTrajectory::SharedPtr track, double dt,
                    RacingMPCStepMode step_mode = RacingMPCStepMode::CONTINUOUS, int delay_step = 0, bool jit = false);

  // one timer tick: `actuation` carries the last published actuation in and the new one out (only when PUBLISHED)
  Result step(const VehicleState& state, VehicleActuation& actuation, MPCTelemetry& telemetry);

  // Diagnostics (racing_mpc_node.cpp:47-48,351-384): solve time (ms) and iteration count of every published step go into two
  // windows of 10; after every 10th published step `diagnostics` holds a fresh array -- "Racing MPC Solve Time" (WARN above the
  // control period dt * 1e3 ms) and "Racing MPC Iteration Count" (WARN above 50) -- and the function returns true once.
  bool take_diagnostics(DiagnosticArray& diagnostics);

  // RacingMPCNode::change_trajectory (racing_mpc_node.cpp:509-571): switch to another reference line.  The previous plan's
  // poses go old Frenet -> global -> new Frenet (only once the QP controller has solved, as upstream), total_length and the
  // curvature under the model step follow the new track.  The measured state needs no conversion here: step() projects the
  // global pose of every state message itself (:181-185).  A null track is ignored (upstream: unknown index).
  void change_trajectory(lmpc::vehicle_model::racing_trajectory::RacingTrajectory::SharedPtr new_track);

  void set_speed_limit(const double& speed_limit);  // racing_mpc_node.cpp:571-581
  void set_speed_scale(const double& speed_scale);  // :583-598 (out of (0, 1] resets to 0.2)
  const DM& last_x() const { return last_x_; }
  const DM& last_u() const { return last_u_; }
  const DM& last_du() const { return last_du_; }
  const DMDict& sol_in() const { return sol_in_; }  // what the last step handed to the controller (step-level parity tests)

 private:
  void discrete_dynamics(const double* x, const double* u, double* xn) const;  // model step with the track curvature at x[s]
  RacingMPC::SharedPtr mpc_, mpc_full_;
  lmpc::vehicle_model::racing_trajectory::RacingTrajectory::SharedPtr track_;
  double dt_;
  RacingMPCStepMode step_mode_;
  int delay_step_;
  bool jitted_;
  double speed_limit_, speed_scale_ = 1.0;
  DM last_x_, last_u_, last_du_, last_convex_combi_;
  DMDict sol_in_;
  CycleWindow profiler_{10}, profiler_iter_count_{10};  // racing_mpc_node.cpp:47-48
  std::size_t profile_step_count_ = 0;
  bool diagnostics_ready_ = false;
  DiagnosticArray diagnostics_;
};

}  // namespace racing_mpc
}  // namespace mpc
}  // namespace lmpc
#endif
