// racing_lmpc.hpp -- C++ facade with the class surface of the reference's second controller plugin,
// lmpc::mpc::racing_lmpc::RacingLMPC (src/controllers/racing_lmpc/include/racing_lmpc/racing_lmpc.hpp:37-54;
// constructed and called at src/controllers/racing_lmpc/src/racing_lmpc_node.cpp:193), over the C ABI
// (include/lmpc_hip.h).  Same namespace, class name, method set and DMDict keys:
//   explicit RacingLMPC(config, model); get_config(); solve(in, out, stats); create_warm_start(in, out);
//   get_model(); solved();
// so a caller switches by changing the include and the DM type.
//
// WHAT IS BEHIND IT.  Upstream this class is a *nonlinear* program handed to IPOPT (racing_lmpc.cpp:31-176): variables
// X and U only, the discrete dynamics as equality rows (model_->add_nlp_constraints), the input-rate limits as rows on
// (u_{i+1} - u_i) / t_i, one boundary slack PER KNOT, cost q_contour e_y^2 + q_heading e_psi^2 + q_vel (|v| - v_ref)^2 on the
// knots 1 .. N-1 (terminal knot x 10) + u' R u.  That arithmetic is out of this repository's scope (SURVEY.md section 2 row
// 13); the surface is not (BASELINE.json north_star: "RacingMPC/RacingLMPC C++ plugin surface stay drop-in").  The facade
// therefore FORWARDS to the nonlinear-dynamics path of the batched solver -- lmpc_solve_full_dynamics_host, the sequential-QP
// loop that stands in for IPOPT behind RacingMPC(full_dynamics = true) -- with a configuration that restates this class's
// problem as closely as that path can.  The differences, all in the problem and none in the surface:
//   * input rates are VARIABLES there (dU, u_i = u_{i-1} + t_i dU_i) with a cost weight R_d the reference's problem does not
//     have: the facade sets R_d = 0.1 diag(R).  It is what makes the sequential-QP loop converge -- its QPs carry the cost's
//     Hessian only, and with a vanishing rate weight they zig-zag (racing_lmpc.cpp has the measurement) -- and it makes the plan
//     smoother in the inputs than the reference's optimum;
//   * ONE boundary slack shared by all knots (racing_mpc.cpp:533) instead of one per knot (racing_lmpc.cpp:83-90): identical
//     while at most one knot is outside the tightened boundary, cheaper than the reference's when several are (the shared
//     slack is charged once, q_boundary sigma^2, not once per violating knot);
//   * the speed term is (vx - v_ref)^2, not (hypot(vx, vy) - v_ref)^2 (they differ by vy^2 / 2 vx); q_vy = q_vyaw = 0;
//   * scale_x / scale_u are ones upstream (racing_lmpc.cpp:35-36): results are in physical units either way;
//   * `tol`, `max_cpu_time` are IPOPT's; here `max_iter` caps the number of QPs (clamped to [1, 200]) and the loop stops when
//     the QP's own step falls below 1e-8 (scaled).  A run that stops on the cap returns the iterate reached, as
//     solve_limited() does upstream (racing_lmpc.cpp:254), and counts as solved.
// 3-control configurations (u = [f_drive, f_brake, steer] in newtons, R 3 x 3: the layout of param/sample_mpc.param.yaml) are
// accepted when they can be folded onto the single longitudinal input of the built model (u_lon in kN:
// single_track_planar_model.cpp:215-216): R(0,0) == R(1,1), no coupling between the force rows and steering; U_ref / U_optm
// then travel in that 3-row layout (f_drive = max(u_lon, 0) * 1000, f_brake = min(u_lon, 0) * 1000).  Anything else throws.
#ifndef LMPC_HOST_RACING_LMPC_HPP_
#define LMPC_HOST_RACING_LMPC_HPP_

#include <cstddef>
#include <cstdint>
#include <memory>

#include "dm.hpp"
#include "lmpc_hip.h"
#include "racing_mpc.hpp"  // VehicleModel (the stand-in for BaseVehicleModel::SharedPtr)

namespace lmpc {
namespace mpc {
namespace racing_lmpc {

using lmpc::DM;
using lmpc::DMDict;
using lmpc::Dict;
using lmpc::mpc::racing_mpc::VehicleModel;

enum RacingLMPCStepMode { STEP, CONTINUOUS };  // racing_lmpc_config.hpp:29-33

// racing_lmpc_config.hpp:35-62, field for field and in the same order (ros_param_loader.cpp:56-76 aggregate-initialises it)
struct RacingLMPCConfig {
  typedef std::shared_ptr<RacingLMPCConfig> SharedPtr;
  double max_cpu_time;         // IPOPT's; not enforced here
  int64_t max_iter;            // cap on the QPs of the sequential-QP loop
  double tol;                  // IPOPT's; not used (the loop's step tolerance is fixed, see above)
  std::size_t N;               // knots
  double margin;
  double average_track_width;  // unused upstream as well
  bool verbose;
  RacingLMPCStepMode step_mode = RacingLMPCStepMode::STEP;
  DM q_contour, q_heading, q_vel, q_boundary;  // 1 x 1
  DM R;                                        // 2 x 2, or 3 x 3 in the 3-control layout
  DM x_max, x_min;                             // 6
  DM u_max, u_min;                             // 2, or 3 ([f_drive, f_brake, steer], newtons)
};

class RacingLMPC {
 public:
  typedef std::shared_ptr<RacingLMPC> SharedPtr;
  typedef std::unique_ptr<RacingLMPC> UniquePtr;

  // Throws std::invalid_argument on a configuration that cannot be folded onto the built model, std::runtime_error when the
  // library rejects it.  `device` (not upstream) selects the GPU.
  explicit RacingLMPC(RacingLMPCConfig::SharedPtr mpc_config, VehicleModel::SharedPtr model, int device = 0);
  ~RacingLMPC();
  RacingLMPC(const RacingLMPC&) = delete;
  RacingLMPC& operator=(const RacingLMPC&) = delete;

  const RacingLMPCConfig& get_config() const;

  // racing_lmpc.cpp:183-267.  Keys read: total_length, x_ic, u_ic, X_ref, U_ref, bound_left, bound_right, curvatures, vel_ref,
  // and either X_optm_ref + U_optm_ref + T_optm_ref (the start iterate and the knot spacing) or T_ref (then the previous
  // solution is the start iterate, its abscissa re-aligned to X_ref's; std::runtime_error("No warm start given and no
  // previous solution found.") when there is none, :226-228).  Keys written: X_optm, U_optm -- ALWAYS, also when the
  // solver fails (upstream writes its debug values, :258-265); solved() turns true on the first success.
  // stats: iter_count (interior-point iterations), sqp_iter_count, dynamics_defect, success (1 / 0).
  void solve(const DMDict& in, DMDict& out, Dict& stats);

  // racing_lmpc.cpp:269-330: throws std::length_error / std::range_error on the same conditions.
  void create_warm_start(const DMDict& in, DMDict& out);

  VehicleModel& get_model();
  const bool& solved() const;

 private:
  RacingLMPCConfig::SharedPtr config_;
  VehicleModel::SharedPtr model_;
  lmpc_config c_{};
  bool three_controls_;
  bool solved_;
  bool have_sol_;  // upstream's sol_ != nullptr
  DM sol_X_, sol_U_;  // the previous solution, 2-control layout
  lmpc_handle* h_;
};

}  // namespace racing_lmpc
}  // namespace mpc
}  // namespace lmpc
#endif
