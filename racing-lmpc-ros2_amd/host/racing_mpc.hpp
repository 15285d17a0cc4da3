// racing_mpc.hpp -- C++ facade with the reference's class surface over the C ABI (include/lmpc_hip.h).
//
// Mirrors lmpc::mpc::racing_mpc::RacingMPC (src/mpc/racing_mpc/include/racing_mpc/racing_mpc.hpp:43-62):
//   explicit RacingMPC(config, model, full_dynamics = false); get_config(); solve(in, out, stats);
//   create_warm_start(in, out); get_model(); solved();
// with `in` / `out` string-keyed maps of dense column-major fp64 matrices (`DM`, a minimal stand-in
// for casadi::DM -- CasADi is not a dependency of this library) and the same keys as
// RacingMPC::solve reads and writes (racing_mpc.cpp:215-228, 347-353).  A caller of the reference
// (RacingMPCNode::on_step_timer, racing_mpc_node.cpp:301,320) switches by changing the include and
// the DM type; see INTEGRATION.md.  No HIP types appear here: the facade talks to lmpc_solve_host and, with
// config.learning, keeps the safe set through SafeSetManager / SafeSetRecorder (safe_set.hpp).
#ifndef LMPC_HOST_RACING_MPC_HPP_
#define LMPC_HOST_RACING_MPC_HPP_

#include <cstddef>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "dm.hpp"
#include "lmpc_hip.h"
#include "safe_set.hpp"

namespace lmpc {
namespace mpc {
namespace racing_mpc {

using lmpc::DM;      // column-major dense matrix standing in for casadi::DM (dm.hpp)
using lmpc::DMDict;
using lmpc::Dict;

// RacingMPCConfig (racing_mpc_config.hpp:37-82): the numeric fields live in the C struct.
struct RacingMPCConfig {
  typedef std::shared_ptr<RacingMPCConfig> SharedPtr;
  lmpc_config c{};
  bool verbose = false;
  bool record = false;
  std::string path_prefix;
  bool load = false;
  std::vector<std::string> load_path;
};

// Stand-in for BaseVehicleModel::SharedPtr: the scalars compile_dynamics reads, plus the selector
// name vehicle_model_factory.cpp:31-49 dispatches on.
struct VehicleModel {
  typedef std::shared_ptr<VehicleModel> SharedPtr;
  std::string name = "single_track_planar_model";
  lmpc_vehicle v{};
  std::size_t nx() const { return 6; }
  std::size_t nu() const { return 2; }
};

class RacingMPC {
 public:
  typedef std::shared_ptr<RacingMPC> SharedPtr;
  typedef std::unique_ptr<RacingMPC> UniquePtr;

  // Throws std::runtime_error when the library rejects the configuration (the reference throws
  // from CasADi for the same reasons).  full_dynamics = true solves the nonlinear-dynamics problem
  // (racing_mpc.cpp:162-166, IPOPT upstream) by sequential QPs over the same kernel.
  explicit RacingMPC(RacingMPCConfig::SharedPtr mpc_config, VehicleModel::SharedPtr model,
                     const bool& full_dynamics = false, int device = 0);
  ~RacingMPC();
  RacingMPC(const RacingMPC&) = delete;
  RacingMPC& operator=(const RacingMPC&) = delete;

  const RacingMPCConfig& get_config() const;
  // Same contract as the reference: on solver failure a message goes to std::cerr, `out` lacks
  // "X_optm" and solved() is unchanged (racing_mpc.cpp:343-371); missing keys throw std::out_of_range.  Warm start keys
  // (X_optm_ref, U_optm_ref, dU_optm_ref, T_optm_ref): all or none; none on a controller whose solver has never run throws
  // std::runtime_error("No warm start given and no previous solution found.") as upstream does (:310-313).
  void solve(const DMDict& in, DMDict& out, Dict& stats);
  // racing_mpc.cpp:374-430: throws std::length_error / std::range_error on the same conditions.
  void create_warm_start(const DMDict& in, DMDict& out);
  VehicleModel& get_model();
  const bool& solved() const;

 private:
  RacingMPCConfig::SharedPtr config_;
  VehicleModel::SharedPtr model_;
  bool full_dynamics_;
  bool solved_;
  bool ran_;  // the solver has produced a solution object before (upstream's sol_ != nullptr)
  lmpc_handle* h_;
  // LMPC (config.learning): the safe set and its recorder, as racing_mpc.hpp:88-90 upstream
  std::unique_ptr<lmpc::vehicle_model::racing_trajectory::SafeSetManager> ss_manager_;
  std::unique_ptr<lmpc::vehicle_model::racing_trajectory::SafeSetRecorder> ss_recorder_;
  bool ss_loaded_ = false;
  std::vector<double> ss_x_, ss_j_;  // last non-empty query, padded (the parameter keeps its value upstream)
  std::vector<double> ss_x_prev_;    // the set the last returned convex_combi_optm weighs (warm start: weights carried over by point identity)
};

}  // namespace racing_mpc
}  // namespace mpc
}  // namespace lmpc
#endif
