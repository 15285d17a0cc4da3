// safe_set.hpp -- host side of the LMPC safe set with the reference's class surface
// (src/vehicle_dynamics_models/racing_trajectory/include/racing_trajectory/safe_set.hpp:112-160):
//   SafeSetManager   add_lap(x, u, k, t, L), query(SSQuery)          safe_set.cpp:139-180
//   SafeSetRecorder  load(files, L), step(x, u, k, t, L)             safe_set.cpp:246-322
// The manager keeps the laps on the host (a ring of max_lap_stored, as the boost::circular_buffer upstream) and
// mirrors them to the device store of the C-ABI handle; the k-nearest-neighbour query runs on the GPU
// (lmpc_ss_query_host -> lmpc_ss_query_kernel).  Lap files are the reference's: `<prefix>_{x,u,k,t}.txt`, whitespace
// text, one sample per row (DM::to_file / from_file(..., "txt") of the transpose, safe_set.cpp:266-269,302-305).
#ifndef LMPC_HOST_SAFE_SET_HPP_
#define LMPC_HOST_SAFE_SET_HPP_

#include <deque>
#include <string>
#include <vector>

#include "dm.hpp"
#include "lmpc_hip.h"

namespace lmpc {
namespace vehicle_model {
namespace racing_trajectory {

// SSQuery / SSResult (safe_set.hpp:35-55); dist_max is carried and ignored, as upstream (safe_set.cpp:42-54)
struct SSQuery {
  DM x;
  double dist_max = 1.0;
  std::size_t max_num_total = 0;
  std::size_t max_num_per_lap = 0;
};
struct SSResult {
  DM x;  // 6 x n_found (nearest first per lap, newest lap first)
  DM J;  // 1 x n_found, steps to finish
};

DM read_txt(const std::string& path);                   // DM::from_file(path, "txt"): one row per line
void write_txt(const DM& m, const std::string& path);   // DM::to_file(path, "txt")

class SafeSetManager {
 public:
  SafeSetManager(lmpc_handle* handle, const std::size_t& max_lap_stored);
  // throws std::runtime_error when the device store rejects the lap
  void add_lap(const DM& x, const DM& u, const DM& k, const DM& t, const double& total_length);
  // needs max_num_total / max_num_per_lap equal to the handle's num_ss_pts / num_ss_pts_per_lap (they size the kernel)
  SSResult query(const SSQuery& query);
  std::size_t size() const { return laps_.size(); }

 private:
  struct Lap {
    DM x, u, k, t;
  };
  lmpc_handle* h_;
  std::size_t max_lap_stored_;
  std::deque<Lap> laps_;
};

class SafeSetRecorder {
 public:
  SafeSetRecorder(SafeSetManager& manager, const bool& to_file, const std::string& file_prefix);
  void load(const std::vector<std::string>& from_files, const double& total_length);
  void step(const DM& x, const DM& u, const DM& k, const DM& t, const double& total_length);
  std::size_t lap_count() const { return lap_count_; }

 private:
  // the lap being driven: one column per control period (state, input, curvature, time stamp)
  struct Lap {
    DM x, u, k, t;
    void restart(const DM& x0, const DM& u0, const DM& k0, const DM& t0);
    void push(const DM& xi, const DM& ui, const DM& ki, const DM& ti);
    double last_abscissa() const { return x(0, x.cols - 1); }
  };
  void finish_lap(double total_length);  // hand the driven lap to the manager (and to disk)

  SafeSetManager& manager_;
  Lap lap_;
  bool seeded_ = false;      // a first sample has been seen
  bool past_first_line_ = false;  // the start line has been crossed once: laps from here on are complete
  bool to_file_;
  std::string file_prefix_;
  std::size_t lap_count_ = 0;
};

}  // namespace racing_trajectory
}  // namespace vehicle_model
}  // namespace lmpc
#endif
