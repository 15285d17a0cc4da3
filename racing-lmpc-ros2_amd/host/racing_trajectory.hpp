// racing_trajectory.hpp -- track tables and their interpolants with the reference's class surface
// (src/vehicle_dynamics_models/racing_trajectory/include/racing_trajectory/racing_trajectory.hpp:37-120,
//  src/racing_trajectory.cpp:25-236): 17-column waypoint table, periodic padding (+4 / -3 waypoints), interpolating
// cubic splines with not-a-knot ends (what casadi::interpolant("bspline") fits), yaw, the curvature expression as the
// reference writes it, Frenet <-> global.  Plain C++17, no dependency; racing_trajectory.py is the same thing in
// Python and tests/test_racing_trajectory.py checks both against an independent implementation of the spline.
#ifndef LMPC_HOST_RACING_TRAJECTORY_HPP_
#define LMPC_HOST_RACING_TRAJECTORY_HPP_

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "dm.hpp"

namespace lmpc {

// lmpc_utils/include/lmpc_utils/primitives.hpp: the poses RacingTrajectory converts between
struct Position2D {
  double x = 0.0, y = 0.0;
};
struct Pose2D {
  Position2D position;
  double yaw = 0.0;
};
struct FrenetPosition2D {
  double s = 0.0, t = 0.0;
};
struct FrenetPose2D {
  FrenetPosition2D position;
  double yaw = 0.0;
};

namespace vehicle_model {
namespace racing_trajectory {

enum TrajectoryIndex : uint8_t {  // racing_trajectory.hpp:37-56
  PX = 0, PY = 1, PZ = 2, YAW = 3, SPEED = 4, CURVATURE = 5, DIST_TO_SF_BWD = 6, DIST_TO_SF_FWD = 7, REGION = 8,
  LEFT_BOUND_X = 9, LEFT_BOUND_Y = 10, RIGHT_BOUND_X = 11, RIGHT_BOUND_Y = 12, BANK = 13, LON_ACC = 14, LAT_ACC = 15,
  TIME = 16
};

// Interpolating C2 cubic through (x_i, y_i), not-a-knot end conditions, evaluated piecewise (value, 1st, 2nd derivative)
class NotAKnotCubic {
 public:
  NotAKnotCubic() = default;
  NotAKnotCubic(const std::vector<double>& x, const std::vector<double>& y);
  double operator()(double xq, int nu = 0) const;

 private:
  std::vector<double> x_, a_, b_, c_, d_;
};

class RacingTrajectory {
 public:
  typedef std::shared_ptr<RacingTrajectory> SharedPtr;
  typedef std::unique_ptr<RacingTrajectory> UniquePtr;

  explicit RacingTrajectory(const DM& traj);               // 17 x n, one waypoint per column (the reference's DM)
  explicit RacingTrajectory(const std::string& file_name);  // whitespace table, one waypoint per row

  void frenet_to_global(const FrenetPose2D& frenet_pose, Pose2D& global_pose) const;
  // initialize_with_previous: start the projection from the abscissa already in frenet_pose (racing_trajectory.cpp:204-212)
  void global_to_frenet(const Pose2D& global_pose, FrenetPose2D& frenet_pose, const bool& initialize_with_previous = false) const;

  // the reference hands out casadi::Function objects; these are the same maps as plain calls
  double x_interpolation(double s) const;
  double y_interpolation(double s) const;
  double yaw_interpolation(double s) const;
  double curvature_interpolation(double s) const;  // x' y'' - y' x'' / |r'|^3, as written (:108-110)
  double left_boundary_interpolation(double s) const;
  double right_boundary_interpolation(double s) const;
  double velocity_interpolation(double s) const;
  const double& total_length() const { return total_length_; }

  // uniform periodic tables for the device (lmpc_track): curvature, bound_left, bound_right, vel at s_j = j L / M
  void to_track_table(std::size_t M, std::vector<double>& curvature, std::vector<double>& bound_left,
                      std::vector<double>& bound_right, std::vector<double>& vel) const;

 private:
  double mod(double s) const;
  DM traj_;
  double total_length_ = 0.0;
  NotAKnotCubic x_, y_, vel_, left_, right_;
};

}  // namespace racing_trajectory
}  // namespace vehicle_model
}  // namespace lmpc
#endif
