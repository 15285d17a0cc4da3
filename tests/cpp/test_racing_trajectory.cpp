// Prints the C++ RacingTrajectory's interpolants and a Frenet round trip for tests/test_racing_trajectory.py.
// usage: test_racing_trajectory <track file> <n samples>
#include <cstdio>
#include <cstdlib>

#include "racing_trajectory.hpp"

using lmpc::vehicle_model::racing_trajectory::RacingTrajectory;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  RacingTrajectory tr(argv[1]);
  const int n = std::atoi(argv[2]);
  const double L = tr.total_length();
  std::printf("%.17g\n", L);
  for (int i = 0; i < n; ++i) {
    const double s = -3.0 + (2.5 * L + 3.0) * i / (n - 1);
    std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", s, tr.x_interpolation(s), tr.y_interpolation(s),
                tr.velocity_interpolation(s), tr.left_boundary_interpolation(s), tr.right_boundary_interpolation(s),
                tr.yaw_interpolation(s), tr.curvature_interpolation(s));
  }
  // Frenet -> global -> Frenet
  lmpc::FrenetPose2D f{{3.7, 0.12}, 0.2}, back;
  lmpc::Pose2D g;
  tr.frenet_to_global(f, g);
  tr.global_to_frenet(g, back);
  std::printf("%.17g %.17g %.17g\n", back.position.s - f.position.s, back.position.t - f.position.t, back.yaw - f.yaw);
  return 0;
}
