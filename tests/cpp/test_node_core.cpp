// Drives RacingMPCNodeCore -- the ROS-free restatement of RacingMPCNode::on_step_timer (racing_mpc_node.cpp:150-477) --
// in closed loop with a host plant on the reference's BARC race line, the way sim_barc_tracking_mpc wires node and
// simulator: state message (global pose + body velocities) in, actuation message out, every 25 ms.
// usage: test_node_core <15_barc_optm.txt> <N> <laps> [step|continuous] [dump_file dump_steps]
// With a dump file, the first dump_steps ticks are written out in full -- the state and actuation messages that went in,
// the plan the node held before the tick, the sol_in it handed to the controller, the plan and the actuation that came
// out -- for the step-for-step comparison with oracle/node_step.py (tests/test_gpu_facade.py).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "racing_mpc_node_core.hpp"
#include "single_track_model.hpp"

using namespace lmpc::mpc::racing_mpc;
namespace rt = lmpc::vehicle_model::racing_trajectory;
namespace stm = lmpc::vehicle_model::single_track_planar_model;

static void dump(std::FILE* f, const char* key, const lmpc::DM& m) {
  std::fprintf(f, "%s %zu %zu", key, m.rows, m.cols);
  for (double v : m.data) std::fprintf(f, " %.17g", v);
  std::fputc('\n', f);
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  std::FILE* df = argc > 6 ? std::fopen(argv[5], "w") : nullptr;
  const int dump_steps = argc > 6 ? std::atoi(argv[6]) : 0;
  const int N = std::atoi(argv[2]);
  const double laps_wanted = std::atof(argv[3]);
  const bool step_mode = argc > 4 && std::strcmp(argv[4], "step") == 0;
  auto track = std::make_shared<rt::RacingTrajectory>(std::string(argv[1]));
  auto cfg = std::make_shared<RacingMPCConfig>();
  auto veh = std::make_shared<VehicleModel>();
  const double inf = std::numeric_limits<double>::infinity();
  lmpc_vehicle& v = veh->v;  // param/barc/*.yaml
  v.m = 2.2187; v.Jzz = 0.02723; v.l = 0.324; v.cg_ratio = 0.5; v.h = 0.07; v.b = 0.281; v.fr = 0.012;
  v.kd = 0.0; v.kb = 0.5; v.cd = 0.0; v.Af = 1.0; v.rho = 1.2; v.cl_f = 0.0; v.cl_r = 0.0; v.mu = 0.9;
  v.Bf = 5.0; v.Cf = 2.28; v.Br = 5.0; v.Cr = 2.28; v.Fd_max = 15.0; v.Fb_max = -15.0; v.Td = 0.1; v.Tb = 0.1;
  v.max_steer = 0.314159; v.max_steer_rate = 10.0;
  lmpc_config& c = cfg->c;   // param/racing_mpc/barc_tracking_mpc.param.yaml
  c.N = N; c.learning = 0; c.num_ss_pts = 96; c.num_ss_pts_per_lap = 32; c.max_lap_stored = 3;
  c.margin = 0.1; c.q_contour = 1.0; c.q_heading = 1.0; c.q_vel = 0.2; c.q_vy = 1e-3; c.q_vyaw = 1e-3; c.q_boundary = 20.0;
  const double R[4] = {0.01, 0, 0, 0.01};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = R[k]; }
  const double xmax[6] = {inf, inf, inf, 6.0, 1.0, 3.0}, xmin[6] = {-inf, -inf, -inf, 0.1, -1.0, -3.0};
  for (int k = 0; k < 6; ++k) { c.x_max[k] = xmax[k]; c.x_min[k] = xmin[k]; c.convex_hull_slack[k] = 20.0; }
  c.u_max[0] = 0.01; c.u_max[1] = 0.33; c.u_min[0] = -0.01; c.u_min[1] = -0.33; c.max_vel_ref_diff = 1.0;

  auto mpc = std::make_shared<RacingMPC>(cfg, veh, false);
  auto mpc_full = std::make_shared<RacingMPC>(cfg, veh, true);
  const double dt = 0.025;
  RacingMPCNodeCore node(mpc, mpc_full, track, dt, step_mode ? RacingMPCStepMode::STEP : RacingMPCStepMode::CONTINUOUS, 0, true);
  node.set_speed_scale(0.9);  // velocity_profile_scale of sim_barc_tracking_mpc.launch.py

  // plant: the same model stepped at 10 ms + 10 ms + 5 ms per control period, state kept in the Frenet frame
  const double L = track->total_length();
  double x[6] = {0.5, 0.02, 0.0, 1.5, 0.0, 0.0};
  VehicleActuation act;
  MPCTelemetry tel;
  int n_published = 0, n_failed = 0, n_initial = 0, n_discarded = 0, n_diag = 0;
  double iters_mean_sum = 0.0, iters_max_sum = 0.0, iters_min_sum = 0.0;
  bool changed = false;
  double travelled = 0.0, worst_excess = -1e9, solve_ms = 0.0, t = 0.0;
  const int max_steps = (int)(laps_wanted * L / 1.0 / dt);  // (bounded: at least 1 m/s average)
  for (int k = 0; k < max_steps && travelled < laps_wanted * L; ++k) {
    lmpc::FrenetPose2D fp;
    fp.position.s = x[0]; fp.position.t = x[1]; fp.yaw = x[2];
    lmpc::Pose2D gp;
    track->frenet_to_global(fp, gp);
    VehicleState st;
    st.t = t; st.x = gp.position.x; st.y = gp.position.y; st.psi = gp.yaw;
    st.v_long = x[3]; st.v_tran = x[4]; st.w_psi = x[5];
    const bool dumping = df && k < dump_steps;
    if (dumping) {
      std::fprintf(df, "tick %d\nstate 7 1 %.17g %.17g %.17g %.17g %.17g %.17g %.17g\nfrenet 3 1 %.17g %.17g %.17g\nact_in 2 1 %.17g %.17g\n", k, st.t,
                   st.x, st.y, st.psi, st.v_long, st.v_tran, st.w_psi, x[0], x[1], x[2], act.u_a, act.u_steer);
      dump(df, "prev_X", node.last_x());
      dump(df, "prev_U", node.last_u());
      dump(df, "prev_dU", node.last_du());
    }
    const auto r = node.step(st, act, tel);
    if (dumping) {
      std::fprintf(df, "result 1 1 %d\n", (int)r);
      for (const auto& kv : node.sol_in()) dump(df, ("in_" + kv.first).c_str(), kv.second);
      dump(df, "X_optm", node.last_x());
      dump(df, "U_optm", node.last_u());
      std::fprintf(df, "act_out 2 1 %.17g %.17g\n", act.u_a, act.u_steer);
    }
    if (r == RacingMPCNodeCore::Result::INITIAL_SOLVE) ++n_initial;
    if (r == RacingMPCNodeCore::Result::INITIAL_SOLVE_FAILED) { std::puts("FAIL: initial full-dynamics solve"); return 1; }
    if (r == RacingMPCNodeCore::Result::JIT_DISCARDED) ++n_discarded;
    if (r == RacingMPCNodeCore::Result::PUBLISHED) {
      ++n_published;
      // racing_mpc_node.cpp:370-384: a diagnostics array after every 10th published step, and only then
      DiagnosticArray da;
      const bool got = node.take_diagnostics(da);
      if (got != (n_published % 10 == 0)) { std::puts("FAIL: diagnostics cadence"); return 1; }
      if (got) {
        ++n_diag;
        const bool shape = da.status.size() == 2 && da.status[0].name == "Racing MPC Solve Time" && da.status[0].message == "(ms)" &&
                           da.status[1].name == "Racing MPC Iteration Count" && da.status[0].values.size() == 3 &&
                           da.status[0].values[0].first == "max" && da.status[0].values[1].first == "mean" && da.status[0].values[2].first == "min";
        if (!shape) { std::puts("FAIL: diagnostics content"); return 1; }
        const double mx = std::atof(da.status[0].values[0].second.c_str()), mean = std::atof(da.status[0].values[1].second.c_str()),
                     mn = std::atof(da.status[0].values[2].second.c_str()), it_max = std::atof(da.status[1].values[0].second.c_str());
        if (!(mn <= mean && mean <= mx && mx > 0.0) || (da.status[0].level == DiagnosticStatus::WARN) != (mx > dt * 1e3) ||
            (da.status[1].level == DiagnosticStatus::WARN) != (it_max > 50) || it_max < 1) { std::puts("FAIL: diagnostics levels"); return 1; }
        if (node.take_diagnostics(da)) { std::puts("FAIL: diagnostics handed out twice"); return 1; }
        iters_mean_sum += std::atof(da.status[1].values[1].second.c_str());  // (mean of the window of 10 this array reports)
        iters_max_sum += it_max;
        iters_min_sum += std::atof(da.status[1].values[2].second.c_str());
      }
      // racing_mpc_node.cpp:509-571, once, mid-run: the same race line loaded again is another reference-line object -- the plan
      // goes old Frenet -> global -> new Frenet and must come back where it was (abscissa modulo the lap), and the run goes on
      if (n_published == 57 && !df) {
        const lmpc::DM before = node.last_x();
        auto again = std::make_shared<rt::RacingTrajectory>(std::string(argv[1]));
        node.change_trajectory(again);
        node.change_trajectory(nullptr);  // ignored
        const lmpc::DM& after = node.last_x();
        double worst = 0.0;
        for (std::size_t i = 0; i < before.cols; ++i) {
          double ds = std::fabs(after(0, i) - before(0, i));
          ds = std::fmin(ds, std::fabs(ds - L));
          worst = std::fmax(worst, std::fmax(ds, std::fmax(std::fabs(after(1, i) - before(1, i)), std::fabs(after(2, i) - before(2, i)))));
          for (int q = 3; q < 6; ++q) if (after(q, i) != before(q, i)) worst = 1.0;
        }
        std::printf("change_trajectory: plan moved by %.2e\n", worst);
        if (!(worst < 1e-6)) { std::puts("FAIL: change_trajectory round trip"); return 1; }
        changed = true;
      }
      n_failed += tel.solved ? 0 : 1;
      solve_ms += tel.solve_time;
      if ((int)tel.state.size() != 6 * N || (int)tel.control.size() != 2 * (N - 1)) { std::puts("FAIL: telemetry sizes"); return 1; }
    }
    // the simulator applies the actuation message through from_base_control (racing_simulator.cpp:97-112)
    const double ub[3] = {act.u_a > 0 ? act.u_a : 0.0, act.u_a < 0 ? act.u_a : 0.0, act.u_steer};
    double u[2];
    stm::from_base_control(ub, u);
    const double sub[3] = {0.01, 0.01, 0.005};
    for (double h : sub) {
      double xn[6];
      stm::discrete_dynamics(v, x, u, track->curvature_interpolation(x[0]), h, xn);
      travelled += xn[0] - x[0];
      for (int q = 0; q < 6; ++q) x[q] = xn[q];
      if (x[0] > L) x[0] -= L;
    }
    const double exc = std::fmax(x[1] + v.b / 2 - track->left_boundary_interpolation(x[0]),
                                 track->right_boundary_interpolation(x[0]) - (x[1] - v.b / 2));
    worst_excess = std::fmax(worst_excess, exc);
    t += dt;
    if (!std::isfinite(x[3])) { std::puts("FAIL: plant state not finite"); return 1; }
  }
  std::printf("laps %.3f time %.3f published %d failed %d initial %d discarded %d worst_excess %.4f mean_step_ms %.3f mean_iters %.2f (windows of 10: min %.2f max %.2f)\n",
              travelled / L, t, n_published, n_failed, n_initial, n_discarded, worst_excess, solve_ms / (n_published ? n_published : 1),
              n_diag ? iters_mean_sum / n_diag : 0.0, n_diag ? iters_min_sum / n_diag : 0.0, n_diag ? iters_max_sum / n_diag : 0.0);
  const bool ok = travelled >= laps_wanted * L && n_initial == 1 && n_discarded == 1 && n_failed <= n_published / 100 && worst_excess < 0.02 &&
                  n_diag == n_published / 10 && (changed || df || n_published < 57);
  if (df) std::fclose(df);
  std::puts(ok ? "PASS" : "FAIL");
  return ok ? 0 : 1;
}
