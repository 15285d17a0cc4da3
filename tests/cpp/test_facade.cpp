// Exercises the RacingMPC facade the way RacingMPCNode::on_step_timer does (racing_mpc_node.cpp:301-332).
// usage: test_facade <problem.txt>   (problem.txt: written by tests/test_gpu_facade.py from a golden vector)
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>

#include "racing_mpc.hpp"

using namespace lmpc::mpc::racing_mpc;

static DM read_dm(std::ifstream& f) {
  std::size_t r, c;
  f >> r >> c;
  DM m(r, c);
  for (auto& v : m.data) f >> v;
  return m;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::ifstream f(argv[1]);
  int N;
  f >> N;
  auto cfg = std::make_shared<RacingMPCConfig>();
  auto veh = std::make_shared<VehicleModel>();
  const double inf = std::numeric_limits<double>::infinity();
  // BARC vehicle + tracking MPC (param/barc/*.yaml, param/racing_mpc/barc_tracking_mpc.param.yaml)
  lmpc_vehicle& v = veh->v;
  v.m = 2.2187; v.Jzz = 0.02723; v.l = 0.324; v.cg_ratio = 0.5; v.h = 0.07; v.b = 0.281; v.fr = 0.012;
  v.kd = 0.0; v.kb = 0.5; v.cd = 0.0; v.Af = 1.0; v.rho = 1.2; v.cl_f = 0.0; v.cl_r = 0.0; v.mu = 0.9;
  v.Bf = 5.0; v.Cf = 2.28; v.Br = 5.0; v.Cr = 2.28; v.Fd_max = 15.0; v.Fb_max = -15.0; v.Td = 0.1; v.Tb = 0.1;
  v.max_steer = 0.314159; v.max_steer_rate = 10.0;
  lmpc_config& c = cfg->c;
  c.N = N; c.learning = 0; c.num_ss_pts = 96; c.num_ss_pts_per_lap = 32; c.max_lap_stored = 3;
  c.margin = 0.1; c.q_contour = 1.0; c.q_heading = 1.0; c.q_vel = 0.2; c.q_vy = 1e-3; c.q_vyaw = 1e-3; c.q_boundary = 20.0;
  const double R[4] = {0.01, 0, 0, 0.01};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = R[k]; }
  const double xmax[6] = {inf, inf, inf, 6.0, 1.0, 3.0}, xmin[6] = {-inf, -inf, -inf, 0.1, -1.0, -3.0};
  for (int k = 0; k < 6; ++k) { c.x_max[k] = xmax[k]; c.x_min[k] = xmin[k]; c.convex_hull_slack[k] = 20.0; }
  c.u_max[0] = 0.01; c.u_max[1] = 0.33; c.u_min[0] = -0.01; c.u_min[1] = -0.33; c.max_vel_ref_diff = 1.0;

  RacingMPC mpc(cfg, veh);
  DMDict in, out;
  Dict stats;
  for (const char* key : {"x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref"})
    in[key] = read_dm(f);
  in["t_ic"] = DM(0.0);
  in["total_length"] = DM(15.6);
  DM Xe = read_dm(f), Ue = read_dm(f);
  if (mpc.solved()) return 3;
  {  // racing_mpc.cpp:310-313: no warm start and no previous solution -> std::runtime_error
    bool refused = false;
    try { mpc.solve(in, out, stats); } catch (const std::runtime_error&) { refused = true; }
    if (!refused || mpc.solved()) { std::puts("FAIL: a first call without warm start keys must throw"); return 1; }
  }
  {  // a FAILED first solve leaves no previous solution behind either (upstream: error_on_fail = true, solve_limited() throws
     // before sol_ is assigned, racing_mpc.cpp:86-103,343-345): the next call without the keys still throws
    RacingMPC fresh(cfg, veh);
    DMDict bad = in, outb;
    bad["X_optm_ref"] = bad["X_ref"];
    bad["U_optm_ref"] = bad["U_ref"];
    bad["dU_optm_ref"] = DM(2, static_cast<std::size_t>(N) - 1);
    bad["T_optm_ref"] = bad["T_ref"];
    bad["x_ic"](3, 0) = 0.01;  // outside the state box: the QP is infeasible
    fresh.solve(bad, outb, stats);
    if (outb.count("X_optm") || fresh.solved()) { std::puts("FAIL: infeasible first solve reported a solution"); return 1; }
    bool refused = false;
    try { fresh.solve(in, outb, stats); } catch (const std::runtime_error&) { refused = true; }
    if (!refused) { std::puts("FAIL: a failed first solve must not count as a previous solution"); return 1; }
  }
  // the node's first call (racing_mpc_node.cpp:225-234): the reference doubles as the warm start
  in["X_optm_ref"] = in["X_ref"];
  in["U_optm_ref"] = in["U_ref"];
  in["dU_optm_ref"] = DM(2, static_cast<std::size_t>(N) - 1);
  in["T_optm_ref"] = in["T_ref"];
  mpc.solve(in, out, stats);
  if (!out.count("X_optm") || !mpc.solved()) { std::puts("FAIL: no X_optm"); return 1; }
  const double sx[6] = {2000.0, 10.0, 0.1, 80.0, 2.0, 2.0}, su[2] = {10.0, 0.3};
  double ex = 0, eu = 0;
  for (std::size_t i = 0; i < (std::size_t)N; ++i)
    for (int k = 0; k < 6; ++k) ex = std::fmax(ex, std::fabs(out["X_optm"](k, i) - Xe(k, i)) / sx[k]);
  for (std::size_t i = 0; i + 1 < (std::size_t)N; ++i)
    for (int k = 0; k < 2; ++k) eu = std::fmax(eu, std::fabs(out["U_optm"](k, i) - Ue(k, i)) / su[k]);
  std::printf("iter_count %g  err_x %.3e  err_u %.3e\n", stats["iter_count"], ex, eu);
  // failure contract: an initial state outside the box leaves `out` without X_optm
  DMDict in2 = in, out2;
  in2["x_ic"](3, 0) = 0.01;
  mpc.solve(in2, out2, stats);
  if (out2.count("X_optm")) { std::puts("FAIL: infeasible problem returned X_optm"); return 1; }
  // once a solution exists a call without the warm start keys goes through (upstream restarts from sol_)
  {
    DMDict in4 = in, out4;
    for (const char* k : {"X_optm_ref", "U_optm_ref", "dU_optm_ref", "T_optm_ref"}) in4.erase(k);
    mpc.solve(in4, out4, stats);
    if (!out4.count("X_optm")) { std::puts("FAIL: solve without warm start keys after a first solution"); return 1; }
    for (std::size_t e = 0; e < out4["X_optm"].data.size(); ++e)
      if (out4["X_optm"].data[e] != out["X_optm"].data[e]) { std::puts("FAIL: warm start keys changed the answer"); return 1; }
  }
  // missing key throws like DMDict::at
  bool threw = false;
  try { DMDict in3 = in; in3.erase("u_ic"); mpc.solve(in3, out2, stats); } catch (const std::out_of_range&) { threw = true; }
  if (!threw) { std::puts("FAIL: missing key did not throw"); return 1; }
  std::puts((ex < 1e-4 && eu < 1e-4) ? "PASS" : "FAIL: tolerance");
  return (ex < 1e-4 && eu < 1e-4) ? 0 : 1;
}
