// LMPC through the RacingMPC facade: safe set loaded from lap files, recorder fed by solve(), terminal set from the
// device query (racing_mpc.cpp:240-281,484-504).
// usage: test_facade_lmpc <problem.txt> <lap prefix 1> <lap prefix 2> <lap prefix 3> <record prefix>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <limits>

#include "racing_mpc.hpp"

using namespace lmpc::mpc::racing_mpc;
namespace rt = lmpc::vehicle_model::racing_trajectory;

static DM read_dm(std::ifstream& f) {
  std::size_t r, c;
  f >> r >> c;
  DM m(r, c);
  for (auto& v : m.data) f >> v;
  return m;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  std::ifstream f(argv[1]);
  int N;
  double L;
  f >> N >> L;
  auto cfg = std::make_shared<RacingMPCConfig>();
  auto veh = std::make_shared<VehicleModel>();
  const double inf = std::numeric_limits<double>::infinity();
  lmpc_vehicle& v = veh->v;  // param/barc/*.yaml
  v.m = 2.2187; v.Jzz = 0.02723; v.l = 0.324; v.cg_ratio = 0.5; v.h = 0.07; v.b = 0.281; v.fr = 0.012;
  v.kd = 0.0; v.kb = 0.5; v.cd = 0.0; v.Af = 1.0; v.rho = 1.2; v.cl_f = 0.0; v.cl_r = 0.0; v.mu = 0.9;
  v.Bf = 5.0; v.Cf = 2.28; v.Br = 5.0; v.Cr = 2.28; v.Fd_max = 15.0; v.Fb_max = -15.0; v.Td = 0.1; v.Tb = 0.1;
  v.max_steer = 0.314159; v.max_steer_rate = 10.0;
  lmpc_config& c = cfg->c;  // param/racing_mpc/barc_lmpc.param.yaml
  c.N = N; c.learning = 1; c.num_ss_pts = 96; c.num_ss_pts_per_lap = 32; c.max_lap_stored = 3;
  c.margin = 0.1; c.q_contour = 1.0; c.q_heading = 1.0; c.q_vel = 0.2; c.q_vy = 1e-3; c.q_vyaw = 1e-3; c.q_boundary = 1000.0;
  const double R[4] = {0.1, 0, 0, 0.1};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = R[k]; }
  const double xmax[6] = {inf, inf, inf, 3.0, 1.0, 3.0}, xmin[6] = {-inf, -inf, -inf, 0.1, -1.0, -3.0};
  const double chs[6] = {40.0, 40.0, 4.0, 40.0, 40.0, 4.0};
  for (int k = 0; k < 6; ++k) { c.x_max[k] = xmax[k]; c.x_min[k] = xmin[k]; c.convex_hull_slack[k] = chs[k]; }
  c.u_max[0] = 0.01; c.u_max[1] = 0.33; c.u_min[0] = -0.01; c.u_min[1] = -0.33; c.max_vel_ref_diff = 1.0;
  cfg->load = true;
  cfg->load_path = {argv[2], argv[3], argv[4]};
  cfg->record = true;
  cfg->path_prefix = argv[5];

  RacingMPC mpc(cfg, veh);
  DMDict in, out;
  Dict stats;
  for (const char* key : {"x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref"})
    in[key] = read_dm(f);
  in["t_ic"] = DM(0.0);
  in["total_length"] = DM(L);
  DM Xe = read_dm(f), Ue = read_dm(f), ssx = read_dm(f), ssj = read_dm(f);
  // warm start keys as the node hands them (racing_mpc_node.cpp:225-234); without them a first call throws
  in["X_optm_ref"] = in["X_ref"];
  in["U_optm_ref"] = in["U_ref"];
  in["dU_optm_ref"] = DM(2, in["U_ref"].cols);
  in["T_optm_ref"] = in["T_ref"];
  mpc.solve(in, out, stats);
  if (!out.count("X_optm") || !out.count("convex_combi_optm") || !out.count("ss_x")) { std::puts("FAIL: outputs missing"); return 1; }
  // the safe set the facade found = the oracle's (points identical, costs up to the J[0] offset the solver removes)
  if (out["ss_x"].size2() != ssx.size2()) { std::printf("FAIL: %zu safe-set points, expected %zu\n", out["ss_x"].size2(), ssx.size2()); return 1; }
  double es = 0, ej = 0;
  for (std::size_t j = 0; j < ssx.size2(); ++j) {
    for (int k = 0; k < 6; ++k) es = std::fmax(es, std::fabs(out["ss_x"](k, j) - ssx(k, j)));
    ej = std::fmax(ej, std::fabs((out["ss_j"](0, j) - out["ss_j"](0, 0)) - ssj(0, j)));
  }
  const double sx[6] = {2000.0, 10.0, 0.1, 80.0, 2.0, 2.0}, su[2] = {10.0, 0.3};
  double ex = 0, eu = 0, sl = 0;
  for (std::size_t i = 0; i < (std::size_t)N; ++i)
    for (int k = 0; k < 6; ++k) ex = std::fmax(ex, std::fabs(out["X_optm"](k, i) - Xe(k, i)) / sx[k]);
  for (std::size_t i = 0; i + 1 < (std::size_t)N; ++i)
    for (int k = 0; k < 2; ++k) eu = std::fmax(eu, std::fabs(out["U_optm"](k, i) - Ue(k, i)) / su[k]);
  for (double w : out["convex_combi_optm"].data) sl += w;
  std::printf("iter_count %g  ss_x %.1e  ss_j %.1e  err_x %.3e  err_u %.3e  sum lambda %.12f\n", stats["iter_count"], es, ej, ex, eu, sl);
  bool ok = es == 0.0 && ej == 0.0 && ex < 1e-6 && eu < 1e-6 && std::fabs(sl - 1.0) < 1e-9;

  // the warm start of the learning controller (round 6; racing_mpc.cpp:281, 293-305): the same problem again with the solution as the
  // plan and its simplex weights as convex_combi_optm_ref -> the active-set attempt is accepted (a handful of rounds, not an
  // interior-point run) and returns the same optimum
  {
    DMDict inw = in, ow;
    inw["X_optm_ref"] = out["X_optm"];
    inw["U_optm_ref"] = out["U_optm"];
    inw["dU_optm_ref"] = out["dU_optm"];
    inw["convex_combi_optm_ref"] = out["convex_combi_optm"];
    Dict sw;
    mpc.solve(inw, ow, sw);
    double dw = 0;
    if (ow.count("X_optm"))
      for (std::size_t i = 0; i < (std::size_t)N; ++i)
        for (int k = 0; k < 6; ++k) dw = std::fmax(dw, std::fabs(ow["X_optm"](k, i) - out["X_optm"](k, i)) / sx[k]);
    std::printf("warm: warm_start %g  iter_count %g (cold %g)  distance from the cold answer %.2e\n", sw["warm_start"], sw["iter_count"], stats["iter_count"], dw);
    ok = ok && ow.count("X_optm") && sw["warm_start"] == 1.0 && sw["iter_count"] <= 4.0 && dw < 1e-9;
  }

  // recorder: drive the abscissa over the line twice; the first (partial) lap is dropped, the second is stored and saved
  DMDict o2;
  const int per_lap = 40;
  for (int s = 0; s < 2 * per_lap + 5; ++s) {
    DMDict in2 = in;
    in2["x_ic"](0, 0) = std::fmod(0.6 * L + s * L / per_lap, L);
    in2["t_ic"] = DM(0.03 * s);
    o2.clear();
    mpc.solve(in2, o2, stats);
  }
  try {
    const DM lx = rt::read_txt(std::string(argv[5]) + "lap_4_x.txt"), lt = rt::read_txt(std::string(argv[5]) + "lap_4_t.txt");
    // 3 loaded laps, then the dropped partial lap bumps lap_count_ to 4: the first completed lap is saved as lap_4
    std::printf("recorded lap: %zu x %zu samples, t %zu\n", lx.rows, lx.cols, lt.rows);
    ok = ok && lx.cols == 6 && lx.rows == (std::size_t)per_lap && lt.rows == (std::size_t)per_lap;
  } catch (const std::exception& e) {
    std::printf("FAIL: %s\n", e.what());
    ok = false;
  }
  std::puts(ok ? "PASS" : "FAIL");
  return ok ? 0 : 1;
}
