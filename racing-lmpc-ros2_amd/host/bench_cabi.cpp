// bench_cabi.cpp -- the batched solve through the C ABI from plain C++ (HIP runtime for the buffers, no Python, no
// torch): the reference's BARC track file -> RacingTrajectory -> device tables -> lmpc_prepare_batch (node cold
// start) -> lmpc_solve_batch, timed with HIP events.  What a C++ caller of include/lmpc_hip.h looks like.
// usage: bench_cabi <track file> [batch=4096] [steps=50] [--gpus N [--same-device] [--gather none|copy|rccl]]
// With --gpus N the batch is PER SHARD (weak scaling, as bench.py --gpus N): N handles on N devices, one host thread and one
// stream each (host/sharded_solver.hpp), results gathered by RCCL all-gather (default when the devices are distinct) or by
// peer copies into shard 0 (--same-device puts every shard on device 0: the way the path is exercised on a one-GPU box).
// The sharded run is checked against ONE handle solving the whole batch: every problem bit for bit, gathered == own records.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "lmpc_hip.h"
#include "racing_trajectory.hpp"
#include "sharded_solver.hpp"

#define HIP_OK(e)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (e);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                       \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

// random initial states around the race line [6][B], inputs [2][B]
static void random_states(const lmpc::vehicle_model::racing_trajectory::RacingTrajectory& traj, size_t B, std::vector<double>& x,
                          std::vector<double>& u) {
  std::mt19937_64 rng(0);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 1.0);
  x.assign(6 * B, 0.0);
  u.assign(2 * B, 0.0);
  for (size_t b = 0; b < B; ++b) {
    const double s = U01(rng) * traj.total_length();
    x[0 * B + b] = s;
    x[1 * B + b] = 0.1 * (U01(rng) - 0.5);
    x[2 * B + b] = 0.03 * G(rng);
    x[3 * B + b] = 0.8 * traj.velocity_interpolation(s);
    x[4 * B + b] = 0.02 * G(rng);
    x[5 * B + b] = 0.1 * G(rng);
  }
}

static double* dev(const std::vector<double>& v) {
  double* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(double)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
  return d;
}

static int sharded(const lmpc_config& c, const lmpc_vehicle& v, const lmpc::vehicle_model::racing_trajectory::RacingTrajectory& traj,
                   const std::vector<double>& kap, const std::vector<double>& bl, const std::vector<double>& br, const std::vector<double>& vel,
                   int M, int B, int steps, int gpus, bool same_device, lmpc::mpc::GatherMode gather);

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  int gpus = 0;
  bool same_device = false;
  int gather = -1;
  std::vector<const char*> pos;
  for (int a = 1; a < argc; ++a) {
    if (!std::strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = std::atoi(argv[++a]);
    else if (!std::strcmp(argv[a], "--same-device")) same_device = true;
    else if (!std::strcmp(argv[a], "--gather") && a + 1 < argc) {
      const std::string g = argv[++a];
      gather = g == "none" ? lmpc::mpc::GATHER_NONE : g == "copy" ? lmpc::mpc::GATHER_COPY : g == "rccl" ? lmpc::mpc::GATHER_RCCL : -2;
      if (gather == -2) { std::fprintf(stderr, "--gather none|copy|rccl\n"); return 2; }
    } else pos.push_back(argv[a]);
  }
  if (pos.empty()) return 2;
  const int B = pos.size() > 1 ? std::atoi(pos[1]) : 4096, steps = pos.size() > 2 ? std::atoi(pos[2]) : 50, N = 20, M = 1024;
  const double inf = std::numeric_limits<double>::infinity();
  lmpc_vehicle v{};  // param/barc/*.yaml
  v.model_id = LMPC_MODEL_SINGLE_TRACK_PLANAR;
  v.m = 2.2187; v.Jzz = 0.02723; v.l = 0.324; v.cg_ratio = 0.5; v.h = 0.07; v.b = 0.281; v.fr = 0.012;
  v.kd = 0.0; v.kb = 0.5; v.cd = 0.0; v.Af = 1.0; v.rho = 1.2; v.cl_f = 0.0; v.cl_r = 0.0; v.mu = 0.9;
  v.Bf = 5.0; v.Cf = 2.28; v.Br = 5.0; v.Cr = 2.28; v.Fd_max = 15.0; v.Fb_max = -15.0; v.Td = 0.1; v.Tb = 0.1;
  v.max_steer = 0.314159; v.max_steer_rate = 10.0;
  lmpc_config c{};  // param/racing_mpc/barc_tracking_mpc.param.yaml
  c.N = N; c.num_ss_pts = 96; c.num_ss_pts_per_lap = 32; c.max_lap_stored = 3;
  c.margin = 0.1; c.q_contour = 1.0; c.q_heading = 1.0; c.q_vel = 0.2; c.q_vy = 1e-3; c.q_vyaw = 1e-3; c.q_boundary = 20.0;
  const double R[4] = {0.01, 0, 0, 0.01};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = R[k]; }
  const double xmax[6] = {inf, inf, inf, 6.0, 1.0, 3.0}, xmin[6] = {-inf, -inf, -inf, 0.1, -1.0, -3.0};
  for (int k = 0; k < 6; ++k) { c.x_max[k] = xmax[k]; c.x_min[k] = xmin[k]; c.convex_hull_slack[k] = 20.0; }
  c.u_max[0] = 0.01; c.u_max[1] = 0.33; c.u_min[0] = -0.01; c.u_min[1] = -0.33; c.max_vel_ref_diff = 1.0;

  // track file -> uniform tables
  lmpc::vehicle_model::racing_trajectory::RacingTrajectory traj(pos[0]);
  std::vector<double> kap, bl, br, vel;
  traj.to_track_table(M, kap, bl, br, vel);
  if (gpus > 0) {
    const lmpc::mpc::GatherMode g = gather >= 0 ? static_cast<lmpc::mpc::GatherMode>(gather) : (same_device ? lmpc::mpc::GATHER_COPY : lmpc::mpc::GATHER_RCCL);
    try {
      return sharded(c, v, traj, kap, bl, br, vel, M, B, steps, gpus, same_device, g);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "sharded: %s\n", e.what());
      return 1;
    }
  }
  lmpc_handle* h = nullptr;
  if (lmpc_create(&c, &v, 0, &h) != LMPC_OK) { std::fprintf(stderr, "lmpc_create: %s\n", lmpc_last_error(h)); return 1; }
  lmpc_track tr{};
  tr.L = traj.total_length(); tr.M = M;
  tr.curvature = dev(kap); tr.bound_left = dev(bl); tr.bound_right = dev(br); tr.vel = dev(vel);

  std::vector<double> x, u;
  random_states(traj, B, x, u);
  double *x_ic = dev(x), *u_ic = dev(u);
  const size_t NB = (size_t)N * B, SB = (size_t)(N - 1) * B;
  double *X_ref, *U_ref, *T_ref, *bL, *bR, *cu, *vr, *X, *Uo, *dU;
  HIP_OK(hipMalloc(&X_ref, 6 * NB * 8)); HIP_OK(hipMalloc(&U_ref, 2 * SB * 8)); HIP_OK(hipMalloc(&T_ref, SB * 8));
  HIP_OK(hipMalloc(&bL, NB * 8)); HIP_OK(hipMalloc(&bR, NB * 8)); HIP_OK(hipMalloc(&cu, NB * 8)); HIP_OK(hipMalloc(&vr, NB * 8));
  HIP_OK(hipMalloc(&X, 6 * NB * 8)); HIP_OK(hipMalloc(&Uo, 2 * SB * 8)); HIP_OK(hipMalloc(&dU, 2 * SB * 8));
  int32_t *status, *iters;
  HIP_OK(hipMalloc(&status, B * sizeof(int32_t))); HIP_OK(hipMalloc(&iters, B * sizeof(int32_t)));
  if (lmpc_reserve(h, B) != LMPC_OK ||
      lmpc_prepare_batch(h, B, &tr, x_ic, 0.025, 0.9, c.x_max[3], X_ref, U_ref, T_ref, bL, bR, cu, vr) != LMPC_OK) {
    std::fprintf(stderr, "prepare: %s\n", lmpc_last_error(h));
    return 1;
  }
  auto solve = [&]() {
    return lmpc_solve_batch(h, B, x_ic, u_ic, X_ref, U_ref, T_ref, bL, bR, cu, vr, tr.L, nullptr, nullptr, X, Uo, dU, nullptr,
                            status, iters, nullptr);
  };
  for (int k = 0; k < 5; ++k)
    if (solve() != LMPC_OK) { std::fprintf(stderr, "solve: %s\n", lmpc_last_error(h)); return 1; }
  HIP_OK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, nullptr));
  for (int k = 0; k < steps; ++k) solve();
  HIP_OK(hipEventRecord(e1, nullptr));
  HIP_OK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<int32_t> st(B), it(B);
  HIP_OK(hipMemcpy(st.data(), status, B * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(it.data(), iters, B * sizeof(int32_t), hipMemcpyDeviceToHost));
  long solved = 0, its = 0;
  for (int b = 0; b < B; ++b) { solved += st[b] == LMPC_SOLVE_OPTIMAL; its += it[b]; }
  std::printf("batch %d  steps %d  %.3f ms/step  %.0f solves/s  solved %.4f  mean iters %.2f\n", B, steps, ms / steps,
              1e3 * (double)B * steps / ms, (double)solved / B, (double)its / B);
  lmpc_destroy(h);
  return 0;
}

// N shards of B problems each against ONE handle solving all N B problems (device of shard 0): per problem bit for bit
static int sharded(const lmpc_config& c, const lmpc_vehicle& v, const lmpc::vehicle_model::racing_trajectory::RacingTrajectory& traj,
                   const std::vector<double>& kap, const std::vector<double>& bl, const std::vector<double>& br, const std::vector<double>& vel,
                   int M, int B, int steps, int gpus, bool same_device, lmpc::mpc::GatherMode gather) {
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  std::vector<int> devices;
  for (int r = 0; r < gpus; ++r) devices.push_back(same_device ? 0 : r);
  if (!same_device && gpus > ndev) { std::fprintf(stderr, "%d devices visible, %d asked for (--same-device puts every shard on device 0)\n", ndev, gpus); return 1; }
  const size_t total = (size_t)B * gpus, N = (size_t)c.N;
  std::vector<double> x, u;
  random_states(traj, total, x, u);
  lmpc::mpc::ShardedSolver sv(c, v, devices, B, gather);
  sv.set_track(traj.total_length(), M, kap.data(), bl.data(), br.data(), vel.data());
  sv.prepare(x.data(), u.data(), 0.025, 0.9, c.x_max[3]);
  sv.solve_many(5);                           // warm-up
  const double ms = sv.solve_many(steps);     // slowest shard's wall-clock over `steps` back-to-back steps
  double one = 0.0;
  sv.solve(&one);
  std::vector<double> all_d, own_d;
  std::vector<int32_t> all_i, own_i;
  sv.fetch(all_d, all_i, gpus - 1);           // (RCCL: read the LAST device's gathered copy)
  size_t gather_bad = 0;
  for (int r = 0; r < gpus; ++r) {
    sv.fetch_own(r, own_d, own_i);
    gather_bad += std::memcmp(own_d.data(), all_d.data() + sv.record_doubles() * r, own_d.size() * sizeof(double)) != 0;
    gather_bad += std::memcmp(own_i.data(), all_i.data() + sv.record_ints() * r, own_i.size() * sizeof(int32_t)) != 0;
  }
  // the unsharded solve of the same cars
  HIP_OK(hipSetDevice(devices[0]));
  lmpc_handle* h = nullptr;
  if (lmpc_create(&c, &v, devices[0], &h) != LMPC_OK) { std::fprintf(stderr, "lmpc_create: %s\n", lmpc_last_error(h)); return 1; }
  lmpc_track tr{};
  tr.L = traj.total_length(); tr.M = M;
  tr.curvature = dev(kap); tr.bound_left = dev(bl); tr.bound_right = dev(br); tr.vel = dev(vel);
  double *x_ic = dev(x), *u_ic = dev(u);
  const size_t NB = N * total, SB = (N - 1) * total;
  double *X_ref, *U_ref, *T_ref, *bL, *bR, *cu, *vr, *X, *Uo, *dU;
  HIP_OK(hipMalloc(&X_ref, 6 * NB * 8)); HIP_OK(hipMalloc(&U_ref, 2 * SB * 8)); HIP_OK(hipMalloc(&T_ref, SB * 8));
  HIP_OK(hipMalloc(&bL, NB * 8)); HIP_OK(hipMalloc(&bR, NB * 8)); HIP_OK(hipMalloc(&cu, NB * 8)); HIP_OK(hipMalloc(&vr, NB * 8));
  HIP_OK(hipMalloc(&X, 6 * NB * 8)); HIP_OK(hipMalloc(&Uo, 2 * SB * 8)); HIP_OK(hipMalloc(&dU, 2 * SB * 8));
  int32_t *status, *iters;
  HIP_OK(hipMalloc(&status, total * sizeof(int32_t))); HIP_OK(hipMalloc(&iters, total * sizeof(int32_t)));
  if (lmpc_prepare_batch(h, (int32_t)total, &tr, x_ic, 0.025, 0.9, c.x_max[3], X_ref, U_ref, T_ref, bL, bR, cu, vr) != LMPC_OK ||
      lmpc_solve_batch(h, (int32_t)total, x_ic, u_ic, X_ref, U_ref, T_ref, bL, bR, cu, vr, tr.L, nullptr, nullptr, X, Uo, dU, nullptr, status, iters,
                       nullptr) != LMPC_OK ||
      lmpc_synchronize(h) != LMPC_OK) {
    std::fprintf(stderr, "unsharded solve: %s\n", lmpc_last_error(h));
    return 1;
  }
  std::vector<double> Xh(6 * NB), Uh(2 * SB), dUh(2 * SB);
  std::vector<int32_t> st(total), it(total);
  HIP_OK(hipMemcpy(Xh.data(), X, Xh.size() * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(Uh.data(), Uo, Uh.size() * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(dUh.data(), dU, dUh.size() * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(st.data(), status, total * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(it.data(), iters, total * 4, hipMemcpyDeviceToHost));
  size_t differ = 0, solved = 0;
  long its = 0;
  for (size_t p = 0; p < total; ++p) {
    const size_t r = p / B, q = p % B;
    const double* rec = all_d.data() + sv.record_doubles() * r;
    const int32_t* ri = all_i.data() + sv.record_ints() * r;
    bool same = ri[q] == st[p] && ri[B + q] == it[p];
    for (size_t e = 0; e < 6 * N && same; ++e) same = rec[e * B + q] == Xh[e * total + p];
    for (size_t e = 0; e < 2 * (N - 1) && same; ++e)
      same = rec[(6 * N + e) * B + q] == Uh[e * total + p] && rec[(6 * N + 2 * (N - 1) + e) * B + q] == dUh[e * total + p];
    differ += !same;
    solved += ri[q] == LMPC_SOLVE_OPTIMAL;
    its += ri[B + q];
  }
  const char* gname = gather == lmpc::mpc::GATHER_RCCL ? "rccl" : gather == lmpc::mpc::GATHER_COPY ? "copy" : "none";
  std::printf("shards %d  devices %s  batch/shard %d  steps %d  gather %s  %.3f ms/step  %.0f solves/s  one step alone %.3f ms  solved %.4f  mean iters %.2f  "
              "differ_from_unsharded %zu  gather_mismatch %zu\n",
              gpus, same_device ? "all-0" : "distinct", B, steps, gname, ms / steps, 1e3 * (double)total * steps / ms, one, (double)solved / total,
              (double)its / total, differ, gather_bad);
  lmpc_destroy(h);
  return (differ == 0 && gather_bad == 0) ? 0 : 1;
}
