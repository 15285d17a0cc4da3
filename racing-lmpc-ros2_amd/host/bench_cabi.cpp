// bench_cabi.cpp -- the batched solve through the C ABI from plain C++ (HIP runtime for the buffers, no Python, no
// torch): the reference's BARC track file -> RacingTrajectory -> device tables -> lmpc_prepare_batch (node cold
// start) -> lmpc_solve_batch, timed with HIP events.  What a C++ caller of include/lmpc_hip.h looks like.
// usage: bench_cabi <track file> [batch=4096] [steps=50]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <random>
#include <vector>

#include "lmpc_hip.h"
#include "racing_trajectory.hpp"

#define HIP_OK(e)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (e);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                       \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

static double* dev(const std::vector<double>& v) {
  double* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(double)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const int B = argc > 2 ? std::atoi(argv[2]) : 4096, steps = argc > 3 ? std::atoi(argv[3]) : 50, N = 20, M = 1024;
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

  lmpc_handle* h = nullptr;
  if (lmpc_create(&c, &v, 0, &h) != LMPC_OK) { std::fprintf(stderr, "lmpc_create: %s\n", lmpc_last_error(h)); return 1; }

  // track file -> uniform device tables
  lmpc::vehicle_model::racing_trajectory::RacingTrajectory traj(argv[1]);
  std::vector<double> kap, bl, br, vel;
  traj.to_track_table(M, kap, bl, br, vel);
  lmpc_track tr{};
  tr.L = traj.total_length(); tr.M = M;
  tr.curvature = dev(kap); tr.bound_left = dev(bl); tr.bound_right = dev(br); tr.vel = dev(vel);

  // random initial states around the race line [6][B], inputs [2][B]
  std::mt19937_64 rng(0);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 1.0);
  std::vector<double> x((size_t)6 * B), u((size_t)2 * B, 0.0);
  for (int b = 0; b < B; ++b) {
    const double s = U01(rng) * tr.L;
    x[0 * (size_t)B + b] = s;
    x[1 * (size_t)B + b] = 0.1 * (U01(rng) - 0.5);
    x[2 * (size_t)B + b] = 0.03 * G(rng);
    x[3 * (size_t)B + b] = 0.8 * traj.velocity_interpolation(s);
    x[4 * (size_t)B + b] = 0.02 * G(rng);
    x[5 * (size_t)B + b] = 0.1 * G(rng);
  }
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
