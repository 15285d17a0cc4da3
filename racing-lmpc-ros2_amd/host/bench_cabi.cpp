// bench_cabi.cpp -- the batched solve through the C ABI from plain C++ (HIP runtime for the buffers, no Python, no
// torch): track tables -> lmpc_prepare_batch (node cold start) -> the solve entry point of the chosen precision, timed with HIP
// events.  What a C++ caller of include/lmpc_hip.h looks like.
// usage: bench_cabi <track file | -> [batch=4096] [steps=50] [--workload tracking|iac|lmpc] [--horizon N] [--precision f64|f32|mixed]
//                   [--regression] [--gpus N [--same-device] [--gather none|copy|rccl]]
//   --workload tracking (default)  BARC tracking MPC (BASELINE configs[1]) on the reference's BARC track file (first argument);
//   --workload iac                 the IAC car on a synthetic Putnam-scale track (L = 2849 m) -- configs[3]'s problem; with
//                                  --horizon 40 --precision f32 the configuration as quoted;
//   --workload lmpc                BARC learning MPC, 5 laps / 160 safe-set points on a synthetic BARC-scale track -- configs[2];
//                                  with --precision mixed --regression configs[4]'s; the safe set goes BY REFERENCE
//                                  (lmpc_ss_query_idx_batch + lmpc_solve_batch_ss_idx).  (`-` for the track file with these two.)
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
#define LMPC_TRY(h, e)                                                                   \
  do {                                                                                   \
    if ((e) != LMPC_OK) {                                                                \
      std::fprintf(stderr, "%s: %s\n", #e, lmpc_last_error(h));                          \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

namespace {
const double PI = 3.14159265358979323846;

// everything that defines a run: configuration, vehicle, track tables, (learning) laps and regression data, initial states
struct Problem {
  lmpc_config c{};
  lmpc_vehicle v{};
  double L = 0.0;
  int M = 1024;
  std::vector<double> kap, bl, br, vel;
  double dt = 0.025, speed_scale = 0.9, speed_limit = 0.0;
  // learning
  std::vector<int32_t> lap_n;
  std::vector<double> lap_x;  // [total][6]
  // regression: two-sample laps (x [2][6], u [2][2], k [2], t [2]) per recorded pair
  bool regression = false;
  std::vector<int32_t> reg_n;
  std::vector<double> reg_x, reg_u, reg_k, reg_t;
  lmpc_regression_spec reg_spec{};
};

lmpc_vehicle barc_vehicle() {  // param/barc/*.yaml
  lmpc_vehicle v{};
  v.model_id = LMPC_MODEL_SINGLE_TRACK_PLANAR;
  v.m = 2.2187; v.Jzz = 0.02723; v.l = 0.324; v.cg_ratio = 0.5; v.h = 0.07; v.b = 0.281; v.fr = 0.012;
  v.kd = 0.0; v.kb = 0.5; v.cd = 0.0; v.Af = 1.0; v.rho = 1.2; v.cl_f = 0.0; v.cl_r = 0.0; v.mu = 0.9;
  v.Bf = 5.0; v.Cf = 2.28; v.Br = 5.0; v.Cr = 2.28; v.Fd_max = 15.0; v.Fb_max = -15.0; v.Td = 0.1; v.Tb = 0.1;
  v.max_steer = 0.314159; v.max_steer_rate = 10.0;
  return v;
}

lmpc_vehicle iac_vehicle() {  // param/iac_car/*.yaml
  lmpc_vehicle v{};
  v.model_id = LMPC_MODEL_SINGLE_TRACK_PLANAR;
  v.m = 811.9303; v.Jzz = 700.0; v.l = 2.9718; v.cg_ratio = 0.45; v.h = 0.35; v.b = 2.0; v.fr = 0.012;
  v.kd = 0.0; v.kb = 0.54; v.cd = 1.0; v.Af = 1.0; v.rho = 1.2; v.cl_f = 1.0; v.cl_r = 1.0; v.mu = 1.3;
  v.Bf = 11.0; v.Cf = 1.7; v.Br = 11.0; v.Cr = 1.7; v.Fd_max = 10000.0; v.Fb_max = -20000.0; v.Td = 0.1; v.Tb = 0.1;
  v.max_steer = 0.314159; v.max_steer_rate = 0.66;
  return v;
}

lmpc_config barc_tracking(int N) {  // param/racing_mpc/barc_tracking_mpc.param.yaml
  const double inf = std::numeric_limits<double>::infinity();
  lmpc_config c{};
  c.N = N; c.num_ss_pts = 96; c.num_ss_pts_per_lap = 32; c.max_lap_stored = 3;
  c.margin = 0.1; c.q_contour = 1.0; c.q_heading = 1.0; c.q_vel = 0.2; c.q_vy = 1e-3; c.q_vyaw = 1e-3; c.q_boundary = 20.0;
  const double R[4] = {0.01, 0, 0, 0.01};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = R[k]; }
  const double xmax[6] = {inf, inf, inf, 6.0, 1.0, 3.0}, xmin[6] = {-inf, -inf, -inf, 0.1, -1.0, -3.0};
  for (int k = 0; k < 6; ++k) { c.x_max[k] = xmax[k]; c.x_min[k] = xmin[k]; c.convex_hull_slack[k] = 20.0; }
  c.u_max[0] = 0.01; c.u_max[1] = 0.33; c.u_min[0] = -0.01; c.u_min[1] = -0.33; c.max_vel_ref_diff = 1.0;
  return c;
}

lmpc_config barc_lmpc(int N, int n_laps) {  // param/racing_mpc/barc_lmpc.param.yaml, with SURVEY.md 8(d) config 3's 5 laps
  lmpc_config c = barc_tracking(N);
  c.learning = 1; c.q_boundary = 1000.0;
  const double R[4] = {0.1, 0, 0, 0.1};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = R[k]; }
  c.x_max[3] = 3.0;
  const double chs[6] = {40.0, 40.0, 4.0, 40.0, 40.0, 4.0};
  for (int k = 0; k < 6; ++k) c.convex_hull_slack[k] = chs[k];
  c.num_ss_pts = 32 * n_laps; c.num_ss_pts_per_lap = 32; c.max_lap_stored = n_laps;
  return c;
}

lmpc_config iac_tracking(int N) {  // param/racing_mpc/iac_car_tracking_mpc.param.yaml
  const double inf = std::numeric_limits<double>::infinity();
  lmpc_config c{};
  c.N = N; c.num_ss_pts = 96; c.num_ss_pts_per_lap = 32; c.max_lap_stored = 3;
  c.margin = 0.5; c.q_contour = 1.0; c.q_heading = 1.0; c.q_vel = 0.2; c.q_vy = 0.01; c.q_vyaw = 0.01; c.q_boundary = 20.0;
  const double R[4] = {1e-5, 0, 0, 1.0}, Rd[4] = {1e-4, 0, 0, 10.0};
  for (int k = 0; k < 4; ++k) { c.R[k] = R[k]; c.R_d[k] = Rd[k]; }
  const double xmax[6] = {inf, inf, inf, 100.0, 15.0, 2.0}, xmin[6] = {-inf, -inf, -inf, 3.0, -15.0, -2.0};
  const double chs[6] = {20.0, 20.0, 2.0, 20.0, 20.0, 2.0};
  for (int k = 0; k < 6; ++k) { c.x_max[k] = xmax[k]; c.x_min[k] = xmin[k]; c.convex_hull_slack[k] = chs[k]; }
  c.u_max[0] = 5.0; c.u_max[1] = 0.314159; c.u_min[0] = -10.0; c.u_min[1] = -0.314159; c.max_vel_ref_diff = 1.0;
  return c;
}

// closed synthetic tracks of the two scales SURVEY.md 8(d) describes (the curvature integrates to 2 pi over a lap) -- the same
// formulas as racing-lmpc-ros2_amd/workloads.py:synthetic_track
void synthetic_track(bool putnam, Problem& p) {
  const int M = p.M;
  p.L = putnam ? 2849.0 : 15.6;
  p.kap.resize(M); p.bl.resize(M); p.br.resize(M); p.vel.resize(M);
  for (int i = 0; i < M; ++i) {
    const double th = 2.0 * PI * i / M;
    if (putnam) {
      const double c3 = std::cos(3 * th + 0.4);
      p.kap[i] = (2.0 * PI / p.L) * (1.0 + 8.0 * c3 * c3 * c3 + 5.0 * std::cos(5 * th - 0.7));
      p.bl[i] = 4.5 + 2.5 * std::sin(th + 0.5);
      p.br[i] = -(4.5 + 2.5 * std::cos(2 * th - 0.2));
      p.vel[i] = std::fmin(65.0, std::fmax(15.0, std::sqrt(1.6 * 9.8 / std::fmax(std::fabs(p.kap[i]), 1e-4))));
    } else {
      p.kap[i] = (2.0 * PI / p.L) * (1.0 + 0.9 * std::cos(2 * th + 0.3) + 0.45 * std::cos(3 * th - 1.1));
      p.bl[i] = 0.55 + 0.25 * std::sin(th + 0.5);
      p.br[i] = -(0.55 + 0.25 * std::cos(2 * th - 0.2));
      p.vel[i] = std::fmin(4.5, std::fmax(1.5, std::sqrt(0.45 * 9.8 / std::fmax(std::fabs(p.kap[i]), 1e-3))));
    }
  }
}

double table_at(const std::vector<double>& t, double L, double s) {  // periodic linear interpolation
  const int M = (int)t.size();
  double x = std::fmod(s, L);
  if (x < 0) x += L;
  x *= M / L;
  const int i = (int)x % M;
  const double f = x - std::floor(x);
  return t[i] * (1.0 - f) + t[(i + 1) % M] * f;
}

// stand-ins for recorded laps (workloads.py:synthetic_laps): n_pts samples per lap, a lateral weave per lap, speed rising lap by lap
void synthetic_laps(Problem& p, int n_laps, int n_pts) {
  for (int l = 0; l < n_laps; ++l) {
    p.lap_n.push_back(n_pts);
    for (int j = 0; j < n_pts; ++j) {
      const double s = (j + 0.37) * p.L / n_pts, k = table_at(p.kap, p.L, s), vx = 1.4 + 0.05 * l, w = 2 * PI * 3 / p.L;
      const double row[6] = {s, 0.06 * std::sin(w * s + 0.9 * l), 0.06 * w * std::cos(w * s + 0.9 * l), vx, 0.0, k * vx};
      p.lap_x.insert(p.lap_x.end(), row, row + 6);
    }
  }
}

// random initial states [6][B], inputs [2][B]
void random_states(const Problem& p, int kind, size_t B, std::vector<double>& x, std::vector<double>& u) {
  std::mt19937_64 rng(0);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 1.0);
  x.assign(6 * B, 0.0);
  u.assign(2 * B, 0.0);
  for (size_t b = 0; b < B; ++b) {
    const double s = U01(rng) * p.L;
    x[0 * B + b] = s;
    if (kind == 1) {  // iac: SURVEY.md 8(d) config 4's ranges, the speed a fraction of the profile so that the cold start is well posed
      x[1 * B + b] = 3.0 * (U01(rng) - 0.5);
      x[2 * B + b] = 0.05 * G(rng);
      x[3 * B + b] = (0.6 + 0.3 * U01(rng)) * table_at(p.vel, p.L, s);
      x[4 * B + b] = 0.5 * G(rng);
      x[5 * B + b] = 0.1 * G(rng);
    } else if (kind == 2) {  // lmpc: near the last stored lap
      const size_t n = (size_t)p.lap_n.back(), j = (size_t)(U01(rng) * n) % n, o = (p.lap_x.size() / 6 - n + j) * 6;
      const double sd[6] = {0.0, 0.03, 0.03, 0.1, 0.02, 0.1};
      for (int k = 0; k < 6; ++k) x[k * B + b] = p.lap_x[o + k] + sd[k] * G(rng);
      x[0 * B + b] = std::fmod(x[0 * B + b] + p.L, p.L);
    } else {
      x[1 * B + b] = 0.1 * (U01(rng) - 0.5);
      x[2 * B + b] = 0.03 * G(rng);
      x[3 * B + b] = 0.8 * table_at(p.vel, p.L, s);
      x[4 * B + b] = 0.02 * G(rng);
      x[5 * B + b] = 0.1 * G(rng);
    }
  }
}

double* dev(const std::vector<double>& v) {
  double* d = nullptr;
  if (hipMalloc(&d, v.size() * sizeof(double)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice);
  return d;
}

// Recorded data of a plant that differs from the model (15 % less grip), as bench.py --regression builds it: states around the stored
// laps, random inputs, each state's successor 30 ms later from the PLANT step kernel of a second handle -- two-sample laps.
int regression_pairs(Problem& p) {
  lmpc_vehicle pv = p.v;
  pv.mu *= 0.85;
  lmpc_handle* h = nullptr;
  if (lmpc_create(&p.c, &pv, 0, &h) != LMPC_OK) { std::fprintf(stderr, "lmpc_create (plant): %s\n", lmpc_last_error(h)); return 1; }
  const size_t n = p.lap_x.size() / 6;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U01(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 1.0);
  std::vector<double> xa(6 * n), ua(2 * n), ka(n);
  const double sd[6] = {0.0, 0.02, 0.02, 0.1, 0.03, 0.2};
  for (size_t j = 0; j < n; ++j) {
    for (int k = 0; k < 6; ++k) xa[k * n + j] = p.lap_x[6 * j + k] + sd[k] * G(rng);
    ua[0 * n + j] = -0.005 + 0.01 * U01(rng);
    ua[1 * n + j] = -0.15 + 0.3 * U01(rng);
    ka[j] = table_at(p.kap, p.L, xa[j]);
  }
  lmpc_track tr{};
  tr.L = p.L; tr.M = p.M;
  tr.curvature = dev(p.kap); tr.bound_left = dev(p.bl); tr.bound_right = dev(p.br); tr.vel = dev(p.vel);
  double *xd = dev(xa), *ud = dev(ua);
  LMPC_TRY(h, lmpc_plant_step_batch(h, (int32_t)n, &tr, xd, ud, 0.03, 1));
  LMPC_TRY(h, lmpc_synchronize(h));
  std::vector<double> xb(6 * n);
  HIP_OK(hipMemcpy(xb.data(), xd, xb.size() * 8, hipMemcpyDeviceToHost));
  lmpc_destroy(h);
  for (size_t j = 0; j < n; ++j) {
    p.reg_n.push_back(2);
    for (int k = 0; k < 6; ++k) p.reg_x.push_back(xa[k * n + j]);
    for (int k = 0; k < 6; ++k) p.reg_x.push_back(xb[k * n + j]);
    for (int r = 0; r < 2; ++r) { p.reg_u.push_back(ua[0 * n + j]); p.reg_u.push_back(ua[1 * n + j]); }
    p.reg_k.push_back(ka[j]); p.reg_k.push_back(ka[j]);
    p.reg_t.push_back(0.0); p.reg_t.push_back(0.03);
  }
  p.reg_spec.n_out = 3; p.reg_spec.out[0] = 3; p.reg_spec.out[1] = 4; p.reg_spec.out[2] = 5;
  p.reg_spec.n_in_state = 3; p.reg_spec.in_state[0] = 3; p.reg_spec.in_state[1] = 4; p.reg_spec.in_state[2] = 5;
  p.reg_spec.n_in_ctrl = 2; p.reg_spec.in_ctrl[0] = 0; p.reg_spec.in_ctrl[1] = 1;
  p.reg_spec.as_written = 0; p.reg_spec.dist_max = 0.6;
  p.regression = true;
  return 0;
}

// ONE handle on `device` solving `B` problems through the entry point of `prec`: results to the host as doubles (floats widened),
// [X 6 N B | U 2 (N-1) B | dU], status, iters; `ms` per step over `steps` timed steps (0: one solve, no timing)
int single_handle(const Problem& p, int device, int prec, const std::vector<double>& x, const std::vector<double>& u, int B, int steps,
                  std::vector<double>& val, std::vector<int32_t>& st, std::vector<int32_t>& it, double* ms_per_step, int* precision_ran) {
  HIP_OK(hipSetDevice(device));
  lmpc_handle* h = nullptr;
  if (lmpc_create(&p.c, &p.v, device, &h) != LMPC_OK) { std::fprintf(stderr, "lmpc_create: %s\n", lmpc_last_error(h)); return 1; }
  lmpc_track tr{};
  tr.L = p.L; tr.M = p.M;
  tr.curvature = dev(p.kap); tr.bound_left = dev(p.bl); tr.bound_right = dev(p.br); tr.vel = dev(p.vel);
  if (p.c.learning) LMPC_TRY(h, lmpc_set_safe_set(h, (int32_t)p.lap_n.size(), p.lap_n.data(), p.lap_x.data(), p.L));
  if (p.regression)
    LMPC_TRY(h, lmpc_set_regression_laps(h, (int32_t)p.reg_n.size(), p.reg_n.data(), p.reg_x.data(), p.reg_u.data(), p.reg_k.data(), p.reg_t.data(), &p.reg_spec));
  const size_t N = (size_t)p.c.N, b = (size_t)B, NB = N * b, SB = (N - 1) * b;
  double *x_ic = dev(x), *u_ic = dev(u);
  double *X_ref, *U_ref, *T_ref, *bL, *bR, *cu, *vr;
  HIP_OK(hipMalloc(&X_ref, 6 * NB * 8)); HIP_OK(hipMalloc(&U_ref, 2 * SB * 8)); HIP_OK(hipMalloc(&T_ref, SB * 8));
  HIP_OK(hipMalloc(&bL, NB * 8)); HIP_OK(hipMalloc(&bR, NB * 8)); HIP_OK(hipMalloc(&cu, NB * 8)); HIP_OK(hipMalloc(&vr, NB * 8));
  const size_t nval = 6 * NB + 4 * SB, es = prec == LMPC_PRECISION_F32 ? 4 : 8;
  void* out = nullptr;
  HIP_OK(hipMalloc(&out, nval * es));
  int32_t *status, *iters;
  HIP_OK(hipMalloc(&status, b * sizeof(int32_t))); HIP_OK(hipMalloc(&iters, b * sizeof(int32_t)));
  LMPC_TRY(h, lmpc_reserve(h, B));
  LMPC_TRY(h, lmpc_prepare_batch(h, B, &tr, x_ic, p.dt, p.speed_scale, p.speed_limit, X_ref, U_ref, T_ref, bL, bR, cu, vr));
  LMPC_TRY(h, lmpc_synchronize(h));
  double* query = nullptr;
  int32_t *ss_idx = nullptr, *n_found = nullptr;
  if (p.c.learning) {  // the query of every problem, as ShardedSolver::prepare forms it (racing_mpc.cpp:219-223, 249-254)
    std::vector<double> last(2 * b), q(2 * b);
    HIP_OK(hipMemcpy(last.data(), X_ref + (N - 1) * b, b * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(last.data() + b, X_ref + (N + N - 1) * b, b * 8, hipMemcpyDeviceToHost));
    for (size_t j = 0; j < b; ++j) {
      const double s1 = last[j], s2 = x[j], k = std::fabs(s2 - s1) + p.L / 2, l = k - std::fmod(k, p.L);
      q[j] = s1 + l * ((s2 > s1) - (s2 < s1));
      q[b + j] = last[b + j];
    }
    query = dev(q);
    HIP_OK(hipMalloc(&ss_idx, (size_t)p.c.num_ss_pts * b * sizeof(int32_t))); HIP_OK(hipMalloc(&n_found, b * sizeof(int32_t)));
  }
  float* f32 = nullptr;
  if (prec == LMPC_PRECISION_F32) {
    const double* src[9] = {x_ic, u_ic, X_ref, U_ref, T_ref, bL, bR, cu, vr};
    const size_t n[9] = {6 * b, 2 * b, 6 * NB, 2 * SB, SB, NB, NB, NB, NB};
    HIP_OK(hipMalloc(&f32, (8 * b + 6 * NB + 3 * SB + 4 * NB) * sizeof(float)));
    size_t o = 0;
    for (int a = 0; a < 9; ++a) {
      std::vector<double> hd(n[a]);
      std::vector<float> hf(n[a]);
      HIP_OK(hipMemcpy(hd.data(), src[a], n[a] * 8, hipMemcpyDeviceToHost));
      for (size_t e = 0; e < n[a]; ++e) hf[e] = (float)hd[e];
      HIP_OK(hipMemcpy(f32 + o, hf.data(), n[a] * 4, hipMemcpyHostToDevice));
      o += n[a];
    }
  }
  auto solve = [&]() -> int {
    if (prec == LMPC_PRECISION_F32) {
      float *X = (float*)out, *U = X + 6 * NB, *dU = U + 2 * SB;
      const float *a0 = f32, *a1 = a0 + 6 * b, *a2 = a1 + 2 * b, *a3 = a2 + 6 * NB, *a4 = a3 + 2 * SB, *a5 = a4 + SB, *a6 = a5 + NB, *a7 = a6 + NB, *a8 = a7 + NB;
      return lmpc_solve_batch_f32(h, B, a0, a1, a2, a3, a4, a5, a6, a7, a8, X, U, dU, status, iters, nullptr);
    }
    double *X = (double*)out, *U = X + 6 * NB, *dU = U + 2 * SB;
    if (p.c.learning) {
      const int rc = lmpc_ss_query_idx_batch(h, B, query, ss_idx, n_found);
      if (rc != LMPC_OK) return rc;
      return lmpc_solve_batch_ss_idx(h, B, prec, x_ic, u_ic, X_ref, U_ref, T_ref, bL, bR, cu, vr, tr.L, ss_idx, X, U, dU, nullptr, status, iters, nullptr);
    }
    return (prec == LMPC_PRECISION_MIXED ? lmpc_solve_batch_mixed : lmpc_solve_batch)(h, B, x_ic, u_ic, X_ref, U_ref, T_ref, bL, bR, cu, vr, tr.L, nullptr,
                                                                                      nullptr, X, U, dU, nullptr, status, iters, nullptr);
  };
  if (steps > 0) {
    for (int k = 0; k < 5; ++k) LMPC_TRY(h, solve());
    HIP_OK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    HIP_OK(hipEventRecord(e0, nullptr));
    for (int k = 0; k < steps; ++k) solve();
    HIP_OK(hipEventRecord(e1, nullptr));
    HIP_OK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_per_step) *ms_per_step = ms / steps;
  } else {
    LMPC_TRY(h, solve());
  }
  LMPC_TRY(h, lmpc_synchronize(h));
  if (precision_ran) {
    int32_t pr = -1;
    LMPC_TRY(h, lmpc_last_solve_precision(h, &pr));
    *precision_ran = pr;
  }
  val.resize(nval);
  st.resize(b);
  it.resize(b);
  if (es == 8) {
    HIP_OK(hipMemcpy(val.data(), out, nval * 8, hipMemcpyDeviceToHost));
  } else {
    std::vector<float> f(nval);
    HIP_OK(hipMemcpy(f.data(), out, nval * 4, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < nval; ++e) val[e] = (double)f[e];
  }
  HIP_OK(hipMemcpy(st.data(), status, b * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(it.data(), iters, b * 4, hipMemcpyDeviceToHost));
  lmpc_destroy(h);
  return 0;
}

const char* PREC_NAME[3] = {"f64", "f32", "mixed"};

// N shards of B problems each against ONE handle solving all N B problems (device of shard 0): per problem bit for bit
int sharded(const Problem& p, int kind, int prec, int B, int steps, int gpus, bool same_device, lmpc::mpc::GatherMode gather) {
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  std::vector<int> devices;
  for (int r = 0; r < gpus; ++r) devices.push_back(same_device ? 0 : r);
  if (!same_device && gpus > ndev) { std::fprintf(stderr, "%d devices visible, %d asked for (--same-device puts every shard on device 0)\n", ndev, gpus); return 1; }
  const size_t total = (size_t)B * gpus, N = (size_t)p.c.N;
  std::vector<double> x, u;
  random_states(p, kind, total, x, u);
  lmpc::mpc::ShardedSolver sv(p.c, p.v, devices, B, gather, static_cast<lmpc::mpc::Precision>(prec));
  sv.set_track(p.L, p.M, p.kap.data(), p.bl.data(), p.br.data(), p.vel.data());
  if (p.c.learning) sv.set_safe_set((int32_t)p.lap_n.size(), p.lap_n.data(), p.lap_x.data(), p.L);
  if (p.regression) sv.set_regression_laps((int32_t)p.reg_n.size(), p.reg_n.data(), p.reg_x.data(), p.reg_u.data(), p.reg_k.data(), p.reg_t.data(), &p.reg_spec);
  sv.prepare(x.data(), u.data(), p.dt, p.speed_scale, p.speed_limit);
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
    gather_bad += std::memcmp(own_d.data(), all_d.data() + sv.record_values() * r, own_d.size() * sizeof(double)) != 0;
    gather_bad += std::memcmp(own_i.data(), all_i.data() + sv.record_ints() * r, own_i.size() * sizeof(int32_t)) != 0;
  }
  // the unsharded solve of the same cars
  std::vector<double> val;
  std::vector<int32_t> st, it;
  int ran = -1;
  if (single_handle(p, devices[0], prec, x, u, (int)total, 0, val, st, it, nullptr, &ran)) return 1;
  const size_t NB = N * total, SB = (N - 1) * total;
  const double *Xh = val.data(), *Uh = Xh + 6 * NB, *dUh = Uh + 2 * SB;
  size_t differ = 0, solved = 0;
  long its = 0;
  for (size_t q0 = 0; q0 < total; ++q0) {
    const size_t r = q0 / B, q = q0 % B;
    const double* rec = all_d.data() + sv.record_values() * r;
    const int32_t* ri = all_i.data() + sv.record_ints() * r;
    bool same = ri[q] == st[q0] && ri[B + q] == it[q0];
    for (size_t e = 0; e < 6 * N && same; ++e) same = rec[e * B + q] == Xh[e * total + q0];
    for (size_t e = 0; e < 2 * (N - 1) && same; ++e)
      same = rec[(6 * N + e) * B + q] == Uh[e * total + q0] && rec[(6 * N + 2 * (N - 1) + e) * B + q] == dUh[e * total + q0];
    differ += !same;
    solved += ri[q] == LMPC_SOLVE_OPTIMAL;
    its += ri[B + q];
  }
  const char* gname = gather == lmpc::mpc::GATHER_RCCL ? "rccl" : gather == lmpc::mpc::GATHER_COPY ? "copy" : "none";
  std::printf("shards %d  devices %s  batch/shard %d  horizon %d  precision %s  ran_in %s  regression %d  steps %d  gather %s  %.3f ms/step  %.0f solves/s  "
              "one step alone %.3f ms  solved %.4f  mean iters %.2f  differ_from_unsharded %zu  gather_mismatch %zu\n",
              gpus, same_device ? "all-0" : "distinct", B, p.c.N, PREC_NAME[prec], PREC_NAME[(int)sv.last_solve_precision()], (int)p.regression, steps, gname,
              ms / steps, 1e3 * (double)total * steps / ms, one, (double)solved / total, (double)its / total, differ, gather_bad);
  return (differ == 0 && gather_bad == 0 && ran == (int)sv.last_solve_precision()) ? 0 : 1;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  int gpus = 0, horizon = 20, prec = LMPC_PRECISION_F64, kind = 0;
  bool same_device = false, regression = false;
  int gather = -1;
  std::vector<const char*> pos;
  for (int a = 1; a < argc; ++a) {
    if (!std::strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = std::atoi(argv[++a]);
    else if (!std::strcmp(argv[a], "--same-device")) same_device = true;
    else if (!std::strcmp(argv[a], "--regression")) regression = true;
    else if (!std::strcmp(argv[a], "--horizon") && a + 1 < argc) horizon = std::atoi(argv[++a]);
    else if (!std::strcmp(argv[a], "--workload") && a + 1 < argc) {
      const std::string w = argv[++a];
      kind = w == "tracking" ? 0 : w == "iac" ? 1 : w == "lmpc" ? 2 : -1;
      if (kind < 0) { std::fprintf(stderr, "--workload tracking|iac|lmpc\n"); return 2; }
    } else if (!std::strcmp(argv[a], "--precision") && a + 1 < argc) {
      const std::string w = argv[++a];
      prec = w == "f64" ? LMPC_PRECISION_F64 : w == "f32" ? LMPC_PRECISION_F32 : w == "mixed" ? LMPC_PRECISION_MIXED : -1;
      if (prec < 0) { std::fprintf(stderr, "--precision f64|f32|mixed\n"); return 2; }
    } else if (!std::strcmp(argv[a], "--gather") && a + 1 < argc) {
      const std::string g = argv[++a];
      gather = g == "none" ? lmpc::mpc::GATHER_NONE : g == "copy" ? lmpc::mpc::GATHER_COPY : g == "rccl" ? lmpc::mpc::GATHER_RCCL : -2;
      if (gather == -2) { std::fprintf(stderr, "--gather none|copy|rccl\n"); return 2; }
    } else pos.push_back(argv[a]);
  }
  if (pos.empty()) return 2;
  const int B = pos.size() > 1 ? std::atoi(pos[1]) : 4096, steps = pos.size() > 2 ? std::atoi(pos[2]) : 50;
  Problem p;
  try {
    if (kind == 0) {  // the reference's BARC track file -> uniform tables
      p.c = barc_tracking(horizon);
      p.v = barc_vehicle();
      lmpc::vehicle_model::racing_trajectory::RacingTrajectory traj(pos[0]);
      traj.to_track_table(p.M, p.kap, p.bl, p.br, p.vel);
      p.L = traj.total_length();
    } else if (kind == 1) {
      p.c = iac_tracking(horizon);
      p.v = iac_vehicle();
      synthetic_track(true, p);
    } else {
      p.c = barc_lmpc(horizon, 5);
      p.v = barc_vehicle();
      synthetic_track(false, p);
      synthetic_laps(p, 5, 440);
      p.speed_scale = 1.0;
      if (regression && regression_pairs(p)) return 1;
    }
    p.speed_limit = p.c.x_max[3];
    if (regression && kind != 2) { std::fprintf(stderr, "--regression goes with --workload lmpc\n"); return 2; }
    if (gpus > 0) {
      const lmpc::mpc::GatherMode g = gather >= 0 ? static_cast<lmpc::mpc::GatherMode>(gather) : (same_device ? lmpc::mpc::GATHER_COPY : lmpc::mpc::GATHER_RCCL);
      return sharded(p, kind, prec, B, steps, gpus, same_device, g);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "bench_cabi: %s\n", e.what());
    return 1;
  }
  std::vector<double> x, u, val;
  std::vector<int32_t> st, it;
  random_states(p, kind, B, x, u);
  double ms = 0.0;
  int ran = -1;
  if (single_handle(p, 0, prec, x, u, B, steps, val, st, it, &ms, &ran)) return 1;
  long solved = 0, its = 0;
  for (int b = 0; b < B; ++b) { solved += st[b] == LMPC_SOLVE_OPTIMAL; its += it[b]; }
  std::printf("batch %d  horizon %d  precision %s  ran_in %s  steps %d  %.3f ms/step  %.0f solves/s  solved %.4f  mean iters %.2f\n", B, p.c.N, PREC_NAME[prec],
              PREC_NAME[ran], steps, ms, 1e3 * (double)B / ms, (double)solved / B, (double)its / B);
  return 0;
}
