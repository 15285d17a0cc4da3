// sharded_solver.cpp -- see sharded_solver.hpp.  HIP runtime + RCCL + the C ABI; compiled as host code by hipcc.
#include "sharded_solver.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <set>
#include <stdexcept>

namespace lmpc {
namespace mpc {

namespace {
#define SH_HIP(s, e)                                                                  \
  do {                                                                                \
    hipError_t e_ = (e);                                                              \
    if (e_ != hipSuccess) {                                                           \
      (s).error = std::string(#e) + ": " + hipGetErrorString(e_);                     \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)
#define SH_LMPC(s, e)                                                                 \
  do {                                                                                \
    if ((e) != LMPC_OK) {                                                             \
      (s).error = std::string(#e) + ": " + lmpc_last_error((s).h);                    \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)
#define SH_NCCL(s, e)                                                                 \
  do {                                                                                \
    ncclResult_t e_ = (e);                                                            \
    if (e_ != ncclSuccess) {                                                          \
      (s).error = std::string(#e) + ": " + ncclGetErrorString(e_);                    \
      return -1;                                                                      \
    }                                                                                 \
  } while (0)
template <typename T>
int dmalloc(T** p, std::size_t n) {
  return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)) == hipSuccess ? 0 : -1;
}
}  // namespace

ShardedSolver::ShardedSolver(const lmpc_config& cfg, const lmpc_vehicle& veh, const std::vector<int>& devices, int32_t shard_batch,
                             GatherMode gather, Precision precision)
    : cfg_(cfg), veh_(veh), b_(shard_batch), gather_(gather), prec_(precision) {
  if (devices.empty() || shard_batch < 1) throw std::runtime_error("ShardedSolver: need at least one device and one problem per shard");
  if (cfg.learning && precision == PRECISION_F32)
    throw std::runtime_error("ShardedSolver: single precision is built for the tracking problem only (lmpc_solve_batch_f32)");
  const std::size_t N = static_cast<std::size_t>(cfg.N), b = static_cast<std::size_t>(shard_batch);
  rec_d_ = (6 * N + 4 * (N - 1)) * b;
  rec_i_ = 2 * b;
  elem_ = precision == PRECISION_F32 ? sizeof(float) : sizeof(double);
  shards_ = std::vector<Shard>(devices.size());
  for (std::size_t r = 0; r < devices.size(); ++r) {
    shards_[r].rank = static_cast<int>(r);
    shards_[r].device = devices[r];
  }
  if (gather_ == GATHER_RCCL) {
    if (std::set<int>(devices.begin(), devices.end()).size() != devices.size())
      throw std::runtime_error("ShardedSolver: GATHER_RCCL needs one distinct device per shard (one communicator rank per GPU)");
    std::vector<ncclComm_t> comms(devices.size());
    const ncclResult_t rc = ncclCommInitAll(comms.data(), static_cast<int>(devices.size()), devices.data());
    if (rc != ncclSuccess) throw std::runtime_error(std::string("ncclCommInitAll: ") + ncclGetErrorString(rc));
    for (std::size_t r = 0; r < devices.size(); ++r) shards_[r].comm = comms[r];
  }
  // A constructor that throws does not run the destructor, and a joinable std::thread that is destroyed terminates the process:
  // whatever fails from here on, the workers are told to quit (they release what they created, on their device) and joined first.
  try {
    for (Shard& s : shards_) s.worker = std::thread([this, &s] { run(s); });
    issue(CMD_INIT);
  } catch (...) {
    quit();
    throw;
  }
}

ShardedSolver::~ShardedSolver() { quit(); }

void ShardedSolver::quit() {
  {
    std::lock_guard<std::mutex> lk(mu_);
    cmd_ = CMD_QUIT;
    ++generation_;
  }
  cv_go_.notify_all();
  for (Shard& s : shards_) {
    if (s.worker.joinable())
      s.worker.join();  // (the worker destroyed -- or, poisoned, aborted -- its communicator and released handle, stream and buffers)
    else if (s.comm)    // its thread never started (std::thread's constructor threw)
      (void)ncclCommAbort(static_cast<ncclComm_t>(s.comm));
    s.comm = nullptr;
  }
}

// Host barrier over the workers, carrying one bit: every worker calls it the same number of times per command (the decisions
// taken on its result are the same on every worker), and gets the AND of what all of them brought.
bool ShardedSolver::agree(bool ok) {
  std::unique_lock<std::mutex> lk(bar_mu_);
  const uint64_t gen = bar_gen_;
  bar_ok_ = bar_ok_ && ok;
  if (++bar_count_ == n_shards()) {
    bar_result_ = bar_ok_;   // (stable until every waiter of this round has returned: the next round cannot complete before)
    bar_ok_ = true;
    bar_count_ = 0;
    ++bar_gen_;
    bar_cv_.notify_all();
    return bar_result_;
  }
  bar_cv_.wait(lk, [&] { return bar_gen_ != gen; });
  return bar_result_;
}

void ShardedSolver::run(Shard& s) {
  uint64_t seen = 0;
  for (;;) {
    Command c;
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_go_.wait(lk, [&] { return generation_ != seen; });
      seen = generation_;
      c = cmd_;
    }
    int rc = 0;
    switch (c) {
      case CMD_INIT: rc = do_init(s); break;
      case CMD_TRACK: rc = do_track(s); break;
      case CMD_SAFE_SET: rc = do_safe_set(s); break;
      case CMD_REGRESSION: rc = do_regression(s); break;
      case CMD_PREPARE: rc = do_prepare(s); break;
      case CMD_SOLVE: rc = do_solve(s); break;
      default: break;
    }
    if (c == CMD_QUIT) {  // release what this thread created, on its device
      (void)hipSetDevice(s.device);
      // a poisoned object may have a collective in flight that will never complete: abort first, and do not wait on the stream
      if (s.comm) (void)(poisoned_ ? ncclCommAbort(static_cast<ncclComm_t>(s.comm)) : ncclCommDestroy(static_cast<ncclComm_t>(s.comm)));
      s.comm = nullptr;
      if (s.stream && !poisoned_) (void)hipStreamSynchronize(static_cast<hipStream_t>(s.stream));
      if (s.h) lmpc_destroy(s.h);
      for (double* p : {s.x_ic, s.u_ic, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref, s.query, static_cast<double*>(s.rec_d),
                        static_cast<double*>(s.all_d), const_cast<double*>(s.track.curvature), const_cast<double*>(s.track.bound_left),
                        const_cast<double*>(s.track.bound_right), const_cast<double*>(s.track.vel)})
        if (p) (void)hipFree(p);
      for (void* p : {static_cast<void*>(s.rec_i), static_cast<void*>(s.all_i), static_cast<void*>(s.f32), static_cast<void*>(s.ss_idx),
                      static_cast<void*>(s.n_found)})
        if (p) (void)hipFree(p);
      if (s.stream) (void)hipStreamDestroy(static_cast<hipStream_t>(s.stream));
      return;
    }
    (void)rc;  // (the message is in s.error)
    {
      std::lock_guard<std::mutex> lk(mu_);
      --pending_;
    }
    cv_done_.notify_all();
  }
}

void ShardedSolver::issue(Command c) {
  if (poisoned_) throw std::runtime_error("ShardedSolver: an earlier call failed on a shard; the object cannot be used any more (destroy it)");
  {
    std::unique_lock<std::mutex> lk(mu_);
    for (Shard& s : shards_) s.error.clear();
    cmd_ = c;
    pending_ = n_shards();
    ++generation_;
  }
  cv_go_.notify_all();
  {
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
  }
  for (const Shard& s : shards_)
    if (!s.error.empty()) {
      poisoned_ = true;  // (streams may hold half a step, communicators may be aborted: nothing after this is well defined)
      throw std::runtime_error("ShardedSolver shard " + std::to_string(s.rank) + " (device " + std::to_string(s.device) + "): " + s.error);
    }
}

int ShardedSolver::do_init(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  hipStream_t st;
  SH_HIP(s, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  s.stream = st;
  if (lmpc_create(&cfg_, &veh_, s.device, &s.h) != LMPC_OK) {
    s.error = std::string("lmpc_create: ") + (s.h ? lmpc_last_error(s.h) : "allocation failed");
    return -1;
  }
  SH_LMPC(s, lmpc_set_stream(s.h, s.stream));
  SH_LMPC(s, lmpc_reserve(s.h, b_));
  const std::size_t N = static_cast<std::size_t>(cfg_.N), b = static_cast<std::size_t>(b_), NB = N * b, SB = (N - 1) * b;
  int bad = dmalloc(&s.x_ic, 6 * b) | dmalloc(&s.u_ic, 2 * b) | dmalloc(&s.X_ref, 6 * NB) | dmalloc(&s.U_ref, 2 * SB) | dmalloc(&s.T_ref, SB) |
            dmalloc(&s.bl, NB) | dmalloc(&s.br, NB) | dmalloc(&s.kap, NB) | dmalloc(&s.vref, NB) | dmalloc(&s.rec_i, rec_i_);
  bad |= hipMalloc(&s.rec_d, rec_d_ * elem_) != hipSuccess;
  if (prec_ == PRECISION_F32) bad |= dmalloc(&s.f32, 8 * b + 6 * NB + 3 * SB + 4 * NB);
  if (cfg_.learning)
    bad |= dmalloc(&s.query, 2 * b) | dmalloc(&s.ss_idx, static_cast<std::size_t>(cfg_.num_ss_pts) * b) | dmalloc(&s.n_found, b);
  const bool holds_all = gather_ == GATHER_RCCL || (gather_ == GATHER_COPY && s.rank == 0);
  if (holds_all) bad |= (hipMalloc(&s.all_d, rec_d_ * elem_ * shards_.size()) != hipSuccess) | dmalloc(&s.all_i, rec_i_ * shards_.size());
  if (bad) {
    s.error = "hipMalloc failed";
    return -1;
  }
  return 0;
}

int ShardedSolver::do_track(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  const std::size_t M = static_cast<std::size_t>(M_);
  double* t[4] = {nullptr, nullptr, nullptr, nullptr};
  const double* src[4] = {t_kap_, t_bl_, t_br_, t_vel_};
  for (int k = 0; k < 4; ++k) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&t[k]), M * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(t[k], src[k], M * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {  // the tables in use stay in use
      for (double* q : t)
        if (q) (void)hipFree(q);
      s.error = std::string("track tables: ") + hipGetErrorString(e);
      return -1;
    }
  }
  for (const double* p : {s.track.curvature, s.track.bound_left, s.track.bound_right, s.track.vel})
    if (p) (void)hipFree(const_cast<double*>(p));
  s.track.L = L_;
  s.track.M = M_;
  s.track.curvature = t[0];
  s.track.bound_left = t[1];
  s.track.bound_right = t[2];
  s.track.vel = t[3];
  return 0;
}

int ShardedSolver::do_safe_set(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  SH_LMPC(s, lmpc_set_safe_set(s.h, l_n_, l_npts_, l_x_, l_L_));
  return 0;
}

int ShardedSolver::do_regression(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  SH_LMPC(s, lmpc_set_regression_laps(s.h, l_n_, l_npts_, l_x_, l_u_, l_k_, l_t_, l_spec_));
  return 0;
}

int ShardedSolver::do_prepare(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  hipStream_t st = static_cast<hipStream_t>(s.stream);
  const std::size_t N = static_cast<std::size_t>(cfg_.N), b = static_cast<std::size_t>(b_), total = b * shards_.size(),
                    off = b * static_cast<std::size_t>(s.rank), NB = N * b, SB = (N - 1) * b;
  // the shard's contiguous slice of every component row of the whole batch
  SH_HIP(s, hipMemcpy2DAsync(s.x_ic, b * sizeof(double), p_x_ + off, total * sizeof(double), b * sizeof(double), 6, hipMemcpyHostToDevice, st));
  SH_HIP(s, hipMemcpy2DAsync(s.u_ic, b * sizeof(double), p_u_ + off, total * sizeof(double), b * sizeof(double), 2, hipMemcpyHostToDevice, st));
  SH_LMPC(s, lmpc_prepare_batch(s.h, b_, &s.track, s.x_ic, p_dt_, p_scale_, p_limit_, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref));
  SH_HIP(s, hipStreamSynchronize(st));
  if (cfg_.learning) {
    // the safe-set query of every problem: (s, e_y) of the last reference knot, the abscissa aligned to x_ic modulo L
    // (racing_mpc.cpp:219-223, 249-254; lmpc_utils/utils.hpp:35-41: k = |s2 - s1| + L/2, l = k - fmod(k, L), s1 + l sign(s2 - s1))
    std::vector<double> last(2 * b), q(2 * b);
    SH_HIP(s, hipMemcpy(last.data(), s.X_ref + (N - 1) * b, b * sizeof(double), hipMemcpyDeviceToHost));
    SH_HIP(s, hipMemcpy(last.data() + b, s.X_ref + (N + N - 1) * b, b * sizeof(double), hipMemcpyDeviceToHost));
    const double L = s.track.L;
    for (std::size_t j = 0; j < b; ++j) {
      const double s1 = last[j], s2 = p_x_[off + j], k = std::fabs(s2 - s1) + L / 2, l = k - std::fmod(k, L);
      q[j] = s1 + l * ((s2 > s1) - (s2 < s1));
      q[b + j] = last[b + j];
    }
    SH_HIP(s, hipMemcpy(s.query, q.data(), 2 * b * sizeof(double), hipMemcpyHostToDevice));
  }
  if (prec_ == PRECISION_F32) {  // lmpc_solve_batch_f32 takes every array in float: converted once, here (prepare is set-up, not a step)
    const double* src[9] = {s.x_ic, s.u_ic, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref};
    const std::size_t n[9] = {6 * b, 2 * b, 6 * NB, 2 * SB, SB, NB, NB, NB, NB};
    std::vector<double> hd;
    std::vector<float> hf;
    std::size_t o = 0;
    for (int a = 0; a < 9; ++a) {
      hd.resize(n[a]);
      hf.resize(n[a]);
      SH_HIP(s, hipMemcpy(hd.data(), src[a], n[a] * sizeof(double), hipMemcpyDeviceToHost));
      for (std::size_t e = 0; e < n[a]; ++e) hf[e] = static_cast<float>(hd[e]);
      SH_HIP(s, hipMemcpy(s.f32 + o, hf.data(), n[a] * sizeof(float), hipMemcpyHostToDevice));
      o += n[a];
    }
  }
  return 0;
}

// one solve of the shard's slice, enqueued on its stream
int ShardedSolver::launch_solve(Shard& s) {
  const std::size_t N = static_cast<std::size_t>(cfg_.N), b = static_cast<std::size_t>(b_), NB = N * b, SB = (N - 1) * b;
  int32_t *status = s.rec_i, *iters = s.rec_i + b;
  if (prec_ == PRECISION_F32) {
    float *X = static_cast<float*>(s.rec_d), *U = X + 6 * NB, *dU = U + 2 * SB;
    const float *x_ic = s.f32, *u_ic = x_ic + 6 * b, *X_ref = u_ic + 2 * b, *U_ref = X_ref + 6 * NB, *T_ref = U_ref + 2 * SB, *bl = T_ref + SB,
                *br = bl + NB, *kap = br + NB, *vref = kap + NB;
    SH_LMPC(s, lmpc_solve_batch_f32(s.h, b_, x_ic, u_ic, X_ref, U_ref, T_ref, bl, br, kap, vref, X, U, dU, status, iters, nullptr));
    return 0;
  }
  double *X = static_cast<double*>(s.rec_d), *U = X + 6 * NB, *dU = U + 2 * SB;
  if (cfg_.learning) {  // the safe set by reference: S int32 codes per problem instead of 7 S doubles
    SH_LMPC(s, lmpc_ss_query_idx_batch(s.h, b_, s.query, s.ss_idx, s.n_found));
    SH_LMPC(s, lmpc_solve_batch_ss_idx(s.h, b_, prec_ == PRECISION_MIXED ? LMPC_PRECISION_MIXED : LMPC_PRECISION_F64, s.x_ic, s.u_ic, s.X_ref, s.U_ref,
                                       s.T_ref, s.bl, s.br, s.kap, s.vref, s.track.L, s.ss_idx, X, U, dU, nullptr, status, iters, nullptr));
  } else if (prec_ == PRECISION_MIXED) {
    SH_LMPC(s, lmpc_solve_batch_mixed(s.h, b_, s.x_ic, s.u_ic, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref, s.track.L, nullptr, nullptr, X, U,
                                      dU, nullptr, status, iters, nullptr));
  } else {
    SH_LMPC(s, lmpc_solve_batch(s.h, b_, s.x_ic, s.u_ic, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref, s.track.L, nullptr, nullptr, X, U, dU,
                                nullptr, status, iters, nullptr));
  }
  return 0;
}

int ShardedSolver::do_solve(Shard& s) {
  // (with GATHER_RCCL every worker must reach every agree(): nothing returns early before the first one)
  int rc0 = 0;
  if (hipSetDevice(s.device) != hipSuccess) {
    s.error = "hipSetDevice failed";
    rc0 = -1;
    if (gather_ != GATHER_RCCL) return -1;
  }
  hipStream_t st = static_cast<hipStream_t>(s.stream);
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps_; ++k) {
    int rc = rc0 != 0 ? rc0 : launch_solve(s);
    if (gather_ == GATHER_RCCL) {
      // No rank enters the collective unless every rank does (a shard whose solve failed used to return before its
      // all-gather, and the others then blocked in hipStreamSynchronize for good -- and the destructor behind them in join()).
      ncclComm_t comm = static_cast<ncclComm_t>(s.comm);
      if (!agree(rc == 0)) {
        if (rc == 0) s.error = "another shard's solve failed before the gather; nothing was gathered";
        return -1;
      }
      auto gather = [&]() -> int {
        SH_NCCL(s, ncclGroupStart());
        SH_NCCL(s, ncclAllGather(s.rec_d, s.all_d, rec_d_, prec_ == PRECISION_F32 ? ncclFloat : ncclDouble, comm, st));
        SH_NCCL(s, ncclAllGather(s.rec_i, s.all_i, rec_i_, ncclInt32, comm, st));
        SH_NCCL(s, ncclGroupEnd());
        return 0;
      };
      rc = gather();
      if (!agree(rc == 0)) {
        // a rank could not enqueue its share: the others' collectives would never complete.  Abort every communicator (that
        // also ends the kernels already waiting in it); the object is poisoned by issue().
        if (rc == 0) s.error = "another shard could not enqueue its all-gather; the communicators were aborted";
        (void)ncclCommAbort(comm);
        s.comm = nullptr;
        return -1;
      }
    } else {
      if (rc != 0) return -1;
      if (gather_ == GATHER_COPY) {
        Shard& root = shards_[0];
        SH_HIP(s, hipMemcpyPeerAsync(static_cast<char*>(root.all_d) + rec_d_ * elem_ * static_cast<std::size_t>(s.rank), root.device, s.rec_d, s.device,
                                     rec_d_ * elem_, st));
        SH_HIP(s, hipMemcpyPeerAsync(root.all_i + rec_i_ * static_cast<std::size_t>(s.rank), root.device, s.rec_i, s.device, rec_i_ * sizeof(int32_t), st));
      }
    }
  }
  SH_HIP(s, hipStreamSynchronize(st));
  s.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

void ShardedSolver::set_track(double L, int32_t M, const double* curvature, const double* bound_left, const double* bound_right, const double* vel) {
  L_ = L;
  M_ = M;
  t_kap_ = curvature;
  t_bl_ = bound_left;
  t_br_ = bound_right;
  t_vel_ = vel;
  issue(CMD_TRACK);
}

void ShardedSolver::set_safe_set(int32_t n_laps, const int32_t* n_pts, const double* x, double total_length) {
  if (!cfg_.learning) throw std::runtime_error("ShardedSolver::set_safe_set: the configuration is the tracking problem (learning = 0)");
  l_n_ = n_laps;
  l_npts_ = n_pts;
  l_x_ = x;
  l_L_ = total_length;
  issue(CMD_SAFE_SET);
}

void ShardedSolver::set_regression_laps(int32_t n_laps, const int32_t* n_pts, const double* x, const double* u, const double* k, const double* t,
                                        const lmpc_regression_spec* spec) {
  if (prec_ == PRECISION_F32) throw std::runtime_error("ShardedSolver::set_regression_laps: the regression is applied in double precision (f64 / mixed)");
  l_n_ = n_laps;
  l_npts_ = n_pts;
  l_x_ = x;
  l_u_ = u;
  l_k_ = k;
  l_t_ = t;
  l_spec_ = spec;
  issue(CMD_REGRESSION);
}

void ShardedSolver::prepare(const double* x_ic, const double* u_ic, double dt, double speed_scale, double speed_limit) {
  p_x_ = x_ic;
  p_u_ = u_ic;
  p_dt_ = dt;
  p_scale_ = speed_scale;
  p_limit_ = speed_limit;
  issue(CMD_PREPARE);
}

void ShardedSolver::solve(double* wall_ms) {
  steps_ = 1;
  const auto t0 = std::chrono::steady_clock::now();
  issue(CMD_SOLVE);
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

double ShardedSolver::solve_many(int steps) {
  steps_ = steps < 1 ? 1 : steps;
  issue(CMD_SOLVE);
  steps_ = 1;
  double worst = 0.0;
  for (const Shard& s : shards_) worst = s.ms > worst ? s.ms : worst;
  return worst;
}

Precision ShardedSolver::last_solve_precision() const {
  int32_t p = LMPC_PRECISION_F64;
  if (lmpc_last_solve_precision(shards_.at(0).h, &p) != LMPC_OK) throw std::runtime_error("lmpc_last_solve_precision failed");
  return static_cast<Precision>(p);
}

namespace {
// device values (double or float) -> host doubles (floats widened exactly)
bool read_values(const void* dev, std::size_t n, std::size_t elem, double* out) {
  if (elem == sizeof(double)) return hipMemcpy(out, dev, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  std::vector<float> f(n);
  if (hipMemcpy(f.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return false;
  for (std::size_t e = 0; e < n; ++e) out[e] = static_cast<double>(f[e]);
  return true;
}
}  // namespace

void ShardedSolver::fetch_own(int shard, std::vector<double>& record_d, std::vector<int32_t>& record_i) {
  Shard& s = shards_.at(static_cast<std::size_t>(shard));
  record_d.resize(rec_d_);
  record_i.resize(rec_i_);
  if (hipSetDevice(s.device) != hipSuccess || !read_values(s.rec_d, rec_d_, elem_, record_d.data()) ||
      hipMemcpy(record_i.data(), s.rec_i, rec_i_ * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
    throw std::runtime_error("ShardedSolver::fetch_own: copy failed");
}

void ShardedSolver::fetch(std::vector<double>& record_d, std::vector<int32_t>& record_i, int from_shard) {
  const std::size_t n = shards_.size();
  record_d.resize(rec_d_ * n);
  record_i.resize(rec_i_ * n);
  if (gather_ == GATHER_NONE) {
    std::vector<double> d;
    std::vector<int32_t> i;
    for (std::size_t r = 0; r < n; ++r) {
      fetch_own(static_cast<int>(r), d, i);
      std::copy(d.begin(), d.end(), record_d.begin() + static_cast<std::ptrdiff_t>(rec_d_ * r));
      std::copy(i.begin(), i.end(), record_i.begin() + static_cast<std::ptrdiff_t>(rec_i_ * r));
    }
    return;
  }
  Shard& s = shards_.at(gather_ == GATHER_COPY ? 0 : static_cast<std::size_t>(from_shard));
  if (hipSetDevice(s.device) != hipSuccess || !read_values(s.all_d, record_d.size(), elem_, record_d.data()) ||
      hipMemcpy(record_i.data(), s.all_i, record_i.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
    throw std::runtime_error("ShardedSolver::fetch: copy failed");
}

}  // namespace mpc
}  // namespace lmpc
