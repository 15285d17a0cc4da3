// sharded_solver.cpp -- see sharded_solver.hpp.  HIP runtime + RCCL + the C ABI; compiled as host code by hipcc.
#include "sharded_solver.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
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
                             GatherMode gather)
    : cfg_(cfg), veh_(veh), b_(shard_batch), gather_(gather) {
  if (devices.empty() || shard_batch < 1) throw std::runtime_error("ShardedSolver: need at least one device and one problem per shard");
  if (cfg.learning) throw std::runtime_error("ShardedSolver: built for the tracking problem (the safe set would be replicated the same way)");
  const std::size_t N = static_cast<std::size_t>(cfg.N), b = static_cast<std::size_t>(shard_batch);
  rec_d_ = (6 * N + 4 * (N - 1)) * b;
  rec_i_ = 2 * b;
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
      s.worker.join();  // (the worker destroyed its communicator, handle, stream and buffers)
    else if (s.comm)    // its thread never started (std::thread's constructor threw)
      (void)ncclCommDestroy(static_cast<ncclComm_t>(s.comm));
    s.comm = nullptr;
  }
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
      case CMD_PREPARE: rc = do_prepare(s); break;
      case CMD_SOLVE: rc = do_solve(s); break;
      default: break;
    }
    if (c == CMD_QUIT) {  // release what this thread created, on its device
      (void)hipSetDevice(s.device);
      if (s.stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(s.stream));
      if (s.comm) (void)ncclCommDestroy(static_cast<ncclComm_t>(s.comm));
      if (s.h) lmpc_destroy(s.h);
      for (double* p : {s.x_ic, s.u_ic, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref, s.rec_d, s.all_d,
                        const_cast<double*>(s.track.curvature), const_cast<double*>(s.track.bound_left),
                        const_cast<double*>(s.track.bound_right), const_cast<double*>(s.track.vel)})
        if (p) (void)hipFree(p);
      if (s.rec_i) (void)hipFree(s.rec_i);
      if (s.all_i) (void)hipFree(s.all_i);
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
    if (!s.error.empty()) throw std::runtime_error("ShardedSolver shard " + std::to_string(s.rank) + " (device " + std::to_string(s.device) + "): " + s.error);
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
            dmalloc(&s.bl, NB) | dmalloc(&s.br, NB) | dmalloc(&s.kap, NB) | dmalloc(&s.vref, NB) | dmalloc(&s.rec_d, rec_d_) |
            dmalloc(&s.rec_i, rec_i_);
  const bool holds_all = gather_ == GATHER_RCCL || (gather_ == GATHER_COPY && s.rank == 0);
  if (holds_all) bad |= dmalloc(&s.all_d, rec_d_ * shards_.size()) | dmalloc(&s.all_i, rec_i_ * shards_.size());
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

int ShardedSolver::do_prepare(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  hipStream_t st = static_cast<hipStream_t>(s.stream);
  const std::size_t b = static_cast<std::size_t>(b_), total = b * shards_.size(), off = b * static_cast<std::size_t>(s.rank);
  // the shard's contiguous slice of every component row of the whole batch
  SH_HIP(s, hipMemcpy2DAsync(s.x_ic, b * sizeof(double), p_x_ + off, total * sizeof(double), b * sizeof(double), 6, hipMemcpyHostToDevice, st));
  SH_HIP(s, hipMemcpy2DAsync(s.u_ic, b * sizeof(double), p_u_ + off, total * sizeof(double), b * sizeof(double), 2, hipMemcpyHostToDevice, st));
  SH_LMPC(s, lmpc_prepare_batch(s.h, b_, &s.track, s.x_ic, p_dt_, p_scale_, p_limit_, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref));
  SH_HIP(s, hipStreamSynchronize(st));
  return 0;
}

int ShardedSolver::do_solve(Shard& s) {
  SH_HIP(s, hipSetDevice(s.device));
  hipStream_t st = static_cast<hipStream_t>(s.stream);
  const std::size_t N = static_cast<std::size_t>(cfg_.N), b = static_cast<std::size_t>(b_);
  double *X = s.rec_d, *U = X + 6 * N * b, *dU = U + 2 * (N - 1) * b;
  int32_t *status = s.rec_i, *iters = s.rec_i + b;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps_; ++k) {
    SH_LMPC(s, lmpc_solve_batch(s.h, b_, s.x_ic, s.u_ic, s.X_ref, s.U_ref, s.T_ref, s.bl, s.br, s.kap, s.vref, s.track.L, nullptr, nullptr,
                                X, U, dU, nullptr, status, iters, nullptr));
    if (gather_ == GATHER_RCCL) {
      ncclComm_t comm = static_cast<ncclComm_t>(s.comm);
      SH_NCCL(s, ncclGroupStart());
      SH_NCCL(s, ncclAllGather(s.rec_d, s.all_d, rec_d_, ncclDouble, comm, st));
      SH_NCCL(s, ncclAllGather(s.rec_i, s.all_i, rec_i_, ncclInt32, comm, st));
      SH_NCCL(s, ncclGroupEnd());
    } else if (gather_ == GATHER_COPY) {
      Shard& root = shards_[0];
      SH_HIP(s, hipMemcpyPeerAsync(root.all_d + rec_d_ * static_cast<std::size_t>(s.rank), root.device, s.rec_d, s.device, rec_d_ * sizeof(double), st));
      SH_HIP(s, hipMemcpyPeerAsync(root.all_i + rec_i_ * static_cast<std::size_t>(s.rank), root.device, s.rec_i, s.device, rec_i_ * sizeof(int32_t), st));
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

void ShardedSolver::fetch_own(int shard, std::vector<double>& record_d, std::vector<int32_t>& record_i) {
  Shard& s = shards_.at(static_cast<std::size_t>(shard));
  record_d.resize(rec_d_);
  record_i.resize(rec_i_);
  if (hipSetDevice(s.device) != hipSuccess || hipMemcpy(record_d.data(), s.rec_d, rec_d_ * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
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
  if (hipSetDevice(s.device) != hipSuccess || hipMemcpy(record_d.data(), s.all_d, record_d.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(record_i.data(), s.all_i, record_i.size() * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)
    throw std::runtime_error("ShardedSolver::fetch: copy failed");
}

}  // namespace mpc
}  // namespace lmpc
