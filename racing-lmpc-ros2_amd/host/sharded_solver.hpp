// sharded_solver.hpp -- the batched solve sharded over the GPUs of one node, on the C++ side of the C ABI
// (BASELINE.json north_star: "Batches shard embarrassingly across the 8 GPUs of one node (RCCL over
// xGMI only to gather results)").  No reference counterpart: the reference has no device or collective code.
//
//   * one lmpc_handle + one HIP stream + one host thread per shard, each on its own device (`devices[r]`); the threads are
//     persistent workers, a step is one wake-up;
//   * problems are cut into contiguous slices of `shard_batch`; vehicle / MPC parameters, track tables and the safe set are
//     replicated (read-only, < 1 MB);
//   * NO data-path collective: a shard's solve touches nothing outside its device;
//   * results are gathered after the solve, on the shard's own stream:
//       GATHER_RCCL   ncclAllGather over the node's xGMI links -- every device ends up with every shard's record
//                     (needs distinct devices: one communicator rank per GPU, ncclCommInitAll);
//       GATHER_COPY   hipMemcpyPeerAsync into shard 0's buffer (works with several shards on ONE device, which is how the
//                     path is exercised on a single-GPU box; on a real node it is the root-gather variant);
//       GATHER_NONE   results stay where they were computed.
//     The gathered double record of shard r is [X_optm 6 N b | U_optm 2 (N-1) b | dU_optm 2 (N-1) b] (each [comp][knot][b], the
//     C ABI's layout for the shard's b problems), the int32 record [status b | iters b]; records are laid end to end in shard
//     order, so problem p of the whole batch is entry p % b of record p / b.
// Plain C++17 + HIP runtime + RCCL; no torch, no Python.  bench_cabi --gpus N drives it.
#ifndef LMPC_HOST_SHARDED_SOLVER_HPP_
#define LMPC_HOST_SHARDED_SOLVER_HPP_

#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "lmpc_hip.h"

namespace lmpc {
namespace mpc {

enum GatherMode { GATHER_NONE = 0, GATHER_COPY = 1, GATHER_RCCL = 2 };

class ShardedSolver {
 public:
  // Throws std::runtime_error with the library's / runtime's message.  `devices` may repeat a device (GATHER_RCCL then throws).
  ShardedSolver(const lmpc_config& cfg, const lmpc_vehicle& veh, const std::vector<int>& devices, int32_t shard_batch,
                GatherMode gather);
  ~ShardedSolver();
  ShardedSolver(const ShardedSolver&) = delete;
  ShardedSolver& operator=(const ShardedSolver&) = delete;

  int n_shards() const { return static_cast<int>(shards_.size()); }
  int32_t shard_batch() const { return b_; }
  int64_t total_batch() const { return static_cast<int64_t>(b_) * n_shards(); }
  std::size_t record_doubles() const { return rec_d_; }  // per shard
  std::size_t record_ints() const { return rec_i_; }

  // HOST tables of the closed track (lmpc_track's meaning), replicated to every device
  void set_track(double L, int32_t M, const double* curvature, const double* bound_left, const double* bound_right, const double* vel);
  // HOST x_ic [6][total], u_ic [2][total] (batch axis fastest over the WHOLE batch): each shard takes its slice and runs the
  // node's cold-start preparation on its device (lmpc_prepare_batch)
  void prepare(const double* x_ic, const double* u_ic, double dt, double speed_scale, double speed_limit);
  // one step on every shard: lmpc_solve_batch on the slice + the gather; returns when every shard's stream has drained.
  // `wall_ms` (optional): host wall-clock of the step, from the wake-up to the last shard's completion
  void solve(double* wall_ms = nullptr);
  // `steps` back-to-back steps per shard without a host rendezvous in between (each worker queues its launches and waits
  // once): the throughput loop of a benchmark.  Returns the wall-clock of the slowest shard in ms.
  double solve_many(int steps);

  // results of the last step, copied to the host: the whole batch in shard order.  With a gather mode they are read from
  // ONE device's gathered buffer (`from_shard`; GATHER_COPY: shard 0 only), without from each shard's own record.
  void fetch(std::vector<double>& record_d, std::vector<int32_t>& record_i, int from_shard = 0);
  // a shard's own (ungathered) record, for checking the gather against
  void fetch_own(int shard, std::vector<double>& record_d, std::vector<int32_t>& record_i);

 private:
  struct Shard {
    int rank = 0, device = 0;
    lmpc_handle* h = nullptr;
    void* stream = nullptr;  // hipStream_t
    void* comm = nullptr;    // ncclComm_t
    lmpc_track track{};
    double *x_ic = nullptr, *u_ic = nullptr, *X_ref = nullptr, *U_ref = nullptr, *T_ref = nullptr, *bl = nullptr, *br = nullptr,
           *kap = nullptr, *vref = nullptr;
    double* rec_d = nullptr;    // [X | U | dU]
    int32_t* rec_i = nullptr;   // [status | iters]
    double* all_d = nullptr;    // gathered (RCCL: every shard; COPY: shard 0)
    int32_t* all_i = nullptr;
    std::thread worker;
    std::string error;
    double ms = 0.0;
  };
  enum Command { CMD_NONE, CMD_INIT, CMD_TRACK, CMD_PREPARE, CMD_SOLVE, CMD_QUIT };
  void run(Shard& s);
  void issue(Command c);  // wake every worker with `c`, wait for all, throw the first error
  void quit();            // CMD_QUIT + join (destructor, and a constructor that is about to throw)
  int do_init(Shard& s);
  int do_track(Shard& s);
  int do_prepare(Shard& s);
  int do_solve(Shard& s);

  lmpc_config cfg_;
  lmpc_vehicle veh_;
  int32_t b_;
  GatherMode gather_;
  std::size_t rec_d_, rec_i_;
  std::vector<Shard> shards_;
  // command hand-over
  std::mutex mu_;
  std::condition_variable cv_go_, cv_done_;
  Command cmd_ = CMD_NONE;
  uint64_t generation_ = 0;
  int pending_ = 0;
  // arguments of the command in flight (host pointers; valid for the duration of issue())
  double L_ = 0.0;
  int32_t M_ = 0;
  const double *t_kap_ = nullptr, *t_bl_ = nullptr, *t_br_ = nullptr, *t_vel_ = nullptr, *p_x_ = nullptr, *p_u_ = nullptr;
  double p_dt_ = 0.0, p_scale_ = 1.0, p_limit_ = 0.0;
  int steps_ = 1;
};

}  // namespace mpc
}  // namespace lmpc
#endif
