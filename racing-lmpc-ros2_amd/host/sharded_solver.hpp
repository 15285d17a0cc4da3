// sharded_solver.hpp -- the batched solve sharded over the GPUs of one node, on the C++ side of the C ABI
// (BASELINE.json north_star: "Batches shard embarrassingly across the 8 GPUs of one node (RCCL over
// xGMI only to gather results)").  No reference counterpart: the reference has no device or collective code.
//
//   * one lmpc_handle + one HIP stream + one host thread per shard, each on its own device (`devices[r]`); the threads are
//     persistent workers, a step is one wake-up;
//   * problems are cut into contiguous slices of `shard_batch`; vehicle / MPC parameters, track tables, the safe set
//     (set_safe_set) and the regression's lap samples (set_regression_laps) are replicated to every handle (read-only, < 1 MB);
//   * every entry point the BASELINE configs name: PRECISION_F64 (lmpc_solve_batch), PRECISION_F32 (lmpc_solve_batch_f32: configs[3],
//     "IAC Putnam tracking MPC, N=40, batch=65536, fp32, sharded across 8 GPUs" -- float arrays and float records) and
//     PRECISION_MIXED (lmpc_solve_batch_mixed: configs[4], "LMPC + error-dynamics residual term, batch=262144, mixed fp32/fp64
//     KKT"); a learning handle solves with the safe set BY REFERENCE: lmpc_ss_query_idx_batch + lmpc_solve_batch_ss_idx per step;
//   * NO data-path collective: a shard's solve touches nothing outside its device;
//   * results are gathered after the solve, on the shard's own stream:
//       GATHER_RCCL   ncclAllGather over the node's xGMI links -- every device ends up with every shard's record
//                     (needs distinct devices: one communicator rank per GPU, ncclCommInitAll);
//       GATHER_COPY   hipMemcpyPeerAsync into shard 0's buffer (works with several shards on ONE device, which is how the
//                     path is exercised on a single-GPU box; on a real node it is the root-gather variant);
//       GATHER_NONE   results stay where they were computed.
//     The gathered value record of shard r is [X_optm 6 N b | U_optm 2 (N-1) b | dU_optm 2 (N-1) b] (each [comp][knot][b], the
//     C ABI's layout for the shard's b problems; doubles, or floats under PRECISION_F32), the int32 record [status b | iters b];
//     records are laid end to end in shard order, so problem p of the whole batch is entry p % b of record p / b.
//   * a failure on one shard never leaves the others inside a collective: with GATHER_RCCL the workers agree on a
//     host barrier that every shard's solve was launched before any of them enqueues its all-gather, and again that every
//     all-gather was enqueued before any of them waits on its stream; a failure aborts the communicators (ncclCommAbort), the
//     call throws the first error, and the object is POISONED: every later call throws at once, the destructor does not block.
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
enum Precision { PRECISION_F64 = LMPC_PRECISION_F64, PRECISION_F32 = LMPC_PRECISION_F32, PRECISION_MIXED = LMPC_PRECISION_MIXED };

class ShardedSolver {
 public:
  // Throws std::runtime_error with the library's / runtime's message.  `devices` may repeat a device (GATHER_RCCL then throws).
  // PRECISION_F32 is the tracking problem only (as lmpc_solve_batch_f32).
  ShardedSolver(const lmpc_config& cfg, const lmpc_vehicle& veh, const std::vector<int>& devices, int32_t shard_batch,
                GatherMode gather, Precision precision = PRECISION_F64);
  ~ShardedSolver();
  ShardedSolver(const ShardedSolver&) = delete;
  ShardedSolver& operator=(const ShardedSolver&) = delete;

  int n_shards() const { return static_cast<int>(shards_.size()); }
  int32_t shard_batch() const { return b_; }
  int64_t total_batch() const { return static_cast<int64_t>(b_) * n_shards(); }
  std::size_t record_values() const { return rec_d_; }  // per shard: 6 N b + 4 (N - 1) b values (double, or float under PRECISION_F32)
  std::size_t record_doubles() const { return rec_d_; }  // (the name of rounds 1 - 5)
  std::size_t record_ints() const { return rec_i_; }
  Precision precision() const { return prec_; }
  bool poisoned() const { return poisoned_; }

  // HOST tables of the closed track (lmpc_track's meaning), replicated to every device
  void set_track(double L, int32_t M, const double* curvature, const double* bound_left, const double* bound_right, const double* vel);
  // lmpc_set_safe_set on every shard's handle (HOST pointers; SafeSetManager's laps, oldest first): learning handles
  void set_safe_set(int32_t n_laps, const int32_t* n_pts, const double* x, double total_length);
  // lmpc_set_regression_laps on every shard's handle (HOST pointers; n_laps = 0 switches it off)
  void set_regression_laps(int32_t n_laps, const int32_t* n_pts, const double* x, const double* u, const double* k, const double* t,
                           const lmpc_regression_spec* spec);
  // HOST x_ic [6][total], u_ic [2][total] (batch axis fastest over the WHOLE batch): each shard takes its slice and runs the
  // node's cold-start preparation on its device (lmpc_prepare_batch); a learning handle also forms its safe-set query points
  // (the last reference knot, abscissa aligned to x_ic: racing_mpc.cpp:219-223, 249-254), PRECISION_F32 its float copies
  void prepare(const double* x_ic, const double* u_ic, double dt, double speed_scale, double speed_limit);
  // one step on every shard: the solve on the slice + the gather; returns when every shard's stream has drained.
  // `wall_ms` (optional): host wall-clock of the step, from the wake-up to the last shard's completion
  void solve(double* wall_ms = nullptr);
  // `steps` back-to-back steps per shard (each worker queues its launches and waits for its stream once; with GATHER_RCCL the
  // workers meet on a host barrier twice per step, microseconds): the throughput loop of a benchmark.  Returns the wall-clock of
  // the slowest shard in ms.
  double solve_many(int steps);

  // results of the last step, copied to the host: the whole batch in shard order (floats widened exactly under PRECISION_F32).
  // With a gather mode they are read from ONE device's gathered buffer (`from_shard`; GATHER_COPY: shard 0 only), without from
  // each shard's own record.
  void fetch(std::vector<double>& record_d, std::vector<int32_t>& record_i, int from_shard = 0);
  // a shard's own (ungathered) record, for checking the gather against
  void fetch_own(int shard, std::vector<double>& record_d, std::vector<int32_t>& record_i);
  // the precision the shards' last solve ran in (lmpc_last_solve_precision of shard 0: PRECISION_MIXED falls back to fp64 where the
  // library has no reduced-precision kernel for the configuration)
  Precision last_solve_precision() const;

 private:
  struct Shard {
    int rank = 0, device = 0;
    lmpc_handle* h = nullptr;
    void* stream = nullptr;  // hipStream_t
    void* comm = nullptr;    // ncclComm_t
    lmpc_track track{};
    double *x_ic = nullptr, *u_ic = nullptr, *X_ref = nullptr, *U_ref = nullptr, *T_ref = nullptr, *bl = nullptr, *br = nullptr,
           *kap = nullptr, *vref = nullptr;
    float* f32 = nullptr;        // PRECISION_F32: the nine input arrays as floats, end to end
    double* query = nullptr;     // learning: [2][b]
    int32_t* ss_idx = nullptr;   // learning: [S][b]
    int32_t* n_found = nullptr;  // learning: [b]
    void* rec_d = nullptr;       // [X | U | dU], double or float
    int32_t* rec_i = nullptr;    // [status | iters]
    void* all_d = nullptr;       // gathered (RCCL: every shard; COPY: shard 0)
    int32_t* all_i = nullptr;
    std::thread worker;
    std::string error;
    double ms = 0.0;
  };
  enum Command { CMD_NONE, CMD_INIT, CMD_TRACK, CMD_SAFE_SET, CMD_REGRESSION, CMD_PREPARE, CMD_SOLVE, CMD_QUIT };
  void run(Shard& s);
  void issue(Command c);  // wake every worker with `c`, wait for all, throw the first error
  void quit();            // CMD_QUIT + join (destructor, and a constructor that is about to throw)
  bool agree(bool ok);    // host barrier over the workers: true when every one of them brought `ok`
  int do_init(Shard& s);
  int do_track(Shard& s);
  int do_safe_set(Shard& s);
  int do_regression(Shard& s);
  int do_prepare(Shard& s);
  int do_solve(Shard& s);
  int launch_solve(Shard& s);

  lmpc_config cfg_;
  lmpc_vehicle veh_;
  int32_t b_;
  GatherMode gather_;
  Precision prec_;
  std::size_t rec_d_, rec_i_, elem_;
  std::vector<Shard> shards_;
  bool poisoned_ = false;
  // command hand-over
  std::mutex mu_;
  std::condition_variable cv_go_, cv_done_;
  Command cmd_ = CMD_NONE;
  uint64_t generation_ = 0;
  int pending_ = 0;
  // the workers' barrier (agree)
  std::mutex bar_mu_;
  std::condition_variable bar_cv_;
  int bar_count_ = 0;
  uint64_t bar_gen_ = 0;
  bool bar_ok_ = true, bar_result_ = true;
  // arguments of the command in flight (host pointers; valid for the duration of issue())
  double L_ = 0.0;
  int32_t M_ = 0;
  const double *t_kap_ = nullptr, *t_bl_ = nullptr, *t_br_ = nullptr, *t_vel_ = nullptr, *p_x_ = nullptr, *p_u_ = nullptr;
  double p_dt_ = 0.0, p_scale_ = 1.0, p_limit_ = 0.0;
  int32_t l_n_ = 0;
  const int32_t* l_npts_ = nullptr;
  const double *l_x_ = nullptr, *l_u_ = nullptr, *l_k_ = nullptr, *l_t_ = nullptr;
  const lmpc_regression_spec* l_spec_ = nullptr;
  double l_L_ = 0.0;
  int steps_ = 1;
};

}  // namespace mpc
}  // namespace lmpc
#endif
