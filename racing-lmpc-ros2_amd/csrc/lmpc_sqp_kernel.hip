// lmpc_sqp_kernel.hip -- globalisation of the sequential-QP solve of the NONLINEAR-dynamics problem.
//
// What it replaces: RacingMPC(full_dynamics = true)::solve, i.e. IPOPT on the problem whose dynamics rows are
// x_{i+1} = f_d(x_i, u_i, k_i, t_i) instead of their linearisation (racing_mpc.cpp:67-84 solver choice, :162-166 the
// rows; used once by the node for its very first solve, racing_mpc_node.cpp:299-314).  Cost and every other row are
// those of the QP (racing_mpc.cpp:442-543), so the NLP differs from the QP in the dynamics equalities only.
//
// The outer loop lives in lmpc_solve_full_dynamics_batch (lmpc_capi.hip): linearise about the iterate, solve the QP with
// the batched kernels, then this kernel takes the step -- one THREAD per problem (the batch is on the lanes, every
// access to the [field][knot][batch] arrays is coalesced):
//   merit    phi(w) = J(w) + nu |c(w)|_1,   c_i = (x_{i+1} - f_d(x_i, u_i, k_i, t_i)) / scale_x   (the l1 exact penalty)
//   J        the QP's own cost with the boundary slack eliminated, sigma*(w) = largest boundary violation (>= 0); for
//            the learning problem  ss_j' lambda + eps' D eps  with eps = x_T - SS lambda  (racing_mpc.cpp:496-504)
//   nu       raised whenever needed so that the step d = w_QP - w is a descent direction:  nu >= dJ / (0.9 |c|_1),
//            dJ = J(w_QP) - J(w) >= J'(w; d) (J is convex)
//   Armijo   phi(w + a d) <= phi(w) + 1e-4 a (dJ - nu |c(w)|_1),   a = 1, 1/2, ... , 2^-7
// Every row of the NLP other than the dynamics is LINEAR and holds at w_QP; it holds at the iterate too from the first
// full step on, hence along the whole segment -- the line search only has to look at cost and defect.  The very first
// step is taken in full (the cold-start iterate is a zero-input rollout: c = 0, but outside the boxes).
//   back-off when the QP about the new iterate is infeasible (its linearised dynamics cannot meet the hard boxes: the step
//            went too far for the linearisation that proposed it) the iterate is moved half way back to the previous one
//            and linearised again, up to LMPC_SQP_BACKOFF times in a row -- the role of IPOPT's restoration phase upstream.
//            Only after that does the problem stop with the QP's status.
//   move     the convergence measure is the size of the QP's own step |w_QP - w| (scaled), not of the shortened step
//            taken: a collapsed line search cannot pass for convergence.
#include <hip/hip_runtime.h>

#include "lmpc_device.h"
#include "lmpc_dynamics.hip.h"

#define LMPC_SQP_BACKOFF 6

struct lmpc_sqp_arrays {
  // iterate (updated in place) and QP solution, [field][knot][batch]
  double *X, *U, *dU, *lam;
  const double *Xq, *Uq, *dUq, *lamq;
  const int* status_q;
  // problem data
  const double *T_ref, *curv, *bl, *br, *vref, *ss_x, *ss_j;
  // the iterate before the last step taken (for the back-off), same layout as the iterate
  double *Xp, *Up, *dUp, *lamp;
  int* backoffs;   // consecutive back-offs (in/out)
  const int* iters_q;  // interior-point iterations of this pass's QP
  int* iters;          // their sum over the QPs this problem took part in (in/out)
  // per problem
  double* nu;      // penalty weight (in/out)
  int* active;     // 1 while the problem is still iterating (in/out)
  int* status;     // status of the last QP taken into the iterate (out)
  int* sqp_iters;  // QPs solved for this problem: steps taken + back-offs (in/out)
  double* move;    // largest scaled |X_QP - X| of the last QP (out): the step proposed, whatever part of it was taken
  double* defect;  // |c|_inf of the iterate after the step (out)
  int* n_active;   // device counter of problems still active after this step (atomic)
};

namespace {

struct sqp_point {  // w(a) = cur + a (q - cur), read on the fly
  const lmpc_sqp_arrays& A;
  int B, b;
  double a;
  __device__ double blend(const double* cur, const double* q, size_t e) const {
    const double c = cur[e * B + b];
    return c + a * (q[e * B + b] - c);
  }
};

// cost J and l1 defect of w(a)
__device__ void sqp_eval(const lmpc_params& P, const lmpc_sqp_arrays& A, int B, int b, double a, double& J, double& c1,
                         double& cinf) {
  const int N = P.N, NS = N - 1, S = P.S;
  const sqp_point w{A, B, b, a};
  const double isc[6] = {1.0 / 2000.0, 1.0 / 10.0, 1.0 / 0.1, 1.0 / 80.0, 1.0 / 2.0, 1.0 / 2.0};  // racing_mpc.cpp:36
  double x[6], xn[6], u[2], viol = 0.0;
  J = 0.0;
  c1 = 0.0;
  cinf = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) x[k] = w.blend(A.X, A.Xq, (size_t)k * N);
  for (int i = 0; i < N; ++i) {
    const bool last = i == N - 1;
    if (!P.learning) {  // tracking terms (racing_mpc.cpp:459-476): Qd = 2 q, qv = -2 q_vel (10x at the terminal knot)
#pragma unroll
      for (int k = 0; k < 6; ++k) J += 0.5 * (last ? P.Qt[k] : P.Qd[k]) * x[k] * x[k];
      J += (last ? P.qv_term : P.qv_stage) * A.vref[(size_t)i * B + b] * x[3];
    }
    viol = fmax(viol, fmax(x[1] - (A.bl[(size_t)i * B + b] - P.marg), (A.br[(size_t)i * B + b] + P.marg) - x[1]));
    if (last) break;
#pragma unroll
    for (int k = 0; k < 2; ++k) u[k] = w.blend(A.U, A.Uq, (size_t)k * NS + i);
    const double v0 = w.blend(A.dU, A.dUq, (size_t)i), v1 = w.blend(A.dU, A.dUq, (size_t)NS + i);
    J += 0.5 * (P.Qu[0] * u[0] * u[0] + (P.Qu[1] + P.Qu[2]) * u[0] * u[1] + P.Qu[3] * u[1] * u[1]);
    J += 0.5 * (P.Sv[0] * v0 * v0 + (P.Sv[1] + P.Sv[2]) * v0 * v1 + P.Sv[3] * v1 * v1);
    lmpc_fd(P.veh, x, u, A.curv[(size_t)i * B + b], A.T_ref[(size_t)i * B + b], xn);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      x[k] = w.blend(A.X, A.Xq, (size_t)k * N + i + 1);
      const double c = fabs(x[k] - xn[k]) * isc[k];
      c1 += c;
      cinf = fmax(cinf, c);
    }
  }
  if (P.has_sigma) J += 0.5 * P.qsig * viol * viol;  // sigma* = the largest violation (0 inside the track)
  if (P.learning) {  // ss_j' lambda + eps' D eps, eps = x_T - SS lambda (x holds the terminal state)
    double eps[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
    for (int j = 0; j < S; ++j) {
      const double l = w.blend(A.lam, A.lamq, (size_t)j);
      J += A.ss_j[(size_t)j * B + b] * l;
#pragma unroll
      for (int k = 0; k < 6; ++k) eps[k] -= A.ss_x[((size_t)k * S + j) * B + b] * l;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) J += 0.5 * P.chs2[k] * eps[k] * eps[k];
  }
}

}  // namespace

__global__ __launch_bounds__(64) void lmpc_sqp_linesearch_kernel(lmpc_params P, int B, lmpc_sqp_arrays A, int first, double step_tol) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || !A.active[b]) return;
  const int N = P.N, NS = N - 1, S = P.S;
  A.status[b] = A.status_q[b];
  A.iters[b] += A.iters_q[b];
  A.sqp_iters[b] += 1;
  if (A.status_q[b] != LMPC_SOLVE_OPTIMAL) {
    if (!first && A.backoffs[b] < LMPC_SQP_BACKOFF) {  // half way back to the iterate the last step started from
      A.backoffs[b] += 1;
      for (int e = 0; e < 6 * N; ++e) A.X[(size_t)e * B + b] = 0.5 * (A.X[(size_t)e * B + b] + A.Xp[(size_t)e * B + b]);
      for (int e = 0; e < 2 * NS; ++e) {
        A.U[(size_t)e * B + b] = 0.5 * (A.U[(size_t)e * B + b] + A.Up[(size_t)e * B + b]);
        A.dU[(size_t)e * B + b] = 0.5 * (A.dU[(size_t)e * B + b] + A.dUp[(size_t)e * B + b]);
      }
      for (int j = 0; j < S; ++j) A.lam[(size_t)j * B + b] = 0.5 * (A.lam[(size_t)j * B + b] + A.lamp[(size_t)j * B + b]);
      atomicAdd(A.n_active, 1);
      return;
    }
    A.active[b] = 0;  // the problem keeps its last iterate and reports the QP's status
    return;
  }
  A.backoffs[b] = 0;
  double J0, c0, ci0, J1, c1, ci1;
  sqp_eval(P, A, B, b, 0.0, J0, c0, ci0);
  sqp_eval(P, A, B, b, 1.0, J1, c1, ci1);
  double a = 1.0, cinf = ci1;
  if (!first) {
    const double dJ = J1 - J0;
    double nu = A.nu[b];
    if (c0 > 0.0 && dJ > 0.0) nu = fmax(nu, dJ / (0.9 * c0));
    nu = fmax(nu, 1e-3);
    A.nu[b] = nu;
    const double D = dJ - nu * c0, phi0 = J0 + nu * c0;
    double Ja = J1, ca = c1;
    for (int t = 0; t < 8; ++t) {
      if (Ja + nu * ca <= phi0 + 1e-4 * a * D + 1e-14 * (1.0 + fabs(phi0))) break;
      if (t == 7) break;
      a *= 0.5;
      sqp_eval(P, A, B, b, a, Ja, ca, cinf);
    }
  }
  // take the step
  const double isc[6] = {1.0 / 2000.0, 1.0 / 10.0, 1.0 / 0.1, 1.0 / 80.0, 1.0 / 2.0, 1.0 / 2.0};
  double mv = 0.0;
  for (int e = 0; e < 6 * N; ++e) {
    const double c = A.X[(size_t)e * B + b], d = A.Xq[(size_t)e * B + b] - c;
    mv = fmax(mv, fabs(d) * isc[e / N]);
    A.Xp[(size_t)e * B + b] = c;
    A.X[(size_t)e * B + b] = c + a * d;
  }
  for (int e = 0; e < 2 * NS; ++e) {
    const double c = A.U[(size_t)e * B + b];
    A.Up[(size_t)e * B + b] = c;
    A.U[(size_t)e * B + b] = c + a * (A.Uq[(size_t)e * B + b] - c);
    const double v = A.dU[(size_t)e * B + b];
    A.dUp[(size_t)e * B + b] = v;
    A.dU[(size_t)e * B + b] = v + a * (A.dUq[(size_t)e * B + b] - v);
  }
  for (int j = 0; j < S; ++j) {
    const double l = A.lam[(size_t)j * B + b];
    A.lamp[(size_t)j * B + b] = l;
    A.lam[(size_t)j * B + b] = l + a * (A.lamq[(size_t)j * B + b] - l);
  }
  A.move[b] = mv;
  A.defect[b] = cinf;
  const int still = mv > step_tol ? 1 : 0;
  A.active[b] = still;
  if (still) atomicAdd(A.n_active, 1);
}

