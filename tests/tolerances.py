"""Stated parity tolerances (scaled units: X / scale_x, U / scale_u, dU / scale_u with the
reference's scale vectors, racing_mpc.cpp:36-37).

Contract: fp64 results within 1e-6 (scaled) of the optimum of the reference's QP in
X, U and dU -- EVERY problem, degenerate or not.  The optimum is the dense, polished, KKT-certified solution of
oracle/qp.py (the reference itself runs OSQP at eps = 1e-3 with polish = true and holds no golden outputs).

Until round 3 problems whose dense multipliers show a strict-complementarity margin below 1e-4 (16 % of the bench's
cold-start distribution at N = 20, 33 % at N = 80: typically an input on its box and its rate limit at once) were held
to a relaxed bound, because an interior-point iterate is O(sqrt(mu)) from such an optimum.  The kernel and its twin now
finish with an active-set polish (the role of OSQP's polish = true: the equality-constrained QP of the rows the interior
point holds, KKT-verified, csrc/lmpc_solve_kernel.hip `polish_attempt`): a weakly active row may sit on either side of
the guess, the solution is the same.  Measured on the unclipped cold starts against the dense optimum: worst 5e-10 at
N = 20, 3e-10 at N = 60 (256 problems each), 1e-11 on the IAC problem, 5e-13 on the learning problem.  The margin is
still computed by the fixtures and reported, but no tolerance depends on it any more.

HIP kernel vs its serial C twin (identical algorithm; FMA contraction and summation order differ): both are within
TOL_XU of the same optimum; iteration counts (interior-point iterations + polish rounds) equal on >= 90 % of problems.

Single precision / mixed precision (lmpc_solve_batch_f32, lmpc_solve_batch_mixed): TOL_F32 on every problem against the
fp64 answer.
"""
TOL_XU = 1e-6
TOL_DU = 1e-6
TOL_MEDIAN = 1e-8
TOL_TWIN = 1e-6
TOL_F32 = 1e-3
TOL_LINEARIZE_REL = 1e-11
