"""Stated parity tolerances (scaled units: X / scale_x, U / scale_u, dU / scale_u with the
reference's scale vectors, racing_mpc.cpp:36-37).

The reference hands the QP to OSQP with default eps_abs = eps_rel = 1e-3 (+ polish), so 1e-3 in
scaled variables is all it guarantees (racing_mpc.cpp:86-103).  The structured interior-point
solver (C oracle and HIP kernel, same algorithm) is held to tighter figures against the dense,
polished, KKT-certified optimum:
  * X, U:  1e-4 on the golden vectors and on >= 90 % of any batch;   dU (= difference quotient of
    U over dt = 25 ms): 2e-3;   median over a batch: 1e-7
  * degenerate problems (an input pinned by its box and its rate limit at once: no strict
    complementarity): an interior-point iterate is only O(sqrt(mu)) from the optimum there, and the
    Riccati recursion cannot take mu below ~1e-11 in fp64 (weights lam/t ~ 1e12 cancel in P), so a
    few problems per thousand sit 1e-4 .. 1e-3 away in X, U (measured on 1024 fresh problems:
    max 5e-4, 99th percentile 5e-5, median 1e-10; scratch/acc_eval4.py).  The iteration stops on such a
    problem as soon as the affine step stalls (mu_aff / mu > 0.4 at mu <= 1e-8) instead of adding noise.  They are bounded by
    TOL_DEGENERATE and, rigorously, by feasibility (1e-9 / 1e-8) and the objective gap
    (1e-7 relative) against the dense optimum, which every problem must meet.
The dense oracle itself is accurate to ~1e-12.
HIP kernel vs its serial C twin (identical algorithm; FMA contraction and summation order
differ, and the ill-conditioned late iterations amplify that): twice the bound against the
optimum (each twin may be 1e-4 off on its own), iteration counts equal on >= 90 % of problems
and never more than 1 apart.
"""
TOL_XU = 1e-4
TOL_DU = 2e-3
TOL_MEDIAN = 1e-7
TOL_DEGENERATE = 5e-3
TOL_TWIN = 2e-4
TOL_LINEARIZE_REL = 1e-11
