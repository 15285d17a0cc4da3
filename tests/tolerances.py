"""Stated parity tolerances (scaled units: X / scale_x, U / scale_u, dU / scale_u with the
reference's scale vectors, racing_mpc.cpp:36-37).

The reference hands the QP to OSQP with default eps_abs = eps_rel = 1e-3 (+ polish), so 1e-3 in
scaled variables is all it guarantees (racing_mpc.cpp:86-103).  The structured interior-point
solver (C oracle and HIP kernel, same algorithm) is held to tighter figures against the dense,
polished, KKT-certified optimum:
  * X, U:  1e-4 worst case;   dU (= difference quotient of U over dt = 25 ms): 2e-3 worst case
  * median over a batch: 1e-7
The worst case is set by the plain Riccati recursion's conditioning late in the iteration
("numerics"); the dense oracle itself is accurate to ~1e-12.
HIP kernel vs its serial C twin (identical algorithm; FMA contraction and summation order
differ, and the ill-conditioned late iterations amplify that): twice the bound against the
optimum (each twin may be 1e-4 off on its own), iteration counts equal on >= 90 % of problems
and never more than 1 apart.
"""
TOL_XU = 1e-4
TOL_DU = 2e-3
TOL_MEDIAN = 1e-7
TOL_TWIN = 2e-4
TOL_LINEARIZE_REL = 1e-11
