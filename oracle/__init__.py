"""CPU oracle for the batched LMPC solve path -- TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference's per-step
MPC/LMPC optimisation (MPC-Berkeley/Racing-LMPC-ROS2, src/mpc/racing_mpc,
src/vehicle_dynamics_models/{single_track_planar_model,racing_trajectory}).
It is the checker for the HIP product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import or execute anything under ``oracle/``; the product package never does.

PARITY UNPINNED: the reference's arithmetic lives in un-vendored third-party
code (CasADi >= 3.6.3 symbolic AD + Opti, OSQP through CasADi's conic plugin,
CGAL's kd-tree; none installed or installable here), and the reference's own
tests for this path assert nothing numeric (every one ends in SUCCEED();
SURVEY.md section 4 / 8c).  There are therefore no golden vectors to pin the
oracle against.  What pins it instead:
  * the QP is assembled variable-for-variable as racing_mpc.cpp:106-201,442-543
    builds it and solved densely to 1e-10 + an active-set polish; every
    solution carries a solver-independent KKT certificate (oracle/qp.py);
  * the analytic RK4 Jacobians are checked against complex-step
    differentiation of the RK4 map (oracle/dynamics.py);
  * the k-NN query is a brute-force sort (ss_query in oracle/c/lmpc_oracle.c, driven by tests/lmpc_scenario.py and
    tests/test_safe_set_oracle.py);
  * the reference's recorded BARC laps (its only data fixtures on this path,
    src/mpc/racing_mpc/test_data/barc_ss) are replayed through the dynamics as
    a plausibility check (tests/test_oracle_dynamics.py);
  * the parameter sets every test and bench line runs on (presets / oracle.params) are compared with the reference's
    shipped *.param.yaml files whenever its checkout is present (tests/test_ros_params.py).
"""
