"""The RacingLMPC facade's first solve (tests/cpp/test_racing_lmpc.cpp) on the CPU with the dense SQP: how R_d affects the SQP's convergence."""
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
from oracle import nlp
RT = pkg.racing_trajectory.RacingTrajectory(str(ROOT / "tests/golden/barc_track/15_barc_optm.txt"))
N, dt, v0, s0 = 20, 0.025, 2.0, 2.0
for rd in [float(a) for a in sys.argv[1:]] or [1e-6, 1e-4, 1e-3, 1e-2]:
    cfg = P.barc_tracking_mpc(N).with_(q_vy=0.0, q_vyaw=0.0, q_boundary=50.0, R_d=np.diag([rd, rd]))
    veh = P.barc_vehicle()
    X = np.zeros((6, N)); X[0] = s0 + dt * v0 * np.arange(N); X[3] = v0
    x_ic = np.array([s0, 0.05, 0, v0, 0, 0.0]); X[:, 0] = x_ic
    s = X[0]
    pr = {"x_ic": x_ic, "u_ic": np.zeros(2), "X_ref": X, "U_ref": np.zeros((2, N - 1)), "T_ref": np.full(N - 1, dt),
          "bound_left": np.array([float(RT.left_boundary(si)) for si in s]), "bound_right": np.array([float(RT.right_boundary(si)) for si in s]),
          "curvatures": np.array([float(RT.curvature(si)) for si in s]), "vel_ref": np.minimum(np.array([float(RT.velocity(si)) for si in s]), 2.5), "L": float(RT.total_length)}
    Xs, Us, dUs, sig, info = nlp.solve_nlp_dense(cfg, veh, pr, max_sqp=60, tol=1e-8)
    print(f"R_d {rd:g}: status {info['status']} sqp_iters {info['sqp_iters']} move {info['move']:.2e} defect {np.abs(nlp.defect(veh, pr, Xs, Us)).max():.2e}  vx_end {Xs[3, -1]:.3f} max|dU| {np.abs(dUs).max(axis=1)}")
