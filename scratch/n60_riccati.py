"""Plain Riccati recursion on a failing N = 60 low-speed problem (numpy, fp64): growth of P, K and the closed loop."""
import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "racing-lmpc-ros2_amd"))
import workloads as wl
from oracle import params as P, qp as Q, scenario as S
N = 60
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(N)
tr = wl.synthetic_track("barc")
x, u = wl.sample_initial_states("barc", 256, tr["L"], [-0.015, -0.314159], [0.015, 0.314159], seed=0)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 17
pr = S.problem(inp, b)
A, B, g = Q.linearise(cfg, veh, pr)
print("A shape", np.shape(A), "B", np.shape(B))
A = np.asarray(A); B = np.asarray(B)
if A.shape[0] == 6: A = np.moveaxis(A, -1, 0); B = np.moveaxis(B, -1, 0)
for i in (0, 20, 40, 58):
    ev = np.linalg.eigvals(A[i])
    print(f"stage {i}: |eig A| {np.sort(np.abs(ev))[::-1].round(2)}  |B| max {np.abs(B[i]).max():.2f} vx_ref {pr['X_ref'][3, i]:.3f}")
# augmented z = [x; u_prev], v = dU: zbar_{i+1} = [A B; 0 I] z + [B t; I t] v
t = 0.025
Qd = np.diag([0, 2*cfg.q_contour, 2*cfg.q_heading, 2*cfg.q_vel, 2*cfg.q_vy, 2*cfg.q_vyaw] + list(np.diag(cfg.R + cfg.R.T)))
Sv = cfg.R_d + cfg.R_d.T
Pm = Qd.copy(); Pm[:6, :6] *= 10
for i in range(N - 2, -1, -1):
    Ab = np.block([[A[i], B[i]], [np.zeros((2, 6)), np.eye(2)]])
    Bb = np.vstack([B[i] * t, np.eye(2) * t])
    H = Sv + Bb.T @ Pm @ Bb
    K = np.linalg.solve(H, Bb.T @ Pm @ Ab)
    Pm = Qd + Ab.T @ Pm @ Ab - (Bb.T @ Pm @ Ab).T @ K
    Pm = 0.5 * (Pm + Pm.T)
    if i % 10 == 0 or i > N - 6:
        cl = np.abs(np.linalg.eigvals(Ab - Bb @ K))
        print(f"i={i}: |P| {np.abs(Pm).max():.2e} cond(H) {np.linalg.cond(H):.1e} |K| {np.abs(K).max():.2e} closed-loop |eig| max {cl.max():.3f}  min eig P {np.linalg.eigvalsh(Pm).min():.2e}")

# the unstable mode at stage 40: right / left eigenvectors and how the inputs reach it
import scipy.linalg as sl
w, vl, vr = sl.eig(A[40], left=True, right=True)
k = int(np.argmax(np.abs(w)))
print("eig", w[k], "right vec (s, ey, epsi, vx, vy, om)", np.real(vr[:, k]).round(3), "left vec", np.real(vl[:, k]).round(3))
print("left' B (reach of u_lon, steer):", (np.real(vl[:, k]) @ B[40]).round(3), " left' B over stages 30..45:", [float(np.abs(np.real(sl.eig(A[i], left=True, right=False)[1][:, int(np.argmax(np.abs(sl.eig(A[i])[0])))]) @ B[i]).max().round(2)) for i in range(30, 46, 3)])
