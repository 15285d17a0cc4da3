import sys, os, ctypes, pickle, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "racing-lmpc-ros2_amd"))
import workloads as wl
from oracle import cbind, params as P, qp as Q, scenario as S
N = int(sys.argv[1]); b = int(sys.argv[2]); B = 256; seed = 0; kind = "barc"
veh = P.barc_vehicle(); cfg = P.barc_tracking_mpc(N); tr = wl.synthetic_track(kind)
u_lo, u_hi = Q.effective_bounds(cfg, veh)[:2]
x, u = wl.sample_initial_states(kind, B, tr["L"], u_lo, u_hi, seed=seed)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
res = pickle.load(open(f"/tmp/dense_{kind}_{N}_{seed}_{B}.pkl", "rb"))
cbind._LIB = ctypes.CDLL("/tmp/liboracle_pol.so")
o = cbind.solve_batch(cfg, veh, inp, b0=b, b1=b + 1, tol=3e-14)
_, st, pol, X, U, dU, sc = res[b]
ex = np.abs(o["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]; eu = np.abs(o["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]
print("status", o["status"][b], "iters", o["iters"][b], "err X", ex.max(), "at", np.unravel_index(ex.argmax(), ex.shape), "err U", eu.max(), np.unravel_index(eu.argmax(), eu.shape), "dense polished", pol, "sc", sc, "kkt", o["kkt"][:, b])
qp = Q.build_qp(cfg, veh, S.problem(inp, b))
def obj(o_):
    y = np.zeros(qp.n) if not hasattr(qp, "pack") else None
    return None
print("U twin", o["U_optm"][:, :, b][0]); print("U dense", U[0])
