"""Accuracy of the C twin vs the dense oracle on fresh problems (the GPU test's measure), CPU only."""
import sys, numpy as np, ctypes as C, os
from pathlib import Path
from concurrent.futures import ProcessPoolExecutor
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
if len(sys.argv) > 1:
    _real = C.CDLL(sys.argv[1]); cbind.lib = lambda: _real
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
B = 48
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, 21)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
out = cbind.solve_batch(cfg, veh, inp)
def dense(b):
    qp = Q.build_qp(cfg, veh, S.problem(inp, b))
    yex, info = Q.solve_dense(qp)
    o = qp.split(yex)
    return o["X_optm"], o["U_optm"], info["status"]
with ProcessPoolExecutor(16) as ex:
    res = list(ex.map(dense, range(B)))
per = []
for b, (X, U, st) in enumerate(res):
    per.append(max(np.abs((out["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max(), np.abs((out["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max()))
per = np.array(per)
print({k: os.environ[k] for k in os.environ if k.startswith("X_")}, "iters mean %.2f" % out["iters"].mean(), "status", np.bincount(out["status"], minlength=3), "max %.2e median %.2e p90 %.2e" % (per.max(), np.median(per), np.percentile(per, 90)), "mu max %.1e" % out["kkt"][2].max())
