"""Accuracy of the C twin vs the dense oracle for several IPM tolerances, 192 fresh problems."""
import sys, numpy as np, os
from pathlib import Path
from concurrent.futures import ProcessPoolExecutor
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
import ctypes as C
if len(sys.argv) > 1:
    _real = C.CDLL(sys.argv[1]); cbind.lib = lambda: _real
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
B = 1024
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, 7)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
def dense(b):
    qp = Q.build_qp(cfg, veh, S.problem(inp, b))
    yex, info = Q.solve_dense(qp)
    o = qp.split(yex)
    return o["X_optm"], o["U_optm"], o["dU_optm"], info["status"]
cache = Path("/tmp/acc_dense_1024.npz")
if cache.exists():
    d = np.load(cache); DX, DU, DD = d["X"], d["U"], d["D"]
else:
    with ProcessPoolExecutor(16) as ex:
        res = list(ex.map(dense, range(B)))
    DX = np.stack([r[0] for r in res], -1); DU = np.stack([r[1] for r in res], -1); DD = np.stack([r[2] for r in res], -1)
    np.savez(cache, X=DX, U=DU, D=DD)
for tol in (1e-11,):
    out = cbind.solve_batch(cfg, veh, inp, tol=tol)
    ex = np.abs((out["X_optm"] - DX) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
    eu = np.abs((out["U_optm"] - DU) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    ed = np.abs((out["dU_optm"] - DD) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    per = np.maximum(ex, eu)
    print("tol %.0e iters %.2f status %s  XU: max %.2e p99 %.2e median %.2e | dU: max %.2e median %.2e" % (tol, out["iters"].mean(), np.bincount(out["status"], minlength=3), per.max(), np.percentile(per, 99), np.median(per), ed.max(), np.median(ed)))
