import sys, numpy as np, ctypes as C, os
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
_real = C.CDLL(str(ROOT / "scratch/_exp_oracle5.so")); cbind.lib = lambda: _real
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
x, u = pkg.workloads.sample_initial_states("barc", 1024, tr["L"], u_lo, u_hi, 7)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
d = np.load("/tmp/acc_dense_1024.npz")
out = cbind.solve_batch(cfg, veh, inp)
per = np.maximum(np.abs((out["X_optm"] - d["X"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1)), np.abs((out["U_optm"] - d["U"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1)))
worst = np.argsort(per)[-6:][::-1]
print("worst", worst, per[worst])
for b in worst[:4]:
    row = []
    for mi in range(5, 15):
        o = cbind.solve_batch(cfg, veh, inp, b0=int(b), b1=int(b) + 1, max_iter=mi)
        e = max(np.abs((o["X_optm"][:, :, b] - d["X"][:, :, b]) / P.SCALE_X[:, None]).max(), np.abs((o["U_optm"][:, :, b] - d["U"][:, :, b]) / P.SCALE_U[:, None]).max())
        row.append("%d:%.0e/%.0e" % (o["iters"][b], o["kkt"][2, b], e))
        if o["status"][b] == 0: break
    print(b, " ".join(row))
print("objective gaps:")
for b in worst:
    qp = Q.build_qp(cfg, veh, S.problem(inp, int(b)))
    y = Q.pack(qp, out["X_optm"][:, :, b], out["U_optm"][:, :, b], out["dU_optm"][:, :, b], sigma=out["kkt"][3, b])
    yd = Q.pack(qp, d["X"][:, :, b], d["U"][:, :, b], d["D"][:, :, b], sigma=None) if False else None
    yex, info = Q.solve_dense(qp)
    f, fex = qp.objective(y), qp.objective(yex)
    cert = Q.kkt_certificate(qp, y)
    print(b, "err %.1e" % per[b], "gap %.2e rel %.1e" % (f - fex, (f - fex) / (1 + abs(fex))), "eq %.0e ineq %.0e" % (np.abs(qp.A @ y - qp.b).max(), (qp.C @ y - qp.d).max()), "cert stat %.1e comp %.1e" % (cert["stat"], cert["comp"]), "n_active", cert["n_active"])
