"""Is the worst-case disagreement with the dense oracle a flat direction of the QP?  objective gap + feasibility of the twin's point."""
import sys, numpy as np
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import cbind, params as P, qp as Q, scenario as S
pkg = load_package()
veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(20)
tr = pkg.workloads.synthetic_track("barc")
u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
B = 192
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, 5)
inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
d = np.load("/tmp/acc_dense_192.npz"); DX, DU, DD = d["X"], d["U"], d["D"]
out = cbind.solve_batch(cfg, veh, inp)
per = np.maximum(np.abs((out["X_optm"] - DX) / P.SCALE_X[:, None, None]).max(axis=(0, 1)), np.abs((out["U_optm"] - DU) / P.SCALE_U[:, None, None]).max(axis=(0, 1)))
for b in np.argsort(per)[-5:]:
    qp = Q.build_qp(cfg, veh, S.problem(inp, b))
    y = Q.pack(qp, out["X_optm"][:, :, b], out["U_optm"][:, :, b], out["dU_optm"][:, :, b], sigma=out["kkt"][3, b])
    yex, info = Q.solve_dense(qp)
    f, fex = qp.objective(y), qp.objective(yex)
    cert = Q.kkt_certificate(qp, y) if hasattr(Q, "kkt_certificate") else {}
    print("b %d err %.2e iters %d  obj %.12e dense %.12e gap %.2e rel %.1e  eq %.1e ineq %.1e  cert %s  |y-yex|inf %.2e" % (b, per[b], out["iters"][b], f, fex, f - fex, (f - fex) / (1 + abs(fex)), np.abs(qp.A @ y - qp.b).max(), (qp.C @ y - qp.d).max(), {k: ("%.1e" % v if isinstance(v, float) else v) for k, v in cert.items()}, np.abs(y - yex).max()))
