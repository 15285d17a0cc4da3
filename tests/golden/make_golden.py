"""Generate the golden QP vectors (run from the repo root: `python tests/golden/make_golden.py`).

Inputs come from the product's synthetic workload generator + the oracle's restatement of the
node's cold start; expected outputs are the dense, polished optimum from oracle/qp.py, each
with its solver-independent KKT certificate stored alongside.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
from oracle import params as OP, qp as OQ, scenario as OS  # noqa: E402

pkg = load_package()
wl = pkg.workloads

CASES = {
    "qp_barc_tracking_n20": (OP.barc_vehicle(), OP.barc_tracking_mpc(20), "barc", 24, 7),
    "qp_barc_tracking_n10": (OP.barc_vehicle(), OP.barc_tracking_mpc(10), "barc", 8, 8),
    "qp_iac_tracking_n40": (OP.iac_vehicle(), OP.iac_tracking_mpc(40), "putnam", 8, 9),
}

# ---- LMPC (safe set from the reference's recorded laps) ----
sys.path.insert(0, str(ROOT / "tests"))
import lmpc_scenario as LS  # noqa: E402

veh, cfg, tr, laps, inp, q = LS.make(16, 5)
ss_x, ss_j, nf = LS.oracle_safe_set(cfg, laps, q)
N, B = cfg.N, 16
X = np.zeros((6, N, B)); U = np.zeros((2, N - 1, B)); dU = np.zeros((2, N - 1, B)); obj = np.zeros(B); ok = np.zeros(B, bool)
cert = np.zeros((4, B)); margin = np.zeros(B)
for b in range(B):
    qp = OQ.build_qp(cfg, veh, OS.problem(inp, b), ss_x=ss_x[:, :, b], ss_j=ss_j[:, b])
    y, info = OQ.solve_dense(qp)
    o = qp.split(y)
    X[:, :, b], U[:, :, b], dU[:, :, b] = o["X_optm"], o["U_optm"], o["dU_optm"]
    obj[b] = qp.objective(y)
    c = OQ.kkt_certificate(qp, y)
    cert[:, b] = [c["stat"], c["eq"], c["ineq"], c["comp"]]
    ok[b] = info["status"] == 0
    margin[b] = OQ.strict_complementarity(qp, y, info["lam"])
    print("qp_barc_lmpc_n20", b, info["status"], info.get("polished"), c, margin[b])
np.savez_compressed(Path(__file__).parent / "qp_barc_lmpc_n20.npz", X_optm=X, U_optm=U, dU_optm=dU, objective=obj,
                    kkt_cert=cert, certified=ok, margin=margin, ss_x=ss_x, ss_j=ss_j, query=q,
                    **{k: np.asarray(v) for k, v in inp.items()})

for name, (veh, cfg, kind, B, seed) in CASES.items():
    tr = wl.synthetic_track(kind)
    u_lo, u_hi, _, _ = OQ.effective_bounds(cfg, veh)
    x, u = wl.sample_initial_states(kind, B, tr["L"], u_lo, u_hi, seed)
    inp = OS.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    N = cfg.N
    X = np.zeros((6, N, B))
    U = np.zeros((2, N - 1, B))
    dU = np.zeros((2, N - 1, B))
    sig = np.zeros(B)
    obj = np.zeros(B)
    cert = np.zeros((4, B))
    ok = np.zeros(B, dtype=bool)
    margin = np.zeros(B)
    for b in range(B):
        qp = OQ.build_qp(cfg, veh, OS.problem(inp, b))
        y, info = OQ.solve_dense(qp)
        o = qp.split(y)
        X[:, :, b], U[:, :, b], dU[:, :, b], sig[b] = o["X_optm"], o["U_optm"], o["dU_optm"], o["sigma"]
        obj[b] = qp.objective(y)
        c = OQ.kkt_certificate(qp, y)
        cert[:, b] = [c["stat"], c["eq"], c["ineq"], c["comp"]]
        ok[b] = info["status"] == 0 and bool(info.get("polished"))
        margin[b] = OQ.strict_complementarity(qp, y, info["lam"])
        print(name, b, info["status"], info.get("polished"), c, margin[b])
    arrs = {k: np.asarray(v) for k, v in inp.items()}
    np.savez_compressed(Path(__file__).parent / f"{name}.npz", X_optm=X, U_optm=U, dU_optm=dU, sigma=sig,
                        objective=obj, kkt_cert=cert, certified=ok, margin=margin, **arrs)
