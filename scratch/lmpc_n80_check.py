"""LMPC at N = 80, 160 safe-set points: kernel and twin against the dense optimum."""
import sys, importlib
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
pkg = importlib.import_module("racing-lmpc-ros2_amd")
import lmpc_scenario as LS
from oracle import cbind, params as P, qp as Q, scenario as S
N, n_laps, B = int(sys.argv[1]) if len(sys.argv) > 1 else 80, 5, 12
veh, _, tr, laps, inp, q = LS.make(B, 70 + N, N=N, n_laps=3)
cfg = P.barc_lmpc(N, n_laps)
stored = (laps * 2)[:n_laps]
solver = pkg.Solver(pkg.presets.barc_lmpc(N, n_laps), pkg.presets.barc_vehicle(), device=0)
solver.set_safe_set(stored, LS.L_BARC_SS)
ss_x, ss_j, nf = solver.ss_query(q)
out = solver.alloc_outputs(B)
out["convex_combi_optm"] = torch.zeros((32 * n_laps, B), dtype=torch.float64, device="cuda")
o = {k: v.cpu().numpy() for k, v in solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j).items() if hasattr(v, "cpu")}
sx, sj = ss_x.cpu().numpy(), ss_j.cpu().numpy()
twin = cbind.solve_batch(cfg, veh, inp, ss_x=sx, ss_j=sj)
for b in range(B):
    qp = Q.build_qp(cfg, veh, S.problem(inp, b), ss_x=sx[:, :, b], ss_j=sj[:, b])
    yex, info = Q.solve_dense(qp)
    ex = qp.split(yex)
    ek = np.abs((o["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max()
    et = np.abs((twin["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max()
    ekt = np.abs((twin["X_optm"][:, :, b] - o["X_optm"][:, :, b]) / P.SCALE_X[:, None]).max()
    print(f"b={b} dense status {info['status']} kernel st {o['status'][b]} it {o['iters'][b]} mu {o['kkt'][2,b]:.1e} | twin st {twin['status'][b]} it {twin['iters'][b]} | err kernel {ek:.2e} twin {et:.2e} kernel-twin {ekt:.2e}", flush=True)
