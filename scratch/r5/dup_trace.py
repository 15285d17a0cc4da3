import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
import lmpc_scenario as LS
N, n_laps = 20, 6
veh, cfg, tr, laps, inp, q = LS.make(32, 70 + N, N=N, n_laps=3)
cfg = P.barc_lmpc(N, n_laps)
stored = (laps * 2)[:n_laps]
rx, rj, rn = cbind.ss_query_batch(stored, LS.L_BARC_SS, 32 * n_laps, 32, q)
tw = cbind.solve_batch(cfg, veh, inp, ss_x=rx, ss_j=rj)
print(tw["status"], tw["iters"])
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["LMPC_ORACLE_POLISH_TRACE"] = "1"
t1 = cbind.solve_batch(cfg, veh, inp, ss_x=rx, ss_j=rj, b0=b, b1=b + 1)
qp = Q.build_qp(cfg, veh, S.problem(inp, b), ss_x=rx[:, :, b], ss_j=rj[:, b])
y, info = Q.solve_dense(qp); o = qp.split(y)
print("dense", info["status"], info.get("polished"), "twin vs dense", np.abs((t1["X_optm"][..., b] - o["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((t1["dU_optm"][..., b] - o["dU_optm"]) / P.SCALE_U[:, None]).max())
print("support dense", np.nonzero(o["convex_combi_optm"] > 1e-9)[0], "twin", np.nonzero(t1["convex_combi_optm"][:, b] > 1e-9)[0])
