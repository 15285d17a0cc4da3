import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
import lmpc_scenario as LS
N = 20
veh, cfg, tr, laps, inp, q = LS.make(64, 7, N=N, n_laps=3)
cfg = P.barc_lmpc(N, 5)
rx, rj, rn = cbind.ss_query_batch(laps[:2], LS.L_BARC_SS, 160, 32, q)
print("n_found", np.unique(rn))
tw = cbind.solve_batch(cfg, veh, inp, ss_x=rx, ss_j=rj)
print("twin status", np.bincount(tw["status"], minlength=3), "iters", tw["iters"].mean())
errs = []
for b in range(16):
    qp = Q.build_qp(cfg, veh, S.problem(inp, b), ss_x=rx[:, :, b], ss_j=rj[:, b])
    y, info = Q.solve_dense(qp); o = qp.split(y)
    errs.append(max(np.abs((tw["X_optm"][..., b] - o["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((tw["dU_optm"][..., b] - o["dU_optm"]) / P.SCALE_U[:, None]).max()))
print("twin vs dense (16 problems, padded set):", max(errs), "lambda sums", np.abs(tw["convex_combi_optm"].sum(0) - 1).max())
