"""twin statistics per workload under the EXP_* settings: mean iterations, exits without an accepted polish, and every such
problem against the dense optimum.  usage: sweep_twin.py what [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(__file__))
from common import *
from concurrent.futures import ProcessPoolExecutor
G = {}
def dense(b):
    kw = {} if G.get("ss_x") is None else {"ss_x": G["ss_x"][:, :, b], "ss_j": G["ss_j"][:, b]}
    qp = Q.build_qp(G["cfg"], G["veh"], S.problem(G["inp"], b), **kw)
    y, info = Q.solve_dense(qp); o = qp.split(y)
    return b, info["status"], info.get("polished"), o["X_optm"], o["U_optm"], o["dU_optm"]
for what in sys.argv[1].split(","):
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    cfg, veh, inp, ss_x, ss_j = batch(what, B)
    G.update(cfg=cfg, veh=veh, inp=inp, ss_x=ss_x, ss_j=ss_j)
    tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j)
    mu = tw["kkt"][2]; ok = tw["status"] == 0
    ref = np.nonzero((mu > 1e-16) & ok)[0]
    worst = (0, 0, 0)
    if len(sys.argv) > 3 and ref.size:
        with ProcessPoolExecutor(8) as ex:
            res = list(ex.map(dense, [int(b) for b in ref[:int(sys.argv[3])]], chunksize=1))
        errs = []
        for b, st, pol, X, U, dU in res:
            e = max(np.abs((tw["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max(), np.abs((tw["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max(), np.abs((tw["dU_optm"][:, :, b] - dU) / P.SCALE_U[:, None]).max())
            errs.append((e, b, st, pol))
        errs.sort(reverse=True); worst = errs[:3]
    print(f"{what}: status {np.bincount(tw['status'], minlength=3).tolist()} mean iters {tw['iters'][ok].mean():.3f} max {tw['iters'][ok].max()} exits with mu > 1e-16: {ref.size}; worst vs dense {worst}", flush=True)
