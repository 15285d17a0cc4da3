import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
what, b = sys.argv[1], int(sys.argv[2])
cfg, veh, inp, ss_x, ss_j = batch(what)
os.environ["LMPC_ORACLE_POLISH_TRACE"] = "1"
tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, b0=b, b1=b + 1)
print("status", tw["status"][b], "iters", tw["iters"][b], "kkt", tw["kkt"][:, b], "x0", inp["x_ic"][:, b])
if len(sys.argv) > 3:
    kw = {} if ss_x is None else {"ss_x": ss_x[:, :, b], "ss_j": ss_j[:, b]}
    qp = Q.build_qp(cfg, veh, S.problem(inp, b), **kw)
    y, info = Q.solve_dense(qp); o = qp.split(y)
    print("dense", info["status"], info["iters"], info["mu"], info.get("polished"), "margin", Q.strict_complementarity(qp, y, info["lam"]))
    print("err X", np.abs((tw["X_optm"][:, :, b] - o["X_optm"]) / P.SCALE_X[:, None]).max(), "dU", np.abs((tw["dU_optm"][:, :, b] - o["dU_optm"]) / P.SCALE_U[:, None]).max())
    if ss_x is not None:
        print("lambda support dense", np.nonzero(o["convex_combi_optm"] > 1e-9)[0], o["convex_combi_optm"][o["convex_combi_optm"] > 1e-9])
        print("lambda support twin ", np.nonzero(tw["convex_combi_optm"][:, b] > 1e-9)[0], tw["convex_combi_optm"][:, b][tw["convex_combi_optm"][:, b] > 1e-9])
