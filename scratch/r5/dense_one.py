import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
what, b = sys.argv[1], int(sys.argv[2])
cfg, veh, inp, ss_x, ss_j = batch(what)
kw = {} if ss_x is None else {"ss_x": ss_x[:, :, b], "ss_j": ss_j[:, b]}
qp = Q.build_qp(cfg, veh, S.problem(inp, b), **kw)
os.environ["DENSE_TRACE"] = "1"
y, info = Q.solve_dense(qp); o = qp.split(y)
print({k: v for k, v in info.items() if k not in ("lam", "pi")})
tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, b0=b, b1=b + 1)
print("twin status", tw["status"][b], "iters", tw["iters"][b], "kkt", tw["kkt"][:, b], "x0", inp["x_ic"][:, b], inp["u_ic"][:, b])
yt = Q.pack(qp, tw["X_optm"][:, :, b], tw["U_optm"][:, :, b], tw["dU_optm"][:, :, b], sigma=tw["kkt"][3, b], lam=None if ss_x is None else tw["convex_combi_optm"][:, b], eps=None if ss_x is None else tw["X_optm"][:, -1, b] - ss_x[:, :, b] @ tw["convex_combi_optm"][:, b])
print("twin point certificate", Q.kkt_certificate(qp, yt), "objective twin", qp.objective(yt), "dense", qp.objective(y))
print("err X", np.abs((tw["X_optm"][:, :, b] - o["X_optm"]) / P.SCALE_X[:, None]).max(), "dU", np.abs((tw["dU_optm"][:, :, b] - o["dU_optm"]) / P.SCALE_U[:, None]).max())
