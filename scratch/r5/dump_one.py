"""One dumped problem: dense optimum's simplex support, the twin re-run with the polish trace.  usage: dump_one.py dir family N j"""
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
d, fam, N, j = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
z = np.load(os.path.join(d, f"{fam}_N{N}.npz"))
if fam == "trk": cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
elif fam == "iac": cfg, veh = P.iac_tracking_mpc(N), P.iac_vehicle()
else: cfg, veh = P.barc_lmpc(N, 3 if fam == "lrn96" else 5), P.barc_vehicle()
K = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
inp = {k: z["in_" + k][..., j] for k in K}; inp["L"] = float(z["L"])
kw = {} if "ss_x" not in z.files else {"ss_x": z["ss_x"][..., j], "ss_j": z["ss_j"][..., j]}
qp = Q.build_qp(cfg, veh, inp, **kw)
y, info = Q.solve_dense(qp); o = qp.split(y)
print("problem", int(z["idx"][j]), "dense", info["status"], info["iters"], info.get("polished"), "objective", qp.objective(y))
if kw:
    lam = o["convex_combi_optm"]; print("dense lambda support (> 1e-9):", np.nonzero(lam > 1e-9)[0], lam[lam > 1e-9], "eps", o["eps"])
    print("distinct safe-set points:", len({tuple(np.round(kw["ss_x"][:, i], 12)) for i in range(kw["ss_x"].shape[1])}), "of", kw["ss_x"].shape[1])
binp = {k: (np.asarray(v)[..., None] if k != "L" else v) for k, v in inp.items()}
os.environ["LMPC_ORACLE_POLISH_TRACE"] = "1"
tw = cbind.solve_batch(cfg, veh, binp, None if not kw else kw["ss_x"][..., None], None if not kw else kw["ss_j"][..., None])
e = max(np.abs((tw["X_optm"][..., 0] - o["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((tw["dU_optm"][..., 0] - o["dU_optm"]) / P.SCALE_U[:, None]).max())
print("twin status", tw["status"][0], "iters", tw["iters"][0], "kkt", tw["kkt"][:, 0], "vs dense", e)
if kw: print("twin lambda support:", np.nonzero(tw["convex_combi_optm"][:, 0] > 1e-9)[0], tw["convex_combi_optm"][:, 0][tw["convex_combi_optm"][:, 0] > 1e-9])
