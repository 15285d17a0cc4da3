"""All dumped problems of a sweep: twin (current build, current env) against dense; status and error."""
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
from concurrent.futures import ProcessPoolExecutor
d = sys.argv[1]; only = sys.argv[2] if len(sys.argv) > 2 else ""
K = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
def work(f):
    z = np.load(os.path.join(d, f)); fam, N = f[:-4].split("_N"); N = int(N)
    if fam == "trk": cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
    elif fam == "iac": cfg, veh = P.iac_tracking_mpc(N), P.iac_vehicle()
    else: cfg, veh = P.barc_lmpc(N, 3 if fam == "lrn96" else 5), P.barc_vehicle()
    inp = {k: z["in_" + k] for k in K}; inp["L"] = float(z["L"])
    ss = ("ss_x" in z.files)
    tw = cbind.solve_batch(cfg, veh, inp, z["ss_x"] if ss else None, z["ss_j"] if ss else None)
    out = []
    for j in range(len(z["idx"])):
        kw = {} if not ss else {"ss_x": z["ss_x"][..., j], "ss_j": z["ss_j"][..., j]}
        qp = Q.build_qp(cfg, veh, S.problem(inp, j), **kw)
        y, info = Q.solve_dense(qp); o = qp.split(y)
        e = max(np.abs((tw["X_optm"][..., j] - o["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((tw["U_optm"][..., j] - o["U_optm"]) / P.SCALE_U[:, None]).max(), np.abs((tw["dU_optm"][..., j] - o["dU_optm"]) / P.SCALE_U[:, None]).max())
        out.append((e, f"{fam} N={N} problem {int(z['idx'][j])}: twin status {tw['status'][j]} iters {tw['iters'][j]} mu {tw['kkt'][2, j]:.1e} vs dense {e:.1e} (dense st {info['status']} pol {info.get('polished')}) vx0 {inp['x_ic'][3, j]:.2f}"))
    return out
if __name__ == "__main__":
    files = sorted(f for f in os.listdir(d) if f.endswith(".npz") and f.startswith(only))
    with ProcessPoolExecutor(8) as ex:
        allr = [r for rs in ex.map(work, files) for r in rs]
    bad = [r for r in allr if r[0] > 1e-6]
    print(len(allr), "problems; above 1e-6:", len(bad))
    for e, line in sorted(bad, reverse=True): print(line)
