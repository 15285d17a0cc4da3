"""The dumped worst problems of the dispatch sweep's failing cases: who is right, kernel or twin?  Dense optimum of each."""
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
from concurrent.futures import ProcessPoolExecutor
d = sys.argv[1]
def work(f):
    z = np.load(os.path.join(d, f)); fam, N = f[:-4].split("_N"); N = int(N)
    if fam == "trk": cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
    elif fam == "iac": cfg, veh = P.iac_tracking_mpc(N), P.iac_vehicle()
    else: cfg, veh = P.barc_lmpc(N, 3 if fam == "lrn96" else 5), P.barc_vehicle()
    out = []
    for j in range(len(z["idx"]) - 1, max(len(z["idx"]) - 3, -1), -1):
        inp = {k: z["in_" + k][..., j] for k in DCK}; inp["L"] = float(z["L"])
        kw = {} if "ss_x" not in z.files else {"ss_x": z["ss_x"][..., j], "ss_j": z["ss_j"][..., j]}
        qp = Q.build_qp(cfg, veh, inp, **kw)
        y, info = Q.solve_dense(qp); o = qp.split(y)
        def e(pre):
            return max(np.abs((z[pre + "X_optm"][..., j] - o["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((z[pre + "U_optm"][..., j] - o["U_optm"]) / P.SCALE_U[:, None]).max(),
                       np.abs((z[pre + "dU_optm"][..., j] - o["dU_optm"]) / P.SCALE_U[:, None]).max())
        out.append(f"{fam} N={N} problem {int(z['idx'][j])}: kernel-twin {z['err'][j]:.1e} | kernel vs dense {e('k_'):.1e} (iters {int(z['kernel_iters'][j])}, mu {z['kernel_kkt'][2, j]:.1e}) | twin vs dense {e('t_'):.1e} (iters {int(z['twin_iters'][j])}, mu {z['twin_kkt'][2, j]:.1e}) | dense st {info['status']} pol {info.get('polished')} margin {Q.strict_complementarity(qp, y, info['lam']):.1e} vx0 {inp['x_ic'][3]:.2f}")
    return out
DCK = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
if __name__ == "__main__":
    files = sorted(f for f in os.listdir(d) if f.endswith(".npz"))
    with ProcessPoolExecutor(8) as ex:
        for lines in ex.map(work, files):
            print("\n".join(lines), flush=True)
