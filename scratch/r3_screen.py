"""Can the fp32 interior point's own quantities say when its answer is already within tolerance?  IPM-only error against the
polished fp64 reference vs (scaled last step, row ambiguity min max(lam/t, t/lam))."""
import sys, os, ctypes, pickle, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import r3_f32_eval as E
from oracle import cbind, params as P
kind, N, B, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg, veh, inp, ss_x, ss_j = E.build(kind, N, B, seed)
ref = pickle.load(open(f"/tmp/ref64_{kind}_{N}_{B}_{seed}.pkl", "rb"))
os.environ["DIAG_AMB"] = "1"
cbind._LIB = ctypes.CDLL("/tmp/liboracle_f32.so")
o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j)
ok = (ref["status"] == 0) & (o["status"] == 0)
ex = np.abs((o["X_optm"] - ref["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1)); eu = np.abs((o["U_optm"] - ref["U_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
e = np.maximum(ex, eu)[ok]; ls = o["kkt"][0][ok]; amb = o["kkt"][3][ok]
print(f"{kind}: {ok.sum()} problems, frac err > 1e-3: {np.mean(e > 1e-3):.4f}, > 3e-4: {np.mean(e > 3e-4):.4f}")
for ls_thr in (1e-4, 3e-4, 1e-3, 3e-3):
    for a_thr in (1, 10, 100, 1000):
        sel = (ls <= ls_thr) & (amb >= a_thr)
        if sel.any(): print(f"  last_step <= {ls_thr:g} & amb >= {a_thr:g}: pass {sel.mean():.3f}  max err among passed {e[sel].max():.2e}  (99.9% {np.quantile(e[sel], .999):.1e})")
