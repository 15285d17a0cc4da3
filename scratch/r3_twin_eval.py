"""The shipped twin (oracle/_build/liblmpc_oracle.so, polish on/off) against the dense optimum; reuses the /tmp dense caches of
r3_pol_eval.py.  usage: r3_twin_eval.py N B seed kind"""
import sys, os, pickle, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import r3_f32_eval as E
from oracle import cbind, params as P
N, B, seed, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg, veh, inp, ss_x, ss_j = E.build(kind, N, B, seed)
res = pickle.load(open(f"/tmp/dense_{kind}_{N}_{seed}_{B}.pkl", "rb"))
dst = np.array([r[1] for r in res])
for polish in (-1, 0):
    o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j, polish=polish)
    e = np.zeros(B); ed = np.zeros(B)
    for b, st, pol, X, U, dU, sc in res:
        e[b] = max((np.abs(o["X_optm"][:, :, b] - X) / P.SCALE_X[:, None]).max(), (np.abs(o["U_optm"][:, :, b] - U) / P.SCALE_U[:, None]).max())
        ed[b] = (np.abs(o["dU_optm"][:, :, b] - dU) / P.SCALE_U[:, None]).max()
    ok = dst == 0
    print(f"{kind} N={N} polish={'on' if polish == 0 else 'off'}: status {np.bincount(o['status'], minlength=3)} iters {o['iters'].mean():.2f} max {o['iters'].max()}  err XU 50/90/99/max {np.quantile(e[ok], [.5, .9, .99, 1.])}  dU max {ed[ok].max():.2e}  kkt: stat {o['kkt'][0][ok].max():.1e} viol {o['kkt'][1][ok].max():.1e} mu {o['kkt'][2][ok].max():.1e}")
