"""twin (under the EXP_* settings) against the cached dense optima: per workload mean iterations, worst error, count > 1e-6 / 1e-7."""
import os, sys
sys.path.insert(0, os.path.dirname(__file__))
from common import *
for what in sys.argv[1].split(","):
    c = np.load(ROOT / f"scratch/r5/cache/{what}.npz")
    B = c["status"].size
    cfg, veh, inp, ss_x, ss_j = batch(what, 4096)
    tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, b0=0, b1=B)
    ok = (tw["status"][:B] == 0) & (c["status"] == 0)
    e = np.maximum(np.abs((tw["X_optm"][:, :, :B] - c["X_optm"]) / P.SCALE_X[:, None, None]).max((0, 1)), np.abs((tw["U_optm"][:, :, :B] - c["U_optm"]) / P.SCALE_U[:, None, None]).max((0, 1)))
    ed = np.abs((tw["dU_optm"][:, :, :B] - c["dU_optm"]) / P.SCALE_U[:, None, None]).max((0, 1))
    em = np.maximum(e, ed)[ok]
    idx = np.nonzero(ok)[0][np.argsort(em)[-3:]]
    print(f"{what}: {B} problems, both solved {ok.sum()}, twin status {np.bincount(tw['status'][:B], minlength=3).tolist()} dense failed {(c['status'] != 0).sum()}; mean iters {tw['iters'][:B][ok].mean():.3f} max {tw['iters'][:B][ok].max()}; "
          f"err max {em.max():.1e} > 1e-6: {(em > 1e-6).sum()} > 1e-7: {(em > 1e-7).sum()} > 1e-8: {(em > 1e-8).sum()}; worst {[(int(i), float('%.1e' % np.maximum(e, ed)[i])) for i in idx]}", flush=True)
