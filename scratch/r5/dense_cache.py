"""Dense certified optima of the first B problems of a bench batch, cached under scratch/r5/cache/ (git-ignored)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(__file__))
from common import *
from concurrent.futures import ProcessPoolExecutor
G = {}
def dense(b):
    kw = {} if G.get("ss_x") is None else {"ss_x": G["ss_x"][:, :, b], "ss_j": G["ss_j"][:, b]}
    qp = Q.build_qp(G["cfg"], G["veh"], S.problem(G["inp"], b), **kw)
    try:
        y, info = Q.solve_dense(qp)
    except np.linalg.LinAlgError:
        return None
    o = qp.split(y)
    return info["status"], bool(info.get("polished")), o["X_optm"], o["U_optm"], o["dU_optm"], Q.strict_complementarity(qp, y, info["lam"])
if __name__ == "__main__":
    out = ROOT / "scratch/r5/cache"; out.mkdir(exist_ok=True)
    for spec in sys.argv[1:]:
        what, B = spec.split(":"); B = int(B)
        cfg, veh, inp, ss_x, ss_j = batch(what, 4096)
        G.update(cfg=cfg, veh=veh, inp=inp, ss_x=ss_x, ss_j=ss_j)
        t0 = time.time()
        with ProcessPoolExecutor(8) as ex:
            res = list(ex.map(dense, range(B), chunksize=2))
        N = cfg.N
        st = np.array([r[0] if r else 9 for r in res]); pol = np.array([r[1] if r else False for r in res])
        X = np.stack([r[2] if r else np.zeros((6, N)) for r in res], -1); U = np.stack([r[3] if r else np.zeros((2, N - 1)) for r in res], -1)
        dU = np.stack([r[4] if r else np.zeros((2, N - 1)) for r in res], -1); mg = np.array([r[5] if r else 0 for r in res])
        np.savez(out / f"{what}.npz", status=st, polished=pol, X_optm=X, U_optm=U, dU_optm=dU, margin=mg)
        print(f"{what}: {B} problems, dense solved {(st == 0).sum()}, polished {pol.sum()}  ({time.time() - t0:.0f} s)", flush=True)
