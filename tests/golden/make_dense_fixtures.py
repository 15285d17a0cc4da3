"""Generate tests/golden/dense_<case>.npz: the dense, polished optimum (oracle/qp.py `solve_dense`, the reference's scaled
variables) of every problem of tests/dense_cases.py, with its margin of strict complementarity and, for the problems the
active-set polish of the dense solver did not accept, the solver-independent KKT certificate of the point stored.

Run from the repo root:  python tests/golden/make_dense_fixtures.py [case ...]      (CPU; ~15 minutes on 8 cores for all)
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402
from oracle import qp as Q, scenario as S  # noqa: E402
import dense_cases as DC  # noqa: E402

G = {}


def dense(b):
    kw = {} if G["ss_x"] is None else {"ss_x": G["ss_x"][:, :, b], "ss_j": G["ss_j"][:, b]}
    qp = Q.build_qp(G["cfg"], G["veh"], S.problem(G["inp"], b), **kw)
    try:
        y, info = Q.solve_dense(qp)
    except np.linalg.LinAlgError:
        return None
    o = qp.split(y)
    c = Q.kkt_certificate(qp, y)
    gs = max(1.0, float(np.abs(qp.H @ y + qp.h).max()))
    return (info["status"], bool(info.get("polished")), o["X_optm"], o["U_optm"], o["dU_optm"], Q.strict_complementarity(qp, y, info["lam"]),
            qp.objective(y), (c["stat"] / gs, c["eq"], c["ineq"], c["comp"]), info["iters"])


if __name__ == "__main__":
    pkg = load_package()
    names = sys.argv[1:] or list(DC.CASES)
    for name in names:
        cfg, veh, inp, ss_x, ss_j = DC.build(pkg, name)
        G.update(cfg=cfg, veh=veh, inp=inp, ss_x=ss_x, ss_j=ss_j)
        B, N = inp["x_ic"].shape[-1], cfg.N
        t0 = time.time()
        with ProcessPoolExecutor(os.cpu_count()) as ex:
            res = list(ex.map(dense, range(B), chunksize=2))
        bad = [b for b, r in enumerate(res) if r is None]
        assert not bad, (name, "singular KKT system", bad)
        st = np.array([r[0] for r in res], dtype=np.int32)
        np.savez(ROOT / "tests" / "golden" / f"dense_{name}.npz",
                 status=st, polished=np.array([r[1] for r in res]), X_optm=np.stack([r[2] for r in res], -1), U_optm=np.stack([r[3] for r in res], -1),
                 dU_optm=np.stack([r[4] for r in res], -1), margin=np.array([r[5] for r in res]), objective=np.array([r[6] for r in res]),
                 kkt_cert=np.array([r[7] for r in res]).T, iters=np.array([r[8] for r in res], dtype=np.int32),
                 x_ic=inp["x_ic"], u_ic=inp["u_ic"], digest=DC.digest(inp, ss_x, ss_j))
        cert = np.array([r[7] for r in res])
        print(f"dense_{name}: {B} problems (N = {N}), solved {(st == 0).sum()}, polished {sum(r[1] for r in res)}, certificate worst: stationarity "
              f"{cert[:, 0].max():.1e} (relative) rows {max(cert[:, 1].max(), cert[:, 2].max()):.1e} complementarity {cert[:, 3].max():.1e}; "
              f"mean iterations {np.mean([r[8] for r in res]):.1f}  ({time.time() - t0:.0f} s)", flush=True)
