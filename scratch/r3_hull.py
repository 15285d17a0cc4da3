"""Hard hull equality: dense oracle (eps pinned to 0) against the twin with a large slack weight."""
import sys, numpy as np, dataclasses
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import lmpc_scenario as LS
from oracle import cbind, qp as Q, params as P, scenario as S
B = 64
veh, cfg, tr, laps, inp, q = LS.make(B, 5)
ss_x, ss_j, _ = LS.oracle_safe_set(cfg, laps, q)
hard = dataclasses.replace(cfg, convex_hull_slack=np.zeros(6))
for W in (0.0,):
    big = hard
    tw = cbind.solve_batch(big, veh, inp, ss_x, ss_j)
    errs = []
    for b in range(B):
        qp = Q.build_qp(hard, veh, S.problem(inp, b), ss_x[:, :, b], ss_j[:, b])
        try:
            y, info = Q.solve_dense(qp)
        except np.linalg.LinAlgError:
            info = {"status": 9}
        if info["status"] != 0:
            errs.append((b, "dense status", info["status"], "twin", int(tw["status"][b]))); continue
        o = qp.split(y)
        e = max(np.abs((tw["X_optm"][:, :, b] - o["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((tw["U_optm"][:, :, b] - o["U_optm"]) / P.SCALE_U[:, None]).max())
        eps = tw["X_optm"][:, -1, b] - ss_x[:, :, b] @ tw["convex_combi_optm"][:, b]
        errs.append((b, f"{e:.1e}", int(tw["status"][b]), int(tw["iters"][b]), f"eps {np.abs(eps / P.SCALE_X).max():.1e}"))
    good=[float(e[1]) for e in errs if e[2]==0 and isinstance(e[2],int) and e[1][0].isdigit()]
    print("W", W, "n ok", len(good), "max", max(good), "median", np.median(good), "bad", [e for e in errs if not (e[2]==0 and e[1][0].isdigit())])
