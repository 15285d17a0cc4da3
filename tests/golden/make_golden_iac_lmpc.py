"""Golden vectors for the IAC learning controller as shipped (iac_car_lmpc.param.yaml: n = 60, IAC vehicle, hull slack
[200, 20, 2, 200, 2, 20], 96 safe-set points from 3 laps): run from the repo root,
`python tests/golden/make_golden_iac_lmpc.py`.  The reference ships no IAC laps (its load_path points at files outside
the repository), so the safe set comes from synthetic laps on the Putnam-scale track of workloads.synthetic_track; inputs
are the node's cold start from states near the newest lap, the safe set is the oracle's k-NN query, and the expected
output is the dense, polished optimum of the QP racing_mpc.cpp:106-201,479-522 defines (oracle/qp.py)."""
import os
import sys
from multiprocessing import Pool
from pathlib import Path

os.environ.setdefault("OMP_NUM_THREADS", "1")
import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402
from oracle import cbind, dynamics as D, params as OP, qp as OQ, scenario as OS  # noqa: E402

B, N, SEED = 8, 60, 17


def scenario(B=B, N=N, seed=SEED):
    pkg = load_package()
    veh, cfg = OP.iac_vehicle(), OP.iac_lmpc(N, 3)
    tr = pkg.workloads.synthetic_track("putnam")
    L = float(tr["L"])
    laps = []
    for l in range(3):   # ~2400 samples per lap at 40..44 m/s, a lateral weave that differs per lap
        n = 2400
        s = (np.arange(n) + 0.37) * L / n
        k = np.interp(s, np.arange(tr["M"]) * L / tr["M"], tr["curvature"], period=L)
        vx = np.full(n, 40.0 + 2.0 * l)
        ey = 0.8 * np.sin(2 * np.pi * 3 * s / L + 0.9 * l)
        epsi = 0.8 * (2 * np.pi * 3 / L) * np.cos(2 * np.pi * 3 * s / L + 0.9 * l)
        laps.append(np.stack([s, ey, epsi, vx, np.zeros(n), k * vx], axis=1))
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, laps[-1].shape[0], B)
    x = laps[-1][idx] + rng.normal(0, 1, (B, 6)) * np.array([0.0, 0.3, 0.01, 1.0, 0.1, 0.02])
    x[:, 0] = np.mod(x[:, 0], L)
    inp = OS.cold_start_inputs(cfg, veh, tr, x, np.zeros((B, 2)), 0.025)
    q = np.stack([D.align_abscissa(inp["X_ref"][0, -1, :], inp["x_ic"][0, :], L), inp["X_ref"][1, -1, :]])
    ss_x, ss_j, nf = cbind.ss_query_batch(laps, L, cfg.num_ss_pts, cfg.num_ss_pts_per_lap, q)
    return veh, cfg, tr, laps, inp, q, ss_x, ss_j


_G = {}


def work(b):
    veh, cfg, inp, ss_x, ss_j = _G["v"]
    qp = OQ.build_qp(cfg, veh, OS.problem(inp, b), ss_x=ss_x[:, :, b], ss_j=ss_j[:, b])
    y, info = OQ.solve_dense(qp)
    o = qp.split(y)
    c = OQ.kkt_certificate(qp, y)
    return (o["X_optm"], o["U_optm"], o["dU_optm"], o["convex_combi_optm"], qp.objective(y), [c["stat"], c["eq"], c["ineq"], c["comp"]],
            info["status"] == 0 and bool(info.get("polished")), OQ.strict_complementarity(qp, y, info["lam"]))


if __name__ == "__main__":
    veh, cfg, tr, laps, inp, q, ss_x, ss_j = scenario()
    _G["v"] = (veh, cfg, inp, ss_x, ss_j)
    with Pool(min(8, B)) as pool:
        res = pool.map(work, range(B))
    for b, r in enumerate(res):
        print(b, "certified", r[6], "cert", r[5], "margin", r[7])
    np.savez_compressed(Path(__file__).parent / "qp_iac_lmpc_n60.npz", X_optm=np.stack([r[0] for r in res], -1),
                        U_optm=np.stack([r[1] for r in res], -1), dU_optm=np.stack([r[2] for r in res], -1),
                        convex_combi_optm=np.stack([r[3] for r in res], -1), objective=np.array([r[4] for r in res]),
                        kkt_cert=np.array([r[5] for r in res]).T, certified=np.array([r[6] for r in res]),
                        margin=np.array([r[7] for r in res]), ss_x=ss_x, ss_j=ss_j, query=q,
                        laps=np.stack(laps), **{k: np.asarray(v) for k, v in inp.items()})
