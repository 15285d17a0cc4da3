"""Round 4: which problems does a miscomputing build get wrong?  usage:
    LMPC_HIP_LIBRARY=<lib> python scratch/r4_cmp_builds.py save <tag>      (iac40 / barc40 fp64 answers of that build -> /tmp/r4cmp_<tag>.npz)
    python scratch/r4_cmp_builds.py diff <good tag> <bad tag>               (per-problem comparison, JSON)"""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
SX = np.array([2000, 10, 0.1, 80, 2, 2.0])
SU = np.array([10, 0.3])


def save(tag):
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.capi._ABI_SYMBOLS = tuple(s for s in pkg.capi._ABI_SYMBOLS if s != "lmpc_query_launch_for")
    dev = torch.device("cuda:0")
    out = {}
    for kind, N, B in (("iac", 40, 8192), ("barc", 40, 4096), ("barc", 80, 2048)):
        tr = pkg.workloads.synthetic_track("putnam" if kind == "iac" else "barc")
        if kind == "iac":
            cfg, veh = dict(pkg.presets.iac_tracking_mpc(N)), pkg.presets.iac_vehicle()
            x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
        else:
            cfg, veh = dict(pkg.presets.barc_tracking_mpc(N)), pkg.presets.barc_vehicle()
            x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
        for pol in (0, -1):
            cfg["polish"] = pol
            sv = pkg.Solver(cfg, veh, device=0)
            inp = sv.prepare(tr, x.T.copy(), 0.025)
            inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
            o = sv.solve(inp)
            torch.cuda.synchronize()
            for k in ("X_optm", "U_optm", "status", "iters", "kkt"):
                out["%s%d_p%d_%s" % (kind, N, pol, k)] = o[k].cpu().numpy()
            sv.close()
    np.savez("/tmp/r4cmp_%s.npz" % tag, **out)


def diff(g, b):
    G, Bd = np.load("/tmp/r4cmp_%s.npz" % g), np.load("/tmp/r4cmp_%s.npz" % b)
    for case in ("iac40_p0", "iac40_p-1", "barc40_p0", "barc40_p-1", "barc80_p0", "barc80_p-1"):
        Xg, Xb = G[case + "_X_optm"], Bd[case + "_X_optm"]
        e = np.maximum((np.abs(Xg - Xb) / SX[:, None, None]).max(axis=(0, 1)), (np.abs(G[case + "_U_optm"] - Bd[case + "_U_optm"]) / SU[:, None, None]).max(axis=(0, 1)))
        sg, sb, ig, ib = G[case + "_status"], Bd[case + "_status"], G[case + "_iters"], Bd[case + "_iters"]
        bitdiff = np.where(e > 0)[0]
        row = {"case": case, "good": g, "bad": b, "problems": int(e.size), "bitwise_different": int(bitdiff.size),
               "status_changed": int((sg != sb).sum()), "iters_changed": int((ig != ib).sum()),
               "err_hist": {"<=1e-12": int(((e > 0) & (e <= 1e-12)).sum()), "1e-12..1e-9": int(((e > 1e-12) & (e <= 1e-9)).sum()),
                            "1e-9..1e-6": int(((e > 1e-9) & (e <= 1e-6)).sum()), "1e-6..1e-3": int(((e > 1e-6) & (e <= 1e-3)).sum()), ">1e-3": int((e > 1e-3).sum())},
               "iters_of_good_where_different": np.bincount(ig[bitdiff], minlength=1).tolist(), "iters_of_good_all": np.bincount(ig).tolist(),
               "examples": [{"b": int(i), "status": [int(sg[i]), int(sb[i])], "iters": [int(ig[i]), int(ib[i])], "err": float(e[i]),
                             "kkt_good": G[case + "_kkt"][:, i].tolist(), "kkt_bad": Bd[case + "_kkt"][:, i].tolist()} for i in bitdiff[:6]]}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "save":
        save(sys.argv[2])
    else:
        diff(sys.argv[2], sys.argv[3])
