"""Round 4: the row state (t, lambda of every inequality row, per lane) the interior point ends with, for builds with -DLMPC_DUMP_ROWS.
    LMPC_HIP_LIBRARY=<lib> python scratch/r4_rowdump.py save <tag>;  python scratch/r4_rowdump.py diff <good> <bad>"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
KQ, N, B = 7, 40, 512


def save(tag):
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.capi._ABI_SYMBOLS = tuple(s for s in pkg.capi._ABI_SYMBOLS if s != "lmpc_query_launch_for")
    dev = torch.device("cuda:0")
    tr = pkg.workloads.synthetic_track("putnam")
    cfg, veh = dict(pkg.presets.iac_tracking_mpc(N)), pkg.presets.iac_vehicle()
    cfg["polish"] = -1
    x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
    sv = pkg.Solver(cfg, veh, device=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    out = sv.alloc_outputs(B)
    out["kkt"] = torch.zeros(4 * B + B * 64 * 6 * KQ, dtype=torch.float64, device=dev)
    o = sv.solve(inp, out)
    torch.cuda.synchronize()
    k = o["kkt"].cpu().numpy()
    np.savez("/tmp/r4rows_%s.npz" % tag, kkt=k[:4 * B].reshape(4, B), rows=k[4 * B:].reshape(B, 64, KQ, 6), status=o["status"].cpu().numpy(),
             iters=o["iters"].cpu().numpy())


def diff(g, b):
    G, Bd = np.load("/tmp/r4rows_%s.npz" % g), np.load("/tmp/r4rows_%s.npz" % b)
    print(json.dumps({"status_good": np.bincount(G["status"], minlength=4).tolist(), "status_bad": np.bincount(Bd["status"], minlength=4).tolist(),
                      "iters_good_mean": float(G["iters"].mean()), "iters_bad_mean": float(Bd["iters"].mean())}))
    rg, rb = G["rows"], Bd["rows"]
    names = ["t_up", "t_lo", "lam_up", "lam_lo", "p_up", "p_lo"]
    # where is the bad build's state not a legal interior-point state?  (t > 0, lam >= 0 always hold in the good build)
    for k, nm in enumerate(names[:4]):
        neg = rb[..., k] < 0
        nan = ~np.isfinite(rb[..., k])
        print(json.dumps({"array": nm, "negative_entries_bad": int(neg.sum()), "nonfinite_bad": int(nan.sum()), "negative_entries_good": int((rg[..., k] < 0).sum()),
                          "by_q_bad": neg.sum(axis=(0, 1)).tolist(), "by_lane_bad_top": np.argsort(-neg.sum(axis=(0, 2)))[:8].tolist(),
                          "by_lane_counts": np.sort(neg.sum(axis=(0, 2)))[::-1][:8].tolist()}))
    for pb in range(3):
        d = np.abs(rg[pb] - rb[pb]) / (1e-300 + np.abs(rg[pb]) + np.abs(rb[pb]))
        idx = np.argwhere(d > 0.5)
        print(json.dumps({"problem": pb, "iters": [int(G["iters"][pb]), int(Bd["iters"][pb])], "entries_grossly_different": int(idx.shape[0]),
                          "first": [[int(a), int(q), names[k], float(rg[pb, a, q, k]), float(rb[pb, a, q, k])] for a, q, k in idx[:12]]}))


if __name__ == "__main__":
    save(sys.argv[2]) if sys.argv[1] == "save" else diff(sys.argv[2], sys.argv[3])
