"""Round 6: the two-wave kernels against the one-wave kernels -- answers, statuses, iteration counts, kernel time."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import params as P
pkg = load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
CASES = (("barc", 24), ("barc", 40), ("barc", 60), ("barc", 80), ("iac", 40), ("iac", 80))
if len(sys.argv) > 2:  # w2_check.py B N,N,N...
    CASES = tuple(("barc", int(n)) for n in sys.argv[2].split(","))
for kind, N in CASES:
    tr = pkg.workloads.synthetic_track("putnam" if kind == "iac" else "barc")
    if kind == "iac":
        sv = pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), device=0)
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
    else:
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    res = {}
    for w in (1, 2):
        sv.set_waves_per_problem(w)
        out = sv.alloc_outputs(B)
        sv.solve(inp, out)
        torch.cuda.synchronize()
        sv.enable_timing(True)
        ms = []
        for _ in range(12):
            sv.solve(inp, out)
            torch.cuda.synchronize()
            ms.append(sv.last_kernel_ms()[1])
        sv.enable_timing(False)
        res[w] = ({k: v.cpu().numpy() for k, v in out.items() if hasattr(v, "cpu")}, float(np.median(ms)))
    o1, o2 = res[1][0], res[2][0]
    ok = (o1["status"] == 0) & (o2["status"] == 0)
    sx = P.SCALE_X[:, None, None]
    e = np.abs((o2["X_optm"] - o1["X_optm"]) / sx).max(axis=(0, 1))
    print("%s N %d B %d: kernel ms one wave %.3f two waves %.3f (x%.2f) | status one %s two %s | two vs one X max %.1e (solved by both %d) | iters one %.2f two %.2f, equal on %.3f"
          % (kind, N, B, res[1][1], res[2][1], res[1][1] / res[2][1], np.bincount(o1["status"], minlength=3).tolist(), np.bincount(o2["status"], minlength=3).tolist(),
             e[ok].max() if ok.any() else -1, ok.sum(), o1["iters"].mean(), o2["iters"].mean(), (o1["iters"] == o2["iters"]).mean()), flush=True)
    sv.close()
