import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
N, B = 20, 64
tracker = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
learner = pkg.Solver(pkg.presets.barc_lmpc(N, 3), pkg.presets.barc_vehicle(), 0)
tr = pkg.workloads.synthetic_track("barc")
rng = np.random.default_rng(0)
x0 = np.stack([np.full(B, 0.5), rng.uniform(-0.05, 0.05, B), np.zeros(B), np.full(B, 2.0), np.zeros(B), np.zeros(B)])
x0[:, 0] = [0.5, 0.0, 0.0, 2.0, 0.0, 0.0]
res = pkg.closed_loop.run_lmpc(tracker, learner, tr, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"),
                              warm_laps=2, learn_laps=int(sys.argv[1]) if len(sys.argv) > 1 else 4, warm_speed_scale=float(sys.argv[2]) if len(sys.argv) > 2 else 0.7, debug=True)
print("lap times", [round(t, 3) for t in res["lap_times"]], res["lap_kind"])
print("steps", res["steps"], "laps in set", res["laps_in_set"], "fails per car max", int(res["n_fail"].max()), "car0", int(res["n_fail"][0]), "worst excess %.3f" % float(res["worst_excess"].max()))
