"""The closed loop of soak_warm.py, eager (every kernel of a period is its own launch), for rocprofv3 --kernel-trace:
loop_prof.py cars periods cold|warm [horizon]"""
import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
B, steps, warm = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "warm"
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], B)
x0 = np.stack([s0, rng.uniform(-0.08, 0.08, B), rng.normal(0, 0.03, B), rng.uniform(0.6, 0.95, B) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]), np.zeros(B), np.zeros(B)])
sv = pkg.Solver(dict(pkg.presets.barc_tracking_mpc(N)), pkg.presets.barc_vehicle(), 0)
r = pkg.closed_loop.run(sv, tab, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"), steps=steps, speed_scale=0.9, graph=False, warm=warm)
torch.cuda.synchronize()
print("cars %d periods %d N %d warm %s: accepted %s" % (B, steps, N, warm, r["warm_hit_rate"]))
