"""Which solves fail in the LMPC closed loop of tests/test_gpu_path.py::test_lmpc_experiment_lap_times_improve?  Dumps the
inputs of the failing problems to gpurun_out/lmpc_fail.npz for a replay on the CPU twin."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from __graft_entry__ import load_package
pkg = load_package()
N, B = 20, 64
tracker = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
learner = pkg.Solver(pkg.presets.barc_lmpc(N, 3), pkg.presets.barc_vehicle(), device=0)
tr = pkg.workloads.synthetic_track("barc")
rng = np.random.default_rng(0)
x0 = np.stack([np.full(B, 0.5), rng.uniform(-0.05, 0.05, B), np.zeros(B), np.full(B, 2.0), np.zeros(B), np.zeros(B)])
x0[:, 0] = [0.5, 0.0, 0.0, 2.0, 0.0, 0.0]
dump = []
orig = learner.solve
def solve(inp, out=None, ss_x=None, ss_j=None, **kw):
    o = orig(inp, out, ss_x=ss_x, ss_j=ss_j, **kw)
    st = o["status"].cpu().numpy()
    for b in np.where(st != 0)[0]:
        if len(dump) < 40:
            rec = {k: (v[..., b].cpu().numpy() if hasattr(v, "cpu") and v.dim() >= 1 else v) for k, v in inp.items() if k != "L"}
            rec.update(ss_x=ss_x[..., b].cpu().numpy(), ss_j=ss_j[..., b].cpu().numpy(), status=st[b], iters=int(o["iters"][b]), kkt=o["kkt"][:, b].cpu().numpy(), L=tr["L"])
            dump.append(rec)
    return o
learner.solve = solve
res = pkg.closed_loop.run_lmpc(tracker, learner, tr, torch.as_tensor(x0, device="cuda"), torch.zeros((2, B), dtype=torch.float64, device="cuda"),
                               warm_laps=2, learn_laps=4, warm_speed_scale=0.7, debug=False)
print("lap times", res["lap_times"], "n_fail", res["n_fail"].cpu().numpy().sum(), "dumped", len(dump))
print("status/iters/mu of dumped:", [(int(d["status"]), d["iters"], f"{d['kkt'][2]:.1e}", f"{d['kkt'][1]:.1e}") for d in dump])
import os
os.makedirs("gpurun_out", exist_ok=True)
keys = [k for k in dump[0] if k not in ("L",)]
np.savez("gpurun_out/lmpc_fail.npz", L=tr["L"], **{k: np.stack([np.asarray(d[k]) for d in dump], -1) for k in keys})
