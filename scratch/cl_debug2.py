import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]; sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
solver = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
tr = pkg.workloads.synthetic_track("barc")
rng = np.random.default_rng(2); B = 512
x0 = np.stack([rng.uniform(0, tr["L"], B), rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), rng.uniform(1.0, 1.5, B), np.zeros(B), np.zeros(B)])
trk = solver.device_track(tr)
x = torch.as_tensor(x0, device="cuda"); u_prev = torch.zeros((2, B), dtype=torch.float64, device="cuda")
inp = solver.prepare(trk, x, 0.025, speed_scale=0.9); out = solver.alloc_outputs(B)
watch = None
for k in range(60):
    inp["x_ic"] = x; inp["u_ic"] = u_prev
    solver.solve(inp, out)
    ok = out["status"] == 0
    bad = torch.nonzero(~ok).flatten().tolist()
    if bad and watch is None and k > 3:
        watch = bad[0]
        np.savez('/root/repo/gpurun_out/cl_fail.npz', **{kk: (v[..., watch:watch+1].cpu().numpy() if hasattr(v,'cpu') else v) for kk, v in inp.items()})
    if bad:
        for b in bad[:4]:
            print(k, "car", b, "status", int(out["status"][b]), "iters", int(out["iters"][b]), "kkt", out["kkt"][:, b].cpu().numpy(), "x", x[:, b].cpu().numpy().round(3), "u_prev", u_prev[:, b].cpu().numpy().round(4))
    u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
    solver.plant_step(trk, x, u_apply, 0.0125, 2)
    u_prev = u_apply
    inp = solver.shift(trk, inp, out, 0.025, speed_scale=0.9)
