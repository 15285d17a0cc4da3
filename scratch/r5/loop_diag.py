"""Which outputs of lmpc_loop_advance_batch differ from the composition of the separate entry points, where, by how much."""
import sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package
from oracle import scenario as S
pkg = load_package()
KEYS = ("X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
B, N, dt = 1024, 20, 0.025
tr = pkg.workloads.synthetic_track("barc")
sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
trk = sv.device_track(tr)
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tr["L"], B)
x = torch.as_tensor(np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), 1.3 * S.track_lookup(tr["vel"], s0, tr["L"]), rng.normal(0, 0.02, B), rng.normal(0, 0.1, B)]), device="cuda")
u_prev = torch.zeros((2, B), dtype=torch.float64, device="cuda")
inp = sv.prepare(trk, x, dt, speed_scale=0.9)
inp["x_ic"], inp["u_ic"] = x, u_prev
out = sv.solve(inp)
out["status"][::7] = 1
ok = out["status"] == 0
for restart in (True, False):
    u_apply = torch.where(ok[None, :], out["U_optm"][:, 0, :], inp["U_ref"][:, 0, :]).contiguous()
    x_ref = sv.plant_step(trk, x.clone(), u_apply, dt / 2, 2)
    nxt = sv.shift(trk, inp, out, dt, speed_scale=0.9)
    if restart:
        sv.prepare_failed(trk, x_ref, out["status"], nxt, dt, speed_scale=0.9)
    inp2 = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()}
    x2, u2 = x.clone(), u_prev.clone()
    sv.loop_advance(trk, inp2, out, x2, u2, dt, dt / 2, 2, speed_scale=0.9, restart_failed=restart)
    torch.cuda.synchronize()
    print("restart", restart, "x equal", torch.equal(x2, x_ref), "max", float((x2 - x_ref).abs().max()), "u equal", torch.equal(u2, u_apply))
    for k in KEYS:
        dfr = (inp2[k] - nxt[k]).abs()
        bad = dfr > 0
        if bad.any():
            idx = bad.nonzero()
            knots = sorted(set(idx[:, -2].tolist()))
            cars = idx[:, -1]
            print("  ", k, "differs in", int(bad.sum()), "entries; max", float(dfr.max()), "knots", knots[:10], "of ok cars", int(ok[cars].sum()), "of failed cars", int((~ok[cars]).sum()),
                  "components", sorted(set(idx[:, 0].tolist())) if idx.shape[1] == 3 else "-")
        else:
            print("  ", k, "equal")
