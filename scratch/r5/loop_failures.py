"""Closed loop on the CPU with the twin, on the reference's BARC track and soak_warm.py's start states: which solves fail, and why.
usage: loop_failures.py [cars] [periods]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import *
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 666
tab = pkg.workloads.track_from_file(ROOT / "tests/golden/barc_track/15_barc_optm.txt", 1024)
cfg, veh = P.barc_tracking_mpc(20), P.barc_vehicle()
rng = np.random.default_rng(3)
s0 = rng.uniform(0, tab["L"], 4096)
x = np.stack([s0, rng.uniform(-0.08, 0.08, 4096), rng.normal(0, 0.03, 4096), rng.uniform(0.6, 0.95, 4096) * np.interp(s0, np.arange(1024) * tab["L"] / 1024, tab["vel"]),
              np.zeros(4096), np.zeros(4096)], axis=1)[:B]
u = np.zeros((B, 2))
dt, sc = 0.025, 0.9
inp = S.cold_start_inputs(cfg, veh, tab, x, u, dt, speed_scale=sc)
sol = cbind.solve_batch(cfg, veh, inp)
fails = []
xmax, xmin = np.array(cfg.x_max), np.array(cfg.x_min)
for k in range(steps):
    ok = sol["status"] == 0
    for b in np.nonzero(~ok)[0]:
        xi = inp["x_ic"][:, b]
        out_of_box = bool(((xi > xmax) | (xi < xmin)).any())
        fails.append((k, int(b), int(sol["status"][b]), int(sol["iters"][b]), out_of_box, xi.copy(), {kk: (np.array(v[..., b]) if isinstance(v, np.ndarray) else v) for kk, v in inp.items()}))
    U0 = np.where(ok[None, :], sol["U_optm"][:, 0, :], inp["U_ref"][:, 0, :])
    x = S.plant_step(veh, tab, x, U0.T, dt / 2, 2)
    Xp = np.where(ok[None, None, :], sol["X_optm"], inp["X_ref"]); Up = np.where(ok[None, None, :], sol["U_optm"], inp["U_ref"])
    nxt = S.shift_inputs(cfg, veh, tab, Xp, Up, dt, speed_scale=sc)
    nxt["x_ic"], nxt["u_ic"] = x.T.copy(), U0.copy()
    if (~ok).any():
        cold = S.cold_start_inputs(cfg, veh, tab, x[~ok], U0.T[~ok], dt, speed_scale=sc)
        for key in ("X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref"):
            nxt[key][..., ~ok] = cold[key]
    inp = nxt
    sol = cbind.solve_batch(cfg, veh, inp)
print(f"{B} cars x {steps} periods: {len(fails)} failed solves on {len(set(f[1] for f in fails))} cars; status histogram {np.bincount([f[2] for f in fails], minlength=3).tolist()}; "
      f"measured state outside the hard state box: {sum(f[4] for f in fails)}")
# the ones inside the box: what does the dense oracle say?
inside = [f for f in fails if not f[4]]
dense_solved = 0
for k, b, st, it, _, xi, prob in inside:
    one = {kk: (v[..., None] if isinstance(v, np.ndarray) else v) for kk, v in prob.items()}
    qp = Q.build_qp(cfg, veh, S.problem(one, 0))
    y, info = Q.solve_dense(qp)
    dense_solved += info["status"] == 0
    print(f"  period {k} car {b}: twin status {st} after {it} iterations; x_ic {np.array2string(xi, precision=3)}; dense status {info['status']} iterations {info.get('iters')}")
print(f"failed solves with the measured state inside the box: {len(inside)}; of those the dense oracle solves: {dense_solved}")
