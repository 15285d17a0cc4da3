"""Closed loop on the CPU with the twin: cold solves against warm (active-set) solves -- hit rate, iteration counts, equality of the answers.
usage: warm_loop.py [cars] [steps] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(__file__))
from common import *
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tr = pkg.workloads.synthetic_track("barc")
cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
rng = np.random.default_rng(5)
s0 = rng.uniform(0, tr["L"], B)
x = np.stack([s0, rng.uniform(-0.1, 0.1, B), rng.normal(0, 0.03, B), 0.8 * S.track_lookup(tr["vel"], s0, tr["L"]) * 0.9, rng.normal(0, 0.02, B), rng.normal(0, 0.1, B)], axis=1)
u = np.zeros((B, 2))
dt, sc = 0.025, 0.9
inp = S.cold_start_inputs(cfg, veh, tr, x, u, dt, speed_scale=sc)
sol = cbind.solve_batch(cfg, veh, inp)        # the first solve is cold
hits = tot = 0; it_w, it_c, worst = [], [], 0.0
t_w = t_c = 0.0
for k in range(steps):
    ok = sol["status"] == 0
    U0 = np.where(ok[None, :], sol["U_optm"][:, 0, :], inp["U_ref"][:, 0, :])
    x = S.plant_step(veh, tr, x, U0.T, dt / 2, 2)
    Xp = np.where(ok[None, None, :], sol["X_optm"], inp["X_ref"]); Up = np.where(ok[None, None, :], sol["U_optm"], inp["U_ref"])
    nxt = S.shift_inputs(cfg, veh, tr, Xp, Up, dt, speed_scale=sc)
    nxt["x_ic"], nxt["u_ic"] = x.T.copy(), U0.copy()
    # cars whose solve failed restart cold (as closed_loop.run does)
    if (~ok).any():
        cold = S.cold_start_inputs(cfg, veh, tr, x[~ok], U0.T[~ok], dt, speed_scale=sc)
        for key in ("X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref"):
            nxt[key][..., ~ok] = cold[key]
    inp = nxt
    t0 = time.perf_counter(); c = cbind.solve_batch(cfg, veh, inp); t_c += time.perf_counter() - t0
    t0 = time.perf_counter(); wm = cbind.solve_batch(cfg, veh, inp, warm=True); t_w += time.perf_counter() - t0
    both = (c["status"] == 0) & (wm["status"] == 0)
    assert ((c["status"] == 0) == (wm["status"] == 0)).all(), (c["status"], wm["status"])
    e = max(np.abs((wm["X_optm"] - c["X_optm"]) / P.SCALE_X[:, None, None])[..., both].max(), np.abs((wm["dU_optm"] - c["dU_optm"]) / P.SCALE_U[:, None, None])[..., both].max())
    worst = max(worst, e)
    hit = both & (wm["iters"] <= 2)
    hits += hit.sum(); tot += both.sum(); it_w += list(wm["iters"][both]); it_c += list(c["iters"][both])
    sol = wm
it_w, it_c = np.array(it_w), np.array(it_c)
print(f"N = {N}: {B} cars x {steps} periods: warm accepted (<= 2 rounds) {hits / tot:.3f}; iterations warm mean {it_w.mean():.2f} (hits {it_w[it_w <= 2].mean():.2f}, misses {it_w[it_w > 2].mean() if (it_w > 2).any() else 0:.2f}) "
      f"cold mean {it_c.mean():.2f}; max warm {it_w.max()} cold {it_c.max()}; warm vs cold answers worst {worst:.1e}; twin time warm {t_w:.2f} s cold {t_c:.2f} s")
