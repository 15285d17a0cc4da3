"""Round 6: why the learning warm start is refused in the closed loop -- how the optimum's support moves from one period to the next."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
B = 64
tr = pkg.workloads.synthetic_track("barc")
rng = np.random.default_rng(5)
x0 = torch.as_tensor(np.stack([np.zeros(B), rng.uniform(-0.05, 0.05, B), np.zeros(B), np.full(B, 1.2), np.zeros(B), np.zeros(B)]), dtype=torch.float64, device="cuda")
u0 = torch.zeros((2, B), dtype=torch.float64, device="cuda")
for adv in (0, 1, 2):
    for rounds in (2, 4):
        tracker = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
        learner = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
        learner.set_warm_rounds(rounds)
        r = pkg.closed_loop.run_lmpc(tracker, learner, tr, x0, u0, warm_laps=2, learn_laps=2, warm=True, advance=adv)
        print("advance %d rounds %d: hit rate %.3f lap times %s" % (adv, rounds, r["warm_hit_rate"], np.round(r["lap_times"], 3).tolist()), flush=True)
        tracker.close(); learner.close()
# how the support moves: run the experiment cold with idx, log car 0's support codes per period
tracker = pkg.Solver(pkg.presets.barc_tracking_mpc(20), pkg.presets.barc_vehicle(), device=0)
learner = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
import types
log = []
orig = learner.solve
def spy(inp, out=None, **kw):
    o = orig(inp, out, **kw)
    if kw.get("ss_idx") is not None:
        lam = o["convex_combi_optm"][:, 0].cpu().numpy(); idx = kw["ss_idx"][:, 0].cpu().numpy()
        sup = np.where(lam > 1e-9)[0]
        log.append([(int(idx[j]) >> 2, int(idx[j]) & 3, round(float(lam[j]), 3)) for j in sup])
    return o
learner.solve = spy
# run_lmpc uses ss_idx only in warm mode: use warm with rounds irrelevant
r = pkg.closed_loop.run_lmpc(tracker, learner, tr, x0, u0, warm_laps=2, learn_laps=1, warm=True, advance=1)
for k in range(0, 0):
    print(k, log[k])
