"""Per-phase cycle accounting of lmpc_solve_kernel (profiling build, GPU box)."""
import ctypes as C, sys, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package
pkg = load_package()
import importlib
capi = importlib.import_module(pkg.__name__ + ".capi")
import os
capi.library_path = lambda: Path(os.environ.get("LMPC_PROF_LIB", str(ROOT / "racing-lmpc-ros2_amd" / "lib" / "liblmpc_hip_prof.so")))
capi._LIB = None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
LMPC = len(sys.argv) > 3 and sys.argv[3] == "lmpc"
MIXED = len(sys.argv) > 4 and sys.argv[4] == "mixed"
tr = pkg.workloads.synthetic_track("barc")
kw = {}
if LMPC:
    laps = pkg.workloads.synthetic_laps(tr, 5)
    solver = pkg.Solver(pkg.presets.barc_lmpc(N, 5), pkg.presets.barc_vehicle(), 0)
    solver.set_safe_set(laps, tr["L"])
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
else:
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), 0)
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], 0)
inp = solver.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
if LMPC:
    s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
    kk = (s0 - s_last).abs() + L / 2
    query = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
    ss_x, ss_j, _ = solver.ss_query(query)
    kw = dict(ss_x=ss_x, ss_j=ss_j)
if "w1" in sys.argv:
    solver.set_waves_per_problem(1)   # (since round 6 the library takes two waves from N = 41 on by itself)
if "w2" in sys.argv:
    solver.set_waves_per_problem(2)   # (round 6: the two-wave kernels; the clock is the chain wave's)
out = solver.alloc_outputs(B)
out["kkt"] = torch.zeros((20, B), dtype=torch.float64, device="cuda")
for _ in range(3):
    solver.solve(inp, out, mixed=MIXED, **kw)
torch.cuda.synchronize()
k = out["kkt"].cpu().numpy()
it = out["iters"].cpu().numpy()
names = ["load", "init(factor+rollout)", "rows+reduce(+terminal block)", "factor", "riccati_solve (all)", "(ModelStream::wait, inside the sweeps)", "schur+steps", "update/exit",
         "  bwd sweep nrhs1", "  bwd sweep nrhs2", "  fwd sweep nrhs1", "  fwd sweep nrhs2", "gradient", "x13", "x14", "x15"]
tot = k[:16].sum(0)
print("mean iters %.2f; mean cycles/wave %.0f" % (it.mean(), tot.mean()))
for n, v in zip(names, k[:16]):
    print("%-22s %10.0f cycles  %5.1f %%   per-iter %8.0f" % (n, v.mean(), 100 * v.mean() / tot.mean(), v.mean() / it.mean()))

w0, w1, hw, xcc = k[16], k[17], k[18].astype(np.int64), k[19].astype(np.int64)
span = (w1.max() - w0.min()) / 100e6
dur = (w1 - w0) / 100e6
print("kernel span %.3f ms; wave duration mean %.3f ms (min %.3f max %.3f); effective clock %.2f GHz" % (span * 1e3, dur.mean() * 1e3, dur.min() * 1e3, dur.max() * 1e3, tot.mean() / dur.mean() / 1e9))
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
key = xcc * 10000 + se * 100 + sh * 16 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu):", len(uniq), "waves per CU: min %d max %d mean %.1f" % (cnt.min(), cnt.max(), cnt.mean()))
# concurrency: for each CU, max overlapping waves
mx = []
for u in uniq[:64]:
    m = key == u
    ev = sorted([(t, 1) for t in w0[m]] + [(t, -1) for t in w1[m]])
    c = best = 0
    for _, d in ev:
        c += d; best = max(best, c)
    mx.append(best)
print("max concurrent waves per CU (first 64 CUs):", np.bincount(mx))
print("xcc histogram", np.bincount(xcc))
# timeline: when do waves start / end (ms from kernel start)
t0 = w0.min()
st_ms = (w0 - t0) / 100e6 * 1e3; en_ms = (w1 - t0) / 100e6 * 1e3
print("start times: pct 50/90/99/max", np.percentile(st_ms, [50, 90, 99, 100]).round(3))
print("end times:   pct 1/10/50/90/99/max", np.percentile(en_ms, [1, 10, 50, 90, 99, 100]).round(3))
order = np.argsort(en_ms)[-8:]
print("last finishers: idx", order, "start", st_ms[order].round(3), "dur", dur[order].round(3) * 1e3, "iters", it[order])
print("dur by iters:", {int(k): round(float(dur[it == k].mean() * 1e3), 3) for k in np.unique(it)})
