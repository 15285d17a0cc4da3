import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from oracle import params as P
B, N = 4096, 40
tr = pkg.workloads.synthetic_track("barc")
laps = pkg.workloads.synthetic_laps(tr, 5)
sv = pkg.Solver(pkg.presets.barc_lmpc(N, 5), pkg.presets.barc_vehicle(), device=0)
sv.set_safe_set(laps, tr["L"])
x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), device="cuda")
s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
kk = (s0 - s_last).abs() + L / 2
q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
ss_x, ss_j, _ = sv.ss_query(q)
def run(mixed):
    out = sv.alloc_outputs(B)
    o = sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=mixed)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, mixed=mixed)
    e1.record(); torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in o.items() if hasattr(v, "cpu")}, e0.elapsed_time(e1) / 10
o64, t64 = run(False); om, tm = run(True)
ok = (o64["status"] == 0) & (om["status"] == 0)
e = np.abs((om["X_optm"] - o64["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))[ok]
print("N=40 learning: fp64 %.3f ms (%.2f M/s) solved %.4f iters %.1f | mixed %.3f ms (%.2f M/s) solved %.4f iters %.1f" % (t64, B / t64 / 1e3, (o64["status"] == 0).mean(), o64["iters"].mean(), tm, B / tm / 1e3, (om["status"] == 0).mean(), om["iters"].mean()))
print("mixed vs fp64: median %.1e p90 %.1e p99 %.1e max %.1e" % (np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
