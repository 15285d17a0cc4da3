"""Dump the unsolved problems of the IAC N = 40 bench batch and of the LMPC 32768 batch for the dense oracle."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from oracle import cbind, params as P
tr = pkg.workloads.synthetic_track("putnam")
sv = pkg.Solver(pkg.presets.iac_tracking_mpc(40), pkg.presets.iac_vehicle(), device=0)
x, u = pkg.workloads.sample_initial_states("putnam", 8192, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
inp = sv.prepare(tr, x.T.copy(), 0.025)
inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
o = {k: v.cpu().numpy() for k, v in sv.solve(inp).items() if hasattr(v, "cpu")}
bad = np.where(o["status"] != 0)[0]
print("IAC unsolved", bad, o["status"][bad], o["iters"][bad], o["kkt"][:, bad].T)
npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
sel = list(bad[:4]) + [0]
sub = {k: (v[..., sel] if isinstance(v, np.ndarray) and v.ndim >= 1 else v) for k, v in npinp.items()}
tw = cbind.solve_batch(P.iac_tracking_mpc(40), P.iac_vehicle(), sub)
print("twin status", tw["status"], "iters", tw["iters"])
np.savez("/root/repo/gpurun_out/iac_unsolved.npz", sel=np.array(sel), **{k: v for k, v in sub.items() if isinstance(v, np.ndarray)})
import test_gpu_mixed_lmpc as T
sv, tr, laps, inp, ss_x, ss_j = T._s160(pkg, 32768)
o = T._solve(sv, inp, ss_x, ss_j, False)
bad = np.where(o["status"] != 0)[0]
print("LMPC 32768 unsolved", bad, o["status"][bad], o["iters"][bad])
npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
sel = list(bad[:6]) + [0]
sub = {k: (v[..., sel] if isinstance(v, np.ndarray) and v.ndim >= 1 else v) for k, v in npinp.items()}
np.savez("/root/repo/gpurun_out/lmpc32k_unsolved.npz", sel=np.array(sel), ss_x=ss_x.cpu().numpy()[..., sel], ss_j=ss_j.cpu().numpy()[..., sel], **{k: v for k, v in sub.items() if isinstance(v, np.ndarray)})
