"""Dump the unsolved problems of the full-size learning batch (inputs + safe set) for the CPU twin / dense oracle."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
from oracle import cbind, params as P
import test_gpu_mixed_lmpc as T
sv, tr, laps, inp, ss_x, ss_j = T._s160(pkg, 4096)
o = T._solve(sv, inp, ss_x, ss_j, False)
bad = np.where(o["status"] != 0)[0]
print("unsolved", bad, o["status"][bad], o["iters"][bad], o["kkt"][:, bad].T)
npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
sel = list(bad[:4]) + [0, 1]
sub = {k: (v[..., sel] if isinstance(v, np.ndarray) and v.ndim >= 1 else v) for k, v in npinp.items()}
sx, sj = ss_x.cpu().numpy()[..., sel], ss_j.cpu().numpy()[..., sel]
tw = cbind.solve_batch(P.barc_lmpc(20, 5), P.barc_vehicle(), sub, ss_x=sx, ss_j=sj)
print("twin status", tw["status"], "iters", tw["iters"])
np.savez("/root/repo/gpurun_out/lmpc_unsolved.npz", ss_x=sx, ss_j=sj, sel=np.array(sel), **{k: v for k, v in sub.items() if isinstance(v, np.ndarray)})
