"""The twin built with -D overrides (algorithm experiments that need no GPU: the kernel runs the twin's iteration), against the
cached dense optima.  usage: twin_variant.py trk20,iac40,lrn20 [-DPOLISH_MU=1e-6 ...]"""
import os, sys, subprocess, ctypes, hashlib
sys.path.insert(0, os.path.dirname(__file__))
from common import *
flags = sys.argv[2:]
tag = hashlib.md5(" ".join(flags).encode()).hexdigest()[:8]
so = ROOT / f"scratch/r5/cache/twin_{tag}.so"
subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-std=c11", f"-I{ROOT}/include", "-shared", "-o", str(so), str(ROOT / "oracle/c/lmpc_oracle.c"), "-lm"] + flags)
cbind._LIB = ctypes.CDLL(str(so))
print("flags:", flags)
for what in sys.argv[1].split(","):
    c = np.load(ROOT / f"scratch/r5/cache/{what}.npz")
    B = c["status"].size
    cfg, veh, inp, ss_x, ss_j = batch(what, 4096)
    tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, b0=0, b1=B)
    ok = (tw["status"][:B] == 0) & (c["status"] == 0)
    e = np.maximum(np.abs((tw["X_optm"][:, :, :B] - c["X_optm"]) / P.SCALE_X[:, None, None]).max((0, 1)), np.abs((tw["U_optm"][:, :, :B] - c["U_optm"]) / P.SCALE_U[:, None, None]).max((0, 1)))
    ed = np.abs((tw["dU_optm"][:, :, :B] - c["dU_optm"]) / P.SCALE_U[:, None, None]).max((0, 1))
    em = np.maximum(e, ed)[ok]
    it = tw["iters"][:B][ok]
    print(f"  {what}: {B} problems, both solved {ok.sum()}, twin status {np.bincount(tw['status'][:B], minlength=3).tolist()}; iters mean {it.mean():.3f} max {it.max()} hist {np.bincount(it).tolist()}; "
          f"err max {em.max():.1e} > 1e-6: {(em > 1e-6).sum()} > 1e-7: {(em > 1e-7).sum()}", flush=True)
