import sys, numpy as np
sys.path.insert(0, "scratch"); sys.path.insert(0, ".")
import r3_sqp_proto as R
from r3_sqp_proto import *
for seed, B in ((8, 96), (3, 256)):
    orig = R.setup
    R.setup = lambda B_=B, seed_=seed, N=20: orig(B, seed)
    veh, cfg, tr, x, u, inp = R.setup()
    conv, _ = R.run("both", B=B, seed=seed, backoff=6, verbose=False)
    R.setup = orig
    rho = np.zeros(len(x))
    for b in range(len(x)):
        A, B_, g = Q.linearise(cfg, veh, S.problem(inp, b))
        rho[b] = max(np.abs(np.linalg.eigvals(A[i])).max() for i in range(A.shape[0]))
    ok = rho <= 2.0
    print(f"seed {seed} B {B}: rho<=2: {ok.sum()} conv among them {conv[ok].mean():.4f}; all {conv.mean():.3f}; failed with rho<=2:", [(int(b), round(float(x[b,3]),2), round(float(rho[b]),2)) for b in np.nonzero(ok & ~conv)[0]])
