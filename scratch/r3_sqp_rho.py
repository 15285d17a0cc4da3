import sys, numpy as np
sys.path.insert(0, "scratch"); sys.path.insert(0, ".")
from r3_sqp_proto import *
veh, cfg, tr, x, u, inp = setup()
conv, _ = run("both", backoff=6, verbose=False)
rho = np.zeros(len(x))
for b in range(len(x)):
    A, B_, g = Q.linearise(cfg, veh, S.problem(inp, b))
    rho[b] = max(np.abs(np.linalg.eigvals(A[i])).max() for i in range(A.shape[0]))
o = np.argsort(x[:, 3])
for b in o[:40]:
    print(f"b={b:2d} vx0={x[b,3]:.2f} rho={rho[b]:8.2f} conv={conv[b]}")
print("converged: max rho", rho[conv].max(), " failed: min rho", rho[~conv].min())
