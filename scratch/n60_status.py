"""BARC tracking at N = 60 on the bench workload: who fails, and does the dense solver agree?"""
import sys, importlib
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
pkg = importlib.import_module("racing-lmpc-ros2_amd")
from oracle import cbind, params as P, qp as Q, scenario as S
N, B = int(sys.argv[1]) if len(sys.argv) > 1 else 60, 256
tr = pkg.workloads.synthetic_track("barc")
s = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.015, -0.314159], [0.015, 0.314159], seed=0)
inp = s.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
o = {k: v.cpu().numpy() for k, v in s.solve(inp).items() if hasattr(v, "cpu")}
npinp = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in inp.items()}
cfg, veh = P.barc_tracking_mpc(N), P.barc_vehicle()
twin = cbind.solve_batch(cfg, veh, npinp)
print("kernel status", np.bincount(o["status"], minlength=3), "twin status", np.bincount(twin["status"], minlength=3), "agree", (o["status"] == twin["status"]).mean())
bad = np.where(o["status"] != 0)[0][:8]
for b in bad:
    qp = Q.build_qp(cfg, veh, S.problem(npinp, b))
    yex, info = Q.solve_dense(qp)
    print(f"b={b} kernel st {o['status'][b]} it {o['iters'][b]} rd {o['kkt'][1,b]:.1e} mu {o['kkt'][2,b]:.1e} | twin st {twin['status'][b]} | dense st {info['status']} x0 {x[b].round(2)}", flush=True)
