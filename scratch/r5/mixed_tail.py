"""The mixed entry's tail on the learning problem with 96 points (tests/dispatch_sweep.py: 1.0 - 1.3e-3 at N = 11, 15, 16, 18): were the
worst problems verified by the fp32 polish (status 0 with lmpc_config.polish = 1) or re-solved by the fp64 pass?"""
import sys, numpy as np, torch, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]; sys.path.insert(0, str(ROOT))
from oracle import params as P
pkg = importlib.import_module("racing-lmpc-ros2_amd")
dev = torch.device("cuda:0"); SX, SU = P.SCALE_X[:, None, None], P.SCALE_U[:, None, None]
def npd(d): return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in d.items()}
B = 1024
out_dir = ROOT / "gpurun_out" / "r5f_mixed_tail"; out_dir.mkdir(parents=True, exist_ok=True)
for N in (11, 15, 16, 18):
    tr = pkg.workloads.synthetic_track("barc"); laps = pkg.workloads.synthetic_laps(tr, 3)
    x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    res = {}
    for tag, pol in (("two_pass", 0), ("one_pass_marks", 1)):
        pc = dict(pkg.presets.barc_lmpc(N, 3)); pc["polish"] = pol
        sv = pkg.Solver(pc, pkg.presets.barc_vehicle(), device=0); sv.reserve(B)
        inp = sv.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
        sv.set_safe_set(laps, tr["L"])
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]; kk = (s0 - s_last).abs() + L / 2
        q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
        ss_x, ss_j, _ = sv.ss_query(q)
        def solve(**kw):
            o = sv.alloc_outputs(B); o["convex_combi_optm"] = torch.zeros((96, B), dtype=torch.float64, device=dev)
            return npd(sv.solve(inp, o, ss_x=ss_x, ss_j=ss_j, **kw))
        if pol == 0: res["f64"] = solve()
        res[tag] = solve(mixed=True)
        keep = (npd(inp), ss_x.cpu().numpy(), ss_j.cpu().numpy())
        sv.close()
    o64, om, o1 = res["f64"], res["two_pass"], res["one_pass_marks"]
    e = np.maximum(np.abs((om["X_optm"] - o64["X_optm"]) / SX).max((0, 1)), np.abs((om["U_optm"] - o64["U_optm"]) / SU).max((0, 1)))
    idx = np.argsort(e)[-6:]
    print(f"N = {N}: marks (status 3 in the one-pass run) {int((o1['status'] == 3).sum())} of {B}; worst problems:")
    for b in idx[::-1]:
        print(f"   problem {b}: mixed vs fp64 {e[b]:.1e}; one-pass status {o1['status'][b]} iters {o1['iters'][b]} kkt {o1['kkt'][:, b]}; two-pass iters {om['iters'][b]} kkt {om['kkt'][:, b]}; fp64 iters {o64['iters'][b]} kkt {o64['kkt'][:, b]}")
    np.savez(out_dir / f"lrn96_N{N}.npz", idx=idx, **{"in_" + k: np.asarray(keep[0][k])[..., idx] for k in ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")},
             L=float(tr["L"]), ss_x=keep[1][..., idx], ss_j=keep[2][..., idx], **{"m_" + k: om[k][..., idx] for k in ("X_optm", "U_optm", "dU_optm", "convex_combi_optm")},
             **{"k_" + k: o64[k][..., idx] for k in ("X_optm", "U_optm", "dU_optm", "convex_combi_optm")})
