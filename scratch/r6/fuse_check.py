"""Round 6: the fused factorisation (predictor's backward sweep inside riccati_factor) against the five-chain iteration.
usage: fuse_check.py TAG        (library by LMPC_HIP_LIBRARY; writes gpurun_out/fuse_TAG.npz)
       fuse_check.py cmp A B    (compares two such files)"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
if sys.argv[1] == "cmp":
    a, b = (np.load(ROOT / "gpurun_out" / ("fuse_%s.npz" % t)) for t in sys.argv[2:4])
    for name in sorted({k.rsplit("/", 1)[0] for k in a.files}):
        ok = (a[name + "/status"] == 0) & (b[name + "/status"] == 0)
        e = np.abs(a[name + "/X"] - b[name + "/X"]).reshape(-1, ok.size).max(axis=0)
        print("%-28s ms %s %.3f -> %s %.3f (x%.3f) | status equal %s | X max abs diff %.1e (bitwise equal on %.3f) | iters equal on %.4f, mean %.2f / %.2f"
              % (name, sys.argv[2], a[name + "/ms"], sys.argv[3], b[name + "/ms"], a[name + "/ms"] / b[name + "/ms"], np.array_equal(a[name + "/status"], b[name + "/status"]),
                 e[ok].max(), (e == 0).mean(), (a[name + "/iters"] == b[name + "/iters"]).mean(), a[name + "/iters"].mean(), b[name + "/iters"].mean()))
    sys.exit(0)
import torch
from __graft_entry__ import load_package
pkg = load_package()
B = 4096
res = {}
def timed(sv, run):
    out = run()
    torch.cuda.synchronize()
    sv.enable_timing(True)
    ms = []
    for _ in range(10):
        out = run()
        torch.cuda.synchronize()
        ms.append(sv.last_kernel_ms()[1])
    sv.enable_timing(False)
    return out, float(np.median(ms))
cases = [("barc", 20, "f64"), ("barc", 24, "f64"), ("barc", 40, "f64"), ("barc", 60, "f64"), ("barc", 80, "f64"), ("iac", 40, "f64"), ("iac", 40, "f32"), ("iac", 40, "mixed"),
         ("lmpc", 20, "f64"), ("lmpc", 20, "mixed"), ("lmpc", 40, "f64"), ("lmpc", 60, "f64")]
if len(sys.argv) > 2:  # fuse_check.py TAG kind:N:prec,...
    cases = [(c.split(":")[0], int(c.split(":")[1]), c.split(":")[2]) for c in sys.argv[2].split(",")]
for kind, N, prec in cases:
    tr = pkg.workloads.synthetic_track("putnam" if kind == "iac" else "barc")
    kw = {}
    if kind == "iac":
        sv = pkg.Solver(pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), device=0)
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1)
    elif kind == "lmpc":
        laps = pkg.workloads.synthetic_laps(tr, 5)
        sv = pkg.Solver(pkg.presets.barc_lmpc(N, 5), pkg.presets.barc_vehicle(), device=0)
        sv.set_safe_set(laps, tr["L"])
        x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=0)
    else:
        sv = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=0)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    if kind == "lmpc":
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = (s0 - s_last).abs() + L / 2
        query = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
        ss_x, ss_j, _ = sv.ss_query(query)
        kw = dict(ss_x=ss_x, ss_j=ss_j)
    if prec == "f32":
        out, ms = timed(sv, lambda: sv.solve_f32(inp))
    else:
        o = sv.alloc_outputs(B)
        out, ms = timed(sv, lambda: sv.solve(inp, o, mixed=(prec == "mixed"), **kw))
    name = "%s_n%d_%s" % (kind, N, prec)
    res[name + "/X"] = out["X_optm"].double().cpu().numpy()
    res[name + "/status"] = out["status"].cpu().numpy()
    res[name + "/iters"] = out["iters"].cpu().numpy()
    res[name + "/ms"] = ms
    print("%-24s kernel %.3f ms, solved %d of %d, iters %.2f" % (name, ms, (res[name + "/status"] == 0).sum(), B, res[name + "/iters"].mean()), flush=True)
    sv.close()
np.savez_compressed(ROOT / "gpurun_out" / ("fuse_%s.npz" % sys.argv[1]), **res)
