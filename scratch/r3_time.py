"""Round-3 timing / accuracy sweep of the QP kernel, polish on and off (one GPU).  Prints one line per case:
kernel ms (HIP events around the QP launches of lmpc_solve_batch*), statuses, iterations, and for the fp32 / mixed
cases the scaled distance from the fp64 answers of the same batch."""
import sys, os, json, numpy as np, torch
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import importlib
pkg = importlib.import_module("racing-lmpc-ros2_amd")
if os.environ.get("LMPC_LIB"):
    pkg.capi.library_path = lambda: Path(os.environ["LMPC_LIB"])
SX = np.array([2000, 10, 0.1, 80, 2, 2.0]); SU = np.array([10, 0.3])
dev = torch.device("cuda:0")

def scaled_err(a, b):
    ex = (torch.abs(a["X_optm"] - b["X_optm"]).cpu().numpy() / SX[:, None, None]).max(axis=(0, 1))
    eu = (torch.abs(a["U_optm"] - b["U_optm"]).cpu().numpy() / SU[:, None, None]).max(axis=(0, 1))
    return np.maximum(ex, eu)

def timed(sv, fn, reps=5):
    sv.enable_timing(True)
    fn(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        o = fn(); torch.cuda.synchronize(); ms.append(sv.last_kernel_ms()[1])
    return o, float(np.median(ms))

def case(kind, N, B, precisions=("f64",), seeds=0):
    tr = pkg.workloads.synthetic_track("putnam" if kind == "iac" else "barc")
    res = {}
    ref = None
    for pol in ((0, -1, 1) if "mixed" in precisions else (0, -1)):
        if kind == "lmpc":
            cfg = dict(pkg.presets.barc_lmpc(N, 5)); laps = pkg.workloads.synthetic_laps(tr, 5)
            x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=seeds)
            veh = pkg.presets.barc_vehicle()
        elif kind == "iac":
            cfg = dict(pkg.presets.iac_tracking_mpc(N)); veh = pkg.presets.iac_vehicle()
            x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=seeds + 1)
        else:
            cfg = dict(pkg.presets.barc_tracking_mpc(N)); veh = pkg.presets.barc_vehicle()
            x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=seeds)
        cfg["polish"] = pol
        sv = pkg.Solver(cfg, veh, device=0)
        sv.reserve(B)
        inp = sv.prepare(tr, x.T.copy(), 0.025)
        inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
        kw = {}
        if kind == "lmpc":
            sv.set_safe_set(laps, tr["L"])
            s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
            kk = (s0 - s_last).abs() + L / 2
            q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
            ss_x, ss_j, _ = sv.ss_query(q)
            kw = dict(ss_x=ss_x, ss_j=ss_j)
        for prec in (precisions if pol != 1 else ("mixed",)):
            out = sv.alloc_outputs(B)
            if kind == "lmpc":
                out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
            if prec == "f32":
                inp32 = {k: (v.float() if hasattr(v, "float") and v.dtype == torch.float64 else v) for k, v in inp.items()}
                o, ms = timed(sv, lambda: sv.solve_f32(inp32))
                o = {k: (v.double() if hasattr(v, "double") and v.dtype == torch.float32 else v) for k, v in o.items()}
            else:
                o, ms = timed(sv, lambda: sv.solve(inp, out, mixed=(prec == "mixed"), **kw))
            st = np.bincount(o["status"].cpu().numpy(), minlength=4); it = o["iters"].cpu().numpy()
            line = f"{kind:5s} N={N:2d} B={B:6d} {prec:5s} polish={'on ' if pol == 0 else 'off' if pol < 0 else 'on, one pass'}: QP kernel {ms:8.3f} ms  {B / ms / 1e3:7.3f} M/s  status {st.tolist()}  iters mean {it.mean():.2f} max {it.max()}"
            if prec == "f64" and pol == 0:
                ref = {k: o[k].clone() for k in ("X_optm", "U_optm", "status")}
            elif ref is not None:
                both = (ref["status"] == 0).cpu().numpy() & (o["status"] == 0).cpu().numpy()
                e = scaled_err(o, ref)[both]
                line += f"  vs polished fp64: median {np.median(e):.1e} 99% {np.quantile(e, .99):.1e} 99.9% {np.quantile(e, .999):.1e} max {e.max():.1e} frac>1e-3 {np.mean(e > 1e-3):.5f}"
            print(line, flush=True)
        sv.close()

if __name__ == "__main__":
    which = sys.argv[1:] or ["trk20", "trk40", "trk60", "lmpc", "iac"]
    if "trk20" in which: case("barc", 20, 4096); case("barc", 20, 65536)
    if "trk40" in which: case("barc", 40, 4096)
    if "trk60" in which: case("barc", 60, 4096)
    if "trkmix" in which:
        for n in (20, 40, 60, 80): case("barc", n, 4096, ("f64", "mixed"))
    if "trk80" in which: case("barc", 80, 4096)
    if "trk48" in which: case("barc", 48, 4096)
    if "lmpc" in which: case("lmpc", 20, 4096, ("f64", "mixed")); case("lmpc", 20, 32768, ("f64", "mixed"))
    if "lmpc40" in which: case("lmpc", 40, 4096, ("f64", "mixed"))
    if "iac" in which: case("iac", 40, 8192, ("f64", "mixed", "f32"))
