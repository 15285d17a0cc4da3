"""Round-4 A/B timing of the QP kernel across builds of the library (one GPU).  One process per build:
    LMPC_HIP_LIBRARY=<lib.so> python scratch/r4_ab.py [case ...]
prints one JSON line per case: kernel ms (HIP events around the QP launches inside lmpc_solve_batch*: median of 9 calls),
statuses, iterations, a checksum of the answers (so that builds that should agree bit for bit can be compared), and for the
reduced-precision entries the scaled distance from the fp64 answers of the same build."""
import hashlib
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
# (older builds under A/B lack the newest entry point; nothing here calls it)
pkg.capi._ABI_SYMBOLS = tuple(s for s in pkg.capi._ABI_SYMBOLS if s != "lmpc_query_launch_for")
SX = np.array([2000, 10, 0.1, 80, 2, 2.0])
SU = np.array([10, 0.3])
dev = torch.device("cuda:0")
LIB = os.path.basename(os.environ.get("LMPC_HIP_LIBRARY", "liblmpc_hip.so"))


def scaled_err(a, b):
    ex = (torch.abs(a["X_optm"].double() - b["X_optm"]).cpu().numpy() / SX[:, None, None]).max(axis=(0, 1))
    eu = (torch.abs(a["U_optm"].double() - b["U_optm"]).cpu().numpy() / SU[:, None, None]).max(axis=(0, 1))
    return np.maximum(ex, eu)


def timed(sv, fn, reps=9):
    sv.enable_timing(True)
    fn()
    torch.cuda.synchronize()
    ms, lin = [], []
    for _ in range(reps):
        o = fn()
        torch.cuda.synchronize()
        a, b = sv.last_kernel_ms()
        lin.append(a)
        ms.append(b)
    return o, float(np.median(ms)), float(np.min(ms)), float(np.median(lin))


def digest(o):
    h = hashlib.sha1()
    for k in ("X_optm", "U_optm", "dU_optm", "status", "iters"):
        h.update(o[k].cpu().numpy().tobytes())
    return h.hexdigest()[:12]


def case(kind, N, B, precisions=("f64",), regression=False, seed=0):
    tr = pkg.workloads.synthetic_track("putnam" if kind == "iac" else "barc")
    kw = {}
    if kind == "lmpc":
        cfg = dict(pkg.presets.barc_lmpc(N, 5))
        laps = pkg.workloads.synthetic_laps(tr, 5)
        x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=seed)
        veh = pkg.presets.barc_vehicle()
    elif kind == "iac":
        cfg = dict(pkg.presets.iac_tracking_mpc(N))
        veh = pkg.presets.iac_vehicle()
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=seed + 1)
    else:
        cfg = dict(pkg.presets.barc_tracking_mpc(N))
        veh = pkg.presets.barc_vehicle()
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=seed)
    sv = pkg.Solver(cfg, veh, device=0)
    sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    if kind == "lmpc":
        sv.set_safe_set(laps, tr["L"])
        if regression:
            pv = dict(veh)
            pv["mu"] *= 0.85
            plant = pkg.Solver(cfg, pv, device=0)
            reg = pkg.workloads.regression_sample_pairs(
                tr, laps, lambda xa, ua: plant.plant_step(tr, torch.as_tensor(xa.T.copy(), device=dev), torch.as_tensor(ua.T.copy(), device=dev),
                                                          0.03).cpu().numpy().T)
            plant.close()
            sv.set_regression_laps(reg, dist_max=0.6)
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = (s0 - s_last).abs() + L / 2
        q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
        ss_x, ss_j, _ = sv.ss_query(q)
        kw = dict(ss_x=ss_x, ss_j=ss_j)
    ref = None
    for prec in precisions:
        out = sv.alloc_outputs(B)
        if kind == "lmpc":
            out["convex_combi_optm"] = torch.zeros((int(cfg["num_ss_pts"]), B), dtype=torch.float64, device=dev)
        if prec == "f32":
            inp32 = {k: (v.float() if hasattr(v, "float") and v.dtype == torch.float64 else v) for k, v in inp.items()}
            o, ms, mn, lin = timed(sv, lambda: sv.solve_f32(inp32))
        else:
            o, ms, mn, lin = timed(sv, lambda: sv.solve(inp, out, mixed=(prec == "mixed"), **kw))
        st = np.bincount(o["status"].cpu().numpy(), minlength=4)
        it = o["iters"].cpu().numpy()
        row = {"lib": LIB, "case": f"{kind}{N}", "B": B, "prec": prec, "reg": regression, "qp_ms": round(ms, 4), "qp_ms_min": round(mn, 4),
               "lin_ms": round(lin, 4), "Msolves_s": round(B / ms / 1e3, 3), "status": st.tolist(), "iters_mean": round(float(it.mean()), 3),
               "iters_max": int(it.max()), "sha": digest(o)}
        if prec == "f64":
            ref = {k: o[k].clone() for k in ("X_optm", "U_optm", "status")}
        elif ref is not None:
            s64, sx_ = (ref["status"] == 0).cpu().numpy(), (o["status"] == 0).cpu().numpy()
            e = scaled_err(o, ref)
            both = s64 & sx_
            eb = e[both]
            bad = np.where(both & (e > 1e-3))[0]
            lost = np.where(s64 & ~sx_)[0]
            row.update({"err_med": float(np.median(eb)), "err_999": float(np.quantile(eb, .999)), "err_max": float(eb.max()),
                        "n_gt_1e3": int((eb > 1e-3).sum()), "n_gt_5e4": int((eb > 5e-4).sum()), "n_gt_3e4": int((eb > 3e-4).sum()),
                        "bad_idx": bad[:8].tolist(), "bad_err": [float(v) for v in e[bad[:8]]],
                        "lost_idx": lost[:8].tolist(), "lost_status": o["status"].cpu().numpy()[lost[:8]].tolist(),
                        "lost_iters": it[lost[:8]].tolist(),
                        "lost_kkt": o["kkt"].cpu().numpy()[:, lost[:8]].T.tolist() if "kkt" in o else None})
        print(json.dumps(row), flush=True)
    sv.close()


CASES = {
    "trk20": lambda: case("barc", 20, 4096),
    "trk20big": lambda: case("barc", 20, 65536),
    "trk10": lambda: case("barc", 10, 4096),
    "trk40": lambda: case("barc", 40, 4096),
    "trk60": lambda: case("barc", 60, 4096),
    "trk80": lambda: case("barc", 80, 4096),
    "lmpc": lambda: case("lmpc", 20, 4096, ("f64", "mixed")),
    "lmpc32k": lambda: case("lmpc", 20, 32768, ("f64", "mixed")),
    "lmpc32kreg": lambda: case("lmpc", 20, 32768, ("f64", "mixed"), regression=True),
    "lmpc40": lambda: case("lmpc", 40, 4096, ("f64",)),
    "lmpc60": lambda: case("lmpc", 60, 4096, ("f64",)),
    "iac": lambda: case("iac", 40, 8192, ("f64", "mixed", "f32")),
    "iac60": lambda: case("iac", 60, 4096, ("f64", "mixed", "f32")),
    "iac80": lambda: case("iac", 80, 4096, ("f64", "mixed", "f32")),
    "trk60m": lambda: case("barc", 60, 4096, ("f64", "mixed")),
    "lmpc80": lambda: case("lmpc", 80, 2048, ("f64",)),
    "lmpc96": lambda: case96(),
}


def case96():
    """the learning problem with 96 points (KS = 2), N = 20 and 40, fp64 and mixed"""
    global pkg
    orig = pkg.presets.barc_lmpc
    pkg.presets.barc_lmpc = lambda N, n=5: orig(N, 3)
    syn = pkg.workloads.synthetic_laps
    pkg.workloads.synthetic_laps = lambda tr, n=5, **kw: syn(tr, 3, **kw)
    try:
        case("lmpc", 20, 4096, ("f64", "mixed"))
        case("lmpc", 40, 2048, ("f64",))
    finally:
        pkg.presets.barc_lmpc, pkg.workloads.synthetic_laps = orig, syn


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["trk20", "lmpc", "iac"]):
        try:
            CASES[name]()
        except Exception as ex:  # a case a build does not serve is reported, the rest still run
            print(json.dumps({"lib": LIB, "case": name, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}), flush=True)
