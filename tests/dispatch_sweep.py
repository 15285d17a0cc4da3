"""Every kernel instantiation the dispatch table can select, against the serial twin.  Helper of
tests/test_gpu_dispatch.py: run as a subprocess so that LMPC_HIP_LIBRARY can name the build under test (the product library and
the debug-hook build).

The library picks a kernel from (N, num_ss_pts, precision): `real` in {double, float}, KQ = slots per lane in {2, 4, 7, 11, 14}
(N <= 11 / 23 / 40 / 64 / 81), KS = safe-set points per lane in {0, 2, 3} (none / <= 128 / <= 192 points), `io` in {double,
float}; lean or fat LDS records, polish inlined or called, DPP or LDS sweeps follow from those (csrc/lmpc_capi.hip `kq_for`,
`ks_for`, `pick_*_fn`; lmpc_query_launch_for reports what an entry point would launch).  Rounds 2 - 4 each met an instantiation
that miscomputed under a source change that should not matter ("the register-starved instantiations"): a
compiler-sensitive corner that only an every-problem check of EVERY instantiation notices.  So: every N from 3 to 81 -- not a
hand-picked list: slots wrap round the lanes differently at every N -- x {BARC tracking, IAC tracking, learning with 96 points
(KS = 2), learning with 160 points (KS = 3)} in fp64 against the twin on `--problems` problems (statuses equal, X / U / dU within
1e-6 scaled, iteration counts equal on >= 90 %), and wherever lmpc_query_launch_for says the entry point has a kernel, the fp32
entry and the mixed entry against the fp64 kernel's answers (tests/tolerances.py TOL_F32_SWEEP; every problem the fp64 kernel
solves is solved).
Prints one line per (family, N) and a JSON summary; exit code 1 on any violation."""
import argparse
import json
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from __graft_entry__ import load_package  # noqa: E402
from oracle import cbind, params as P  # noqa: E402
from tolerances import TOL_DU, TOL_F32_SWEEP, TOL_TWIN  # noqa: E402

pkg = load_package()
dev = torch.device("cuda:0")
SX, SU = P.SCALE_X[:, None, None], P.SCALE_U[:, None, None]
POOL = ThreadPoolExecutor(16)
DUMP = ""
SEED = 0   # --seed: offset of the sampling seeds (0: the problems the test suite runs on)


def twin_parallel(cfg, veh, inp, ss_x=None, ss_j=None, chunks=16):
    """cbind.solve_batch over `chunks` ranges on a thread pool (ctypes releases the GIL; the C entry writes its own range only)."""
    B = inp["x_ic"].shape[-1]
    edges = np.linspace(0, B, chunks + 1).astype(int)
    parts = list(POOL.map(lambda r: cbind.solve_batch(cfg, veh, inp, ss_x, ss_j, b0=int(r[0]), b1=int(r[1])), zip(edges[:-1], edges[1:])))
    out = parts[0]
    for (a, b), p in zip(zip(edges[1:-1], edges[2:]), parts[1:]):
        for k in ("X_optm", "U_optm", "dU_optm", "status", "iters"):
            out[k][..., a:b] = p[k][..., a:b]
    return out


def npd(d):
    return {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in d.items()}


def err(a, b, ok):
    ex = np.abs((a["X_optm"].astype(np.float64) - b["X_optm"]) / SX).max(axis=(0, 1))[ok]
    eu = np.abs((a["U_optm"].astype(np.float64) - b["U_optm"]) / SU).max(axis=(0, 1))[ok]
    ed = np.abs((a["dU_optm"].astype(np.float64) - b["dU_optm"]) / SU).max(axis=(0, 1))[ok]
    return (float(max(ex.max(), eu.max())), float(ed.max())) if ok.any() else (0.0, 0.0)


def has_kernel(sv, precision):
    import ctypes as C

    lds, ppc = C.c_int32(0), C.c_int32(0)
    return sv.lib.lmpc_query_launch_for(sv._h, C.c_int32(precision), C.byref(lds), C.byref(ppc)) == 0, lds.value, ppc.value


def one(family, N, B, failures):
    iac, learning = family == "iac", family.startswith("lrn")
    tr = pkg.workloads.synthetic_track("putnam" if iac else "barc")
    if iac:
        x, u = pkg.workloads.sample_initial_states("putnam", B, tr["L"], [-10.0, -0.314159], [5.0, 0.314159], seed=1 + SEED)
        pc, pv, oc, ov = pkg.presets.iac_tracking_mpc(N), pkg.presets.iac_vehicle(), P.iac_tracking_mpc(N), P.iac_vehicle()
    elif learning:
        n_laps = 3 if family == "lrn96" else 5
        pc, pv, oc, ov = dict(pkg.presets.barc_lmpc(N, n_laps)), pkg.presets.barc_vehicle(), P.barc_lmpc(N, n_laps), P.barc_vehicle()
        laps = pkg.workloads.synthetic_laps(tr, n_laps)
        x, u = pkg.workloads.sample_states_near_laps(laps, B, tr["L"], seed=SEED)
    else:
        x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.01, -0.314159], [0.01, 0.314159], seed=SEED)
        pc, pv, oc, ov = pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), P.barc_tracking_mpc(N), P.barc_vehicle()
    sv = pkg.Solver(pc, pv, device=0)
    sv.reserve(B)
    inp = sv.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
    ss_x = ss_j = None
    if learning:
        sv.set_safe_set(laps, tr["L"])
        s_last, s0, L = inp["X_ref"][0, -1], inp["x_ic"][0], tr["L"]
        kk = (s0 - s_last).abs() + L / 2
        q = torch.stack([s_last + (kk - torch.fmod(kk, L)) * torch.sign(s0 - s_last), inp["X_ref"][1, -1]]).contiguous()
        ss_x, ss_j, _ = sv.ss_query(q)

    def solve(**kw):
        out = sv.alloc_outputs(B)
        if learning:
            out["convex_combi_optm"] = torch.zeros((int(pc["num_ss_pts"]), B), dtype=torch.float64, device=dev)
        return npd(sv.solve(inp, out, ss_x=ss_x, ss_j=ss_j, **kw))

    o = solve()
    tw = twin_parallel(oc, ov, npd(inp), None if ss_x is None else ss_x.cpu().numpy(), None if ss_j is None else ss_j.cpu().numpy())
    ok = (o["status"] == 0) & (tw["status"] == 0)
    exu, ed = err(o, tw, ok)
    di = np.abs(o["iters"][ok] - tw["iters"][ok])
    rec = {"family": family, "N": N, "solved": int(ok.sum()), "status_equal": bool((o["status"] == tw["status"]).all()), "xu": exu, "du": ed,
           "iters_equal": float((di == 0).mean()) if ok.any() else 1.0, "iters_mean": float(o["iters"][ok].mean()) if ok.any() else 0.0,
           "lds": has_kernel(sv, 0)[1], "per_cu": has_kernel(sv, 0)[2]}
    bad = []
    # Statuses: equal, except that a borderline polish acceptance may fall either way (a held row met to 0.9e-9 or 1.1e-9 decides
    # between "polished: OPTIMAL" and "refused by step noise: MAX_ITER", include/lmpc_hip.h) on at most two problems per case;
    # INFEASIBLE against anything else is never borderline.
    differ = np.nonzero(o["status"] != tw["status"])[0]
    rec["status_differ"] = int(differ.size)
    if differ.size > 2 or ((o["status"][differ] == 2) | (tw["status"][differ] == 2)).any():
        bad.append("statuses differ from the twin's at %s: %s vs %s" % (differ[:6].tolist(), o["status"][differ[:6]].tolist(), tw["status"][differ[:6]].tolist()))
    if not (exu < TOL_TWIN and ed < TOL_DU):
        bad.append("fp64 vs twin %.1e / %.1e" % (exu, ed))
    if ok.any() and not (rec["iters_equal"] >= 0.9 and di.max() <= 8):
        bad.append("iteration counts: equal %.3f, max difference %d" % (rec["iters_equal"], di.max()))
    if ok.mean() < (0.95 if N >= 6 else 0.5):
        bad.append("solved fraction %.3f" % ok.mean())
    s64 = o["status"] == 0
    for name, prec in (("mixed", 2), ("f32", 1)):
        if (name == "f32" and learning) or family == "trk" or not has_kernel(sv, prec)[0]:
            continue  # (BARC tracking is not a single-precision problem: DESIGN.md section 3, "What fp32 cannot do here")
        if name == "mixed":
            r = solve(mixed=True)
        else:
            r = npd(sv.solve_f32({k: (v.to(torch.float32).contiguous() if hasattr(v, "to") else v) for k, v in inp.items()}))
        sr = r["status"] == 0
        both = s64 & sr
        e, e_du = err(r, o, both)
        rec[name] = {"xu": e, "du": e_du, "lost": int((s64 & ~sr).sum()), "unverified": int((r["status"] == 3).sum())}
        if rec[name]["lost"] or rec[name]["unverified"] or not (e < TOL_F32_SWEEP and e_du < TOL_F32_SWEEP / 0.025):
            bad.append("%s vs fp64: %s" % (name, rec[name]))
    sv.close()
    print("%-6s N = %2d: solved %4d/%d  fp64 vs twin %.1e / %.1e  iterations equal %.3f (mean %.2f)  LDS %6d B, %d per CU%s%s%s"
          % (family, N, rec["solved"], B, exu, ed, rec["iters_equal"], rec["iters_mean"], rec["lds"], rec["per_cu"],
             "  mixed %.1e" % rec["mixed"]["xu"] if "mixed" in rec else "", "  f32 %.1e" % rec["f32"]["xu"] if "f32" in rec else "",
             "   <-- " + "; ".join(bad) if bad else ""), flush=True)
    if bad:
        failures.append({"family": family, "N": N, "what": bad})
        if DUMP:  # the worst problems of a failing case with everything needed to re-solve them on the CPU (dense oracle, twin)
            e = np.maximum(np.maximum(np.abs((o["X_optm"] - tw["X_optm"]) / SX).max(axis=(0, 1)), np.abs((o["U_optm"] - tw["U_optm"]) / SU).max(axis=(0, 1))),
                           np.abs((o["dU_optm"] - tw["dU_optm"]) / SU).max(axis=(0, 1)))
            e[~ok] = -1.0
            idx = np.argsort(e)[-8:]
            ninp = npd(inp)
            np.savez(Path(DUMP) / ("%s_N%d.npz" % (family, N)), idx=idx, err=e[idx], kernel_iters=o["iters"][idx], twin_iters=tw["iters"][idx],
                     kernel_kkt=o["kkt"][:, idx], twin_kkt=tw["kkt"][:, idx],
                     **{"k_" + k: o[k][..., idx] for k in ("X_optm", "U_optm", "dU_optm")}, **{"t_" + k: tw[k][..., idx] for k in ("X_optm", "U_optm", "dU_optm")},
                     **{"in_" + k: np.asarray(ninp[k])[..., idx] for k in ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")},
                     L=float(tr["L"]), **({} if ss_x is None else {"ss_x": ss_x.cpu().numpy()[..., idx], "ss_j": ss_j.cpu().numpy()[..., idx]}))
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", type=int, default=1024)
    ap.add_argument("--families", default="trk,iac,lrn96,lrn160")
    ap.add_argument("--nmin", type=int, default=3)
    ap.add_argument("--nmax", type=int, default=81)
    ap.add_argument("--nstep", type=int, default=1)
    ap.add_argument("--dump", default="")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    DUMP = a.dump
    SEED = a.seed
    if DUMP:
        Path(DUMP).mkdir(parents=True, exist_ok=True)
    t0 = time.time()
    failures, recs = [], []
    for family in a.families.split(","):
        for N in range(a.nmin, a.nmax + 1, a.nstep):
            recs.append(one(family, N, a.problems, failures))
    worst = {f: max((r["xu"], r["du"], r["N"]) for r in recs if r["family"] == f) for f in a.families.split(",")}
    print(json.dumps({"library": str(pkg.library_path().name), "cases": len(recs), "problems": a.problems,
                      "failures": failures, "worst_fp64_vs_twin": {f: "%.1e / %.1e at N = %d" % w for f, w in worst.items()},
                      "worst_mixed": max([r["mixed"]["xu"] for r in recs if "mixed" in r] or [0.0]),
                      "worst_f32": max([r["f32"]["xu"] for r in recs if "f32" in r] or [0.0]), "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if failures else 0)
