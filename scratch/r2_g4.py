"""Grouped kernel (LMPC_GROUPED=1: four problems per workgroup, merged Riccati sweeps) against the one-wave-per-problem
kernel on the bench workload: bitwise comparison and kernel time.  usage: python scratch/r2_g4.py [batch]"""
import sys, os, subprocess, numpy as np
if len(sys.argv) > 2 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, "/root/repo")
    from __graft_entry__ import load_package
    pkg = load_package()
    import importlib
    capi = importlib.import_module(pkg.__name__ + ".capi")
    _orig = capi.library_path
    capi.library_path = lambda: _orig().with_name(os.environ.get("LMPC_LIB", "liblmpc_hip.so"))
    B = int(sys.argv[3])
    NH = int(os.environ.get("LMPC_N", "20"))
    tr = pkg.workloads.synthetic_track("barc")
    solver = pkg.Solver(pkg.presets.barc_tracking_mpc(NH), pkg.presets.barc_vehicle(), device=0)
    P = solver.config
    u_lo = [max(P["u_min"][0], -0.015), max(P["u_min"][1], -0.314159)]
    u_hi = [min(P["u_max"][0], 0.015), min(P["u_max"][1], 0.314159)]
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, seed=0)
    solver.reserve(B)
    inp = solver.prepare(tr, x.T.copy(), 0.025)
    inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device="cuda")
    out = solver.alloc_outputs(B)
    res = []
    for k in range(3):
        o = solver.solve(inp, out)
        torch.cuda.synchronize()
        res.append({k2: o[k2].clone().cpu().numpy() for k2 in ("X_optm", "U_optm", "dU_optm", "iters", "status")})
    rep = all((res[k][n] == res[0][n]).all() for k in range(1, 3) for n in res[0])
    solver.enable_timing(True)
    ms = []
    for _ in range(30):
        solver.solve(inp, out)
        torch.cuda.synchronize()
        ms.append(solver.last_kernel_ms()[1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    solver.enable_timing(False)
    e0.record()
    for _ in range(50):
        solver.solve(inp, out)
    e1.record(); torch.cuda.synchronize()
    print("LMPC_GROUPED=%s" % os.environ.get("LMPC_GROUPED"), "reproducible:", rep, "qp kernel ms %.4f (min %.4f)" % (np.mean(ms[5:]), np.min(ms)),
          "step ms %.4f -> %.3f M solves/s" % (e0.elapsed_time(e1) / 50, B / (e0.elapsed_time(e1) / 50) / 1e3),
          "solved", float((res[0]["status"] == 0).mean()), "mean iters", float(res[0]["iters"].mean()), "status", np.bincount(res[0]["status"]))
    np.savez(sys.argv[2], **res[0])
    sys.exit(0)
B = sys.argv[1] if len(sys.argv) > 1 else "4096"
outs = []
# default: grouped against ungrouped kernel of the current library; "libs": the base build against the current one
variants = [dict(LMPC_GROUPED="0"), dict(LMPC_GROUPED="1")] if len(sys.argv) < 3 else [dict(LMPC_LIB="liblmpc_hip_base.so"), dict(LMPC_LIB="liblmpc_hip.so")]
for k, v in enumerate(variants):
    o = "/tmp/g4_%d.npz" % k
    print(v)
    subprocess.run([sys.executable, __file__, "child", o, B], env=dict(os.environ, **v), check=True)
    outs.append(np.load(o))
for n in outs[0].files:
    a, b = outs[0][n], outs[1][n]
    print(n, "identical" if (a == b).all() else "DIFFERENT: %d entries, max |d| %.3g" % (int((a != b).sum()), float(np.abs(a.astype(float) - b).max())))
