"""Bitwise comparison of two builds of the library on the learning problem (+ run-to-run reproducibility and kernel time).
usage: python scratch/r2_bitcmp.py            -> runs itself once per library, compares
       LMPC_LIB=<name> python scratch/r2_bitcmp.py child <out.npz>"""
import sys, os, subprocess, numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
    from __graft_entry__ import load_package
    pkg = load_package()
    import importlib
    capi = importlib.import_module(pkg.__name__ + ".capi")
    _orig = capi.library_path
    capi.library_path = lambda: _orig().with_name(os.environ.get("LMPC_LIB", "liblmpc_hip.so"))
    import lmpc_scenario as LS
    B = 4096
    veh, cfg, tr, laps, inp, q = LS.make(B, 9)
    solver = pkg.Solver(pkg.presets.barc_lmpc(20, 3), pkg.presets.barc_vehicle(), device=0)
    solver.set_safe_set(laps, LS.L_BARC_SS)
    ss_x, ss_j, nf = solver.ss_query(q)
    res = []
    for k in range(4):
        out = solver.alloc_outputs(B)
        out["convex_combi_optm"] = torch.zeros((96, B), dtype=torch.float64, device="cuda")
        o = solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j)
        torch.cuda.synchronize()
        res.append({k2: o[k2].clone().cpu().numpy() for k2 in ("X_optm", "U_optm", "iters", "status", "convex_combi_optm")})
    rep = all((res[k][n] == res[0][n]).all() for k in range(1, 4) for n in res[0])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = solver.alloc_outputs(B)
    e0.record()
    for _ in range(20):
        solver.solve(inp, out, ss_x=ss_x, ss_j=ss_j)
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("LMPC_LIB"), "reproducible:", rep, "ms per 4096-batch: %.3f" % (e0.elapsed_time(e1) / 20),
          "solved", float((res[0]["status"] == 0).mean()), "mean iters", float(res[0]["iters"].mean()))
    np.savez(sys.argv[2], **res[0])
    sys.exit(0)
libs = sys.argv[1:] or ["liblmpc_hip_base.so", "liblmpc_hip.so"]
outs = []
for l in libs:
    o = "/tmp/bitcmp_%s.npz" % l
    subprocess.run([sys.executable, __file__, "child", o], env=dict(os.environ, LMPC_LIB=l), check=True)
    outs.append(np.load(o))
for n in outs[0].files:
    a, b = outs[0][n], outs[1][n]
    print(n, "identical" if (a == b).all() else "DIFFERENT: %d entries, max |d| %.3g" % (int((a != b).sum()), float(np.abs(a.astype(float) - b).max())))
