"""The twin with other early-polish thresholds (-DPOLISH_MU=...), on the dense fixtures' problems: iterations and accuracy.
usage: twin_mu_early.py barc_tracking_n20,iac_tracking_n40,...  1e-8 1e-7 1e-6"""
import ctypes, os, subprocess, sys, hashlib
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import dense_cases as DC
from oracle import cbind, params as P
from parity import per_problem_err
class _NoGpu:  # dense_cases.build wants the package for workloads / presets only
    pass
from __graft_entry__ import load_package
pkg = load_package()
for mu in sys.argv[2:]:
    so = f"/tmp/twin_mu_{mu}.so"
    subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-std=c11", f"-I{ROOT}/include", "-shared", "-o", so, str(ROOT / "oracle/c/lmpc_oracle.c"), "-lm", f"-D{os.environ.get('TWIN_MACRO', 'POLISH_MU')}={mu}"])
    cbind._LIB = None
    cbind._lib = None
    lib = ctypes.CDLL(so)
    for name in sys.argv[1].split(","):
        d = np.load(ROOT / "tests" / "golden" / f"dense_{name}.npz")
        cfg, veh, inp, ss_x, ss_j = DC.build(pkg, name)
        old = cbind.lib
        cbind.lib = lambda: lib
        for f in ("lmpc_oracle_solve_range", "lmpc_oracle_set_warm_rounds"):
            pass
        try:
            tw = cbind.solve_batch(cfg, veh, inp, ss_x, ss_j)
        finally:
            cbind.lib = old
        ok = (tw["status"] == 0) & (d["status"] == 0)
        exu, ed = per_problem_err({k: tw[k][..., ok] for k in ("X_optm", "U_optm", "dU_optm")}, {k: d[k][..., ok] for k in ("X_optm", "U_optm", "dU_optm")})
        it = tw["iters"][ok]
        print(f"{os.environ.get('TWIN_MACRO', 'POLISH_MU')} {mu} {name}: solved {ok.sum()} of {ok.size}; iters mean {it.mean():.3f} max {it.max()}; vs dense X/U max {exu.max():.1e} dU max {ed.max():.1e}", flush=True)
