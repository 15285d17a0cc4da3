"""ctypes access to oracle/_build/liblmpc_oracle.so (test infrastructure).

Host-pointer twins of the C-ABI entry points, same [field][knot][batch] layout.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from .params import MPCConfig, Vehicle

_HERE = Path(__file__).resolve().parent
_LIB = None


class CVehicle(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("integrator", C.c_int32)] + [
        (n, C.c_double) for n in
        ("m Jzz l cg_ratio h b fr kd kb cd Af rho cl_f cl_r mu Bf Cf Br Cr Fd_max Fb_max Td Tb "
         "max_steer max_steer_rate").split()]


class CConfig(C.Structure):
    _fields_ = [("N", C.c_int32), ("learning", C.c_int32), ("num_ss_pts", C.c_int32),
                ("num_ss_pts_per_lap", C.c_int32), ("max_lap_stored", C.c_int32),
                ("max_iter", C.c_int32), ("polish", C.c_int32), ("reserved", C.c_int32),
                ("tol", C.c_double), ("margin", C.c_double),
                ("q_contour", C.c_double), ("q_heading", C.c_double), ("q_vel", C.c_double),
                ("q_vy", C.c_double), ("q_vyaw", C.c_double), ("q_boundary", C.c_double),
                ("R", C.c_double * 4), ("R_d", C.c_double * 4),
                ("x_max", C.c_double * 6), ("x_min", C.c_double * 6),
                ("u_max", C.c_double * 2), ("u_min", C.c_double * 2),
                ("convex_hull_slack", C.c_double * 6), ("max_vel_ref_diff", C.c_double)]


def c_vehicle(v: Vehicle) -> CVehicle:
    cv = CVehicle()
    cv.model_id = 0
    cv.integrator = 1 if getattr(v, "integrator", "rk4") == "euler" else 0
    for name, _ in CVehicle._fields_[2:]:
        setattr(cv, name, float(getattr(v, name)))
    return cv


def c_config(cfg: MPCConfig, max_iter: int = 0, tol: float = 0.0, polish: int = 0) -> CConfig:
    cc = CConfig()
    cc.N, cc.learning = cfg.N, int(cfg.learning)
    cc.num_ss_pts, cc.num_ss_pts_per_lap, cc.max_lap_stored = cfg.num_ss_pts, cfg.num_ss_pts_per_lap, cfg.max_lap_stored
    cc.max_iter, cc.tol, cc.polish = max_iter, tol, polish
    for n in ("margin", "q_contour", "q_heading", "q_vel", "q_vy", "q_vyaw", "q_boundary", "max_vel_ref_diff"):
        setattr(cc, n, float(getattr(cfg, n)))
    cc.R[:] = list(np.asarray(cfg.R, dtype=float).reshape(-1))
    cc.R_d[:] = list(np.asarray(cfg.R_d, dtype=float).reshape(-1))
    cc.x_max[:] = list(cfg.x_max)
    cc.x_min[:] = list(cfg.x_min)
    cc.u_max[:] = list(cfg.u_max)
    cc.u_min[:] = list(cfg.u_min)
    cc.convex_hull_slack[:] = list(cfg.convex_hull_slack)
    return cc


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "_build" / "liblmpc_oracle.so"
        if not so.exists():
            subprocess.check_call(["make", "-C", str(_HERE)], stdout=subprocess.DEVNULL)
        _LIB = C.CDLL(str(so))
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def solve_batch(cfg: MPCConfig, veh: Vehicle, inp: dict, ss_x=None, ss_j=None, b0=0, b1=None,
                max_iter: int = 0, tol: float = 0.0, polish: int = 0, warm: bool = False, warm_plan: dict | None = None, warm_rounds: int = 0) -> dict:
    """inp as produced by oracle.scenario.cold_start_inputs (batch axis last).  warm: the active-set warm start of
    lmpc_solve_batch_warm; the plan is warm_plan's (X_ref, U_ref) when given, else the linearisation trajectory itself.  The
    learning problem's warm start (lmpc_solve_batch_warm_ss) also takes warm_plan["lam"] [S][B], the plan's simplex weights."""
    N = cfg.N
    B = inp["x_ic"].shape[-1]
    b1 = B if b1 is None else b1
    arrs = [_c(inp[k]) for k in ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right",
                                 "curvatures", "vel_ref")]
    ss_x, ss_j = _c(ss_x), _c(ss_j)
    X = np.zeros((6, N, B))
    U = np.zeros((2, N - 1, B))
    dU = np.zeros((2, N - 1, B))
    lam = np.zeros((cfg.num_ss_pts, B)) if cfg.learning else None
    status = np.full(B, -1, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    kkt = np.zeros((4, B))
    cc, cv = c_config(cfg, max_iter, tol, polish), c_vehicle(veh)
    plan = []
    lib().lmpc_oracle_set_warm_rounds(C.c_int(warm_rounds))  # (lmpc_set_warm_rounds; 0: the default)
    fn = lib().lmpc_oracle_solve_range_warm if warm else lib().lmpc_oracle_solve_range
    if warm and warm_plan is not None:
        fn = lib().lmpc_oracle_solve_range_warm_plan
        plan = [_c(warm_plan["X_ref"]), _c(warm_plan["U_ref"])]
        assert plan[0].shape == (6, N, B) and plan[1].shape == (2, N - 1, B)
        if warm_plan.get("lam") is not None:
            fn = lib().lmpc_oracle_solve_range_warm_lam
            plan.append(_c(warm_plan["lam"]))
            assert plan[2].shape == (cfg.num_ss_pts, B)
    rc = fn(C.byref(cc), C.byref(cv), C.c_int32(B), C.c_int32(b0), C.c_int32(b1),
            *[_p(a) for a in arrs], _p(ss_x), _p(ss_j), *[_p(a) for a in plan], _p(X), _p(U), _p(dU),
            _p(lam), _p(status), _p(iters), _p(kkt))
    if rc != 0:
        raise RuntimeError(f"lmpc_oracle_solve_range -> {rc}")
    return {"X_optm": X, "U_optm": U, "dU_optm": dU, "convex_combi_optm": lam, "status": status,
            "iters": iters, "kkt": kkt}


def linearize_batch(cfg: MPCConfig, veh: Vehicle, inp: dict):
    N = cfg.N
    B = inp["X_ref"].shape[-1]
    A = np.zeros((6, 6, N - 1, B))
    Bm = np.zeros((6, 2, N - 1, B))
    g = np.zeros((6, N - 1, B))
    cc, cv = c_config(cfg), c_vehicle(veh)
    arrs = [_c(inp[k]) for k in ("X_ref", "U_ref", "T_ref", "curvatures")]
    rc = lib().lmpc_oracle_linearize_batch(C.byref(cc), C.byref(cv), C.c_int32(B), *[_p(a) for a in arrs],
                                           _p(A), _p(Bm), _p(g))
    if rc != 0:
        raise RuntimeError(f"lmpc_oracle_linearize_batch -> {rc}")
    return A, Bm, g


def ss_query_batch(laps_x, L: float, S: int, K: int, query: np.ndarray):
    """laps_x: list of (n_j, 6) arrays oldest first; query (2, B)."""
    n_pts = np.array([a.shape[0] for a in laps_x], dtype=np.int32)
    x = (np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64) for a in laps_x], axis=0))
         if len(laps_x) else np.zeros((1, 6)))
    if n_pts.size == 0:
        n_pts = np.zeros(1, dtype=np.int32)
    query = _c(query)
    B = query.shape[1]
    ss_x = np.zeros((6, S, B))
    ss_j = np.zeros((S, B))
    nf = np.zeros(B, dtype=np.int32)
    f = lib().lmpc_oracle_ss_query_batch
    rc = f(C.c_int32(len(laps_x)), _p(n_pts), _p(x), C.c_double(L), C.c_int32(S), C.c_int32(K),
           C.c_int32(B), _p(query), _p(ss_x), _p(ss_j), _p(nf))
    if rc != 0:
        raise RuntimeError(f"lmpc_oracle_ss_query_batch -> {rc}")
    return ss_x, ss_j, nf
