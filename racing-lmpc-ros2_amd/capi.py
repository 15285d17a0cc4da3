"""ctypes binding of include/lmpc_hip.h (lib/liblmpc_hip.so).

`Solver` mirrors the reference's RacingMPC surface for a whole batch: it is constructed from a
config and a vehicle (RacingMPC::RacingMPC, racing_mpc.cpp:31-35) and `solve()` takes/returns the
same keys as RacingMPC::solve's DMDict (racing_mpc.cpp:215-228, 347-353) with a trailing batch
axis.  Tensors are torch CUDA (ROCm) fp64 tensors; only their device pointers cross the ABI.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LIB = None

SOLVE_OPTIMAL, SOLVE_MAX_ITER, SOLVE_INFEASIBLE = 0, 1, 2

_ABI_SYMBOLS = ("lmpc_create", "lmpc_destroy", "lmpc_last_error", "lmpc_set_stream", "lmpc_synchronize",
                "lmpc_linearize_batch", "lmpc_solve_batch", "lmpc_set_safe_set", "lmpc_ss_query_batch",
                "lmpc_prepare_batch", "lmpc_reserve", "lmpc_query_launch", "lmpc_enable_timing",
                "lmpc_last_kernel_ms", "lmpc_query_residency", "lmpc_query_launch_for", "lmpc_solve_host", "lmpc_shift_batch",
                "lmpc_plant_step_batch", "lmpc_solve_full_dynamics_batch", "lmpc_solve_full_dynamics_host", "lmpc_set_regression_laps", "lmpc_regress_batch", "lmpc_ss_query_host", "lmpc_solve_batch_f32",
                "lmpc_solve_batch_mixed", "lmpc_prepare_failed_batch", "lmpc_set_launch_order", "lmpc_set_warm_rounds", "lmpc_loop_advance_batch", "lmpc_launch_order_from_iters", "lmpc_set_output_layout",
                "lmpc_ss_query_idx_batch", "lmpc_solve_batch_ss_idx", "lmpc_solve_batch_warm", "lmpc_solve_host_warm",
                "lmpc_last_solve_precision", "lmpc_solve_batch_warm_ss", "lmpc_solve_host_warm_ss", "lmpc_shift_lambda_batch",
                "lmpc_get_warm_accepted", "lmpc_set_waves_per_problem")


class LmpcError(RuntimeError):
    pass


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


class CRegressionSpec(C.Structure):
    """lmpc_regression_spec: RegQuery's index lists and bandwidth (safe_set.hpp:57-75)."""
    _fields_ = [("n_out", C.c_int32), ("out", C.c_int32 * 6), ("n_in_state", C.c_int32), ("in_state", C.c_int32 * 6),
                ("n_in_ctrl", C.c_int32), ("in_ctrl", C.c_int32 * 2), ("as_written", C.c_int32), ("dist_max", C.c_double)]


class CTrack(C.Structure):
    _fields_ = [("L", C.c_double), ("M", C.c_int32), ("reserved", C.c_int32),
                ("curvature", C.c_void_p), ("bound_left", C.c_void_p), ("bound_right", C.c_void_p),
                ("vel", C.c_void_p)]


def _fill(struct, values: dict):
    for name, ctype in struct._fields_:
        if name == "reserved":
            continue
        if name == "polish":         # 0 (default): on, < 0: off -- presets need not carry the key
            setattr(struct, name, int(values.get("polish", 0)))
            continue
        if name == "integrator":     # modeling.integrator_type, "rk4" (default) or "euler"
            iv = values.get("integrator", 0)
            setattr(struct, name, {"rk4": 0, "euler": 1}[iv] if isinstance(iv, str) else int(iv))
            continue
        v = values[name]
        if hasattr(ctype, "_length_"):
            getattr(struct, name)[:] = [float(a) for a in v]
        elif ctype is C.c_int32:
            setattr(struct, name, int(v))
        else:
            setattr(struct, name, float(v))
    return struct


def library_path() -> Path:
    """The built library; LMPC_HIP_LIBRARY names another build of it (A/B timing of a variant, scratch/)."""
    import os
    return Path(os.environ["LMPC_HIP_LIBRARY"]) if os.environ.get("LMPC_HIP_LIBRARY") else _HERE / "lib" / "liblmpc_hip.so"


def load_library():
    """Load liblmpc_hip.so.  Raises if it has not been built (no CPU fallback exists)."""
    global _LIB
    if _LIB is None:
        so = library_path()
        if not so.exists():
            raise LmpcError(f"{so} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            f"or `make -C {_HERE / 'csrc'}`")
        lib = C.CDLL(str(so))
        for sym in _ABI_SYMBOLS:
            getattr(lib, sym)
        lib.lmpc_last_error.restype = C.c_char_p
        lib.lmpc_last_error.argtypes = [C.c_void_p]
        lib.lmpc_destroy.restype = None
        lib.lmpc_destroy.argtypes = [C.c_void_p]
        _LIB = lib
    return _LIB


def _ptr(t):
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())


class Solver:
    """Batched RacingMPC on one GPU.  Not re-entrant; one instance per device."""

    def __init__(self, config: dict, vehicle: dict, device: int = 0):
        import torch

        self._torch = torch
        self.lib = load_library()
        self.config = dict(config)
        self.vehicle = dict(vehicle)
        self.N = int(config["N"])
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p(0)
        cc, cv = _fill(CConfig(), config), _fill(CVehicle(), vehicle)
        rc = self.lib.lmpc_create(C.byref(cc), C.byref(cv), C.c_int(device), C.byref(self._h))
        if rc != 0:
            msg = self.lib.lmpc_last_error(self._h).decode() if self._h else "allocation failed"
            if self._h:
                self.lib.lmpc_destroy(self._h)
                self._h = C.c_void_p(0)
            raise LmpcError(f"lmpc_create -> {rc}: {msg}")
        self._track_keepalive = None

    def close(self):
        if getattr(self, "_h", None):
            self.lib.lmpc_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise LmpcError(f"{what} -> {rc}: {self.lib.lmpc_last_error(self._h).decode()}")

    def _t(self, x):
        torch = self._torch
        t = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        return t.contiguous()

    def use_current_stream(self):
        """Queue launches on torch's current stream, so that torch copies, ops and events order
        with the kernels.  Called by every method below before it launches."""
        s = self._torch.cuda.current_stream(self.device).cuda_stream
        if s != getattr(self, "_stream", None):
            self._check(self.lib.lmpc_set_stream(self._h, C.c_void_p(s)), "lmpc_set_stream")
            self._stream = s

    def synchronize(self):
        self._check(self.lib.lmpc_synchronize(self._h), "lmpc_synchronize")

    def reserve(self, max_batch: int):
        self._check(self.lib.lmpc_reserve(self._h, C.c_int32(max_batch)), "lmpc_reserve")

    def enable_timing(self, on: bool = True):
        self._check(self.lib.lmpc_enable_timing(self._h, C.c_int32(int(on))), "lmpc_enable_timing")

    def last_kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        self._check(self.lib.lmpc_last_kernel_ms(self._h, C.byref(a), C.byref(b)), "lmpc_last_kernel_ms")
        return a.value, b.value

    def launch_info(self, precision: str = "f64"):
        """LDS bytes and resident problems per CU of the kernel the given entry point launches ("f64", "f32", "mixed")."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.lmpc_query_launch(self._h, C.byref(a), C.byref(b)), "lmpc_query_launch")
        c = C.c_int32(0)
        self._check(self.lib.lmpc_query_launch_for(self._h, C.c_int32({"f64": 0, "f32": 1, "mixed": 2}[precision]), C.byref(a), C.byref(c)),
                    "lmpc_query_launch_for")
        return {"lds_bytes_per_problem": a.value, "threads_per_problem": b.value, "resident_problems_per_cu": c.value}

    # ---- input preparation (racing_mpc_node.cpp:210-235,261-292) ----
    def prepare(self, track: dict, x_ic, dt: float, speed_scale: float = 1.0, speed_limit: float | None = None):
        torch = self._torch
        self.use_current_stream()
        x_ic = self._t(x_ic)
        B, N = x_ic.shape[1], self.N
        tabs = {k: self._t(track[k]) for k in ("curvature", "bound_left", "bound_right", "vel")}
        ct = CTrack(float(track["L"]), int(tabs["curvature"].numel()), 0, tabs["curvature"].data_ptr(),
                    tabs["bound_left"].data_ptr(), tabs["bound_right"].data_ptr(), tabs["vel"].data_ptr())
        if speed_limit is None:
            speed_limit = float(self.config["x_max"][3])
        kw = dict(dtype=torch.float64, device=self.device)
        out = {"X_ref": torch.empty((6, N, B), **kw), "U_ref": torch.empty((2, N - 1, B), **kw),
               "T_ref": torch.empty((N - 1, B), **kw), "bound_left": torch.empty((N, B), **kw),
               "bound_right": torch.empty((N, B), **kw), "curvatures": torch.empty((N, B), **kw),
               "vel_ref": torch.empty((N, B), **kw)}
        rc = self.lib.lmpc_prepare_batch(self._h, C.c_int32(B), C.byref(ct), _ptr(x_ic), C.c_double(dt),
                                         C.c_double(speed_scale), C.c_double(speed_limit),
                                         *[_ptr(out[k]) for k in ("X_ref", "U_ref", "T_ref", "bound_left",
                                                                  "bound_right", "curvatures", "vel_ref")])
        self._check(rc, "lmpc_prepare_batch")
        self._track_keepalive = tabs
        out["x_ic"] = x_ic
        out["L"] = float(track["L"])
        return out

    def prepare_failed(self, track: dict, x_ic, status, inp: dict, dt: float, speed_scale: float = 1.0,
                       speed_limit: float | None = None):
        """lmpc_prepare_failed_batch: cold start, in place in `inp`, of the problems with status != 0."""
        self.use_current_stream()
        ct = self._ctrack(track)
        if speed_limit is None:
            speed_limit = float(self.config["x_max"][3])
        x_ic = self._t(x_ic)
        rc = self.lib.lmpc_prepare_failed_batch(self._h, C.c_int32(x_ic.shape[1]), C.byref(ct), _ptr(x_ic), _ptr(status),
                                                C.c_double(dt), C.c_double(speed_scale), C.c_double(speed_limit),
                                                *[_ptr(inp[k]) for k in ("X_ref", "U_ref", "T_ref", "bound_left",
                                                                         "bound_right", "curvatures", "vel_ref")])
        self._check(rc, "lmpc_prepare_failed_batch")
        return inp

    def _ctrack(self, track: dict):
        tabs = {k: self._t(track[k]) for k in ("curvature", "bound_left", "bound_right", "vel")}
        ct = CTrack(float(track["L"]), int(tabs["curvature"].numel()), 0, tabs["curvature"].data_ptr(),
                    tabs["bound_left"].data_ptr(), tabs["bound_right"].data_ptr(), tabs["vel"].data_ptr())
        self._track_keepalive = tabs
        return ct

    def device_track(self, track: dict) -> dict:
        """Upload the track tables once; the returned dict can be passed wherever `track` is expected."""
        out = {k: self._t(track[k]) for k in ("curvature", "bound_left", "bound_right", "vel")}
        out["L"] = float(track["L"])
        out["M"] = int(track["M"])
        return out

    # ---- warm-start shift (racing_mpc_node.cpp:245-254,261-292) ----
    def shift(self, track: dict, prev_inp: dict, sol: dict, dt: float, speed_scale: float = 1.0,
              speed_limit: float | None = None):
        torch = self._torch
        self.use_current_stream()
        ct = self._ctrack(track)
        N = self.N
        B = sol["X_optm"].shape[2]
        if speed_limit is None:
            speed_limit = float(self.config["x_max"][3])
        kw = dict(dtype=torch.float64, device=self.device)
        out = {"X_ref": torch.empty((6, N, B), **kw), "U_ref": torch.empty((2, N - 1, B), **kw),
               "T_ref": torch.empty((N - 1, B), **kw), "bound_left": torch.empty((N, B), **kw),
               "bound_right": torch.empty((N, B), **kw), "curvatures": torch.empty((N, B), **kw),
               "vel_ref": torch.empty((N, B), **kw), "L": float(track["L"])}
        rc = self.lib.lmpc_shift_batch(self._h, C.c_int32(B), C.byref(ct), _ptr(sol["X_optm"]), _ptr(sol["U_optm"]),
                                       _ptr(prev_inp["X_ref"]), _ptr(prev_inp["U_ref"]), _ptr(sol["status"]),
                                       C.c_double(dt), C.c_double(speed_scale), C.c_double(speed_limit),
                                       *[_ptr(out[k]) for k in ("X_ref", "U_ref", "T_ref", "bound_left",
                                                                "bound_right", "curvatures", "vel_ref")])
        self._check(rc, "lmpc_shift_batch")
        return out

    def set_output_layout(self, layout: str = "soa"):
        """"soa": results [component][knot][batch] (default); "aos": [batch][knot][component] (include/lmpc_hip.h).  The caller
        passes result tensors of the matching shape (alloc_outputs follows the setting)."""
        self._aos = {"soa": False, "aos": True}[layout]
        self._check(self.lib.lmpc_set_output_layout(self._h, C.c_int32(int(self._aos))), "lmpc_set_output_layout")

    def set_warm_rounds(self, rounds: int = 0):
        """Repair rounds a warm start may spend before the cold start takes over (include/lmpc_hip.h): 1 .. 4, or 0 for the default
        (2 rounds; 4 when the batch is at least four times what the device holds at once)."""
        self._check(self.lib.lmpc_set_warm_rounds(self._h, C.c_int32(int(rounds))), "lmpc_set_warm_rounds")

    # ---- launch order (longest job first; include/lmpc_hip.h) ----
    def set_launch_order(self, order):
        """order: int32 device tensor [batch] (kept alive here; applies to solves of that batch size) or None for the default mapping."""
        self._launch_order = None if order is None else self._torch.as_tensor(order, dtype=self._torch.int32, device=self.device).contiguous()
        n = 0 if self._launch_order is None else int(self._launch_order.numel())
        self._check(self.lib.lmpc_set_launch_order(self._h, _ptr(self._launch_order), C.c_int32(n)), "lmpc_set_launch_order")

    def launch_order_from_iters(self, iters, order=None):
        """Fills (and returns) `order` from the iteration counts of the previous solve of the same batch, longest first."""
        torch = self._torch
        self.use_current_stream()
        iters = torch.as_tensor(iters, dtype=torch.int32, device=self.device).contiguous()
        if order is None:
            order = torch.empty_like(iters)
        self._check(self.lib.lmpc_launch_order_from_iters(self._h, C.c_int32(iters.numel()), _ptr(iters), _ptr(order)),
                    "lmpc_launch_order_from_iters")
        return order

    # ---- plant (racing_simulator.cpp:97-112) ----
    def plant_step(self, track: dict, x, u, dt_sim: float, n_sub: int = 1):
        self.use_current_stream()
        ct = self._ctrack(track)
        rc = self.lib.lmpc_plant_step_batch(self._h, C.c_int32(x.shape[1]), C.byref(ct), _ptr(x), _ptr(u),
                                            C.c_double(dt_sim), C.c_int32(n_sub))
        self._check(rc, "lmpc_plant_step_batch")
        return x

    def loop_advance(self, track: dict, inp: dict, sol: dict, x, u_prev, dt: float, dt_sim: float, n_sub: int = 1, speed_scale: float = 1.0,
                     speed_limit: float | None = None, restart_failed: bool = True, distance=None, worst_excess=None, n_fail=None,
                     n_accepted=None):
        """lmpc_loop_advance_batch: input selection + plant step + the next period's inputs (shift, or cold restart of failed cars) +
        bookkeeping in one launch.  `inp`'s reference arrays, `x` [6][B] and `u_prev` [2][B] are updated in place; the accumulators
        (float64 [B], float64 [B], int64 [B], int64 scalar tensor) are optional."""
        self.use_current_stream()
        ct = self._ctrack(track)
        if speed_limit is None:
            speed_limit = float(self.config["x_max"][3])
        rc = self.lib.lmpc_loop_advance_batch(
            self._h, C.c_int32(x.shape[1]), C.byref(ct), _ptr(sol["status"]), _ptr(sol["iters"]), _ptr(sol["X_optm"]), _ptr(sol["U_optm"]),
            _ptr(x), _ptr(u_prev), C.c_double(dt), C.c_double(dt_sim), C.c_int32(n_sub), C.c_double(speed_scale), C.c_double(speed_limit),
            C.c_int32(1 if restart_failed else 0), *[_ptr(inp[k]) for k in ("X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")],
            _ptr(distance), _ptr(worst_excess), _ptr(n_fail), _ptr(n_accepted))
        self._check(rc, "lmpc_loop_advance_batch")

    # ---- discrete_dynamics_jacobian (single_track_planar_model.cpp:377-387) ----
    def linearize(self, inp: dict):
        torch = self._torch
        self.use_current_stream()
        X, U, T, kap = (self._t(inp[k]) for k in ("X_ref", "U_ref", "T_ref", "curvatures"))
        B, N = X.shape[2], self.N
        kw = dict(dtype=torch.float64, device=self.device)
        A = torch.empty((6, 6, N - 1, B), **kw)
        Bm = torch.empty((6, 2, N - 1, B), **kw)
        g = torch.empty((6, N - 1, B), **kw)
        rc = self.lib.lmpc_linearize_batch(self._h, C.c_int32(B), _ptr(X), _ptr(U), _ptr(T), _ptr(kap), _ptr(A),
                                           _ptr(Bm), _ptr(g))
        self._check(rc, "lmpc_linearize_batch")
        return A, Bm, g

    # ---- RacingMPC::solve (racing_mpc.cpp:209-372) ----
    def alloc_outputs(self, B: int, aos: bool | None = None):
        """Result tensors for a batch of B in the layout lmpc_solve_batch / _mixed write (set_output_layout), or, with
        aos=False, in the default layout whatever the setting (what every other entry point writes)."""
        torch = self._torch
        N = self.N
        kw = dict(dtype=torch.float64, device=self.device)
        if aos is None:
            aos = getattr(self, "_aos", False)
        shapes = ((B, N, 6), (B, N - 1, 2)) if aos else ((6, N, B), (2, N - 1, B))
        return {"X_optm": torch.empty(shapes[0], **kw), "U_optm": torch.empty(shapes[1], **kw),
                "dU_optm": torch.empty(shapes[1], **kw),
                "status": torch.empty((B,), dtype=torch.int32, device=self.device),
                "iters": torch.empty((B,), dtype=torch.int32, device=self.device),
                "kkt": torch.empty((4, B), **kw)}

    def solve(self, inp: dict, out: dict | None = None, ss_x=None, ss_j=None, mixed: bool = False, ss_idx=None, warm=None):
        """lmpc_solve_batch, or with mixed=True lmpc_solve_batch_mixed (same fp64 arrays, fp32 interior-point iteration); with
        ss_idx (int32 [S][B] from ss_query_idx) lmpc_solve_batch_ss_idx: the safe set by reference instead of ss_x / ss_j."""
        self.use_current_stream()
        keys = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
        a = [self._t(inp[k]) for k in keys]
        B = a[0].shape[1]
        if out is None:
            out = self.alloc_outputs(B)
        if warm is not None and mixed:
            # (this combination used to run the fp64 warm solve and say nothing)
            raise ValueError("Solver.solve: warm start and mixed precision do not combine (lmpc_solve_batch_warm is the fp64 entry)")
        if warm is not None and bool(self.config["learning"]):
            # lmpc_solve_batch_warm_ss: the plan AND its simplex weights (convex_combi_optm_ref, aligned with this call's points)
            wx = a[2] if warm is True else self._t(warm["X_optm_ref"])
            wu = a[3] if warm is True else self._t(warm["U_optm_ref"])
            wl = None if warm is True or warm.get("convex_combi_optm_ref") is None else self._t(warm["convex_combi_optm_ref"])
            sx = None if (ss_x is None or ss_idx is not None) else self._t(ss_x)
            sj = None if (ss_j is None or ss_idx is not None) else self._t(ss_j)
            rc = self.lib.lmpc_solve_batch_warm_ss(self._h, C.c_int32(B), *[_ptr(t) for t in a], C.c_double(float(inp["L"])), _ptr(sx), _ptr(sj),
                                                   _ptr(ss_idx), _ptr(wx), _ptr(wu), _ptr(wl), _ptr(out["X_optm"]), _ptr(out["U_optm"]),
                                                   _ptr(out["dU_optm"]), _ptr(out.get("convex_combi_optm")), _ptr(out["status"]), _ptr(out["iters"]),
                                                   _ptr(out.get("kkt")))
            self._check(rc, "lmpc_solve_batch_warm_ss")
            out["_inputs_keepalive"] = a + [wx, wu, wl, sx, sj, ss_idx]
            return out
        if warm is not None and ss_idx is not None:
            raise ValueError("Solver.solve: ss_idx goes with a learning handle")
        if warm is not None:
            # lmpc_solve_batch_warm: warm = True takes (X_ref, U_ref) as the plan (what the node does), or a dict with X_optm_ref / U_optm_ref
            wx = a[2] if warm is True else self._t(warm["X_optm_ref"])
            wu = a[3] if warm is True else self._t(warm["U_optm_ref"])
            rc = self.lib.lmpc_solve_batch_warm(self._h, C.c_int32(B), *[_ptr(t) for t in a], C.c_double(float(inp["L"])), _ptr(wx), _ptr(wu),
                                                _ptr(out["X_optm"]), _ptr(out["U_optm"]), _ptr(out["dU_optm"]), _ptr(out["status"]),
                                                _ptr(out["iters"]), _ptr(out.get("kkt")))
            self._check(rc, "lmpc_solve_batch_warm")
            out["_inputs_keepalive"] = a + [wx, wu]
            return out
        if ss_idx is not None:
            rc = self.lib.lmpc_solve_batch_ss_idx(self._h, C.c_int32(B), C.c_int32(2 if mixed else 0), *[_ptr(t) for t in a],
                                                  C.c_double(float(inp["L"])), _ptr(ss_idx), _ptr(out["X_optm"]), _ptr(out["U_optm"]),
                                                  _ptr(out["dU_optm"]), _ptr(out.get("convex_combi_optm")), _ptr(out["status"]),
                                                  _ptr(out["iters"]), _ptr(out.get("kkt")))
            self._check(rc, "lmpc_solve_batch_ss_idx")
            out["_inputs_keepalive"] = a + [ss_idx]
            return out
        ss_x = None if ss_x is None else self._t(ss_x)
        ss_j = None if ss_j is None else self._t(ss_j)
        fn = self.lib.lmpc_solve_batch_mixed if mixed else self.lib.lmpc_solve_batch
        rc = fn(self._h, C.c_int32(B), *[_ptr(t) for t in a], C.c_double(float(inp["L"])),
                _ptr(ss_x), _ptr(ss_j), _ptr(out["X_optm"]), _ptr(out["U_optm"]),
                _ptr(out["dU_optm"]), _ptr(out.get("convex_combi_optm")),
                _ptr(out["status"]), _ptr(out["iters"]), _ptr(out.get("kkt")))
        self._check(rc, "lmpc_solve_batch_mixed" if mixed else "lmpc_solve_batch")
        out["_inputs_keepalive"] = a
        return out

    def shift_lambda(self, idx_prev, lam_prev, idx, advance: int = 1, out=None):
        """lmpc_shift_lambda_batch: the previous solution's simplex weights carried onto this period's safe-set codes."""
        torch = self._torch
        self.use_current_stream()
        S, B = idx.shape
        if out is None:
            out = torch.empty((S, B), dtype=torch.float64, device=self.device)
        self._check(self.lib.lmpc_shift_lambda_batch(self._h, C.c_int32(B), _ptr(idx_prev), _ptr(lam_prev), _ptr(idx), C.c_int32(int(advance)), _ptr(out)),
                    "lmpc_shift_lambda_batch")
        return out

    def warm_accepted(self, B: int, out=None):
        """lmpc_get_warm_accepted: int32 [B], 1 where the last warm solve's active-set attempt was accepted."""
        torch = self._torch
        self.use_current_stream()
        if out is None:
            out = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._check(self.lib.lmpc_get_warm_accepted(self._h, C.c_int32(B), _ptr(out)), "lmpc_get_warm_accepted")
        return out

    def set_waves_per_problem(self, waves: int):
        """lmpc_set_waves_per_problem: 0 the library's choice, 1 / 2 wavefronts per problem (2: fp64 tracking, N >= 24, cold solves)."""
        self._check(self.lib.lmpc_set_waves_per_problem(self._h, C.c_int32(int(waves))), "lmpc_set_waves_per_problem")

    def last_solve_precision(self) -> str:
        """lmpc_last_solve_precision: "f64", "f32" or "mixed" -- what the most recent batched solve ran in (solve(mixed=True) falls
        back to fp64 where no reduced-precision kernel exists for this (N, num_ss_pts): include/lmpc_hip.h)."""
        p = C.c_int32(-1)
        self._check(self.lib.lmpc_last_solve_precision(self._h, C.byref(p)), "lmpc_last_solve_precision")
        return {0: "f64", 1: "f32", 2: "mixed"}[p.value]

    # ---- single precision (BASELINE configs[3]) ----
    def solve_f32(self, inp: dict, out: dict | None = None):
        """lmpc_solve_batch_f32: every array in float32 (inputs are converted if they are not), tracking problem only."""
        torch = self._torch
        self.use_current_stream()
        keys = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
        a = [torch.as_tensor(inp[k], device=self.device).to(torch.float32).contiguous() for k in keys]
        B, N = a[0].shape[1], self.N
        if out is None:
            kw = dict(dtype=torch.float32, device=self.device)
            # (always the default layout: lmpc_set_output_layout applies to lmpc_solve_batch / _mixed only -- include/lmpc_hip.h --
            # and until round 5 this allocated [B][N][6] under the AOS setting for a kernel that writes [6][N][B]: ADVICE r4)
            shapes = ((6, N, B), (2, N - 1, B))
            out = {"X_optm": torch.empty(shapes[0], **kw), "U_optm": torch.empty(shapes[1], **kw),
                   "dU_optm": torch.empty(shapes[1], **kw), "kkt": torch.empty((4, B), **kw),
                   "status": torch.empty((B,), dtype=torch.int32, device=self.device),
                   "iters": torch.empty((B,), dtype=torch.int32, device=self.device)}
        rc = self.lib.lmpc_solve_batch_f32(self._h, C.c_int32(B), *[_ptr(t) for t in a], _ptr(out["X_optm"]),
                                           _ptr(out["U_optm"]), _ptr(out["dU_optm"]), _ptr(out["status"]),
                                           _ptr(out["iters"]), _ptr(out.get("kkt")))
        self._check(rc, "lmpc_solve_batch_f32")
        out["_inputs_keepalive"] = a
        return out

    # ---- full_dynamics = true (racing_mpc.cpp:162-166; IPOPT upstream, racing_mpc_node.cpp:299-314) ----
    def solve_full_dynamics(self, inp: dict, max_sqp: int = 10, tol: float = 1e-9, ss_x=None, ss_j=None):
        """lmpc_solve_full_dynamics_batch: the nonlinear-dynamics problem x_{i+1} = f_d(x_i, u_i, k_i, t_i) by sequential
        QPs over the same kernels with a line search on the l1 merit function (csrc/lmpc_sqp_kernel.hip); the per-knot
        parameters -- curvature, bounds, vel_ref -- stay fixed, as they are parameters of the reference's NLP too.
        Problems whose QP fails keep their last iterate and report that QP's status.  Returns the iterate reached plus
        "sqp_iters" [B], "sqp_move" [B] (scaled size of the last step) and "defect" [B] (scaled dynamics defect)."""
        torch = self._torch
        self.use_current_stream()
        a = {k: self._t(inp[k]) for k in ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right",
                                          "curvatures", "vel_ref")}
        B = a["x_ic"].shape[1]
        out = self.alloc_outputs(B, aos=False)   # (the output layout setting applies to lmpc_solve_batch / _mixed only)
        kw = dict(device=self.device)
        out["sqp_iters"] = torch.zeros((B,), dtype=torch.int32, **kw)
        out["sqp_move"] = torch.zeros((B,), dtype=torch.float64, **kw)
        out["defect"] = torch.zeros((B,), dtype=torch.float64, **kw)
        learning = bool(self.config["learning"])
        if learning:
            out["convex_combi_optm"] = torch.zeros((int(self.config["num_ss_pts"]), B), dtype=torch.float64, **kw)
            ss_x, ss_j = self._t(ss_x), self._t(ss_j)
        rc = self.lib.lmpc_solve_full_dynamics_batch(
            self._h, C.c_int32(B), *[_ptr(a[k]) for k in ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left",
                                                           "bound_right", "curvatures", "vel_ref")],
            C.c_double(float(inp.get("L", 0.0))), _ptr(ss_x) if learning else None, _ptr(ss_j) if learning else None,
            C.c_int32(int(max_sqp)), C.c_double(float(tol)), _ptr(out["X_optm"]), _ptr(out["U_optm"]), _ptr(out["dU_optm"]),
            _ptr(out["convex_combi_optm"]) if learning else None, _ptr(out["status"]), _ptr(out["iters"]),
            _ptr(out["sqp_iters"]), _ptr(out["sqp_move"]), _ptr(out["defect"]))
        self._check(rc, "lmpc_solve_full_dynamics_batch")
        out["_inputs_keepalive"] = a
        return out

    # ---- safe set (safe_set.cpp:116-180) ----
    def set_safe_set(self, laps_x, total_length: float):
        import numpy as np

        n_pts = np.array([np.asarray(a).shape[0] for a in laps_x], dtype=np.int32)
        x = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1, 6) for a in laps_x], 0)) \
            if len(laps_x) else np.zeros((0, 6))
        rc = self.lib.lmpc_set_safe_set(self._h, C.c_int32(len(laps_x)), n_pts.ctypes.data_as(C.c_void_p),
                                        x.ctypes.data_as(C.c_void_p), C.c_double(total_length))
        self._check(rc, "lmpc_set_safe_set")

    # ---- error-dynamics regression (safe_set.cpp:56-114, 182-245) ----
    def set_regression_laps(self, laps, in_state=(3, 4, 5), in_ctrl=(0, 1), out_rows=(3, 4, 5), dist_max: float = 1.0,
                            as_written: bool = False):
        """laps: list of (x [n,6], u [n,2], k [n], t [n]) host arrays; an empty list switches the regression off.
        as_written = True reproduces the reference's literal signs (backward time step, b = -M'K y; lmpc_hip.h)."""
        import numpy as np

        if not laps:
            self._check(self.lib.lmpc_set_regression_laps(self._h, C.c_int32(0), None, None, None, None, None, None),
                        "lmpc_set_regression_laps")
            return
        n_pts = np.array([np.asarray(l[0]).shape[0] for l in laps], dtype=np.int32)
        cat = [np.ascontiguousarray(np.concatenate([np.asarray(l[i], dtype=np.float64).reshape(n, -1) for l, n in zip(laps, n_pts)], 0))
               for i in range(4)]
        spec = CRegressionSpec()
        spec.n_out, spec.n_in_state, spec.n_in_ctrl, spec.dist_max = len(out_rows), len(in_state), len(in_ctrl), float(dist_max)
        spec.out[:len(out_rows)] = list(out_rows)
        spec.in_state[:len(in_state)] = list(in_state)
        spec.in_ctrl[:len(in_ctrl)] = list(in_ctrl)
        spec.as_written = 1 if as_written else 0
        rc = self.lib.lmpc_set_regression_laps(self._h, C.c_int32(len(laps)), n_pts.ctypes.data_as(C.c_void_p),
                                               *[a.ctypes.data_as(C.c_void_p) for a in cat], C.byref(spec))
        self._check(rc, "lmpc_set_regression_laps")

    def regress(self, inp: dict, A, Bm, g):
        """Adds RegResult onto A [6,6,N-1,B], Bm [6,2,N-1,B], g [6,N-1,B] (device tensors, in place)."""
        self.use_current_stream()
        X, U = self._t(inp["X_ref"]), self._t(inp["U_ref"])
        rc = self.lib.lmpc_regress_batch(self._h, C.c_int32(X.shape[2]), _ptr(X), _ptr(U), _ptr(A), _ptr(Bm), _ptr(g))
        self._check(rc, "lmpc_regress_batch")
        return A, Bm, g

    def ss_query_idx(self, query, out=None):
        """lmpc_ss_query_idx_batch: the safe set of every query by reference -- (ss_idx int32 [S][B], n_found [B]); `out` reuses buffers."""
        torch = self._torch
        self.use_current_stream()
        q = self._t(query)
        B = q.shape[1]
        S = int(self.config["num_ss_pts"])
        if out is not None:
            ss_idx, nf = out
        else:
            ss_idx = torch.full((S, B), -1, dtype=torch.int32, device=self.device)
            nf = torch.zeros((B,), dtype=torch.int32, device=self.device)
        rc = self.lib.lmpc_ss_query_idx_batch(self._h, C.c_int32(B), _ptr(q), _ptr(ss_idx), _ptr(nf))
        self._check(rc, "lmpc_ss_query_idx_batch")
        return ss_idx, nf

    def ss_query(self, query, out=None):
        """lmpc_ss_query_batch.  `out` = (ss_x [6][S][B], ss_j [S][B], n_found [B]) reuses the caller's buffers (the
        kernel writes every entry of a query that found at least one point); otherwise zero-filled ones are allocated."""
        torch = self._torch
        self.use_current_stream()
        q = self._t(query)
        B = q.shape[1]
        S = int(self.config["num_ss_pts"])
        kw = dict(dtype=torch.float64, device=self.device)
        if out is not None:
            ss_x, ss_j, nf = out
        else:
            ss_x = torch.zeros((6, S, B), **kw)
            ss_j = torch.zeros((S, B), **kw)
            nf = torch.zeros((B,), dtype=torch.int32, device=self.device)
        rc = self.lib.lmpc_ss_query_batch(self._h, C.c_int32(B), _ptr(q), _ptr(ss_x), _ptr(ss_j), _ptr(nf))
        self._check(rc, "lmpc_ss_query_batch")
        return ss_x, ss_j, nf
