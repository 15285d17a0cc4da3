"""The reference's ROS 2 parameter files -> the C ABI's lmpc_vehicle / lmpc_config.

A node of the reference receives several `*.param.yaml` files (vehicle base, single-track, MPC) merged by the launch
file; each is `/**: ros__parameters: <group>: <key>: value`.  The loaders below read the same dotted keys as
  base_vehicle_model/src/ros_param_loader.cpp:30-174        (chassis.*, aero.*, steer.*, front_tyre.*, ...)
  single_track_planar_model/src/ros_param_loader.cpp:30-52  (single_track_planar.*)
  mpc/racing_mpc/src/ros_param_loader.cpp:30-104            (racing_mpc.*)
and keep their contract: every key the reference declares is mandatory (declare_parameter rethrows,
lmpc_utils/ros_param_helper.hpp:28-54) -- a missing one raises KeyError naming it.  Keys the device path has no use for
(tyre geometry, brake hardware, powertrain map) are not required here.

Field mapping onto lmpc_vehicle follows where single_track_planar_model.cpp:53-159 reads each quantity.
"""
from __future__ import annotations

import math
from pathlib import Path

__all__ = ["load_ros_params", "vehicle_from_params", "mpc_config_from_params", "host_options_from_params"]


def _scalar(v):
    """rcl's YAML parser reads `1e-3` as a double; PyYAML (YAML 1.1) leaves it a string."""
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return v
    return v


def _flatten(tree, prefix, out):
    for k, v in tree.items():
        key = f"{prefix}.{k}" if prefix else str(k)
        if isinstance(v, dict):
            _flatten(v, key, out)
        elif isinstance(v, (list, tuple)):
            out[key] = [_scalar(e) for e in v]
        else:
            out[key] = _scalar(v)


def load_ros_params(*paths) -> dict:
    """Merge ROS 2 parameter files into one {dotted key: value} map (later files win, as later `parameters=` entries do).
    Accepts the wildcard node name `/**` and any explicit node name at the top level."""
    import yaml

    out: dict = {}
    for p in paths:
        doc = yaml.safe_load(Path(p).read_text())
        if not isinstance(doc, dict):
            raise ValueError(f"{p}: not a ROS 2 parameter file")
        for node, body in doc.items():
            if not isinstance(body, dict) or "ros__parameters" not in body:
                raise ValueError(f"{p}: node entry {node!r} has no ros__parameters")
            _flatten(body["ros__parameters"] or {}, "", out)
    return out


def _need(params: dict, key: str, kind):
    if key not in params:
        raise KeyError(f"parameter {key!r} is not set (the reference declares it without a default)")
    v = params[key]
    if kind is float:
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise TypeError(f"parameter {key!r}: expected a number, got {v!r}")
        return float(v)
    if kind is int:
        if isinstance(v, bool) or not isinstance(v, int):
            if isinstance(v, float) and v.is_integer():
                return int(v)
            raise TypeError(f"parameter {key!r}: expected an integer, got {v!r}")
        return int(v)
    if kind is bool:
        if not isinstance(v, bool):
            raise TypeError(f"parameter {key!r}: expected a bool, got {v!r}")
        return v
    if kind is list:
        if not isinstance(v, list) or any(isinstance(e, bool) or not isinstance(e, (int, float)) for e in v):
            raise TypeError(f"parameter {key!r}: expected a list of numbers, got {v!r}")
        return [float(e) for e in v]
    if kind is str:
        if not isinstance(v, str):
            raise TypeError(f"parameter {key!r}: expected a string, got {v!r}")
        return v
    raise AssertionError(kind)


def vehicle_from_params(params: dict, model: str = "single_track_planar_model") -> dict:
    """lmpc_vehicle fields from the merged vehicle parameter files.  Raises NotImplementedError for what the device
    path does not build: other models (vehicle_model_factory.cpp:31-49), separate throttle / brake inputs
    (simplify_lon_control = false, nu = 3), a Cartesian model or the Euler integrator."""
    if model != "single_track_planar_model":
        raise NotImplementedError(f"vehicle model {model!r}: only single_track_planar_model is built")
    f = lambda k: _need(params, k, float)  # noqa: E731
    if not _need(params, "single_track_planar.simplify_lon_control", bool):
        raise NotImplementedError("single_track_planar.simplify_lon_control = false (nu = 3) is not built")
    if not _need(params, "modeling.use_frenet", bool):
        raise NotImplementedError("modeling.use_frenet = false is not built (RacingMPC uses the Frenet model)")
    integ = _need(params, "modeling.integrator_type", str)
    if integ not in ("rk4", "euler"):   # base_vehicle_model_config.cpp: any other string throws
        raise ValueError(f"Unknown integrator type: {integ}")
    return dict(model_id=0, integrator=integ,
                m=f("chassis.total_mass"), Jzz=f("chassis.moi"), l=f("chassis.wheel_base"),
                cg_ratio=f("chassis.cg_ratio"), h=f("chassis.cg_height"), b=f("chassis.b"), fr=f("chassis.fr"),
                kd=f("powertrain.kd"), kb=f("front_brake.bias"),
                cd=f("aero.drag_coeff"), Af=f("aero.frontal_area"), rho=f("aero.air_density"),
                cl_f=f("aero.cl_f"), cl_r=f("aero.cl_r"), mu=f("single_track_planar.mu"),
                Bf=f("front_tyre.pacejka_b"), Cf=f("front_tyre.pacejka_c"),
                Br=f("rear_tyre.pacejka_b"), Cr=f("rear_tyre.pacejka_c"),
                Fd_max=f("single_track_planar.fd_max"), Fb_max=f("single_track_planar.fb_max"),
                Td=f("single_track_planar.td"), Tb=f("single_track_planar.tb"),
                max_steer=f("steer.max_steer"), max_steer_rate=f("steer.max_steer_rate"))


def mpc_config_from_params(params: dict, horizon: int | None = None) -> dict:
    """lmpc_config fields from `racing_mpc.*`.  `horizon` overrides racing_mpc.n (the BASELINE configurations run the
    shipped weights at N = 20 / 40).

    racing_mpc.tol / max_iter / max_cpu_time / jit configure IPOPT and CasADi code generation upstream; the QP path
    (OSQP, racing_mpc.cpp:86-104) is constructed without them, so they do not set this library's interior-point
    tolerance either: max_iter = 0 and tol = 0 select the library defaults (40 iterations, complementarity 1e-11)."""
    f = lambda k: _need(params, "racing_mpc." + k, float)  # noqa: E731
    i = lambda k: _need(params, "racing_mpc." + k, int)  # noqa: E731
    v = lambda k: _need(params, "racing_mpc." + k, list)  # noqa: E731
    for k in ("max_cpu_time", "tol", "average_track_width"):
        f(k)
    i("max_iter")
    for k in ("verbose", "jit"):
        _need(params, "racing_mpc." + k, bool)
    mode = _need(params, "racing_mpc.step_mode", str)
    if mode not in ("step", "continuous"):
        raise ValueError("Invalid step mode: " + mode)
    R, R_d = v("r"), v("r_d")
    for name, m in (("r", R), ("r_d", R_d)):
        n = math.isqrt(len(m))
        if n * n != len(m):
            raise ValueError(f"racing_mpc.{name}: {len(m)} entries do not form a square matrix")
        if n != 2:
            raise NotImplementedError(f"racing_mpc.{name} is {n}x{n}: the device path is built for nu = 2")
    sizes = {"x_max": 6, "x_min": 6, "u_max": 2, "u_min": 2, "convex_hull_slack": 6}
    vec = {}
    for k, n in sizes.items():
        vec[k] = v(k)
        if len(vec[k]) != n:
            raise ValueError(f"racing_mpc.{k}: expected {n} entries, got {len(vec[k])}")
    N = i("n") if horizon is None else int(horizon)
    return dict(N=N, learning=int(_need(params, "racing_mpc.learning", bool)), num_ss_pts=i("num_ss_pts"),
                num_ss_pts_per_lap=i("num_ss_pts_per_lap"), max_lap_stored=i("max_lap_stored"), max_iter=0, tol=0.0,
                margin=f("margin"), q_contour=f("q_contour"), q_heading=f("q_heading"), q_vel=f("q_vel"),
                q_vy=f("q_vy"), q_vyaw=f("q_vyaw"), q_boundary=f("q_boundary"), R=R, R_d=R_d,
                x_max=vec["x_max"], x_min=vec["x_min"], u_max=vec["u_max"], u_min=vec["u_min"],
                convex_hull_slack=vec["convex_hull_slack"], max_vel_ref_diff=f("max_vel_ref_diff"))


def host_options_from_params(params: dict, strict: bool = True) -> dict:
    """The host-side members of RacingMPCConfig (racing_mpc_config.hpp:37-82): safe-set recording and loading, step
    mode, verbosity.  strict=False fills `load` / `load_path` / `record` / `path_prefix` with "off" when a file leaves
    them out (barc_lmpc.param.yaml ships with `load` commented out, which the reference's loader rejects)."""
    def get(key, kind, default):
        full = "racing_mpc." + key
        if full in params or strict:
            return _need(params, full, kind)
        return default
    load_path = params.get("racing_mpc.load_path", None)
    if load_path is None:
        if strict:
            raise KeyError("parameter 'racing_mpc.load_path' is not set (the reference declares it without a default)")
        load_path = []
    if not isinstance(load_path, list) or any(not isinstance(e, str) for e in load_path):
        raise TypeError(f"parameter 'racing_mpc.load_path': expected a list of strings, got {load_path!r}")
    return dict(record=get("record", bool, False), path_prefix=get("path_prefix", str, ""),
                load=get("load", bool, False), load_path=list(load_path),
                step_mode=_need(params, "racing_mpc.step_mode", str), verbose=_need(params, "racing_mpc.verbose", bool))
