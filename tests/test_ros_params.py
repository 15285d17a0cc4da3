"""The reference's ROS 2 parameter files -> lmpc_vehicle / lmpc_config (ros_params.py) -- CPU only.

The files written here are this repository's own, in the reference's format (`/**: ros__parameters:` trees with the keys
its loaders declare).  When the reference checkout is present (this container, not the GPU box) its shipped files are
read as well, which pins `presets.py` -- the values every parity test and bench line runs on -- to them."""
from pathlib import Path

import pytest
import yaml

REF_PARAM = Path("/root/reference/src/launch/racing_lmpc_launch/param")

VEHICLE_KEYS = {  # lmpc_vehicle field -> the reference's parameter (where single_track_planar_model.cpp reads it)
    "m": "chassis.total_mass", "Jzz": "chassis.moi", "l": "chassis.wheel_base", "cg_ratio": "chassis.cg_ratio",
    "h": "chassis.cg_height", "b": "chassis.b", "fr": "chassis.fr", "kd": "powertrain.kd", "kb": "front_brake.bias",
    "cd": "aero.drag_coeff", "Af": "aero.frontal_area", "rho": "aero.air_density", "cl_f": "aero.cl_f",
    "cl_r": "aero.cl_r", "mu": "single_track_planar.mu", "Bf": "front_tyre.pacejka_b", "Cf": "front_tyre.pacejka_c",
    "Br": "rear_tyre.pacejka_b", "Cr": "rear_tyre.pacejka_c", "Fd_max": "single_track_planar.fd_max",
    "Fb_max": "single_track_planar.fb_max", "Td": "single_track_planar.td", "Tb": "single_track_planar.tb",
    "max_steer": "steer.max_steer", "max_steer_rate": "steer.max_steer_rate"}


def nest(flat: dict) -> dict:
    tree: dict = {}
    for key, v in flat.items():
        node = tree
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return {"/**": {"ros__parameters": tree}}


def vehicle_files(tmp_path, veh: dict, **override):
    base = {VEHICLE_KEYS[k]: v for k, v in veh.items() if k in VEHICLE_KEYS and not VEHICLE_KEYS[k].startswith("single_track")}
    base.update({"modeling.use_frenet": True, "modeling.integrator_type": "rk4", "modeling.sample_throttle": 60.0})
    st = {VEHICLE_KEYS[k]: v for k, v in veh.items() if k in VEHICLE_KEYS and VEHICLE_KEYS[k].startswith("single_track")}
    st.update({"single_track_planar.v_max": 5.0, "single_track_planar.p_max": 550.0,
               "single_track_planar.simplify_lon_control": True})
    for k, v in override.items():
        (st if k.startswith("single_track") else base)[k] = v
    a, b = tmp_path / "veh_base.param.yaml", tmp_path / "veh_single_track.param.yaml"
    a.write_text(yaml.safe_dump(nest(base)))
    b.write_text(yaml.safe_dump(nest(st)))
    return a, b


def mpc_text(cfg: dict, n: int, extra: str = "") -> str:
    """Hand-written in the style of the shipped files: bare `1e-3`, `.inf`, multi-line lists, comments."""
    inf = lambda v: (".inf" if v == float("inf") else "-.inf" if v == float("-inf") else repr(v))  # noqa: E731
    lst = lambda xs: "[" + ", ".join(inf(x) for x in xs) + "]"  # noqa: E731
    return f"""/**:
  ros__parameters:
    racing_mpc:
      max_cpu_time: 0.085
      max_iter: 200
      tol: 1e-3
      n: {n}
      margin: {cfg['margin']}
      average_track_width: 1.0
      verbose: false
      jit: true
      q_contour: {cfg['q_contour']}
      q_heading: {cfg['q_heading']}
      q_boundary: {cfg['q_boundary']} # 0 to disable
      q_vel: {cfg['q_vel']}
      q_vy: {cfg['q_vy']}
      q_vyaw: {cfg['q_vyaw']}
      r: [
        {cfg['R'][0]}, {cfg['R'][1]},
        {cfg['R'][2]}, {cfg['R'][3]},
      ]
      r_d: {lst(cfg['R_d'])}
      max_vel_ref_diff: {cfg['max_vel_ref_diff']}
      x_max: {lst(cfg['x_max'])}
      x_min: {lst(cfg['x_min'])}
      u_max: {lst(cfg['u_max'])}
      u_min: {lst(cfg['u_min'])}
      step_mode: "continuous"
      learning: {'true' if cfg['learning'] else 'false'}
      convex_hull_slack: {lst(cfg['convex_hull_slack'])}
      num_ss_pts: {cfg['num_ss_pts']}
      num_ss_pts_per_lap: {cfg['num_ss_pts_per_lap']}
      max_lap_stored: {cfg['max_lap_stored']}
      record: false
      path_prefix: "/tmp/ss_"
{extra}"""


LOAD = """      load: true
      load_path:
        - /data/ss_lap_1
        - /data/ss_lap_2
"""


@pytest.mark.parametrize("vehicle", ["barc_vehicle", "iac_vehicle"])
def test_vehicle_files_round_trip(pkg, tmp_path, vehicle):
    want = getattr(pkg.presets, vehicle)()
    got = pkg.ros_params.vehicle_from_params(pkg.ros_params.load_ros_params(*vehicle_files(tmp_path, want)))
    assert got == want


@pytest.mark.parametrize("preset,n", [("barc_tracking_mpc", 60), ("barc_lmpc", 40), ("iac_tracking_mpc", 80)])
def test_mpc_file_round_trip(pkg, tmp_path, preset, n):
    want = getattr(pkg.presets, preset)(n)
    f = tmp_path / "mpc.param.yaml"
    f.write_text(mpc_text(want, n, LOAD))
    params = pkg.ros_params.load_ros_params(f)
    assert params["racing_mpc.tol"] == 1e-3                      # bare exponent read as a number, as rcl does
    assert pkg.ros_params.mpc_config_from_params(params) == want
    assert pkg.ros_params.mpc_config_from_params(params, horizon=20) == getattr(pkg.presets, preset)(20)
    host = pkg.ros_params.host_options_from_params(params)
    assert host == dict(record=False, path_prefix="/tmp/ss_", load=True, load_path=["/data/ss_lap_1", "/data/ss_lap_2"],
                        step_mode="continuous", verbose=False)


def test_every_declared_key_is_mandatory(pkg, tmp_path):
    rp = pkg.ros_params
    cfg = pkg.presets.barc_tracking_mpc(20)
    f = tmp_path / "mpc.param.yaml"
    f.write_text("\n".join(ln for ln in mpc_text(cfg, 20, LOAD).splitlines() if "q_vyaw" not in ln))
    with pytest.raises(KeyError, match="racing_mpc.q_vyaw"):
        rp.mpc_config_from_params(rp.load_ros_params(f))
    f.write_text(mpc_text(cfg, 20))                               # `load` left out, as barc_lmpc.param.yaml ships
    params = rp.load_ros_params(f)
    with pytest.raises(KeyError, match="racing_mpc.load"):
        rp.host_options_from_params(params)
    assert rp.host_options_from_params(params, strict=False)["load"] is False
    a, b = vehicle_files(tmp_path, pkg.presets.barc_vehicle())
    params = rp.load_ros_params(a, b)
    del params["chassis.moi"]
    with pytest.raises(KeyError, match="chassis.moi"):
        rp.vehicle_from_params(params)
    f.write_text(mpc_text(cfg, 20, LOAD).replace('step_mode: "continuous"', 'step_mode: "sometimes"'))
    with pytest.raises(ValueError, match="Invalid step mode"):
        rp.mpc_config_from_params(rp.load_ros_params(f))
    f.write_text(mpc_text(cfg, 20, LOAD).replace("q_vel: 0.2", 'q_vel: "fast"'))
    with pytest.raises(TypeError, match="racing_mpc.q_vel"):
        rp.mpc_config_from_params(rp.load_ros_params(f))


def test_what_the_device_path_does_not_build_is_refused(pkg, tmp_path):
    rp, veh = pkg.ros_params, pkg.presets.barc_vehicle()
    with pytest.raises(NotImplementedError, match="simplify_lon_control"):
        rp.vehicle_from_params(rp.load_ros_params(*vehicle_files(tmp_path, veh, **{"single_track_planar.simplify_lon_control": False})))
    # modeling.integrator_type = euler is built (utils.cpp:110-123): it selects the integrator field of lmpc_vehicle
    assert rp.vehicle_from_params(rp.load_ros_params(*vehicle_files(tmp_path, veh, **{"modeling.integrator_type": "euler"})))["integrator"] == "euler"
    assert rp.vehicle_from_params(rp.load_ros_params(*vehicle_files(tmp_path, veh)))["integrator"] == "rk4"
    with pytest.raises(ValueError, match="Unknown integrator type"):
        rp.vehicle_from_params(rp.load_ros_params(*vehicle_files(tmp_path, veh, **{"modeling.integrator_type": "rk9"})))
    with pytest.raises(NotImplementedError, match="only single_track_planar_model"):
        rp.vehicle_from_params({}, model="kinematic_bicycle_model")
    (tmp_path / "bad.yaml").write_text("racing_mpc:\n  n: 3\n")
    with pytest.raises(ValueError, match="ros__parameters"):
        rp.load_ros_params(tmp_path / "bad.yaml")


@pytest.mark.skipif(not REF_PARAM.is_dir(), reason="the reference checkout is not on this machine")
def test_presets_equal_the_reference_shipped_files(pkg):
    rp, pr = pkg.ros_params, pkg.presets
    for car, preset in (("barc", pr.barc_vehicle), ("iac_car", pr.iac_vehicle)):
        params = rp.load_ros_params(REF_PARAM / car / f"{car}_base.param.yaml", REF_PARAM / car / f"{car}_single_track.param.yaml")
        assert rp.vehicle_from_params(params) == preset()
    for name, preset, n in (("barc_tracking_mpc", pr.barc_tracking_mpc, 60), ("barc_lmpc", pr.barc_lmpc, 40),
                            ("iac_car_tracking_mpc", pr.iac_tracking_mpc, 80), ("iac_car_lmpc", pr.iac_lmpc, 60)):
        params = rp.load_ros_params(REF_PARAM / "racing_mpc" / f"{name}.param.yaml")
        assert params["racing_mpc.n"] == n
        assert rp.mpc_config_from_params(params) == preset(n), name


def test_oracle_parameter_sets_equal_the_package_presets(pkg):
    """oracle.params (what the golden vectors and the twin run on) and presets.py (what the HIP path runs on) are two
    transcriptions of the same files: field by field equal, so the check against the shipped YAML covers both."""
    import dataclasses

    import numpy as np
    from oracle import params as OP

    for a, b in ((OP.barc_vehicle(), pkg.presets.barc_vehicle()), (OP.iac_vehicle(), pkg.presets.iac_vehicle())):
        for k, v in dataclasses.asdict(a).items():
            assert v == b[k], k
    for a, b in ((OP.barc_tracking_mpc(60), pkg.presets.barc_tracking_mpc(60)), (OP.barc_lmpc(40, 3), pkg.presets.barc_lmpc(40, 3)),
                 (OP.barc_lmpc(20, 5), pkg.presets.barc_lmpc(20, 5)), (OP.iac_tracking_mpc(80), pkg.presets.iac_tracking_mpc(80)),
                 (OP.iac_lmpc(60, 3), pkg.presets.iac_lmpc(60, 3))):
        for k, v in dataclasses.asdict(a).items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(v.ravel(), np.asarray(b[k], dtype=float).ravel()), k
            else:
                assert float(v) == float(b[k]), k
