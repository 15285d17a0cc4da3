"""Vehicle and MPC parameter sets (oracle side; test infrastructure).

Values are transcribed from the reference's shipped YAML:
  BARC vehicle  src/launch/racing_lmpc_launch/param/barc/barc_base.param.yaml:3-152,
                .../barc/barc_single_track.param.yaml:4-11
  IAC vehicle   .../param/iac_car/iac_car_base.param.yaml, iac_car_single_track.param.yaml
  MPC           .../param/racing_mpc/{barc_tracking_mpc,barc_lmpc,iac_car_tracking_mpc}.param.yaml
Field meaning follows RacingMPCConfig (racing_mpc_config.hpp:37-82),
BaseVehicleModelConfig (base_vehicle_model_config.hpp:30-154) and
SingleTrackPlanarModelConfig (single_track_planar_model.hpp:31-43).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field

import numpy as np

INF = float("inf")

# racing_mpc.cpp:36-37 (hard-coded for every vehicle)
SCALE_X = np.array([2000.0, 10.0, 0.1, 80.0, 2.0, 2.0])
SCALE_U = np.array([10.0, 0.3])


@dataclass
class Vehicle:
    # chassis
    m: float
    Jzz: float
    l: float
    cg_ratio: float
    h: float
    b: float
    fr: float
    # powertrain / brake split
    kd: float
    kb: float
    # aero
    cd: float
    Af: float
    rho: float
    cl_f: float
    cl_r: float
    # tyres
    mu: float
    Bf: float
    Cf: float
    Br: float
    Cr: float
    # actuators (single_track_planar)
    Fd_max: float
    Fb_max: float
    Td: float
    Tb: float
    max_steer: float
    max_steer_rate: float
    integrator: str = "rk4"   # modeling.integrator_type: "rk4" | "euler"

    def as_array(self) -> np.ndarray:
        """Order of lmpc_vehicle in include/lmpc_hip.h."""
        return np.array([
            self.m, self.Jzz, self.l, self.cg_ratio, self.h, self.b, self.fr,
            self.kd, self.kb, self.cd, self.Af, self.rho, self.cl_f, self.cl_r,
            self.mu, self.Bf, self.Cf, self.Br, self.Cr,
            self.Fd_max, self.Fb_max, self.Td, self.Tb,
            self.max_steer, self.max_steer_rate], dtype=np.float64)


@dataclass
class MPCConfig:
    N: int
    margin: float
    q_contour: float
    q_heading: float
    q_vel: float
    q_vy: float
    q_vyaw: float
    q_boundary: float
    R: np.ndarray
    R_d: np.ndarray
    x_max: np.ndarray
    x_min: np.ndarray
    u_max: np.ndarray
    u_min: np.ndarray
    max_vel_ref_diff: float = 1.0
    learning: bool = False
    convex_hull_slack: np.ndarray = field(default_factory=lambda: np.array([20.0, 20.0, 2.0, 20.0, 20.0, 2.0]))
    num_ss_pts: int = 96
    num_ss_pts_per_lap: int = 32
    max_lap_stored: int = 3

    def with_(self, **kw) -> "MPCConfig":
        c = copy.deepcopy(self)
        for k, v in kw.items():
            setattr(c, k, v)
        return c


def barc_vehicle() -> Vehicle:
    return Vehicle(m=2.2187, Jzz=0.02723, l=0.324, cg_ratio=0.5, h=0.07, b=0.281, fr=0.012,
                   kd=0.0, kb=0.5, cd=0.0, Af=1.0, rho=1.2, cl_f=0.0, cl_r=0.0,
                   mu=0.9, Bf=5.0, Cf=2.28, Br=5.0, Cr=2.28,
                   Fd_max=15.0, Fb_max=-15.0, Td=0.1, Tb=0.1,
                   max_steer=0.314159, max_steer_rate=10.0)


def iac_vehicle() -> Vehicle:
    return Vehicle(m=811.9303, Jzz=700.0, l=2.9718, cg_ratio=0.45, h=0.35, b=2.0, fr=0.012,
                   kd=0.0, kb=0.54, cd=1.0, Af=1.0, rho=1.2, cl_f=1.0, cl_r=1.0,
                   mu=1.3, Bf=11.0, Cf=1.7, Br=11.0, Cr=1.7,
                   Fd_max=10000.0, Fb_max=-20000.0, Td=0.1, Tb=0.1,
                   max_steer=0.314159, max_steer_rate=0.66)


def barc_tracking_mpc(N: int = 20) -> MPCConfig:
    return MPCConfig(N=N, margin=0.1, q_contour=1.0, q_heading=1.0, q_vel=0.2, q_vy=1e-3,
                     q_vyaw=1e-3, q_boundary=20.0,
                     R=np.diag([0.01, 0.01]), R_d=np.diag([0.01, 0.01]),
                     x_max=np.array([INF, INF, INF, 6.0, 1.0, 3.0]),
                     x_min=np.array([-INF, -INF, -INF, 0.1, -1.0, -3.0]),
                     u_max=np.array([0.01, 0.33]), u_min=np.array([-0.01, -0.33]))


def barc_lmpc(N: int = 20, n_laps: int = 3) -> MPCConfig:
    return MPCConfig(N=N, margin=0.1, q_contour=1.0, q_heading=1.0, q_vel=0.2, q_vy=1e-3,
                     q_vyaw=1e-3, q_boundary=1000.0,
                     R=np.diag([0.1, 0.1]), R_d=np.diag([0.1, 0.1]),
                     x_max=np.array([INF, INF, INF, 3.0, 1.0, 3.0]),
                     x_min=np.array([-INF, -INF, -INF, 0.1, -1.0, -3.0]),
                     u_max=np.array([0.01, 0.33]), u_min=np.array([-0.01, -0.33]),
                     learning=True,
                     convex_hull_slack=np.array([40.0, 40.0, 4.0, 40.0, 40.0, 4.0]),
                     num_ss_pts=32 * n_laps, num_ss_pts_per_lap=32, max_lap_stored=n_laps)


def iac_tracking_mpc(N: int = 40) -> MPCConfig:
    return MPCConfig(N=N, margin=0.5, q_contour=1.0, q_heading=1.0, q_vel=0.2, q_vy=0.01,
                     q_vyaw=0.01, q_boundary=20.0,
                     R=np.diag([1e-5, 1.0]), R_d=np.diag([1e-4, 10.0]),
                     x_max=np.array([INF, INF, INF, 100.0, 15.0, 2.0]),
                     x_min=np.array([-INF, -INF, -INF, 3.0, -15.0, -2.0]),
                     u_max=np.array([5.0, 0.314159]), u_min=np.array([-10.0, -0.314159]))


def iac_lmpc(N: int = 60, n_laps: int = 3) -> MPCConfig:
    """iac_car_lmpc.param.yaml (ships n = 60)."""
    return iac_tracking_mpc(N).with_(learning=True, R=np.diag([1e-4, 1e-3]), R_d=np.diag([5e-4, 1e-1]),
                                     convex_hull_slack=np.array([200.0, 20.0, 2.0, 200.0, 2.0, 20.0]),
                                     num_ss_pts=32 * n_laps, num_ss_pts_per_lap=32, max_lap_stored=n_laps)
