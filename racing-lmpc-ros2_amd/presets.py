"""Shipped parameter sets of the reference, as plain dicts for the C ABI structs.

Transcribed from the reference's YAML (values only):
  vehicles  src/launch/racing_lmpc_launch/param/{barc,iac_car}/*_base.param.yaml, *_single_track.param.yaml
  MPC       src/launch/racing_lmpc_launch/param/racing_mpc/{barc_tracking_mpc,barc_lmpc,iac_car_tracking_mpc}.param.yaml
Field names are those of lmpc_vehicle / lmpc_config in include/lmpc_hip.h.
"""
from __future__ import annotations

INF = float("inf")


def barc_vehicle() -> dict:
    return dict(model_id=0, integrator="rk4", m=2.2187, Jzz=0.02723, l=0.324, cg_ratio=0.5, h=0.07, b=0.281, fr=0.012,
                kd=0.0, kb=0.5, cd=0.0, Af=1.0, rho=1.2, cl_f=0.0, cl_r=0.0, mu=0.9,
                Bf=5.0, Cf=2.28, Br=5.0, Cr=2.28, Fd_max=15.0, Fb_max=-15.0, Td=0.1, Tb=0.1,
                max_steer=0.314159, max_steer_rate=10.0)


def iac_vehicle() -> dict:
    return dict(model_id=0, integrator="rk4", m=811.9303, Jzz=700.0, l=2.9718, cg_ratio=0.45, h=0.35, b=2.0, fr=0.012,
                kd=0.0, kb=0.54, cd=1.0, Af=1.0, rho=1.2, cl_f=1.0, cl_r=1.0, mu=1.3,
                Bf=11.0, Cf=1.7, Br=11.0, Cr=1.7, Fd_max=10000.0, Fb_max=-20000.0, Td=0.1, Tb=0.1,
                max_steer=0.314159, max_steer_rate=0.66)


def barc_tracking_mpc(N: int = 20) -> dict:
    return dict(N=N, learning=0, num_ss_pts=96, num_ss_pts_per_lap=32, max_lap_stored=3, max_iter=0, tol=0.0,
                margin=0.1, q_contour=1.0, q_heading=1.0, q_vel=0.2, q_vy=1e-3, q_vyaw=1e-3, q_boundary=20.0,
                R=[0.01, 0.0, 0.0, 0.01], R_d=[0.01, 0.0, 0.0, 0.01],
                x_max=[INF, INF, INF, 6.0, 1.0, 3.0], x_min=[-INF, -INF, -INF, 0.1, -1.0, -3.0],
                u_max=[0.01, 0.33], u_min=[-0.01, -0.33],
                convex_hull_slack=[20.0, 20.0, 2.0, 20.0, 20.0, 2.0], max_vel_ref_diff=1.0)


def barc_lmpc(N: int = 20, n_laps: int = 3) -> dict:
    c = barc_tracking_mpc(N)
    c.update(learning=1, q_boundary=1000.0, R=[0.1, 0.0, 0.0, 0.1], R_d=[0.1, 0.0, 0.0, 0.1],
             x_max=[INF, INF, INF, 3.0, 1.0, 3.0], convex_hull_slack=[40.0, 40.0, 4.0, 40.0, 40.0, 4.0],
             num_ss_pts=32 * n_laps, num_ss_pts_per_lap=32, max_lap_stored=n_laps)
    return c


def iac_tracking_mpc(N: int = 40) -> dict:
    return dict(N=N, learning=0, num_ss_pts=96, num_ss_pts_per_lap=32, max_lap_stored=3, max_iter=0, tol=0.0,
                margin=0.5, q_contour=1.0, q_heading=1.0, q_vel=0.2, q_vy=0.01, q_vyaw=0.01, q_boundary=20.0,
                R=[1e-5, 0.0, 0.0, 1.0], R_d=[1e-4, 0.0, 0.0, 10.0],
                x_max=[INF, INF, INF, 100.0, 15.0, 2.0], x_min=[-INF, -INF, -INF, 3.0, -15.0, -2.0],
                u_max=[5.0, 0.314159], u_min=[-10.0, -0.314159],
                convex_hull_slack=[20.0, 20.0, 2.0, 20.0, 20.0, 2.0], max_vel_ref_diff=1.0)


def iac_lmpc(N: int = 60, n_laps: int = 3) -> dict:
    """iac_car_lmpc.param.yaml (n = 60): the IAC learning controller."""
    c = iac_tracking_mpc(N)
    c.update(learning=1, R=[1e-4, 0.0, 0.0, 1e-3], R_d=[5e-4, 0.0, 0.0, 1e-1],
             convex_hull_slack=[200.0, 20.0, 2.0, 200.0, 2.0, 20.0],
             num_ss_pts=32 * n_laps, num_ss_pts_per_lap=32, max_lap_stored=n_laps)
    return c
