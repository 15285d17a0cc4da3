"""Single-track planar model, RK4 and its exact Jacobian (oracle; test infrastructure).

Restates, in numpy (vectorised over a leading batch axis, complex-safe):
  * SingleTrackPlanarModel::compile_dynamics
      src/vehicle_dynamics_models/single_track_planar_model/src/single_track_planar_model.cpp:195-332
    (simplify_lon_control=true, use_frenet=true -- every shipped vehicle YAML);
  * lmpc::utils::rk4_function  src/tools/lmpc_utils/src/utils.cpp:88-108;
  * the discrete Jacobian outputs A, B, g of
      single_track_planar_model.cpp:377-387  (A = d xip1/dx, B = d xip1/du,
      g = xip1 - A x - B u).

CasADi obtains A and B by symbolic differentiation of the RK4 expression; the
oracle's reference derivative is complex-step differentiation of the same RK4
map (exact to rounding for analytic functions), and an independent hand-derived
forward-mode Jacobian is provided and checked against it.

State  x = [s, e_y, e_psi, vx, vy, omega]   (base_vehicle_model.hpp:32-40)
Input  u = [u_lon, steer]                   (single_track_planar_model.hpp:62-66)
"""
from __future__ import annotations

import numpy as np

from .params import Vehicle

GRAVITY = 9.8  # single_track_planar_model.cpp:18


def f_continuous(x, u, k, v: Vehicle):
    """x_dot = f(x, u, k).  x: (..., 6), u: (..., 2), k: (...,).  Complex-safe."""
    ey = x[..., 1]
    phi = x[..., 2]
    vx = x[..., 3]
    vy = x[..., 4]
    om = x[..., 5]
    ul = u[..., 0]
    delta = u[..., 1]

    # :215-216
    fd = ul * (np.tanh(ul) * 0.5 + 0.5) * 1000.0
    fb = ul * (np.tanh(-ul) * 0.5 + 0.5) * 1000.0
    m, l = v.m, v.l
    lr = v.cg_ratio * l  # :229
    lf = l - lr          # :230
    v_sq = vx * vx       # :210

    # :258-262
    Fx_f = 0.5 * v.kd * fd + 0.5 * v.kb * fb - 0.5 * v.fr * m * GRAVITY * lr / l
    Fx_r = 0.5 * (1 - v.kd) * fd + 0.5 * (1.0 - v.kb) * fb - 0.5 * v.fr * m * GRAVITY * lf / l
    # :267 (no air density in this term, as written upstream)
    ax = (fd + fb - 0.5 * v.cd * v.Af * v_sq - v.fr * m * GRAVITY) / m
    # :270-275
    Fz_f = 0.5 * m * GRAVITY * lr / (lf + lr) - 0.5 * v.h / (lf + lr) * m * ax + 0.25 * v.cl_f * v.rho * v.Af * v_sq
    Fz_r = 0.5 * m * GRAVITY * lf / (lf + lr) + 0.5 * v.h / (lf + lr) * m * ax + 0.25 * v.cl_r * v.rho * v.Af * v_sq
    # :280-283
    a_f = delta - np.arctan((lf * om + vy) / (vx + 1e-3))
    a_r = np.arctan((lr * om - vy) / (vx + 1e-3))
    # :299-300
    Fy_f = v.mu * Fz_f * np.sin(v.Cf * np.arctan(v.Bf * a_f))
    Fy_r = v.mu * Fz_r * np.sin(v.Cr * np.arctan(v.Br * a_r))
    cd_, sd_ = np.cos(delta), np.sin(delta)
    # :309-319
    om_dot = 1.0 / v.Jzz * (-(2 * Fy_r) * lr + ((2 * Fy_f) * cd_ + (2 * Fx_f) * sd_) * lf)
    vx_dot = 1.0 / m * ((2 * Fx_r) + (2 * Fx_f) * cd_ - (2 * Fy_f) * sd_ - 0.5 * v.cd * v.rho * v.Af * v_sq) + om * vy
    vy_dot = 1.0 / m * ((2 * Fy_r) + (2 * Fy_f) * cd_ + (2 * Fx_f) * sd_) - om * vx
    # :322-330 (Frenet)
    px_dot = (vx * np.cos(phi) - vy * np.sin(phi)) / (1 - ey * k)
    py_dot = vx * np.sin(phi) + vy * np.cos(phi)
    phi_dot = om - k * px_dot
    return np.stack([px_dot, py_dot, phi_dot, vx_dot, vy_dot, om_dot], axis=-1)


def rk4(x, u, k, dt, v: Vehicle):
    """The model's discrete dynamics: utils.cpp:88-108 -- classic RK4 with u, k held over the step -- or, for a vehicle
    with integrator = "euler" (modeling.integrator_type, single_track_planar_model.cpp:357-368), x + dt f (utils.cpp:110-123)."""
    dt_ = np.asarray(dt)[..., None] if np.ndim(dt) else dt
    k1 = f_continuous(x, u, k, v)
    if getattr(v, "integrator", "rk4") == "euler":
        return x + dt_ * k1
    k2 = f_continuous(x + dt_ / 2.0 * k1, u, k, v)
    k3 = f_continuous(x + dt_ / 2.0 * k2, u, k, v)
    k4 = f_continuous(x + dt_ * k3, u, k, v)
    return x + dt_ / 6 * (k1 + 2 * k2 + 2 * k3 + k4)


def rk4_jacobian_cs(x, u, k, dt, v: Vehicle, h: float = 1e-30):
    """(A, B, g) by complex-step differentiation of the RK4 map.

    x: (..., 6) real, u: (..., 2) real.  Returns A (..., 6, 6), B (..., 6, 2), g (..., 6).
    """
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    xp = rk4(x, u, k, dt, v)
    A = np.empty(x.shape[:-1] + (6, 6))
    B = np.empty(x.shape[:-1] + (6, 2))
    for j in range(6):
        xc = x.astype(np.complex128)
        xc[..., j] += 1j * h
        A[..., :, j] = rk4(xc, u.astype(np.complex128), k, dt, v).imag / h
    for j in range(2):
        uc = u.astype(np.complex128)
        uc[..., j] += 1j * h
        B[..., :, j] = rk4(x.astype(np.complex128), uc, k, dt, v).imag / h
    g = xp - np.einsum("...ij,...j->...i", A, x) - np.einsum("...ij,...j->...i", B, u)
    return A, B, g


def f_and_partials(x, u, k, v: Vehicle):
    """Hand-derived value and partials of f: returns (f (...,6), Fx (...,6,6), Fu (...,6,2)).

    Independent of CasADi-style AD; validated against complex-step in the tests.
    """
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    ey, phi, vx, vy, om = x[..., 1], x[..., 2], x[..., 3], x[..., 4], x[..., 5]
    ul, delta = u[..., 0], u[..., 1]
    m, l = v.m, v.l
    lr = v.cg_ratio * l
    lf = l - lr
    th = np.tanh(ul)
    sech2 = 1.0 - th * th
    fd = 1000.0 * ul * (0.5 * th + 0.5)
    fb = 1000.0 * ul * (0.5 - 0.5 * th)
    dfd = 1000.0 * ((0.5 * th + 0.5) + ul * 0.5 * sech2)
    dfb = 1000.0 * ((0.5 - 0.5 * th) - ul * 0.5 * sech2)
    Fx_f = 0.5 * v.kd * fd + 0.5 * v.kb * fb - 0.5 * v.fr * m * GRAVITY * lr / l
    Fx_r = 0.5 * (1 - v.kd) * fd + 0.5 * (1 - v.kb) * fb - 0.5 * v.fr * m * GRAVITY * lf / l
    dFxf_du = 0.5 * v.kd * dfd + 0.5 * v.kb * dfb
    dFxr_du = 0.5 * (1 - v.kd) * dfd + 0.5 * (1 - v.kb) * dfb
    vsq = vx * vx
    ax = (fd + fb - 0.5 * v.cd * v.Af * vsq - v.fr * m * GRAVITY) / m
    dax_dvx = -v.cd * v.Af * vx / m
    dax_du = (dfd + dfb) / m
    hl = v.h / l
    Fz_f = 0.5 * m * GRAVITY * lr / l - 0.5 * hl * m * ax + 0.25 * v.cl_f * v.rho * v.Af * vsq
    Fz_r = 0.5 * m * GRAVITY * lf / l + 0.5 * hl * m * ax + 0.25 * v.cl_r * v.rho * v.Af * vsq
    dFzf_dvx = -0.5 * hl * m * dax_dvx + 0.5 * v.cl_f * v.rho * v.Af * vx
    dFzr_dvx = 0.5 * hl * m * dax_dvx + 0.5 * v.cl_r * v.rho * v.Af * vx
    dFzf_du = -0.5 * hl * m * dax_du
    dFzr_du = 0.5 * hl * m * dax_du
    den = vx + 1e-3
    rf = (lf * om + vy) / den
    rr = (lr * om - vy) / den
    wf = 1.0 / ((1.0 + rf * rf) * den)
    wr = 1.0 / ((1.0 + rr * rr) * den)
    a_f = delta - np.arctan(rf)
    a_r = np.arctan(rr)
    daf_dvx, daf_dvy, daf_dom = rf * wf, -wf, -lf * wf
    dar_dvx, dar_dvy, dar_dom = -rr * wr, -wr, lr * wr
    tf = np.arctan(v.Bf * a_f)
    tr = np.arctan(v.Br * a_r)
    Sf, Sr = np.sin(v.Cf * tf), np.sin(v.Cr * tr)
    Df = np.cos(v.Cf * tf) * v.Cf * v.Bf / (1.0 + (v.Bf * a_f) ** 2)
    Dr = np.cos(v.Cr * tr) * v.Cr * v.Br / (1.0 + (v.Br * a_r) ** 2)
    Fy_f = v.mu * Fz_f * Sf
    Fy_r = v.mu * Fz_r * Sr
    # partials of Fy_f, Fy_r wrt (vx, vy, om, ul, delta)
    dFyf = {
        "vx": v.mu * (dFzf_dvx * Sf + Fz_f * Df * daf_dvx),
        "vy": v.mu * Fz_f * Df * daf_dvy,
        "om": v.mu * Fz_f * Df * daf_dom,
        "ul": v.mu * dFzf_du * Sf,
        "de": v.mu * Fz_f * Df,
    }
    dFyr = {
        "vx": v.mu * (dFzr_dvx * Sr + Fz_r * Dr * dar_dvx),
        "vy": v.mu * Fz_r * Dr * dar_dvy,
        "om": v.mu * Fz_r * Dr * dar_dom,
        "ul": v.mu * dFzr_du * Sr,
        "de": 0.0 * vx,
    }
    cd_, sd_ = np.cos(delta), np.sin(delta)
    cph, sph = np.cos(phi), np.sin(phi)
    q = 1.0 / (1.0 - ey * k)
    num = vx * cph - vy * sph
    s_dot = num * q
    ey_dot = vx * sph + vy * cph
    phi_dot = om - k * s_dot
    om_dot = (-2 * Fy_r * lr + (2 * Fy_f * cd_ + 2 * Fx_f * sd_) * lf) / v.Jzz
    drag = 0.5 * v.cd * v.rho * v.Af
    vx_dot = (2 * Fx_r + 2 * Fx_f * cd_ - 2 * Fy_f * sd_ - drag * vsq) / m + om * vy
    vy_dot = (2 * Fy_r + 2 * Fy_f * cd_ + 2 * Fx_f * sd_) / m - om * vx
    f = np.stack([s_dot, ey_dot, phi_dot, vx_dot, vy_dot, om_dot], axis=-1)

    shp = x.shape[:-1]
    Fx = np.zeros(shp + (6, 6))
    Fu = np.zeros(shp + (6, 2))
    # row 0: s_dot
    Fx[..., 0, 1] = num * q * q * k
    Fx[..., 0, 2] = (-vx * sph - vy * cph) * q
    Fx[..., 0, 3] = cph * q
    Fx[..., 0, 4] = -sph * q
    # row 1: ey_dot
    Fx[..., 1, 2] = vx * cph - vy * sph
    Fx[..., 1, 3] = sph
    Fx[..., 1, 4] = cph
    # row 2: phi_dot = om - k*s_dot
    Fx[..., 2, :] = -np.asarray(k)[..., None] * Fx[..., 0, :] if np.ndim(k) else -k * Fx[..., 0, :]
    Fx[..., 2, 5] += 1.0
    # rows 3..5 wrt vx, vy, om
    for key, col in (("vx", 3), ("vy", 4), ("om", 5)):
        Fx[..., 3, col] = (-2 * dFyf[key] * sd_) / m
        Fx[..., 4, col] = (2 * dFyr[key] + 2 * dFyf[key] * cd_) / m
        Fx[..., 5, col] = (-2 * dFyr[key] * lr + 2 * dFyf[key] * cd_ * lf) / v.Jzz
    Fx[..., 3, 3] += -2 * drag * vx / m
    Fx[..., 3, 4] += om
    Fx[..., 3, 5] += vy
    Fx[..., 4, 3] += -om
    Fx[..., 4, 5] += -vx
    # wrt u_lon
    Fu[..., 3, 0] = (2 * dFxr_du + 2 * dFxf_du * cd_ - 2 * dFyf["ul"] * sd_) / m
    Fu[..., 4, 0] = (2 * dFyr["ul"] + 2 * dFyf["ul"] * cd_ + 2 * dFxf_du * sd_) / m
    Fu[..., 5, 0] = (-2 * dFyr["ul"] * lr + (2 * dFyf["ul"] * cd_ + 2 * dFxf_du * sd_) * lf) / v.Jzz
    # wrt steer
    Fu[..., 3, 1] = (-2 * Fx_f * sd_ - 2 * dFyf["de"] * sd_ - 2 * Fy_f * cd_) / m
    Fu[..., 4, 1] = (2 * dFyf["de"] * cd_ - 2 * Fy_f * sd_ + 2 * Fx_f * cd_) / m
    Fu[..., 5, 1] = ((2 * dFyf["de"] * cd_ - 2 * Fy_f * sd_ + 2 * Fx_f * cd_) * lf) / v.Jzz
    return f, Fx, Fu


def rk4_jacobian_analytic(x, u, k, dt, v: Vehicle):
    """(A, B, g, xip1) by forward-mode chain rule through the four RK4 stages."""
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    shp = x.shape[:-1]
    dt_ = np.broadcast_to(np.asarray(dt, dtype=np.float64), shp)[..., None]
    dtm = dt_[..., None]
    I6 = np.broadcast_to(np.eye(6), shp + (6, 6))
    Z62 = np.zeros(shp + (6, 2))

    def stage(xs, Xx, Xu):
        fs, Fx, Fu = f_and_partials(xs, u, k, v)
        return fs, Fx @ Xx, Fx @ Xu + Fu

    k1, K1x, K1u = stage(x, I6, Z62)
    k2, K2x, K2u = stage(x + dt_ / 2 * k1, I6 + dtm / 2 * K1x, dtm / 2 * K1u)
    k3, K3x, K3u = stage(x + dt_ / 2 * k2, I6 + dtm / 2 * K2x, dtm / 2 * K2u)
    k4, K4x, K4u = stage(x + dt_ * k3, I6 + dtm * K3x, dtm * K3u)
    xp = x + dt_ / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    A = I6 + dtm / 6 * (K1x + 2 * K2x + 2 * K3x + K4x)
    B = dtm / 6 * (K1u + 2 * K2u + 2 * K3u + K4u)
    g = xp - np.einsum("...ij,...j->...i", A, x) - np.einsum("...ij,...j->...i", B, u)
    return A, B, g, xp


def align_abscissa(s1, s2, s_total):
    """lmpc_utils/utils.hpp:35-41, as written (sign() of CasADi: sign(0) = 0)."""
    k = np.abs(s2 - s1) + s_total / 2.0
    l = k - np.fmod(np.abs(s2 - s1) + s_total / 2.0, s_total)
    return s1 + l * np.sign(s2 - s1)
