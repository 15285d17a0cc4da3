"""Track tables and their interpolants: host-side mirror of the reference's RacingTrajectory
(src/vehicle_dynamics_models/racing_trajectory/src/racing_trajectory.cpp:25-236, column enum
include/racing_trajectory/racing_trajectory.hpp:37-56) -- SURVEY.md 8(f) rank 3.

A track is a 17-column whitespace table, one waypoint per row.  Only PX, PY, SPEED, DIST_TO_SF_BWD (the abscissa),
DIST_TO_SF_FWD[0] (the lap length) and the four boundary columns are consumed (racing_trajectory.cpp:27-29,64-94).
The reference closes the loop by appending the first four waypoints (+L) and prepending the last three (-L)
(:48-59), fits `casadi::interpolant("bspline", ...)` through (abscissa, value) -- an interpolating cubic spline with
not-a-knot end conditions -- and derives yaw = atan2(y', x') and the curvature expression *as written*,
    x' y'' - y' x'' / sqrt((x'^2 + y'^2)^3)                                                           (:108-110)
(the division binds to the second product only).  Every interpolant is evaluated at align_abscissa(s, L/2, L)
(:98), i.e. the abscissa wrapped into [0, L).

The device kernels consume uniform periodic tables (`lmpc_track`); `RacingTrajectory.to_track_table` samples them.
The spline here is hand-written (banded not-a-knot system -> piecewise cubics); tests compare it with scipy's
make_interp_spline, which implements the same published algorithm (oracle/trajectory.py)."""
from __future__ import annotations

import numpy as np

PX, PY, PZ, YAW, SPEED, CURVATURE, DIST_TO_SF_BWD, DIST_TO_SF_FWD, REGION = range(9)
LEFT_BOUND_X, LEFT_BOUND_Y, RIGHT_BOUND_X, RIGHT_BOUND_Y, BANK, LON_ACC, LAT_ACC, TIME = range(9, 17)


def align_abscissa(s1, s2, s_total):
    """lmpc_utils/utils.hpp:35-41."""
    s1 = np.asarray(s1, dtype=np.float64)
    k = np.abs(s2 - s1) + s_total / 2.0
    return s1 + (k - np.fmod(k, s_total)) * np.sign(s2 - s1)


def align_yaw(yaw_1, yaw_2):
    """lmpc_utils/utils.hpp:25-31."""
    d = np.asarray(yaw_1, dtype=np.float64) - yaw_2
    return np.arctan2(np.sin(d), np.cos(d)) + yaw_2


class NotAKnotCubic:
    """Interpolating C2 cubic spline through (x_i, y_i) with not-a-knot end conditions, as piecewise cubics
    y(x) = a_i + b_i h + c_i h^2 + d_i h^3 on [x_i, x_{i+1}], h = x - x_i (extrapolated with the end pieces)."""

    def __init__(self, x, y):
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        n = x.size
        if n < 4 or not (np.diff(x) > 0).all():
            raise ValueError("need >= 4 strictly increasing abscissae")
        h = np.diff(x)
        dlt = np.diff(y) / h
        # unknowns: second derivatives m_i; interior rows are the C1 conditions, the end rows the not-a-knot
        # conditions (third derivative continuous across x_1 and x_{n-2})
        A = np.zeros((n, n))
        r = np.zeros(n)
        for i in range(1, n - 1):
            A[i, i - 1], A[i, i], A[i, i + 1] = h[i - 1], 2.0 * (h[i - 1] + h[i]), h[i]
            r[i] = 6.0 * (dlt[i] - dlt[i - 1])
        A[0, 0], A[0, 1], A[0, 2] = h[1], -(h[0] + h[1]), h[0]
        A[-1, -3], A[-1, -2], A[-1, -1] = h[-1], -(h[-2] + h[-1]), h[-2]
        m = np.linalg.solve(A, r)
        self.x = x
        self.a = y[:-1]
        self.b = dlt - h * (2.0 * m[:-1] + m[1:]) / 6.0
        self.c = m[:-1] / 2.0
        self.d = (m[1:] - m[:-1]) / (6.0 * h)

    def __call__(self, xq, nu: int = 0):
        xq = np.asarray(xq, dtype=np.float64)
        i = np.clip(np.searchsorted(self.x, xq, side="right") - 1, 0, self.x.size - 2)
        h = xq - self.x[i]
        a, b, c, d = self.a[i], self.b[i], self.c[i], self.d[i]
        if nu == 0:
            return a + h * (b + h * (c + h * d))
        if nu == 1:
            return b + h * (2.0 * c + 3.0 * h * d)
        if nu == 2:
            return 2.0 * c + 6.0 * h * d
        raise ValueError("nu in 0..2")


class RacingTrajectory:
    def __init__(self, traj):
        """traj: path of a track file, or the [n, 17] table (one waypoint per row)."""
        tab = np.loadtxt(traj, dtype=np.float64, ndmin=2) if isinstance(traj, (str, bytes)) or hasattr(traj, "__fspath__") \
            else np.asarray(traj, dtype=np.float64)
        if tab.ndim != 2 or tab.shape[1] != 17 or tab.shape[0] < 8:
            raise ValueError("a track table has 17 columns and at least 8 waypoints")
        self.table = tab
        self.total_length = float(tab[0, DIST_TO_SF_FWD])
        L = self.total_length
        ext = np.concatenate([tab, tab[:4]], axis=0)            # racing_trajectory.cpp:48-53
        ext[-4:, DIST_TO_SF_BWD] += L
        ext = np.concatenate([ext[-7:-4], ext], axis=0)          # :56-59 (the last three original waypoints)
        ext[:3, DIST_TO_SF_BWD] -= L
        s = ext[:, DIST_TO_SF_BWD]
        left = np.hypot(ext[:, PX] - ext[:, LEFT_BOUND_X], ext[:, PY] - ext[:, LEFT_BOUND_Y])
        right = -np.hypot(ext[:, PX] - ext[:, RIGHT_BOUND_X], ext[:, PY] - ext[:, RIGHT_BOUND_Y])
        self._x, self._y = NotAKnotCubic(s, ext[:, PX]), NotAKnotCubic(s, ext[:, PY])
        self._vel, self._left, self._right = NotAKnotCubic(s, ext[:, SPEED]), NotAKnotCubic(s, left), NotAKnotCubic(s, right)

    # ---- interpolants, all at the wrapped abscissa (racing_trajectory.cpp:98,112-118) ----
    def _mod(self, s):
        return align_abscissa(s, self.total_length / 2.0, self.total_length)

    def x(self, s):
        return self._x(self._mod(s))

    def y(self, s):
        return self._y(self._mod(s))

    def velocity(self, s):
        return self._vel(self._mod(s))

    def left_boundary(self, s):
        return self._left(self._mod(s))

    def right_boundary(self, s):
        return self._right(self._mod(s))

    def yaw(self, s):
        sm = self._mod(s)
        return np.arctan2(self._y(sm, 1), self._x(sm, 1))

    def curvature(self, s):
        sm = self._mod(s)
        dx, dy, d2x, d2y = self._x(sm, 1), self._y(sm, 1), self._x(sm, 2), self._y(sm, 2)
        return dx * d2y - dy * d2x / np.sqrt((dx ** 2 + dy ** 2) ** 3)   # as written (:108-110)

    # ---- Frenet <-> global (racing_trajectory.cpp:122-186, 204-236) ----
    def frenet_to_global(self, s, t, xi):
        yaw0 = self.yaw(s)
        return self.x(s) - np.sin(yaw0) * t, self.y(s) + np.cos(yaw0) * t, align_yaw(yaw0 + xi, 0.0)

    def global_to_frenet(self, x, y, phi, s0=None):
        """Projection of one pose: minimise the squared distance to the centre line over the abscissa, started from
        `s0` or from the closest waypoint (the kd-tree lookup upstream, :213-216).  The reference hands the scalar
        problem to CasADi's sqpmethod; here it is a safeguarded Newton iteration on the same objective."""
        L = self.total_length
        if s0 is None:
            i = int(np.argmin((self.table[:, PX] - x) ** 2 + (self.table[:, PY] - y) ** 2))
            s0 = self.table[i, DIST_TO_SF_BWD]
        s = float(self._mod(s0))

        def f(sv):
            return (self._x(sv) - x) ** 2 + (self._y(sv) - y) ** 2

        for _ in range(50):
            ex, ey = self._x(s) - x, self._y(s) - y
            dx, dy, d2x, d2y = self._x(s, 1), self._y(s, 1), self._x(s, 2), self._y(s, 2)
            g = 2.0 * (ex * dx + ey * dy)
            H = 2.0 * (dx * dx + dy * dy + ex * d2x + ey * d2y)
            step = -g / H if H > 1e-12 else -g / (2.0 * (dx * dx + dy * dy))
            f0, a = f(s), 1.0
            while f(s + a * step) > f0 and a > 1e-6:
                a *= 0.5
            s += a * step
            if abs(a * step) < 1e-12:
                break
        s_out = float(self._mod(s))
        xo, yo, yaw_o = float(self._x(s_out)), float(self._y(s_out)), float(np.arctan2(self._y(s_out, 1), self._x(s_out, 1)))
        sign = np.sign(np.cos(yaw_o) * (y - yo) - np.sin(yaw_o) * (x - xo))     # lateral_sign (utils.hpp)
        return s_out, float(np.hypot(x - xo, y - yo) * sign), float(align_yaw(phi, yaw_o) - yaw_o)

    # ---- what the device kernels consume ----
    def to_track_table(self, M: int = 1024) -> dict:
        """Uniform periodic tables (lmpc_track): curvature, signed boundary offsets and speed at s_j = j L / M."""
        s = np.arange(M) * self.total_length / M
        return {"L": self.total_length, "M": M, "curvature": self.curvature(s), "bound_left": self.left_boundary(s),
                "bound_right": self.right_boundary(s), "vel": self.velocity(s)}
