"""CPU restatement (test infrastructure) of the reference's track interpolants, racing_trajectory.cpp:25-120.

The arithmetic lives in a third-party dependency that is not vendored: CasADi (>= 3.6.3, branch `main` unpinned
upstream) `interpolant("bspline", grid, values)`, whose default algorithm "not_a_knot" is the interpolating cubic
B-spline with not-a-knot end conditions.  scipy.interpolate.make_interp_spline(k=3) (default boundary conditions:
not-a-knot) is an independent implementation of that published algorithm and stands in for it here.  Parity
unpinned: the reference's test of this class (test_racing_trajectory.cpp) only prints."""
from __future__ import annotations

import numpy as np
from scipy.interpolate import make_interp_spline

from .dynamics import align_abscissa

PX, PY, SPEED, S_BWD, S_FWD, LBX, LBY, RBX, RBY = 0, 1, 4, 6, 7, 9, 10, 11, 12


class TrackOracle:
    def __init__(self, table: np.ndarray):
        tab = np.asarray(table, dtype=np.float64)
        self.L = float(tab[0, S_FWD])
        ext = np.concatenate([tab, tab[:4]], axis=0)              # :48-53
        ext[-4:, S_BWD] += self.L
        ext = np.concatenate([ext[-7:-4], ext], axis=0)            # :56-59
        ext[:3, S_BWD] -= self.L
        s = ext[:, S_BWD]
        self.sx, self.sy = make_interp_spline(s, ext[:, PX], k=3), make_interp_spline(s, ext[:, PY], k=3)
        self.sv = make_interp_spline(s, ext[:, SPEED], k=3)
        self.sl = make_interp_spline(s, np.hypot(ext[:, PX] - ext[:, LBX], ext[:, PY] - ext[:, LBY]), k=3)
        self.sr = make_interp_spline(s, -np.hypot(ext[:, PX] - ext[:, RBX], ext[:, PY] - ext[:, RBY]), k=3)

    def mod(self, s):
        return align_abscissa(np.asarray(s, dtype=np.float64), self.L / 2.0, self.L)

    def eval(self, s) -> dict:
        sm = self.mod(s)
        dx, dy, d2x, d2y = self.sx(sm, 1), self.sy(sm, 1), self.sx(sm, 2), self.sy(sm, 2)
        return {"x": self.sx(sm), "y": self.sy(sm), "vel": self.sv(sm), "left": self.sl(sm), "right": self.sr(sm),
                "yaw": np.arctan2(dy, dx), "curvature": dx * d2y - dy * d2x / np.sqrt((dx ** 2 + dy ** 2) ** 3)}
