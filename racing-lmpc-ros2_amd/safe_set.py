"""Host side of the LMPC safe set, Python mirror of the reference's classes
(src/vehicle_dynamics_models/racing_trajectory/src/safe_set.cpp):

  SafeSetManager.add_lap      :144-151  ring of max_lap_stored laps (boost::circular_buffer, :139-142)
  SafeSetRecorder.load / step :260-322  lap files and lap segmentation on the abscissa wrap

The laps live on the host; `SafeSetManager.sync` mirrors them into a Solver's device store (lmpc_set_safe_set, and
lmpc_set_regression_laps when a regression spec is given), where the queries run.  Lap files are the reference's:
`<prefix>_{x,u,k,t}.txt`, whitespace text, one sample per row (DM::to_file(..., "txt") of the transpose)."""
from __future__ import annotations

from collections import deque
from pathlib import Path

import numpy as np


def read_txt(path) -> np.ndarray:
    """DM::from_file(path, "txt"): rows = samples.  Always 2-D (a one-column file gives [n, 1])."""
    a = np.loadtxt(path, dtype=np.float64, ndmin=2)
    return a


def write_txt(a, path) -> None:
    np.savetxt(path, np.atleast_2d(np.asarray(a, dtype=np.float64)), fmt="% .16e", delimiter="  ")


class SafeSetManager:
    def __init__(self, max_lap_stored: int):
        self.laps = deque(maxlen=max_lap_stored if max_lap_stored > 0 else None)
        self.total_length = None

    def add_lap(self, x, u, k, t, total_length: float):
        """x [n, 6], u [n, 2], k [n], t [n] (one sample per row, as the lap files store them)."""
        x = np.asarray(x, dtype=np.float64).reshape(-1, 6)
        n = x.shape[0]
        self.laps.append((x, np.asarray(u, dtype=np.float64).reshape(n, 2), np.asarray(k, dtype=np.float64).reshape(n),
                          np.asarray(t, dtype=np.float64).reshape(n)))
        self.total_length = float(total_length)

    def sync(self, solver, regression: dict | None = None):
        """Upload the stored laps to the solver's device store (oldest first, as lmpc_set_safe_set expects)."""
        solver.set_safe_set([l[0] for l in self.laps], self.total_length)
        if regression is not None:
            solver.set_regression_laps(list(self.laps), **regression)


class SafeSetRecorder:
    def __init__(self, manager: SafeSetManager, to_file: bool = False, file_prefix: str = ""):
        self.manager, self.to_file, self.file_prefix = manager, to_file, file_prefix
        self.last_x_valid = self.initialized = False
        self.lap_count = 0
        self.x, self.u, self.k, self.t = [], [], [], []

    def load(self, from_files, total_length: float):
        for name in from_files:
            try:
                x, u, k, t = (read_txt(f"{name}_{s}.txt") for s in ("x", "u", "k", "t"))
                self.manager.add_lap(x, u, k[:, 0], t[:, 0], total_length)
                self.lap_count += 1
            except Exception as e:  # the reference prints and carries on (safe_set.cpp:271-275)
                print(f"Failed to load lap from {name}\n{e}")

    def step(self, x, u, k, t, total_length: float) -> bool:
        """One sample per control step; returns True when a completed lap was handed to the manager."""
        x = np.asarray(x, dtype=np.float64).reshape(6)
        if not self.last_x_valid:  # the very first sample only seeds the abscissa (safe_set.cpp:278-282)
            self.x = [x]
            self.last_x_valid = True
            return False
        added = False
        if self.x[-1][0] - x[0] > 0.5 * total_length:  # crossed the start line
            if self.initialized:
                lap = (np.stack(self.x), np.stack(self.u), np.array(self.k), np.array(self.t))
                self.manager.add_lap(*lap, total_length)
                if self.to_file:
                    name = f"{self.file_prefix}lap_{self.lap_count}"
                    for a, s in zip(lap, ("x", "u", "k", "t")):
                        write_txt(a if a.ndim == 2 else a[:, None], f"{name}_{s}.txt")
                added = True
            else:
                self.initialized = True  # the first, partial lap is discarded
            self.lap_count += 1
            self.x, self.u, self.k, self.t = [x], [np.asarray(u, dtype=np.float64).reshape(2)], [float(k)], [float(t)]
        else:
            self.x.append(x)
            self.u.append(np.asarray(u, dtype=np.float64).reshape(2))
            self.k.append(float(k))
            self.t.append(float(t))
        return added
