"""racing-lmpc-ros2_amd -- MI355X-native batched LMPC solve path (host-side Python mirror).

Thin ctypes layer over the C ABI in include/lmpc_hip.h (lib/liblmpc_hip.so, hand-written HIP
for gfx950).  PyTorch is used only for device memory, streams and torch.distributed; all
compute happens in the HIP kernels.  There is no CPU fallback: importing `capi` without the
built library raises.
"""
from . import closed_loop, presets, racing_trajectory, ros_params, safe_set, workloads  # noqa: F401
from .capi import (LmpcError, Solver, SOLVE_INFEASIBLE, SOLVE_MAX_ITER, SOLVE_OPTIMAL,  # noqa: F401
                   library_path, load_library)

__all__ = ["presets", "ros_params", "workloads", "closed_loop", "safe_set", "racing_trajectory", "Solver", "LmpcError", "load_library", "library_path",
           "SOLVE_OPTIMAL", "SOLVE_MAX_ITER", "SOLVE_INFEASIBLE"]
