"""Every instantiation the dispatch table can select, every N from 3 to 81, >= 1024 problems each, against the twin -- in the product
library AND in the debug-hook build (tests/dispatch_sweep.py does the work, one subprocess per library so
that LMPC_HIP_LIBRARY selects the build).  This is the guard behind the compiler-sensitive instantiations of DESIGN.md section 4
that does not depend on code shape: whatever a later edit does to register pressure, a miscomputing (N, problem, precision) fails
here by name."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
LIB = ROOT / "racing-lmpc-ros2_amd" / "lib"


def _sweep(lib, *args):
    assert lib.exists(), "%s not built: __graft_entry__.build()" % lib.name
    env = dict(os.environ, LMPC_HIP_LIBRARY=str(lib))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "dispatch_sweep.py"), *args], capture_output=True, text=True, timeout=1500, env=env)
    lines = r.stdout.strip().splitlines()
    assert lines, r.stderr[-3000:]
    summary = json.loads(lines[-1]) if lines[-1].startswith("{") else None
    marked = [ln for ln in lines if "<--" in ln]
    print("\n".join(lines[-1:] + marked[:20]))
    assert r.returncode == 0 and summary is not None and not summary["failures"], (marked[:20], r.stderr[-2000:])
    return summary


def test_every_instantiation_against_the_twin_product_library():
    s = _sweep(LIB / "liblmpc_hip.so", "--problems", "1024")
    assert s["cases"] == 4 * 79 and s["library"] == "liblmpc_hip.so"


def test_every_instantiation_against_the_twin_debug_build():
    """the same sources with LMPC_DEBUG_HOOKS compiled in: a second register allocation of every kernel"""
    s = _sweep(LIB / "liblmpc_hip_dbg.so", "--problems", "1024")
    assert s["cases"] == 4 * 79 and s["library"] == "liblmpc_hip_dbg.so"
