"""warm_loop.py with a twin built with -D overrides: warm_variant.py cars steps N [-DWARM_ROUNDS=4 ...]"""
import os, sys, subprocess, ctypes, hashlib
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, here)
flags = sys.argv[4:]
from common import *
tag = hashlib.md5(" ".join(flags).encode()).hexdigest()[:8]
so = ROOT / f"scratch/r5/cache/twin_{tag}.so"
subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fPIC", "-std=c11", f"-I{ROOT}/include", "-shared", "-o", str(so), str(ROOT / "oracle/c/lmpc_oracle.c"), "-lm"] + flags)
cbind._LIB = ctypes.CDLL(str(so))
print("flags:", flags)
sys.argv = sys.argv[:4]
src = open(os.path.join(here, "warm_loop.py")).read().replace("os.path.dirname(__file__)", repr(here))
src = src.replace("it_w, it_c = np.array(it_w), np.array(it_c)", "it_w, it_c = np.array(it_w), np.array(it_c)\nprint('  cold hist', np.bincount(it_c).tolist()); print('  warm hist', np.bincount(it_w).tolist())")
exec(compile(src, "warm_loop", "exec"))
