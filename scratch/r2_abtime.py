"""A/B timing of library builds on the bench workload, alternating: usage r2_abtime.py libA libB [reps] (LMPC_N, LMPC_B)"""
import sys, os, subprocess, re
libs = sys.argv[1:3]; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B = os.environ.get("LMPC_B", "4096")
res = {l: [] for l in libs}
for k in range(reps):
    for l in libs:
        o = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "r2_g4.py"), "child", "/tmp/ab.npz", B], env=dict(os.environ, LMPC_LIB=l), capture_output=True, text=True).stdout
        m = re.search(r"qp kernel ms ([0-9.]+) \(min ([0-9.]+)\) step ms ([0-9.]+)", o)
        res[l].append(tuple(float(x) for x in m.groups()))
for l in libs:
    print(l, " ".join("%.4f/%.4f/%.4f" % t for t in res[l]))
