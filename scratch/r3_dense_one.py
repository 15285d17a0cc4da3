import sys, os, ctypes, pickle, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import r3_f32_eval as E
from oracle import cbind, params as P, qp as Q, scenario as S
kind, N, B, seed, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cfg, veh, inp, ss_x, ss_j = E.build(kind, N, B, seed)
ref = pickle.load(open(f"/tmp/ref64_{kind}_{N}_{B}_{seed}.pkl", "rb"))
kw = {} if ss_x is None else {"ss_x": ss_x[:, :, b], "ss_j": ss_j[:, b]}
qp = Q.build_qp(cfg, veh, S.problem(inp, b), **kw)
y, info = Q.solve_dense(qp)
ex = qp.split(y)
print("dense status", info["status"], "polished", info.get("polished"), "margin", Q.strict_complementarity(qp, y, info["lam"]))
for name, o in (("ref64", ref),):
    e = max(np.abs((o["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]).max(), np.abs((o["U_optm"][:, :, b] - ex["U_optm"]) / P.SCALE_U[:, None]).max())
    print(name, "vs dense", e, "status", o["status"][b])
cbind._LIB = ctypes.CDLL(os.environ.get("TWIN_LIB", "/tmp/liboracle_f32.so"))
o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j, b0=b, b1=b + 1)
exx = np.abs((o["X_optm"][:, :, b] - ex["X_optm"]) / P.SCALE_X[:, None]); euu = np.abs((o["U_optm"][:, :, b] - ex["U_optm"]) / P.SCALE_U[:, None])
print("f32 vs dense", exx.max(), np.unravel_index(exx.argmax(), exx.shape), euu.max(), np.unravel_index(euu.argmax(), euu.shape))
np.set_printoptions(linewidth=200, precision=5)
print("vy dense ", ex["X_optm"][4, -8:]); print("vy f32   ", o["X_optm"][4, -8:, b]); print("vy ref64 ", ref["X_optm"][4, -8:, b])
print("ulon dense", ex["U_optm"][0, -8:]); print("ulon f32 ", o["U_optm"][0, -8:, b])
print("steer dense", ex["U_optm"][1, -8:]); print("steer f32 ", o["U_optm"][1, -8:, b])
