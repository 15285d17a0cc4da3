import sys, os, ctypes, pickle, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import r3_f32_eval as E
from oracle import cbind, params as P
kind, N, B, seed, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cfg, veh, inp, ss_x, ss_j = E.build(kind, N, B, seed)
ref = pickle.load(open(f"/tmp/ref64_{kind}_{N}_{B}_{seed}.pkl", "rb"))
cbind._LIB = ctypes.CDLL(os.environ.get("TWIN_LIB", "/tmp/liboracle_f32.so"))
o = cbind.solve_batch(cfg, veh, inp, ss_x=ss_x, ss_j=ss_j, b0=b, b1=b + 1)
ex = np.abs((o["X_optm"][:, :, b] - ref["X_optm"][:, :, b]) / P.SCALE_X[:, None]); eu = np.abs((o["U_optm"][:, :, b] - ref["U_optm"][:, :, b]) / P.SCALE_U[:, None])
print("status", o["status"][b], "iters", o["iters"][b], "ref iters", ref["iters"][b], "err X", ex.max(), np.unravel_index(ex.argmax(), ex.shape), "err U", eu.max(), np.unravel_index(eu.argmax(), eu.shape), "kkt", o["kkt"][:, b])
print("x_ic", inp["x_ic"][:, b])
