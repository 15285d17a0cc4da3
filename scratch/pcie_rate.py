"""Rates with the inputs coming from and the results going back to HOST memory (never bench.py's `value`)."""
import sys, time, importlib, ctypes as C
import numpy as np, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("racing-lmpc-ros2_amd")
dev = torch.device("cuda", 0)
B, N = 4096, 20
tr = pkg.workloads.synthetic_track("barc")
s = pkg.Solver(pkg.presets.barc_tracking_mpc(N), pkg.presets.barc_vehicle(), device=0)
x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], [-0.015, -0.314159], [0.015, 0.314159], seed=0)
inp = s.prepare(tr, x.T.copy(), 0.025); inp["u_ic"] = torch.as_tensor(u.T.copy(), dtype=torch.float64, device=dev)
keys = ("x_ic", "u_ic", "X_ref", "U_ref", "T_ref", "bound_left", "bound_right", "curvatures", "vel_ref")
host_in = {k: inp[k].cpu().pin_memory() for k in keys}
dev_in = {k: torch.empty_like(inp[k]) for k in keys}; dev_in["L"] = inp["L"]
out = s.alloc_outputs(B)
okeys = ("X_optm", "U_optm", "dU_optm", "status", "iters")
host_out = {k: torch.empty_like(out[k], device="cpu").pin_memory() for k in okeys}
def step():
    for k in keys: dev_in[k].copy_(host_in[k], non_blocking=True)
    s.solve(dev_in, out)
    for k in okeys: host_out[k].copy_(out[k], non_blocking=True)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 50
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
nbytes = sum(host_in[k].numel() * 8 for k in keys) + sum(host_out[k].numel() * host_out[k].element_size() for k in okeys)
print(f"batch {B}: host->device inputs, solve, device->host results, one stream, pinned memory: {dt*1e3:.3f} ms per batch = {B/dt/1e6:.2f} M solves/s ({nbytes/1e6:.1f} MB over PCIe per batch)")
# one problem through lmpc_solve_host (column-major host pointers, the facade's path)
lib = s.lib
g = {k: np.ascontiguousarray(inp[k][..., 0].cpu().numpy()) for k in keys}
X_ref = np.asfortranarray(g["X_ref"]); U_ref = np.asfortranarray(g["U_ref"])
Xo = np.zeros((6, N), order="F"); Uo = np.zeros((2, N - 1), order="F"); dUo = np.zeros((2, N - 1), order="F")
st, it = C.c_int32(0), C.c_int32(0)
p = lambda a: a.ctypes.data_as(C.c_void_p)
def one():
    rc = lib.lmpc_solve_host(s._h, p(g["x_ic"]), p(g["u_ic"]), p(X_ref), p(U_ref), p(g["T_ref"]), p(g["bound_left"]), p(g["bound_right"]),
                             p(g["curvatures"]), p(g["vel_ref"]), C.c_double(float(inp["L"])), None, None, p(Xo), p(Uo), p(dUo), None, C.byref(st), C.byref(it))
    assert rc == 0 and st.value == 0, (rc, st.value)
for _ in range(20): one()
t0 = time.perf_counter()
for _ in range(300): one()
dt1 = (time.perf_counter() - t0) / 300
print(f"one problem, lmpc_solve_host (pageable host pointers in, host pointers out, synchronous): {dt1*1e3:.3f} ms per call, {it.value} iterations")
