"""CPU prototype of the SQP globalisation (C twin as the QP solver, numpy merit function): which cold starts fail, and why."""
import sys, numpy as np, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from oracle import cbind, dynamics as D, params as P, qp as Q, scenario as S, nlp as NLP
pkg = importlib.import_module("racing-lmpc-ros2_amd")

def setup(B=96, seed=8, N=20):
    veh, cfg = P.barc_vehicle(), P.barc_tracking_mpc(N)
    tr = pkg.workloads.synthetic_track("barc")
    u_lo, u_hi, _, _ = Q.effective_bounds(cfg, veh)
    x, u = pkg.workloads.sample_initial_states("barc", B, tr["L"], u_lo, u_hi, seed)
    inp = S.cold_start_inputs(cfg, veh, tr, x, u, 0.025)
    return veh, cfg, tr, x, u, inp

def defect1(veh, inp, X, U):
    nxt = D.rk4(X[:, :-1].transpose(1, 2, 0), U.transpose(1, 2, 0), inp["curvatures"][:-1], inp["T_ref"], veh)  # (N-1, B, 6)
    c = np.abs((X[:, 1:].transpose(1, 2, 0) - nxt) / P.SCALE_X)
    return c.sum(axis=(0, 2)), c.max(axis=(0, 2))

def run(variant="base", B=96, seed=8, max_sqp=40, tol=1e-9, verbose=True, backoff=0, first_ls=False, soc=False):
    veh, cfg, tr, x, u, inp = setup(B, seed)
    N = cfg.N
    qps = [Q.build_qp(cfg, veh, S.problem(inp, b)) for b in range(B)]
    def J(b, X, U, dU):
        return NLP.merit_cost(qps[b], Q.pack(qps[b], X[:, :, b], U[:, :, b], dU[:, :, b], sigma=0.0))
    X, U, dU = inp["X_ref"].copy(), inp["U_ref"].copy(), np.zeros_like(inp["U_ref"])
    nu = np.zeros(B); active = np.ones(B, bool); status = np.zeros(B, int); move = np.full(B, np.inf); its = np.zeros(B, int)
    alpha_hist = [[] for _ in range(B)]
    Xp, Up, dUp = X.copy(), U.copy(), dU.copy(); nback = np.zeros(B, int)
    theta = np.ones(B)  # share of the start's linear-row violation still in the iterate
    for it in range(max_sqp):
        q = cbind.solve_batch(cfg, veh, dict(inp, X_ref=X, U_ref=U))
        c0all, _ = defect1(veh, inp, X, U)
        for b in np.nonzero(active)[0]:
            status[b] = q["status"][b]
            if q["status"][b] != 0:
                if backoff and it > 0 and nback[b] < backoff:   # the QP about this iterate is infeasible: go back half way
                    nback[b] += 1
                    X[:, :, b] = 0.5 * (X[:, :, b] + Xp[:, :, b]); U[:, :, b] = 0.5 * (U[:, :, b] + Up[:, :, b]); dU[:, :, b] = 0.5 * (dU[:, :, b] + dUp[:, :, b])
                    theta[b] = 0.5 * (theta[b] + thp[b]) if first_ls else theta[b]
                    alpha_hist[b].append(-1)
                    continue
                active[b] = False; continue
            nback[b] = 0
            a = 1.0
            Xq, Uq, dUq = q["X_optm"], q["U_optm"], q["dU_optm"]
            if it > 0 or first_ls:
                J0 = J(b, X, U, dU); J1 = J(b, Xq, Uq, dUq); dJ = J1 - J0; c0 = c0all[b]
                if c0 > 0 and dJ > 0: nu[b] = max(nu[b], dJ / (0.9 * c0))
                nu[b] = max(nu[b], 1e-3)
                if first_ls:   # the linear rows' violation shrinks by (1 - a): part of the merit, weight = nu
                    if dJ > 0 and c0 + theta[b] > 0: nu[b] = max(nu[b], dJ / (0.9 * (c0 + theta[b])))
                    c0 = c0 + theta[b]
                Dd, phi0 = dJ - nu[b] * c0, J0 + nu[b] * c0
                took_soc = False
                if soc:
                    sub = {k: (v[..., b:b+1] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
                    c1 = defect1(veh, sub, Xq[:, :, b:b+1], Uq[:, :, b:b+1])[0][0]
                    if J1 + nu[b] * c1 > phi0 + 1e-4 * Dd + 1e-14 * (1 + abs(phi0)):   # full step refused: second-order correction
                        pr = S.problem(dict(inp, X_ref=X, U_ref=U), b)
                        A_, B_, g_ = Q.linearise(cfg, veh, pr)
                        xt, ut = Xq[:, :, b], Uq[:, :, b]
                        fx = D.rk4(xt[:, :-1].T, ut.T, pr["curvatures"][:-1], pr["T_ref"], veh)     # (N-1, 6)
                        r = fx - (np.einsum("irc,ic->ir", A_, xt[:, :-1].T) + np.einsum("irc,ic->ir", B_, ut.T) + g_)
                        orig = Q.linearise
                        Q.linearise = lambda *_: (A_, B_, g_ + r)
                        try:
                            qp2 = Q.build_qp(cfg, veh, pr); y2, qi2 = Q.solve_dense(qp2)
                        except np.linalg.LinAlgError:
                            qi2 = {"status": 9}
                        finally:
                            Q.linearise = orig
                        if qi2["status"] == 0:
                            o2 = qp2.split(y2)
                            X2, U2, dU2 = o2["X_optm"][:, :, None], o2["U_optm"][:, :, None], o2["dU_optm"]
                            c2 = defect1(veh, sub, X2, U2)[0][0]
                            J2 = NLP.merit_cost(qps[b], Q.pack(qps[b], X2[:, :, 0], U2[:, :, 0], dU2, sigma=0.0))
                            if J2 + nu[b] * c2 <= phi0 + 1e-4 * Dd + 1e-14 * (1 + abs(phi0)):
                                took_soc = True
                                Xq = Xq.copy(); Uq = Uq.copy(); dUq = dUq.copy()
                                Xq[:, :, b], Uq[:, :, b], dUq[:, :, b] = X2[:, :, 0], U2[:, :, 0], dU2
                for t in range(8):
                    if took_soc: a = 1.0; break
                    Xa = X[:, :, b:b+1] + a * (Xq[:, :, b:b+1] - X[:, :, b:b+1]); Ua = U[:, :, b:b+1] + a * (Uq[:, :, b:b+1] - U[:, :, b:b+1])
                    dUa = dU[:, :, b:b+1] + a * (dUq[:, :, b:b+1] - dU[:, :, b:b+1])
                    sub = {k: (v[..., b:b+1] if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
                    ca = defect1(veh, sub, Xa, Ua)[0][0] + ((1 - a) * theta[b] if first_ls else 0.0)
                    Ja = NLP.merit_cost(qps[b], Q.pack(qps[b], Xa[:, :, 0], Ua[:, :, 0], dUa[:, :, 0], sigma=0.0))
                    if Ja + nu[b] * ca <= phi0 + 1e-4 * a * Dd + 1e-14 * (1 + abs(phi0)) or t == 7: break
                    a *= 0.5
            alpha_hist[b].append(2.0 if (it > 0 and soc and took_soc) else a)
            Xp[:, :, b], Up[:, :, b], dUp[:, :, b] = X[:, :, b], U[:, :, b], dU[:, :, b]
            thp = theta.copy() if it == 0 else thp; thp[b] = theta[b]; theta[b] *= (1 - a)
            d = a * (Xq[:, :, b] - X[:, :, b])
            move[b] = np.abs(d / P.SCALE_X[:, None]).max()
            X[:, :, b] += d; U[:, :, b] += a * (Uq[:, :, b] - U[:, :, b]); dU[:, :, b] += a * (dUq[:, :, b] - dU[:, :, b])
            its[b] += 1
            if move[b] <= tol: active[b] = False
        if not active.any(): break
    conv = (status == 0) & (move <= 1e-8) & (theta <= 1e-9)
    fast = x[:, 3] >= 1.6
    print(f"{variant}: converged {conv.mean():.3f} of all, {conv[fast].mean():.3f} of fast; status {np.bincount(status, minlength=3)}")
    if verbose:
        for b in np.nonzero(~conv)[0]:
            print(f"  b={b} vx0={x[b,3]:.2f} status={status[b]} its={its[b]} theta={theta[b]:.1e} move={move[b]:.2e} nu={nu[b]:.2e} alphas={alpha_hist[b][-8:]}")
    return conv, x

if __name__ == "__main__":
    v = sys.argv[1] if len(sys.argv) > 1 else "base"
    kw = dict(base={}, backoff=dict(backoff=6), first=dict(first_ls=True), both=dict(backoff=6, first_ls=True), both100=dict(backoff=6, max_sqp=100), soc=dict(backoff=6, soc=True))[v]
    run(v, **kw)
