"""Shared parity bookkeeping: per-problem scaled error against the dense optimum, and the contract check that holds
strictly complementary problems to 1e-6 and only oracle-labelled degenerate ones to the relaxed bound
(tests/tolerances.py)."""
import numpy as np

from oracle import params as P, qp as Q
from tolerances import TOL_DEGENERATE, TOL_DU, TOL_XU


def per_problem_err(out, ref):
    """max scaled |X - X*|, |U - U*|, |dU - dU*| per problem (batch axis last)."""
    ex = np.abs((np.asarray(out["X_optm"]) - ref["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
    eu = np.abs((np.asarray(out["U_optm"]) - ref["U_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    ed = np.abs((np.asarray(out["dU_optm"]) - ref["dU_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    return np.maximum(ex, eu), ed


def assert_contract(out, ref, margin, certified=None, who="kernel"):
    """Every problem solved; strictly complementary ones (oracle's margin >= DEGENERATE_MARGIN, certified dense optimum)
    within TOL_XU / TOL_DU of the dense optimum, the others within TOL_DEGENERATE.  Returns the degenerate fraction."""
    status = np.asarray(out["status"])
    assert (status == 0).all(), (who, np.where(status != 0)[0], status[status != 0])
    exu, ed = per_problem_err(out, ref)
    strict = np.asarray(margin) >= Q.DEGENERATE_MARGIN
    if certified is not None:
        strict &= np.asarray(certified, dtype=bool)
    assert strict.mean() > 0.5, strict.mean()
    worst = int(np.argmax(np.where(strict, exu, 0.0)))
    assert exu[strict].max() < TOL_XU, (who, "strict", worst, exu[worst], float(np.asarray(margin)[worst]))
    assert ed[strict].max() < TOL_DU, (who, "strict dU", ed[strict].max())
    if (~strict).any():
        assert exu[~strict].max() < TOL_DEGENERATE, (who, "degenerate", exu[~strict].max())
        assert ed[~strict].max() < 40 * TOL_DEGENERATE, (who, "degenerate dU", ed[~strict].max())
    return float((~strict).mean())


def dense_reference(cfg, veh, inp, problems, ss_x=None, ss_j=None):
    """Dense optimum + margin for the listed problems of a batch (seconds each at N >= 40: keep the list short)."""
    from oracle import scenario as S

    N = cfg.N
    B = len(problems)
    ref = {"X_optm": np.zeros((6, N, B)), "U_optm": np.zeros((2, N - 1, B)), "dU_optm": np.zeros((2, N - 1, B))}
    margin, ok, qps, ys = np.zeros(B), np.zeros(B, bool), [], []
    for j, b in enumerate(problems):
        kw = {} if ss_x is None else {"ss_x": ss_x[:, :, b], "ss_j": ss_j[:, b]}
        qp = Q.build_qp(cfg, veh, S.problem(inp, b), **kw)
        y, info = Q.solve_dense(qp)
        o = qp.split(y)
        for k in ref:
            ref[k][..., j] = o[k]
        margin[j] = Q.strict_complementarity(qp, y, info["lam"])
        ok[j] = info["status"] == 0 and bool(info.get("polished"))
        qps.append(qp)
        ys.append(y)
    return ref, margin, ok, qps, ys


def assert_same_iterations(a, b):
    """Kernel and serial twin run the same iteration: the counts are equal on >= 90 % of the problems and within one on
    >= 97 %.  The few that differ by more are problems that crawl towards the tolerance at the floor of fp64 (mu a few
    1e-14, changing by a factor 0.6 .. 0.9 per iteration): which iteration first dips under it depends on the last bits,
    i.e. on FMA contraction and summation order."""
    d = np.abs(np.asarray(a).astype(int) - np.asarray(b).astype(int))
    assert (d == 0).mean() >= 0.9 and (d <= 1).mean() >= 0.97 and d.max() <= 6, (float((d == 0).mean()), float((d <= 1).mean()), int(d.max()))
