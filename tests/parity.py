"""Shared parity bookkeeping: per-problem scaled error against the dense optimum, and the contract check that holds every
problem to 1e-6 (tests/tolerances.py)."""
import numpy as np

from oracle import params as P, qp as Q
from tolerances import TOL_DU, TOL_XU


def per_problem_err(out, ref):
    """max scaled |X - X*|, |U - U*|, |dU - dU*| per problem (batch axis last)."""
    ex = np.abs((np.asarray(out["X_optm"]) - ref["X_optm"]) / P.SCALE_X[:, None, None]).max(axis=(0, 1))
    eu = np.abs((np.asarray(out["U_optm"]) - ref["U_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    ed = np.abs((np.asarray(out["dU_optm"]) - ref["dU_optm"]) / P.SCALE_U[:, None, None]).max(axis=(0, 1))
    return np.maximum(ex, eu), ed


def assert_contract(out, ref, margin, certified=None, who="kernel"):
    """Every problem solved and within TOL_XU / TOL_DU of the dense optimum -- strictly complementary or degenerate alike
    (tests/tolerances.py).  Returns the degenerate fraction (oracle's margin below DEGENERATE_MARGIN) for the record."""
    status = np.asarray(out["status"])
    assert (status == 0).all(), (who, np.where(status != 0)[0], status[status != 0])
    exu, ed = per_problem_err(out, ref)
    worst = int(np.argmax(exu))
    assert exu.max() < TOL_XU, (who, worst, exu[worst], float(np.asarray(margin)[worst]))
    assert ed.max() < TOL_DU, (who, "dU", ed.max())
    return float((np.asarray(margin) < Q.DEGENERATE_MARGIN).mean())


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
    """Kernel and serial twin run the same iteration (the count is interior-point iterations + polish rounds): equal on
    >= 90 % of the problems and within one on >= 95 %.  The few that differ by more are borderline decisions that depend on
    the last bits, i.e. on FMA contraction and summation order: a polish attempt that one of the two accepts and the other
    refuses (a held row met to 0.9e-9 or 1.1e-9) costs the refusing one the three to five interior-point iterations down to
    its own tolerance; a problem crawling at the floor of fp64 (mu a few 1e-14) dips under the tolerance an iteration
    apart.  Both end at the same optimum (the value tests hold them to 1e-6 of each other)."""
    a, b = np.asarray(a).astype(int), np.asarray(b).astype(int)
    d = np.abs(a - b)
    assert (d == 0).mean() >= 0.9 and (d <= 1).mean() >= 0.95 and d.max() <= 8, (float((d == 0).mean()), float((d <= 1).mean()), int(d.max()))
    # ... and the borderline decisions fall both ways: a kernel whose polish were refused (or accepted) systematically more
    # often than the twin's would move the MEAN count, which a handful of individual differences of up to 8 does not
    # (the per-problem bound alone was loosened from 2 to 8 when the polish arrived)
    if a.size >= 16:
        assert abs(a.mean() - b.mean()) <= 0.25, (float(a.mean()), float(b.mean()))
