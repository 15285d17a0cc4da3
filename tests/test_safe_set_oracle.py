"""Safe-set query oracle (C) against a numpy brute force on the reference's recorded laps (CPU)."""
from pathlib import Path

import numpy as np

from oracle import cbind

GOLD = Path(__file__).parent / "golden" / "barc_ss"


def load_laps():
    return [np.loadtxt(GOLD / f"ss_lap_{i}_x.txt") for i in (1, 2, 3)]


def brute(laps, L, S, K, q):
    xs, js = [], []
    for lap in reversed(laps):
        n = lap.shape[0]
        rep = np.concatenate([lap - [L, 0, 0, 0, 0, 0], lap, lap + [L, 0, 0, 0, 0, 0]])
        J = np.linspace(n - 1, 0, n)
        Jr = np.concatenate([J + n - 1, J, J - n + 1])  # safe_set.cpp:122,128
        d = (rep[:, 0] - q[0]) ** 2 + (rep[:, 1] - q[1]) ** 2
        idx = np.lexsort((np.arange(3 * n), d))[:K]
        xs.append(rep[idx])
        js.append(Jr[idx])
        if sum(len(a) for a in xs) >= S:
            break
    x = np.concatenate(xs)[:S]
    j = np.concatenate(js)[:S]
    nf = len(x)
    if nf < S:
        x = np.concatenate([x, np.repeat(x[-1:], S - nf, 0)])
        j = np.concatenate([j, np.repeat(j[-1:], S - nf)])
    return x.T, j - j[0], nf


def test_ss_query_matches_brute_force():
    laps = load_laps()
    L = 17.05
    rng = np.random.default_rng(0)
    q = np.stack([rng.uniform(-1.0, L + 1.0, 50), rng.uniform(-0.3, 0.3, 50)])
    for S, K in ((96, 32), (40, 32), (160, 32), (7, 3)):
        ss_x, ss_j, nf = cbind.ss_query_batch(laps, L, S, K, q)
        for b in range(q.shape[1]):
            x, j, n = brute(laps, L, S, K, q[:, b])
            assert nf[b] == n
            assert np.array_equal(ss_x[:, :, b], x)
            assert np.array_equal(ss_j[:, b], j)


def test_ss_query_edge_cases():
    laps = load_laps()
    q = np.array([[1.0], [0.0]])
    ss_x, ss_j, nf = cbind.ss_query_batch([], 17.0, 8, 4, q)  # empty store
    assert nf[0] == 0
    short = [laps[0][:2]]  # 2-sample lap: 6 unrolled points < K
    ss_x, ss_j, nf = cbind.ss_query_batch(short, 17.0, 8, 32, q)
    assert nf[0] == 6 and np.all(ss_x[:, 6:, 0] == ss_x[:, 5:6, 0])
    assert ss_j[0, 0] == 0.0
