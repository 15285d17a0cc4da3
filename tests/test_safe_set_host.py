"""Host logic of the safe set (SafeSetManager ring, SafeSetRecorder segmentation and lap files; safe_set.cpp:139-151,
260-322) -- CPU only.  The lap files under tests/golden/barc_ss are the reference's own test data."""
from pathlib import Path

import numpy as np

GOLD = Path(__file__).resolve().parent / "golden" / "barc_ss"


def test_lap_files_round_trip(pkg, tmp_path):
    ss = pkg.safe_set
    for s, cols in (("x", 6), ("u", 2), ("k", 1), ("t", 1)):
        a = ss.read_txt(GOLD / f"ss_lap_2_{s}.txt")
        assert a.ndim == 2 and a.shape[1] == cols and a.shape[0] > 400
        ss.write_txt(a, tmp_path / f"lap_{s}.txt")
        assert (ss.read_txt(tmp_path / f"lap_{s}.txt") == a).all()     # %.16e is lossless for doubles
    # same text layout as the reference writes: one sample per row, scientific with 16 digits
    first = open(tmp_path / "lap_x.txt").readline().split()
    assert len(first) == 6 and all("e" in w and len(w.split("e")[0].lstrip("-").replace(".", "")) == 17 for w in first)


def test_recorder_loads_reference_laps_and_keeps_a_ring(pkg):
    ss = pkg.safe_set
    man = ss.SafeSetManager(max_lap_stored=2)
    rec = ss.SafeSetRecorder(man)
    rec.load([str(GOLD / f"ss_lap_{i}") for i in (1, 2, 3)] + [str(GOLD / "missing_lap")], 17.06)
    assert rec.lap_count == 3 and len(man.laps) == 2              # the failed file is skipped, the ring keeps the newest two
    x3 = np.loadtxt(GOLD / "ss_lap_3_x.txt")
    assert (man.laps[-1][0] == x3).all() and man.laps[-1][1].shape == (x3.shape[0], 2)


def test_recorder_segments_laps_on_the_abscissa_wrap(pkg, tmp_path):
    ss = pkg.safe_set
    man = ss.SafeSetManager(max_lap_stored=5)
    rec = ss.SafeSetRecorder(man, to_file=True, file_prefix=str(tmp_path) + "/ss_")
    L, per_lap = 10.0, 25
    added = []
    for j in range(3 * per_lap + 7):
        s = (0.4 * L + j * L / per_lap) % L
        x = np.array([s, 0.01 * j, 0.0, 1.5, 0.0, 0.0])
        added.append(rec.step(x, [0.001 * j, -0.002 * j], 0.1 * j, 0.03 * j, L))
    # sample 0 only seeds the abscissa; the partial lap up to the first wrap is discarded; two full laps are stored
    assert sum(added) == 2 and len(man.laps) == 2 and rec.lap_count == 3
    for lap in man.laps:
        assert lap[0].shape == (per_lap, 6) and lap[1].shape == (per_lap, 2) and lap[2].shape == (per_lap,)
        assert lap[0][0, 0] < lap[0][-1, 0] and lap[0][0, 0] < L / per_lap + 1e-12   # starts right after the line
    first_wrap = next(j for j in range(1, 200) if (0.4 * L + j * L / per_lap) % L < (0.4 * L + (j - 1) * L / per_lap) % L)
    assert man.laps[0][3][0] == 0.03 * first_wrap and man.laps[0][2][0] == 0.1 * first_wrap
    # files carry the lap counter at completion: the discarded lap is number 0, the stored ones 1 and 2
    for i in (1, 2):
        assert (ss.read_txt(f"{tmp_path}/ss_lap_{i}_x.txt") == man.laps[i - 1][0]).all()
        assert ss.read_txt(f"{tmp_path}/ss_lap_{i}_t.txt").shape == (per_lap, 1)
