"""Property test: the two CPU restatements (C dense band, NumPy per-frame port) agree on random tiny
lattices beyond the goldens — odd beams, max_move up to 8, -inf entries, label 0, ValueError parity."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.mark.parametrize("seed", range(6))
def test_c_oracle_equals_numpy_port(seed):
    rng = np.random.default_rng(1000 + seed)
    n_ok = n_err = 0
    for _ in range(60):
        V = int(rng.integers(2, 12))
        S = int(rng.integers(0, 40))
        T = int(rng.integers(1, 90))
        beam = int(rng.choice([1, 2, 3, 5, 8, 13, 33, 1000]))
        mm = int(rng.integers(1, 9))
        lp = np.round(rng.standard_normal((T, V)) * 2, int(rng.integers(0, 3))).astype(np.float32)
        if rng.random() < 0.3:
            lp = np.where(rng.random((T, V)) < 0.2, -np.inf, lp).astype(np.float32)
        labels = rng.integers(0, V, size=S).astype(np.int32)
        try:
            want = O.ctc_best_path_numpy(lp, labels, beam, mm)
        except ValueError:
            with pytest.raises(ValueError):
                O.ctc_best_path_c(lp, labels, beam, mm)
            n_err += 1
            continue
        got = O.ctc_best_path_c(lp, labels, beam, mm)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.array_equal(got[2].view(np.int32), want[2].view(np.int32))
        n_ok += 1
    assert n_ok >= 20 and n_err >= 1


def test_band_cells_helper():
    # sum of band widths for cfg2
    assert O.band_cells(50000, 5000, 1000) == 49376749
