"""The CPU oracle (oracle/) against the goldens produced by the reference implementation
(kokoro_align/align.py:43-109).  Bit-exact on path/labels, bit-exact on scores (the scores
are gathers of the inputs)."""
import numpy as np
import pytest

from golden_util import g1_cases, g2_cases, g3_case, sha
from oracle import oracle as O


@pytest.mark.parametrize("impl", ["c", "numpy"])
def test_g1_tiny(impl):
    fn = O.ctc_best_path_c if impl == "c" else O.ctc_best_path_numpy
    n_ok = n_err = 0
    for c in g1_cases():
        if c["status"] == 1:
            with pytest.raises(ValueError):
                fn(c["lp"], c["labels"], beam_size=c["beam"], max_move=c["max_move"])
            n_err += 1
            continue
        p, l, s = fn(c["lp"], c["labels"], beam_size=c["beam"], max_move=c["max_move"])
        assert np.array_equal(p, c["path"]), c["idx"]
        assert np.array_equal(l, c["best_labels"]), c["idx"]
        assert np.array_equal(s.view(np.uint32), c["best_scores"].view(np.uint32)), c["idx"]
        n_ok += 1
    assert n_ok > 150 and n_err > 20


def test_g2_medium_c():
    for c in g2_cases():
        lp = O.hash_logprobs(c["T"], c["V"], c["seed"])
        labels = O.hash_labels(c["S"], c["V"], c["seed"])
        p, l, s = O.ctc_best_path_c(lp, labels, c["beam"], c["max_move"])
        assert np.array_equal(p, c["path"]), c["idx"]
        assert sha(l) == c["sha_labels"] and sha(s) == c["sha_scores"], c["idx"]


def test_g2_numpy_port_one():
    c = g2_cases()[4]
    lp = O.hash_logprobs(c["T"], c["V"], c["seed"])
    labels = O.hash_labels(c["S"], c["V"], c["seed"])
    p, l, s = O.ctc_best_path_numpy(lp, labels, c["beam"], c["max_move"])
    assert np.array_equal(p, c["path"])
    assert sha(l) == c["sha_labels"] and sha(s) == c["sha_scores"]


def test_g3_cfg2_c():
    c = g3_case()
    lp = O.hash_logprobs(c["T"], c["V"], c["seed"])
    labels = O.hash_labels(c["S"], c["V"], c["seed"])
    p, l, s = O.ctc_best_path_c(lp, labels, c["beam"], c["max_move"])
    assert np.array_equal(p, c["path"])
    assert sha(l) == c["sha_labels"] and sha(s) == c["sha_scores"]
    assert abs(float(np.sum(s.astype(np.float64))) - c["sum_scores"]) < 1e-6


def test_hash_generator_c_matches_numpy():
    T, V, S = 257, 39, 100
    lp = np.empty((T, V), np.float32)
    O.lib().kao_hash_logprobs_f32(lp.ctypes.data, T, V, V, 5)
    assert np.array_equal(lp, O.hash_logprobs(T, V, 5))
    lab = np.empty(S, np.int32)
    O.lib().kao_hash_labels_i32(lab.ctypes.data, S, V, 5)
    assert np.array_equal(lab, O.hash_labels(S, V, 5))
    assert lab.min() >= 1 and lab.max() < V
