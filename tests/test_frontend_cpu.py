"""Audio front end, CPU side: the oracle's restatement of the reference's silence
splitting against the goldens made by the reference itself (tests/golden/make_golden.py g6), and consistency
checks of the MFCC restatement (which has no reference output to be pinned to: torchaudio is not in the image)."""
import numpy as np
import pytest

from golden_util import g6
from oracle import frontend_oracle as F


def test_silent_ranges_match_the_reference():
    for c in g6()["silent_ranges"]:
        v = np.array([ch == "1" for ch in c["mask"]])
        if "raises" in c:
            with pytest.raises(Exception):
                F.silent_ranges(v)
        else:
            assert F.silent_ranges(v).tolist() == c["ranges"], c["mask"]


def test_split_points_match_the_reference():
    g = g6()
    par = g["parameters"]
    assert par == F.split_parameters(g["sample_rate"], 512)
    for c in g["cases"]:
        x = F.hash_waveform(c["n"], c["seed"], c["pieces"])
        if c["status"] == 1:
            with pytest.raises(ValueError, match="cannot be split"):
                F.split_points(x, **par)
            continue
        got = F.split_points(x, **par)
        assert [int(v) for v in got] == c["split_points"], c["name"]


def test_mfcc_restatement_is_self_consistent():
    """Independent formulations of the pieces: the DCT matrix is orthonormal, a pure tone lands in the mel filter
    that contains it, scaling the input by 10 moves every un-clamped mel level by 20 dB, the frame count follows
    center=True framing, and the top_db floor holds."""
    d = F.dct_matrix(40, 40)
    assert np.allclose(d.T @ d, np.eye(40), atol=1e-12)
    fb = F.mel_filterbank(257, 40, 22050)
    assert fb.shape == (257, 40) and (fb >= 0).all() and fb.max() <= 1.0 + 1e-12
    sr, n = 22050, 22050
    t = np.arange(n) / sr
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    m = F.mfcc(tone)
    assert m.shape == (1 + n // 256, 40)
    db = m @ F.dct_matrix(40, 40).T                       # back to mel dB (orthonormal DCT)
    k = int(np.argmax(db[40]))
    centre = np.argmax(fb[:, k]) * (sr // 2) / 256.0
    assert abs(centre - 1000.0) < 150.0
    assert db.min() >= db.max() - 80.0 - 1e-9
    rng = np.random.default_rng(0)
    y = rng.standard_normal(5000) * 0.1
    a = F.mfcc(y) @ F.dct_matrix(40, 40).T
    b = F.mfcc(10.0 * y) @ F.dct_matrix(40, 40).T
    assert np.allclose(b - a, 20.0, atol=1e-9)            # nothing near the floor for white noise


def test_product_merge_of_short_pieces_is_the_reference_order():
    """kokoro_align_amd.preprocess replaces the reference's quadratic merge loop (preprocess.py:81-95) by a heap over a
    linked list; the removals must be the same, ties included (host logic: no GPU needed)."""
    import importlib
    P = importlib.import_module("kokoro_align_amd.preprocess")

    def loop(points, num_frames, m):          # the loop as oracle/frontend_oracle.py restates it
        points = np.asarray(points)
        while len(points):
            dist = np.append(points, num_frames) - np.insert(points, 0, 0)
            i = np.argmin(dist)
            if dist[i] > m:
                break
            if i == 0:
                points = np.delete(points, i)
            elif i == len(points):
                points = np.delete(points, len(points) - 1)
            elif dist[i - 1] < dist[i + 1]:
                points = np.delete(points, i - 1)
            else:
                points = np.delete(points, i)
        return points

    rng = np.random.default_rng(8)
    for _ in range(1500):
        k = int(rng.integers(0, 60))
        n = int(rng.integers(k + 1, 500))
        pts = np.sort(rng.choice(np.arange(1, n), size=min(k, n - 1), replace=False)) if k else np.array([], dtype=np.int64)
        m = float(rng.uniform(0, 80))
        assert list(loop(pts, n, m)) == list(P._merge_short_pieces(pts, n, m))
