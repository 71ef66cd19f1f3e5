"""GPU parity on the book-shaped workloads of BASELINE.json (configs[0], [2], [3]) at FULL size.

Every chapter lattice of the Kokoro (64 chapters, 2.77 M frames) and Meian (120 chapters, 5.27 M frames) stand-ins
goes through the HIP path in ONE launch per kernel form and is compared, bit for bit, with the C oracle
(path, labels, per-frame scores, cumulative score, end position).  configs[3] is "Meian sharded across 8 GPUs": the
split is replayed on one GPU, each rank's shard as a launch of its own, and the union must equal the one-launch
result.  configs[0] (gongitsune: T = 81140 frames, V = 39, S = 2000) runs at full size as a single lattice.

Reference: the per-file loop of run_example.py:248-254 around kokoro_align/align.py:112-124.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

MODES = ["wave", "workgroup", "wave_exact", "tiled", "wave+parallel", "tiled+parallel", "auto"]
_oracle_cache = {}


def _book(name):
    from kokoro_align_amd import workloads as W
    return {"kokoro": W.kokoro_book, "meian": W.meian_book}[name]()[1]


def _oracle(name):
    """C oracle of every chapter (host threads; computed once per session)."""
    if name not in _oracle_cache:
        from kokoro_align_amd import workloads as W
        threads = max(1, min(16, (os.cpu_count() or 2) - 1))
        _oracle_cache[name] = O.lattice_batch_c(_book(name), W.V_MODEL, W.BOOK_SEED0, 1000, 4, threads=threads)
    return _oracle_cache[name]


@pytest.fixture(scope="module")
def books_on_device():
    import torch
    assert torch.cuda.is_available()
    from kokoro_align_amd import workloads as W
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = W.device_book(_book(name))
        return cache[name]
    return get


def _engine():
    import torch
    from kokoro_align_amd import _lib
    return _lib.default_engine(torch.cuda.current_device())


def _set(eng, mode):
    """'form' or 'form+parallel' (chunk-parallel backtrace forced); 'auto' leaves both choices to the library"""
    form, _, bt = mode.partition("+")
    eng.set_mode(form)
    eng.set_backtrace(bt or ("auto" if form == "auto" else "serial"))


def _check_against_oracle(batch, want, idxs=None):
    idxs = range(batch.n) if idxs is None else idxs
    for k, i in enumerate(idxs):
        w = want[i]
        assert w is not None and batch.status[k] == 0, i
        assert np.array_equal(batch.path[k].cpu().numpy(), w[0]), f"chapter {i}: best_path differs"
        assert np.array_equal(batch.best_labels[k].cpu().numpy(), w[1]), f"chapter {i}: best_labels differ"
        assert np.array_equal(batch.best_scores[k].cpu().numpy().view(np.int32), w[2].view(np.int32)), f"chapter {i}: best_scores differ"
        assert np.float32(batch.total[k]).view(np.int32) == np.float32(w[3]).view(np.int32), f"chapter {i}: total score differs"


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["kokoro", "meian"])
def test_book_one_launch_every_chapter_vs_oracle(books_on_device, name, mode):
    from kokoro_align_amd.align import DeviceBatch
    lps, labs = books_on_device(name)
    want = _oracle(name)
    eng = _engine()
    _set(eng, mode)
    try:
        b = DeviceBatch(lps, labs)
        b.run()
        _check_against_oracle(b, want)
    finally:
        _set(eng, "auto")


@pytest.mark.parametrize("mode", ["auto", "tiled", "wave+parallel"])
def test_meian_sharded_over_8_ranks_equals_one_launch(books_on_device, mode):
    """BASELINE configs[3]: the LPT split of sharding.shard_for_rank, every rank's shard as its own launch."""
    from kokoro_align_amd.align import DeviceBatch
    from kokoro_align_amd.sharding import shard_for_rank
    shapes = _book("meian")
    lps, labs = books_on_device("meian")
    want = _oracle("meian")
    eng = _engine()
    _set(eng, mode)
    try:
        seen = []
        for rank in range(8):
            mine = shard_for_rank(shapes, rank, 8)
            assert mine, "an 8-way split of 120 chapters leaves no rank empty"
            b = DeviceBatch([lps[i] for i in mine], [labs[i] for i in mine])
            b.run()
            _check_against_oracle(b, want, mine)
            seen += mine
        assert sorted(seen) == list(range(len(shapes)))
    finally:
        _set(eng, "auto")


@pytest.mark.parametrize("mode", MODES)
def test_cfg1_gongitsune_full_size(mode):
    """BASELINE configs[0] stand-in at full size: one MP3 of 0:15:42 = 81140 frames, V = 39, S = 2000."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    c = W.CFG1
    lps, labs = W.device_book([(c["T"], c["S"])], V=c["V"], seed0=77)
    want = O.lattice_batch_c([(c["T"], c["S"])], c["V"], 77, threads=1)
    eng = _engine()
    _set(eng, mode)
    try:
        b = DeviceBatch(lps, labs)
        b.run()
        _check_against_oracle(b, want)
        assert int(b.path[0][-1]) == 2 * c["S"]
    finally:
        _set(eng, "auto")
    del lps, labs
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------
# BASELINE configs[4]: the long-form stress lattice, banded (beam 1000) and as the whole lattice (beam >= 2L, "tiled DP")
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,S,V", [(8000, 10000, 64), (6000, 6000, 39), (3000, 9000, 20)])
def test_full_lattice_mode_vs_oracle_reduced_size(T, S, V):
    """beam_size >= 2L: every position of every frame is in the band (align.py:64-65 degenerates to lo = 0, hi = L).
    L up to 20001 = 79 tiles alive from the first frame to the last, one wavefront each; bit-exact vs the C oracle."""
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    L = 2 * S + 1
    lps, labs = W.device_book([(T, S)], V=V, seed0=31)
    lp = O.hash_logprobs_c(T, V, 31)
    lab = O.hash_labels(S, V, 31)
    eng = _engine()
    for beam in (2 * L, 2 * L + 7):
        try:
            want = O.ctc_best_path_c(lp, lab, beam, 4, return_total=True)
        except ValueError:
            want = None
        for mode in ("auto", "tiled+parallel", "tiled"):
            _set(eng, mode)
            try:
                b = DeviceBatch(lps, labs, beam)
                st = b.run(raise_on_error=False)
                if want is None:
                    assert st[0] == -1
                    continue
                assert st[0] == 0
                assert np.array_equal(b.path[0].cpu().numpy(), want[0]), (mode, beam)
                assert np.array_equal(b.best_labels[0].cpu().numpy(), want[1])
                assert np.array_equal(b.best_scores[0].cpu().numpy().view(np.int32), want[2].view(np.int32))
                assert np.float32(b.total[0]).view(np.int32) == np.float32(want[3]).view(np.int32)
            finally:
                _set(eng, "auto")


@pytest.mark.parametrize("beam", [1000, None])
def test_cfg5_stress_lattice_at_full_size_properties(beam):
    """T = 500000 x V = 64, S = 50000 (L = 100001): the banded case (beam 1000) and the whole lattice (beam >= 2L,
    5e10 cells - the CPU reference would need hours), checked through size-independent properties: monotone path with
    steps <= 3 that ends on the trailing blank, labels = lab'[path], scores = the emissions, and the float32 chain of
    the scores along the path equal to the forward pass's best cumulative score bit for bit."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    c = W.CFG5
    T, S, V = c["T"], c["S"], c["V"]
    L = 2 * S + 1
    lps, labs = W.device_book([(T, S)], V=V, seed0=5)
    b = DeviceBatch(lps, labs, beam if beam else 2 * L + 2)
    _set(_engine(), "auto")
    b.run()
    p = b.path[0].cpu().numpy(); l = b.best_labels[0].cpu().numpy(); s = b.best_scores[0].cpu().numpy()
    ext = np.zeros(L, np.int32); ext[1::2] = labs[0].cpu().numpy()
    d = np.diff(p)
    assert (d >= 0).all() and (d <= 3).all() and p[-1] == L - 1 and p[0] in (0, 1, 3)
    assert np.array_equal(l, ext[p])
    assert np.array_equal(s, lps[0].cpu().numpy()[np.arange(T), l])
    chain = np.add.accumulate(s, dtype=np.float32)[-1]
    assert np.float32(chain).view(np.int32) == np.float32(b.total[0]).view(np.int32)
    del b, lps, labs
    torch.cuda.empty_cache()
