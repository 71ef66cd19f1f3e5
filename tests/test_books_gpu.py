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

# "tiled/256", "tiled/128": the tile pipeline with the tile width forced (ka_tiled2.hpp / ka_tiled_stream.hpp; plain "tiled" lets the
# library choose by the number of tiles alive at once)
MODES = ["wave", "wave_exact", "tiled/256", "tiled/128", "wave+parallel", "tiled/256+parallel", "tiled/128+parallel", "auto"]
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
    form, _, width = form.partition("/")
    eng.set_tile_width(int(width or 0))
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
        for mode in ("auto", "tiled/128+parallel", "tiled/256", "tiled/128"):
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


# ------------------------------------------------------------------------------------------
# round 3: the input classes around the green suite
# ------------------------------------------------------------------------------------------
def _lattice_with_neg_inf(T, S, V, seed, frac=0.10):
    """hash log-probs with `frac` of the entries -inf (masked CTC posteriors: legal input, align.py:67-85 tracks the live
    set explicitly), column 0 kept finite in most frames so that paths survive"""
    lp = O.hash_logprobs_c(T, V, seed).copy()
    rng = np.random.default_rng(seed)
    lp[rng.random(lp.shape) < frac] = -np.inf
    return lp, O.hash_labels(S, V, seed)


def _compare(b, k, want, tag):
    assert np.array_equal(b.path[k].cpu().numpy(), want[0]), f"{tag}: best_path differs"
    assert np.array_equal(b.best_labels[k].cpu().numpy(), want[1]), f"{tag}: best_labels differ"
    assert np.array_equal(b.best_scores[k].cpu().numpy().view(np.int32), want[2].view(np.int32)), f"{tag}: best_scores differ"
    assert np.float32(b.total[k]).view(np.int32) == np.float32(want[3]).view(np.int32), f"{tag}: total differs"


@pytest.mark.parametrize("beam", [2000, 5000, None])
def test_wide_band_with_neg_inf_log_probs_is_answered(beam):
    """Bands wider than the exact kernels' 1009-position ring whose log-probs hold -inf: the reference answers them
    (align.py:67-85), so KA_MODE_AUTO must too (the tiled form declines, ka_batch_finish hands the lattice to the generic
    kernels); only the explicit KA_MODE_TILED reports KA_ERR_NONFINITE (-7).  One launch mixes a finite wide lattice, two
    with -inf and a narrow one with -inf (redone by the exact kernels inside the launch): indices must not get mixed up."""
    import torch
    from kokoro_align_amd.align import DeviceBatch
    shapes = [(6000, 3000, 64, 41, 0.0), (6000, 3000, 64, 42, 0.10), (2500, 400, 64, 43, 0.10), (5000, 2600, 64, 44, 0.10)]
    L_max = 2 * 3000 + 1
    bm = beam if beam else 2 * L_max + 2
    host = [_lattice_with_neg_inf(T, S, V, seed, frac) for T, S, V, seed, frac in shapes]
    want = []
    for lp, lab in host:
        try:
            want.append(O.ctc_best_path_c(lp, lab, bm, 4, return_total=True))
        except ValueError:
            want.append(None)
    assert sum(w is not None for w in want) >= 3, "the case must exercise paths, not only empty beams"
    lps = [torch.from_numpy(lp).cuda() for lp, _ in host]
    labs = [torch.from_numpy(np.asarray(lab, np.int32)).cuda() for _, lab in host]
    eng = _engine()
    try:
        for mode in ("auto", "auto+parallel"):
            _set(eng, mode)
            b = DeviceBatch(lps, labs, bm)
            st = b.run(raise_on_error=False)
            for k, w in enumerate(want):
                if w is None:
                    assert st[k] == -1, (mode, k, st[k])
                else:
                    assert st[k] == 0, (mode, k, st[k])
                    _compare(b, k, w, f"{mode} beam {bm} lattice {k}")
        # host buffers in, host buffers out (the boundary's KA_MEM_HOST path goes through the same hand-over)
        _set(eng, "auto")
        import kokoro_align_amd as ka
        if want[1] is not None:
            p, l, s = ka.ctc_best_path(host[1][0], host[1][1], beam_size=bm, verbose=False)
            assert np.array_equal(p, want[1][0]) and np.array_equal(l, want[1][1]) and np.array_equal(s.view(np.int32), want[1][2].view(np.int32))
        # the explicit tiled form has no answer for the wide ones with -inf and says so
        for mode in ("tiled/256", "tiled/128"):
            _set(eng, mode)
            b = DeviceBatch(lps, labs, bm)
            st = b.run(raise_on_error=False)
            assert st[0] == 0 and st[1] == -7 and st[3] == -7, (mode, st)
            _compare(b, 0, want[0], f"{mode}, finite lattice")
            if want[2] is not None:
                assert st[2] == 0
                _compare(b, 2, want[2], f"{mode}, narrow lattice redone exactly")
    finally:
        _set(eng, "auto")


def test_cfg2_whole_lattice_50k_frames_vs_oracle():
    """T = 50000, S = 5000, beam_size >= 2L: BASELINE configs[1]'s lattice without a band (5e8 cells, 40 tiles
    alive for all 50000 frames) against the C oracle, bit for bit - the largest unbanded case the oracle's back-pointer
    array (T x L bytes = 500 MB) allows; configs[4]'s 5e10 cells are covered by properties above."""
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    c = W.CFG2
    T, S, V = c["T"], c["S"], c["V"]
    L = 2 * S + 1
    lps, labs = W.device_book([(T, S)], V=V, seed0=0)
    want = O.ctc_best_path_c(O.hash_logprobs_c(T, V, 0), O.hash_labels(S, V, 0), 2 * L + 2, 4, return_total=True)
    eng = _engine()
    try:
        for mode in ("auto", "tiled/256", "tiled/128"):
            _set(eng, mode)
            b = DeviceBatch(lps, labs, 2 * L + 2)
            b.run()
            _compare(b, 0, want, f"{mode}, whole 50000 x 10001 lattice")
    finally:
        _set(eng, "auto")


def test_meian_book_with_the_hand_off_self_check_on(books_on_device):
    """ka_engine_set_verify(1): the halo region starts as a NaN sentinel and every packet a tile consumes is checked against
    it - a packet read before it was written gives KA_ERR_INTERNAL (-8).  The Meian book (120 chapters, ~2600 tiles, every
    hand-off of the pipeline) must come through with status 0 and the oracle's paths; verify 2 (full drain before every
    publish) must give the same."""
    from kokoro_align_amd.align import DeviceBatch
    lps, labs = books_on_device("meian")
    want = _oracle("meian")
    eng = _engine()
    try:
        for mode, flags in (("tiled/256", 1), ("tiled/256", 3), ("tiled/128", 1), ("tiled/128", 3)):
            _set(eng, mode)
            eng.set_verify(flags)
            b = DeviceBatch(lps, labs)
            st = b.run(raise_on_error=False)
            assert not (st != 0).any(), (mode, flags, sorted(set(st[st != 0].tolist())))
            _check_against_oracle(b, want)
    finally:
        eng.set_verify(0)
        _set(eng, "auto")


def test_corpus_under_auto_sample_vs_oracle_and_properties_on_all():
    """The unit of work of the reference's run_example.py without --dataset: the 14 enabled datasets of example.json as ONE
    launch (462 chapter lattices, 20k..94k frames) under KA_MODE_AUTO / KA_BACKTRACE_AUTO - the regime in which the library
    has to pick a kernel form per lattice (the longest tiled, the others one wavefront each; the longest walked back
    chunk-parallel, the others serially).  40 chapters spread over the length range against the C oracle, bit for bit; every
    chapter through size-independent properties.  The same with the split forced to a few other places: results must not
    depend on it."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    shapes, seeds = [], []
    for k, (_, sh) in enumerate(W.corpus()):
        shapes += sh
        seeds += [W.corpus_seed0(k) + i for i in range(len(sh))]
    lps, labs = [], []
    for (T, S), seed in zip(shapes, seeds):
        lp1, lab1 = W.device_book([(T, S)], seed0=seed)
        lps += lp1; labs += lab1
    n = len(shapes)
    by_len = sorted(range(n), key=lambda i: shapes[i][0])
    sample = [by_len[j] for j in np.linspace(0, n - 1, 40).astype(int)]
    from concurrent.futures import ThreadPoolExecutor

    def oracle_one(i):
        T, S = shapes[i]
        return O.ctc_best_path_c(O.hash_logprobs_c(T, W.V_MODEL, seeds[i]), O.hash_labels(S, W.V_MODEL, seeds[i]), 1000, 4, return_total=True)
    O.lib()
    with ThreadPoolExecutor(max_workers=max(1, min(16, (os.cpu_count() or 2) - 1))) as ex:
        want = dict(zip(sample, ex.map(oracle_one, sample)))
    eng = _engine()
    _set(eng, "auto")
    try:
        first = None
        # (all 462 tiled = ~2300 tiles for 1024 workgroup slots: tiles start late, next to producers that are about to finish;
        #  repeated, because what can go wrong there - a progress word that vouches for more than is in memory - is a race)
        for split in ((-1, -1), (n // 3, n // 5), (7, n - 3), (0, n), (n, 0), (n, 0), (n, 0), (n, n // 2)):
            eng.set_split(*split)
            b = DeviceBatch(lps, labs)
            st = b.run(raise_on_error=False)
            assert not (st != 0).any(), split
            for i in sample:
                _compare(b, i, want[i], f"split {split}, chapter {i} (T = {shapes[i][0]})")
            if first is None:
                first = [p.clone() for p in b.path]
                for i in range(n):      # properties on every chapter
                    T, S = shapes[i]
                    p = b.path[i]
                    d = p[1:] - p[:-1]
                    assert int(d.min()) >= 0 and int(d.max()) <= 3 and int(p[-1]) == 2 * S, i
                chain_ok = [np.add.accumulate(b.best_scores[i].cpu().numpy(), dtype=np.float32)[-1].view(np.int32) == np.float32(b.total[i]).view(np.int32)
                            for i in range(0, n, 5)]
                assert all(chain_ok)
            else:
                assert all(torch.equal(a, c) for a, c in zip(first, b.path)), split
    finally:
        eng.set_split(-1, -1)
        _set(eng, "auto")


@pytest.mark.parametrize("width,lds", [(256, 1024), (256, 40 * 1024), (256, 80 * 1024), (128, 1024), (128, 64 * 1024)])
def test_tile_workgroups_per_cu_do_not_change_results(books_on_device, width, lds):
    """ka_debug_set_tile_lds: how much LDS a tile workgroup asks for, i.e. how many share a CU - a request below what the kernel uses
    (27-52 KB by tile width, V and row layout) is raised to that: five 256-position tiles per CU with V = 39, what the library
    picks by itself when a launch's tiles outnumber four per CU; 80 KB keeps two per CU.  Same paths, same scores."""
    from kokoro_align_amd.align import DeviceBatch
    lps, labs = books_on_device("kokoro")
    want = _oracle("kokoro")
    eng = _engine()
    try:
        _set(eng, f"tiled/{width}")
        eng.set_tile_lds(lds)
        b = DeviceBatch(lps, labs)
        b.run()
        _check_against_oracle(b, want)
    finally:
        eng.set_tile_lds(0)
        _set(eng, "auto")


@pytest.mark.parametrize("gather", [0, 1])
def test_serial_backtrace_output_forms_on_a_book(books_on_device, gather):
    """backtrace_rc_kernel<.., GO>: labels and scores collected by the walk in registers (few lattices) or gathered from memory
    after it (launches that fill the chip; ka_debug_set_rc_gather forces either): the same three outputs, bit for bit."""
    from kokoro_align_amd.align import DeviceBatch
    lps, labs = books_on_device("kokoro")
    want = _oracle("kokoro")
    eng = _engine()
    try:
        for mode in ("wave", "tiled"):
            _set(eng, mode)
            eng.set_rc_gather(gather)
            b = DeviceBatch(lps, labs)
            b.run()
            _check_against_oracle(b, want)
    finally:
        eng.set_rc_gather(-1)
        _set(eng, "auto")


def test_halo_sentinel_refill_survives_other_launches_in_between():
    """ka_tiled_stream.hpp's packets vouch for themselves through a NaN sentinel that the engine refills BEHIND a launch's tiles
    for the next launch (ka_engine.hip: refill_halo_sentinel).  Every launch lays its regions out from the start of the same
    workspace, so a launch of another kind in between (one wavefront per lattice, the exact form, the generic kernels) writes
    over the refilled slots: the next tiled launch must fill them itself.  A stale slot would pass for a packet."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    shapes = [(9000, 1500), (7000, 900), (4000, 700)]
    want = O.lattice_batch_c(shapes, W.V_MODEL, 321, 1000, 4, threads=4)
    lps, labs = W.device_book(shapes, seed0=321)
    eng = _engine()
    try:
        tiled = DeviceBatch(lps, labs)
        for between in ("wave", "wave_exact", "tiled/256", "generic", None):
            _set(eng, "tiled/128+parallel")
            tiled.run()
            _check_against_oracle(tiled, want)
            if between == "generic":        # V > 64: the generic kernels, whose byte-per-cell back-pointers cover the front of the workspace
                glp = torch.zeros((3000, 80), dtype=torch.float32, device="cuda")
                glab = torch.arange(1, 301, dtype=torch.int32, device="cuda") % 79 + 1
                _set(eng, "auto")
                DeviceBatch([glp], [glab]).run()
            elif between is not None:
                _set(eng, between)
                other = DeviceBatch(lps, labs)
                other.run()
                _check_against_oracle(other, want)
        _set(eng, "tiled/128+parallel")
        tiled.run()
        _check_against_oracle(tiled, want)
    finally:
        _set(eng, "auto")


def test_a_launch_right_behind_a_big_tiled_one_is_not_hit_by_its_sentinel_refill():
    """The refill of a book's halo region (hundreds of MB, ~0.2 ms, on the engine's own side stream) is still running when the
    launch returns; the next launch - other shapes, so its labels and checkpoints lie where the book's halo slots were - must
    wait for it.  Without the wait its workspace was overwritten under it (a memory fault with four engines, wrong paths here)."""
    import torch
    from kokoro_align_amd import workloads as W
    from kokoro_align_amd.align import DeviceBatch
    big = [(60000 - 700 * i, int(0.14 * (60000 - 700 * i))) for i in range(36)]
    small = [(9000 - 500 * i, 1200 - 60 * i) for i in range(8)]
    want = O.lattice_batch_c(small, W.V_MODEL, 777, 1000, 4, threads=8)
    blp, blab = W.device_book(big, seed0=555)
    slp, slab = W.device_book(small, seed0=777)
    eng = _engine()
    try:
        _set(eng, "tiled/128+parallel")
        book = DeviceBatch(blp, blab)
        book.run()
        assert all(int(p[-1]) == 2 * s for p, (_, s) in zip(book.path, big))
        for mode in ("wave", "tiled/128+parallel", "wave_exact", "auto"):
            _set(eng, "tiled/128+parallel")
            book.run()
            _set(eng, mode)
            other = DeviceBatch(slp, slab)
            other.run()
            _check_against_oracle(other, want)
    finally:
        _set(eng, "auto")
