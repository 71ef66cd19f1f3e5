"""GPU parity tests: the HIP path (through the C ABI) against the reference-made goldens and
the CPU oracle.  Integer outputs bit-exact; float outputs bit-exact too (they are gathers of
the inputs; the cumulative score is the same float32 add chain) — north_star allows 1e-4."""
import os
import tempfile

import numpy as np
import pytest

from golden_util import g1_cases, g2_cases, g3_case, g4, sha
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["wave", "wave_exact", "tiled/256", "tiled/128", "wave+parallel", "tiled/256+parallel", "tiled/128+parallel"])
def ka(request):
    """Every test runs in every kernel form: one wavefront per lattice checkpointed / exact,
    four wavefronts per lattice, the tile pipeline with tiles of 256 and of 128 positions - and the checkpointed forms once
    more with the chunk-parallel backtrace forced (the default picks it by batch size).  Results must be identical."""
    import torch
    assert torch.cuda.is_available()
    import kokoro_align_amd as ka
    from kokoro_align_amd import _lib
    assert os.path.exists(ka.library_path()), "HIP library not built"
    mode, _, bt = request.param.partition("+")
    mode, _, width = mode.partition("/")
    eng = _lib.default_engine(torch.cuda.current_device())
    eng.set_mode(mode)
    eng.set_tile_width(int(width or 0))
    eng.set_backtrace(bt or "serial")
    yield ka
    eng.set_mode("auto")
    eng.set_tile_width(0)
    eng.set_backtrace("auto")


def _same(got, want):
    return all(np.array_equal(np.asarray(g).view(np.int32), np.asarray(w).view(np.int32)) for g, w in zip(got, want))


def test_g1_tiny_single_calls(ka):
    for c in g1_cases():
        if c["status"] == 1:
            with pytest.raises(ValueError):
                ka.ctc_best_path(c["lp"], c["labels"], beam_size=c["beam"], max_move=c["max_move"], verbose=False)
            continue
        got = ka.ctc_best_path(c["lp"], c["labels"], beam_size=c["beam"], max_move=c["max_move"], verbose=False)
        assert _same(got, (c["path"], c["best_labels"], c["best_scores"])), (c["idx"], c["T"], c["S"], c["beam"], c["max_move"])


def test_g1_tiny_batched(ka):
    cases = g1_cases()
    groups = {}
    for c in cases:
        groups.setdefault((c["V"], c["beam"], c["max_move"]), []).append(c)
    n = 0
    for (V, beam, mm), cs in groups.items():
        res, status, total = ka.ctc_best_path_batch([c["lp"] for c in cs], [c["labels"] for c in cs], beam, mm,
                                                    return_status=True)
        for c, r, st in zip(cs, res, status):
            if c["status"] == 1:
                assert st == -1, c["idx"]
            else:
                assert st == 0, c["idx"]
                assert _same(r, (c["path"], c["best_labels"], c["best_scores"])), c["idx"]
                n += 1
    assert n > 150


def test_g2_medium(ka):
    for c in g2_cases():
        lp = O.hash_logprobs(c["T"], c["V"], c["seed"])
        labels = O.hash_labels(c["S"], c["V"], c["seed"])
        p, l, s = ka.ctc_best_path(lp, labels, beam_size=c["beam"], max_move=c["max_move"], verbose=False)
        assert np.array_equal(p, c["path"]), c["idx"]
        assert sha(l) == c["sha_labels"] and sha(s) == c["sha_scores"], c["idx"]


def test_g3_cfg2_device_generated(ka):
    """BASELINE configs[1]: inputs generated in HBM by the HIP hash generator, path vs the reference's."""
    import torch
    c = g3_case()
    T, V, S = c["T"], c["V"], c["S"]
    lib = ka.load_library()
    lp = torch.empty((T, V), dtype=torch.float32, device="cuda")
    lab = torch.empty(S, dtype=torch.int32, device="cuda")
    assert lib.ka_hash_logprobs_f32(lp.data_ptr(), T, V, V, c["seed"], None) == 0
    assert lib.ka_hash_labels_i32(lab.data_ptr(), S, V, c["seed"], None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(lp.cpu().numpy(), O.hash_logprobs(T, V, c["seed"]))
    assert np.array_equal(lab.cpu().numpy(), O.hash_labels(S, V, c["seed"]))
    (p, l, s), = ka.ctc_best_path_device([lp], [lab], beam_size=c["beam"], max_move=c["max_move"])
    assert np.array_equal(p.cpu().numpy(), c["path"])
    assert sha(l.cpu().numpy()) == c["sha_labels"] and sha(s.cpu().numpy()) == c["sha_scores"]


def _rand_case(rng, T, V, S, zero_labels=False, ties=False, ninf=False):
    lp = rng.standard_normal((T, V)).astype(np.float32)
    lp = lp - np.log(np.sum(np.exp(lp), axis=-1, keepdims=True))
    if ties:
        lp = (np.round(lp * 2) / 2).astype(np.float32)
    if ninf:
        lp = np.where(rng.random((T, V)) < 0.1, -np.inf, lp).astype(np.float32)
    labels = rng.integers(0 if zero_labels else 1, V, size=S).astype(np.int32)
    return lp, labels


@pytest.mark.parametrize("mm", [1, 2, 3, 4])
def test_random_vs_oracle_fast_path(ka, mm):
    rng = np.random.default_rng(100 + mm)
    lps, labs, want = [], [], []
    for i in range(24):
        V = int(rng.choice([5, 39, 64]))
        S = int(rng.integers(1, 1500))
        T = int(rng.integers(max(1, (2 * S + 1) // 3), 3 * S + 50))
        lp, labels = _rand_case(rng, T, V, S, zero_labels=(i % 3 == 0), ties=(i % 4 == 1), ninf=(i % 6 == 5))
        try:
            w = O.ctc_best_path_c(lp, labels, 1000, mm, return_total=True)
        except ValueError:
            w = None
        lps.append(lp); labs.append(labels); want.append(w)
    by_v = {}
    for i, lp in enumerate(lps):
        by_v.setdefault(lp.shape[1], []).append(i)
    for V, idxs in by_v.items():
        res, status, total = ka.ctc_best_path_batch([lps[i] for i in idxs], [labs[i] for i in idxs], 1000, mm,
                                                    return_status=True)
        for j, i in enumerate(idxs):
            if want[i] is None:
                assert status[j] == -1
            else:
                assert status[j] == 0
                assert _same(res[j], want[i][:3]), (i, lps[i].shape, labs[i].shape)
                assert np.float32(total[j]).view(np.int32) == np.float32(want[i][3]).view(np.int32)


def test_band_edges_and_long_transcripts(ka):
    """lo/hi sliding over many 16-position blocks, wrap of the 1024 slots, several beams."""
    rng = np.random.default_rng(7)
    for T, V, S, beam in [(9000, 39, 4000, 1000), (7000, 64, 3000, 1009), (6000, 39, 2500, 333),
                          (4000, 20, 5500, 1000), (2000, 39, 2900, 800), (12000, 39, 1500, 1000)]:
        lp, labels = _rand_case(rng, T, V, S)
        try:
            want = O.ctc_best_path_c(lp, labels, beam, 4)
        except ValueError:
            with pytest.raises(ValueError):
                ka.ctc_best_path(lp, labels, beam_size=beam, verbose=False)
            continue
        got = ka.ctc_best_path(lp, labels, beam_size=beam, verbose=False)
        assert _same(got, want), (T, V, S, beam)


def test_degenerate_jumps(ka):
    """L/T far above 3: the band outruns every state (ValueError), or T tiny."""
    rng = np.random.default_rng(8)
    for T, V, S, beam in [(10, 39, 5000, 1000), (3, 39, 5000, 1000), (40, 8, 2000, 1000), (100, 8, 1700, 1000),
                          (2, 5, 600, 1000), (700, 8, 1100, 50)]:
        lp, labels = _rand_case(rng, T, V, S)
        try:
            want = O.ctc_best_path_c(lp, labels, beam, 4)
        except ValueError:
            with pytest.raises(ValueError):
                ka.ctc_best_path(lp, labels, beam_size=beam, verbose=False)
            continue
        got = ka.ctc_best_path(lp, labels, beam_size=beam, verbose=False)
        assert _same(got, want), (T, V, S, beam)


def test_generic_path(ka):
    """Arguments outside the w16 layout: V > 64, beam > 1009 on a long transcript, max_move > 4."""
    rng = np.random.default_rng(9)
    for T, V, S, beam, mm in [(600, 100, 300, 1000, 4), (900, 39, 1200, 2000, 4), (500, 39, 200, 1000, 6),
                              (800, 130, 900, 5000, 5), (300, 39, 700, 100000, 4)]:
        lp, labels = _rand_case(rng, T, V, S, zero_labels=True)
        try:
            want = O.ctc_best_path_c(lp, labels, beam, mm)
        except ValueError:
            with pytest.raises(ValueError):
                ka.ctc_best_path(lp, labels, beam_size=beam, max_move=mm, verbose=False)
            continue
        got = ka.ctc_best_path(lp, labels, beam_size=beam, max_move=mm, verbose=False)
        assert _same(got, want), (T, V, S, beam, mm)


def test_errors(ka):
    lp = np.zeros((10, 5), np.float32)
    with pytest.raises(IndexError):
        ka.ctc_best_path(lp, np.array([1, 7], np.int32), verbose=False)      # label >= V
    with pytest.raises(IndexError):
        ka.ctc_best_path(np.zeros((0, 5), np.float32), np.array([1], np.int32), verbose=False)  # T = 0
    with pytest.raises(ValueError):
        ka.ctc_best_path(lp, np.array([1, 2], np.int32), beam_size=0, verbose=False)   # empty band
    p, l, s = ka.ctc_best_path(lp, np.zeros(0, np.int32), verbose=False)     # S = 0: all blank
    assert np.array_equal(p, np.zeros(10, np.int32))


def test_float64_and_plus_inf_inputs_behave_as_documented(ka):
    """kokoro-align_amd/align.py (docstring of ctc_best_path).  float64 log-probs: cast to float32 FIRST (the reference adds in
    float64 and rounds the sum into its float32 score array, align.py:77, and returns float64 best_scores) - the result is that of
    the float32 array, best_scores float32.  +inf log-probs: the scores-only forms hand the lattice to the exact kernels, whose
    add-then-compare gives what the reference gives (oracle == reference on such input, checked in the dev container) as long as
    +inf never meets -inf."""
    rng = np.random.default_rng(77)
    lp64 = -rng.random((300, 9)) * 8.0
    labels = rng.integers(1, 9, 60).astype(np.int32)
    got = ka.ctc_best_path(lp64, labels, verbose=False)
    want = O.ctc_best_path_c(lp64.astype(np.float32), labels, 1000, 4)
    assert got[2].dtype == np.float32 and _same(got, want)
    for trial in range(12):
        T, V, S = int(rng.integers(20, 400)), int(rng.integers(3, 40)), int(rng.integers(1, 60))
        lp = (-rng.random((T, V)) * 8).astype(np.float32)
        for _ in range(int(rng.integers(1, 4))):
            lp[rng.integers(0, T), rng.integers(0, V)] = np.inf
        lab = rng.integers(1, V, S).astype(np.int32)
        try:
            want = O.ctc_best_path_c(lp, lab, 1000, 4)
        except ValueError:
            with pytest.raises(ValueError):
                ka.ctc_best_path(lp, lab, verbose=False)
            continue
        assert _same(ka.ctc_best_path(lp, lab, verbose=False), want), trial


def test_strided_device_input(ka):
    import torch
    rng = np.random.default_rng(10)
    lp, labels = _rand_case(rng, 700, 39, 300)
    want = O.ctc_best_path_c(lp, labels, 1000, 4)
    wide = torch.zeros((700, 64), dtype=torch.float32, device="cuda")
    wide[:, :39] = torch.from_numpy(lp).cuda()
    view = wide[:, :39]                       # row stride 64, V = 39
    (p, l, s), = ka.ctc_best_path_device([view], [labels])
    assert _same((p.cpu().numpy(), l.cpu().numpy(), s.cpu().numpy()), want)


def test_rows_that_span_more_than_4_gb_go_to_the_exact_kernels(ka):
    """forward_ck addresses the log-probs with 32-bit byte offsets (a buffer descriptor); a lattice whose rows span 4 GB or more
    is declined there and done by the exact kernels, whose addresses are 64-bit: same result, status ok."""
    import torch
    rng = np.random.default_rng(12)
    T, V, S = 130, 39, 40
    lp, labels = _rand_case(rng, T, V, S)
    want = O.ctc_best_path_c(lp, labels, 1000, 4)
    ld = (1 << 23) + 64                        # floats between two rows: 130 rows x 32 MB = 4.36 GB
    assert T * ld * 4 > (1 << 32)
    wide = torch.zeros((T, ld), dtype=torch.float32, device="cuda")
    wide[:, :V] = torch.from_numpy(lp).cuda()
    (p, l, s), = ka.ctc_best_path_device([wide[:, :V]], [labels])
    assert _same((p.cpu().numpy(), l.cpu().numpy(), s.cpu().numpy()), want)
    del wide
    torch.cuda.empty_cache()


def test_log_softmax_kernel(ka):
    import torch
    rng = np.random.default_rng(11)
    for V in (39, 64, 100):
        x = (rng.standard_normal((513, V)) * 3).astype(np.float32)
        c = x - np.mean(x, axis=-1, keepdims=True)
        want = c - np.log(np.sum(np.exp(c), axis=-1, keepdims=True))
        got = ka.log_softmax_device(torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.allclose(got, want, rtol=0, atol=2e-6 * max(1.0, np.abs(want).max()))


def test_file_round_trip_matches_reference(ka):
    """best_path() -> align() on files, against the text the reference wrote (g4)."""
    g = g4()
    with tempfile.TemporaryDirectory() as td:
        voca = os.path.join(td, "x.voca.txt")
        with open(voca, "wt") as f:
            f.write(g["voca_txt"])
        for name in ("rt_a", "rt_b", "rt_c"):
            rt = g[name]
            T, segs = rt["T"], rt["segments"]
            logits = O.hash_logprobs(T, 39, rt["logits_seed"]) + np.float32(4.0)
            labels = np.array(g["read_transcript"], np.int64)
            S = labels.shape[0]
            ext = np.zeros(2 * S + 1, np.int64); ext[1::2] = labels
            logits[np.arange(T), ext[np.arange(T) * (2 * S + 1) // T]] += 4.0
            assert sha(logits) == rt["logits_sha"]
            lf, mf, bf = (os.path.join(td, f"{name}.{e}.npz") for e in ("logits", "mfcc", "best_path"))
            np.savez(lf, indices=np.array(segs, np.int32), data=logits)
            np.savez(mf, indices=np.array(segs, np.int32), data=np.zeros((T, 1), np.float32))
            ka.best_path(lf, voca, bf)
            with np.load(bf) as f:
                assert {k: str(f[k].dtype) for k in f.files} == rt["dtypes"]
                assert f["best_path"].tolist() == rt["best_path"]
                assert f["best_labels"].tolist() == rt["best_labels"]
                assert np.allclose(f["best_scores"], np.array(rt["best_scores"], np.float32), atol=1e-4)
                same_scores = np.array_equal(f["best_scores"], np.array(rt["best_scores"], np.float32))
            for rw in (True, False):
                af = os.path.join(td, f"{name}.{int(rw)}.align.txt")
                ka.align(bf, mf, voca, af, rw)
                got = open(af).read().splitlines()
                want = rt[f"align_txt_{int(rw)}"].splitlines()
                assert len(got) == len(want)
                # The two float fields are sums of best_scores, which come from the HOST log-softmax (NumPy's exp / log: ulp-level
                # platform dependent).  Where this host's NumPy reproduces the reference's best_scores bit
                # for bit the file must be the reference's byte for byte; elsewhere the north-star tolerance applies.
                if same_scores:
                    assert got == want
                for gl, wl in zip(got, want):
                    gp, wp = gl.split("|"), wl.split("|")
                    assert gp[:5] == wp[:5]
                    assert abs(float(gp[5]) - float(wp[5])) < 1e-4 and abs(float(gp[6]) - float(wp[6])) < 1e-4
            # device log-softmax variant gives the same path
            bf2 = os.path.join(td, f"{name}.dev.best_path.npz")
            ka.best_path(lf, voca, bf2, device_softmax=True)
            with np.load(bf2) as f:
                assert f["best_path"].tolist() == rt["best_path"]


def test_repeatability_tiny_lattices(ka):
    """Regression for a start-up race (a log-prob row used before its prefetch had landed): it showed up
    on tiny lattices about once in 60 runs.  Every g1 case, many times, single and batched calls."""
    cases = [c for c in g1_cases() if c["status"] == 0 and c["T"] <= 80]
    for rep in range(25):
        for c in cases[rep % 3::3]:
            p, l, s = ka.ctc_best_path(c["lp"], c["labels"], beam_size=c["beam"], max_move=c["max_move"], verbose=False)
            assert np.array_equal(p, c["path"]), (rep, c["idx"])


def test_many_lattices_under_memory_pressure(ka):
    """Regression: with hundreds of lattices in flight the log-prob prefetches land late; a prefetch that was
    still in flight after the last frame used to overwrite registers of the end-position reduction
    (about 2 % of the lattices ended one to a few positions low).  Every end must be the trailing blank
    and sampled lattices must equal the oracle."""
    import torch
    B, T, V, S = 768, 6000, 64, 600
    lib = ka.load_library()
    lps = torch.empty((B, T, V), dtype=torch.float32, device="cuda")
    labs = torch.empty((B, S), dtype=torch.int32, device="cuda")
    assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, 7000, None) == 0
    assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 7000, None) == 0
    torch.cuda.synchronize()
    from kokoro_align_amd.align import DeviceBatch
    batch = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)])
    for rep in range(3):
        batch.run()
        ends = torch.stack([p[-1] for p in batch.path]).cpu().numpy()
        assert (ends == 2 * S).all(), np.nonzero(ends != 2 * S)[0][:10]
    for i in (0, 1, B // 2, B - 1):
        want = O.ctc_best_path_c(O.hash_logprobs(T, V, 7000 + i), O.hash_labels(S, V, 7000 + i))
        assert np.array_equal(batch.path[i].cpu().numpy(), want[0]), i
        assert np.array_equal(batch.best_labels[i].cpu().numpy(), want[1]), i


def test_full_occupancy_batch_sampled_against_the_oracle(ka):
    """Regression: 4096 lattices (4 wavefronts per SIMD) and a random sample of them against the oracle, all three
    outputs.  The checkpointed forward kernel once staged its 16-byte checkpoint stores through one set of
    registers with a single wait state between a store and the next write of its data registers; gfx950 needs
    two, and only under load did the first dword of a group come out as the next group's - every end position
    was right, lattice 0 was right, and about half of the other paths differed from the reference for a few
    thousand frames."""
    import torch
    B, T, V, S = 4096, 4000, 64, 400
    lib = ka.load_library()
    lps = torch.empty((B, T, V), dtype=torch.float32, device="cuda")
    labs = torch.empty((B, S), dtype=torch.int32, device="cuda")
    assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, 9000, None) == 0
    assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 9000, None) == 0
    torch.cuda.synchronize()
    from kokoro_align_amd.align import DeviceBatch
    batch = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)])
    batch.run()
    batch.run()
    # every lattice, without the oracle: the float32 chain of the per-frame scores along the returned path is the
    # forward pass's best cumulative score, bit for bit (each score on the best path IS fl(previous + emission))
    chain = np.add.accumulate(torch.stack(batch.best_scores).cpu().numpy(), axis=1, dtype=np.float32)[:, -1]
    assert np.array_equal(chain.view(np.int32), np.asarray(batch.total, np.float32).view(np.int32))
    sample = sorted(set([0, B - 1] + np.random.default_rng(4).integers(0, B, size=30).tolist()))
    for i in sample:
        want = O.ctc_best_path_c(O.hash_logprobs(T, V, 9000 + i), O.hash_labels(S, V, 9000 + i))
        assert np.array_equal(batch.path[i].cpu().numpy(), want[0]), i
        assert np.array_equal(batch.best_labels[i].cpu().numpy(), want[1]), i
        assert np.array_equal(batch.best_scores[i].cpu().numpy().view(np.int32), want[2].view(np.int32)), i


@pytest.mark.parametrize("mm", [3, 4])
def test_full_occupancy_batch_with_label_zero_and_strided_rows(ka, mm):
    """The same at 4096 lattices for the kernel instances the previous test does not reach: transcripts that contain
    label 0 (the zero-label instances run beside the ordinary ones), V = 39 columns in rows of 64 floats, max_move 3
    and 4, beam 700."""
    import torch
    B, T, V, S, LD = 4096, 3000, 39, 350, 64
    lib = ka.load_library()
    lps = torch.full((B, T, LD), float("nan"), dtype=torch.float32, device="cuda")     # the padding columns are never read
    labs = torch.empty((B, S), dtype=torch.int32, device="cuda")
    assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, LD, T * LD, 9500, None) == 0
    assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 9500, None) == 0
    labs[::2, ::7] = 0            # every other lattice has label 0 in its transcript
    torch.cuda.synchronize()
    from kokoro_align_amd.align import DeviceBatch
    batch = DeviceBatch([lps[i][:, :V] for i in range(B)], [labs[i] for i in range(B)], 700, mm)
    batch.run()
    batch.run()
    sample = sorted(set([0, 1, B - 2, B - 1] + np.random.default_rng(5).integers(0, B, size=24).tolist()))
    for i in sample:
        lab = O.hash_labels(S, V, 9500 + i)
        if i % 2 == 0:
            lab[::7] = 0
        want = O.ctc_best_path_c(O.hash_logprobs(T, V, 9500 + i), lab, 700, mm)
        assert np.array_equal(batch.path[i].cpu().numpy(), want[0]), i
        assert np.array_equal(batch.best_labels[i].cpu().numpy(), want[1]), i
        assert np.array_equal(batch.best_scores[i].cpu().numpy().view(np.int32), want[2].view(np.int32)), i


def test_cfg2_batch_at_full_size_properties(ka):
    """1024 lattices of BASELINE configs[1] (T = 50 000, V = 64, S = 5 000, beam 1000, max_move 4), every one checked
    through size-independent properties: the path ends on the trailing blank, never goes back and never moves more
    than 3, best_labels is the expanded transcript read along the path, best_scores are the log-probs read at those
    labels, and their float32 running sum is the forward pass's best cumulative score bit for bit."""
    import torch
    B, T, V, S = 1024, 50000, 64, 5000
    lib = ka.load_library()
    lps = torch.empty((B, T, V), dtype=torch.float32, device="cuda")
    labs = torch.empty((B, S), dtype=torch.int32, device="cuda")
    assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, 100, None) == 0
    assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 100, None) == 0
    torch.cuda.synchronize()
    from kokoro_align_amd.align import DeviceBatch
    batch = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)], 1000, 4)
    batch.run()
    path = torch.stack(batch.path)                       # [B, T] on the device
    assert bool((path[:, -1] == 2 * S).all()) and bool((path[:, 0] >= 0).all()) and bool((path[:, 0] <= 3).all())
    step = path[:, 1:] - path[:, :-1]
    assert bool((step >= 0).all()) and bool((step <= 3).all())
    expanded = torch.zeros((B, 2 * S + 1), dtype=torch.int32, device="cuda")
    expanded[:, 1::2] = labs
    want_labels = torch.gather(expanded, 1, path.long())
    got_labels = torch.stack(batch.best_labels)
    assert bool((got_labels == want_labels).all())
    want_scores = torch.gather(lps, 2, got_labels.long().unsqueeze(2)).squeeze(2)
    got_scores = torch.stack(batch.best_scores)
    assert bool((got_scores.view(torch.int32) == want_scores.view(torch.int32)).all())
    chain = np.add.accumulate(got_scores.cpu().numpy(), axis=1, dtype=np.float32)[:, -1]
    assert np.array_equal(chain.view(np.int32), np.asarray(batch.total, np.float32).view(np.int32))


def test_cfg5_long_form_band(ka):
    """BASELINE configs[4] with the default band: T=500000 x V=64, S=50000 (L=100001); 128 MB of back-pointers."""
    import torch
    T, V, S = 500000, 64, 50000
    lib = ka.load_library()
    lp = torch.empty((T, V), dtype=torch.float32, device="cuda")
    lab = torch.empty(S, dtype=torch.int32, device="cuda")
    assert lib.ka_hash_logprobs_f32(lp.data_ptr(), T, V, V, 55, None) == 0
    assert lib.ka_hash_labels_i32(lab.data_ptr(), S, V, 55, None) == 0
    torch.cuda.synchronize()
    (p, l, s), = ka.ctc_best_path_device([lp], [lab])
    want = O.ctc_best_path_c(O.hash_logprobs(T, V, 55), O.hash_labels(S, V, 55))
    assert np.array_equal(p.cpu().numpy(), want[0])
    assert np.array_equal(l.cpu().numpy(), want[1])
    assert np.array_equal(s.cpu().numpy().view(np.int32), want[2].view(np.int32))


def test_host_strided_rows_and_engine_reuse(ka):
    """Host buffers whose row stride exceeds V go through the C ABI directly; one engine serves calls of
    very different sizes back to back (workspace growth) and an explicit reserve."""
    import ctypes
    from kokoro_align_amd import _lib
    rng = np.random.default_rng(21)
    eng = _lib.Engine(0)
    eng.set_mode("auto")
    try:
        for T, V, ld, S in [(300, 39, 48, 120), (9000, 39, 39, 3000), (50, 5, 64, 10), (4000, 64, 80, 1500)]:
            buf = np.full((T, ld), np.nan, np.float32)          # NaN padding must never be read
            lp = rng.standard_normal((T, V)).astype(np.float32)
            buf[:, :V] = lp
            labels = rng.integers(1, V, size=S).astype(np.int32)
            want = O.ctc_best_path_c(lp, labels, 1000, 4, return_total=True)
            path = np.empty(T, np.int32); lout = np.empty(T, np.int32); sout = np.empty(T, np.float32)
            total = ctypes.c_float(0)
            rc = eng.lib.ka_ctc_best_path_f32(eng.handle, buf.ctypes.data, T, V, ld, labels.ctypes.data, S, 1000, 4,
                                              path.ctypes.data, lout.ctypes.data, sout.ctypes.data,
                                              ctypes.addressof(total), _lib.KA_MEM_HOST, None)
            assert rc == 0, _lib.last_error()
            assert np.array_equal(path, want[0]) and np.array_equal(lout, want[1])
            assert np.array_equal(sout.view(np.int32), want[2].view(np.int32))
            assert np.float32(total.value).view(np.int32) == np.float32(want[3]).view(np.int32)
        eng.reserve(64 << 20)
        n = (ctypes.c_int64 * 1)(1000); s = (ctypes.c_int64 * 1)(100)
        assert eng.lib.ka_workspace_bytes(1, n, s, 39, 1000, 4) >= 1000 * 256
        assert eng.lib.ka_workspace_bytes(1, n, s, 39, 1000, 300) == 0        # max_move out of range -> 0
    finally:
        eng.close()


def test_non_default_stream_and_two_engines(ka):
    """Device buffers on a side stream; a second engine in another host thread at the same time."""
    import threading
    import torch
    from kokoro_align_amd import _lib
    rng = np.random.default_rng(22)
    lp, labels = _rand_case(rng, 5000, 39, 1800)
    want = O.ctc_best_path_c(lp, labels, 1000, 4)
    results = {}

    def worker(name):
        eng = _lib.Engine(0)
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                d = torch.from_numpy(lp).cuda()
                from kokoro_align_amd.align import DeviceBatch
                b = DeviceBatch([d] * 3, [torch.from_numpy(labels).cuda()] * 3)
                b.engine = eng
                for _ in range(3):
                    b.run()
                results[name] = [x.cpu().numpy() for x in b.path]
        finally:
            eng.close()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for name in (0, 1):
        assert all(np.array_equal(p, want[0]) for p in results[name]), name


def test_checkpointed_form_hands_over_what_it_cannot_do(ka):
    """The scores-only forward kernel is valid for finite log-probs of sane magnitude; lattices with -inf
    entries or absurd magnitudes must come out identical through the exact kernels, in the same batch as
    lattices that stay on the checkpointed path (they share one workspace region per lattice)."""
    rng = np.random.default_rng(77)
    lps, labs = [], []
    for i in range(9):
        T, V, S = 700 + 37 * i, 39, 150 + 11 * i
        lp = np.log(rng.dirichlet(np.ones(V) * 0.3, size=T)).astype(np.float32)
        if i % 3 == 1:
            lp[rng.random((T, V)) < 0.05] = -np.inf          # a state can be live with score -inf
        if i % 3 == 2:
            lp[rng.integers(0, T), rng.integers(0, V)] = -3e31   # finite, but beyond the magnitude the check allows
        lps.append(lp)
        labs.append(rng.integers(1, V, size=S).astype(np.int32))
    res, status, total = ka.ctc_best_path_batch(lps, labs, 1000, 4, return_status=True)
    for i, (lp, lab) in enumerate(zip(lps, labs)):
        try:
            w = O.ctc_best_path_c(lp, lab, 1000, 4, return_total=True)
        except ValueError:
            assert status[i] == -1, i
            continue
        assert status[i] == 0, i
        assert _same(res[i], w[:3]), i
        assert np.float32(total[i]).view(np.int32) == np.float32(w[3]).view(np.int32), i


def test_nan_log_probs_are_rejected_not_aligned(ka):
    """A NaN log-prob is an explicit per-lattice error (KA_ERR_NAN -> ValueError), in a single call and inside a
    batch whose other lattices still come out right.  (The library is built with -fno-honor-nans: a float compare
    would let the NaN through, the kernels test the bits.)"""
    rng = np.random.default_rng(5)
    T, V, S = 900, 39, 200
    lps = [np.log(rng.dirichlet(np.ones(V), size=T)).astype(np.float32) for _ in range(4)]
    labs = [rng.integers(1, V, size=S).astype(np.int32) for _ in range(4)]
    lps[2][T // 2, 7] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        ka.ctc_best_path(lps[2], labs[2], verbose=False)
    res, status, total = ka.ctc_best_path_batch(lps, labs, 1000, 4, return_status=True)
    assert status == [0, 0, -6, 0]
    for i in (0, 1, 3):
        assert _same(res[i], O.ctc_best_path_c(lps[i], labs[i], 1000, 4))


@pytest.mark.parametrize("mm", [1, 2, 3, 4])
def test_random_small_shapes_and_narrow_beams(ka, mm):
    """Many small lattices with odd shapes: T not a multiple of the checkpoint interval, beams narrower than the
    recompute window (its cells straddle both band edges), L >> T and L << T, S = 0, V = 1."""
    rng = np.random.default_rng(900 + mm)
    groups = {}
    for i in range(160):
        V = int(rng.choice([1, 2, 5, 64]))
        beam = int(rng.choice([1, 2, 5, 16, 64, 1000]))
        S = int(rng.integers(0, 150))
        T = int(rng.integers(1, 400))
        lp = (rng.standard_normal((T, V)) * 3).astype(np.float32)
        if i % 5 == 0:
            lp = np.round(lp)                                  # exact ties
        labels = rng.integers(0 if (V > 1 and i % 4 == 0) or V == 1 else 1, V, size=S).astype(np.int32)
        groups.setdefault((V, beam), []).append((lp, labels))
    n_ok = 0
    for (V, beam), cases in groups.items():
        res, status, total = ka.ctc_best_path_batch([c[0] for c in cases], [c[1] for c in cases], beam, mm,
                                                    return_status=True)
        for j, (lp, labels) in enumerate(cases):
            try:
                w = O.ctc_best_path_c(lp, labels, beam, mm, return_total=True)
            except ValueError:
                assert status[j] == -1, (V, beam, j, lp.shape, labels.shape)
                continue
            assert status[j] == 0, (V, beam, j, lp.shape, labels.shape)
            assert _same(res[j], w[:3]), (V, beam, j, lp.shape, labels.shape)
            assert np.float32(total[j]).view(np.int32) == np.float32(w[3]).view(np.int32)
            n_ok += 1
    assert n_ok > 20
