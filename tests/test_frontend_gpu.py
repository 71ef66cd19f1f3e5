"""Audio front end on the GPU: silence splitting bit-equal to the reference-made
goldens, MFCCs against the float64 restatement of the torchaudio transform the reference calls."""
import os
import wave

import numpy as np
import pytest

from golden_util import g6
from oracle import frontend_oracle as F

pytestmark = pytest.mark.gpu


def test_window_energy_is_numpys_sum_bit_for_bit():
    from kokoro_align_amd import preprocess as P
    x = F.hash_waveform(256 * 4001 + 77, 11, [(0, 400000, 0.4), (400000, 800000, 0.003), (900000, 1024333, 0.9)])
    want = np.mean(x[:256 * 4001].reshape(-1, 256) ** 2, axis=1)
    got = P.window_energy(x, 256)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.int32), want.view(np.int32))
    rng = np.random.default_rng(1)
    y = (rng.standard_normal(256 * 999) * np.exp(rng.uniform(-12, 2, size=256 * 999))).astype(np.float32)
    assert np.array_equal(P.window_energy(y, 256).view(np.int32), np.mean(y.reshape(-1, 256) ** 2, axis=1).view(np.int32))


def test_split_points_equal_the_reference_goldens():
    from kokoro_align_amd import preprocess as P
    g = g6()
    par = g["parameters"]
    for c in g["cases"]:
        x = F.hash_waveform(c["n"], c["seed"], c["pieces"])
        if c["status"] == 1:
            with pytest.raises(ValueError, match="cannot be split"):
                P.get_split_points(x, **par)
            continue
        got = P.get_split_points(x, **par)
        assert [int(v) for v in got] == c["split_points"], c["name"]
    for c in g["silent_ranges"]:
        v = np.array([ch == "1" for ch in c["mask"]])
        if "raises" not in c:
            assert P.get_silent_ranges(v).tolist() == c["ranges"]


def _speechlike(n, seed):
    """noise bursts with a spectral tilt and a noise floor (so that mel levels span ~60 dB, not the full 80)"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    x = np.convolve(x, np.ones(8) / 8.0, mode="same") + 0.02 * rng.standard_normal(n)
    env = np.repeat(rng.uniform(0.05, 1.0, size=n // 2000 + 1), 2000)[:n]
    return (0.3 * x * env).astype(np.float32)


def test_mfcc_of_all_segments_matches_the_float64_restatement():
    """float32 transform on the device (DFT as a GEMM) against the float64 oracle, one segment per oracle call like
    the reference (top_db is relative to the segment's own maximum).  Tolerance 1e-4 (north-star) on speech-like
    material: measured worst 2.3e-5, mean 1.5e-6.  With digital silence inside a segment the frames next to it sit
    close to the -80 dB floor, where a level is the log of a float32 sum of 512 products that nearly cancel:
    measured 1.2e-4, asserted 5e-4."""
    from kokoro_align_amd import preprocess as P
    y = _speechlike(22050 * 12 + 123, 3)
    ends = np.array([30000, 30300, 95000, 95000 + 257, 200017, len(y)], dtype=np.int64)
    got, idx = P.mfcc_segments(y, ends)
    got = got.cpu().numpy()
    assert got.dtype == np.float32
    k, a = 0, 0
    worst, mean = 0.0, []
    for e, stop in zip(ends.tolist(), idx.tolist()):
        want = F.mfcc(y[a:e])
        assert stop - k == want.shape[0] == 1 + (e - a) // 256
        d = np.abs(got[k:stop] - want)
        worst = max(worst, float(d.max()))
        mean.append(float(d.mean()))
        k, a = stop, e
    assert worst < 1e-4 and max(mean) < 1e-5, (worst, mean)
    # digital silence inside a segment: frames at the 1e-10 clamp, floored at max - 80 dB exactly
    z = y.copy()
    z[40000:60000] = 0.0
    got2, _ = P.mfcc_segments(z, np.array([len(z)]))
    want2 = F.mfcc(z)
    assert np.abs(got2.cpu().numpy() - want2).max() < 5e-4
    with pytest.raises(ValueError):
        P.mfcc_segments(y, np.array([100, len(y)]))


def test_split_audio_writes_the_reference_file_formats(tmp_path):
    from kokoro_align_amd import preprocess as P
    c = g6()["cases"][0]
    x = F.hash_waveform(c["n"], c["seed"], c["pieces"])
    pcm = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2")
    wav = str(tmp_path / "ch01.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes(pcm.tobytes())
    seg, mf = str(tmp_path / "ch01.split.txt"), str(tmp_path / "ch01.mfcc.npz")
    P.split_audio(wav, seg, mf)
    xq = pcm.astype(np.float32) / np.float32(32768.0)
    par = g6()["parameters"]
    pts = F.split_points(xq, **par) * par["window_size"]
    ends = [int(v) for v in pts] + [len(xq)]
    assert [int(l) for l in open(seg).read().split()] == ends
    with np.load(mf) as f:
        assert f["indices"].dtype == np.int32 and f["data"].dtype == np.float32 and f["data"].shape[1] == 40
        assert f["indices"].tolist() == np.cumsum([1 + (b - a) // 256 for a, b in zip([0] + ends[:-1], ends)]).tolist()
        a = 0
        k = 0
        for e, stop in zip(ends, f["indices"].tolist()):
            assert np.abs(f["data"][k:stop] - F.mfcc(xq[a:e])).max() < 2e-3
            k, a = stop, e
    np.save(str(tmp_path / "ch02.npy"), xq)
    P.split_audio(str(tmp_path / "ch02.npy"), str(tmp_path / "ch02.split.txt"), str(tmp_path / "ch02.mfcc.npz"))
    assert open(str(tmp_path / "ch02.split.txt")).read() == open(seg).read()
    with pytest.raises(ValueError):
        P.split_audio(str(tmp_path / "x.mp3"), seg, mf)


def test_stage_runner_starts_from_the_waveform(tmp_path):
    """run_example.py:205-275 from the split_audio stage on: decoded audio + voca.txt -> split.txt, mfcc.npz,
    logits, best_path, align.txt, metadata; a second run skips every stage."""
    import torch
    from golden_util import g4
    from kokoro_align_amd import pipeline
    from kokoro_align_amd.model import AudioToChar
    torch.manual_seed(5)
    model = AudioToChar().cuda().eval()
    cases = g6()["cases"]
    audio_files = []
    for i in (0, 1):
        c = cases[i]
        base = str(tmp_path / f"ch{i:02d}")
        np.save(base + ".npy", F.hash_waveform(c["n"], c["seed"], c["pieces"]))
        with open(base + ".voca.txt", "wt") as f:
            f.write(g4()["voca_txt"])
        audio_files.append(base + ".mp3")
    meta = str(tmp_path / "out" / "ds.metadata.txt")
    pipeline.process_alignment("ds", audio_files, meta, model=model, remove_wordsep=False, verbose=False)
    par = g6()["parameters"]
    for i, af in zip((0, 1), audio_files):
        base = af[:-4]
        ends = [int(v) * par["window_size"] for v in cases[i]["split_points"]] + [cases[i]["n"]]
        assert [int(l) for l in open(base + ".split.txt").read().split()] == ends
        with np.load(base + ".mfcc.npz") as f:
            assert len(f["indices"]) == len(ends)
        assert len(open(base + ".align.txt").read().splitlines()) == len(ends)
    assert os.path.exists(meta)
    before = {f: os.path.getmtime(f) for f in [meta] + [a[:-4] + e for a in audio_files for e in (".split.txt", ".mfcc.npz", ".best_path.npz")]}
    pipeline.process_alignment("ds", audio_files, meta, model=None, remove_wordsep=False, verbose=False)
    assert before == {f: os.path.getmtime(f) for f in before}
    with pytest.raises(FileNotFoundError):
        pipeline.process_alignment("ds", [str(tmp_path / "nothing.mp3")], meta + "2", model=model, verbose=False)
