"""End-to-end device pipeline on the GPU: AudioToChar (PyTorch-ROCm) -> HIP log-softmax -> batched HIP DP
-> align.txt -> metadata.txt, checked against the CPU oracle fed with the same logits."""
import os

import numpy as np
import pytest

from golden_util import g4, g5, g7
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_audio_to_char_on_gpu_matches_reference(tmp_path):
    import torch
    from kokoro_align_amd.model import AudioToChar, segment_logits
    a = g5()["audio_to_char"]
    model = AudioToChar(**a["params"])
    model.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in a["state_dict"].items()})
    model = model.cuda().eval()
    segs = [O.hash_logprobs(n, 40, seed) + np.float32(4.0) for n, seed in zip(a["segment_lens"], a["segment_seeds"])]
    from kokoro_align_amd.model import segment_logits_device
    for got in (segment_logits(model, segs),            # PyTorch-ROCm (MIOpen) vs the reference's CPU LSTM
                segment_logits_device(model, segs),     # library GEMMs + ka_lstm_layer_f32 vs the same golden logits
                segment_logits_device(model, segs, persistent=False)):   # ... + per-step GEMM and ka_lstm_step_f32
        for g, w in zip(got, a["logits"]):
            assert g.is_cuda
            assert np.allclose(g.cpu().numpy(), np.array(w, np.float32), atol=1e-4, rtol=0)


def test_persistent_lstm_kernel_matches_the_reference_at_default_params():
    """g7 (reference-made: AudioToChar(**DEFAULT_PARAMS), train.py:16-20, :54-65, :92-95; 19 ragged segments of 1..300 frames =
    two 16-sequence tiles): the hand-written MFMA kernel ka_lstm_layer_f32 / ka_lstm_layer0_f32 - what the pipeline runs - against
    the logits the reference computed.  North-star tolerance 1e-4."""
    import torch
    from kokoro_align_amd.model import AudioToChar, lstm_logits_device, segment_logits_device
    g = g7()
    model = AudioToChar()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g["state"].items()})
    model = model.cuda().eval()
    assert model.hidden_dim == 128            # the size that takes the persistent path (model.py: persistent and H == 128)
    segs = [O.hash_logprobs(n, 40, sd) * np.float32(g["scale"]) + np.float32(g["offset"]) for n, sd in zip(g["lens"], g["seeds"])]
    worst = {}
    for name, kw in (("persistent, layer-0 projection inside", dict(persistent=True)), ("per step", dict(persistent=False))):
        got = segment_logits_device(model, segs, **kw)
        assert len(got) == len(g["logits"])
        worst[name] = max(float(np.max(np.abs(a.cpu().numpy() - w))) for a, w in zip(got, g["logits"]))
    # ... and with the layer-0 projection as a library GEMM in front of the persistent kernel
    data = np.concatenate(segs, 0)
    c = lstm_logits_device(model, data, np.cumsum(g["lens"]), persistent=True, fuse_layer0=False).cpu().numpy()
    worst["persistent, projection as a GEMM"] = float(np.max(np.abs(c - np.concatenate(g["logits"], 0))))
    assert all(v <= 1e-4 for v in worst.values()), worst


def test_process_alignment_end_to_end(tmp_path):
    import torch
    from kokoro_align_amd import pipeline
    from kokoro_align_amd.model import AudioToChar
    from kokoro_align_amd.transcript import read_transcript
    torch.manual_seed(5)
    model = AudioToChar().cuda().eval()
    voca_txt = g4()["voca_txt"]
    audio_files = []
    for i, seg_lens in enumerate([[120, 300, 90], [400, 250], [77, 88, 99, 111]]):
        base = str(tmp_path / f"ch{i:02d}")
        audio_files.append(base + ".mp3")
        with pipeline.open_index_data_for_write(base + ".mfcc.npz") as w:
            for j, n in enumerate(seg_lens):
                w.write(O.hash_logprobs(n, 40, 50 * i + j) * np.float32(0.5) + np.float32(2.0))
        with open(base + ".split.txt", "wt") as f:
            for j in range(len(seg_lens)):
                f.write(f"{(j + 1) * 40000}\n")
        with open(base + ".voca.txt", "wt") as f:
            f.write(voca_txt)
    meta = str(tmp_path / "out" / "ds.metadata.txt")
    pipeline.process_alignment("ds", audio_files, meta, model=model, remove_wordsep=False, verbose=False)
    for af in audio_files:
        base = af[:-4]
        with np.load(base + ".logits.npz") as f:
            logits, idx = f["data"], f["indices"]
        with np.load(base + ".best_path.npz") as f:
            got = (f["best_path"], f["best_labels"], f["best_scores"])
            assert {k: str(f[k].dtype) for k in f.files} == {"best_path": "int32", "best_labels": "int32", "best_scores": "float32"}
        # the DP itself must be exact given the device log-probs; re-derive them on the host (1e-6) and allow
        # the path to differ only if the oracle on host log-probs differs too (it never should here)
        c = logits - np.mean(logits, axis=-1, keepdims=True)
        lp = c - np.log(np.sum(np.exp(c), axis=-1, keepdims=True))
        want = O.ctc_best_path_c(lp, read_transcript(base + ".voca.txt"))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        assert np.allclose(got[2], want[2], atol=1e-4)
        lines = open(base + ".align.txt").read().splitlines()
        assert len(lines) == len(idx) and [int(x.split("|")[0]) for x in lines] == idx.tolist()
        assert os.path.exists(base + ".greed.txt")
    assert os.path.exists(meta)
    # second run: every stage is skipped (outputs exist), nothing is rewritten
    before = {f: os.path.getmtime(f) for f in [meta] + [a[:-4] + ".best_path.npz" for a in audio_files]}
    pipeline.process_alignment("ds", audio_files, meta, model=None, remove_wordsep=False, verbose=False)
    assert before == {f: os.path.getmtime(f) for f in before}


@pytest.mark.gpu
def test_device_lstm_matches_the_reference_network():
    """segment_logits_device (library GEMMs + the fused HIP step kernel, all segments at once) against the
    PyTorch CPU network of the reference's architecture (train.py:54-65) with the same weights: float32
    rounding only (north-star tolerance 1e-4), ragged segment lengths, original order kept."""
    import torch
    from kokoro_align_amd.model import AudioToChar, segment_logits, segment_logits_device
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    cpu = AudioToChar().eval()
    segs = [rng.standard_normal((int(n), 40)).astype(np.float32) for n in [1, 7, 300, 64, 2, 129, 300, 511, 33]]
    want = segment_logits(cpu, segs, device="cpu")
    gpu = AudioToChar().eval()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda()
    for persistent in (True, False):
        got = segment_logits_device(gpu, segs, persistent=persistent)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g.shape == w.shape
            np.testing.assert_allclose(g.cpu().numpy(), w.numpy(), rtol=0, atol=1e-4)


@pytest.mark.gpu
def test_persistent_lstm_many_tiles_and_empty_segments():
    """More sequences than one 32-row tile, lengths that straddle tile boundaries, empty segments in the index
    table (they own no rows), against the per-step device path (itself checked against the CPU network above)."""
    import torch
    from kokoro_align_amd.model import AudioToChar, lstm_logits_device
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    model = AudioToChar().cuda().eval()
    lens = rng.integers(0, 90, size=150)
    lens[[0, 17, 149]] = 0
    lens[40] = 257
    data = rng.standard_normal((int(lens.sum()) + 5, 40)).astype(np.float32)   # rows past the last index are ignored
    a = lstm_logits_device(model, data, np.cumsum(lens), persistent=True)
    b = lstm_logits_device(model, data, np.cumsum(lens), persistent=False)
    assert a.shape == b.shape == (int(lens.sum()), 39)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-4)
    # layer 0 with its input projection inside the recurrence kernel (the default) against the library GEMM in front of it
    c = lstm_logits_device(model, data, np.cumsum(lens), persistent=True, fuse_layer0=False)
    np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=0, atol=2e-5)


def test_best_path_stage_keeps_per_file_progress_when_a_chapter_fails(tmp_path):
    """run_example.py:248-254 loops the files: the `*.best_path.npz` written before a failing chapter stay written and are
    skipped on the rerun.  Here all chapters are ONE launch, so the stage writes every chapter whose status is 0 and THEN
    raises what the reference's best_path() raises for the first bad one (ValueError, align.py:101: empty beam)."""
    import torch
    from kokoro_align_amd import pipeline
    voca_txt = g4()["voca_txt"]
    logits, voca, out = [], [], []
    for i, (T, reps) in enumerate([(900, 1), (700, 1), (20, 6), (1100, 1), (800, 1)]):   # chapter 2: 20 frames for ~600 phonemes
        base = str(tmp_path / f"ch{i:02d}")
        with pipeline.open_index_data_for_write(base + ".logits.npz") as w:
            w.write(O.hash_logprobs(T, 39, 900 + i) + np.float32(4.0))
        with open(base + ".voca.txt", "wt") as f:
            f.write(voca_txt * reps)
        logits.append(base + ".logits.npz"); voca.append(base + ".voca.txt"); out.append(base + ".best_path.npz")
    with pytest.raises(ValueError, match="argmax of an empty sequence"):
        pipeline.best_path_files(logits, voca, out)
    assert [os.path.exists(f) for f in out] == [True, True, False, True, True]
    from kokoro_align_amd.transcript import read_transcript
    for k in (0, 1, 3, 4):
        with np.load(logits[k]) as f:
            lg = f["data"]
        c = lg - np.mean(lg, axis=-1, keepdims=True)
        want = O.ctc_best_path_c(c - np.log(np.sum(np.exp(c), axis=-1, keepdims=True)), read_transcript(voca[k]))
        with np.load(out[k]) as f:
            assert np.array_equal(f["best_path"], want[0]) and np.array_equal(f["best_labels"], want[1])
    # the rerun skips the four that exist and fails on the same chapter again, writing nothing
    before = {f: os.path.getmtime(f) for f in out if os.path.exists(f)}
    with pytest.raises(ValueError):
        pipeline.best_path_files(logits, voca, out)
    assert before == {f: os.path.getmtime(f) for f in out if os.path.exists(f)} and not os.path.exists(out[2])
    torch.cuda.synchronize()
