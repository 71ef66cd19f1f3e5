"""Rows either side of the hot path against reference-made goldens (g5):
combine_files metadata filter (run_example.py:73-131), the AudioToChar producer network
(train.py:54-65) on CPU, the IndexDataArray npz format (preprocess.py:12-35)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from golden_util import g5
from oracle import oracle as O

from kokoro_align_amd import pipeline
from kokoro_align_amd.model import DEFAULT_PARAMS, AudioToChar, segment_logits


@pytest.fixture(scope="module")
def G():
    return g5()


@pytest.mark.parametrize("rw", [True, False])
def test_combine_files_matches_reference(G, tmp_path, rw):
    c = G["combine_files"][str(int(rw))]
    align_files, split_files = [], []
    for i, (a, s) in enumerate(zip(c["align"], c["split"])):
        af, sf = tmp_path / f"{i}.align.txt", tmp_path / f"{i}.split.txt"
        af.write_text(a); sf.write_text(s)
        align_files.append(str(af)); split_files.append(str(sf))
    meta = tmp_path / "out" / "ds.metadata.txt"
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pipeline.combine_files("ds", align_files, c["audio_files"], split_files, str(meta), rw)
    assert meta.read_text() == c["metadata"]
    assert buf.getvalue() == c["stdout"]


def test_combine_files_removes_partial_output(tmp_path):
    af, sf = tmp_path / "a.align.txt", tmp_path / "a.split.txt"
    af.write_text("1|x|k o|k o|2|0.0\n")          # 6 fields: unpack error, like the reference
    sf.write_text("100\n")
    meta = tmp_path / "m" / "ds.metadata.txt"
    with pytest.raises(ValueError):
        pipeline.combine_files("ds", [str(af)], ["a.mp3"], [str(sf)], str(meta), True, verbose=False)
    assert not meta.exists()


def test_audio_to_char_matches_reference_cpu(G):
    a = G["audio_to_char"]
    model = AudioToChar(**a["params"])
    model.load_state_dict({k: torch.tensor(v, dtype=torch.float32) for k, v in a["state_dict"].items()})
    model.eval()
    segs = [O.hash_logprobs(n, 40, seed) + np.float32(4.0) for n, seed in zip(a["segment_lens"], a["segment_seeds"])]
    got = segment_logits(model, segs, device=torch.device("cpu"))
    for g, w in zip(got, a["logits"]):
        assert np.allclose(g.numpy(), np.array(w, np.float32), atol=1e-5, rtol=0)
    full = AudioToChar(**DEFAULT_PARAMS)
    assert sum(p.numel() for p in full.parameters()) == a["default_param_count"] == 579367
    assert list(full.state_dict().keys()) == a["state_keys"]


def test_index_data_array_format(tmp_path):
    f = str(tmp_path / "x.npz")
    parts = [np.arange(6, dtype=np.float32).reshape(3, 2), np.ones((1, 2), np.float32), np.zeros((4, 2), np.float32)]
    with pipeline.open_index_data_for_write(f) as w:
        for p in parts:
            w.write(p)
    idx, data = pipeline.read_index_data(f)
    assert idx.dtype == np.int32 and idx.tolist() == [3, 4, 8] and data.shape == (8, 2)
    back = pipeline.split_segments(idx, data)
    assert all(np.array_equal(a, b) for a, b in zip(back, parts))
    g = str(tmp_path / "y.npz")
    with pytest.raises(RuntimeError):
        with pipeline.open_index_data_for_write(g) as w:
            w.write(parts[0])
            raise RuntimeError("producer failed")
    assert not os.path.exists(g)          # nothing written on failure (preprocess.py:27)


def test_mirror_network_reproduces_the_reference_at_default_params():
    """g7: the reference's AudioToChar(**DEFAULT_PARAMS) (train.py:16-20, :54-65) - our mirror module with the stored weights gives
    the stored logits on the CPU (same PyTorch LSTM arithmetic), so the GPU test of the persistent kernel compares against
    numbers the reference produced, not against this repository's own network."""
    import numpy as np
    import torch
    from golden_util import g7
    from oracle import oracle as O
    from kokoro_align_amd.model import AudioToChar, DEFAULT_PARAMS, segment_logits
    g = g7()
    assert g["params"] == [DEFAULT_PARAMS["n_mfcc"], DEFAULT_PARAMS["hidden_dim"], DEFAULT_PARAMS["vocab_size"]] == [40, 128, 39]
    model = AudioToChar().eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in g["state"].items()})
    segs = [O.hash_logprobs(n, 40, sd) * np.float32(g["scale"]) + np.float32(g["offset"]) for n, sd in zip(g["lens"], g["seeds"])]
    got = segment_logits(model, segs, device="cpu")
    for a, w in zip(got, g["logits"]):
        np.testing.assert_allclose(a.numpy(), w, rtol=0, atol=1e-5)
