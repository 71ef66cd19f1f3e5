"""Host-side mirror of the reference's boundary helpers against the reference-made goldens
(tests/golden/g4_text.json): encoder.py:14-31, transcript.py:13-67, align.py:127-190."""
import os

import numpy as np
import pytest

from golden_util import g4

import kokoro_align_amd as ka
from kokoro_align_amd import encoder, transcript


@pytest.fixture(scope="module")
def G():
    return g4()


@pytest.fixture()
def voca_file(G, tmp_path):
    p = tmp_path / "x.voca.txt"
    p.write_text(G["voca_txt"])
    return str(p)


def test_vocab(G):
    assert encoder.vocab == G["vocab"] and encoder.VOCAB_SIZE == 39 and encoder.v2i["_"] == 0


def test_is_valid_text(G):
    for text, want in G["is_valid_text"]:
        assert encoder.is_valid_text(text) == want, text


def test_encode_text(G):
    for text, want in G["encode_text"]:
        got = encoder.encode_text(text)
        assert got.dtype == np.int8 and got.tolist() == want, text


def test_decode_and_merge(G):
    for ids, want in G["decode_text"]:
        assert encoder.decode_text(ids) == want
        assert encoder.decode_text(np.array(ids, np.int32)) == want
    for text, want in G["merge_repeated"]:
        assert encoder.merge_repeated(text) == want, text
    for ids, want in G["decode_merge"]:
        assert encoder.merge_repeated(encoder.decode_text(ids)) == want


def test_read_transcript(G, voca_file):
    got = transcript.read_transcript(voca_file)
    assert str(got.dtype) == G["read_transcript_dtype"] and got.tolist() == G["read_transcript"]


def test_voca_aligner(G, voca_file):
    al = transcript.VocaAligner(voca_file)
    assert al.token_pos == G["token_pos"] and len(al) == G["aligner_len"]
    for a, b, rw, want in G["get_token"]:
        assert list(al.get_token(a, b, remove_wordsep=rw)) == want, (a, b, rw)


def test_align_writer_matches_reference_text(G, voca_file, tmp_path):
    """align() on the reference's own best_path arrays must reproduce its align.txt byte for byte
    (same float32 sums, same repr formatting)."""
    for name in ("rt_a", "rt_b", "rt_c"):
        rt = G[name]
        bp_file, mfcc_file = str(tmp_path / f"{name}.bp.npz"), str(tmp_path / f"{name}.mfcc.npz")
        np.savez(bp_file, best_path=np.array(rt["best_path"], np.int32),
                 best_labels=np.array(rt["best_labels"], np.int32),
                 best_scores=np.array(rt["best_scores"], np.float32))
        np.savez(mfcc_file, indices=np.array(rt["segments"], np.int32), data=np.zeros((rt["T"], 1), np.float32))
        for rw in (True, False):
            out = str(tmp_path / f"{name}.{int(rw)}.align.txt")
            ka.align(bp_file, mfcc_file, voca_file, out, rw)
            assert open(out).read() == rt[f"align_txt_{int(rw)}"], (name, rw)
        df = ka.pandas_read_align([out])
        assert list(df.columns) == ["audio_start", "audio_end", "text", "voca", "decoded", "non_blanks",
                                    "non_blanks_score", "all_score", "audio_len"]
        assert df["audio_end"].tolist() == rt["segments"]
        assert df["audio_start"].tolist() == [0] + rt["segments"][:-1]


def test_align_removes_partial_file_on_error(G, voca_file, tmp_path):
    rt = G["rt_b"]
    bp_file, mfcc_file = str(tmp_path / "e.bp.npz"), str(tmp_path / "e.mfcc.npz")
    np.savez(bp_file, best_path=np.array(rt["best_path"], np.int32), best_labels=np.array(rt["best_labels"], np.int32),
             best_scores=np.array(rt["best_scores"], np.float32))
    # second segment starts beyond the path: IndexError in the reference too (align.py:151)
    np.savez(mfcc_file, indices=np.array([rt["T"] + 5, rt["T"] + 9], np.int32), data=np.zeros((1, 1), np.float32))
    out = str(tmp_path / "e.align.txt")
    with pytest.raises(IndexError):
        ka.align(bp_file, mfcc_file, voca_file, out, True)
    assert not os.path.exists(out)
