"""Synthetic stand-ins for the BASELINE.json configurations.

The reference's real inputs (LibriVox MP3s, Aozora texts, the pretrained checkpoint) need the network;
what the alignment step sees of them is a list of (T frames, S phonemes) lattices, one per audio file
(run_example.py:248-254 loops over them), with V = 39 log-prob columns (encoder.py:5-11):

    cfg1  gongitsune, one MP3 of 0:15:42            T = 81140 (942 s x 86.13 frames/s), S = 2000
    cfg2  the synthetic headline lattice             T = 50000, V = 64, S = 5000
    cfg3  Kokoro, 08:46:41 of audio                  64 chapters, 2.72 M frames, T in [20k, 160k], S = 0.14 T
    cfg4  Meian, 16:39:29 of audio                   120 chapters, 5.17 M frames, same recipe
    cfg5  the long-form stress lattice               T = 500000, V = 64, S = 50000

Durations are example.json's ``totaltime``; the frame rate is 22050 / 256 (preprocess.py:102-104).
Chapter lengths are drawn from a fixed-seed generator so that every tool, test and bench sees the
same book.  Inputs of lattice i of a book are hash-generated with seed ``seed0 + i`` (the generator of
include/kokoro_align_amd.h), so the CPU oracle can rebuild any of them bit for bit.
"""
import numpy as np

V_MODEL = 39
CFG1 = dict(T=81140, V=39, S=2000)
CFG2 = dict(T=50000, V=64, S=5000)
CFG5 = dict(T=500000, V=64, S=50000)
BOOK_SEED0 = 10000


def book_shapes(total_frames, n_chapters, seed):
    """[(T, S)] of one audio book: chapter weights uniform in [0.4, 3.0], T clipped to [20k, 160k], S = 0.14 T."""
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.4, 3.0, n_chapters)
    T = np.maximum(20000, (w / w.sum() * total_frames).astype(int))
    T = np.minimum(T, 160000)
    return [(int(t), int(0.14 * t)) for t in T]


def kokoro_book():
    """BASELINE configs[2] stand-in: (name, shapes)."""
    return "Kokoro stand-in (8.78 h)", book_shapes(2_720_000, 64, 1)


def meian_book():
    """BASELINE configs[3] stand-in."""
    return "Meian stand-in (16.66 h)", book_shapes(5_170_000, 120, 2)


def device_book(shapes, V=V_MODEL, seed0=BOOK_SEED0, device="cuda"):
    """Hash-generated log-probs and labels of every chapter, resident on the device.
    Returns (log_probs list, labels list)."""
    import torch
    from . import _lib
    lib = _lib.load_library()
    stream = torch.cuda.current_stream().cuda_stream
    lps, labs = [], []
    for i, (T, S) in enumerate(shapes):
        lp = torch.empty((T, V), dtype=torch.float32, device=device)
        lab = torch.empty(max(S, 1), dtype=torch.int32, device=device)[:S]
        _lib.check(lib.ka_hash_logprobs_f32(lp.data_ptr(), T, V, V, seed0 + i, stream), "ka_hash_logprobs_f32")
        if S > 0:
            _lib.check(lib.ka_hash_labels_i32(lab.data_ptr(), S, V, seed0 + i, stream), "ka_hash_labels_i32")
        lps.append(lp)
        labs.append(lab)
    torch.cuda.synchronize()
    return lps, labs
