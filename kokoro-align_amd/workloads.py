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


# example.json of the reference: the 14 datasets with "enabled": true and their `totaltime` (h:mm:ss), in file order -
# the unit of work of `python run_example.py` without --dataset (run_example.py:283-304 loops them).  63.95 hours.
CORPUS_DATASETS = (
    ("meian-by-soseki-natsume", "16:39:29"), ("kokoro-by-soseki-natsume", "08:46:41"), ("inakakyoshi-by-katai-tayama", "08:13:26"),
    ("nowaki-by-soseki-natsume", "4:40:49"), ("kusamakura-by-soseki-natsume", "04:27:35"), ("botchan-by-soseki-natsume-2", "04:26:27"),
    ("gan-by-ogai-mori", "03:41:31"), ("umareizuru-nayami-by-takeo-arishima", "2:43:12"), ("garasudono-uchi-by-natsume-soseki", "2:39:53"),
    ("eijitsu-syohin-by-soseki-natsume", "2:33:54"), ("futon-by-katai-tayama", "2:28:58"), ("kouyahijiri-by-kyoka-izumi", "2:06:23"),
    ("gongitsune-by-nankichi-niimi", "0:15:42"), ("caucasus-no-hagetaka-by-yoshio-toyoshima", "0:13:04"),
)
FRAMES_PER_SECOND = 22050.0 / 256.0      # preprocess.py:102-104 (sample rate / hop length)
CORPUS_SEED0 = 20000


def _seconds(hms):
    h, m, s = (int(x) for x in hms.split(":"))
    return 3600 * h + 60 * m + s


def corpus():
    """The whole of example.json as the alignment step sees it: [(dataset id, [(T, S), ...])] for the 14 enabled
    datasets.  Chapter counts are not in example.json (they are the MP3s of each LibriVox zip): the two books whose
    stand-ins exist keep theirs (120 and 64 chapters, 8.3 minutes on average) and the others get one chapter per 8.25
    minutes of audio, lengths by the recipe of book_shapes (weights uniform in [0.4, 3.0], 20k..160k frames, S = 0.14 T);
    the two recordings of a quarter of an hour are single files.  ~460 lattices, ~19.8 M frames."""
    out = []
    for k, (name, total) in enumerate(CORPUS_DATASETS):
        frames = int(_seconds(total) * FRAMES_PER_SECOND)
        if name.startswith("meian"):
            shapes = meian_book()[1]
        elif name.startswith("kokoro-by"):
            shapes = kokoro_book()[1]
        elif frames < 160000:
            shapes = [(frames, int(0.14 * frames))]
        else:
            shapes = book_shapes(frames, max(1, round(_seconds(total) / 60.0 / 8.25)), 100 + k)
        out.append((name, shapes))
    return out


def corpus_seed0(dataset_index):
    """hash seed of chapter 0 of dataset `dataset_index` of corpus(): chapter i uses seed0 + i"""
    return CORPUS_SEED0 + 1000 * dataset_index


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
