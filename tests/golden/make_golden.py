#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE implementation.

Runs only in the dev container (needs /root/reference).  Nothing from the reference is
copied: the fixtures are inputs + the outputs the reference computed for them.

    PYTHONDONTWRITEBYTECODE=1 TQDM_DISABLE=1 python tests/golden/make_golden.py [g1 g2 g3 g4]

Fixtures
  g1_tiny.npz        ~260 tiny lattices (inputs and outputs), incl. true ties, label-0
                     transcripts, unreachable ends, empty-beam errors, S=0, T=1, max_move 1..6
  g2_medium.npz      band-binding medium lattices; inputs come from the hash generator
                     (oracle/oracle.py: hash_logprobs/hash_labels), only outputs stored
  g3_cfg2.npz        BASELINE.json configs[1] (T=50000, V=64, S=5000, seed 0) — path deltas,
                     SHA-256 of labels/scores
  g4_text.json       encoder / transcript / VocaAligner I/O pairs and a best_path()->align()
                     file round trip (inputs embedded)
  g5_pipeline.json   combine_files metadata filter (run_example.py:73-131) on the g4 align.txt
                     texts + lines that hit every blocking rule; AudioToChar (hidden 8) weights,
                     packed 3-segment input and logits (train.py:54-65, :92-95)

  g7_audio_to_char_default.npz   the reference's AudioToChar at DEFAULT_PARAMS (hidden 128): seeded weights, 19 ragged
                     segments, logits (train.py:16-20, :54-65, :92-95) - pins the persistent LSTM kernel

Reference entry points exercised: kokoro_align/align.py:43 (ctc_best_path), :112 (best_path),
:127 (align); kokoro_align/encoder.py:14-31; kokoro_align/transcript.py:13-67.

Note on g4: kokoro_align/transcript.py imports kokoro_align/_text2voca.py at module import,
which constructs a `fugashi.Tagger()` (a MeCab binding absent from this image).  The functions
pinned here (read_transcript, VocaAligner) never call the tagger, so a placeholder module
object is registered for the import only.  The DP goldens g1-g3 need no such placeholder.
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
os.environ.setdefault("TQDM_DISABLE", "1")
sys.dont_write_bytecode = True

from oracle.oracle import hash_labels, hash_logprobs  # noqa: E402  (input generator only)
from kokoro_align.align import ctc_best_path as ref_ctc_best_path  # noqa: E402


def run_ref(lp, labels, beam, max_move):
    """-> (status, path, labels, scores); status 1 = ValueError (empty beam)."""
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
            p, l, s = ref_ctc_best_path(lp, labels, beam_size=beam, max_move=max_move)
        return 0, p.astype(np.int32), l.astype(np.int32), s.astype(np.float32)
    except ValueError:
        return 1, None, None, None


def make_g1():
    rng = np.random.default_rng(20261003)
    cases = []

    def add(lp, labels, beam=1000, max_move=4):
        cases.append((np.ascontiguousarray(lp, np.float32), np.asarray(labels, np.int8), beam, max_move))

    beams = [4, 7, 10, 16, 50, 1000]
    for i in range(240):
        mode = i % 8
        V = int(rng.choice([3, 4, 5, 8, 12, 39, 64]))
        S = int(rng.integers(0, 60)) if mode != 5 else 0
        T = int(rng.integers(1, 120))
        beam = int(rng.choice(beams))
        mm = 4
        lp = rng.standard_normal((T, V)).astype(np.float32)
        lp = lp - np.log(np.sum(np.exp(lp), axis=-1, keepdims=True))
        labels = rng.integers(1, V, size=S)
        if mode == 1:      # quantised to 1/4: many exact ties
            lp = np.round(lp * 4) / 4
        elif mode == 2:    # label 0 inside the transcript (veto tests label VALUE)
            labels = rng.integers(0, V, size=S)
            if S:
                labels[rng.integers(0, S)] = 0
        elif mode == 3:    # short audio: T < L/3, end unreachable or beam outrun
            T = max(1, (2 * S + 1) // int(rng.integers(3, 6)))
            lp = lp[:T] if T <= lp.shape[0] else rng.standard_normal((T, V)).astype(np.float32)
        elif mode == 4:    # coarse integers: ties everywhere, repeated labels
            lp = -rng.integers(0, 3, size=(T, V)).astype(np.float32)
            labels = rng.integers(1, min(V, 3), size=S)
        elif mode == 6:    # other max_move values
            mm = int(rng.choice([1, 2, 3, 5, 6]))
        elif mode == 7:    # -inf entries (a state can be live with score -inf)
            mask = rng.random((T, V)) < 0.15
            lp = np.where(mask, -np.inf, lp).astype(np.float32)
        add(lp, labels, beam, mm)
    # hand-picked edges
    add(rng.standard_normal((1, 5)), [1, 2])               # T=1
    add(rng.standard_normal((3, 5)), np.ones(20) * 2)      # T=3, L=41
    add(rng.standard_normal((40, 5)), [])                  # S=0
    add(rng.standard_normal((10, 5)), np.ones(40), beam=4)  # band outruns states -> ValueError
    add(np.zeros((30, 4)), [1, 1, 1, 2, 2, 3])             # all ties, repeats
    add(rng.standard_normal((200, 6)), rng.integers(1, 6, 30), beam=16)
    add(rng.standard_normal((64, 6)), rng.integers(1, 6, 90), beam=1000)   # L/T ~ 2.8
    add(rng.standard_normal((100, 39)), rng.integers(1, 39, 150), beam=50)  # L/T = 3.01
    add(rng.standard_normal((100, 39)), rng.integers(1, 39, 149), beam=1000)
    add(rng.standard_normal((50, 4)), rng.integers(0, 2, 40), beam=10, max_move=6)

    meta, lps, labs, paths, blabs, bscs = [], [], [], [], [], []
    lp_off = lab_off = out_off = 0
    for lp, labels, beam, mm in cases:
        status, p, l, s = run_ref(lp, labels, beam, mm)
        T, V = lp.shape
        S = labels.shape[0]
        meta.append([T, V, S, beam, mm, status, lp_off, lab_off, out_off])
        lps.append(lp.ravel())
        labs.append(labels.astype(np.int32))
        lp_off += T * V
        lab_off += S
        if status == 0:
            assert p.shape[0] == T
            paths.append(p); blabs.append(l); bscs.append(s)
            out_off += T
    np.savez_compressed(
        os.path.join(HERE, "g1_tiny.npz"),
        meta=np.array(meta, np.int64), lp=np.concatenate(lps), labels=np.concatenate(labs),
        path=np.concatenate(paths), best_labels=np.concatenate(blabs), best_scores=np.concatenate(bscs))
    n_err = sum(m[5] for m in meta)
    print(f"g1: {len(cases)} cases, {n_err} ValueError cases, {lp_off * 4 / 1e6:.2f} MB of log-probs")


def path_pack(p):
    d = np.diff(p.astype(np.int64))
    assert d.min() >= 0 and d.max() <= 255
    return np.int64(p[0]), d.astype(np.uint8)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_g2():
    specs = [  # T, V, S, beam, max_move, seed
        (8000, 39, 2000, 1000, 4, 11),
        (5000, 39, 600, 1000, 4, 12),
        (6000, 64, 1500, 200, 4, 13),
        (3000, 39, 2600, 1000, 4, 14),   # L/T = 1.73: band moves >1 per frame
        (4000, 17, 900, 64, 3, 15),
        (2500, 8, 3700, 1000, 4, 16),    # L/T = 2.96: close to the reachability limit
    ]
    out = {"specs": np.array(specs, np.int64)}
    for i, (T, V, S, beam, mm, seed) in enumerate(specs):
        lp = hash_logprobs(T, V, seed)
        labels = hash_labels(S, V, seed)
        status, p, l, s = run_ref(lp, labels.astype(np.int8), beam, mm)
        assert status == 0, (i, "unexpected ValueError")
        first, d = path_pack(p)
        out[f"first_{i}"] = first
        out[f"delta_{i}"] = d
        out[f"sha_labels_{i}"] = np.array(sha(l))
        out[f"sha_scores_{i}"] = np.array(sha(s))
        out[f"sum_scores_{i}"] = np.float64(np.sum(s.astype(np.float64)))
        print(f"g2[{i}] T={T} S={S} beam={beam} end={p[-1]} (L-1={2 * S})")
    np.savez_compressed(os.path.join(HERE, "g2_medium.npz"), **out)


def make_g3():
    T, V, S, seed = 50000, 64, 5000, 0
    lp = hash_logprobs(T, V, seed)
    labels = hash_labels(S, V, seed)
    status, p, l, s = run_ref(lp, labels.astype(np.int8), 1000, 4)
    assert status == 0
    first, d = path_pack(p)
    np.savez_compressed(
        os.path.join(HERE, "g3_cfg2.npz"), spec=np.array([T, V, S, 1000, 4, seed], np.int64),
        first=first, delta=d, sha_labels=np.array(sha(l)), sha_scores=np.array(sha(s)),
        sum_scores=np.float64(np.sum(s.astype(np.float64))))
    print(f"g3 cfg2: end={p[-1]} sum={np.sum(s.astype(np.float64)):.6f}")


VOCA_TXT = """こころ|k o k o r o
、|,
「|
夏目|n a ts u m e
っ|q
漱石|s o: s e k i
。|.
私|w a t a sh i
は|w a
その|s o n o
人|h i t o
を|o
常|ts u n e
に|n i
先生|s e N s e:
と|t o
呼ん|y o N
で|d e
い|i
た|t a
。|.
だ|d a
から|k a r a
ここ|k o k o
で|d e
も|m o
ただ|t a d a
先生|s e N s e:
と|t o
書く|k a k u
だけ|d a k e
で|d e
本名|h o N m y o:
は|w a
打ち明け|u ch i a k e
ない|n a i
！|!
？|?
"""


def make_g4():
    # placeholder for the import-time-only dependency (see module docstring)
    ph = types.ModuleType("fugashi")
    ph.Tagger = lambda *a, **k: None
    sys.modules.setdefault("fugashi", ph)
    from kokoro_align import encoder as renc
    from kokoro_align import transcript as rtr
    from kokoro_align import align as ralign

    out = {}
    texts = [
        "k o k o r o", "_ _ k k k _ o o _ _ k _ o _ r r o _", "_", "_ _ _", "", "a a a a", "n n n n n n n",
        "a b a b a b", "k o k o k o r o", "s e N s e: s e N s e:", "a _ a _ a", "sh i sh i sh i _ _ ts u",
        "q . , ! ?", "k o q r o .", "x y z", "a: a: i: i: _ u: u:", "N N N _ N", "a a _ a a", "ky o: ky o:",
    ]
    out["is_valid_text"] = [[t, bool(renc.is_valid_text(t))] for t in texts]
    out["encode_text"] = [[t, renc.encode_text(t).tolist()] for t in texts]
    out["merge_repeated"] = [[t, renc.merge_repeated(t)] for t in texts]
    rng = np.random.default_rng(7)
    dec = [rng.integers(0, 39, size=int(n)).tolist() for n in (0, 1, 5, 30)]
    dec.append([0, 0, 18, 18, 18, 0, 24, 24, 0, 0, 18, 0, 24, 0, 28, 28, 24, 0])
    out["decode_text"] = [[ids, renc.decode_text(ids)] for ids in dec]
    out["decode_merge"] = [[ids, renc.merge_repeated(renc.decode_text(ids))] for ids in dec]
    out["vocab"] = list(renc.vocab)

    with tempfile.TemporaryDirectory() as td:
        voca_file = os.path.join(td, "x.voca.txt")
        with open(voca_file, "wt") as f:
            f.write(VOCA_TXT)
        out["voca_txt"] = VOCA_TXT
        labels = rtr.read_transcript(voca_file)
        out["read_transcript"] = labels.tolist()
        out["read_transcript_dtype"] = str(labels.dtype)
        al = rtr.VocaAligner(voca_file)
        out["token_pos"] = list(al.token_pos)
        out["aligner_len"] = len(al)
        n = len(al)
        pairs = [(0, 0), (0, 3), (0, 6), (3, 9), (5, 20), (0, n), (n - 1, n), (n, n + 5), (10, 40), (40, n),
                 (7, 7), (12, 13), (0, n + 100), (n // 2, n), (2, 30), (25, 60)]
        out["get_token"] = [[a, b, rw, list(al.get_token(a, b, remove_wordsep=rw))]
                            for a, b in pairs for rw in (True, False)]

        # file round trip: logits.npz -> best_path() -> align()
        S = labels.shape[0]
        for name, T, segs, seed in (("rt_a", 360, [90, 200, 290, 360], 101), ("rt_b", 150, [40, 150], 102),
                                    ("rt_c", 420, [100, 101, 250, 419], 103)):
            # logits are regenerated by the tests from the hash generator (no MBs of floats stored)
            logits = hash_logprobs(T, 39, seed) + np.float32(4.0)
            # make the transcript likely: boost the right label along a rough diagonal
            ext = np.zeros(2 * S + 1, np.int64); ext[1::2] = labels
            pos = (np.arange(T) * (2 * S + 1) // T)
            logits[np.arange(T), ext[pos]] += 4.0
            logits_file = os.path.join(td, name + ".logits.npz")
            mfcc_file = os.path.join(td, name + ".mfcc.npz")
            bp_file = os.path.join(td, name + ".best_path.npz")
            np.savez(logits_file, indices=np.array(segs, np.int32), data=logits)
            np.savez(mfcc_file, indices=np.array(segs, np.int32), data=np.zeros((T, 1), np.float32))
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(buf):
                ralign.best_path(logits_file, voca_file, bp_file)
            with np.load(bp_file) as f:
                bpd = {k: f[k] for k in f.files}
            rt = {"T": T, "segments": segs, "logits_seed": seed, "logits_sha": sha(logits),
                  "best_path": bpd["best_path"].tolist(), "best_labels": bpd["best_labels"].tolist(),
                  "best_scores": [float(x) for x in bpd["best_scores"]],
                  "dtypes": {k: str(v.dtype) for k, v in bpd.items()},
                  "stdout": buf.getvalue()}
            for rw in (True, False):
                align_file = os.path.join(td, f"{name}.{int(rw)}.align.txt")
                ralign.align(bp_file, mfcc_file, voca_file, align_file, rw)
                with open(align_file) as f:
                    rt[f"align_txt_{int(rw)}"] = f.read()
            out[name] = rt
    with open(os.path.join(HERE, "g4_text.json"), "wt") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("g4: written; aligner_len", out["aligner_len"], "S", len(out["read_transcript"]))


def make_g5():
    """combine_files (run_example.py:73-131) and the AudioToChar network (train.py:54-65, :92-95)."""
    import torch
    import run_example as rex                      # /root/reference/run_example.py (stdlib imports only)
    from kokoro_align import train as rtrain

    out = {}
    g4 = json.load(open(os.path.join(HERE, "g4_text.json")))
    extra = [  # hand-written align.txt lines that hit every blocking rule
        "100|リブリ ボックス の 録音 です|r i b u r i b o q k u s u|r i b u|10|-3.5|-9.25",
        "200||||0|0.0|-1.0",
        "300|こころ|k o k o r o|k o k o r o|6|-1.5|-2.5",
        "400|夏目 漱石|n a ts u m e _ s o: s e k i|n a ts u m e s o: s e k i|12|-2.0|-3.0",
        "500|先生|s e N s e: x|s e N s e:|5|-1.0|-1.0",
        "600|私 は|w a t a sh i w a|w a|1|-0.5|-7.0",
        "700|その 人|s o n o h i t o|s o n o h i t|7|-0.25|-0.5",
    ]
    with tempfile.TemporaryDirectory() as td:
        cases = {}
        for rw in (True, False):
            align_files, split_files, audio_files = [], [], []
            for name in ("rt_a", "rt_b", "rt_c", "extra"):
                lines = extra if name == "extra" else g4[name][f"align_txt_{int(rw)}"].splitlines()
                af = os.path.join(td, f"{name}.{int(rw)}.align.txt")
                sf = os.path.join(td, f"{name}.{int(rw)}.split.txt")
                with open(af, "wt") as f:
                    f.write("\n".join(lines) + "\n")
                with open(sf, "wt") as f:
                    for i in range(len(lines)):
                        f.write(f"{(i + 1) * 22050 + 7 * i}\n")
                align_files.append(af); split_files.append(sf); audio_files.append(f"/some/dir/{name}.mp3")
            meta = os.path.join(td, "out", f"ds.{int(rw)}.metadata.txt")
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                rex.combine_files("ds", align_files, audio_files, split_files, meta, rw)
            cases[str(int(rw))] = {
                "align": [open(a).read() for a in align_files], "split": [open(s_).read() for s_ in split_files],
                "audio_files": audio_files, "metadata": open(meta).read(), "stdout": buf.getvalue()}
        out["combine_files"] = cases

    # AudioToChar with a small hidden size (tiny fixture); weights from the reference's own init
    torch.manual_seed(1234)
    model = rtrain.AudioToChar(n_mfcc=40, hidden_dim=8, vocab_size=39).eval()
    state = {k: v.numpy().tolist() for k, v in model.state_dict().items()}
    lens = [17, 5, 29]
    segs = [np.ascontiguousarray(hash_logprobs(n, 40, 900 + i) + np.float32(4.0)) for i, n in enumerate(lens)]
    with torch.no_grad():
        packed = rtrain.generate_batch_audio([torch.from_numpy(x) for x in segs])
        logits, out_lens = model(packed)
    out["audio_to_char"] = {
        "params": {"n_mfcc": 40, "hidden_dim": 8, "vocab_size": 39}, "state_dict": state, "segment_lens": lens,
        "segment_seeds": [900, 901, 902], "out_lens": out_lens.tolist(),
        "logits": [logits[:n, j, :].numpy().tolist() for j, n in enumerate(lens)],
        "default_param_count": int(sum(p.numel() for p in rtrain.AudioToChar(**rtrain.DEFAULT_PARAMS).parameters())),
        "state_keys": list(rtrain.AudioToChar(**rtrain.DEFAULT_PARAMS).state_dict().keys())}
    with open(os.path.join(HERE, "g5_pipeline.json"), "wt") as f:
        json.dump(out, f, ensure_ascii=False)
    print("g5: written;", len(out["combine_files"]["1"]["metadata"].splitlines()), "metadata lines (remove_wordsep=True)")


def make_g6():
    """Silence splitting: the reference's own get_silent_ranges / get_split_points (kokoro_align/preprocess.py:38-97)
    on synthetic waveforms (oracle.frontend_oracle.hash_waveform: only the recipe is stored, not the audio).
    kokoro_align/preprocess.py imports torchaudio at module import; the two functions pinned here are plain NumPy
    and never touch it, so - as for g4 - a placeholder module object is registered for the import only.  The MFCC
    transform itself (torchaudio.transforms.MFCC) cannot be pinned: torchaudio is not in this image."""
    ph = types.ModuleType("torchaudio")
    sys.modules.setdefault("torchaudio", ph)
    from kokoro_align import preprocess as rpre
    from oracle.frontend_oracle import hash_waveform, split_parameters

    sr = 22050
    par = split_parameters(sr, 512)
    rng = np.random.default_rng(6)

    def speech(total_s, voiced=(0.4, 6.0), silent=(0.05, 1.2), lead=0.0, tail=0.0, amp=(0.05, 0.6), floor=0.0):
        """pieces of one synthetic recording: voiced bursts separated by silences (seconds -> samples)"""
        pieces, t = [], lead
        if floor:
            pieces.append((0, int(total_s * sr), floor))
        while t < total_s - tail:
            d = float(rng.uniform(*voiced))
            pieces.append((int(t * sr), int(min(t + d, total_s - tail) * sr), float(rng.uniform(*amp))))
            t += d + float(rng.uniform(*silent))
        return int(total_s * sr), pieces

    cases = []
    recipes = [
        ("typical", speech(70.0)),
        ("leading and trailing silence", speech(45.0, lead=1.3, tail=2.1)),
        ("noise floor", speech(60.0, floor=0.002)),
        ("short pauses only: the minimum silence is halved until pieces are short enough", speech(50.0, voiced=(1.0, 4.0), silent=(0.06, 0.2))),
        ("long pauses, short bursts: many merges", speech(40.0, voiced=(0.3, 1.0), silent=(0.3, 0.9))),
        ("one burst", (int(6.0 * sr), [(int(1.0 * sr), int(4.0 * sr), 0.3)])),
        ("two bursts, pause in the middle", (int(9.0 * sr), [(0, int(4.0 * sr), 0.3), (int(5.0 * sr), int(9.0 * sr), 0.2)])),
        ("no silence at all and too long: cannot be split", (int(20.0 * sr), [(0, int(20.0 * sr), 0.3)])),
        ("length not a multiple of the window", speech(33.3337)),
        ("quiet recording", speech(55.0, amp=(0.001, 0.004))),
        ("digital silence, too long: raises", (int(20.0 * sr), [])),
        ("digital silence, short", (int(5.0 * sr), [])),
    ]
    for k, (name, (n, pieces)) in enumerate(recipes):
        seed = 600 + k
        x = hash_waveform(n, seed, pieces)
        try:
            pts = rpre.get_split_points(x, par["minimum_silent_frames"], par["minimum_split_distance"],
                                        par["maximum_split_distance"], par["window_size"])
            case = {"status": 0, "split_points": [int(v) for v in pts]}
        except ValueError as e:
            case = {"status": 1, "error": str(e)}
        case.update(name=name, n=n, seed=seed, pieces=[[int(a), int(b), float(v)] for a, b, v in pieces])
        cases.append(case)
    # get_silent_ranges on hand-made masks
    masks = ["0011100111000", "1110001110011", "1111", "0000", "10", "01", "010", "101", "1100110011", "0110"]
    sil = []
    for mk in masks:
        v = np.array([c == "1" for c in mk])
        try:
            sil.append({"mask": mk, "ranges": rpre.get_silent_ranges(v).tolist()})
        except Exception as e:   # noqa: BLE001  (recorded as the reference's behaviour)
            sil.append({"mask": mk, "raises": type(e).__name__})
    out = {"sample_rate": sr, "parameters": par, "cases": cases, "silent_ranges": sil}
    with open(os.path.join(HERE, "g6_split.json"), "wt") as f:
        json.dump(out, f)
    print("g6: written;", [(c["name"][:20], c["status"], len(c.get("split_points", []))) for c in cases])


def make_g7():
    """The reference's AudioToChar at its DEFAULT_PARAMS (train.py:16-20: n_mfcc 40, hidden_dim 128, vocab 39 - the size the
    persistent HIP kernel ka_lstm_layer_f32 is built for), weights from the reference's own initialisation under a fixed
    seed, ragged segments (lengths 1 .. 300, one longer than a 16-sequence tile's neighbours) packed by the reference's
    generate_batch_audio (train.py:92-95) and run through its forward (train.py:61-65).  Stored: the weights (float32, so that
    a different initialisation stream in another torch build is seen rather than silently compared), their SHA-256, the
    recipe of the inputs (hash generator) and the logits of every segment."""
    import torch
    from kokoro_align import train as rtrain

    torch.manual_seed(4321)
    model = rtrain.AudioToChar(**rtrain.DEFAULT_PARAMS).eval()
    state = {k: v.numpy().astype(np.float32) for k, v in model.state_dict().items()}
    lens = [1, 7, 129, 300, 64, 2, 33, 18, 257, 96, 5, 41, 300, 17, 150, 3, 80, 222, 9]     # 19 segments: two tiles of 16
    seeds = [7000 + i for i in range(len(lens))]
    segs = [np.ascontiguousarray(hash_logprobs(n, 40, sd) * np.float32(0.75) + np.float32(3.0)) for n, sd in zip(lens, seeds)]
    with torch.no_grad():
        packed = rtrain.generate_batch_audio([torch.from_numpy(x) for x in segs])
        logits, out_lens = model(packed)
    assert out_lens.tolist() == lens
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k]).tobytes())
    arrays = {"w:" + k: v for k, v in state.items()}
    arrays["segment_lens"] = np.array(lens, np.int32)
    arrays["segment_seeds"] = np.array(seeds, np.int32)
    arrays["input_scale_offset"] = np.array([0.75, 3.0], np.float32)
    arrays["logits"] = np.concatenate([logits[:n, j, :].numpy() for j, n in enumerate(lens)], 0).astype(np.float32)
    arrays["state_sha256"] = np.frombuffer(h.hexdigest().encode(), dtype=np.uint8)
    arrays["params"] = np.array([rtrain.DEFAULT_PARAMS["n_mfcc"], rtrain.DEFAULT_PARAMS["hidden_dim"], rtrain.DEFAULT_PARAMS["vocab_size"]], np.int32)
    np.savez_compressed(os.path.join(HERE, "g7_audio_to_char_default.npz"), **arrays)
    print("g7: written;", len(lens), "segments,", int(sum(lens)), "frames,", sum(v.size for v in state.values()), "weights, sha", h.hexdigest()[:16])


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]
    for w in which:
        {"g1": make_g1, "g2": make_g2, "g3": make_g3, "g4": make_g4, "g5": make_g5, "g6": make_g6, "g7": make_g7}[w]()
