"""Helpers to read the golden fixtures under tests/golden/ (made by make_golden.py from the
reference implementation).  Shared by the CPU oracle tests and the GPU parity tests."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def g1_cases():
    """Yield dicts: lp [T,V] f32, labels [S] i32, beam, max_move, status, path/best_labels/best_scores."""
    with np.load(os.path.join(GOLDEN, "g1_tiny.npz")) as f:
        meta, lp, labels = f["meta"], f["lp"], f["labels"]
        path, bl, bs = f["path"], f["best_labels"], f["best_scores"]
    out = []
    for i, (T, V, S, beam, mm, status, lp_off, lab_off, out_off) in enumerate(meta.tolist()):
        c = dict(idx=i, T=T, V=V, S=S, beam=beam, max_move=mm, status=status,
                 lp=lp[lp_off:lp_off + T * V].reshape(T, V).copy(),
                 labels=labels[lab_off:lab_off + S].copy())
        if status == 0:
            c.update(path=path[out_off:out_off + T], best_labels=bl[out_off:out_off + T],
                     best_scores=bs[out_off:out_off + T])
        out.append(c)
    return out


def _unpack(first, delta):
    return np.concatenate([[first], first + np.cumsum(delta.astype(np.int64))]).astype(np.int32)


def g2_cases():
    with np.load(os.path.join(GOLDEN, "g2_medium.npz")) as f:
        specs = f["specs"].tolist()
        out = []
        for i, (T, V, S, beam, mm, seed) in enumerate(specs):
            out.append(dict(idx=i, T=T, V=V, S=S, beam=beam, max_move=mm, seed=seed,
                            path=_unpack(int(f[f"first_{i}"]), f[f"delta_{i}"]),
                            sha_labels=str(f[f"sha_labels_{i}"]), sha_scores=str(f[f"sha_scores_{i}"]),
                            sum_scores=float(f[f"sum_scores_{i}"])))
    return out


def g3_case():
    with np.load(os.path.join(GOLDEN, "g3_cfg2.npz")) as f:
        T, V, S, beam, mm, seed = f["spec"].tolist()
        return dict(T=T, V=V, S=S, beam=beam, max_move=mm, seed=seed,
                    path=_unpack(int(f["first"]), f["delta"]),
                    sha_labels=str(f["sha_labels"]), sha_scores=str(f["sha_scores"]),
                    sum_scores=float(f["sum_scores"]))


def g4():
    with open(os.path.join(GOLDEN, "g4_text.json")) as f:
        return json.load(f)


def g5():
    with open(os.path.join(GOLDEN, "g5_pipeline.json")) as f:
        return json.load(f)


def g6():
    """Silence-splitting goldens (reference get_split_points / get_silent_ranges on synthetic recordings)."""
    with open(os.path.join(GOLDEN, "g6_split.json")) as f:
        return json.load(f)


def g7():
    """The reference's AudioToChar at DEFAULT_PARAMS: dict with `state` (name -> float32 array), `lens`, `seeds`, `scale`,
    `offset`, `logits` (list of [len_i, 39]) and `sha` (of the stored weights, recomputed here: a damaged file is seen)."""
    with np.load(os.path.join(GOLDEN, "g7_audio_to_char_default.npz")) as f:
        state = {k[2:]: f[k] for k in f.files if k.startswith("w:")}
        lens = f["segment_lens"].tolist()
        seeds = f["segment_seeds"].tolist()
        scale, offset = (float(x) for x in f["input_scale_offset"])
        flat = f["logits"]
        sha_stored = bytes(f["state_sha256"]).decode()
        params = f["params"].tolist()
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k]).tobytes())
    assert h.hexdigest() == sha_stored, "g7: stored weights do not match their checksum"
    logits, k = [], 0
    for n in lens:
        logits.append(flat[k:k + n])
        k += n
    return dict(state=state, lens=lens, seeds=seeds, scale=scale, offset=offset, logits=logits, sha=sha_stored, params=params)
