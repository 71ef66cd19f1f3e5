#!/usr/bin/env python3
"""Side benchmark (not the headline): BASELINE configs[2]/[3] stand-ins — a whole audio book as one batch of
independent chapter lattices: Kokoro ~2.72 M frames, Meian ~5.17 M frames, V=39,
S ~ 0.14 T, per-chapter T in [20k, 160k].  Reports wall time per book for both forward-kernel forms.

    python tools/bench_books.py
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import kokoro_align_amd as ka
from kokoro_align_amd.align import DeviceBatch


def book(total_frames, n_chapters, seed):
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.4, 3.0, n_chapters)
    T = np.maximum(20000, (w / w.sum() * total_frames).astype(int))
    T = np.minimum(T, 160000)
    return [(int(t), int(0.14 * t)) for t in T]


def run(name, shapes):
    lib = ka.load_library()
    V = 39
    lps, labs = [], []
    for i, (T, S) in enumerate(shapes):
        lp = torch.empty((T, V), dtype=torch.float32, device="cuda")
        lab = torch.empty(S, dtype=torch.int32, device="cuda")
        lib.ka_hash_logprobs_f32(lp.data_ptr(), T, V, V, 10000 + i, None)
        lib.ka_hash_labels_i32(lab.data_ptr(), S, V, 10000 + i, None)
        lps.append(lp); labs.append(lab)
    torch.cuda.synchronize()
    frames = sum(t for t, _ in shapes)
    out = {"book": name, "chapters": len(shapes), "frames": frames, "longest_chapter": max(t for t, _ in shapes)}
    ref = None
    for mode in ("wave_exact", "wave"):
        b = DeviceBatch(lps, labs)
        b.engine.set_mode(mode)
        b.run()
        t0 = time.perf_counter()
        for _ in range(3):
            b.run()
        dt = (time.perf_counter() - t0) / 3
        ends = [int(p[-1]) for p in b.path]
        if ref is None:
            ref = [p.clone() for p in b.path]
        same = all(torch.equal(a, c) for a, c in zip(ref, b.path))
        out[mode] = {"ms": dt * 1e3, "frames_per_s": frames / dt, "identical_paths": same,
                     "all_ends_at_trailing_blank": all(e == 2 * s for e, (_, s) in zip(ends, shapes))}
    b.engine.set_mode("auto")
    return out


if __name__ == "__main__":
    res = [run("Kokoro stand-in (8.78 h)", book(2_720_000, 64, 1)), run("Meian stand-in (16.66 h)", book(5_170_000, 120, 2))]
    print(json.dumps(res, indent=1))
