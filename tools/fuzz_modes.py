#!/usr/bin/env python3
"""Random lattice shapes through every kernel form and backtrace against the C oracle (run on the GPU box).

    python tools/fuzz_modes.py [seconds] [seed]

Shapes are drawn to hit what the hand-written paths special-case: bands that bind on both sides, bands narrower than a
tile, unbanded lattices, L/T from 0.01 to 3.2, transcripts with label 0, frame counts around multiples of 32 (chunk and
block boundaries), S = 0, one-frame lattices, beams around 1009 (the one-wavefront ring's limit).  Every case is run as a
small batch (several lattices of different length in one launch).  Prints a summary line; exit status 1 on any mismatch.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kokoro_align_amd as ka
from kokoro_align_amd import _lib
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
# (mode, backtrace, positions per tile: 0 = the library's choice)
FORMS = [("auto", "auto", 0), ("tiled", "parallel", 256), ("tiled", "serial", 256), ("tiled", "parallel", 128), ("tiled", "serial", 128),
         ("wave", "serial", 0), ("wave", "parallel", 0), ("wave_exact", "auto", 0)]
eng = _lib.default_engine(0)
t_end = time.time() + budget
n_cases = n_lattices = 0
bad = []
while time.time() < t_end and len(bad) < 10:
    V = int(rng.choice([3, 12, 39, 64]))
    mm = int(rng.choice([1, 2, 3, 4, 4, 4]))
    beam = int(rng.choice([7, 50, 300, 1000, 1000, 1000, 1008, 1009, 1010, 1500, 4000, 100000]))
    zero_labels = bool(rng.random() < 0.25)
    quantised = bool(rng.random() < 0.3)      # true ties
    shapes = []
    for _ in range(int(rng.integers(1, 5))):
        T = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 100, 511, 512, 1000, 2049, 5000, 9000]))
        T += int(rng.integers(0, 3)) if T > 2 else 0
        ratio = float(rng.choice([0.01, 0.1, 0.2, 0.28, 0.5, 1.0, 2.0, 2.9, 3.2]))
        S = int(max(0, min(6000, round(T * ratio / 2))))
        if rng.random() < 0.05:
            S = 0
        shapes.append((T, S))
    lps, labs, wants = [], [], []
    for i, (T, S) in enumerate(shapes):
        s = int(rng.integers(0, 1 << 30))
        lp = O.hash_logprobs(T, V, s)
        if quantised:
            lp = (np.round(lp * 2) / 2).astype(np.float32)
        lab = O.hash_labels(S, V, s) if S else np.zeros(0, np.int32)
        if zero_labels and S:
            lab = lab.copy()
            lab[rng.random(S) < 0.2] = 0
        lps.append(lp)
        labs.append(lab)
        try:
            wants.append(O.ctc_best_path_c(lp, lab, beam, mm))
        except ValueError:
            wants.append(None)     # empty beam
    n_cases += 1
    n_lattices += len(shapes)
    for mode, bt, width in FORMS:
        eng.set_mode(mode)
        eng.set_backtrace(bt)
        eng.set_tile_width(width)
        res, status, total = ka.ctc_best_path_batch(lps, labs, beam, mm, return_status=True)
        for i, (r, st, w) in enumerate(zip(res, status, wants)):
            if w is None:
                if st != -1:
                    bad.append((mode, bt, width, gather, shapes[i], V, beam, mm, "status", int(st)))
            elif st != 0 or not (np.array_equal(r[0], w[0]) and np.array_equal(r[1], w[1]) and
                                 np.array_equal(r[2].view(np.int32), w[2].view(np.int32))):
                first = int(np.argmax(r[0] != w[0])) if st == 0 and r[0].shape == w[0].shape else -1
                bad.append((mode, bt, width, gather, shapes[i], V, beam, mm, "zero" if zero_labels else "", "ties" if quantised else "", int(st), first))
eng.set_mode("auto")
eng.set_backtrace("auto")
eng.set_tile_width(0)
print(f"fuzz seed {seed}: {n_cases} cases, {n_lattices} lattices x {len(FORMS)} forms, mismatches: {len(bad)}")
for b in bad[:10]:
    print("  ", b)
sys.exit(1 if bad else 0)
