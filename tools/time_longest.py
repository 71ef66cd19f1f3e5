#!/usr/bin/env python3
"""The longest chapter of a book alone against the whole book: how much of a book's forward pass is the chain of its longest
chapter, and how much the other chapters' tiles cost it (shared SIMDs, LDS pipes, memory).   python tools/time_longest.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch


def fwd(shapes, seed0, reps=3):
    lps, labs = W.device_book(shapes, seed0=seed0)
    b = DeviceBatch(lps, labs)
    b.engine.set_mode("tiled")
    b.engine.set_profiling(True)
    b.run()
    ms = []
    for _ in range(reps):
        b.run()
        ms.append(b.engine.last_kernel_ms()["forward"])
    b.engine.set_mode("auto"); b.engine.set_profiling(False)
    del b, lps, labs
    torch.cuda.empty_cache()
    return min(ms)


for name, shapes in (("kokoro", W.kokoro_book()[1]), ("meian", W.meian_book()[1])):
    order = sorted(range(len(shapes)), key=lambda i: -shapes[i][0])
    row = {"book": name, "chapters": len(shapes), "longest": shapes[order[0]]}
    for k in (1, 2, 4, 8, 16, 32, len(shapes)):
        sub = [shapes[i] for i in order[:k]]
        row[f"forward_ms_longest_{k}"] = round(fwd(sub, W.BOOK_SEED0), 4)
    print(json.dumps(row), flush=True)
