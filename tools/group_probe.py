#!/usr/bin/env python3
"""Does splitting a book's chapters into length groups, each its own launch on its own stream, hide the backtrace of the
shorter groups behind the forward pass of the longest?   python tools/group_probe.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
from kokoro_align_amd.streams import StreamedAligner
cases = [("kokoro", W.kokoro_book()[1], W.BOOK_SEED0), ("meian", W.meian_book()[1], W.BOOK_SEED0), ("corpus", [s for _, sh in W.corpus() for s in sh], W.CORPUS_SEED0)]
for name, shapes, seed0 in cases:
    lps, labs = W.device_book(shapes, seed0=seed0)
    one = DeviceBatch(lps, labs)
    one.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        one.run()
    torch.cuda.synchronize(); base = (time.perf_counter() - t0) / 5
    ref = [p.clone() for p in one.path]
    order = sorted(range(len(shapes)), key=lambda i: -shapes[i][0])
    row = {"case": name, "one_launch_ms": round(base * 1e3, 2)}
    for cuts in ((0.34,), (0.2, 0.5), (0.25,), (0.5,), (0.15, 0.4, 0.7)):
        bounds = [0] + [int(len(order) * c) for c in cuts] + [len(order)]
        groups = [order[bounds[j]:bounds[j + 1]] for j in range(len(bounds) - 1)]
        sa = StreamedAligner(len(groups))
        batches = sa.bind([DeviceBatch([lps[i] for i in g], [labs[i] for i in g]) for g in groups])
        sa.run(batches)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sa.run(batches, repeat=5)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        same = all(torch.equal(ref[i], b.path[k]) for g, b in zip(groups, batches) for k, i in enumerate(g))
        row[f"groups cut at {cuts}"] = (round(dt * 1e3, 2), same)
        sa.close()
    print(json.dumps(row), flush=True)
    del one, lps, labs
    torch.cuda.empty_cache()
