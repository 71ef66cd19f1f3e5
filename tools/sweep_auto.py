#!/usr/bin/env python3
"""Calibration of KA_MODE_AUTO / KA_BACKTRACE_AUTO on launches of SKEWED lengths: for several sets of
lattices, forward / backtrace kernel time as a function of how many of the longest lattices run tiled (k) and are walked
back chunk-parallel (m) - ka_debug_set_split - next to what the library's cost model picks by itself (k = m = -1).

    python tools/sweep_auto.py > profiles/r03_sweep_auto_skewed.jsonl
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch


def shapes_uniform(n, lo, hi, seed):
    rng = np.random.default_rng(seed)
    return [(int(t), int(0.14 * t)) for t in rng.integers(lo, hi, n)]


cases = [("kokoro", W.kokoro_book()[1]), ("meian", W.meian_book()[1]), ("corpus", [s for _, sh in W.corpus() for s in sh]),
         ("300 long (80k-160k)", shapes_uniform(300, 80000, 160000, 1)), ("300 short (20k-30k)", shapes_uniform(300, 20000, 30000, 2)),
         ("40 long + 1500 short", shapes_uniform(40, 100000, 160000, 3) + shapes_uniform(1500, 20000, 40000, 4)),
         ("2000 mixed (20k-100k)", shapes_uniform(2000, 20000, 100000, 5))]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, shapes in cases:
    if only and only not in name:
        continue
    lps, labs = W.device_book(shapes, seed0=30000)
    b = DeviceBatch(lps, labs)
    e = b.engine
    e.set_mode("auto"); e.set_backtrace("auto"); e.set_profiling(True)
    n = len(shapes)
    ref = None
    grid = [(-1, -1)] + [(k, -1) for k in sorted({0, n // 16, n // 8, n // 4, n // 2, 3 * n // 4, n})] + \
           [(-1, m) for m in sorted({0, n // 16, n // 8, n // 4, n // 2, 3 * n // 4, n})]
    for k, m in grid:
        e.set_split(k, m)
        b.run()
        rows = []
        for _ in range(3):
            b.run()
            rows.append(e.last_kernel_ms())
        best = min(rows, key=lambda r: r["forward"] + r["backtrace"])
        paths = [p.clone() for p in b.path] if ref is None else None
        same = True if ref is None else all(torch.equal(a, c) for a, c in zip(ref, b.path))
        if ref is None:
            ref = paths
        print(json.dumps({"case": name, "lattices": n, "frames": sum(t for t, _ in shapes), "longest": max(t for t, _ in shapes),
                          "n_tiled": k, "n_parallel": m, "forward_ms": round(best["forward"], 3), "backtrace_ms": round(best["backtrace"], 3),
                          "same_paths_as_auto": same}), flush=True)
    e.set_split(-1, -1); e.set_profiling(False)
    del b, lps, labs, ref
    torch.cuda.empty_cache()
