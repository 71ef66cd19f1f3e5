#!/usr/bin/env python3
"""Tile width of the tiled form (256 positions, 128, 128 with the emissions looked up by the feeder wavefront) against the
number of lattices in the launch: forward kernel time of the first K chapters of the corpus stand-in, all tiled, for each -
the data behind the engine's choice (ka_engine.hip, "tile width").   python tools/sweep_width.py [K ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
Ks = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64, 96, 128, 160, 200, 250, 320, 462]
corpus = [s for _, sh in W.corpus() for s in sh]
for K in Ks:
    shapes = corpus[:K]
    lps, labs = W.device_book(shapes, seed0=W.CORPUS_SEED0)
    b = DeviceBatch(lps, labs)
    b.engine.set_mode("tiled")
    b.engine.set_profiling(True)
    row = {"lattices": K, "frames": sum(t for t, _ in shapes), "longest": max(t for t, _ in shapes)}
    ref = None
    for width in (256, 128):
        b.engine.set_tile_width(width)
        b.run()
        ms = []
        for _ in range(3):
            b.run()
            ms.append(b.engine.last_kernel_ms()["forward"])
        row[f"forward_ms_{width}"] = round(min(ms), 4)
        paths = [p.clone() for p in b.path]
        if ref is None:
            ref = paths
        else:
            row["same_paths"] = row.get("same_paths", True) and all(torch.equal(a, c) for a, c in zip(ref, paths))
    print(json.dumps(row), flush=True)
    b.engine.set_tile_width(0); b.engine.set_mode("auto"); b.engine.set_profiling(False)
    del b, lps, labs
    torch.cuda.empty_cache()
