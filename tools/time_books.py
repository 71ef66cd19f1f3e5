#!/usr/bin/env python3
"""Forward / backtrace kernel times of the two book stand-ins and the corpus as ONE launch each, in a given mode
(tiled|auto|wave) - for timing experiments on the tile kernel.   python tools/time_books.py [mode] [backtrace]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
mode = sys.argv[1] if len(sys.argv) > 1 else "tiled"
bt = sys.argv[2] if len(sys.argv) > 2 else "auto"
lds = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cases = [("kokoro", W.kokoro_book()[1], W.BOOK_SEED0), ("meian", W.meian_book()[1], W.BOOK_SEED0)]
corpus = [s for _, sh in W.corpus() for s in sh]
cases.append(("corpus", corpus, W.CORPUS_SEED0))
for name, shapes, seed0 in cases:
    lps, labs = W.device_book(shapes, seed0=seed0)
    b = DeviceBatch(lps, labs)
    b.engine.set_mode(mode)
    b.engine.set_backtrace(bt)
    b.engine.set_tile_lds(lds)
    b.engine.set_tile_width(int(os.environ.get("KA_TILE_WIDTH", "0")))
    b.engine.set_profiling(True)
    b.run()
    rows = []
    for _ in range(3):
        b.run()
        rows.append(b.engine.last_kernel_ms())
    best = min(rows, key=lambda k: k["forward"] + k["backtrace"])
    ok = all(int(p[-1]) == 2 * s for p, (_, s) in zip(b.path, shapes))
    print(json.dumps({"case": name, "mode": mode, "backtrace": bt, "tile_lds": lds, "lattices": len(shapes), "frames": sum(t for t, _ in shapes),
                      "longest": max(t for t, _ in shapes), "forward_ms": best["forward"], "backtrace_ms": best["backtrace"], "ends_ok": ok}), flush=True)
    b.engine.set_tile_width(0); b.engine.set_mode("auto"); b.engine.set_backtrace("auto"); b.engine.set_profiling(False); b.engine.set_tile_lds(0)
    del b, lps, labs
    torch.cuda.empty_cache()
