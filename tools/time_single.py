#!/usr/bin/env python3
"""Forward/backtrace kernel times of single lattices in a given mode (no result check: for timing experiments with
KA_LIBRARY pointing at a variant build)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
mode = sys.argv[1] if len(sys.argv) > 1 else "tiled"
for name, T, S, V, beam in [("one tile T=50000 S=100", 50000, 100, 64, 1000), ("two tiles S=200", 50000, 200, 64, 1000), ("four tiles S=500", 50000, 500, 64, 1000), ("cfg2", 50000, 5000, 64, 1000), ("cfg1", 81140, 2000, 39, 1000), ("full 50000x5000", 50000, 5000, 64, 30000)]:
    lps, labs = W.device_book([(T, S)], V=V, seed0=0)
    b = DeviceBatch(lps, labs, beam)
    b.engine.set_mode(mode)
    b.engine.set_tile_width(int(os.environ.get("KA_TILE_WIDTH", "0")))
    b.engine.set_tile_lds(int(os.environ.get("KA_TILE_LDS", "0")))
    b.engine.set_profiling(True)
    b.run(raise_on_error=False)
    ms = []
    for _ in range(3):
        b.run(raise_on_error=False)
        ms.append(b.engine.last_kernel_ms()["forward"])
    print(json.dumps({"case": name, "mode": mode, "lib": os.path.basename(os.environ.get("KA_LIBRARY", "default")), "forward_ms": min(ms), "ns_per_frame": min(ms) * 1e6 / T, "status": int(b.status[0])}), flush=True)
    b.engine.set_tile_width(0); b.engine.set_mode("auto"); b.engine.set_tile_lds(0)
