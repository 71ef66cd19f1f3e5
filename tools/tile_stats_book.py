#!/usr/bin/env python3
"""Tile statistics (ka_engine_set_verify(4)) of the longest chapter inside a book launch: shader clock while its tiles ran,
cycles per frame in the frame blocks, busy ns per frame, time at barriers.   python tools/tile_stats_book.py [kokoro|meian] [K longest chapters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
book = sys.argv[1] if len(sys.argv) > 1 else "kokoro"
shapes = {"kokoro": W.kokoro_book, "meian": W.meian_book}[book]()[1]
order = sorted(range(len(shapes)), key=lambda i: -shapes[i][0])
K = int(sys.argv[2]) if len(sys.argv) > 2 else len(shapes)
sub = [shapes[i] for i in order[:K]]
lps, labs = W.device_book(sub, seed0=W.BOOK_SEED0)
b = DeviceBatch(lps, labs)
b.engine.set_mode("tiled")
b.engine.set_tile_lds(int(os.environ.get("KA_TILE_LDS", "0")))
b.engine.set_profiling(True)
b.engine.set_verify(4)
b.run(); b.run()
print("chapters", K, "forward_ms", b.engine.last_kernel_ms()["forward"])
out = np.zeros((20000, 8), np.uint64)
n = b.engine.lib.ka_debug_tile_stats(b.engine.handle, out.ctypes.data, 20000)
o = out[:n]
lat = o[:, 0].astype(np.int64)
t_in = (o[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
t_end = o[:, 3].astype(np.int64)
frames = t_end - t_in
cyc_frames = (o[:, 4] >> np.uint64(32)).astype(np.float64)
alive_ticks = o[:, 5].astype(np.float64)
clock = o[:, 7].astype(np.float64) / (alive_ticks * 10)
for sel, name in ((lat == 0, "longest chapter"), (lat >= 0, "all chapters")):
    print(f"{name}: tiles {int(sel.sum())}, shader clock GHz mean {clock[sel].mean():.3f} min {clock[sel].min():.3f}; cycles per frame in frame blocks mean {np.mean(cyc_frames[sel] / np.maximum(frames[sel], 1)):.1f}")
sel = lat == 0
idx = np.argsort(o[sel, 1].astype(np.int64))
tiles = o[sel][idx]
fr = (tiles[:, 3].astype(np.int64) - (tiles[:, 2] & np.uint64(0xffffffff)).astype(np.int64))
alive_us = tiles[:, 5].astype(np.float64) / 100
print("longest chapter: first tiles' alive_us per frame (ns):", np.round(alive_us[:12] * 1000 / fr[:12], 1))
d = np.diff(alive_us + 0)      # (all tiles start at ~the same time when resident; differences ~ chain progress)
b.engine.set_verify(0); b.engine.set_mode("auto"); b.engine.set_tile_lds(0)
