#!/usr/bin/env python3
"""Where the tiles of a lattice spend their time (ka_engine_set_verify(4)): per tile, frames, alive time, time waiting for the
tile below, ns per frame while not waiting.   python tools/tile_stats.py [T S V beam]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
T, S, V, beam = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (50000, 5000, 64, 1000)
lps, labs = W.device_book([(T, S)], V=V, seed0=0)
b = DeviceBatch(lps, labs, beam)
b.engine.set_mode("tiled")
b.engine.set_tile_width(int(os.environ.get("KA_TILE_WIDTH", "0")))
b.engine.set_profiling(True)
b.engine.set_verify(4)
b.run(); b.run()
print("forward_ms", b.engine.last_kernel_ms()["forward"])
out = np.zeros((4096, 8), np.uint64)
n = b.engine.lib.ka_debug_tile_stats(b.engine.handle, out.ctypes.data, 4096)
t0 = 0
comp_hw = (out[:n, 2] >> np.uint64(32)).astype(np.int64)      # HW_ID of the compute wavefront (two-wavefront tiles)
out[:n, 2] &= np.uint64(0xffffffff)
print("cycles per frame inside the frame blocks (tiles 0..):", np.round((out[:n, 4] >> np.uint64(32)).astype(float) / np.maximum(1, (out[:n, 3] - out[:n, 2]).astype(float)), 1)[:12])
out[:n, 4] &= np.uint64(0xffffffff)
print("shader clock while the tiles ran (GHz):", np.round(out[:n, 7].astype(float) / (out[:n, 5].astype(float) * 10), 2)[:12])
print(" tile   t_in  t_end frames  start_us alive_us wait_us waits ns/frame(busy)  xcc se cu simd slot")
for i, r in enumerate(out[:n]):
    fr = int(r[3]) - int(r[2])
    print(f"{int(r[1]):5d} {int(r[2]):6d} {int(r[3]):6d} {fr:6d} {0.0:9.1f} {int(r[5]) / 100:8.1f} {int(r[4]) / 100:7.1f} {int(r[6]) & 0xffffffff:5d} {(int(r[5]) - int(r[4])) * 10 / max(fr, 1):8.1f}"
          f"      {(int(r[6]) >> 48) & 0xf:3d} {(int(r[6]) >> 45) & 7:2d} {(int(r[6]) >> 40) & 0xf:2d} {(int(r[6]) >> 36) & 3:4d} {(int(r[6]) >> 32) & 0xf:4d}"
          f"   compute simd {(int(comp_hw[i]) >> 4) & 3} slot {int(comp_hw[i]) & 0xf}")
import collections
simd = collections.Counter(((int(r[6]) >> 48) & 0xf, (int(r[6]) >> 45) & 7, (int(r[6]) >> 40) & 0xf, (int(r[6]) >> 36) & 3) for r in out[:n])
cu = collections.Counter(k[:3] for k in simd.elements())
print("tiles per SIMD (max):", max(simd.values()), " tiles per CU (max):", max(cu.values()), " distinct CUs:", len(cu))
b.engine.set_mode("auto")
b.engine.set_tile_width(0)
b.engine.set_verify(0)
