#!/usr/bin/env python3
"""Where the tiles of a lattice spend their time (KA_TP_VERIFY=4): per tile, frames, alive time, time waiting for the
tile below, ns per frame while not waiting.   KA_TP_VERIFY=4 python tools/tile_stats.py [T S V beam]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KA_TP_VERIFY", "4")
import numpy as np
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
T, S, V, beam = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (50000, 5000, 64, 1000)
lps, labs = W.device_book([(T, S)], V=V, seed0=0)
b = DeviceBatch(lps, labs, beam)
b.engine.set_mode("tiled")
b.engine.set_profiling(True)
b.run(); b.run()
print("forward_ms", b.engine.last_kernel_ms()["forward"])
out = np.zeros((4096, 8), np.uint64)
n = b.engine.lib.ka_debug_tile_stats(b.engine.handle, out.ctypes.data, 4096)
t0 = int(out[:n, 7].min())
print(" tile   t_in  t_end frames  start_us alive_us wait_us waits ns/frame(busy)")
for r in out[:n]:
    fr = int(r[3]) - int(r[2])
    print(f"{int(r[1]):5d} {int(r[2]):6d} {int(r[3]):6d} {fr:6d} {(int(r[7]) - t0) / 100:9.1f} {int(r[5]) / 100:8.1f} {int(r[4]) / 100:7.1f} {int(r[6]):5d} {(int(r[5]) - int(r[4])) * 10 / max(fr, 1):8.1f}")
b.engine.set_mode("auto")
