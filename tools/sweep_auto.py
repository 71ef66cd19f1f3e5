#!/usr/bin/env python3
"""Where does the tile pipeline stop paying off?  n cfg2-shaped lattices (reduced T to keep it short) in the tiled form and
in the one-wavefront form, with the parallel and the serial backtrace: ms per launch."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
T, S, V = 20000, 2000, 64
for n in (16, 64, 128, 192, 256, 384, 512, 1024, 2048):
    lps, labs = W.device_book([(T, S)] * n, V=V, seed0=500)
    b = DeviceBatch(lps, labs)
    b.engine.set_profiling(True)
    row = {"lattices": n}
    for mode, bt in (("tiled", "parallel"), ("wave", "parallel"), ("wave", "serial"), ("tiled", "serial")):
        b.engine.set_mode(mode); b.engine.set_backtrace(bt)
        b.run()
        t0 = time.perf_counter()
        for _ in range(2):
            b.run()
        dt = (time.perf_counter() - t0) / 2
        k = b.engine.last_kernel_ms()
        row[f"{mode}+{bt}"] = round(dt * 1e3, 2)
        row[f"{mode}+{bt} fwd/bt"] = (round(k["forward"], 2), round(k["backtrace"], 2))
    print(json.dumps(row), flush=True)
    b.engine.set_mode("auto"); b.engine.set_backtrace("auto")
    del lps, labs, b
    torch.cuda.empty_cache()
