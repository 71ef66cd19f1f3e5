#!/usr/bin/env python3
"""Tile statistics (ka_engine_set_verify(4)) of the corpus launch as KA_MODE_AUTO runs it (the longest ~320 chapters in 256-position
tiles beside the others' one-wavefront kernels): how long a tile is alive per frame against how long its frames take - what share
of a slot's time is frames, code between blocks and waiting for the tile below."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
shapes = [s for _, sh in W.corpus() for s in sh]
lps, labs = W.device_book(shapes, seed0=W.CORPUS_SEED0)
b = DeviceBatch(lps, labs)
e = b.engine
e.set_mode(sys.argv[1] if len(sys.argv) > 1 else "auto")
e.set_tile_width(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
e.set_tile_lds(int(sys.argv[3]) if len(sys.argv) > 3 else 0)      # bytes of LDS a tile workgroup asks for (0: the library's rule)
e.set_profiling(True)
e.set_verify(4)
b.run(); b.run()
print("forward_ms", e.last_kernel_ms()["forward"], "backtrace_ms", e.last_kernel_ms()["backtrace"])
out = np.zeros((60000, 8), np.uint64)
n = e.lib.ka_debug_tile_stats(e.handle, out.ctypes.data, 60000)
o = out[:n]
t_in = (o[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
frames = np.maximum(o[:, 3].astype(np.int64) - t_in, 1)
cyc_frames = (o[:, 4] >> np.uint64(32)).astype(np.float64)
alive_ns = o[:, 5].astype(np.float64) * 10
clock = o[:, 7].astype(np.float64) / np.maximum(alive_ns, 1)
print(f"tiles {n}, frames per tile mean {frames.mean():.0f}; shader clock GHz mean {clock.mean():.3f}")
print(f"alive ns per frame: mean {np.mean(alive_ns / frames):.1f}, weighted {alive_ns.sum() / frames.sum():.1f}")
print(f"cycles per frame inside the frame blocks: weighted {cyc_frames.sum() / frames.sum():.1f} = {cyc_frames.sum() / frames.sum() / clock.mean():.1f} ns")
print(f"tile-frames {frames.sum() / 1e6:.1f} M; sum of alive time {alive_ns.sum() / 1e6:.1f} ms of slot time")
e.set_verify(0); e.set_mode("auto"); e.set_tile_width(0); e.set_tile_lds(0)
