#!/usr/bin/env python3
"""BASELINE configs[4], the "tiled DP" stress case at full size: T=500000 x V=64, S=50000 (L=100001) with beam_size >= 2L,
i.e. the whole lattice (5e10 cells), through the tile pipeline; checked through size-independent properties (the CPU
reference would need hours): the path is monotone with steps <= 3, ends on the trailing blank, labels match positions, and
the float32 chain of the per-frame scores along the path is the forward pass's best cumulative score bit for bit."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
c = W.CFG5
T, S, V = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (c["T"], c["S"], c["V"])
L = 2 * S + 1
lps, labs = W.device_book([(T, S)], V=V, seed0=5)
b = DeviceBatch(lps, labs, 2 * L + 2)
b.engine.set_profiling(True)
b.engine.set_mode("auto")
t0 = time.perf_counter(); b.run(); first = time.perf_counter() - t0
t0 = time.perf_counter(); b.run(); dt = time.perf_counter() - t0
k = b.engine.last_kernel_ms()
p = b.path[0].cpu().numpy(); l = b.best_labels[0].cpu().numpy(); s = b.best_scores[0].cpu().numpy()
lab = labs[0].cpu().numpy()
ext = np.zeros(L, np.int32); ext[1::2] = lab
d = np.diff(p)
chain = np.add.accumulate(s, dtype=np.float32)[-1]
out = {"workload": f"full lattice T={T} V={V} S={S} (L={L}), beam_size={2 * L + 2} >= 2L: {T * L:.3g} cells", "ms": dt * 1e3, "first_call_ms": first * 1e3,
       "forward_ms": k["forward"], "backtrace_ms": k["backtrace"], "frames_per_s": T / dt, "cells_per_s": T * L / dt,
       "monotone_steps_le_3": bool((d >= 0).all() and (d <= 3).all()), "ends_at_trailing_blank": bool(p[-1] == L - 1), "starts_low": int(p[0]),
       "labels_match_positions": bool(np.array_equal(l, ext[p])), "scores_are_the_emissions": bool(np.array_equal(s, lps[0].cpu().numpy()[np.arange(T), l])),
       "chain_equals_total_bitwise": bool(np.float32(chain).view(np.int32) == np.float32(b.total[0]).view(np.int32))}
print(json.dumps(out))
