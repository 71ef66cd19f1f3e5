#!/usr/bin/env python3
"""Host-side cost of one launch: how long ka_ctc_best_path_batch_enqueue_f32 takes to return (planning, descriptors, tile tasks,
the enqueue of memsets / copies / kernels) against the launch's wall time, for the book stand-ins and the corpus."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
cases = [("cfg2 lone lattice", [(50000, 5000)], 0, 64), ("kokoro", W.kokoro_book()[1], W.BOOK_SEED0, 39), ("meian", W.meian_book()[1], W.BOOK_SEED0, 39),
         ("corpus", [s for _, sh in W.corpus() for s in sh], W.CORPUS_SEED0, 39)]
for name, shapes, seed0, V in cases:
    lps, labs = W.device_book(shapes, V=V, seed0=seed0)
    b = DeviceBatch(lps, labs)
    b.engine.set_profiling(True)
    b.run()
    rows = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.enqueue()
        t1 = time.perf_counter()
        b.finish()
        t2 = time.perf_counter()
        k = b.engine.last_kernel_ms()
        rows.append({"enqueue_returns_ms": (t1 - t0) * 1e3, "wall_ms": (t2 - t0) * 1e3, "kernels_ms": k["prep"] + k["forward"] + k["backtrace"] + k["gather"],
                     "prep": k["prep"], "forward": k["forward"], "backtrace": k["backtrace"], "gather": k["gather"]})
    best = min(rows, key=lambda r: r["wall_ms"])
    print(json.dumps(dict(case=name, lattices=len(shapes), **{k: round(v, 4) for k, v in best.items()})), flush=True)
    b.engine.set_profiling(False)
    del b, lps, labs
    torch.cuda.empty_cache()
