#!/usr/bin/env python3
"""The corpus dataset by dataset on four engines / streams (StreamedAligner), many times: several tile pipelines share the
device, every engine reuses one workspace for launches of different shapes.  Every run's paths against a reference launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
from kokoro_align_amd.streams import StreamedAligner
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
datasets = W.corpus()
per_ds = []
for k, (name, shapes) in enumerate(datasets):
    lps, labs = W.device_book(shapes, seed0=W.corpus_seed0(k))
    per_ds.append((lps, labs, shapes))
# reference: every dataset alone, one wavefront per lattice, serial backtrace
ref = []
for lps, labs, shapes in per_ds:
    b = DeviceBatch(lps, labs)
    b.engine.set_mode("wave"); b.engine.set_backtrace("serial")
    b.run()
    ref.append([p.clone() for p in b.path])
    b.engine.set_mode("auto"); b.engine.set_backtrace("auto")
sa = StreamedAligner(4)
batches = sa.bind([DeviceBatch(lps, labs) for lps, labs, _ in per_ds])
bad = 0
t0 = time.time()
for r in range(reps):
    st = sa.run(batches, repeat=2, raise_on_error=False)
    for k, (b, s) in enumerate(zip(batches, st)):
        if (s != 0).any():
            bad += 1
            print("rep", r, "dataset", k, "status", s.tolist(), flush=True)
        elif not all(torch.equal(a, c) for a, c in zip(ref[k], b.path)):
            bad += 1
            print("rep", r, "dataset", k, "paths differ", flush=True)
print(f"{reps} x 2 passes over {len(per_ds)} datasets on 4 streams in {time.time() - t0:.1f} s: {bad} bad")
sa.close()
