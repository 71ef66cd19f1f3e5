#!/usr/bin/env python3
"""Which lattices of the corpus launch differ between kernel forms?  (debugging aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from kokoro_align_amd import workloads as W
from kokoro_align_amd.align import DeviceBatch
shapes, seeds = [], []
for k, (_, sh) in enumerate(W.corpus()):
    shapes += sh; seeds += [W.corpus_seed0(k) + i for i in range(len(sh))]
lps, labs = [], []
for (T, S), seed in zip(shapes, seeds):
    a, b_ = W.device_book([(T, S)], seed0=seed); lps += a; labs += b_
b = DeviceBatch(lps, labs)
e = b.engine
e.set_tile_width(int(os.environ.get("KA_TILE_WIDTH", "0")))     # 0: the library chooses; 128 / 256 forced
def run(mode, bt, split=(-1, -1)):
    e.set_mode(mode); e.set_backtrace(bt); e.set_split(*split)
    st = b.run(raise_on_error=False)
    return [p.clone() for p in b.path], b.total.copy(), st.copy()
ref, rt, _ = run("wave", "serial")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for cfg in [("tiled", "serial")] * reps + [("tiled", "parallel"), ("auto", "auto", (len(shapes), 0)), ("auto", "auto")]:
    p, t, st = run(*cfg)
    bad = [i for i in range(len(shapes)) if not torch.equal(ref[i], p[i])]
    badt = [i for i in range(len(shapes)) if np.float32(rt[i]).view(np.int32) != np.float32(t[i]).view(np.int32)]
    print(cfg, "status!=0:", int((st != 0).sum()), "paths differ:", bad[:10], len(bad), "totals differ:", badt[:10], len(badt), flush=True)
    for i in bad[:3]:
        d = (ref[i] != p[i]).nonzero().flatten()
        print("   lattice", i, "T", shapes[i][0], "first diff frame", int(d[0]), "last", int(d[-1]), "count", len(d), flush=True)
e.set_mode("auto"); e.set_backtrace("auto"); e.set_split(-1, -1); e.set_tile_width(0)
