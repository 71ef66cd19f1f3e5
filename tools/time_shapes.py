#!/usr/bin/env python3
"""Kernel times of the batch forms for other lattice shapes than cfg2 (how much of a frame is band handling?).

    python tools/time_shapes.py [lattices] [T] [S ...]
"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import kokoro_align_amd as ka
from kokoro_align_amd.align import DeviceBatch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
Ss = [int(a) for a in sys.argv[3:]] or [499, 5000]
V = 64
lib = ka.load_library()
lps = torch.empty((B, T, V), dtype=torch.float32, device="cuda")
assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, 0, None) == 0
for S in Ss:
    labs = torch.empty((B, S), dtype=torch.int32, device="cuda")
    assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 0, None) == 0
    torch.cuda.synchronize()
    batch = DeviceBatch([lps[i] for i in range(B)], [labs[i] for i in range(B)], 1000, 4)
    batch.engine.set_mode("wave")
    batch.engine.set_profiling(True)
    for rep in range(2):
        batch.run()
    k = batch.engine.last_kernel_ms()
    fw, bt = k["forward"], k["backtrace"]
    print(f"S={S} L={2*S+1}: forward {fw:.2f} ms  backtrace {bt:.2f} ms  ({fw*1e6/(B*T):.3f} + {bt*1e6/(B*T):.3f} ns per frame)")
    del batch
