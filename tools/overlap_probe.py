#!/usr/bin/env python3
"""Does running the forward kernel of one sub-batch beside the backtrace of another pay?  

G engines, each with its own stream and workspace, each owning B/G cfg2 lattices; G host threads loop enqueue + finish K
times, thread g started g/G of a period late so that the phases interleave.  Compared with one engine over all B lattices.

    python tools/overlap_probe.py [B] [K] [G:stagger_ms,...]
"""
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import kokoro_align_amd as ka
from kokoro_align_amd import _lib
from kokoro_align_amd.align import DeviceBatch

T, V, S = 50000, 64, 5000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
lib = ka.load_library()
lps = torch.empty((B, T, V), dtype=torch.float32, device=dev)
labs = torch.empty((B, S), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
assert lib.ka_hash_logprobs_batch_f32(lps.data_ptr(), B, T, V, V, T * V, 0, st) == 0
assert lib.ka_hash_labels_batch_i32(labs.data_ptr(), B, S, V, S, 0, st) == 0
torch.cuda.synchronize()


def run(G, stagger_ms, period_ms=72.0):
    n = B // G
    batches, streams, raw = [], [], []
    for g in range(G):
        b = DeviceBatch([lps[i] for i in range(g * n, (g + 1) * n)], [labs[i] for i in range(g * n, (g + 1) * n)], 1000, 4)
        b.engine = _lib.Engine(0)
        b.engine.set_mode(os.environ.get("PROBE_MODE", "wave"))
        b.engine.set_backtrace(os.environ.get("PROBE_BACKTRACE", "serial"))
        b.engine.set_profiling(os.environ.get("PROBE_PROFILING", "0") == "1")
        b.engine.set_rc_gather(int(os.environ.get("PROBE_RC_GATHER", "-1")))
        b.engine.reserve(b.workspace_bytes() + (1 << 20))
        batches.append(b)
        if os.environ.get("PROBE_OWN_STREAMS", "1") == "1":
            h = ctypes.c_void_p()
            assert lib.ka_stream_create(0, ctypes.byref(h)) == 0
            raw.append(h)
            streams.append(torch.cuda.ExternalStream(h.value, device=dev))
        else:
            streams.append(torch.cuda.current_stream(dev) if g == 0 and os.environ.get('PROBE_NULL_STREAM', '1') == '1' else torch.cuda.Stream(device=dev))
    for b, s in zip(batches, streams):      # warm-up, one after the other
        with torch.cuda.stream(s):
            b.run()
    torch.cuda.synchronize()
    start = threading.Barrier(G + 1)
    done = [0.0] * G

    def worker(g):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[g]):
            start.wait()
            if stagger_ms:
                time.sleep(g * stagger_ms * 1e-3)
            for _ in range(K):
                batches[g].run()
            done[g] = time.perf_counter()

    th = [threading.Thread(target=worker, args=(g,)) for g in range(G)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    el = max(done) - t0
    ok = all(int(b.path[0][-1]) == 2 * S for b in batches)
    res = {"engines": G, "lattices_each": n, "stagger_ms": stagger_ms, "steps": K, "ms_per_step_all": el / K * 1e3,
           "frames_per_s": n * G * T * K / el, "ms_per_8192": el / K * 1e3 * 8192 / (n * G), "ends_ok": ok}
    print(json.dumps(res), flush=True)
    for b in batches:
        b.engine.close()
    for h in raw:
        lib.ka_stream_destroy(0, h)
    del batches
    torch.cuda.empty_cache()
    return res


plan = [(1, 0), (2, 0), (2, 36), (4, 0), (4, 18), (8, 9)]
if len(sys.argv) > 3:     # "G:stagger_ms,G:stagger_ms,..."
    plan = [tuple(int(x) for x in item.split(":")) for item in sys.argv[3].split(",")]
for G, stagger in plan:
    run(G, stagger)
