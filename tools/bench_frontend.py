#!/usr/bin/env python3
"""Stage times of the device front end on a synthetic recording: silence splitting
(kokoro_align/preprocess.py:51-97) and the MFCCs of every segment (preprocess.py:110-127).

    python tools/bench_frontend.py [hours_of_audio] [cpu_sample_seconds]

Waveform: noise bursts (0.4-6 s) separated by pauses (0.05-1.2 s) over a noise floor, 22 050 Hz, generated on the
device.  The CPU leg times the NumPy restatement (oracle/frontend_oracle.py) on a bounded sample of the same
recording.  Prints one JSON object.  A side measurement: `value` in bench.py is the DP alone.
"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch

from kokoro_align_amd import preprocess as P
from oracle import frontend_oracle as F

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cpu_s = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
sr = 22050
n = int(hours * 3600 * sr)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
rng = np.random.default_rng(0)
# envelope: one value per 220 samples (10 ms), bursts and pauses
env = np.full(n // 220 + 1, 0.002, dtype=np.float32)
t = 0.0
while t < hours * 3600:
    d = float(rng.uniform(0.4, 6.0))
    env[int(t * sr / 220):int((t + d) * sr / 220)] = float(rng.uniform(0.05, 0.6))
    t += d + float(rng.uniform(0.05, 1.2))
y = torch.randn(n, generator=g, device=dev, dtype=torch.float32)
y *= torch.repeat_interleave(torch.from_numpy(env).to(dev), 220)[:n]
torch.cuda.synchronize()

par = F.split_parameters(sr, 512)
times = {}
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pts = P.get_split_points(y, **par) * par["window_size"]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ends = np.append(pts, n).astype(np.int64)
    mf, idx = P.mfcc_segments(y, ends)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    times = {"split_points_ms": (t1 - t0) * 1e3, "mfcc_ms": (t2 - t1) * 1e3}

# CPU leg on a bounded sample (same NumPy work as the reference's get_split_points; float64 NumPy MFCC)
m = min(n, int(cpu_s * sr))
yh = y[:m].cpu().numpy()
c0 = time.perf_counter()
cp = F.split_points(yh, **par) * par["window_size"]
c1 = time.perf_counter()
ce = np.append(cp, m)
a = 0
worst = 0.0
k = 0
for e in ce.tolist():
    want = F.mfcc(yh[a:e])
    a = e
c2 = time.perf_counter()
# accuracy of the device MFCCs on the first segments of the recording
a = 0
for e, stop in list(zip(ends.tolist(), idx.tolist()))[:20]:
    want = F.mfcc(y[a:e].cpu().numpy())
    worst = max(worst, float(np.abs(mf[k:stop].cpu().numpy() - want).max()))
    k, a = stop, e
frames = int(idx[-1])
print(json.dumps({"audio_hours": hours, "samples": n, "segments": int(len(ends)), "mfcc_frames": frames, **times,
                  "audio_seconds_per_second": hours * 3600 / ((times["split_points_ms"] + times["mfcc_ms"]) * 1e-3),
                  "cpu_numpy": {"sample_audio_s": m / sr, "split_points_ms": (c1 - c0) * 1e3, "mfcc_float64_ms": (c2 - c1) * 1e3,
                                "audio_seconds_per_second": (m / sr) / (c2 - c0), "cores": 1},
                  "max_abs_mfcc_error_vs_float64_first_20_segments": worst}))
