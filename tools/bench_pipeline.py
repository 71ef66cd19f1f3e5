#!/usr/bin/env python3
"""Stage times of the device-resident pipeline around the hot path on a synthetic
book: MFCC segments -> AudioToChar (PyTorch-ROCm / MIOpen LSTM) -> HIP log-softmax -> batched CTC best path.

    python tools/bench_pipeline.py [chapters] [frames_per_chapter] [segments_per_lstm_call] [across|device|device_steps]

Random-init network of the reference's architecture (train.py:54-65), random MFCC-shaped input cut into
segments of 200-1200 frames like the reference's splitter produces, labels = S ~ 0.14*T phonemes per chapter.
Prints one JSON object.  This is a side measurement: `value` in bench.py is the DP alone.
"""
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch

import kokoro_align_amd as ka
from kokoro_align_amd.model import load_model, segment_logits, lstm_logits_device

chapters = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 43000
BATCH = int(sys.argv[3]) if len(sys.argv) > 3 else 128      # segments per LSTM call (the reference uses 128)
ACROSS = len(sys.argv) > 4 and sys.argv[4] == "across"     # batch the segments of all chapters together (MIOpen)
DEVICE_LSTM = len(sys.argv) > 4 and sys.argv[4] in ("device", "device_steps")   # kokoro_align_amd.model.lstm_logits_device
PERSISTENT = len(sys.argv) > 4 and sys.argv[4] == "device"   # ka_lstm_layer_f32 (one launch per layer) vs one GEMM + ka_lstm_step_f32 per step
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
torch.manual_seed(0)
model = load_model(None, device=dev)
V = model.dense.out_features


def sync():
    torch.cuda.synchronize()


books = []
for c in range(chapters):
    T = int(frames * rng.uniform(0.6, 1.4))
    segs, left = [], T
    while left > 0:
        n = int(min(left, rng.integers(200, 1200)))
        segs.append(torch.from_numpy(rng.standard_normal((n, 40)).astype(np.float32)))
        left -= n
    books.append((T, segs, rng.integers(1, V, size=int(0.14 * T)).astype(np.int32)))

# the book as one IndexDataArray (data + cumulative segment ends), which is how the MFCC files store it
book_data = torch.cat([s for _, segs, _ in books for s in segs], dim=0)
book_ends = np.cumsum([int(s.shape[0]) for _, segs, _ in books for s in segs])
times = {}
for rep in range(2):      # second pass is the measured one (MIOpen find, allocator warm-up)
    sync(); t0 = time.perf_counter()
    if DEVICE_LSTM:   # library GEMMs + HIP recurrence kernels, every segment of the book at once
        all_logits = lstm_logits_device(model, book_data, book_ends, device=dev, persistent=PERSISTENT)
        logits, k = [], 0
        for T, _, _ in books:
            logits.append(all_logits[k:k + T])
            k += T
    elif ACROSS:   # segments of ALL chapters share the LSTM calls (a time step costs the same for 60 or 4000 sequences)
        flat = [s for _, segs, _ in books for s in segs]
        outs = segment_logits(model, flat, device=dev, batch_size=BATCH)
        logits, k = [], 0
        for _, segs, _ in books:
            logits.append(torch.cat(outs[k:k + len(segs)], dim=0))
            k += len(segs)
    else:        # one call per chapter, like the reference's per-file predict (train.py:201-231)
        logits = [torch.cat(segment_logits(model, segs, device=dev, batch_size=BATCH), dim=0) for _, segs, _ in books]
    sync(); t1 = time.perf_counter()
    lps = [ka.log_softmax_device(lg) for lg in logits]
    sync(); t2 = time.perf_counter()
    res = ka.ctc_best_path_device(lps, [lab for _, _, lab in books])
    sync(); t3 = time.perf_counter()
    times = {"acoustic_model_ms": (t1 - t0) * 1e3, "log_softmax_ms": (t2 - t1) * 1e3, "ctc_best_path_ms": (t3 - t2) * 1e3}
total_frames = sum(T for T, _, _ in books)
ends_ok = all(int(p[-1]) == 2 * len(lab) for (p, _, _), (_, _, lab) in zip(res, books))
print(json.dumps({"segments_per_lstm_call": BATCH, "acoustic_model": ("device_lstm_persistent" if PERSISTENT else "device_lstm_steps") if DEVICE_LSTM else ("miopen_across_chapters" if ACROSS else "miopen_per_chapter"), "chapters": chapters, "frames": total_frames, "audio_hours": total_frames * 0.0116 / 3600, "vocab": V,
                  **times, "frames_per_s_end_to_end": total_frames / (sum(times.values()) * 1e-3),
                  "all_paths_end_at_the_trailing_blank": bool(ends_ok)}))
