#!/usr/bin/env python3
"""Chunk entries and chunk maps of the parallel backtrace against the true path (oracle)."""
import faulthandler, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.dump_traceback_later(90, exit=True)
import numpy as np, torch
import kokoro_align_amd as ka
from kokoro_align_amd import _lib
from oracle import oracle as O
T, V, S, beam = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (3000, 39, 700, 1000)
mode = sys.argv[5] if len(sys.argv) > 5 else "wave"
eng = _lib.default_engine(0)
eng.set_mode(mode); eng.set_backtrace("parallel")
lp = O.hash_logprobs(T, V, 3); lab = O.hash_labels(S, V, 3)
want = O.ctc_best_path_c(lp, lab, beam, 4)[0]
dlp = torch.from_numpy(lp).cuda()
(p, l, s), = ka.ctc_best_path_device([dlp], [lab], beam_size=beam)
got = p.cpu().numpy()
nck = (T - 1) // 32 + 1; nsup = (nck + 31) // 32
R = 1024 if mode == "wave" else None
ent = np.zeros(nck + nsup, np.int32)
m0 = np.zeros(nck * 4096, np.uint8)
n = eng.lib.ka_debug_chunk_entries(eng.handle, ent.ctypes.data, ent.size, m0.ctypes.data, m0.size)
print("entries returned", n, "path equal", np.array_equal(got, want))
te = np.minimum(np.arange(nck) * 32 + 31, T - 1)
true_entry = want[te]
bad = np.nonzero(ent[:nck] != true_entry)[0]
print("chunks", nck, "supers", nsup, "wrong chunk entries", len(bad), bad[:20], "true", true_entry[bad[:8]], "got", ent[bad[:8]])
print("super entries", ent[nck:], "true", [int(want[min(min(s * 32 + 32, nck) * 32 - 1, T - 1)]) for s in range(nsup)])
if R:
    m0 = m0[:nck * R].reshape(nck, R)
    # the map of the cell on the true path: rise over chunk c = path[te(c)] - path[te(c-1)]
    wrong = []
    for c in range(1, nck):
        pe = int(want[te[c]]); rise = pe - int(want[te[c - 1]])
        if m0[c, pe & (R - 1)] != rise:
            wrong.append((c, pe, rise, int(m0[c, pe & (R - 1)])))
    print("maps wrong on the true path:", len(wrong), wrong[:12])
eng.set_mode("auto"); eng.set_backtrace("auto")

# ---- every position of some chunks against a dense NumPy DP with back-pointers (no band: L <= beam) ----
L = 2 * S + 1
if L <= beam and R:
    ext = np.zeros(L, np.int64); ext[1::2] = lab
    NEG = np.float32(-np.inf)
    sc = np.full(L, NEG, np.float32); sc[0] = 0
    bp = np.zeros((T, L), np.int8)
    for t in range(T):
        e = lp[t, ext]
        cand = np.full((4, L), NEG, np.float32)
        for j in range(4):
            cand[j, j:] = sc[:L - j] + e[j:]
        cand[2, ext == 0] = NEG
        mv = np.argmax(cand, axis=0)
        sc = cand[mv, np.arange(L)]
        bp[t] = mv
    for c in (28, 29, 44):
        pos = np.arange(L)
        q = pos.copy()
        for t in range(min(c * 32 + 31, T - 1), c * 32 - 1, -1):
            q = q - bp[t, q]
        true_rise = pos - q
        got_rise = m0[c, pos & (R - 1)].astype(int)
        live = np.isfinite(sc) | True
        bad = np.nonzero(true_rise != got_rise)[0]
        print(f"chunk {c}: {len(bad)} of {L} positions differ; first {bad[:24]}")
        print("   true", true_rise[bad[:16]], "got", got_rise[bad[:16]])
