"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes loader for the C oracle (oracle/ctc_oracle.c) plus two NumPy
restatements of the reference's CTC best-path:

* ``ctc_best_path_c``      — the C dense-band restatement (fast; the checker used by the
                             GPU parity tests and by ``__graft_entry__.smoke()``).
* ``ctc_best_path_numpy``  — a per-frame NumPy port that issues the same kind of NumPy
                             work per frame as kokoro_align/align.py:62-93 (4x candidate
                             scatter, argmax over moves, choose, compaction).  This is the
                             "NumPy CPU path" that bench.py's ``cpu_baseline`` times on the
                             GPU box's host cores (the reference itself cannot travel).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (kokoro-align_amd/) never does.

Parity pin: both functions are checked bit-for-bit against the goldens under
tests/golden/ that were produced by the imported reference (tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KAO_OK = 0
KAO_EMPTY_BEAM = -1
KAO_BAD_ARGS = -2


def build(force=False):
    """Compile oracle/ctc_oracle.c -> oracle/libka_oracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "libka_oracle.so")
    src = os.path.join(_HERE, "ctc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libka_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(so)
        c_i64, c_i32, c_u64 = ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64
        vp = ctypes.c_void_p
        L.kao_ctc_best_path_f32.restype = ctypes.c_int
        L.kao_ctc_best_path_f32.argtypes = [vp, c_i64, c_i32, c_i64, vp, c_i64, c_i32, c_i32,
                                            vp, vp, vp, vp, vp]
        L.kao_band_cells.restype = c_i64
        L.kao_band_cells.argtypes = [c_i64, c_i64, c_i32]
        L.kao_hash_logprobs_f32.restype = None
        L.kao_hash_logprobs_f32.argtypes = [vp, c_i64, c_i32, c_i64, c_u64]
        L.kao_hash_labels_i32.restype = None
        L.kao_hash_labels_i32.argtypes = [vp, c_i64, c_i32, c_u64]
        _LIB = L
    return _LIB


def ctc_best_path_c(log_probs, labels, beam_size=1000, max_move=4, return_total=False):
    """C oracle for kokoro_align/align.py:43-109.  Raises ValueError on the empty-beam
    condition exactly where the reference does (align.py:101)."""
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    lab = np.ascontiguousarray(labels, dtype=np.int32)
    T, V = lp.shape
    S = lab.shape[0]
    path = np.empty(T, np.int32)
    lout = np.empty(T, np.int32)
    sout = np.empty(T, np.float32)
    total = ctypes.c_float(0.0)
    end = ctypes.c_int64(-1)
    rc = lib().kao_ctc_best_path_f32(lp.ctypes.data, T, V, V, lab.ctypes.data, S,
                                     int(beam_size), int(max_move),
                                     path.ctypes.data, lout.ctypes.data, sout.ctypes.data,
                                     ctypes.addressof(total), ctypes.addressof(end))
    if rc == KAO_EMPTY_BEAM:
        raise ValueError("attempt to get argmax of an empty sequence")
    if rc != KAO_OK:
        raise RuntimeError(f"oracle: bad arguments (rc={rc})")
    if return_total:
        return path, lout, sout, np.float32(total.value), int(end.value)
    return path, lout, sout


def band_cells(T, S, beam_size=1000):
    return int(lib().kao_band_cells(int(T), int(S), int(beam_size)))


# ----------------------------------------------------------------------------------------
# hash generator (same definition as ctc_oracle.c / the HIP generator kernel)
# ----------------------------------------------------------------------------------------
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_LABEL_SALT = 0x4C4142454C53


def _mix(seed, idx):
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * _GOLD + idx.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def hash_logprobs(T, V, seed=0):
    """lp[t,c] = -8 * u24(mix(seed, t*V+c)), float32 — exact on every platform."""
    h = _mix(seed, np.arange(T * V, dtype=np.uint64))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.float32(-8.0) * u).reshape(T, V)


def hash_labels(S, V, seed=0):
    h = _mix(np.uint64(seed) ^ np.uint64(_LABEL_SALT), np.arange(S, dtype=np.uint64))
    return (np.uint64(1) + h % np.uint64(V - 1)).astype(np.int32)


# ----------------------------------------------------------------------------------------
# per-frame NumPy port (the "NumPy CPU path" baseline)
# ----------------------------------------------------------------------------------------
def ctc_best_path_numpy(log_probs, labels, beam_size=1000, max_move=4, frame_limit=None):
    """Per-frame NumPy port of kokoro_align/align.py:43-109.

    Per frame it does what the reference does: for each move j scatter the candidates of
    the compacted live list into a (max_move, width) table (align.py:70-81), take the first
    argmax over moves and pick score/back-index with np.choose (align.py:83-85), compact the
    live set (align.py:87-91).  The back-indices of all frames are kept and one full
    backtrace is run at the end (equal to the reference's periodic flushes, align.py:95-102).

    ``frame_limit`` stops the forward pass after that many frames (used only to time a
    bounded sample in bench.py; the returned path is then None).
    """
    ext = np.zeros(2 * labels.shape[0] + 1, dtype=np.int32)
    ext[1::2] = labels
    L = ext.shape[0]
    T = log_probs.shape[0]
    half = beam_size // 2

    live_pos = np.zeros(1, dtype=np.int64)
    live_score = np.zeros(1, dtype=np.float32)
    trail = []
    n_frames = T if frame_limit is None else min(T, frame_limit)
    for t in range(n_frames):
        lo = max(0, L * t // T - half)
        hi = min(lo + beam_size, L)
        width = hi - lo
        back = np.full((max_move, width), -1, dtype=np.int32)
        cand = np.full((max_move, width), -np.inf, dtype=np.float32)
        row = log_probs[t]
        blank_cols = ext[lo:hi] == 0
        for j in range(max_move):
            tgt = live_pos + j
            sel, = np.nonzero((tgt >= lo) & (tgt < hi))
            dst = tgt[sel]
            back[j, dst - lo] = sel
            cand[j, dst - lo] = live_score[sel] + row[ext[dst]]
            if j > 0 and j % 2 == 0:
                cand[j, blank_cols] = -np.inf
        move = np.argmax(cand, axis=0)
        back = np.choose(move, back)
        cand = np.choose(move, cand)
        keep, = np.nonzero(back >= 0)
        live_score = cand[keep].copy()
        trail.append((keep + lo, back[keep].copy()))
        live_pos = keep + lo
    if frame_limit is not None and frame_limit < T:
        return None
    last_pos = trail[-1][0]
    cur = int(np.argmax(last_pos))  # ValueError on empty, as align.py:101
    path = np.empty(T, dtype=np.int32)
    for t in range(T - 1, -1, -1):
        pos, back = trail[t]
        path[t] = pos[cur]
        cur = back[cur]
    best_labels = ext[path]
    best_scores = log_probs[np.arange(T), best_labels]
    return path, best_labels, best_scores
