"""The cost model behind KA_MODE_AUTO / KA_BACKTRACE_AUTO (ka_engine.hip: auto_split_forward, auto_split_backtrace) through its
host-only probe: which lattices of a launch run tiled / are walked back chunk-parallel must follow the launch's LENGTHS, not
its lattice count ("300 long chapters and 300 short ones get the same form").  The expected regimes are the
measured ones of profiles/r03_sweep_auto_skewed.jsonl."""
import ctypes

import numpy as np
import pytest

from kokoro_align_amd import _lib
from kokoro_align_amd import workloads as W


def _split(T, alive=5, n_simd=1024):
    lib = _lib.load_library()
    arr = (ctypes.c_int64 * len(T))(*[int(t) for t in T])
    k, m = ctypes.c_int32(-7), ctypes.c_int32(-7)
    rc = lib.ka_debug_auto_split(ctypes.cast(arr, ctypes.POINTER(ctypes.c_int64)), len(T), alive, n_simd, ctypes.byref(k), ctypes.byref(m))
    assert rc == 0
    return k.value, m.value


def _uniform(n, lo, hi, seed):
    return np.random.default_rng(seed).integers(lo, hi, n).tolist()


def test_a_full_batch_runs_one_wavefront_per_lattice_serial_backtrace():
    assert _split([50000] * 8192) == (0, 0)          # BASELINE configs[1]: the throughput form
    assert _split([50000] * 2048) == (0, 0)          # one of the bench's four launches in flight


def test_a_lone_lattice_and_a_book_run_tiled_chunk_parallel():
    assert _split([50000]) == (1, 1)
    for book in (W.kokoro_book()[1], W.meian_book()[1]):
        T = [t for t, _ in book]
        assert _split(T) == (len(T), len(T))


def test_the_corpus_mixes_forms_by_length():
    T = [t for _, sh in W.corpus() for t, _ in sh]
    k, m = _split(T)
    n = len(T)
    assert n // 2 <= k <= n            # most chapters tiled (all of them is within 5 % of the best split)
    assert n // 4 <= m <= 3 * n // 4   # the longest half walked back chunk-parallel, the rest serially beside it


def test_same_count_different_lengths_different_answer():
    long_, short = _uniform(300, 80000, 160000, 1), _uniform(300, 20000, 30000, 2)
    assert _split(long_)[0] == 300 and _split(short)[0] == 300
    assert _split(long_)[1] < 300                       # 36 M frames of chunk maps are not worth it for the shortest
    k, m = _split(_uniform(40, 100000, 160000, 3) + _uniform(1500, 20000, 40000, 4))
    assert 20 <= k <= 400 and 20 <= m <= 400           # a few long chapters among many short ones: only those are tiled
    assert _split(_uniform(2000, 20000, 100000, 5))[0] == 0   # tiles and 2000 one-wavefront lattices stretch each other


def test_bad_arguments():
    lib = _lib.load_library()
    k, m = ctypes.c_int32(), ctypes.c_int32()
    assert lib.ka_debug_auto_split(None, 3, 5, 1024, ctypes.byref(k), ctypes.byref(m)) == -2
    assert _split([]) == (0, 0)
