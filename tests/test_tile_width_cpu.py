"""Which tile width a tiled launch gets (ka_engine.hip: narrow_tiles_pay) through its host-only probe: 128 positions while the
tiles alive at once fit the device 3.2 times over and the tiles that never die all have a workgroup slot, else 256.  The
expected regimes are the measured ones of profiles/r04_sweep_width.jsonl (round 4's tile kernel) and profiles/r03_bench_tiled.jsonl."""
import ctypes

import pytest

from kokoro_align_amd import _lib
from kokoro_align_amd import workloads as W


def _width(shapes, V=39, beam=1000, max_move=4, n_simd=1024):
    lib = _lib.load_library()
    T = (ctypes.c_int64 * len(shapes))(*[int(t) for t, _ in shapes])
    S = (ctypes.c_int64 * len(shapes))(*[int(s) for _, s in shapes])
    return lib.ka_debug_tile_width_choice(ctypes.cast(T, ctypes.c_void_p), ctypes.cast(S, ctypes.c_void_p), len(shapes), V, beam, max_move, n_simd)


def test_a_lone_lattice_and_the_books_get_128_positions():
    assert _width([(W.CFG2["T"], W.CFG2["S"])], V=64) == 128           # BASELINE configs[1] as a single lattice
    assert _width([(W.CFG1["T"], W.CFG1["S"])]) == 128                 # configs[0]
    assert _width(W.kokoro_book()[1]) == 128                           # configs[2]: 64 chapters, ~580 tiles alive
    assert _width(W.meian_book()[1]) == 128                            # configs[3]: 120 chapters
    assert _width([(W.CFG5["T"], W.CFG5["S"])], V=64) == 128           # configs[4] with the band of 1000


def test_launches_that_oversubscribe_the_device_keep_256_positions():
    corpus = [s for _, sh in W.corpus() for s in sh]
    assert _width(corpus[:200]) == 128        # ~1800 tiles alive on 768 slots: ahead (6.15 against 7.42 ms)
    assert _width(corpus[:250]) == 128        # ~2250: still ahead (7.38 against 7.91)
    assert _width(corpus[:320]) == 256        # ~2900: behind (9.53 against 9.17)
    assert _width(corpus) == 256              # all 462: 13.9 against 11.6
    # V = 64: 52 KB of LDS per tile, still three workgroups per CU
    assert _width([(50000, 5000)] * 250, V=64) == 128
    assert _width([(50000, 5000)] * 320, V=64) == 256


def test_tiles_that_never_die_must_all_fit():
    # the whole lattice (beam_size >= 2L): every tile is alive from the first frame to the last
    assert _width([(50000, 5000)], V=64, beam=30000) == 128            # 79 tiles of 128 positions
    assert _width([(500000, 50000)], V=64, beam=200002) == 256         # 782 on 768 slots (89 ms when forced onto 512, 56 with 256 positions)
    assert _width([(500000, 50000)], V=64, beam=200002, n_simd=4096) == 128   # (a device four times the size would hold them)


def test_shapes_outside_the_tiled_form_and_bad_arguments():
    assert _width([(1000, 100)], V=65) == 0                            # V > 64: not tiled at all
    assert _width([(400, 30000)], V=39) == 128                         # L/T = 150 with the band of 1000: every tile still lives 7 frames
    assert _width([(300, 30000)], V=39, beam=16) != 128                # L/T = 200, band 16: a 128-position tile is jumped over in one frame
    assert _width([(0, 10)]) < 0
    assert _width([(1000, 100)], n_simd=0) < 0
    assert _width([]) == 256                                           # nothing tiled: nothing to narrow
