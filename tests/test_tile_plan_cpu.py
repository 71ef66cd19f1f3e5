"""Host logic of the tiled form (no GPU): the tile plan against the band of align.py:64-65 evaluated frame by frame.

A tile that starts a frame late reads a halo slot nobody wrote; one that ends a frame early drops live cells.  The
closed forms in plan_tiles (ka_engine.hip) are checked here by brute force over shapes that include the BASELINE
configs, bands narrower than a tile, the unbanded case (beam >= 2L), L/T above 1 and the right-edge truncation.
"""
import ctypes

import numpy as np
import pytest

WIDTHS = [256, 128]       # positions per tile: ka_tiled2.hpp, ka_tiled_stream.hpp


def _plan(T, S, V, beam, max_move=4, cap=4096, width=256):
    from kokoro_align_amd import _lib
    L = _lib.load_library()
    t_in = np.zeros(cap, np.int32)
    t_end = np.zeros(cap, np.int32)
    pitch = ctypes.c_int64(0)
    if width == 256:
        n = L.ka_debug_plan_tiles(T, S, V, beam, max_move, t_in.ctypes.data, t_end.ctypes.data, cap, ctypes.byref(pitch))
    else:
        n = L.ka_debug_plan_tiles_width(T, S, V, beam, max_move, width, t_in.ctypes.data, t_end.ctypes.data, cap, ctypes.byref(pitch))
    return n, t_in[:max(n, 0)], t_end[:max(n, 0)], pitch.value


def _band(T, S, beam):
    """lo(t), hi(t) for every frame, exactly as the reference (Python integers)."""
    L = 2 * S + 1
    t = np.arange(T, dtype=object)
    lo = np.array([max(0, (L * int(i)) // T - beam // 2) for i in t], dtype=np.int64)
    hi = np.minimum(lo + beam, L)
    return L, lo, hi


SHAPES = [
    (50000, 5000, 64, 1000),      # BASELINE configs[1]
    (81140, 2000, 39, 1000),      # configs[0]
    (50000, 5000, 64, 30000),     # unbanded: beam >= 2L
    (8000, 10000, 64, 40007),     # unbanded, L/T = 2.5
    (3000, 4000, 39, 1000),       # L/T = 2.67: the band runs up fast, right-edge truncation
    (20000, 300, 39, 1000),       # L = 601 < beam: one band, three tiles for all frames
    (12000, 2500, 20, 100),       # band narrower than a tile
    (7001, 1777, 64, 999),        # odd sizes
    (5000, 5000, 64, 1000),       # L/T = 2
    (257, 700, 39, 1000),         # short
]


@pytest.mark.parametrize("TILE", WIDTHS)
@pytest.mark.parametrize("T,S,V,beam", SHAPES)
def test_tile_plan_matches_the_band_frame_by_frame(T, S, V, beam, TILE):
    n, t_in, t_end, pitch = _plan(T, S, V, beam, width=TILE)
    L, lo, hi = _band(T, S, beam)
    assert n > 0, "shape should be tileable"
    # which tiles does the band ever touch, and in which frames?  tile b is touched in frame t iff lo(t) < TILE (b+1) and hi(t) > TILE b
    n_tiles = (L + TILE - 1) // TILE
    want_in, want_end = [], []
    for b in range(n_tiles):
        alive = np.nonzero((hi > TILE * b) & (lo < TILE * (b + 1)))[0]
        if alive.size == 0:
            break
        # alive frames are one interval (both edges only move up)
        assert alive[-1] - alive[0] + 1 == alive.size
        want_in.append(int(alive[0]))
        want_end.append(int(alive[-1]) + 1)
    assert n == len(want_in)
    assert t_in.tolist() == want_in
    assert t_end.tolist() == want_end
    # every frame's band is covered by the tiles alive in it, and the last frame's tiles reach T
    assert t_end.max() == T
    # checkpoint row: wide enough that the positions alive at once never alias
    span = max(int(hi[t]) - (int(lo[t]) // TILE) * TILE for t in range(0, T, max(1, T // 500)))
    assert pitch // 4 >= min(span, n_tiles * TILE)
    assert pitch // 4 % 256 == 0      # (whatever the tile width: the readers of the checkpoints work in windows of 256 positions)


def test_shapes_outside_the_tiled_form_are_reported():
    from kokoro_align_amd import _lib
    assert _lib.load_library().ka_debug_plan_tiles_width(1000, 100, 39, 1000, 4, 64, None, None, 0, None) < 0   # widths are 128 and 256
    assert _plan(1000, 100, 65, 1000)[0] == 0          # V > 64
    assert _plan(1000, 100, 39, 1000, max_move=5)[0] == 0
    assert _plan(0, 100, 39, 1000)[0] < 0              # bad arguments: a status, not a crash


@pytest.mark.parametrize("TILE", WIDTHS)
def test_random_shapes(TILE):
    rng = np.random.default_rng(7)
    for _ in range(40):
        T = int(rng.integers(40, 6000))
        S = int(rng.integers(1, 4000))
        beam = int(rng.choice([16, 100, 500, 1000, 1009, 1010, 3000, 2 * (2 * S + 1) + 3]))
        if (2 * S + 1) / T > 200:
            continue
        n, t_in, t_end, _ = _plan(T, S, 39, beam, width=TILE)
        L, lo, hi = _band(T, S, beam)
        if n == 0:
            continue   # (the band jumps over a whole tile in one frame: not planned)
        for b in range(n):
            alive = np.nonzero((hi > TILE * b) & (lo < TILE * (b + 1)))[0]
            assert alive.size and int(alive[0]) == t_in[b] and int(alive[-1]) + 1 == t_end[b], (T, S, beam, b)
