"""ctypes binding of include/kokoro_align_amd.h (the C-ABI drop-in boundary).

The shared library is built in-tree by ``build_library()`` (kokoro-align_amd/csrc/Makefile,
hipcc --offload-arch=gfx950) and loaded from the package directory.  A missing library is an
error: there is no fallback implementation.
"""
import atexit
import ctypes
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("KA_LIBRARY") or os.path.join(_PKG, "libkokoro_align_amd.so")   # KA_LIBRARY: debug builds
_lib = None

KA_OK = 0
KA_ERR_EMPTY_BEAM = -1
KA_ERR_BAD_ARGS = -2
KA_ERR_HIP = -3
KA_ERR_NOMEM = -4
KA_ERR_BAD_LABEL = -5
KA_ERR_NAN = -6
KA_ERR_NONFINITE = -7
KA_MEM_HOST = 0
KA_MEM_DEVICE = 1

EXPORTS = [
    "ka_version", "ka_last_error", "ka_engine_create", "ka_engine_destroy", "ka_engine_reserve",
    "ka_workspace_bytes", "ka_ctc_best_path_f32", "ka_ctc_best_path_batch_f32",
    "ka_ctc_best_path_batch_enqueue_f32", "ka_batch_finish", "ka_engine_set_profiling",
    "ka_engine_last_kernel_ms", "ka_log_softmax_f32", "ka_hash_logprobs_f32", "ka_hash_labels_i32",
    "ka_hash_logprobs_batch_f32", "ka_hash_labels_batch_i32", "ka_engine_set_mode", "ka_lstm_step_f32",
    "ka_lstm_layer_f32", "ka_window_energy_f32", "ka_stft_frames_f32", "ka_power_f32", "ka_power_to_db_f32",
    "ka_debug_tile_stats", "ka_engine_set_backtrace", "ka_debug_chunk_entries", "ka_debug_plan_tiles",
    "ka_engine_set_verify", "ka_stream_create", "ka_stream_destroy",
    "ka_debug_set_split", "ka_engine_workspace_bytes", "ka_debug_set_tile_lds",
    "ka_debug_auto_split", "ka_debug_set_rc_gather", "ka_lstm_layer0_f32", "ka_debug_set_tile_width", "ka_debug_tile_width_choice", "ka_debug_plan_tiles_width",
]


class KAError(RuntimeError):
    """A C-ABI call failed (HIP error, bad arguments, out of memory)."""


def library_path():
    return _SO


def build_library(force=False):
    """Compile the HIP library for gfx950 (cross-compiles without a GPU)."""
    # every source of the library (a hand-kept list once missed the header of the default tiled kernel: a stale .so was tested)
    csrc = os.path.join(_PKG, "csrc")
    srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".hpp", ".h", ".cpp")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "kokoro_align_amd.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc")] + (["-B"] if force else []))
    return _SO


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO) and not os.environ.get("KA_LIBRARY") and shutil.which("hipcc", path=os.environ.get("PATH", "") + ":/opt/rocm/bin"):
        build_library()          # same HIP sources, built in-tree; never a different code path
    if not os.path.exists(_SO):
        raise KAError(
            f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C kokoro-align_amd/csrc` (there is no CPU fallback)")
    L = ctypes.CDLL(_SO)
    i32, i64, u64, vp, sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t
    pp = ctypes.POINTER(vp)
    pi64 = ctypes.POINTER(i64)
    L.ka_version.restype = i32
    L.ka_version.argtypes = []
    L.ka_last_error.restype = ctypes.c_char_p
    L.ka_last_error.argtypes = []
    L.ka_engine_create.restype = ctypes.c_int
    L.ka_engine_create.argtypes = [i32, pp]
    L.ka_engine_destroy.restype = None
    L.ka_engine_destroy.argtypes = [vp]
    L.ka_engine_reserve.restype = ctypes.c_int
    L.ka_engine_reserve.argtypes = [vp, sz]
    L.ka_workspace_bytes.restype = sz
    L.ka_workspace_bytes.argtypes = [i32, pi64, pi64, i32, i32, i32]
    L.ka_engine_workspace_bytes.restype = sz
    L.ka_engine_workspace_bytes.argtypes = [vp, i32, pi64, pi64, i32, i32, i32, i32]
    L.ka_ctc_best_path_f32.restype = ctypes.c_int
    L.ka_ctc_best_path_f32.argtypes = [vp, vp, i64, i32, i64, vp, i64, i32, i32, vp, vp, vp, vp, i32, vp]
    batch_common = [vp, i32, pp, pi64, i32, pi64, pp, pi64, i32, i32, pp, pp, pp]
    L.ka_ctc_best_path_batch_f32.restype = ctypes.c_int
    L.ka_ctc_best_path_batch_f32.argtypes = batch_common + [vp, vp, i32, vp]
    L.ka_ctc_best_path_batch_enqueue_f32.restype = ctypes.c_int
    L.ka_ctc_best_path_batch_enqueue_f32.argtypes = batch_common + [vp]
    L.ka_batch_finish.restype = ctypes.c_int
    L.ka_batch_finish.argtypes = [vp, vp, vp]
    L.ka_engine_set_mode.restype = ctypes.c_int
    L.ka_engine_set_mode.argtypes = [vp, i32]
    L.ka_engine_set_backtrace.restype = ctypes.c_int
    L.ka_engine_set_backtrace.argtypes = [vp, i32]
    L.ka_stream_create.restype = ctypes.c_int
    L.ka_stream_create.argtypes = [i32, pp]
    L.ka_stream_destroy.restype = ctypes.c_int
    L.ka_stream_destroy.argtypes = [i32, vp]
    L.ka_debug_auto_split.restype = ctypes.c_int
    L.ka_debug_auto_split.argtypes = [pi64, i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    L.ka_debug_set_rc_gather.restype = ctypes.c_int
    L.ka_debug_set_rc_gather.argtypes = [vp, i32]
    L.ka_debug_set_tile_lds.restype = ctypes.c_int
    L.ka_debug_set_tile_lds.argtypes = [vp, i32]
    L.ka_debug_tile_width_choice.restype = ctypes.c_int
    L.ka_debug_tile_width_choice.argtypes = [vp, vp, i32, i32, i32, i32, i32]
    L.ka_debug_set_tile_width.restype = ctypes.c_int
    L.ka_debug_set_tile_width.argtypes = [vp, i32]
    L.ka_debug_set_split.restype = ctypes.c_int
    L.ka_debug_set_split.argtypes = [vp, i32, i32]
    L.ka_engine_set_verify.restype = ctypes.c_int
    L.ka_engine_set_verify.argtypes = [vp, i32]
    L.ka_engine_set_profiling.restype = ctypes.c_int
    L.ka_engine_set_profiling.argtypes = [vp, i32]
    L.ka_engine_last_kernel_ms.restype = ctypes.c_int
    L.ka_engine_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.ka_debug_chunk_entries.restype = ctypes.c_int
    L.ka_debug_chunk_entries.argtypes = [vp, vp, i32, vp, i64]
    L.ka_debug_plan_tiles.restype = ctypes.c_int
    L.ka_debug_plan_tiles.argtypes = [i64, i64, i32, i32, i32, vp, vp, i32, vp]
    L.ka_debug_plan_tiles_width.restype = ctypes.c_int
    L.ka_debug_plan_tiles_width.argtypes = [i64, i64, i32, i32, i32, i32, vp, vp, i32, vp]
    L.ka_debug_tile_stats.restype = ctypes.c_int
    L.ka_debug_tile_stats.argtypes = [vp, vp, i32]
    L.ka_log_softmax_f32.restype = ctypes.c_int
    L.ka_log_softmax_f32.argtypes = [vp, vp, i64, i32, i64, i64, vp]
    L.ka_lstm_layer_f32.restype = ctypes.c_int
    L.ka_lstm_layer_f32.argtypes = [vp, i64, vp, vp, i64, vp, vp, i32, i32, vp]
    L.ka_lstm_layer0_f32.restype = ctypes.c_int
    L.ka_lstm_layer0_f32.argtypes = [vp, i64, i32, vp, vp, vp, vp, i64, vp, vp, i32, i32, vp]
    L.ka_lstm_step_f32.restype = ctypes.c_int
    L.ka_lstm_step_f32.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, i32, i32, vp]
    L.ka_window_energy_f32.restype = ctypes.c_int
    L.ka_window_energy_f32.argtypes = [vp, i64, i32, vp, vp]
    L.ka_stft_frames_f32.restype = ctypes.c_int
    L.ka_stft_frames_f32.argtypes = [vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, i64, vp]
    L.ka_power_f32.restype = ctypes.c_int
    L.ka_power_f32.argtypes = [vp, i64, vp, i64, i64, i32, vp]
    L.ka_power_to_db_f32.restype = ctypes.c_int
    L.ka_power_to_db_f32.argtypes = [vp, i64, i32, vp, i32, i64, ctypes.c_float, vp, vp]
    L.ka_hash_logprobs_f32.restype = ctypes.c_int
    L.ka_hash_logprobs_f32.argtypes = [vp, i64, i32, i64, u64, vp]
    L.ka_hash_labels_i32.restype = ctypes.c_int
    L.ka_hash_labels_i32.argtypes = [vp, i64, i32, u64, vp]
    L.ka_hash_logprobs_batch_f32.restype = ctypes.c_int
    L.ka_hash_logprobs_batch_f32.argtypes = [vp, i32, i64, i32, i64, i64, u64, vp]
    L.ka_hash_labels_batch_i32.restype = ctypes.c_int
    L.ka_hash_labels_batch_i32.argtypes = [vp, i32, i64, i32, i64, u64, vp]
    _lib = L
    return L


def last_error():
    return load_library().ka_last_error().decode("utf-8", "replace")


def check(rc, what):
    """Map a C status to the exception the reference would raise."""
    if rc == KA_OK:
        return
    msg = last_error()
    if rc == KA_ERR_EMPTY_BEAM:
        # reference: np.argmax of an empty array at kokoro_align/align.py:101
        raise ValueError("attempt to get argmax of an empty sequence")
    if rc == KA_ERR_BAD_LABEL:
        # reference: log_probs[i, labels[v]] at kokoro_align/align.py:77
        raise IndexError(f"{what}: label out of bounds for the vocabulary axis ({msg})")
    if rc == KA_ERR_NAN:
        raise ValueError(f"{what}: log_probs contain NaN ({msg})")
    if rc == KA_ERR_NONFINITE:
        raise ValueError(f"{what}: {msg}")
    if rc == KA_ERR_BAD_ARGS:
        raise ValueError(f"{what}: {msg}")
    if rc == KA_ERR_NOMEM:
        raise MemoryError(f"{what}: {msg}")
    raise KAError(f"{what}: rc={rc}: {msg}")


class Engine:
    """Owns a ka_engine (device workspace + staging) for one device."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        check(self.lib.ka_engine_create(int(device), ctypes.byref(h)), "ka_engine_create")
        self.handle = h
        self.device = int(device)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ka_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_mode(self, mode):
        """'auto' | 'wave' (1 wavefront per lattice, checkpointed) | 'wave_exact' (1 wavefront per lattice, every back-pointer
        stored) | 'tiled' (a pipeline of 128- or 256-position tiles, a workgroup of two or three wavefronts each: few lattices,
        any band width)"""
        code = {"auto": 0, "wave": 1, "wave_exact": 3, "tiled": 4}[mode] if isinstance(mode, str) else int(mode)
        check(self.lib.ka_engine_set_mode(self.handle, code), "ka_engine_set_mode")

    def set_backtrace(self, how):
        """'auto' | 'serial' (chunk after chunk) | 'parallel' (every chunk at once; ka_parallel_bt.hpp)"""
        code = {"auto": 0, "serial": 1, "parallel": 2}[how] if isinstance(how, str) else int(how)
        check(self.lib.ka_engine_set_backtrace(self.handle, code), "ka_engine_set_backtrace")

    def set_split(self, n_tiled=-1, n_parallel=-1):
        """Calibration of the AUTO modes: the longest n_tiled lattices of a launch run tiled, the longest n_parallel are walked
        back chunk-parallel; -1 = the library's cost model."""
        check(self.lib.ka_debug_set_split(self.handle, int(n_tiled), int(n_parallel)), "ka_debug_set_split")

    def set_rc_gather(self, how):
        check(self.lib.ka_debug_set_rc_gather(self.handle, int(how)), "ka_debug_set_rc_gather")

    def set_tile_lds(self, nbytes):
        check(self.lib.ka_debug_set_tile_lds(self.handle, int(nbytes)), "ka_debug_set_tile_lds")

    def set_tile_width(self, positions):
        """Positions per tile of the tiled form: 256, 128 (two-wavefront tiles only) or 0 = the library's choice."""
        check(self.lib.ka_debug_set_tile_width(self.handle, int(positions)), "ka_debug_set_tile_width")

    def set_verify(self, flags):
        """Self-checks of the tiled form's hand-off (ka_engine_set_verify): 1 = sentinel-filled halos, every consumed packet
        checked; 2 = full drain before every publish; 4 = per-tile phase stamps for ka_debug_tile_stats; 0 = off."""
        check(self.lib.ka_engine_set_verify(self.handle, int(flags)), "ka_engine_set_verify")

    def set_profiling(self, on=True):
        check(self.lib.ka_engine_set_profiling(self.handle, int(bool(on))), "ka_engine_set_profiling")

    def last_kernel_ms(self):
        ms = (ctypes.c_float * 4)()
        check(self.lib.ka_engine_last_kernel_ms(self.handle, ms), "ka_engine_last_kernel_ms")
        return {"prep": ms[0], "forward": ms[1], "backtrace": ms[2], "gather": ms[3]}

    def reserve(self, nbytes):
        check(self.lib.ka_engine_reserve(self.handle, int(nbytes)), "ka_engine_reserve")


_engines = {}


def _close_engines():
    # release device memory while the HIP runtime is still alive (not from __del__ at interpreter teardown)
    for e in list(_engines.values()):
        e.close()
    _engines.clear()


atexit.register(_close_engines)


def default_engine(device=0):
    e = _engines.get(device)
    if e is None:
        e = _engines[device] = Engine(device)
    return e
