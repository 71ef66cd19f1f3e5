"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "kokoro_align_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ka_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("ka_ctc_best_path_f32", "ka_ctc_best_path_batch_f32", "ka_ctc_best_path_batch_enqueue_f32",
                 "ka_batch_finish", "ka_engine_create", "ka_engine_destroy", "ka_last_error", "ka_log_softmax_f32"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import kokoro_align_amd as ka
    so = ka.build_library()
    lib = ctypes.CDLL(so)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.ka_version() >= 100


def test_binding_table_matches_header():
    from kokoro_align_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()
    L = _lib.load_library()
    for n in _lib.EXPORTS:
        assert getattr(L, n).argtypes is not None, n


def test_no_gpu_is_a_loud_error():
    """Without a device the engine cannot be created: an exception, never a silent fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np
    import kokoro_align_amd as ka
    with pytest.raises((ka.KAError, ValueError)):
        ka.ctc_best_path(np.zeros((4, 3), np.float32), np.array([1], np.int32), verbose=False)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under kokoro-align_amd/ may reference it."""
    pkg = os.path.join(ROOT, "kokoro-align_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libka_oracle" not in src and "ctc_oracle" not in src, f
