"""The oracle's C restatement under AddressSanitizer + UBSan (CPU only: GPU sanitizers are not
available on this pool).  Runs the g1 goldens through the instrumented build in a subprocess."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, sys, os
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import numpy as np
from golden_util import g1_cases
L = ctypes.CDLL(os.path.join(%(root)r, "oracle", "libka_oracle_asan.so"))
vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
L.kao_ctc_best_path_f32.restype = ctypes.c_int
L.kao_ctc_best_path_f32.argtypes = [vp, i64, i32, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp]
n = 0
for c in g1_cases():
    lp = np.ascontiguousarray(c["lp"], np.float32); lab = np.ascontiguousarray(c["labels"], np.int32)
    T, V = lp.shape
    p = np.empty(T, np.int32); l = np.empty(T, np.int32); s = np.empty(T, np.float32)
    tot = ctypes.c_float(0); end = ctypes.c_int64(0)
    rc = L.kao_ctc_best_path_f32(lp.ctypes.data, T, V, V, lab.ctypes.data, lab.shape[0], c["beam"], c["max_move"],
                                 p.ctypes.data, l.ctypes.data, s.ctypes.data, ctypes.addressof(tot), ctypes.addressof(end))
    assert rc == (-1 if c["status"] == 1 else 0), (c["idx"], rc)
    if rc == 0:
        assert np.array_equal(p, c["path"]), c["idx"]
    n += 1
print("asan-ok", n)
"""


def test_oracle_under_asan_ubsan():
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libka_oracle_asan.so"])
    except (subprocess.CalledProcessError, FileNotFoundError):
        pytest.skip("sanitizer build not available")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.exists(asan):
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "asan-ok 250" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
