"""The kernels release registers written by inline-asm loads with counted waits; a compiler-inserted copy of such
a register ahead of its wait reads it before the load has landed.  This test disassembles
the gfx950 code object the way `make asm` does and runs tools/lint_inflight.py over it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kokoro-align_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc", path=os.environ.get("PATH", "") + ":/opt/rocm/bin") is None,
                    reason="needs hipcc to produce the kernel assembly")
def test_no_copy_of_a_register_whose_load_is_in_flight():
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    subprocess.run(["make", "-C", CSRC, "asm"], check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lint_inflight
    paths = lint_inflight.default_paths()
    assert len(paths) >= 6, "one assembly file per device translation unit"
    report, total = lint_inflight.check_all(paths)
    assert len(report) >= 30, "kernels not found in the assembly"
    assert total == 0, [(n, f[:3]) for n, f in report if f]
