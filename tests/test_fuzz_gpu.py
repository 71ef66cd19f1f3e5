"""A short run of tools/fuzz_modes.py inside the GPU suite: random shapes (band edges, chunk/block boundaries, label 0,
ties, unbanded, empty beams) through every kernel form and backtrace against the C oracle."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fuzz_every_form_against_the_oracle():
    argv = sys.argv
    sys.argv = ["fuzz_modes.py", "20", "3"]       # seconds, seed
    try:
        with pytest.raises(SystemExit) as exc:
            runpy.run_path(os.path.join(ROOT, "tools", "fuzz_modes.py"), run_name="__main__")
        assert exc.value.code == 0
    finally:
        sys.argv = argv
