"""Import shim: the package directory is named ``kokoro-align_amd`` (not an importable
identifier), so ``import kokoro_align_amd`` resolves to this file, which loads the package
from that directory and registers it under this module's name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kokoro-align_amd")
_spec = importlib.util.spec_from_file_location(
    "kokoro_align_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["kokoro_align_amd"] = _mod
_spec.loader.exec_module(_mod)
