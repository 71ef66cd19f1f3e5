"""MI355X-native CTC forced-alignment hot path — drop-in for kokoro_align.align.

Public surface mirrors the reference (kokoro_align/align.py):
    ctc_best_path(log_probs, labels, beam_size=1000, max_move=4)   align.py:43
    best_path(input_file, voca_file, output_file)                  align.py:112
    align(best_path_file, mfcc_file, voca_file, align_file, remove_wordsep)   align.py:127
    pandas_read_align(files)                                       align.py:172
plus batched / device-resident entry points (ctc_best_path_batch, ctc_best_path_device).

The DP and backtrace run in the HIP C-ABI library (include/kokoro_align_amd.h); there is no
CPU fallback — importing works without a GPU, computing does not.
"""
from . import encoder, transcript  # noqa: F401
from .align import (  # noqa: F401
    align,
    best_path,
    ctc_best_path,
    ctc_best_path_batch,
    ctc_best_path_device,
    log_softmax_device,
    pandas_read_align,
)
from ._lib import KAError, build_library, library_path, load_library  # noqa: F401

__version__ = "0.1.0"
