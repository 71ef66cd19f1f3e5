"""Drop-in for kokoro_align/align.py: same functions, keyword names, defaults, file formats.

    ctc_best_path(log_probs, labels, beam_size=1000, max_move=4)      <- align.py:43-109
    best_path(input_file, voca_file, output_file)                     <- align.py:112-124
    align(best_path_file, mfcc_file, voca_file, align_file, remove_wordsep)  <- align.py:127-169
    pandas_read_align(files)                                          <- align.py:172-190

The DP + backtrace (the reference's per-frame NumPy loop) run in the HIP library through the C
ABI of include/kokoro_align_amd.h.  NumPy arrays are handed over as host buffers; torch tensors
on a ROCm device are handed over by pointer and the results stay on the device.
"""
import ctypes
import os

import numpy as np

from . import _lib
from .encoder import decode_text, merge_repeated


# ------------------------------------------------------------------------------------------
# helpers
# ------------------------------------------------------------------------------------------
def _is_tensor(x):
    return hasattr(x, "data_ptr") and hasattr(x, "device")


def _stream_ptr(device_index):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device_index).cuda_stream)


def _ptr_array(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    return ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p)), arr


def _i64_array(vals):
    arr = (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])
    return ctypes.cast(arr, ctypes.POINTER(ctypes.c_int64)), arr


# ------------------------------------------------------------------------------------------
# ctc_best_path
# ------------------------------------------------------------------------------------------
def ctc_best_path(log_probs, labels, beam_size=1000, max_move=4, verbose=True):
    """CTC best path of ``labels`` through ``log_probs`` (reference: align.py:43-109).

    log_probs [T, V] float32, labels [S] integer ids (blanks are inserted here).  Returns
    (best_path int32 [T] in blank-expanded positions, best_labels int32 [T], best_scores
    float32 [T]).  NumPy in -> NumPy out; ROCm torch tensors in -> torch tensors out (same
    device).  Raises ValueError where the reference does (no live state in the last frame).

    Differences from the reference, by design: float64 ``log_probs`` are cast to float32 first;
    NaN / +inf log-probs are not supported (the reference's np.argmax treats NaN as a maximum).
    """
    if _is_tensor(log_probs):
        out = ctc_best_path_device([log_probs], [labels], beam_size, max_move, verbose=verbose)
        return out[0]
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    lab = np.ascontiguousarray(np.asarray(labels).reshape(-1), dtype=np.int32)
    if lp.ndim != 2:
        raise ValueError("log_probs must be [T, V]")
    T, V = lp.shape
    S = lab.shape[0]
    if verbose:  # the reference prints these two lines (align.py:53-54)
        print(f"Label length: {2 * S + 1}")
        print(f"Time length: {T}")
    if T == 0:
        raise IndexError("list index out of range")  # reference: beams[-1] on an empty list, align.py:101
    path = np.empty(T, np.int32)
    lout = np.empty(T, np.int32)
    sout = np.empty(T, np.float32)
    eng = _lib.default_engine(_current_device())
    rc = eng.lib.ka_ctc_best_path_f32(eng.handle, lp.ctypes.data, T, V, V, lab.ctypes.data, S, int(beam_size),
                                      int(max_move), path.ctypes.data, lout.ctypes.data, sout.ctypes.data,
                                      None, _lib.KA_MEM_HOST, None)
    _lib.check(rc, "ctc_best_path")
    return path, lout, sout


def _current_device():
    dev = os.environ.get("KA_DEVICE")
    if dev is not None:
        return int(dev)
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except ImportError:
        pass
    return 0


def ctc_best_path_batch(log_probs_list, labels_list, beam_size=1000, max_move=4, device=None,
                        return_status=False):
    """Independent lattices (one per audio file) in ONE launch; host NumPy buffers in and out.

    Returns a list of (best_path, best_labels, best_scores); with ``return_status`` also the
    per-lattice status list (0 ok, -1 empty beam = the reference's ValueError) and total scores,
    in which case failures do not raise.
    """
    n = len(log_probs_list)
    assert n == len(labels_list)
    if n == 0:
        return ([], [], []) if return_status else []
    lps = [np.ascontiguousarray(x, dtype=np.float32) for x in log_probs_list]
    labs = [np.ascontiguousarray(np.asarray(x).reshape(-1), dtype=np.int32) for x in labels_list]
    V = lps[0].shape[1]
    for x in lps:
        if x.ndim != 2 or x.shape[1] != V:
            raise ValueError("all log_probs must be [T_i, V] with one V")
        if x.shape[0] == 0:
            raise IndexError("list index out of range")
    Ts = [x.shape[0] for x in lps]
    Ss = [x.shape[0] for x in labs]
    paths = [np.empty(t, np.int32) for t in Ts]
    louts = [np.empty(t, np.int32) for t in Ts]
    souts = [np.empty(t, np.float32) for t in Ts]
    status = np.zeros(n, np.int32)
    total = np.zeros(n, np.float32)
    eng = _lib.default_engine(_current_device() if device is None else device)
    p_lp, _k1 = _ptr_array([x.ctypes.data for x in lps])
    p_lab, _k2 = _ptr_array([x.ctypes.data for x in labs])
    p_path, _k3 = _ptr_array([x.ctypes.data for x in paths])
    p_lout, _k4 = _ptr_array([x.ctypes.data for x in louts])
    p_sout, _k5 = _ptr_array([x.ctypes.data for x in souts])
    p_T, _k6 = _i64_array(Ts)
    p_S, _k7 = _i64_array(Ss)
    p_ld, _k8 = _i64_array([V] * n)
    rc = eng.lib.ka_ctc_best_path_batch_f32(eng.handle, n, p_lp, p_T, V, p_ld, p_lab, p_S, int(beam_size),
                                            int(max_move), p_path, p_lout, p_sout, total.ctypes.data,
                                            status.ctypes.data, _lib.KA_MEM_HOST, None)
    results = list(zip(paths, louts, souts))
    if return_status:
        if rc not in (_lib.KA_OK, _lib.KA_ERR_EMPTY_BEAM, _lib.KA_ERR_BAD_LABEL, _lib.KA_ERR_NAN, _lib.KA_ERR_NONFINITE):
            _lib.check(rc, "ctc_best_path_batch")
        return results, status.tolist(), total
    _lib.check(rc, "ctc_best_path_batch")
    return results


class DeviceBatch:
    """A batch of device-resident lattices with its pointer tables prebuilt, so that a
    launch is one C call.  Tensors are torch ROCm tensors (float32 log-probs [T_i, V] with
    unit column stride, int32 labels [S_i]); outputs are allocated here and reused."""

    def __init__(self, log_probs, labels, beam_size=1000, max_move=4, outputs=None):
        import torch
        assert len(log_probs) == len(labels) and len(log_probs) > 0
        dev = log_probs[0].device
        self.device_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self.n = len(log_probs)
        self.V = int(log_probs[0].shape[1])
        self.log_probs, self.labels = [], []
        for lp, lab in zip(log_probs, labels):
            if lp.dtype != torch.float32:
                lp = lp.float()
            if lp.dim() != 2 or lp.shape[1] != self.V or lp.shape[0] == 0:
                raise ValueError("log_probs must be non-empty [T_i, V] tensors with one V")
            if lp.stride(1) != 1:
                lp = lp.contiguous()
            lab = lab.reshape(-1)
            if lab.dtype != torch.int32 or not lab.is_contiguous() or lab.device != dev:
                lab = lab.to(device=dev, dtype=torch.int32).contiguous()
            self.log_probs.append(lp)
            self.labels.append(lab)
        self.T = [int(x.shape[0]) for x in self.log_probs]
        self.S = [int(x.shape[0]) for x in self.labels]
        if outputs is None:
            self.path = [torch.empty(t, dtype=torch.int32, device=dev) for t in self.T]
            self.best_labels = [torch.empty(t, dtype=torch.int32, device=dev) for t in self.T]
            self.best_scores = [torch.empty(t, dtype=torch.float32, device=dev) for t in self.T]
        else:
            self.path, self.best_labels, self.best_scores = outputs
        self.beam_size, self.max_move = int(beam_size), int(max_move)
        self.engine = _lib.default_engine(self.device_index)
        self._p_lp, self._k1 = _ptr_array([x.data_ptr() for x in self.log_probs])
        self._p_lab, self._k2 = _ptr_array([x.data_ptr() for x in self.labels])
        self._p_path, self._k3 = _ptr_array([x.data_ptr() for x in self.path])
        self._p_lout, self._k4 = _ptr_array([x.data_ptr() for x in self.best_labels])
        self._p_sout, self._k5 = _ptr_array([x.data_ptr() for x in self.best_scores])
        self._p_T, self._k6 = _i64_array(self.T)
        self._p_S, self._k7 = _i64_array(self.S)
        self._p_ld, self._k8 = _i64_array([x.stride(0) for x in self.log_probs])
        self.status = np.zeros(self.n, np.int32)
        self.total = np.zeros(self.n, np.float32)

    def workspace_bytes(self):
        """device workspace this batch's engine will carve for it, with the engine's current mode settings"""
        e = self.engine
        return int(e.lib.ka_engine_workspace_bytes(e.handle, self.n, self._p_T, self._p_S, self.V, self.beam_size,
                                                   self.max_move, _lib.KA_MEM_DEVICE))

    def enqueue(self):
        """Launch prep + forward DP + backtrace on torch's current stream; no host sync."""
        e = self.engine
        rc = e.lib.ka_ctc_best_path_batch_enqueue_f32(
            e.handle, self.n, self._p_lp, self._p_T, self.V, self._p_ld, self._p_lab, self._p_S,
            self.beam_size, self.max_move, self._p_path, self._p_lout, self._p_sout,
            _stream_ptr(self.device_index))
        _lib.check(rc, "ctc_best_path_batch_enqueue")

    def finish(self, raise_on_error=True):
        """Synchronise the stream and fetch per-lattice status / total scores."""
        e = self.engine
        rc = e.lib.ka_batch_finish(e.handle, self.total.ctypes.data, self.status.ctypes.data)
        if raise_on_error:
            _lib.check(rc, "ctc_best_path_batch")
        elif rc not in (_lib.KA_OK, _lib.KA_ERR_EMPTY_BEAM, _lib.KA_ERR_BAD_LABEL, _lib.KA_ERR_NAN, _lib.KA_ERR_NONFINITE):
            _lib.check(rc, "ctc_best_path_batch")
        return self.status

    def run(self, raise_on_error=True):
        self.enqueue()
        return self.finish(raise_on_error)

    def results(self):
        return list(zip(self.path, self.best_labels, self.best_scores))


def ctc_best_path_device(log_probs, labels, beam_size=1000, max_move=4, verbose=False):
    """Lists of ROCm torch tensors in, list of (best_path, best_labels, best_scores) tensors out.
    One launch for the whole list; nothing leaves the device except 16 B of status per lattice."""
    import torch
    dev = log_probs[0].device
    labels = [x if _is_tensor(x) else torch.as_tensor(np.asarray(x).reshape(-1).astype(np.int32)) for x in labels]
    for lp, lab in zip(log_probs, labels):
        if verbose:
            print(f"Label length: {2 * int(lab.numel()) + 1}")
            print(f"Time length: {int(lp.shape[0])}")
        if lp.shape[0] == 0:
            raise IndexError("list index out of range")
    with torch.cuda.device(dev):
        batch = DeviceBatch(log_probs, labels, beam_size, max_move)
        batch.run()
    return batch.results()


def log_softmax_device(logits, out=None):
    """Mean-subtracted log-softmax of align.py:116-117 on the device (HIP kernel), float32."""
    import torch
    x = logits if logits.dtype == torch.float32 else logits.float()
    if x.stride(1) != 1:
        x = x.contiguous()
    if out is None:
        out = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    lib = _lib.load_library()
    idx = x.device.index if x.device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(x.device):
        rc = lib.ka_log_softmax_f32(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.stride(0),
                                    out.stride(0), _stream_ptr(idx))
    _lib.check(rc, "log_softmax_device")
    return out


# ------------------------------------------------------------------------------------------
# file-level wrappers (same file names, npz keys, dtypes and text format as the reference)
# ------------------------------------------------------------------------------------------
def _host_log_softmax(logits):
    """align.py:116-117 verbatim in behaviour: float32 NumPy, mean-subtracted, NOT max-subtracted."""
    centred = logits - np.mean(logits, axis=-1, keepdims=True)
    return centred - np.log(np.sum(np.exp(centred), axis=-1, keepdims=True))


def best_path(input_file, voca_file, output_file, device_softmax=False):
    """``*.logits.npz`` + ``*.voca.txt`` -> ``*.best_path.npz`` (reference: align.py:112-124).

    Keys/dtypes of the output: best_path int32, best_labels int32, best_scores float32.
    ``device_softmax=True`` computes the log-softmax with the HIP kernel (1e-6 of the NumPy
    formula) and keeps the log-probs on the device; the default reproduces the reference's
    host NumPy arithmetic bit for bit and hands the DP a host buffer.
    """
    from .transcript import read_transcript
    with np.load(input_file) as f:
        logits = f['data']
    labels = read_transcript(voca_file)
    if device_softmax:
        import torch
        dev = torch.device("cuda", _current_device())
        lp = log_softmax_device(torch.from_numpy(np.ascontiguousarray(logits, np.float32)).to(dev))
        (p, l, s), = ctc_best_path_device([lp], [labels], verbose=True)
        p, l, s = p.cpu().numpy(), l.cpu().numpy(), s.cpu().numpy()
    else:
        p, l, s = ctc_best_path(_host_log_softmax(logits), labels)
    np.savez(output_file, best_path=p, best_labels=l, best_scores=s)


def align(best_path_file, mfcc_file, voca_file, align_file, remove_wordsep):
    """Best path -> one line per silence-delimited audio segment (reference: align.py:127-169).

    Line format: audio_end|text|voca|decoded|non_blanks|non_blanks_score|all_score, floats
    printed as Python repr of float(np.float32 sum).  A partially written file is removed
    on any error, like the reference.
    """
    from .transcript import VocaAligner
    with np.load(best_path_file) as f:
        path = f['best_path'] // 2          # expanded position -> phoneme index (align.py:135)
        best_labels = f['best_labels']
        best_scores = f['best_scores']
    with np.load(mfcc_file) as f:
        seg_ends = f['indices']
    aligner = VocaAligner(voca_file)
    n_phonemes = len(aligner)
    n_frames = len(path)
    try:
        with open(align_file, 'wt') as out:
            for i in range(len(seg_ends)):
                a = seg_ends[i - 1] if i > 0 else 0
                b = seg_ends[i]
                text_start = min(path[a], n_phonemes)
                text_end = min(path[b], n_phonemes) if b < n_frames else n_phonemes
                seg_labels = best_labels[a:b]
                seg_scores = best_scores[a:b]
                voiced = seg_labels != 0
                decoded = merge_repeated(decode_text(seg_labels))
                non_blanks = np.sum(voiced).item()
                non_blanks_score = np.sum(seg_scores[voiced]).item()
                all_score = np.sum(seg_scores).item()
                text, voca = aligner.get_token(text_start, text_end, remove_wordsep=remove_wordsep)
                out.write(f'{b}|{text}|{voca}|{decoded}|{non_blanks}|{non_blanks_score}|{all_score}\n')
    except BaseException:
        os.unlink(align_file)
        raise


def pandas_read_align(files):
    """Read ``*.align.txt`` files into one DataFrame (reference: align.py:172-190)."""
    import pandas as pd
    rows = []
    for file in files:
        prev_end = '0'
        with open(file) as f:
            for line in f:
                fields = line.rstrip().split('|')
                rows.append([prev_end] + fields)
                prev_end = fields[0]
    cols = ['audio_start', 'audio_end', 'text', 'voca', 'decoded', 'non_blanks', 'non_blanks_score', 'all_score']
    df = pd.DataFrame(rows, columns=cols)
    for c in ('audio_start', 'audio_end', 'non_blanks'):
        df[c] = df[c].astype(int)
    for c in ('non_blanks_score', 'all_score'):
        df[c] = df[c].astype(float)
    df['audio_len'] = df['audio_end'] - df['audio_start']
    return df
