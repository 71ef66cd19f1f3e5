"""Audio front end on the device: silence splitting and MFCC extraction.

Mirror of kokoro_align/preprocess.py:38-131 (``get_silent_ranges``, ``get_split_points``, ``split_audio``) with the
same names, argument meaning and outputs (``*.split.txt``: one segment end, in samples, per line; ``*.mfcc.npz``:
IndexDataArray ``indices`` = cumulative frame counts, ``data`` = float32 [frames, 40]).

Where the work runs:
  * the per-window level (mean square of every 256 samples: the only pass over the waveform in get_split_points)
    is one HIP kernel, ``ka_window_energy_f32``, summing in NumPy's float32 order so that the split points are the
    reference's bit for bit; the threshold / fill / merge logic that follows works on one value per 11.6 ms of audio
    and stays on the host, written with the same NumPy calls as the reference;
  * the MFCCs of ALL segments of a recording are computed together: HIP kernels for framing + window
    (``ka_stft_frames_f32``), |X|^2 (``ka_power_f32``) and the per-segment dB conversion with its top_db floor
    (``ka_power_to_db_f32``), library GEMMs (PyTorch-ROCm) for the three contractions: frames x DFT basis
    [512, 514], power x mel filters [257, 40], dB x DCT [40, 40].  Float32 like the reference's transform.
There is no CPU path: without the HIP library / a GPU these functions raise.

Audio decoding is not part of this package: ``split_audio`` takes 16-bit PCM ``.wav`` (standard library) or ``.npy``
float32 waveforms; the reference's mp3 input goes through ``torchaudio.load``, which is not a dependency here.
"""
import os

import numpy as np


# ----------------------------------------------------------------------------------------
# constants of the transform (float64 on the host, used as float32 on the device)
# ----------------------------------------------------------------------------------------
def _hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def _dft_basis(n_fft):
    """[n_fft, 2*(n_fft//2+1)] = (cos | -sin): frames @ basis = (real | imaginary) of the one-sided transform."""
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[None, :]
    n = np.arange(n_fft, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * ((n * k) % n_fft) / n_fft
    return np.concatenate([np.cos(ang), -np.sin(ang)], axis=1)


def _mel_filterbank(n_freqs, n_mels, sample_rate):
    """HTK mel scale, triangular, no area normalisation, 0 .. sample_rate/2 (torchaudio's defaults)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_max = 2595.0 * np.log10(1.0 + (sample_rate // 2) / 700.0)
    f_pts = 700.0 * (10.0 ** (np.linspace(0.0, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    return np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))


def _dct_ortho(n_mfcc, n_mels):
    """[n_mels, n_mfcc], DCT-II, norm="ortho"."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    d = np.cos(np.pi / n_mels * (n + 0.5) * k)
    d[0] *= 1.0 / np.sqrt(2.0)
    return (d * np.sqrt(2.0 / n_mels)).T


# ----------------------------------------------------------------------------------------
# silence splitting
# ----------------------------------------------------------------------------------------
def get_silent_ranges(voiced):
    """[[start, end)] of every run of silent windows that has voiced windows on BOTH sides (int array [n, 2]); a
    silence the recording begins or ends with is not a place to split (kokoro_align/preprocess.py:38-48)."""
    v = np.asarray(voiced, dtype=bool)
    change = np.flatnonzero(v[1:] != v[:-1]) + 1          # window indices at which voiced/silent flips
    falls = change[~v[change]]                             # ... into silence
    rises = change[v[change]]                              # ... back into voice
    if len(rises) and (len(falls) == 0 or rises[0] < falls[0]):
        rises = rises[1:]                                  # the recording began silent: that rise closes no inner silence
    n = min(len(falls), len(rises))                        # a silence that runs to the end has no rise
    return np.stack([falls[:n], rises[:n]], axis=1)


def window_energy(x, window_size, device=None):
    """float32 [len(x)//window_size] on the host: mean square of every window, computed on the device
    (``x``: NumPy array or torch tensor; a CUDA tensor is used in place)."""
    import torch
    from . import _lib
    lib = _lib.load_library()
    if not torch.cuda.is_available():
        raise _lib.KAError("kokoro_align_amd.preprocess needs a GPU")
    xd = torch.as_tensor(x, dtype=torch.float32)
    xd = xd.to(device if device is not None else "cuda").contiguous() if not xd.is_cuda else xd.contiguous()
    n = int(xd.numel()) // int(window_size)
    out = torch.empty((n,), dtype=torch.float32, device=xd.device)
    stream = torch.cuda.current_stream(xd.device).cuda_stream
    _lib.check(lib.ka_window_energy_f32(xd.data_ptr(), n, int(window_size), out.data_ptr(), stream), "ka_window_energy_f32")
    return out.cpu().numpy()


def get_split_points(x, minimum_silent_frames, minimum_split_distance, maximum_split_distance, window_size, eps=1e-12,
                     device=None):
    """Where to cut a recording, in windows of ``window_size`` samples (kokoro_align/preprocess.py:51-97).

    A window is voiced when its level, 10 ln(mean square + eps), lies above the midpoint between the loudest and the
    quietest window.  Silences shorter than ``minimum_silent_frames`` windows are filled in, every remaining inner
    silence gives one cut at its centre; while some piece is still ``maximum_split_distance`` windows or longer the
    minimum silence is halved and the search repeated (ValueError below 0.05 windows).  Pieces not longer than
    ``minimum_split_distance`` are then merged into a neighbour."""
    level = window_energy(x, window_size, device=device)
    n_windows = len(level)
    level = 10 * np.log(level + eps)
    threshold = (np.max(level) + np.min(level)) / 2
    while True:
        voiced = level > threshold
        # fill the short silences.  The ranges are disjoint, so +1 at every start, -1 at every end and a running sum
        # fill them all at once (the reference fills range by range; 8 hours of audio have ~10^4 of them and this
        # loop runs up to 9 times)
        ranges = get_silent_ranges(voiced)
        short = (ranges[:, 1] - ranges[:, 0]) < minimum_silent_frames
        if short.any():
            step = np.zeros(n_windows + 1, dtype=np.int32)
            step[ranges[short, 0]] = 1
            step[ranges[short, 1]] -= 1
            voiced |= np.cumsum(step[:-1]) > 0
        ranges = get_silent_ranges(voiced)
        silent_points = (ranges[:, 0] + ranges[:, 1]) // 2
        pieces = np.diff(np.concatenate(([0], silent_points, [n_windows])))
        if np.max(pieces) < maximum_split_distance:
            break
        minimum_silent_frames *= 0.5
        if minimum_silent_frames < 0.05:
            raise ValueError("Audio cannot be split into")
    num_frames = n_windows
    return _merge_short_pieces(silent_points, num_frames, minimum_split_distance)


def _merge_short_pieces(points, num_frames, minimum_split_distance):
    """preprocess.py:81-95: while the shortest piece (first one among equals) is not longer than the minimum, remove
    one of its end points - the only one it has at either end of the recording, otherwise the one towards its shorter
    neighbour (towards the later one when they are equal).  The reference rebuilds and scans the whole distance array
    for every removal (quadratic: 50 ms for 8 hours of audio); here the pieces sit in a doubly linked list with a heap
    of (length, position) - the same removals in the same order."""
    import heapq
    k = len(points)
    if k == 0:
        return points
    bounds = [0] + [int(p) for p in points] + [int(num_frames)]       # piece j = [bounds[j], bounds[j+1])
    npieces = k + 1
    left = list(range(-1, npieces - 1))                                # neighbouring live pieces
    right = list(range(1, npieces + 1))
    start = bounds[:-1]
    end = bounds[1:]
    alive = [True] * npieces
    heap = [(end[j] - start[j], start[j], j) for j in range(npieces)]
    heapq.heapify(heap)
    live = npieces
    while live > 1:
        d, st, j = heap[0]
        if not alive[j] or st != start[j] or d != end[j] - start[j]:
            heapq.heappop(heap)                                        # stale entry
            continue
        if d > minimum_split_distance:
            break
        lj, rj = left[j], right[j]
        if lj < 0:
            other = rj                                                 # first piece: drop its right end point
        elif rj >= npieces:
            other = lj                                                 # last piece: drop its left end point
        elif end[lj] - start[lj] < end[rj] - start[rj]:
            other = lj
        else:
            other = rj
        a, b = (other, j) if other == lj else (j, other)               # a is the left one of the two pieces that fuse
        heapq.heappop(heap)
        end[a] = end[b]
        alive[b] = False
        right[a] = right[b]
        if right[b] < npieces:
            left[right[b]] = a
        live -= 1
        heapq.heappush(heap, (end[a] - start[a], start[a], a))
    out = []
    j = 0
    while right[j] < npieces:
        out.append(end[j])
        j = right[j]
    return np.asarray(out, dtype=np.asarray(points).dtype)




# ----------------------------------------------------------------------------------------
# MFCC of all segments at once
# ----------------------------------------------------------------------------------------
_CONST = {}


def _constants(device, sample_rate, n_mfcc, n_mels, n_fft):
    import torch
    key = (str(device), sample_rate, n_mfcc, n_mels, n_fft)
    if key not in _CONST:
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)   # noqa: E731
        _CONST[key] = (f32(_hann_periodic(n_fft)), f32(_dft_basis(n_fft)), f32(_mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate)),
                       f32(_dct_ortho(n_mfcc, n_mels)))
    return _CONST[key]


def mfcc_segments(y, ends, sample_rate=22050, n_mfcc=40, n_mels=40, n_fft=512, top_db=80.0, device=None):
    """MFCCs of the segments ``y[0:ends[0]], y[ends[0]:ends[1]], ...`` as the reference's transform computes them one
    segment per call (preprocess.py:110-127, hop = n_fft//2): (float32 [frames, n_mfcc] on the device in segment
    order, int64 cumulative frame counts).  ``y``: float32 waveform (NumPy / torch, host or device)."""
    import torch
    from . import _lib
    lib = _lib.load_library()
    if not torch.cuda.is_available():
        raise _lib.KAError("kokoro_align_amd.preprocess needs a GPU")
    hop = n_fft // 2
    yd = torch.as_tensor(y, dtype=torch.float32)
    yd = yd.to(device if device is not None else "cuda").contiguous() if not yd.is_cuda else yd.contiguous()
    dev = yd.device
    ends = np.asarray(ends, dtype=np.int64).reshape(-1)
    starts = np.concatenate([[0], ends[:-1]])
    lens = ends - starts
    if len(ends) == 0:
        return torch.zeros((0, n_mfcc), dtype=torch.float32, device=dev), np.zeros((0,), np.int64)
    if int(ends[-1]) > int(yd.numel()) or (lens <= n_fft // 2).any():
        raise ValueError("mfcc_segments: every segment must be longer than n_fft/2 samples and lie inside the waveform")
    nfr = 1 + lens // hop
    foff = np.concatenate([[0], np.cumsum(nfr)])
    total = int(foff[-1])
    win, basis, fb, dct = _constants(dev, sample_rate, n_mfcc, n_mels, n_fft)
    stream = torch.cuda.current_stream(dev).cuda_stream
    i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(dev)   # noqa: E731
    d_start, d_len, d_foff = i64(starts), i64(lens), i64(foff)
    nf = n_fft // 2 + 1
    out = torch.empty((total, n_mfcc), dtype=torch.float32, device=dev)
    # in slabs of whole segments, so that the [frames, 514] transform of a long recording stays a few GB
    max_rows = 1 << 21
    s0 = 0
    while s0 < len(ends):
        s1 = s0 + 1
        while s1 < len(ends) and foff[s1 + 1] - foff[s0] <= max_rows and s1 - s0 < 65535:
            s1 += 1
        r0, r1 = int(foff[s0]), int(foff[s1])
        n = r1 - r0
        frames = torch.empty((n, n_fft), dtype=torch.float32, device=dev)
        rel = i64(foff[s0:s1 + 1] - foff[s0])
        _lib.check(lib.ka_stft_frames_f32(yd.data_ptr(), d_start[s0:].data_ptr(), d_len[s0:].data_ptr(), rel.data_ptr(), s1 - s0,
                                          int(nfr[s0:s1].max()), n_fft, hop, win.data_ptr(), frames.data_ptr(), frames.stride(0), stream),
                   "ka_stft_frames_f32")
        reim = frames @ basis                                   # [n, 2*nf]
        del frames
        power = torch.empty((n, nf), dtype=torch.float32, device=dev)
        _lib.check(lib.ka_power_f32(reim.data_ptr(), reim.stride(0), power.data_ptr(), power.stride(0), n, nf, stream), "ka_power_f32")
        del reim
        mel = power @ fb                                        # [n, n_mels]
        del power
        segmax = torch.full((s1 - s0,), float("-inf"), dtype=torch.float32, device=dev)
        _lib.check(lib.ka_power_to_db_f32(mel.data_ptr(), mel.stride(0), n_mels, rel.data_ptr(), s1 - s0, int(nfr[s0:s1].max()),
                                          float(top_db), segmax.data_ptr(), stream), "ka_power_to_db_f32")
        torch.mm(mel, dct, out=out[r0:r1])
        s0 = s1
    return out, foff[1:].copy()


def split_waveform(y, sample_rate=22050, n_mfcc=40, n_mels=40, n_fft=512, device=None):
    """The body of split_audio (preprocess.py:100-127) for a waveform already in memory: (segment ends in samples,
    MFCCs float32 [frames, n_mfcc] on the device, cumulative frame counts)."""
    window_size = n_fft // 2
    minimum_silent_frames = 0.25 * sample_rate / window_size
    minimum_split_distance = 3.0 * sample_rate / window_size
    maximum_split_distance = 15.0 * sample_rate / window_size
    import torch
    yd = torch.as_tensor(y, dtype=torch.float32)
    yd = yd.to(device if device is not None else "cuda").contiguous() if not yd.is_cuda else yd.contiguous()
    points = get_split_points(yd, minimum_silent_frames, minimum_split_distance, maximum_split_distance, window_size) * window_size
    ends = np.append(points, int(yd.numel())).astype(np.int64)
    mfcc, indices = mfcc_segments(yd, ends, sample_rate, n_mfcc, n_mels, n_fft)
    return ends, mfcc, indices


def load_waveform(audio_file, expected_sample_rate=22050):
    """float32 mono waveform of a 16-bit PCM .wav (checked against expected_sample_rate like preprocess.py:119) or .npy."""
    if audio_file.endswith(".npy"):
        y = np.load(audio_file)
        if y.ndim == 2:
            assert y.shape[0] == 1
            y = y[0]
        return np.ascontiguousarray(y, dtype=np.float32)
    if audio_file.endswith(".wav"):
        import wave
        with wave.open(audio_file, "rb") as w:
            assert w.getnchannels() == 1 and w.getsampwidth() == 2
            assert w.getframerate() == expected_sample_rate
            pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        return (pcm.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    raise ValueError("split_audio reads .wav (16-bit PCM) or .npy waveforms; decode other formats first")


def split_audio(audio_file, segment_file, audio_data_file, expected_sample_rate=22050, n_mfcc=40, n_mels=40, n_fft=512):
    """kokoro_align/preprocess.py:100-127: ``segment_file`` gets one segment end (samples) per line,
    ``audio_data_file`` the IndexDataArray of the segments' MFCCs."""
    from .pipeline import open_index_data_for_write
    y = load_waveform(audio_file, expected_sample_rate)
    ends, mfcc, indices = split_waveform(y, expected_sample_rate, n_mfcc, n_mels, n_fft)
    host = mfcc.cpu().numpy()
    try:
        with open(segment_file, "wt") as segf:
            with open_index_data_for_write(audio_data_file) as data:
                k = 0
                for end, stop in zip(ends.tolist(), indices.tolist()):
                    data.write(host[k:stop])
                    k = stop
                    segf.write(f"{end}\n")
    except BaseException:
        for f in (segment_file, audio_data_file):
            if os.path.exists(f):
                os.unlink(f)
        raise
