"""oracle/frontend_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's audio front end, kokoro_align/preprocess.py:38-131:

* ``split_points``  — get_silent_ranges + get_split_points (preprocess.py:38-97), NumPy like the reference.
                      PINNED: bit-equal to tests/golden/g6_split.json, which tests/golden/make_golden.py produced
                      by calling the reference's own functions in this container.
* ``mfcc``          — what ``torchaudio.transforms.MFCC(sample_rate=22050, n_mfcc=40, melkwargs={'n_fft': 512,
                      'n_mels': 40, 'hop_length': 256})`` computes (the call at preprocess.py:110-113), restated in
                      float64 from torchaudio's published algorithm.  PARITY UNPINNED: torchaudio is a dependency
                      of the reference (setup.py:17, no version pinned) that is absent from /root/reference and
                      from this image, and the reference holds no MFCC fixture, so no output of the real transform
                      exists to check against.  The algorithm restated (torchaudio 0.8-2.x, unchanged over those):
                        Spectrogram: center=True, pad_mode="reflect", periodic Hann window of n_fft samples,
                                     onesided, power 2.0, not normalised;
                        MelScale:    n_mels triangular filters, mel_scale="htk", norm=None, f_min 0, f_max sr/2,
                                     all_freqs = linspace(0, sr//2, n_fft//2+1);
                        AmplitudeToDB("power", top_db=80): 10*log10(clamp(x, 1e-10)), then clamped from below at
                                     (max over the whole call) - 80;
                        DCT-II, norm="ortho": dct[n, k] = cos(pi/n_mels*(n+0.5)*k) * sqrt(2/n_mels), column 0
                                     additionally * 1/sqrt(2);   mfcc = (mel_db^T @ dct)^T.
* ``hash_waveform`` — exact-everywhere synthetic audio (bursts of hash noise with silences).

Only tests/, __graft_entry__.smoke() and tools/ benchmarks' CPU legs may import this module.
"""
import numpy as np

from .oracle import _mix


def hash_waveform(n, seed, pieces):
    """float32 [n]: x[i] = amp(i) * (u24(mix(seed, i)) - 0.5), amp piecewise constant: ``pieces`` = list of
    (start, end, amp) in samples (later pieces override earlier ones); elsewhere amp = 0."""
    h = _mix(seed, np.arange(n, dtype=np.uint64))
    u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)
    amp = np.zeros(n, dtype=np.float32)
    for a, b, v in pieces:
        amp[int(a):int(b)] = np.float32(v)
    return (u * amp).astype(np.float32)


# ----------------------------------------------------------------------------------------
# silence splitting (preprocess.py:38-97)
# ----------------------------------------------------------------------------------------
def silent_ranges(voiced):
    """preprocess.py:38-48: [k, 2] (first silent frame, first voiced frame after it), leading / trailing
    silence dropped."""
    s2v = np.where((~voiced[:-1]) & voiced[1:])[0] + 1
    v2s = np.where((voiced[:-1]) & ~voiced[1:])[0] + 1
    if not voiced[0]:
        s2v = s2v[1:]
    if not voiced[-1]:
        v2s = v2s[:-1]
    return np.stack([v2s, s2v]).T


def window_energy_db(x, window_size, eps=1e-12):
    """preprocess.py:53-55: 10*ln(mean square per window + eps), float32 like the reference's NumPy."""
    num_frames = len(x) // window_size
    mX = np.mean(x[:window_size * num_frames].reshape((-1, window_size)) ** 2, axis=1)
    return 10 * np.log(mX + eps)


def split_points_from_db(mX, minimum_silent_frames, minimum_split_distance, maximum_split_distance):
    """preprocess.py:57-97 given the per-window level: split points in windows."""
    num_frames = len(mX)
    thr = (np.max(mX) + np.min(mX)) / 2
    while True:
        voiced = mX > thr
        for s, e in silent_ranges(voiced):
            if e - s < minimum_silent_frames:      # fill short silences
                voiced[s:e] = True
        rng = silent_ranges(voiced)
        points = (rng[:, 0] + rng[:, 1]) // 2      # split in the centre of a silence
        dist = np.append(points, num_frames) - np.insert(points, 0, 0)
        if np.max(dist) < maximum_split_distance:
            break
        minimum_silent_frames *= 0.5
        if minimum_silent_frames < 0.05:
            raise ValueError("Audio cannot be split into")
    while len(points):                             # merge short pieces
        dist = np.append(points, num_frames) - np.insert(points, 0, 0)
        i = np.argmin(dist)
        if dist[i] > minimum_split_distance:
            break
        if i == 0:
            points = np.delete(points, i)
        elif i == len(points):
            points = np.delete(points, len(points) - 1)
        elif dist[i - 1] < dist[i + 1]:
            points = np.delete(points, i - 1)
        else:
            points = np.delete(points, i)
    return points


def split_points(x, minimum_silent_frames, minimum_split_distance, maximum_split_distance, window_size, eps=1e-12):
    """get_split_points (preprocess.py:51-97)."""
    return split_points_from_db(window_energy_db(x, window_size, eps), minimum_silent_frames,
                                minimum_split_distance, maximum_split_distance)


def split_parameters(sample_rate=22050, n_fft=512):
    """The constants split_audio derives (preprocess.py:103-108): window, min silence, min / max piece, in windows."""
    window = n_fft // 2
    return dict(window_size=window, minimum_silent_frames=0.25 * sample_rate / window,
                minimum_split_distance=3.0 * sample_rate / window, maximum_split_distance=15.0 * sample_rate / window)


# ----------------------------------------------------------------------------------------
# MFCC (the torchaudio transform called at preprocess.py:110-113) - see the module docstring
# ----------------------------------------------------------------------------------------
def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def mel_filterbank(n_freqs=257, n_mels=40, sample_rate=22050, f_min=0.0, f_max=None):
    """[n_freqs, n_mels] triangular HTK filters, no area normalisation."""
    f_max = float(sample_rate // 2) if f_max is None else f_max
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * np.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * np.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10.0 ** (np.linspace(m_min, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def dct_matrix(n_mfcc=40, n_mels=40):
    """[n_mels, n_mfcc], DCT-II with norm="ortho"."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(np.pi / n_mels * (n + 0.5) * k)
    dct[0] *= 1.0 / np.sqrt(2.0)
    dct *= np.sqrt(2.0 / n_mels)
    return dct.T


def mfcc(y, sample_rate=22050, n_mfcc=40, n_mels=40, n_fft=512, hop=256, top_db=80.0):
    """float64 [1 + len(y)//hop, n_mfcc] of one segment (one call of the transform: top_db is relative to the
    segment's own maximum)."""
    y = np.asarray(y, dtype=np.float64)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="reflect")
    n_frames = 1 + len(y) // hop
    idx = np.arange(n_frames)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = yp[idx] * hann_periodic(n_fft)[None, :]
    power = np.abs(np.fft.rfft(frames, axis=1)) ** 2                       # [frames, 257]
    mel = power @ mel_filterbank(n_fft // 2 + 1, n_mels, sample_rate)      # [frames, n_mels]
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - top_db)
    return db @ dct_matrix(n_mfcc, n_mels)
