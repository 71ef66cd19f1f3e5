"""Stages around the hot path, device-resident: logits producer -> log-softmax -> CTC best path
-> align.txt -> metadata.txt.  Same files, keys and text formats as the reference, so the output
of any stage can be consumed by the reference's next stage and vice versa.

Reference stages mirrored (run_example.py:146-280 `process`, stages 4-7):
  predict        kokoro_align/train.py:201-231   *.mfcc.npz  -> *.logits.npz + *.greed.txt
  best_path      kokoro_align/align.py:112-124   *.logits.npz + *.voca.txt -> *.best_path.npz
  align          kokoro_align/align.py:127-169   -> *.align.txt
  combine_files  run_example.py:73-131           *.align.txt + *.split.txt -> <id>.metadata.txt
Every stage is skipped when its output exists (run_example.py:227-228, :249-250, :262-263, :274-275).
process_alignment_sharded runs a dataset over the ranks of a node (one process per GPU, files split by work).
Upstream stages (Aozora text, G2P, MP3 split + MFCC) are not part of this package.
"""
import os

import numpy as np

from .align import _host_log_softmax, align as _write_align, log_softmax_device
from .encoder import decode_text, encode_text, is_valid_text, merge_repeated


# ------------------------------------------------------------------------------------------
# IndexDataArray npz format (kokoro_align/preprocess.py:12-35): `indices` = cumulative row ends
# (int32), `data` = all segments concatenated along axis 0
# ------------------------------------------------------------------------------------------
class IndexDataArray:
    """Writer of the segmented-array npz the reference's stages exchange (kokoro_align/preprocess.py:12-35): segments are
    appended with write(); on a clean exit of the ``with`` block ONE npz is written with `indices` (int32, cumulative
    row count after every segment) and `data` (all segments stacked on axis 0); after an exception nothing is written."""

    def __init__(self, file):
        self.file = file
        self._segments = []

    def __enter__(self):
        return self

    def write(self, segment):
        self._segments.append(np.asarray(segment))

    def __exit__(self, exc_type, exc_value, traceback):
        if exc_type is not None:
            return False
        ends = np.cumsum([seg.shape[0] for seg in self._segments], dtype=np.int64).astype(np.int32)
        np.savez(self.file, indices=ends, data=np.concatenate(self._segments, axis=0))
        return False


def open_index_data_for_write(file):
    return IndexDataArray(file)


def read_index_data(file):
    """-> (indices int32 [n_seg], data [rows, ...])"""
    with np.load(file) as f:
        return f['indices'], f['data']


def split_segments(indices, data):
    out, start = [], 0
    for end in indices.tolist():
        out.append(data[start:end])
        start = end
    return out


# ------------------------------------------------------------------------------------------
# stage: predict (logits producer)
# ------------------------------------------------------------------------------------------
def _write_logits(indices, logits, output_file, text_file):
    """One file's *.logits.npz (IndexDataArray) and *.greed.txt as the reference writes them (train.py:215-231)."""
    import torch
    try:
        with open_index_data_for_write(output_file) as out, open(text_file, 'wt') as txt:
            start = 0
            greedy_all = torch.argmax(logits, dim=-1).cpu().numpy()
            host = logits.detach().cpu().numpy().astype(np.float32)
            for i, end in enumerate(np.asarray(indices).tolist()):
                out.write(host[start:end])
                txt.write(f'{i + 1}|{merge_repeated(decode_text(greedy_all[start:end]))}\n')
                start = end
    except BaseException:
        for f in (output_file, text_file):
            if os.path.exists(f):
                os.unlink(f)
        raise


def predict_files(model, audio_files, output_files, text_files, device=None, log_probs_out=None):
    """predict() for many files at once: the MFCC segments of ALL files go through the network together
    (kokoro_align_amd.model.lstm_logits_device: a time step of the LSTM costs the same for 60 sequences or
    4000), then every file's outputs are written in the reference's formats.  Returns {output_file: logits
    [T_file, vocab] on the device}.  ``log_probs_out`` (a dict): also filled with {output_file: log-probs of align.py:116-117},
    computed for the whole dataset by ONE launch of the HIP log-softmax kernel (the files' rows are one tensor here)."""
    import torch
    from .model import lstm_logits_device
    device = device or next(model.parameters()).device
    datas, ends, per_file, base = [], [], [], 0
    for af in audio_files:
        indices, data = read_index_data(af)
        rows = int(indices[-1]) if len(indices) else 0
        datas.append(np.asarray(data[:rows], dtype=np.float32))
        ends.append(np.asarray(indices, dtype=np.int64) + base)
        per_file.append((np.asarray(indices, dtype=np.int64), base, rows))
        base += rows
    if not audio_files:
        return {}
    logits = lstm_logits_device(model, torch.from_numpy(np.concatenate(datas, axis=0)), np.concatenate(ends), device=device)
    log_probs = log_softmax_device(logits) if log_probs_out is not None else None
    result = {}
    for (indices, b, rows), of, tf in zip(per_file, output_files, text_files):
        lg = logits[b:b + rows]
        _write_logits(indices, lg, of, tf)
        result[of] = lg
        if log_probs is not None:
            log_probs_out[of] = log_probs[b:b + rows]
    return result


def predict(model, audio_file, output_file, text_file, device=None, batch_size=128):
    """MFCC segments -> per-segment logits, written as the reference writes them
    (train.py:215-231: `*.logits.npz` via IndexDataArray, `*.greed.txt` lines
    f'{index+1}|{merge_repeated(greedy decode)}').  Returns the logits [T_file, vocab] as ONE
    device tensor (segments concatenated, like the file's `data`) so that the DP can take it by
    pointer without re-reading the file.  On a GPU the network runs through lstm_logits_device (all
    segments at once); on the CPU through PyTorch, ``batch_size`` segments per call like the reference."""
    import torch
    from .model import segment_logits
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if device.type == "cuda":
        return predict_files(model, [audio_file], [output_file], [text_file], device=device)[output_file]
    indices, data = read_index_data(audio_file)
    seg_logits = segment_logits(model, split_segments(indices, data), device=device, batch_size=batch_size)
    logits = torch.cat(seg_logits, dim=0) if seg_logits else torch.zeros((0, 0), device=device)
    _write_logits(indices, logits, output_file, text_file)
    return logits


# ------------------------------------------------------------------------------------------
# stage: best_path for many files in one launch
# ------------------------------------------------------------------------------------------
def best_path_files(logits_files, voca_files, best_path_files_out, device=None, logits_on_device=None,
                    host_softmax=False, log_probs_on_device=None):
    """All files of a dataset as ONE batched launch (the reference loops them, run_example.py:248-254).

    ``logits_on_device`` (optional) maps a logits file name to a device tensor produced by
    ``predict`` in this process; files not in the map are read from disk.  ``host_softmax``
    computes align.py:116-117 with NumPy on the host (bit-identical log-probs to the
    reference); the default runs the HIP log-softmax kernel on the device (1e-6) - once per file, or not at all for the
    files whose log-probs ``log_probs_on_device`` already holds (predict_files computes them for a whole dataset in one launch).
    Files whose output exists are skipped.  Returns the list of files written.
    """
    import torch
    from .transcript import read_transcript
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    todo = [(lf, vf, bf) for lf, vf, bf in zip(logits_files, voca_files, best_path_files_out)
            if not os.path.exists(bf)]
    if not todo:
        return []
    lps, labs = [], []
    for lf, vf, _ in todo:
        if not host_softmax and log_probs_on_device is not None and lf in log_probs_on_device:
            lps.append(log_probs_on_device[lf])
            labs.append(read_transcript(vf))
            continue
        if logits_on_device is not None and lf in logits_on_device:
            logits = logits_on_device[lf]
        else:
            with np.load(lf) as f:
                logits = f['data']
        if host_softmax:
            lg = logits.detach().cpu().numpy() if hasattr(logits, "detach") else logits
            lps.append(torch.from_numpy(np.ascontiguousarray(_host_log_softmax(lg), np.float32)).to(dev))
        else:
            t = logits if hasattr(logits, "detach") else torch.from_numpy(np.ascontiguousarray(logits, np.float32))
            lps.append(log_softmax_device(t.to(dev)))
        labs.append(read_transcript(vf))
    # One launch for all chapters, but the reference's per-file progress (run_example.py:248-254 loops the files: what was
    # written before a failing chapter stays written and is skipped on the rerun): every chapter whose status is 0 gets
    # its file, then the first failing chapter raises what the reference's best_path() would have raised for it.
    from . import _lib
    from .align import DeviceBatch
    for lp in lps:
        if lp.shape[0] == 0:
            raise IndexError("list index out of range")
    with torch.cuda.device(dev):
        batch = DeviceBatch(lps, [torch.as_tensor(np.asarray(x).reshape(-1).astype(np.int32)) for x in labs])
        status = batch.run(raise_on_error=False)
    written, first_bad = [], None
    for (_, _, bf), (p, l, s), st in zip(todo, batch.results(), status.tolist()):
        if st != 0:
            first_bad = first_bad if first_bad is not None else (bf, st)
            continue
        try:
            np.savez(bf, best_path=p.cpu().numpy(), best_labels=l.cpu().numpy(), best_scores=s.cpu().numpy())
        except BaseException:
            if os.path.exists(bf):
                os.unlink(bf)
            raise
        written.append(bf)
    if first_bad is not None:
        _lib.check(first_bad[1], f"best_path -> {first_bad[0]}")
    return written


# ------------------------------------------------------------------------------------------
# stage: metadata (run_example.py:73-131)
# ------------------------------------------------------------------------------------------
_NG_WORDS = ('リブリボックス', 'ボランティアについてなど', 'この録音はパブリックドメイン', 'ために録音されました')


def _blocked_by_ng_word(text, voca):
    """LibriVox boiler-plate or an empty line (run_example.py:84-88)."""
    if text.strip() and voca.strip():
        squeezed = text.replace(' ', '')
        return any(w in squeezed for w in _NG_WORDS)
    return True


def _blocked_by_unknown_yomi(voca, remove_wordsep):
    """run_example.py:97-100"""
    if remove_wordsep and '_' in voca:
        return True
    return not is_valid_text(voca)


def _blocked_by_few_matches(voca, decoded):
    """Keep only segments where the decoded label count exceeds 70 % of the transcript's
    (run_example.py:90-95)."""
    n_voca = int(np.count_nonzero(encode_text(voca)))
    n_dec = len(decoded.split())
    return not (n_voca and n_dec and n_dec / n_voca > 0.7)


def combine_files(dataset, align_files, audio_files, segment_files, metadata_file, remove_wordsep, verbose=True):
    """align.txt + split.txt -> `<id>|<audio file>|<start>|<end>|<text>|<voca>` lines
    (run_example.py:73-131).  start/end are audio sample offsets from split.txt."""
    say = print if verbose else (lambda *a, **k: None)
    os.makedirs(os.path.dirname(metadata_file), exist_ok=True)
    try:
        with open(metadata_file, 'wt') as out:
            idx = 1
            for align_file, audio_file, segment_file in zip(align_files, audio_files, segment_files):
                audio_name = os.path.basename(audio_file)
                with open(align_file, 'rt') as af, open(segment_file, 'rt') as sf:
                    start = 0
                    for aline, sline in zip(af, sf):
                        _, text, voca, decoded, _, _, _ = aline.rstrip('\r\n').split('|')
                        end, = sline.rstrip('\r\n').split('|')
                        if _blocked_by_ng_word(text, voca):
                            say(f'Blocking by NG word: {text}')
                        elif _blocked_by_unknown_yomi(voca, remove_wordsep):
                            say(f'Blocking by unknown yomi {voca}')
                        elif _blocked_by_few_matches(voca, decoded):
                            say('Blocking by too few match')
                            say(f'voca:    {voca}')
                            say(f'decoded: {decoded}')
                        else:
                            out.write(f'{dataset}-{idx:05d}|{audio_name}|{start}|{end}|{text}|{voca}\n')
                            idx += 1
                        start = end
    except BaseException:
        os.unlink(metadata_file)
        raise


# ------------------------------------------------------------------------------------------
# stages 4-7 of run_example.process for one dataset
# ------------------------------------------------------------------------------------------
def _swap_ext(files, old, new):
    return [f[:-len(old)] + new if f.endswith(old) else f for f in files]


def process_alignment_sharded(dataset, audio_files, metadata_file, model=None, remove_wordsep=False, device=None,
                              host_softmax=False, verbose=True, rank=None, world_size=None, barrier=None,
                              best_path_files_fn=None, any_rank=None):
    """process_alignment for one dataset over the ranks of a node: one process per GPU (torch.distributed, RCCL);
    BASELINE.json configs[3] ("Meian sharded across 8 GPUs").

    The reference loops the files of a dataset one after the other (run_example.py:224-267) and they share nothing, so
    the files are split over the ranks - longest-processing-time first on the work T x min(beam, 2S+1) of their
    lattices (sharding.shard_for_rank; T from the `*.mfcc.npz` indices, S from the `*.voca.txt`) - and every rank runs
    predict -> best_path -> align for ITS files in one batched launch and writes their per-file outputs, exactly the
    files the reference would write.  No collective touches the data path; after a barrier rank 0 joins all
    `*.align.txt` into the metadata file (run_example.py:273-278).  The model is assumed to be on every rank already
    (sharding.broadcast_model_weights at start-up).  `rank` / `world_size` / `barrier` default to torch.distributed's;
    `best_path_files_fn` replaces the DP stage (the CPU tests put the oracle there: the product has no CPU path).
    `any_rank(flag) -> bool` tells every rank whether ANY rank raised the flag (default: an all-reduce MAX over
    torch.distributed; it is also the second barrier): when one rank fails, every rank raises and rank 0 does not merge
    files the failed rank never wrote.  Returns the metadata file name on rank 0, None elsewhere."""
    from .sharding import shard_for_rank
    from .transcript import read_transcript
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank, world_size = dist.get_rank(), dist.get_world_size()
        if barrier is None:
            barrier = dist.barrier
        if any_rank is None:
            def any_rank(flag):
                import torch
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
                t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return bool(int(t.item()))
    # The split_audio stage first, files dealt round-robin (its outputs size the shards below): a dataset whose MFCCs have not
    # been made yet goes through the same stage process_alignment would run, instead of failing on a missing file.
    failure = None
    try:
        for af in audio_files[rank::world_size]:
            _ensure_mfcc(af, print if verbose else (lambda *a, **k: None))
    except Exception as exc:           # every rank must reach the barrier: a rank that raised before it would hang the others
        failure = exc
    if barrier is not None:
        barrier()
    if failure is None:
        try:
            shapes = []
            for af in audio_files:
                mf, vf = _swap_ext([af], '.mp3', '.mfcc.npz')[0], _swap_ext([af], '.mp3', '.voca.txt')[0]
                with np.load(mf) as f:
                    ends = f['indices']
                shapes.append((int(ends[-1]) if len(ends) else 0, int(len(read_transcript(vf)))))
            mine = [audio_files[i] for i in shard_for_rank(shapes, rank, world_size)]
            if mine:
                process_alignment(dataset, mine, None, model=model, remove_wordsep=remove_wordsep, device=device,
                                  host_softmax=host_softmax, verbose=verbose, best_path_files_fn=best_path_files_fn)
        except Exception as exc:
            failure = exc
    # the failure is shared: the other ranks must not report success, and rank 0 must not merge a partial dataset
    if any_rank is not None:
        somebody_failed = any_rank(failure is not None)
    else:
        somebody_failed = failure is not None
        if barrier is not None:
            barrier()
    if failure is not None:
        raise failure
    if somebody_failed:
        raise RuntimeError(f"process_alignment_sharded({dataset}): another rank failed; {metadata_file} was not written")
    if rank != 0:
        return None
    say = print if verbose else (lambda *a, **k: None)
    if os.path.exists(metadata_file):
        say(f'Skip writing {metadata_file}')
    else:
        say(f'Writing {metadata_file}')
        combine_files(dataset, _swap_ext(audio_files, '.mp3', '.align.txt'), audio_files,
                      _swap_ext(audio_files, '.mp3', '.split.txt'), metadata_file, remove_wordsep, verbose=verbose)
    return metadata_file


def _ensure_mfcc(af, say):
    """The split_audio stage of one file (run_example.py:205-218): `<x>.split.txt` + `<x>.mfcc.npz` from the decoded audio
    `<x>.wav` / `<x>.npy`, skipped when both exist."""
    sf, mf = _swap_ext([af], '.mp3', '.split.txt')[0], _swap_ext([af], '.mp3', '.mfcc.npz')[0]
    if os.path.exists(sf) and os.path.exists(mf):
        say(f'Skip converting {af} to MFCC')
        return
    decoded = [f for f in _swap_ext([af], ".mp3", ".wav") + _swap_ext([af], ".mp3", ".npy") if os.path.exists(f)]
    if not decoded:
        raise FileNotFoundError(f'{sf} / {mf} are missing and there is no decoded audio ({af[:-4]}.wav or .npy) to make them from')
    say(f'Converting {decoded[0]} to MFCC')
    from .preprocess import split_audio
    split_audio(decoded[0], sf, mf)


def process_alignment(dataset, audio_files, metadata_file, model=None, remove_wordsep=False, device=None,
                      host_softmax=False, verbose=True, best_path_files_fn=None):
    """Given `<x>.mp3` names whose `<x>.voca.txt` exist (made by the reference's upstream stages), run
    split_audio -> predict -> best_path -> align -> combine_files with skip-if-exists (run_example.py:205-275),
    all lattices of the dataset in one DP launch.  The split_audio stage (run_example.py:205-218) runs where
    `<x>.split.txt` / `<x>.mfcc.npz` are missing and needs the decoded audio as `<x>.wav` (16-bit PCM) or
    `<x>.npy` next to the mp3 name: audio decoding is not part of this package."""
    say = print if verbose else (lambda *a, **k: None)
    mfcc = _swap_ext(audio_files, '.mp3', '.mfcc.npz')
    split = _swap_ext(audio_files, '.mp3', '.split.txt')
    for af in audio_files:
        _ensure_mfcc(af, say)
    voca = _swap_ext(audio_files, '.mp3', '.voca.txt')
    logits = _swap_ext(audio_files, '.mp3', '.logits.npz')
    greed = _swap_ext(audio_files, '.mp3', '.greed.txt')
    bpath = _swap_ext(audio_files, '.mp3', '.best_path.npz')
    align_out = _swap_ext(audio_files, '.mp3', '.align.txt')
    on_device, lp_on_device = {}, {}
    missing = []
    for mf, lf, gf in zip(mfcc, logits, greed):
        if os.path.exists(lf) and os.path.exists(gf):   # both outputs, like run_example.py:227
            say(f'Skip writing {lf}')
        else:
            if model is None:
                raise ValueError(f'{lf} is missing and no model was given')
            say(f'Writing {lf}')
            missing.append((mf, lf, gf))
    if missing:
        import torch
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type == "cuda":     # every missing file through the network in one go
            on_device = predict_files(model, [m[0] for m in missing], [m[1] for m in missing], [m[2] for m in missing], device=dev,
                                      log_probs_out=None if host_softmax else lp_on_device)
        else:
            for mf, lf, gf in missing:
                on_device[lf] = predict(model, mf, lf, gf, device=dev)
    extra = {} if best_path_files_fn is not None else {"log_probs_on_device": lp_on_device}
    written = (best_path_files_fn or best_path_files)(logits, voca, bpath, device=device, logits_on_device=on_device,
                                                       host_softmax=host_softmax, **extra)
    for bf in bpath:
        say(f'Writing {bf}' if bf in written else f'Skip writing {bf}')
    for bf, mf, vf, af in zip(bpath, mfcc, voca, align_out):
        if os.path.exists(af):
            say(f'Skip writing {af}')
        else:
            say(f'Writing {af}')
            _write_align(bf, mf, vf, af, remove_wordsep=remove_wordsep)
    if metadata_file is None:      # a rank's share of a dataset (process_alignment_sharded): rank 0 writes the metadata
        return None
    if os.path.exists(metadata_file):
        say(f'Skip writing {metadata_file}')
    else:
        say(f'Writing {metadata_file}')
        combine_files(dataset, align_out, audio_files, split, metadata_file, remove_wordsep, verbose=verbose)
    return metadata_file
