"""Producer of the hot path's input: the AudioToChar CTC network under PyTorch-ROCm.

Mirror of kokoro_align/train.py:54-65 (2-layer bidirectional LSTM 40 -> 128x2, Linear 256 -> 39)
with the same state-dict keys (``lstm.*``, ``dense.*``), so the reference's ``ctc-last.pth``
(checkpoint dict key 'model', train.py:164-169) loads unchanged.  Inference only; PyTorch is
plumbing here (MIOpen runs the LSTM) — the product is the alignment kernel it feeds.
"""
import torch
from torch import nn
from torch.nn.utils.rnn import pack_sequence, pad_packed_sequence

from .encoder import VOCAB_SIZE

DEFAULT_PARAMS = dict(n_mfcc=40, hidden_dim=128, vocab_size=VOCAB_SIZE)


class AudioToChar(nn.Module):
    def __init__(self, n_mfcc=40, hidden_dim=128, vocab_size=VOCAB_SIZE):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.lstm = nn.LSTM(n_mfcc, hidden_dim, num_layers=2, dropout=0.5, bidirectional=True)
        self.dense = nn.Linear(2 * hidden_dim, vocab_size)

    def forward(self, packed_audio):
        """PackedSequence of MFCC segments -> (logits [max_len, batch, vocab], lengths)."""
        hidden, _ = self.lstm(packed_audio)
        hidden, lengths = pad_packed_sequence(hidden)
        return self.dense(hidden), lengths


def load_model(ckpt_path=None, device="cpu", params=None):
    model = AudioToChar(**(params or DEFAULT_PARAMS))
    if ckpt_path is not None:
        state = torch.load(ckpt_path, map_location="cpu")
        model.load_state_dict(state["model"] if "model" in state else state)
    return model.to(device).eval()


@torch.no_grad()
def segment_logits(model, segments, device=None, batch_size=128):
    """MFCC segments (list of [len_i, n_mfcc] arrays/tensors) -> list of logits [len_i, vocab]
    kept on ``device``, in the order given (reference loop: train.py:217-229, which only works
    on CPU because it never moves ``audio`` to the model's device)."""
    device = device or next(model.parameters()).device
    out = []
    for i in range(0, len(segments), batch_size):
        chunk = [torch.as_tensor(s, dtype=torch.float32).to(device) for s in segments[i:i + batch_size]]
        logits, lengths = model(pack_sequence(chunk, enforce_sorted=False))
        for j, n in enumerate(lengths.tolist()):
            out.append(logits[:n, j, :])
    return out


@torch.no_grad()
def lstm_logits_device(model, data, indices, device=None, persistent=True, timings=None, fuse_layer0=True):
    """Logits of every frame of an IndexDataArray (``data`` [rows, n_mfcc], ``indices`` = cumulative segment ends,
    kokoro_align/preprocess.py:12-35), [rows, vocab] on ``device`` in the file's row order - what the reference's
    predict() writes to *.logits.npz (train.py:215-231) - computed for ALL segments at once.

    The reference runs the network file by file, 128 segments per call; MIOpen's LSTM then spends ~100 us per
    time step whatever the batch, and refuses large batches.  Here every layer is
      * one library GEMM for the input projections of all frames of all segments (x @ W_ih^T + b_ih + b_hh),
      * then ONE launch of the persistent HIP kernel ka_lstm_layer_f32 (``persistent``, hidden size 128: a
        workgroup carries 32 sequences through all their steps, h @ W_hh^T on the f32 MFMA with W_hh resident in
        registers), or per time step one batched library GEMM + the fused cell kernel ka_lstm_step_f32,
    with the segments sorted by length.  Inference only (no dropout); float32; equal to the PyTorch network
    within rounding (tests: 1e-4).  ``timings`` (a dict): filled with the milliseconds of every stage (HIP events on the
    current stream; one synchronisation at the end).
    """
    import numpy as np
    from . import _lib
    lib = _lib.load_library()
    device = torch.device(device) if device is not None else next(model.parameters()).device
    if device.type != "cuda":
        raise _lib.KAError("lstm_logits_device needs a GPU (the CPU path is segment_logits)")
    ends = np.asarray(indices, dtype=np.int64).reshape(-1)
    n = int(ends.size)
    total = int(ends[-1]) if n else 0
    marks = []

    def mark(name):
        if timings is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(device))
            marks.append((name, ev))
    mark("start")
    x_file = torch.as_tensor(data, dtype=torch.float32)[:total].to(device)
    mark("h2d_mfcc")
    if total == 0:
        return torch.zeros((0, model.dense.out_features), dtype=torch.float32, device=device)
    starts = np.concatenate([[0], ends[:-1]])
    lens = ends - starts
    keep = np.nonzero(lens > 0)[0]                      # empty segments contribute no rows
    order = keep[np.argsort(-lens[keep], kind="stable")]
    n = int(order.size)
    slen = lens[order]
    max_len = int(slen[0])
    H = model.hidden_dim
    persistent = persistent and H == 128
    if persistent:
        # the persistent kernel addresses each sequence by (first row, length): the frames stay in the file's
        # row order, only the small per-segment tables are sorted (a tile of 32 runs for its longest member)
        offs = starts[order]
        x, perm = x_file, None
    else:
        offs = np.concatenate([[0], np.cumsum(slen)[:-1]])  # row offsets in the length-sorted layout
        # sorted row r of segment i (sorted position s): file row starts[order[s]] + (r - offs[s])
        seg_of_row = np.repeat(np.arange(n), slen)
        perm = torch.from_numpy((starts[order][seg_of_row] + (np.arange(total) - offs[seg_of_row])).astype(np.int64)).to(device)
        x = x_file.index_select(0, perm)
    d_offs = torch.from_numpy(np.ascontiguousarray(offs)).to(device)
    d_len = torch.from_numpy(np.ascontiguousarray(slen)).to(device)
    d_offs32, d_len32 = d_offs.to(torch.int32), d_len.to(torch.int32)
    if not persistent:
        # sequences still running at step t (lengths are sorted descending), frame rows of every (direction, step, sequence)
        n_run = (n - np.searchsorted(slen[::-1], np.arange(max_len), side="right")).tolist()
        steps = torch.arange(max_len, dtype=torch.int64, device=device).unsqueeze(1)
        rows = torch.stack([d_offs.unsqueeze(0) + steps, d_offs.unsqueeze(0) + d_len.unsqueeze(0) - 1 - steps], 0)   # [2, max_len, n]
        rows = rows.clamp_(0, total - 1).to(torch.int32).contiguous()
        rows_ptr, rows_step, rows_dir = rows.data_ptr(), n * 4, rows.stride(0)
    stream = torch.cuda.current_stream(device).cuda_stream
    sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in model.state_dict().items()}
    inp = x
    for layer in range(model.lstm.num_layers):
        sfx = [f"_l{layer}", f"_l{layer}_reverse"]
        w_ih = torch.cat([sd["lstm.weight_ih" + s] for s in sfx], 0)                       # [8H, in]
        bias = torch.cat([sd["lstm.bias_ih" + s] + sd["lstm.bias_hh" + s] for s in sfx], 0)  # [8H]
        w_hh_t = torch.stack([sd["lstm.weight_hh" + s].t().contiguous() for s in sfx], 0)  # [2, H, 4H]
        out = torch.empty((total, 2 * H), dtype=torch.float32, device=device)
        if persistent and layer == 0 and fuse_layer0 and inp.shape[1] == 40:
            # layer 0: the input projection (K = 40) runs inside the recurrence kernel - no [total, 8H] intermediate
            w_hh = torch.stack([sd["lstm.weight_hh" + s] for s in sfx], 0).contiguous()      # [2, 4H, H]
            w_ih_c = w_ih.contiguous()                                                       # [2 * 4H, 40]
            _lib.check(lib.ka_lstm_layer0_f32(inp.data_ptr(), inp.stride(0), 40, w_ih_c.data_ptr(), bias.data_ptr(), w_hh.data_ptr(),
                                              out.data_ptr(), out.stride(0), d_offs32.data_ptr(), d_len32.data_ptr(), n, H, stream),
                       "ka_lstm_layer0_f32")
            mark(f"projection+recurrence_l{layer}")
            inp = out
            continue
        gin = torch.addmm(bias, inp, w_ih.t())                                             # [total, 8H]
        mark(f"input_projection_l{layer}")
        if persistent:
            # the whole layer in one launch: ka_lstm_layer_f32 (f32 MFMA, W_hh register-resident, h in LDS)
            w_hh = torch.stack([sd["lstm.weight_hh" + s] for s in sfx], 0).contiguous()      # [2, 4H, H]
            _lib.check(lib.ka_lstm_layer_f32(gin.data_ptr(), gin.stride(0), w_hh.data_ptr(), out.data_ptr(), out.stride(0),
                                             d_offs32.data_ptr(), d_len32.data_ptr(), n, H, stream), "ka_lstm_layer_f32")
            mark(f"recurrence_l{layer}")
            del gin
            inp = out
            continue
        h = torch.zeros((2, n, H), dtype=torch.float32, device=device)
        c = torch.zeros((2, n, H), dtype=torch.float32, device=device)
        rec = torch.empty((2, n, 4 * H), dtype=torch.float32, device=device)
        args = (gin.data_ptr(), gin.stride(0), rec.data_ptr(), rec.stride(0), c.data_ptr(), h.data_ptr(), h.stride(0),
                out.data_ptr(), out.stride(0))
        step = lib.ka_lstm_step_f32
        for t in range(max_len):
            # fixed shape on purpose (finished sequences included, their result is ignored): a new GEMM shape
            # per step would cost a library heuristic lookup each
            torch.bmm(h, w_hh_t, out=rec)                                                  # [2, n, 4H]
            rc = step(*args, rows_ptr + t * rows_step, rows_dir, n_run[t], H, stream)
            if rc:
                _lib.check(rc, "ka_lstm_step_f32")
        mark(f"recurrence_l{layer}")
        del gin
        inp = out
    logits_sorted = torch.addmm(sd["dense.bias"], inp, sd["dense.weight"].t())
    mark("dense")
    if timings is not None:
        torch.cuda.synchronize(device)
        for (_, a), (name, b) in zip(marks[:-1], marks[1:]):
            timings[name] = timings.get(name, 0.0) + a.elapsed_time(b)
    if perm is None:
        return logits_sorted
    logits = torch.empty_like(logits_sorted)
    logits.index_copy_(0, perm, logits_sorted)           # back to the file's row order
    return logits


@torch.no_grad()
def segment_logits_device(model, segments, device=None, persistent=True):
    """``segment_logits`` through ``lstm_logits_device``: list of [len_i, n_mfcc] -> list of [len_i, vocab] on the
    device, in the order given."""
    import numpy as np
    if len(segments) == 0:
        return []
    lens = [int(s.shape[0]) for s in segments]
    data = torch.cat([torch.as_tensor(s, dtype=torch.float32) for s in segments], dim=0)
    logits = lstm_logits_device(model, data, np.cumsum(lens), device=device, persistent=persistent)
    out, k = [], 0
    for n in lens:
        out.append(logits[k:k + n])
        k += n
    return out
