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
