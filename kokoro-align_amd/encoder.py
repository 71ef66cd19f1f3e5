"""Phoneme codec used at the alignment boundary (mirror of kokoro_align/encoder.py:5-31).

39 symbols; index 0 ('_') is both the CTC blank and the word separator."""
import re

import numpy as np

vocab = ('_ N a a: b by ch d e e: f g gy h hy i i: j k ky m my'
         ' n ny o o: p py r ry s sh t ts u u: w y z').split(' ')
v2i = {sym: idx for idx, sym in enumerate(vocab)}
accepted_vocab = set(vocab) | {'q', '.', ',', '!', '?'}

VOCAB_SIZE = len(vocab)

_REPEAT_RX = re.compile(r'(.+)( \1)+')


def is_valid_text(text):
    """True when every space-separated token is a phoneme or accepted punctuation (encoder.py:14-15)."""
    for token in text.split():
        if token not in accepted_vocab:
            return False
    return True


def encode_text(text):
    """Phoneme string -> int8 ids; tokens outside the 39-symbol vocab are dropped (encoder.py:18-19)."""
    ids = [v2i[token] for token in text.split() if token in v2i]
    return np.array(ids, dtype=np.int8)


def decode_text(encoded):
    """ids -> space-joined phoneme string (encoder.py:22-23)."""
    return ' '.join(vocab[int(i)] for i in encoded)


def merge_repeated(text):
    """Collapse repeats with the reference's greedy group regex, then drop blanks (encoder.py:26-31).
    Not a plain run-length merge: '(.+)( \\1)+' can merge multi-token groups and leaves some
    odd-length runs partly unmerged; reproduced as is."""
    merged = _REPEAT_RX.sub(r'\1', text)
    merged = merged.replace(' _', '').replace('_ ', '')
    return '' if merged == '_' else merged
