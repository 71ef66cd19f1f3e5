"""Transcript side of the alignment boundary (mirror of kokoro_align/transcript.py:13-67).

``*.voca.txt`` holds one ``text|voca`` pair per token.  ``read_transcript`` gives the label
ids fed to the DP; ``VocaAligner`` maps a phoneme index on the best path back to a token
index so that ``align()`` can cut the text at segment boundaries.
(The G2P writer ``write_transcript`` is upstream of the hot path and not part of this package.)
"""
import re

from .encoder import encode_text

_PUNCT = (',', '.', '!', '?')
_SEP_BEFORE_PUNCT = re.compile(r'_ ([.,!?])')
_SEP_AFTER_PUNCT = re.compile(r'([.,!?]) _')


def _read_pairs(path):
    with open(path) as f:
        for line in f:
            yield line.rstrip('\r\n').split('|')


def read_transcript(input_file):
    """All voca columns joined -> int8 label ids (transcript.py:60-67)."""
    return encode_text(' '.join(parts[1] for parts in _read_pairs(input_file)))


class VocaAligner:
    """token_pos[p] = token index at which a cut falling on phoneme p is placed
    (transcript.py:14-42): phonemes up to a token's midpoint cut before it (swallowing
    preceding tokens without phonemes unless they are punctuation), later ones after it."""

    def __init__(self, input_file):
        self.text_tokens = []
        self.voca_tokens = []
        self.attach_dirs = []
        self.token_pos = []
        n_tokens = 0      # tokens seen so far
        n_phonemes = 0    # phonemes seen so far
        cut = 0           # token index a cut is currently placed at
        for text, voca in _read_pairs(input_file):
            n = len(encode_text(voca))
            self.text_tokens.append(text)
            self.voca_tokens.append(voca)
            n_tokens += 1
            if n > 0:
                n_phonemes += n
                upto = n_phonemes - n // 2
                self.token_pos.extend([cut] * (upto - len(self.token_pos)))
                cut = n_tokens
            elif voca in _PUNCT:
                cut = n_tokens

    def __len__(self):
        return len(self.token_pos)

    def get_token(self, start, end, remove_wordsep=True):
        """Text and phonemes of the tokens between two phoneme indices (transcript.py:47-57)."""
        n = len(self.token_pos)
        first = self.token_pos[start] if start < n else len(self.text_tokens)
        last = self.token_pos[end] if end < n else len(self.text_tokens)
        text = ' '.join(tok for tok in self.text_tokens[first:last] if tok)
        vocas = [tok for tok in self.voca_tokens[first:last] if tok]
        if remove_wordsep:
            voca = ' '.join(vocas)
        else:
            voca = ' _ '.join(vocas)
            voca = _SEP_BEFORE_PUNCT.sub(r'\1', voca)
            voca = _SEP_AFTER_PUNCT.sub(r'\1', voca)
        return text.strip(), voca.strip()
