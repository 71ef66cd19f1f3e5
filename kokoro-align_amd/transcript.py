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
    """Where to cut the text when the audio is cut at a given phoneme.

    ``token_pos[p]`` is the index of the token in front of which the text is cut when the cut in the audio falls on
    phoneme ``p`` of the transcript (all tokens' phonemes counted through).  Rule of the reference
    (transcript.py:14-42): the phonemes of a token up to its midpoint - ``len - len // 2`` of them - send the cut
    to where it stood before the token, the rest send it behind the token; tokens without phonemes (``「``, a
    geminate ``q``, ...) are carried along by the next cut, except punctuation (`` , . ! ? ``), which pulls the cut
    behind itself.  ``len(aligner)`` is the number of phonemes that have a cut position."""

    def __init__(self, input_file):
        self._texts, self._vocas = [], []
        self.token_pos = []
        phonemes_seen = 0
        cut_at = 0
        for text, voca in _read_pairs(input_file):
            self._texts.append(text)
            self._vocas.append(voca)
            here = len(self._texts)              # a cut behind this token
            count = len(encode_text(voca))
            if count:
                phonemes_seen += count
                first_half_end = phonemes_seen - count // 2
                self.token_pos += [cut_at] * (first_half_end - len(self.token_pos))
                cut_at = here
            elif voca in _PUNCT:
                cut_at = here

    def __len__(self):
        return len(self.token_pos)

    def _cut(self, phoneme):
        return self.token_pos[phoneme] if phoneme < len(self.token_pos) else len(self._texts)

    def get_token(self, start, end, remove_wordsep=True):
        """(text, phonemes) of the tokens between the cuts of two phoneme indices (transcript.py:47-57); with
        ``remove_wordsep=False`` the tokens' phonemes are joined with the word separator `` _ ``, which is dropped next
        to punctuation."""
        a, b = self._cut(start), self._cut(end)
        text = ' '.join(t for t in self._texts[a:b] if t)
        vocas = [v for v in self._vocas[a:b] if v]
        if remove_wordsep:
            voca = ' '.join(vocas)
        else:
            voca = _SEP_AFTER_PUNCT.sub(r'\1', _SEP_BEFORE_PUNCT.sub(r'\1', ' _ '.join(vocas)))
        return text.strip(), voca.strip()
