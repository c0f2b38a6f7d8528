"""Minimal text front end for the drop-in (`tokenize`, `convert_tokens_to_ids`, `split_sentences`).

The reference's normaliser / tokenizer (/root/reference/indextts/utils/front.py) is host-side string work and
out of the hot-path scope: this shim wraps sentencepiece when `bpe.model` is present and
otherwise accepts pre-tokenised input.  Sentence splitting follows the documented behaviour of
`split_sentences_by_token` (front.py:344-428): cut after sentence punctuation, fall back to commas / hyphens /
hard cuts above the token cap, then merge short neighbours."""
from __future__ import annotations

import os
from typing import List, Sequence

PUNCT = (".", "!", "?", "▁.", "▁?", "▁...")


def _split(tokens: Sequence[str], marks: Sequence[str], cap: int) -> List[List[str]]:
    out: List[List[str]] = []
    cur: List[str] = []
    for i, tok in enumerate(tokens):
        cur.append(tok)
        nxt = tokens[i + 1] if i + 1 < len(tokens) else None
        if tok in marks and len(cur) > 2 and nxt not in ("'", "▁'") and len(cur) <= cap:
            out.append(cur)
            cur = []
        elif len(cur) > cap:
            if not any(m in marks for m in (",", "▁,")) and any(t in (",", "▁,") for t in cur):
                out.extend(_split(cur, (",", "▁,"), cap))
            elif "-" not in marks and "-" in cur:
                out.extend(_split(cur, ("-",), cap))
            else:
                out.extend([cur[j:j + cap] for j in range(0, len(cur), cap)])
            cur = []
    if cur:
        out.append(cur)
    return out


def merge_short(sentences: List[List[str]], cap: int) -> List[List[str]]:
    merged: List[List[str]] = []
    for s in sentences:
        if merged and len(merged[-1]) + len(s) <= cap:
            merged[-1] = merged[-1] + list(s)
        else:
            merged.append(list(s))
    return merged


class TextTokenizer:
    def __init__(self, vocab_file: str, normalizer=None):
        self.normalizer = normalizer
        self.sp = None
        if vocab_file and os.path.exists(vocab_file):
            import sentencepiece as spm

            self.sp = spm.SentencePieceProcessor(model_file=vocab_file)

    def _need(self):
        if self.sp is None:
            raise RuntimeError("no bpe.model: pass pre-tokenised ids (list of int lists) instead of a string")

    def tokenize(self, text: str) -> List[str]:
        self._need()
        if self.normalizer is not None:
            text = self.normalizer.normalize(text)
        return self.sp.encode(text, out_type=str)

    def convert_tokens_to_ids(self, tokens):
        self._need()
        return [self.sp.piece_to_id(t) for t in tokens]

    def convert_ids_to_tokens(self, ids):
        self._need()
        return [self.sp.id_to_piece(int(i)) for i in ids]

    def encode(self, text: str):
        return self.convert_tokens_to_ids(self.tokenize(text))

    def split_sentences(self, tokenized: Sequence[str], max_tokens_per_sentence: int = 120) -> List[List[str]]:
        return merge_short(_split(list(tokenized), PUNCT, max_tokens_per_sentence), max_tokens_per_sentence)
