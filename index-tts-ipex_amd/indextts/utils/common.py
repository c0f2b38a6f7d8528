"""Small text helpers of the drop-in (/root/reference/indextts/utils/common.py:29-84): CJK pre-tokenisation in front of
SentencePiece and its inverse, `safe_log`."""
from __future__ import annotations

import re

import torch

# CJK blocks (Hangul Jamo, CJK radicals .. Yi, Hangul syllables, compatibility ideographs / forms, half-width Hangul, plane 2)
_CJK = re.compile("([\u1100-\u11ff\u2e80-\ua4cf\ua840-\ud7af\uf900-\ufaff\ufe30-\ufe4f\uff65-\uffdc\U00020000-\U0002ffff])")
_LATIN_RUN = re.compile(r"([A-Z]+(?:[\s-][A-Z-]+)*)", re.IGNORECASE)


def tokenize_by_CJK_char(line: str, do_upper_case: bool = True) -> str:
    """"你好世界是 hello world 的中文" -> "你 好 世 界 是 HELLO WORLD 的 中 文": every CJK character becomes a word of its own,
    everything between is kept (upper-cased by default)."""
    parts = [w.strip() for w in _CJK.split(line.strip())]
    return " ".join((w.upper() if do_upper_case else w) for w in parts if w)


def de_tokenized_by_CJK_char(line: str, do_lower_case: bool = False) -> str:
    """Inverse: drop the spaces between CJK characters, keep those inside runs of Latin words."""
    runs = _LATIN_RUN.findall(line)
    for i, run in enumerate(runs):
        line = line.replace(run, f"<sent_{i}>")
    words = line.split()
    mark = re.compile(r"^.*?(<sent_(\d+)>)")
    for i, w in enumerate(words):
        m = mark.match(w)
        if m:
            w = w.replace(m.group(1), runs[int(m.group(2))])
            words[i] = w.lower() if do_lower_case else w
    return "".join(words)


def safe_log(x: torch.Tensor, clip_val: float = 1e-7) -> torch.Tensor:
    return torch.log(torch.clip(x, min=clip_val))
