"""CPU: host-side front-end pieces of the drop-in (mel features, sentence splitting) - analytic checks, the
reference's torchaudio / WeTextProcessing being absent offline."""
import math

import numpy as np
import torch

from indextts.utils.feature_extractors import MelSpectrogramFeatures, mel_filterbank
from indextts.utils.front import TextTokenizer


def test_mel_shapes_and_tone():
    m = MelSpectrogramFeatures()
    sr, f0 = 24000, 1000.0
    x = torch.sin(2 * math.pi * f0 * torch.arange(sr) / sr)[None]
    y = m(x)
    assert y.shape == (1, 100, sr // 256 + 1)
    assert float(y.min()) >= math.log(1e-7) - 1e-6
    fb = mel_filterbank(513, 0.0, 12000.0, 100, sr)
    assert fb.shape == (513, 100) and float(fb.min()) >= 0
    centre = int(torch.argmax(fb[int(round(f0 / (sr / 1024)))]))
    assert int(torch.argmax(y[0, :, 40])) in (centre - 1, centre, centre + 1)
    # magnitude (power=1) spectrogram: doubling the amplitude adds log 2
    y2 = m(2 * x)
    k = int(torch.argmax(y[0, :, 40]))
    assert abs(float(y2[0, k, 40] - y[0, k, 40]) - math.log(2)) < 1e-3


def test_split_sentences():
    import warnings

    split = TextTokenizer.split_sentences_by_token
    toks = list("ab.cde!fghij,klm?")
    s = split(toks, (".", "!", "?"), 6)
    assert sum(s, []) == toks and max(len(x) for x in s) <= 6
    long = ["w"] * 25
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = split(long, TextTokenizer.punctuation_marks_tokens, 10)
    assert sum(s, []) == long and max(len(x) for x in s) <= 10  # hard cut above the cap (front.py:389-396)
    assert split(list("a.b.c."), (".",), 120) == [list("a.b.c.")]
