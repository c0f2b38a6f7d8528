"""CPU: the drop-in text front end (indextts/utils/front.py, common.py, voices.py) against known answers produced by the
REFERENCE's own functions (tests/golden/front_cases.json, oracle/make_golden.py --front): sentence splitting on token
lists, CJK pre-tokenisation, TextNormalizer.normalize with the third-party written-form normalisers stubbed to identity
on both sides, and the web UI's saved-voice file format."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from indextts.utils import common, voices
from indextts.utils.front import TextNormalizer, TextTokenizer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "front_cases.json")


@pytest.fixture(scope="module")
def cases():
    with open(GOLD, encoding="utf-8") as f:
        return json.load(f)


def test_split_sentences_by_token(cases):
    for c in cases["splits"]:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = TextTokenizer.split_sentences_by_token(c["tokens"], TextTokenizer.punctuation_marks_tokens, c["cap"])
        assert out == c["out"], (c["tokens"], c["cap"])
        assert [t for s in out for t in s] == c["tokens"] and all(len(s) <= c["cap"] for s in out)


def test_normalizer_matches_reference_with_identity_third_party(cases):
    class Ident:
        def normalize(self, t):
            return t

    tn = TextNormalizer()
    tn._loaded = True
    tn.zh_normalizer, tn.en_normalizer = Ident(), Ident()
    for c in cases["normalize"]:
        assert tn.use_chinese(c["text"]) == c["use_chinese"], c["text"]
        assert tn.normalize(c["text"]) == c["out"], c["text"]
    for c in cases["correct_pinyin"]:
        assert tn.correct_pinyin(c["in"]) == c["out"]


def test_normalizer_without_third_party_warns_and_still_folds():
    tn = TextNormalizer()
    try:
        import tn as _tn  # noqa: F401

        pytest.skip("WeTextProcessing installed")
    except ImportError:
        pass
    with pytest.warns(RuntimeWarning, match="WeTextProcessing"):
        out = tn.normalize("受不liao3你了：“好”。")
    assert out == "受不liao3你了,'好'."  # pinyin protected and restored (only j/q/x + u forms are re-spelled)


def test_cjk_tokenisation(cases):
    for c in cases["cjk"]:
        assert common.tokenize_by_CJK_char(c["in"]) == c["tok"]
        assert common.tokenize_by_CJK_char(c["in"], do_upper_case=False) == c["tok_keep"]
    for c in cases["detok"]:
        assert common.de_tokenized_by_CJK_char(c["in"]) == c["out"]
        assert common.de_tokenized_by_CJK_char(c["in"], do_lower_case=True) == c["out_lower"]


def test_tokenizer_without_bpe_model_is_loud():
    tk = TextTokenizer("/nonexistent/bpe.model", None)
    with pytest.raises(RuntimeError, match="SentencePiece"):
        tk.tokenize("你好")
    assert tk.split_sentences(["▁A", "B", ".", "C"], 120) == [["▁A", "B", ".", "C"]]
    assert (tk.bos_token_id, tk.eos_token_id, tk.pad_token_id, tk.bos_token, tk.eos_token) == (0, 1, -1, "<s>", "</s>")


def test_saved_voice_roundtrip(tmp_path):
    mel = torch.randn(1, 100, 37)
    vid = voices.save_voice(str(tmp_path), "My Voice #1 (测试)", mel)
    assert vid == voices.sanitize_filename("My Voice #1 (测试)") == "My-Voice-1-测试"
    # the web UI's own reader: torch.from_numpy(np.load(<id>.cond_mel.npy)) (webui.py:311-313)
    raw = np.load(os.path.join(tmp_path, f"{vid}.cond_mel.npy"))
    assert raw.dtype == np.float32 and raw.shape == (1, 100, 37)
    assert torch.equal(voices.load_voice(str(tmp_path), "My Voice #1 (测试)"), mel)
    meta = voices.list_voices(str(tmp_path))
    assert meta == [{"id": vid, "user_given_name": "My Voice #1 (测试)"}]
    with pytest.raises(FileNotFoundError):
        voices.load_voice(str(tmp_path), "nobody")
    with pytest.raises(ValueError):
        voices.save_voice(str(tmp_path), "x", torch.zeros(100, 5))
