"""Text front end of the drop-in: `TextNormalizer` and `TextTokenizer` with the reference's interface
(/root/reference/indextts/utils/front.py:11-428) - host-side string work in front of the hot path.

Behaviour mirrored, and pinned by tests/golden/front_cases.json (known answers produced by the reference's own functions):
  * punctuation folding tables, protection of pinyin-with-tone (`xuan4`) and of dotted person names across the
    third-party number/date normaliser, `jqx + u/ue -> v` pinyin correction, "'s -> is" expansion, the
    Chinese-vs-English routing (`use_chinese`);
  * CJK pre-tokenisation (every CJK character its own word, Latin upper-cased) in front of SentencePiece;
  * sentence splitting on token lists: cut after sentence punctuation, fall back to commas, hyphens and hard cuts when a
    sentence exceeds the token cap, then merge short neighbours (front.py:344-428).
Third-party pieces stay third-party: the zh/en written-form normalisers (`tn` = WeTextProcessing, `wetext` on macOS) are
imported when installed and skipped with a RuntimeWarning otherwise; the BPE model is SentencePiece's `bpe.model`."""
from __future__ import annotations

import os
import re
import warnings
from typing import List, Optional, Sequence, Tuple, Union

from indextts.utils.common import de_tokenized_by_CJK_char, tokenize_by_CJK_char

# punctuation folded to the small set the acoustic model was trained on (front.py:15-53)
_PUNCT = {
    "：": ",", "；": ",", ";": ",", "，": ",", "。": ".", "！": "!", "？": "?", "\n": " ", "·": "-", "、": ",", "...": "…",
    ",,,": "…", "，，，": "…", "……": "…", "“": "'", "”": "'", '"': "'", "‘": "'", "’": "'", "（": "'", "）": "'", "(": "'",
    ")": "'", "《": "'", "》": "'", "【": "'", "】": "'", "[": "'", "]": "'", "—": "-", "～": "-", "~": "-", "「": "'",
    "」": "'", ":": ",",
}


def _fold(text: str, table: dict) -> str:
    """One left-to-right pass replacing every key of `table` (first alternative that matches wins, like re's `|`)."""
    rx = re.compile("|".join(re.escape(k) for k in table))
    return rx.sub(lambda m: table[m.group()], text)


class TextNormalizer:
    # pinyin syllable + tone digit 1-5 (5 = neutral), not preceded by a letter: xuan4, jve2, ying1 - but not beta1, voice2
    PINYIN_TONE_PATTERN = (r"(?<![a-z])((?:[bpmfdtnlgkhjqxzcsryw]|[zcs]h)?(?:[aeiouüv]|[ae]i|u[aio]|ao|ou|i[aue]|[uüv]e|[uvü]ang?|uai|"
                           r"[aeiuv]n|[aeio]ng|ia[no]|i[ao]ng)|ng|er)([1-5])")
    # transliterated person names: 克里斯托弗·诺兰, 约瑟夫·高登-莱维特
    NAME_PATTERN = r"[一-鿿]+(?:[-·—][一-鿿]+){1,2}"
    ENGLISH_CONTRACTION_PATTERN = r"(what|where|who|which|how|t?here|it|s?he|that|this)'s"

    def __init__(self):
        self.zh_normalizer = None
        self.en_normalizer = None
        self.char_rep_map = dict(_PUNCT)
        self.zh_char_rep_map = {"$": ".", **self.char_rep_map}
        self._loaded = False

    # ---- routing ----
    def match_email(self, email: str) -> bool:
        return re.match(r"^[a-zA-Z0-9]+@[a-zA-Z0-9]+\.[a-zA-Z]+$", email) is not None

    def use_chinese(self, s: str) -> bool:
        if re.search(r"[一-鿿]", s) or not re.search(r"[a-zA-Z]", s) or self.match_email(s):
            return True
        return re.search(self.PINYIN_TONE_PATTERN, s, re.IGNORECASE) is not None

    # ---- third-party written-form normalisers ----
    def load(self):
        if self._loaded:
            return
        self._loaded = True
        try:
            import platform

            if platform.system() == "Darwin":
                from wetext import Normalizer

                self.zh_normalizer = Normalizer(remove_erhua=False, lang="zh", operator="tn")
                self.en_normalizer = Normalizer(lang="en", operator="tn")
            else:
                from tn.chinese.normalizer import Normalizer as Zh
                from tn.english.normalizer import Normalizer as En

                cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tagger_cache")
                os.makedirs(cache, exist_ok=True)
                self.zh_normalizer = Zh(cache_dir=cache, remove_interjections=False, remove_erhua=False, overwrite_cache=False)
                self.en_normalizer = En(overwrite_cache=False)
        except ImportError:
            warnings.warn("TextNormalizer: WeTextProcessing (`tn`) / `wetext` is not installed - numbers, dates and units are "
                          "passed through as written; punctuation folding, pinyin and name handling still apply", RuntimeWarning)

    # ---- protect / restore spans across the third-party normaliser ----
    @staticmethod
    def _protect(text: str, pattern: str, tag: str) -> Tuple[str, Optional[List[str]]]:
        found = re.findall(re.compile(pattern, re.IGNORECASE), text)
        if not found:
            return text, None
        spans = list(set("".join(f) for f in found))
        for i, sp in enumerate(spans):
            text = text.replace(sp, f"<{tag}_{chr(ord('a') + i)}>")
        return text, spans

    @staticmethod
    def _restore(text: str, spans: Optional[List[str]], tag: str, fix=None) -> str:
        for i, sp in enumerate(spans or []):
            text = text.replace(f"<{tag}_{chr(ord('a') + i)}>", fix(sp) if fix else sp)
        return text

    def save_names(self, original_text: str):
        return self._protect(original_text, self.NAME_PATTERN, "n")

    def restore_names(self, normalized_text: str, original_name_list):
        return self._restore(normalized_text, original_name_list, "n")

    def save_pinyin_tones(self, original_text: str):
        return self._protect(original_text, self.PINYIN_TONE_PATTERN, "pinyin")

    def restore_pinyin_tones(self, normalized_text: str, original_pinyin_list):
        return self._restore(normalized_text, original_pinyin_list, "pinyin", self.correct_pinyin)

    def correct_pinyin(self, pinyin: str) -> str:
        """ju -> JV, que -> QVE, xün -> XVN (the vocabulary spells j/q/x + ü with v); everything else is left alone."""
        if pinyin[0] not in "jqxJQX":
            return pinyin
        return re.sub(r"([jqx])[uü](n|e|an)*(\d)", r"\g<1>v\g<2>\g<3>", pinyin, flags=re.IGNORECASE).upper()

    def normalize(self, text: str) -> str:
        if not self._loaded:
            self.load()
        text = re.sub(self.ENGLISH_CONTRACTION_PATTERN, r"\1 is", text, flags=re.IGNORECASE)
        if self.use_chinese(text):
            body, pinyins = self.save_pinyin_tones(text.rstrip())
            body, names = self.save_names(body)
            if self.zh_normalizer is not None:
                try:
                    body = self.zh_normalizer.normalize(body)
                except Exception:  # noqa: BLE001 - the reference prints the traceback and continues with ""
                    import traceback

                    print(traceback.format_exc())
                    body = ""
            body = self.restore_pinyin_tones(self.restore_names(body, names), pinyins)
            return _fold(body, self.zh_char_rep_map)
        if self.en_normalizer is not None:
            try:
                text = self.en_normalizer.normalize(text)
            except Exception:  # noqa: BLE001
                import traceback

                print(traceback.format_exc())
        return _fold(text, self.char_rep_map)


class TextTokenizer:
    punctuation_marks_tokens = [".", "!", "?", "▁.", "▁?", "▁..."]

    def __init__(self, vocab_file: str, normalizer: TextNormalizer = None):
        self.vocab_file = vocab_file
        self.normalizer = normalizer
        self.sp_model = None
        if vocab_file and os.path.exists(vocab_file):
            from sentencepiece import SentencePieceProcessor

            self.sp_model = SentencePieceProcessor(model_file=vocab_file)
        if self.normalizer:
            self.normalizer.load()
        self.pre_tokenizers = [tokenize_by_CJK_char]

    def _sp(self):
        if self.sp_model is None:
            raise RuntimeError(f"no SentencePiece model at {self.vocab_file!r}: pass pre-tokenised ids (list of int lists) "
                               "instead of a string")
        return self.sp_model

    # ---- vocabulary / special tokens (front.py:249-300) ----
    vocab_size = property(lambda self: self._sp().GetPieceSize())
    unk_token = property(lambda self: "<unk>")
    pad_token = property(lambda self: None)
    bos_token = property(lambda self: "<s>")
    eos_token = property(lambda self: "</s>")
    pad_token_id = property(lambda self: -1)
    bos_token_id = property(lambda self: 0)
    eos_token_id = property(lambda self: 1)
    unk_token_id = property(lambda self: self._sp().unk_id())

    @property
    def special_tokens_map(self):
        return {"unk_token": self.unk_token, "pad_token": self.pad_token, "bos_token": self.bos_token, "eos_token": self.eos_token}

    def get_vocab(self):
        return {self.convert_ids_to_tokens(i): i for i in range(self.vocab_size)}

    def convert_ids_to_tokens(self, ids: Union[List[int], int]):
        return self._sp().IdToPiece(ids)

    def convert_tokens_to_ids(self, tokens: Union[List[str], str]) -> List[int]:
        return [self._sp().PieceToId(t) for t in ([tokens] if isinstance(tokens, str) else tokens)]

    # ---- text -> pieces ----
    def _prepare(self, text: str) -> str:
        if self.normalizer:
            text = self.normalizer.normalize(text)
        for pre in self.pre_tokenizers:
            text = pre(text)
        return text

    def tokenize(self, text: str) -> List[str]:
        return self.encode(text, out_type=str)

    def encode(self, text: str, **kwargs):
        if len(text) == 0:
            return []
        out_type = kwargs.pop("out_type", int)
        if len(text.strip()) == 1:  # a lone character skips normalisation and pre-tokenisation (front.py:312-313)
            return self._sp().Encode(text, out_type=out_type, **kwargs)
        return self._sp().Encode(self._prepare(text), out_type=out_type, **kwargs)

    def batch_encode(self, texts: List[str], **kwargs):
        return self._sp().Encode([self._prepare(t) for t in texts], out_type=kwargs.pop("out_type", int), **kwargs)

    def decode(self, ids: Union[List[int], int], do_lower_case=False, **kwargs):
        ids = [ids] if isinstance(ids, int) else ids
        return de_tokenized_by_CJK_char(self._sp().Decode(ids, out_type=kwargs.pop("out_type", str), **kwargs), do_lower_case=do_lower_case)

    # ---- sentence splitting on pieces (front.py:344-428) ----
    @staticmethod
    def split_sentences_by_token(tokenized_str: Sequence[str], split_tokens: Sequence[str], max_tokens_per_sentence: int) -> List[List[str]]:
        toks = list(tokenized_str)
        if not toks:
            return []
        cap = max_tokens_per_sentence
        done: List[List[str]] = []
        cur: List[str] = []
        for i, tok in enumerate(toks):
            cur.append(tok)
            quote_next = i + 1 < len(toks) and toks[i + 1] in ("'", "▁'")  # never cut in front of a closing quote
            if len(cur) <= cap and tok in split_tokens and len(cur) > 2 and not quote_next:
                done.append(cur)
                cur = []
            elif len(cur) > cap:
                # the sentence outgrew the cap before its punctuation: retry on commas, then hyphens, else cut blindly
                if not any(t in split_tokens for t in (",", "▁,")) and any(t in cur for t in (",", "▁,")):
                    done.extend(TextTokenizer.split_sentences_by_token(cur, [",", "▁,"], cap))
                elif "-" not in split_tokens and "-" in cur:
                    done.extend(TextTokenizer.split_sentences_by_token(cur, ["-"], cap))
                else:
                    warnings.warn(f"[WARNING] Sentence token length exceeds max ({cap}): {cur}", RuntimeWarning)
                    done.extend(cur[j:j + cap] for j in range(0, len(cur), cap))
                cur = []
        if cur:
            done.append(cur)
        return TextTokenizer._merge_short_sentences(done, cap)

    @staticmethod
    def _merge_short_sentences(sentences: List[List[str]], max_len: int) -> List[List[str]]:
        merged: List[List[str]] = []
        for s in sentences:
            if merged and len(merged[-1]) + len(s) <= max_len:
                merged[-1] = merged[-1] + list(s)
            else:
                merged.append(list(s))
        return merged

    def split_sentences(self, tokenized: Sequence[str], max_tokens_per_sentence: int = 120) -> List[List[str]]:
        return TextTokenizer.split_sentences_by_token(tokenized, self.punctuation_marks_tokens, max_tokens_per_sentence)
