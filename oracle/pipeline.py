"""ORACLE (test infrastructure, not product): CPU restatement of the `IndexTTS.infer` orchestration pieces
that are arithmetic (integer code clean-up, per-sentence pipeline over token ids), /root/reference
indextts/infer.py.  See oracle/gpt.py header for the usage rules.
"""
from __future__ import annotations

import time
from typing import List, Tuple

import torch

from . import gpt as ogpt
from . import vocoder as ovoc


def remove_long_silence(codes: torch.Tensor, stop_mel_token: int, silent_token: int = 52,
                        max_consecutive: int = 30) -> Tuple[torch.Tensor, torch.Tensor]:
    """infer.py:244-298.  codes [B,T] int -> (codes', code_lens).  Cut each row at the first stop token;
    if the row holds more than `max_consecutive` silent tokens IN TOTAL (:262-263), keep at most 10
    consecutive ones; pad ragged rows with stop (:287); clip to the max kept length (:294-296)."""
    code_lens: List[int] = []
    rows: List[torch.Tensor] = []
    isfix = False
    for i in range(codes.shape[0]):
        code = codes[i]
        stop_idx = (code == stop_mel_token).nonzero(as_tuple=False)
        len_ = int(stop_idx[0].item()) if len(stop_idx) > 0 else code.shape[0]
        count = int((code == silent_token).sum().item())
        if count > max_consecutive:
            keep, n = [], 0
            for k in range(len_):
                if int(code[k]) != silent_token:
                    keep.append(k)
                    n = 0
                elif n < 10:
                    keep.append(k)
                    n += 1
            len_ = len(keep)
            rows.append(code[keep])
            isfix = True
        else:
            rows.append(code[:len_])
        code_lens.append(len_)
    if isfix:
        if len(rows) > 1:
            codes = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=stop_mel_token)
        else:
            codes = rows[0].unsqueeze(0)
    max_len = max(code_lens)
    if max_len < codes.shape[1]:
        codes = codes[:, :max_len]
    return codes, torch.tensor(code_lens, dtype=torch.long)


def infer_sentence(prompt_mel, text_tokens, wg, wb, cfg, max_mel_tokens: int = 600, suppress_eos: bool = False,
                   timers: dict = None):
    """One iteration of the `for sent in sentences` loop of infer.py:134-212 under greedy kwargs
    (tests/padding_test.py:35-46): codes -> silence fix -> latent -> BigVGAN -> clamp(32767*wav)."""
    from importlib import import_module

    ecapa = import_module("itts_hip.config").ecapa_dims(cfg["bigvgan"])
    g = cfg["gpt"]
    t0 = time.perf_counter()
    cond = ogpt.get_conditioning(prompt_mel, wg, g)
    codes = ogpt.greedy_generate(cond, text_tokens, wg, g, max_mel_tokens, suppress_eos=suppress_eos)
    t1 = time.perf_counter()
    codes, code_lens = remove_long_silence(codes, g["stop_mel_token"])
    cond2 = ogpt.get_conditioning(prompt_mel, wg, g)  # the reference recomputes it (model.py:540)
    latent = ogpt.latent_forward(cond2, text_tokens, codes, wg, g)
    t2 = time.perf_counter()
    wav = ovoc.bigvgan_forward(latent, prompt_mel.transpose(1, 2), wb, cfg["bigvgan"], ecapa)
    t3 = time.perf_counter()
    if timers is not None:
        timers["gpt_gen"] = timers.get("gpt_gen", 0.0) + t1 - t0
        timers["gpt_forward"] = timers.get("gpt_forward", 0.0) + t2 - t1
        timers["bigvgan"] = timers.get("bigvgan", 0.0) + t3 - t2
    wav = torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0)
    return codes, latent, wav


def infer_fast_sentences(prompt_mel, sentences: List[torch.Tensor], wg, wb, cfg, max_mel_tokens: int = 600,
                         bucket_max_size: int = 4):
    """`infer_fast` (infer.py:332-537) under greedy kwargs over pre-tokenised sentences: length-sorted buckets
    (:303-315), one batched decode per bucket with stop-padded ids (:316-318,412-439), per sentence silence fix +
    latent at batch 1 (:446-477), original order restored (:481), BigVGAN over chunks of 2 latents concatenated along
    time (:480-498), clamp(32767 * wav).  Returns (codes per sentence, wav [1, n])."""
    from importlib import import_module

    ecapa = import_module("itts_hip.config").ecapa_dims(cfg["bigvgan"])
    g = cfg["gpt"]
    items = [{"idx": i, "sent": s, "len": int(s.numel())} for i, s in enumerate(sentences)]
    if len(items) <= bucket_max_size:
        buckets = [items]
    else:
        buckets = []
        for it in sorted(items, key=lambda x: x["len"]):
            if not buckets or len(buckets[-1]) >= bucket_max_size:
                buckets.append([it])
            else:
                buckets[-1].append(it)
    cond = ogpt.get_conditioning(prompt_mel, wg, g)
    codes_by_idx, lat_by_idx = {}, {}
    for bk in buckets:
        rows = [x["sent"].reshape(-1) for x in bk]
        batch = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=g["stop_text_token"])
        codes = ogpt.greedy_generate(cond, batch, wg, g, max_mel_tokens)
        for i, x in enumerate(bk):
            c, _ = remove_long_silence(codes[i:i + 1], g["stop_mel_token"])
            codes_by_idx[x["idx"]] = codes[i]
            lat_by_idx[x["idx"]] = ogpt.latent_forward(cond, rows[i].view(1, -1), c, wg, g)
    lats = [lat_by_idx[i] for i in range(len(sentences))]
    wavs = []
    for lo in range(0, len(lats), 2):
        wav = ovoc.bigvgan_forward(torch.cat(lats[lo:lo + 2], dim=1), prompt_mel.transpose(1, 2), wb, cfg["bigvgan"], ecapa)
        wavs.append(torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0))
    return [codes_by_idx[i] for i in range(len(sentences))], torch.cat(wavs, dim=1)
