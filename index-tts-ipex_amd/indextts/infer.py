"""`IndexTTS` drop-in (/root/reference/indextts/infer.py:25-537) on the MI355X HIP engine.

Same constructor and methods (`extract_features`, `infer`, `infer_fast`, `set_gr_progress_callback`,
`remove_long_silence`, `bucket_sentences`, `pad_tokens_cat`), same attributes (`device`, `cfg`, `tokenizer`,
`gpt`, `bigvgan`, `stop_mel_token`).  Differences, all deliberate:
  * `infer` accepts this fork's `prompt_mel=<tensor [1,100,F]>` AND upstream's `audio_prompt=<wav path>` (the
    reference's own cli.py:70 and tests call the latter, which this fork's signature rejects);
  * conditioning latents and the ECAPA speaker vector are computed once per prompt and reused across sentences;
  * all sentences of a text are decoded as one batch (what `infer_fast` does per bucket);
  * greedy search and multinomial sampling (do_sample with top_k <= 64 / top_p / temperature / repetition_penalty,
    HF GenerationMixin.sample semantics) run on the device; beam search (num_beams > 1, the reference default of 3) is
    not implemented and decodes with num_beams = 1 after a RuntimeWarning.  Draws come from a numpy Generator seeded from
    torch's global RNG, so `torch.manual_seed` makes a run reproducible (the reference draws with torch.multinomial);
  * `is_fp16=True` selects the bf16 throughput engine, `False` the fp32 parity engine; `use_cuda_kernel` is accepted
    and ignored (the fused HIP activation is always used).
There is no CPU fallback: without a GPU / libitts_hip.so construction raises."""
from __future__ import annotations

import os
import time
import warnings
from typing import List

import numpy as np
import torch

from itts_hip import config as icfg
from itts_hip import engine as ieng
from itts_hip import infer_core, pack
from indextts.BigVGAN.models import BigVGAN as Generator
from indextts.gpt.model import UnifiedVoice
from indextts.utils.checkpoint import read_state_dict
from indextts.utils.feature_extractors import MelSpectrogramFeatures, load_wav_mono, resample
from indextts.utils.front import TextTokenizer


class IndexTTS:
    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", is_fp16=True, device=None,
                 use_cuda_kernel=None, state_dicts=None, cfg=None):
        if device is None:
            device = "cuda:0"
        if not str(device).startswith("cuda") or not torch.cuda.is_available():
            raise RuntimeError("IndexTTS (itts_hip): an MI355X GPU is required; there is no CPU fallback")
        self.device = str(device)
        self.is_fp16 = bool(is_fp16)
        self.use_cuda_kernel = True
        self.cfg = cfg if cfg is not None else icfg.load_yaml(cfg_path)
        self.model_dir = model_dir
        self.dtype = torch.bfloat16 if self.is_fp16 else torch.float32
        self.stop_mel_token = self.cfg.gpt.stop_mel_token
        sds = dict(state_dicts or {})
        if "gpt" not in sds:
            self.gpt_path = os.path.join(model_dir, self.cfg.gpt_checkpoint)
            sds["gpt"] = read_state_dict(self.gpt_path)
            print(">> GPT weights restored from:", self.gpt_path)
        if "bigvgan" not in sds:
            self.bigvgan_path = os.path.join(model_dir, self.cfg.bigvgan_checkpoint)
            sds["bigvgan"] = read_state_dict(self.bigvgan_path, key="generator")
            print(">> bigvgan weights restored from:", self.bigvgan_path)
        self.engine = ieng.Engine(self.cfg, "bf16" if self.is_fp16 else "fp32", self.device)
        self.engine.load_packed(pack.pack_gpt(sds["gpt"], self.cfg))
        self.engine.load_packed(pack.pack_bigvgan(sds["bigvgan"], self.cfg))
        if "dvae" in sds:
            self.engine.load_packed(pack.pack_dvae(sds["dvae"], self.cfg))
        self.engine.finalize()
        self.gpt = UnifiedVoice(self.engine, self.cfg.gpt)
        self.bigvgan = Generator(self.engine)
        self.bpe_path = os.path.join(model_dir, self.cfg.dataset["bpe_model"])
        self.normalizer = None
        self.tokenizer = TextTokenizer(self.bpe_path, self.normalizer)
        self.wav2mel = MelSpectrogramFeatures()
        self.gr_progress = None

    def set_gr_progress_callback(self, _callback):
        self.gr_progress = _callback

    def _set_gr_progress(self, value, desc):
        if self.gr_progress is not None:
            self.gr_progress(value, desc)

    def extract_features(self, audio_prompt_path: str) -> torch.Tensor:
        """wav path -> log-mel [1, 100, F] on the device (infer.py:82-93)."""
        audio, sr = load_wav_mono(audio_prompt_path)
        audio = resample(audio, sr, 24000)
        return self.wav2mel(audio).to(self.device)

    # ---- integer host logic (infer.py:244-318) ----
    def remove_long_silence(self, codes: torch.Tensor, silent_token=52, max_consecutive=30):
        c, n = infer_core.remove_long_silence(codes.detach().cpu().numpy(), self.stop_mel_token, silent_token, max_consecutive)
        return torch.from_numpy(c).to(codes.device), torch.from_numpy(n).to(codes.device)

    def bucket_sentences(self, sentences, bucket_max_size=4):
        return infer_core.bucket_sentences(sentences, bucket_max_size)

    def pad_tokens_cat(self, tokens: List[torch.Tensor]) -> torch.Tensor:
        arr = infer_core.pad_tokens_cat([t.detach().cpu().numpy() for t in tokens], self.cfg.gpt.stop_text_token)
        return torch.from_numpy(arr).to(self.device)

    def torch_empty_cache(self):
        torch.cuda.empty_cache()

    # ---- synthesis ----
    def _sentences_to_ids(self, text, max_tokens):
        if isinstance(text, str):
            toks = self.tokenizer.tokenize(text)
            sents = self.tokenizer.split_sentences(toks, max_tokens)
            return [np.asarray(self.tokenizer.convert_tokens_to_ids(s), dtype=np.int32) for s in sents]
        return [np.asarray(s, dtype=np.int32).reshape(-1) for s in text]  # pre-tokenised: list of id lists

    def _synthesize(self, prompt_mel, text, output_path, max_text_tokens_per_sentence, bucket, verbose, kw):
        start = time.perf_counter()
        do_sample = kw.pop("do_sample", True)
        num_beams = kw.pop("num_beams", 3)
        top_p, top_k, temperature = kw.pop("top_p", 0.8), kw.pop("top_k", 30), kw.pop("temperature", 1.0)
        kw.pop("length_penalty", None)
        rep = kw.pop("repetition_penalty", 10.0)
        max_mel_tokens = kw.pop("max_mel_tokens", 600)
        sample_kw = infer_core.sampling_kwargs(do_sample, num_beams, top_k, top_p, temperature)
        sents = self._sentences_to_ids(text, max_text_tokens_per_sentence)
        self._set_gr_progress(0.1, "text processing...")
        cond = self.gpt.get_conditioning(prompt_mel)
        spk = self.engine.ecapa(prompt_mel.transpose(1, 2))
        t_gen = t_fwd = t_voc = 0.0
        buckets = infer_core.bucket_sentences(sents, bucket)
        wav_by_idx = {}
        for bi, bk in enumerate(buckets):
            t0 = time.perf_counter()
            ids = infer_core.pad_tokens_cat([x["sent"] for x in bk], self.cfg.gpt.stop_text_token)
            codes = self.engine.generate(cond, ids, max_mel_tokens, repetition_penalty=rep, **sample_kw)
            t_gen += time.perf_counter() - t0
            if (codes[:, -1] != self.stop_mel_token).any():
                warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}).", RuntimeWarning)
            # latent pass for the whole bucket in one launch sequence (left-padded, masked: each sentence gets exactly its
            # batch-1 latent), then the vocoder per sentence (code lengths differ after remove_long_silence)
            clean = []
            for r in range(len(bk)):
                c, n = infer_core.remove_long_silence(codes[r:r + 1], self.stop_mel_token)
                clean.append(c[0, : int(n[0])])
            t0 = time.perf_counter()
            lats = self.engine.latent_batch(cond, [item["sent"] for item in bk], clean)
            t_fwd += time.perf_counter() - t0
            t0 = time.perf_counter()
            for lat, item in zip(lats, bk):
                wav = self.engine.bigvgan(lat, spk)
                wav_by_idx[item["idx"]] = torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0).cpu()
            t_voc += time.perf_counter() - t0
            self._set_gr_progress(0.2 + 0.7 * (bi + 1) / len(buckets), f"synthesis {bi + 1}/{len(buckets)}")
        wav = torch.cat([wav_by_idx[i] for i in range(len(sents))], dim=1)
        total = time.perf_counter() - start
        wav_len = wav.shape[-1] / 24000
        if verbose:
            print(f">> gpt_gen_time: {t_gen:.2f}s  gpt_forward_time: {t_fwd:.2f}s  bigvgan_time: {t_voc:.2f}s")
        print(f">> Total inference time: {total:.2f} seconds; generated audio: {wav_len:.2f} s; RTF: {total / max(wav_len, 1e-9):.4f}")
        wav16 = wav.type(torch.int16)
        if output_path:
            from scipy.io import wavfile

            if os.path.dirname(output_path):
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            wavfile.write(output_path, 24000, wav16.numpy().T)
            return output_path
        return (24000, wav16.numpy().T)

    def _prompt(self, prompt_mel, audio_prompt):
        if prompt_mel is None and audio_prompt is None:
            raise TypeError("infer() needs prompt_mel=<tensor> or audio_prompt=<wav path>")
        if prompt_mel is None:
            prompt_mel = self.extract_features(audio_prompt)
        if isinstance(prompt_mel, str):
            prompt_mel = self.extract_features(prompt_mel)
        return prompt_mel.to(self.device)

    def infer(self, prompt_mel=None, text=None, output_path=None, max_text_tokens_per_sentence=120, verbose=False,
              audio_prompt=None, **generation_kwargs):
        return self._synthesize(self._prompt(prompt_mel, audio_prompt), text, output_path, max_text_tokens_per_sentence,
                                10 ** 9, verbose, generation_kwargs)

    def infer_fast(self, prompt_mel=None, text=None, output_path=None, max_text_tokens_per_sentence=120, verbose=False,
                   sentences_bucket_max_size=4, audio_prompt=None, **generation_kwargs):
        return self._synthesize(self._prompt(prompt_mel, audio_prompt), text, output_path, max_text_tokens_per_sentence,
                                sentences_bucket_max_size, verbose, generation_kwargs)
