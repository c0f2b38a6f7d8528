"""`IndexTTS` drop-in (/root/reference/indextts/infer.py:25-537) on the MI355X HIP engine.

Same constructor and methods (`extract_features`, `infer`, `infer_fast`, `set_gr_progress_callback`,
`remove_long_silence`, `bucket_sentences`, `pad_tokens_cat`), same attributes (`device`, `cfg`, `tokenizer`,
`gpt`, `bigvgan`, `stop_mel_token`).  Differences, all deliberate:
  * `infer` accepts this fork's `prompt_mel=<tensor [1,100,F]>` AND upstream's `audio_prompt=<wav path>` (the
    reference's own cli.py:70 and tests call the latter, which this fork's signature rejects);
  * conditioning latents and the ECAPA speaker vector are computed once per prompt and reused across sentences;
  * `infer` decodes the sentences of a text in batches of up to max_batch rows (the reference loops at batch 1; rows of a
    batch are independent), `infer_fast` in the reference's length-sorted buckets; both vocode as the reference does
    (`infer` per sentence, `infer_fast` over time-concatenated chunks of 2 latents); `infer_batch` takes several utterances;
  * every public call takes the engine lock: several host threads may share one IndexTTS (webui.py:441-452);
  * generation runs on the device for every mode the kwargs of infer.py:116-124 select: beam-sample (the default:
    do_sample=True, num_beams=3; up to 10 beams, per-beam cache ancestry instead of HF's per-step cache copy), beam search
    (do_sample=False, num_beams > 1), multinomial sampling (num_beams=1) and greedy - with top_k <= 128 / top_p /
    temperature / repetition_penalty / length_penalty and `typical_sampling` (TypicalLogitsWarper) in HF 4.36.2 semantics;
    `top_k = 0 / None` (HF: TopK warper off) or > 128 is exact too, with one beam or several - the warpers and draws then run on
    the host over the whole vocabulary, one logits read-back per token (under beams BeamSearchScorer.process, the beam
    re-ordering and finalize stay on the device).
    Draws come from a numpy Generator seeded from torch's global RNG, so `torch.manual_seed` makes a run reproducible
    (torch.multinomial's own stream cannot be reproduced on a device);
  * the text normaliser is built and loaded as in infer.py:69-71; its third-party written-form normalisers (`tn` /
    `wetext`) are used when installed and skipped with a RuntimeWarning otherwise (punctuation folding, pinyin and name
    protection always apply);
  * `is_fp16=True` selects the IEEE-half engine (the reference's own GPU precision; ITTS_HALF=bf16: the bfloat16 engine, same
    speed), `False` the fp32 parity engine; `use_cuda_kernel` is accepted
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
from indextts.utils.front import TextNormalizer, TextTokenizer


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
        # is_fp16=True is IEEE half in the reference (autocast(dtype=torch.float16) + .half(), infer.py:39,44,52): the f16 build of the
        # library.  ITTS_HALF=bf16 selects the bfloat16 engine instead (same speed, 8 instead of 11 significand bits, wider range)
        self.half = os.environ.get("ITTS_HALF", "f16") if self.is_fp16 else None
        if self.half not in (None, "f16", "bf16"):
            raise ValueError(f"ITTS_HALF={self.half}: expected f16 or bf16")
        self.dtype = (torch.float16 if self.half == "f16" else torch.bfloat16) if self.is_fp16 else torch.float32
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
        self.engine = ieng.Engine(self.cfg, self.half if self.is_fp16 else "fp32", self.device)
        self.engine.load_packed(pack.pack_gpt(sds["gpt"], self.cfg))
        self.engine.load_packed(pack.pack_bigvgan(sds["bigvgan"], self.cfg))
        if "dvae" in sds:
            self.engine.load_packed(pack.pack_dvae(sds["dvae"], self.cfg))
        self.engine.finalize()
        self.gpt = UnifiedVoice(self.engine, self.cfg.gpt)
        self.bigvgan = Generator(self.engine)
        self.bpe_path = os.path.join(model_dir, self.cfg.dataset["bpe_model"])
        self.normalizer = TextNormalizer()  # infer.py:69-71
        self.normalizer.load()
        self.tokenizer = TextTokenizer(self.bpe_path, self.normalizer)
        print(">> bpe model loaded from:", self.bpe_path)
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

    def _gen_kwargs(self, kw):
        """The generation kwargs `infer` / `infer_fast` pop (infer.py:116-124), with the reference's defaults."""
        kw = dict(kw)
        g = dict(do_sample=kw.pop("do_sample", True), top_p=kw.pop("top_p", 0.8), top_k=kw.pop("top_k", 30),
                 temperature=kw.pop("temperature", 1.0), length_penalty=kw.pop("length_penalty", 0.0),
                 num_beams=kw.pop("num_beams", 3), repetition_penalty=kw.pop("repetition_penalty", 10.0),
                 max_mel_tokens=kw.pop("max_mel_tokens", 600))
        g["typical_sampling"] = kw.pop("typical_sampling", False)
        g["typical_mass"] = kw.pop("typical_mass", 0.9)
        return g

    def _generate_rows(self, cond, sents, g):
        """Mel codes for a list of sentences: decode batches of at most `engine max_batch` rows (HF pads / stops per
        batch, so a group is exactly one `inference_speech` call of the reference)."""
        sample_kw = infer_core.sampling_kwargs(g["do_sample"], g["num_beams"], g["top_k"], g["top_p"], g["temperature"],
                                               g["typical_sampling"], g["typical_mass"], g["length_penalty"])
        cap = max(1, self.engine.ccfg.max_batch // max(1, sample_kw.get("num_beams", 1)))
        rows = []
        for lo in range(0, len(sents), cap):
            grp = sents[lo:lo + cap]
            ids = infer_core.pad_tokens_cat(grp, self.cfg.gpt.stop_text_token)
            codes = self.engine.generate(cond, ids, g["max_mel_tokens"], repetition_penalty=g["repetition_penalty"], **sample_kw)
            rows.extend(codes[r] for r in range(len(grp)))
        return rows

    def _clean_and_latents(self, cond, sents, code_rows, max_mel_tokens):
        """remove_long_silence per sentence (infer.py:190 / :463), then the latent pass (batch-1 semantics per sentence,
        stacked as masked left-padded rows in one launch sequence)."""
        if any(r[-1] != self.stop_mel_token for r in code_rows):
            warnings.warn(f"WARN: generation stopped due to exceeding `max_mel_tokens` ({max_mel_tokens}). "
                          "Consider reducing `max_text_tokens_per_sentence` or increasing `max_mel_tokens`.", RuntimeWarning)
        clean = []
        for r in code_rows:
            c, n = infer_core.remove_long_silence(np.asarray(r)[None], self.stop_mel_token)
            clean.append(c[0, : int(n[0])])
        lats = []
        cap = self.engine.ccfg.max_batch
        for lo in range(0, len(sents), cap):
            lats.extend(self.engine.latent_batch(cond, sents[lo:lo + cap], clean[lo:lo + cap]))
        return lats

    def _finish(self, wavs, output_path, start, timers, verbose, tag=""):
        wav = torch.cat(wavs, dim=1)
        total = time.perf_counter() - start
        wav_len = wav.shape[-1] / 24000
        if verbose:
            print(f">> gpt_gen_time: {timers[0]:.2f}s  gpt_forward_time: {timers[1]:.2f}s  bigvgan_time: {timers[2]:.2f}s")
        print(f">> Total {tag}inference time: {total:.2f} seconds; generated audio: {wav_len:.2f} s; RTF: {total / max(wav_len, 1e-9):.4f}")
        wav16 = wav.type(torch.int16)
        if output_path:
            from scipy.io import wavfile

            if os.path.isfile(output_path):
                os.remove(output_path)
            if os.path.dirname(output_path):
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            wavfile.write(output_path, 24000, wav16.numpy().T)
            return output_path
        return (24000, wav16.numpy().T)

    def _to_int16_range(self, wav):
        return torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0).cpu()  # infer.py:208-212

    def _synthesize(self, prompt_mel, text, output_path, max_text_tokens_per_sentence, bucket, verbose, kw, fast):
        """`infer` (fast=False: vocoder per sentence, infer.py:134-212) and `infer_fast` (fast=True: length-sorted buckets
        for the AR decode, vocoder over time-concatenated chunks of 2 latents, infer.py:385-498)."""
        start = time.perf_counter()
        g = self._gen_kwargs(kw)
        sents = self._sentences_to_ids(text, max_text_tokens_per_sentence)
        if not sents:
            # the reference ends in torch.cat([]) -> RuntimeError; same class, clearer message
            raise RuntimeError("IndexTTS: the text produced no sentences (empty input)")
        self._set_gr_progress(0.1, "text processing...")
        with self.engine.lock:
            # the speaker embedding (needed by the vocoder only) on the engine's side stream, beside the conditioning encoder and the prefill
            spk = self.engine.ecapa(prompt_mel.transpose(1, 2), overlap=True)
            cond = self.gpt.get_conditioning(prompt_mel)
            t_gen = t_fwd = t_voc = 0.0
            t0 = time.perf_counter()
            code_rows = [None] * len(sents)
            if fast:
                for bk in infer_core.bucket_sentences(sents, bucket):
                    for item, row in zip(bk, self._generate_rows(cond, [x["sent"] for x in bk], g)):
                        code_rows[item["idx"]] = row
            else:
                code_rows = self._generate_rows(cond, sents, g)
            t_gen = time.perf_counter() - t0
            self._set_gr_progress(0.5, "gpt inference latents...")
            t0 = time.perf_counter()
            lats = self._clean_and_latents(cond, sents, code_rows, g["max_mel_tokens"])  # original sentence order
            t_fwd = time.perf_counter() - t0
            self._set_gr_progress(0.7, "bigvgan decode...")
            t0 = time.perf_counter()
            wavs = []
            if fast:
                chunk = 2  # infer.py:480: BigVGAN runs over pairs of sentences concatenated along time
                for lo in range(0, len(lats), chunk):
                    wavs.append(self._to_int16_range(self.engine.bigvgan(torch.cat(lats[lo:lo + chunk], dim=1), spk)))
            else:
                for lat in lats:
                    wavs.append(self._to_int16_range(self.engine.bigvgan(lat, spk)))
            t_voc = time.perf_counter() - t0
        self._set_gr_progress(0.9, "save audio...")
        return self._finish(wavs, output_path, start, (t_gen, t_fwd, t_voc), verbose, "fast " if fast else "")

    @torch.no_grad()
    def infer_batch(self, prompt_mels, texts, output_paths=None, max_text_tokens_per_sentence=120, verbose=False,
                    **generation_kwargs):
        """Several utterances in one call (the multi-utterance / data-parallel entry point): the sentences of
        all utterances that share a prompt are decoded together in length-sorted batches of up to `max_batch` rows,
        the latent pass is stacked, the vocoder runs per sentence.  Under torch.distributed (one process per GPU) each
        rank should pass its own shard - see itts_hip.dp.run_sharded.  prompt_mels: one tensor for all utterances or a
        list with one per utterance.  Returns a list of (24000, int16 [n, 1]) or of written paths."""
        n = len(texts)
        prompts = [prompt_mels] * n if isinstance(prompt_mels, torch.Tensor) else list(prompt_mels)
        assert len(prompts) == n and (output_paths is None or len(output_paths) == n)
        g = self._gen_kwargs(generation_kwargs)
        start = time.perf_counter()
        utt_sents = [self._sentences_to_ids(t, max_text_tokens_per_sentence) for t in texts]
        results = [None] * n
        groups = {}
        for u, p in enumerate(prompts):
            groups.setdefault(id(p), []).append(u)
        with self.engine.lock:
            for us in groups.values():
                mel = self._prompt(prompts[us[0]], None)
                spk = self.engine.ecapa(mel.transpose(1, 2), overlap=True)
                cond = self.gpt.get_conditioning(mel)
                flat = [(u, k, s) for u in us for k, s in enumerate(utt_sents[u])]
                if not flat:
                    continue
                order = sorted(range(len(flat)), key=lambda i: (len(flat[i][2]), i))  # length-sorted decode batches
                rows = self._generate_rows(cond, [flat[i][2] for i in order], g)
                code_rows = [None] * len(flat)
                for i, r in zip(order, rows):
                    code_rows[i] = r
                lats = self._clean_and_latents(cond, [f[2] for f in flat], code_rows, g["max_mel_tokens"])
                wav_parts = {u: [] for u in us}
                for (u, k, _), wav in zip(flat, self.engine.bigvgan_grouped(lats, spk)):  # equal lengths share a launch
                    wav_parts[u].append(self._to_int16_range(wav))
                for u in us:
                    if wav_parts[u]:
                        results[u] = wav_parts[u]
        out = []
        for u in range(n):
            if results[u] is None:
                raise RuntimeError(f"IndexTTS.infer_batch: utterance {u} produced no sentences (empty input)")
            out.append(self._finish(results[u], output_paths[u] if output_paths else None, start, (0, 0, 0), False, "batch "))
        return out

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
                                10 ** 9, verbose, generation_kwargs, fast=False)

    def infer_fast(self, prompt_mel=None, text=None, output_path=None, max_text_tokens_per_sentence=120, verbose=False,
                   sentences_bucket_max_size=4, audio_prompt=None, **generation_kwargs):
        return self._synthesize(self._prompt(prompt_mel, audio_prompt), text, output_path, max_text_tokens_per_sentence,
                                sentences_bucket_max_size, verbose, generation_kwargs, fast=True)
