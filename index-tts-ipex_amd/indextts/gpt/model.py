"""`UnifiedVoice` drop-in (/root/reference/indextts/gpt/model.py:300-708) on the HIP engine.

Implements the inference surface `infer.py` and `tests/padding_test.py` use: `get_conditioning`,
`inference_speech(...)`, `forward(..., return_latent=True)`, `post_init_gpt2_config`."""
from __future__ import annotations

import warnings

import numpy as np
import torch

from itts_hip import infer_core


class UnifiedVoice:
    def __init__(self, engine, gpt_cfg):
        self._eng = engine
        for k in ("stop_mel_token", "start_mel_token", "start_text_token", "stop_text_token", "mel_length_compression",
                  "max_mel_tokens", "max_text_tokens", "model_dim", "layers", "heads", "number_mel_codes",
                  "number_text_tokens"):
            setattr(self, k, gpt_cfg[k])
        self.cond_num = gpt_cfg.get("condition_num_latent", 32)
        self._cond_key, self._cond = None, None

    # no-ops kept for call compatibility (infer.py:50-52,59)
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def half(self):
        return self

    def post_init_gpt2_config(self, use_deepspeed=False, kv_cache=False, half=False):
        return None

    @torch.no_grad()
    def get_conditioning(self, speech_conditioning_input, cond_mel_lengths=None):
        """[b, n_mels, F] (+ lengths [b]) -> [b, 32, D] (model.py:490-502); the single full-length prompt infer.py passes is
        cached per prompt tensor object (the reference recomputes it twice per sentence, model.py:540,670).
        A padded prompt of length n is the prompt cut to n frames (the subsampled mask keeps exactly the rows a valid 3 x 3
        stride-2 convolution of n frames has, masked keys are keys the shorter sequence does not have) except that the
        convolution module's depthwise convolution sees GLU(pointwise bias) behind the end instead of zeros - handled in
        the engine (itts_conditioning_padded)."""
        x = speech_conditioning_input
        if x.ndim == 2:
            x = x.unsqueeze(0)
        lens = None if cond_mel_lengths is None else [int(v) for v in torch.as_tensor(cond_mel_lengths).reshape(-1).tolist()]
        if x.shape[0] > 1 or (lens is not None and lens[0] != x.shape[-1]):
            if lens is None:
                lens = [x.shape[-1]] * x.shape[0]
            if len(lens) == 1 and x.shape[0] > 1:
                lens = lens * x.shape[0]
            with self._eng.lock:
                return torch.cat([self._eng.conditioning(x[i:i + 1], lens[i]) for i in range(x.shape[0])], 0)
        # cached per prompt TENSOR OBJECT: the key holds a reference to it, so its storage cannot be freed and handed to a
        # different same-shape prompt while the entry is alive (an address-only key would alias two speakers)
        with self._eng.lock:
            hit = self._cond_key is not None and self._cond_key[0] is x and self._cond_key[1] == x._version
            if not hit:
                self._cond, self._cond_key = self._eng.conditioning(x), (x, x._version)
            return self._cond

    @torch.no_grad()
    def inference_speech(self, speech_conditioning_mel, text_inputs, cond_mel_lengths=None, input_tokens=None,
                         num_return_sequences=1, max_generate_length=None, typical_sampling=False, typical_mass=.9,
                         **hf_generate_kwargs):
        """Mel codes [b, <= max_generate_length] (model.py:655-708).  HF `generate` kwargs are accepted: greedy search
        (do_sample=False), multinomial sampling (do_sample=True), beam-sample (do_sample=True, num_beams 2..10) and beam
        search (do_sample=False, num_beams > 1) with top_p, temperature, repetition_penalty, length_penalty and typical_sampling
        in HF 4.36.2 semantics: top_k in [1, 128] entirely on the device, `top_k = 0 / None` (warper off) or > 128 with the warpers
        and draws on the host over the whole vocabulary (one or several beams).  `input_tokens` [b or 1, n] (model.py:672-686):
        given mel tokens the generation continues after; like the reference, the returned codes start after them - with
        num_return_sequences > 1 as well (the reference's row expansion, see below)."""
        nrs = int(num_return_sequences)
        if nrs < 1:
            raise ValueError("num_return_sequences has to be >= 1")
        nbeams = int(hf_generate_kwargs.get("num_beams", 1) or 1)
        if nrs > 1 and nbeams > 1:
            # beam modes: HF 4.36.2 expands the rows by num_beams only and BeamSearchScorer keeps num_beam_hyps_to_keep =
            # num_return_sequences hypotheses per row (model.py:655,698-703): the n best of every row, best first
            if nrs > nbeams:
                raise ValueError("`num_return_sequences` has to be smaller or equal to `num_beams`.")
        elif nrs > 1:
            # one beam: HF expands every input row num_return_sequences times (repeat_interleave) and samples the copies
            # independently (GenerationMixin._expand_inputs_for_generation); greedy search with nrs > 1 is an error in HF too
            if not hf_generate_kwargs.get("do_sample", False):
                raise ValueError("num_return_sequences has to be 1 when doing greedy search")
        sample_kw = infer_core.sampling_kwargs(hf_generate_kwargs.get("do_sample", False), hf_generate_kwargs.get("num_beams", 1),
                                               hf_generate_kwargs.get("top_k", 50), hf_generate_kwargs.get("top_p", 1.0),
                                               hf_generate_kwargs.get("temperature", 1.0), typical_sampling, typical_mass,
                                               hf_generate_kwargs.get("length_penalty", 1.0))  # HF's own default is 1.0
        cond = self.get_conditioning(speech_conditioning_mel, cond_mel_lengths)
        ids = text_inputs.detach().cpu().numpy() if isinstance(text_inputs, torch.Tensor) else np.asarray(text_inputs)
        if ids.ndim == 1:
            ids = ids[None]
        it = None
        if input_tokens is not None:
            it = input_tokens.detach().cpu().numpy() if isinstance(input_tokens, torch.Tensor) else np.asarray(input_tokens)
            it = np.atleast_2d(it).astype(np.int32)
        if cond.shape[0] not in (1, ids.shape[0]):
            raise ValueError(f"{cond.shape[0]} conditioning prompts for {ids.shape[0]} text rows")
        if it is not None and nrs > 1:
            # model.py:672-686: BEFORE generate() the reference repeats the text rows to num_return_sequences rows
            # (`input_ids.repeat(nrs // b, 1)`) and the given tokens likewise; generate() then expands every one of those rows
            # again (x nrs sampled copies, or x num_beams with nrs hypotheses returned per row): nrs * nrs sequences come back.
            # Row j of the nrs pre-expansion rows continues tokens j % bt; its prefix embedding is text row j // (nrs // b)
            # (store_mel_emb + repeat_interleave, model.py:131-139) while its attention mask is row j % b of the tiled masks -
            # the same row only if b == 1 or every text row has the same number of valid tokens, which is required here.
            b, bt = ids.shape[0], it.shape[0]
            assert nrs % bt == 0, "The num_return_sequences must be divisible by the batch number of input_tokens"
            assert nrs % b == 0, "The num_return_sequences must be divisible by the batch number of text_inputs"
            valid = [int(((r != self.start_text_token) & (r != self.stop_text_token)).sum()) for r in ids]
            if b > 1 and len(set(valid)) > 1:
                raise ValueError("num_return_sequences > 1 with input_tokens and a padded text batch: the reference pairs the prefix of "
                                 "one row with the attention mask of another (model.py:131-139 vs :680-683); pass rows of equal length")
            row_of = [j // (nrs // b) for j in range(nrs)]
            ids = ids[row_of]
            it = it[[j % bt for j in range(nrs)]]
            if cond.shape[0] > 1:
                cond = cond[row_of]
        if nrs > 1 and nbeams <= 1:
            ids = np.repeat(ids, nrs, axis=0)
            if it is not None and it.shape[0] > 1:
                it = np.repeat(it, nrs, axis=0)
            if cond.shape[0] > 1:
                cond = cond.repeat_interleave(nrs, 0)
        max_gen = self.max_mel_tokens - 1 if max_generate_length is None else int(max_generate_length)
        rep = float(hf_generate_kwargs.get("repetition_penalty", 1.0) or 1.0)
        n_forced = 0
        with self._eng.lock:
            if it is not None:
                n_forced = it.shape[1]
                self._eng.set_input_tokens(it)
            try:
                # max_length = trunc_index + max_generate_length with trunc_index counting the given tokens (model.py:687,695)
                codes = self._eng.generate(cond, ids, max_gen + n_forced, repetition_penalty=rep,
                                           num_return_sequences=nrs if nbeams > 1 else 1, **sample_kw)
            finally:
                if n_forced:
                    self._eng.set_input_tokens(None)
        return torch.from_numpy(codes[:, n_forced:]).to(self._eng.device)

    @torch.no_grad()
    def forward(self, speech_conditioning_latent, text_inputs, text_lengths, mel_codes, wav_lengths,
                cond_mel_lengths=None, types=None, text_first=True, raw_mels=None, return_attentions=False,
                return_latent=False, clip_inputs=False):
        """Only the `return_latent=True`, batch-1 form infer.py:194-200 uses (model.py:521-589)."""
        if not return_latent or not text_first or raw_mels is not None or types is not None:
            raise NotImplementedError("UnifiedVoice.forward: only return_latent=True inference is implemented")
        cond = self.get_conditioning(speech_conditioning_latent, cond_mel_lengths)
        t = text_inputs.detach().cpu().numpy().reshape(-1)
        c = mel_codes.detach().cpu().numpy().reshape(-1)
        with self._eng.lock:
            return self._eng.latent(cond, t, c)

    __call__ = forward
