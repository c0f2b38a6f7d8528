"""Python face of the HIP engine: owns the weight arena (one contiguous device buffer, so a single RCCL
broadcast replicates a model across ranks), binds it into libitts_hip and exposes the hot-path stages
as tensor-in / tensor-out calls.  torch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import lib as L
from .config import ecapa_dims, perceiver_inner

_TORCH_DT = {L.F32: torch.float32, L.BF16: torch.bfloat16, L.FP8: torch.uint8}
_DT_BYTES = {L.F32: 4, L.BF16: 2, L.FP8: 1}


def make_config(cfg, dtype: int, max_batch: int = 64) -> L.Config:
    g, h, v = cfg["gpt"], cfg["bigvgan"], cfg["vqvae"]
    cm = g["condition_module"]
    c = L.Config()
    c.dtype = dtype
    c.model_dim, c.layers, c.heads = g["model_dim"], g["layers"], g["heads"]
    c.max_mel_tokens, c.max_text_tokens = g["max_mel_tokens"], g["max_text_tokens"]
    c.number_text_tokens, c.number_mel_codes = g["number_text_tokens"], g["number_mel_codes"]
    c.start_mel_token, c.stop_mel_token = g["start_mel_token"], g["stop_mel_token"]
    c.start_text_token, c.stop_text_token = g["start_text_token"], g["stop_text_token"]
    c.cond_latents = g.get("condition_num_latent", 32)
    c.cond_dim, c.cond_ff, c.cond_heads = cm["output_size"], cm["linear_units"], cm["attention_heads"]
    c.cond_blocks, c.cond_idim = cm["num_blocks"], 100
    c.perc_inner, c.perc_layers = perceiver_inner(g), 2
    c.bv_gpt_dim, c.bv_init_ch = h["gpt_dim"], h["upsample_initial_channel"]
    c.bv_num_up = len(h["upsample_rates"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        c.bv_up_rates[i], c.bv_up_kernels[i] = u, k
    c.bv_num_res = len(h["resblock_kernel_sizes"])
    c.bv_num_dil = len(h["resblock_dilation_sizes"][0])
    for j, (k, ds) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
        c.bv_res_kernels[j] = k
        for l, d in enumerate(ds):
            c.bv_res_dils[j][l] = d
    c.bv_spk_dim, c.bv_num_mels = h["speaker_embedding_dim"], h["num_mels"]
    e = ecapa_dims(h)
    for i in range(5):
        c.ec_channels[i], c.ec_kernels[i], c.ec_dils[i] = e["channels"][i], e["kernel_sizes"][i], e["dilations"][i]
    c.ec_att, c.ec_scale, c.ec_se = e["attention_channels"], e["res2net_scale"], e["se_channels"]
    c.dv_channels, c.dv_tokens, c.dv_hidden = v["channels"], v["num_tokens"], v["hidden_dim"]
    c.dv_resblocks, c.dv_codebook, c.dv_layers, c.dv_kernel = v["num_resnet_blocks"], v["codebook_dim"], v["num_layers"], v["kernel_size"]
    c.max_batch = max_batch
    return c


class WeightArena:
    """All packed tensors in ONE device buffer (256-byte aligned slots) + a manifest."""

    def __init__(self, packed: Dict[str, Tuple[str, np.ndarray]], dtype: int, device, half=torch.bfloat16):
        """half: the torch dtype of the engine's 16-bit storage type (bfloat16, or float16 for the f16 build of the library)."""
        self.dtype = dtype
        self.half = half
        self.manifest: List[Tuple[str, int, int, Tuple[int, ...]]] = []  # name, offset, dt, shape
        off = 0
        for name, (tag, arr) in packed.items():
            dt = dtype if tag == "w" else (L.FP8 if tag == "q" else L.F32)  # "q": fp8 e4m3 bytes (uint8 array)
            nbytes = int(np.prod(arr.shape)) * _DT_BYTES[dt]
            self.manifest.append((name, off, dt, tuple(int(x) for x in arr.shape)))
            off = (off + nbytes + 255) // 256 * 256
        self.nbytes = off
        self.buf = torch.empty(off, dtype=torch.uint8, device=device)
        # stage through pinned-free host tensors in chunks (bf16 rounding = torch RNE, same as the device cast)
        for (name, o, dt, shape), (tag, arr) in zip(self.manifest, packed.values()):
            if dt == L.FP8:
                t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.uint8))
                self.buf[o:o + t.numel()].copy_(t.reshape(-1), non_blocking=False)
                continue
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
            if dt == L.BF16:
                t = t.to(half)
            n = t.numel() * t.element_size()
            self.buf[o:o + n].copy_(t.reshape(-1).view(torch.uint8), non_blocking=False)

    def view(self, name: str) -> torch.Tensor:
        for n, o, dt, shape in self.manifest:
            if n == name:
                nb = int(np.prod(shape)) * _DT_BYTES[dt]
                return self.buf[o:o + nb].view(self.half if dt == L.BF16 else _TORCH_DT[dt]).view(*shape)
        raise KeyError(name)


class Engine:
    def __init__(self, cfg, dtype: str = "bf16", device: str = "cuda:0", max_batch: int = 64):
        if not torch.cuda.is_available():
            raise RuntimeError("itts_hip.Engine needs an MI355X (no CPU fallback in the product path)")
        # "f16": IEEE half storage (the reference's GPU precision, infer.py:39,44,52) = the second build of the library, in
        # which the 16-bit dtype code means binary16; everything else is the same engine
        self.half_name = "f16" if dtype in ("f16", "fp16", "half") else "bf16"
        self.lib = L.load(self.half_name)
        self.cfg = cfg
        self.dt = {"fp32": L.F32, "f32": L.F32, "bf16": L.BF16, "f16": L.BF16, "fp16": L.BF16, "half": L.BF16}[dtype]
        self.half_dtype = torch.float16 if self.half_name == "f16" else torch.bfloat16
        self.tdt = self.half_dtype if self.dt == L.BF16 else _TORCH_DT[self.dt]
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.side = torch.cuda.Stream(device=self.device)  # ecapa(overlap=True): the speaker encoder beside conditioning / prefill
        self._side_busy = False
        self.ccfg = make_config(cfg, self.dt, max_batch)
        h = C.c_void_p()
        self._ck(self.lib.itts_engine_create(C.byref(self.ccfg), C.byref(h)), "engine_create")
        self.h = h
        self.arenas: List[WeightArena] = []
        # one engine = one decode state + captured graphs: callers on several host threads (the reference web UI starts a
        # worker thread per request into one IndexTTS, webui.py:441-452) serialise on this lock in the drop-in classes
        self.lock = threading.RLock()
        self.up_total = int(np.prod(cfg["bigvgan"]["upsample_rates"]))

    def _ck(self, status, what=""):
        L.check(status, what, self.lib)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                torch.cuda.synchronize(self.device)
                self.lib.itts_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- weights ----
    def load_packed(self, packed: Dict[str, Tuple[str, np.ndarray]], arena: Optional[WeightArena] = None):
        a = arena or WeightArena(packed, self.dt, self.device, self.half_dtype)
        self.arenas.append(a)
        base = a.buf.data_ptr()
        for name, off, dt, shape in a.manifest:
            dims = (C.c_int64 * len(shape))(*shape)
            self._ck(self.lib.itts_engine_bind_tensor(self.h, name.encode(), C.c_void_p(base + off), dt, len(shape), dims),
                    f"bind {name}")
        return a

    def finalize(self):
        self._ck(self.lib.itts_engine_finalize(self.h), "finalize")

    def debug(self, taps: bool = False, force_simple: bool = False, no_graph: bool = False, fuse: bool = False,
              no_engine: bool = False, engine: bool = False):
        """no_engine: keep the five-launches-per-layer decode step instead of the persistent decode engine; engine: use
        the persistent engine (where it applies) whatever ITTS_ENGINE / the built-in default says (A/B, parity tests)."""
        self._ck(self.lib.itts_debug_enable(self.h, int(taps) | (int(force_simple) << 1) | (int(no_graph) << 2) | (int(fuse) << 3)
                                           | (int(no_engine) << 4) | (int(engine) << 5)))

    def fetch_tap(self, name: str) -> np.ndarray:
        n = self.lib.itts_debug_fetch(self.h, name.encode(), None, 0)
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, dtype=np.float32)
        self.lib.itts_debug_fetch(self.h, name.encode(), out.ctypes.data_as(C.c_void_p), n)
        return out

    # ---- helpers ----
    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _exit(self):
        # (NOT the side stream: the caller's stream is what the next engine call waits for in _enter(), and an overlapped ecapa()
        #  must not hold that call up - its result is joined where it is consumed: _join_side())
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def join_side(self):
        """After ecapa(overlap=True): make the engine's stream AND the caller's current stream wait for the speaker embedding."""
        if self._side_busy:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        self._join_side()

    def _join_side(self):
        """The engine's own stream waits for an overlapped ecapa(): in front of the decode steps (the persistent decode engine wants
        every CU of the device) and of the vocoder (which reads the embedding)."""
        if self._side_busy:
            self.stream.wait_stream(self.side)
            self._side_busy = False

    def to_act(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(device=self.device, dtype=self.tdt).contiguous()

    # ---- stages ----
    def conditioning(self, mel_bcf: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:
        """mel [1, n_mels, F] (the reference's layout) -> cond fp32 [1, latents, D]; length < F: the prompt is the first
        `length` frames of a padded tensor (get_conditioning with cond_mel_lengths)."""
        assert mel_bcf.ndim == 3 and mel_bcf.shape[0] == 1
        mel = self.to_act(mel_bcf.transpose(1, 2))
        F = mel.shape[1]
        out = torch.empty(1, self.ccfg.cond_latents, self.ccfg.model_dim, dtype=torch.float32, device=self.device)
        self._enter()
        if length is not None and int(length) < F:
            self._ck(self.lib.itts_conditioning_padded(self.h, mel.data_ptr(), int(length), F, out.data_ptr(), self._s()), "conditioning")
        else:
            self._ck(self.lib.itts_conditioning(self.h, mel.data_ptr(), F, out.data_ptr(), self._s()), "conditioning")
        self._exit()
        mel.record_stream(self.stream)
        return out

    def ecapa(self, mel_bfc: torch.Tensor, overlap: bool = False) -> torch.Tensor:
        """mel_ref [B, F, n_mels] -> spk fp32 [B, E].
        overlap=True: enqueued on the engine's SIDE stream (its own scratch arena in the library), so that the engine calls that
        FOLLOW - conditioning, prefill: chains of small kernels, as this one - run beside it (conditioning + speaker encoder 3.65 ->
        2.40 ms at IndexTTS-1.5 sizes).  CONTRACT: the returned tensor is for THIS engine's vocoder calls (bigvgan / bigvgan_grouped
        join the side stream themselves, and so does prefill() in front of the decode steps); any other consumer calls
        join_side() first.  IndexTTS.infer / infer_batch and bench.py use it exactly so."""
        mel = self.to_act(mel_bfc)
        B, F, _ = mel.shape
        out = torch.empty(B, self.ccfg.bv_spk_dim, dtype=torch.float32, device=self.device)
        if overlap:
            self._join_side()  # one overlapped call at a time (one side arena)
            self.side.wait_stream(torch.cuda.current_stream(self.device))
            self._ck(self.lib.itts_ecapa(self.h, mel.data_ptr(), B, F, out.data_ptr(), C.c_void_p(self.side.cuda_stream)), "ecapa")
            self._side_busy = True
            mel.record_stream(self.side)
            out.record_stream(self.side)
            return out
        self._enter()
        self._ck(self.lib.itts_ecapa(self.h, mel.data_ptr(), B, F, out.data_ptr(), self._s()), "ecapa")
        self._exit()
        mel.record_stream(self.stream)
        return out

    def prefill(self, cond: torch.Tensor, text_ids: np.ndarray, max_gen: int, repetition_penalty: float = 10.0,
                suppress_stop: bool = False):
        ids = np.ascontiguousarray(text_ids, dtype=np.int32)
        assert ids.ndim == 2
        B, Lt = ids.shape
        cond = cond.to(device=self.device, dtype=torch.float32).contiguous().view(-1, self.ccfg.model_dim)
        per_row = cond.shape[0] == B * self.ccfg.cond_latents and B > 1  # one prompt per row (a batch of prompts)
        assert per_row or cond.shape[0] == self.ccfg.cond_latents, cond.shape
        self._ck(self.lib.itts_gpt_set_cond_per_row(self.h, int(per_row)), "gpt_set_cond_per_row")
        self._enter()
        self._ck(self.lib.itts_gpt_prefill(self.h, cond.data_ptr(), ids.ctypes.data_as(C.c_void_p), B, Lt, max_gen,
                                          float(repetition_penalty), int(suppress_stop), self._s()), "gpt_prefill")
        self._join_side()  # an overlapped ecapa() ends before the first decode step
        self._gen = (B, max_gen)

    def set_sampling(self, do_sample: bool, top_k: int = 30, top_p: float = 0.8, temperature: float = 1.0,
                     uniforms: Optional[np.ndarray] = None):
        """HF GenerationMixin.sample configuration (infer.py:116-124) for the following prefill/decode calls;
        uniforms [max_gen, B] float32 in [0, 1) are the draws (step k, row b)."""
        if not do_sample:
            self._ck(self.lib.itts_gpt_set_sampling(self.h, 0, 0, 1.0, 1.0, None, 0), "gpt_set_sampling")
            return
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        self._ck(self.lib.itts_gpt_set_sampling(self.h, 1, int(top_k), float(top_p), float(temperature),
                                               u.ctypes.data_as(C.c_void_p), u.size), "gpt_set_sampling")

    def set_beam_sample(self, num_beams: int, top_k: int = 30, top_p: float = 0.8, temperature: float = 1.0,
                        uniforms: Optional[np.ndarray] = None, do_sample: bool = True, length_penalty: float = 0.0,
                        num_return_sequences: int = 1, host: bool = False):
        """HF beam_sample (do_sample: the reference's default generate() mode, infer.py:116-124; uniforms
        [max_gen, B, 2 * num_beams] float32 in [0, 1)) or beam_search (not do_sample: deterministic) for the following
        generations.  num_beams <= 1 switches beams off.  num_return_sequences: the n best hypotheses per row (fetch
        returns [B * n, max_gen], best first)."""
        if num_beams <= 1:
            self._ck(self.lib.itts_gpt_set_beams(self.h, 1, 1, 1, 1.0, 1.0, 0.0, None, 0), "gpt_set_beams")
            self._ck(self.lib.itts_gpt_set_beam_returns(self.h, 1), "gpt_set_beam_returns")
            self._nb, self._nret = 1, 1
            return
        self._ck(self.lib.itts_gpt_set_beam_returns(self.h, int(num_return_sequences)), "gpt_set_beam_returns")
        self._nret = int(num_return_sequences)
        dev_draws = do_sample and not host  # host: the caller warps and draws (any top_k), the library needs no uniforms
        u = np.ascontiguousarray(uniforms, dtype=np.float32) if dev_draws else None
        self._ck(self.lib.itts_gpt_set_beams(self.h, int(num_beams), int(bool(do_sample)), int(top_k), float(top_p), float(temperature),
                                            float(length_penalty), u.ctypes.data_as(C.c_void_p) if dev_draws else None,
                                            u.size if dev_draws else 0), "gpt_set_beams")
        self._nb = int(num_beams)

    def set_forced(self, ids: Optional[np.ndarray]):
        """Forced tokens [B or 1, n] (int, -1 = free) for the first n steps of the following generations; None clears."""
        if ids is None or np.asarray(ids).size == 0:
            self._ck(self.lib.itts_gpt_set_forced(self.h, None, 0, 0), "gpt_set_forced")
            return
        a = np.ascontiguousarray(np.atleast_2d(ids), dtype=np.int32)
        self._ck(self.lib.itts_gpt_set_forced(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1]), "gpt_set_forced")

    def set_input_tokens(self, ids: Optional[np.ndarray]):
        """HF `input_tokens` [B or 1, n] (inference_speech, model.py:672-686) for the following generations: forced like
        set_forced, at the reference's positions (token k at mel position k + 1); None clears."""
        if ids is None or np.asarray(ids).size == 0:
            self._ck(self.lib.itts_gpt_set_input_tokens(self.h, None, 0, 0), "gpt_set_input_tokens")
            return
        a = np.ascontiguousarray(np.atleast_2d(ids), dtype=np.int32)
        self._ck(self.lib.itts_gpt_set_input_tokens(self.h, a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1]), "gpt_set_input_tokens")

    def decode_mode(self) -> int:
        """1 if the last decode step ran on the persistent decode engine, 0 for the five-launches-per-block path."""
        return int(self.lib.itts_gpt_decode_mode(self.h))

    def decode(self, nsteps: int):
        self._ck(self.lib.itts_gpt_decode(self.h, nsteps, self._s()), "gpt_decode")

    def status(self) -> Tuple[int, int]:
        a, b = C.c_int(), C.c_int()
        self._ck(self.lib.itts_gpt_status(self.h, C.byref(a), C.byref(b), self._s()), "gpt_status")
        return a.value, b.value

    def fetch(self, logits: bool = False):
        B, mg = self._gen
        codes = np.empty((B * (getattr(self, "_nret", 1) if getattr(self, "_nb", 1) > 1 else 1), mg), dtype=np.int32)
        lg = np.empty((B * getattr(self, "_nb", 1), self.ccfg.number_mel_codes), dtype=np.float32) if logits else None
        self._ck(self.lib.itts_gpt_fetch(self.h, codes.ctypes.data_as(C.c_void_p),
                                        lg.ctypes.data_as(C.c_void_p) if logits else None, self._s()), "gpt_fetch")
        return (codes, lg) if logits else codes

    def generate(self, cond: torch.Tensor, text_ids: np.ndarray, max_gen: int, repetition_penalty: float = 10.0,
                 suppress_stop: bool = False, check_every: int = 16, do_sample: bool = False, top_k: int = 30,
                 top_p: float = 0.8, temperature: float = 1.0, seed: Optional[int] = None,
                 uniforms: Optional[np.ndarray] = None, num_beams: int = 1, typical_mass: float = 0.0,
                 length_penalty: float = 0.0, num_return_sequences: int = 1) -> np.ndarray:
        """Greedy decode (do_sample=False, num_beams=1 of tests/padding_test.py:35-46) or, with do_sample, HF
        GenerationMixin.sample (top-k / top-p / temperature, num_beams=1; draws from `uniforms` or a numpy Generator
        seeded with `seed`).  Returns int64 codes [B, n] with n <= max_gen: HF stops when every row has emitted stop
        or at max length.  num_beams > 1: HF beam_sample (do_sample; uniforms [max_gen, B, 2 * num_beams]) or beam_search
        (not do_sample) over num_beams beams per row; returns the best finalized hypothesis per row, or with
        num_return_sequences = n the n best of every row ([B * n, len], best first)."""
        kw = dict(repetition_penalty=repetition_penalty, suppress_stop=suppress_stop, check_every=check_every, do_sample=do_sample,
                  top_k=top_k, top_p=top_p, temperature=temperature, seed=seed, uniforms=uniforms, num_beams=num_beams,
                  typical_mass=typical_mass, length_penalty=length_penalty, num_return_sequences=num_return_sequences)
        try:
            return self._generate_once(cond, text_ids, max_gen, **kw)
        except L.HandoffTimeout as e:
            # The persistent decode engine needs every CU of its GPU; a wait inside it gave up (another process or a long
            # foreign kernel held CUs).  The library has switched this engine object to the launch path: redo the utterance
            # there, once, in this process - same prefill, same uniforms / seed, so the same ids as an undisturbed run.
            _warn_downgrade(e)
            return self._generate_once(cond, text_ids, max_gen, **kw)

    def _generate_once(self, cond, text_ids, max_gen, repetition_penalty, suppress_stop, check_every, do_sample, top_k, top_p,
                       temperature, seed, uniforms, num_beams, typical_mass, length_penalty, num_return_sequences) -> np.ndarray:
        beams = num_beams > 1
        if num_return_sequences != 1 and not beams:
            raise ValueError("num_return_sequences > 1 without beams: repeat the rows (indextts/gpt/model.py does, as HF does)")
        nrow = np.asarray(text_ids).shape[0]
        if do_sample and (not top_k or int(top_k) < 1 or int(top_k) > 128):
            # HF: TopK warper off (top_k = 0 / None) or wider than the device samplers' 128 candidates - exact on the host
            if beams:
                return self._generate_host_beams(cond, text_ids, max_gen, repetition_penalty, suppress_stop, int(top_k or 0), top_p,
                                                 temperature, seed, uniforms, typical_mass, num_beams, length_penalty,
                                                 num_return_sequences)
            return self._generate_host_sampled(cond, text_ids, max_gen, repetition_penalty, suppress_stop, int(top_k or 0), top_p,
                                               temperature, seed, uniforms, typical_mass)
        typical = bool(typical_mass)
        if typical:  # typical_sampling=True (model.py:690-697): TypicalLogitsWarper behind the repetition penalty, in every mode
            self._ck(self.lib.itts_gpt_set_typical(self.h, float(typical_mass)), "gpt_set_typical")
        if beams:
            if uniforms is None and do_sample:
                uniforms = np.random.default_rng(seed).random((max_gen, nrow, 2 * num_beams), dtype=np.float32)
            self.set_beam_sample(num_beams, top_k, top_p, temperature, uniforms, do_sample=do_sample, length_penalty=length_penalty,
                                 num_return_sequences=num_return_sequences)
        elif do_sample:
            if uniforms is None:
                uniforms = np.random.default_rng(seed).random((max_gen, nrow), dtype=np.float32)
            self.set_sampling(True, top_k, top_p, temperature, uniforms)
        try:
            self.prefill(cond, text_ids, max_gen, repetition_penalty, suppress_stop)
            done = 1
            while done < max_gen:
                if not suppress_stop:
                    step, unf = self.status()
                    done = step
                    if unf == 0:
                        break
                n = min(check_every, max_gen - done)
                self.decode(n)
                done += n
            step, unf = self.status()
            codes = self.fetch()[:, :step].astype(np.int64)
            self._exit()
        finally:
            if typical:
                self._ck(self.lib.itts_gpt_set_typical(self.h, 0.0), "gpt_set_typical")
            if beams:
                self.set_beam_sample(1)
            elif do_sample:
                self.set_sampling(False)
        # HF stops right after the step in which the last running row emitted stop: trim the look-ahead steps
        stop = self.ccfg.stop_mel_token
        n = 0
        for row in codes:
            hit = np.nonzero(row == stop)[0]
            n = max(n, int(hit[0]) + 1 if len(hit) else step)
        return codes[:, :n]

    def _generate_host_sampled(self, cond, text_ids, max_gen, repetition_penalty, suppress_stop, top_k, top_p, temperature, seed,
                               uniforms, typical_mass) -> np.ndarray:
        """HF sample() with the token choice on the host (infer_core.host_sample_step: warpers over the WHOLE vocabulary): the
        step stops behind the head GEMV, the logits come back, the chosen tokens go in through itts_gpt_commit.  One stream
        sync per token - the price of a mode the device samplers (<= 128 kept candidates) do not cover."""
        from . import infer_core

        nrow = np.asarray(text_ids).shape[0]
        if uniforms is None:
            uniforms = np.random.default_rng(seed).random((max_gen, nrow), dtype=np.float32)
        stop, start = self.ccfg.stop_mel_token, self.ccfg.start_mel_token
        seen = [{1, start} for _ in range(nrow)]  # fake prefix ids are all 1, the last one start_mel (model.py:644-653)
        unfinished = np.ones(nrow, dtype=bool)
        out = np.full((nrow, max_gen), stop, dtype=np.int64)
        n = 0
        self._ck(self.lib.itts_gpt_set_host_sampling(self.h, 1), "gpt_set_host_sampling")
        try:
            self.prefill(cond, text_ids, max_gen, repetition_penalty, suppress_stop)
            for k in range(max_gen):
                _, lg = self.fetch(logits=True)
                toks = infer_core.host_sample_step(lg, seen, float(repetition_penalty), float(temperature), top_k, top_p,
                                                   float(typical_mass or 0.0), uniforms[k], stop, bool(suppress_stop))
                toks = np.where(unfinished, toks, stop).astype(np.int32)
                self._ck(self.lib.itts_gpt_commit(self.h, toks.ctypes.data_as(C.c_void_p), self._s()), "gpt_commit")
                out[:, k] = toks
                n = k + 1
                for r in range(nrow):
                    seen[r].add(int(toks[r]))
                unfinished &= toks != stop
                if not unfinished.any() or n == max_gen:
                    break
                self.decode(1)
            self._exit()
        finally:
            self._ck(self.lib.itts_gpt_set_host_sampling(self.h, 0), "gpt_set_host_sampling")
        return out[:, :n]

    def _generate_host_beams(self, cond, text_ids, max_gen, repetition_penalty, suppress_stop, top_k, top_p, temperature, seed,
                             uniforms, typical_mass, num_beams, length_penalty, num_return_sequences) -> np.ndarray:
        """HF beam_sample with the warpers and the draws on the host (infer_core.host_beam_step: any top_k, whole vocabulary),
        BeamSearchScorer.process / beam re-ordering / finalize on the device (itts_gpt_commit_beams).  One logits + beam-state
        read-back per token."""
        from . import infer_core

        ids_in = np.asarray(text_ids)
        items, nb, nd = ids_in.shape[0], int(num_beams), 2 * int(num_beams)
        if uniforms is None:
            uniforms = np.random.default_rng(seed).random((max_gen, items, nd), dtype=np.float32)
        u = np.ascontiguousarray(uniforms, dtype=np.float32).reshape(-1, items, nd)
        stop, start = self.ccfg.stop_mel_token, self.ccfg.start_mel_token
        hist = np.empty((items * nb, max_gen), dtype=np.int32)
        scores = np.empty(items * nb, dtype=np.float32)
        done = np.empty(items, dtype=np.int32)
        step = C.c_int()
        self._ck(self.lib.itts_gpt_set_host_sampling(self.h, 1), "gpt_set_host_sampling")
        try:
            self.set_beam_sample(nb, top_k, top_p, temperature, None, do_sample=True, length_penalty=length_penalty,
                                 num_return_sequences=num_return_sequences, host=True)
            self.prefill(cond, text_ids, max_gen, repetition_penalty, suppress_stop)
            while True:
                self._ck(self.lib.itts_gpt_beam_state(self.h, hist.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p),
                                                     done.ctypes.data_as(C.c_void_p), C.byref(step), self._s()), "gpt_beam_state")
                k = step.value
                if k >= max_gen or done.all():
                    break
                lg = np.empty((items * nb, self.ccfg.number_mel_codes), dtype=np.float32)
                self._ck(self.lib.itts_gpt_fetch(self.h, None, lg.ctypes.data_as(C.c_void_p), self._s()), "gpt_fetch")
                psc, ptok, pbeam = infer_core.host_beam_step(lg, hist, k, scores, done, nb, float(repetition_penalty), float(temperature),
                                                             top_k, top_p, float(typical_mass or 0.0), u[k], stop, bool(suppress_stop),
                                                             start)
                self._ck(self.lib.itts_gpt_commit_beams(self.h, psc.ctypes.data_as(C.c_void_p), ptok.ctypes.data_as(C.c_void_p),
                                                       pbeam.ctypes.data_as(C.c_void_p), self._s()), "gpt_commit_beams")
                if k + 1 >= max_gen:
                    break
                self.decode(1)
            step2, _ = self.status()
            codes = self.fetch()[:, :step2].astype(np.int64)
            self._exit()
        finally:
            self.set_beam_sample(1)
            self._ck(self.lib.itts_gpt_set_host_sampling(self.h, 0), "gpt_set_host_sampling")
        n = 0
        for row in codes:
            hit = np.nonzero(row == stop)[0]
            n = max(n, int(hit[0]) + 1 if len(hit) else step2)
        return codes[:, :n]

    def latent(self, cond: torch.Tensor, text_ids: np.ndarray, codes: np.ndarray) -> torch.Tensor:
        """-> latent [1, T, D] engine dtype."""
        t = np.ascontiguousarray(text_ids, dtype=np.int32).reshape(-1)
        c = np.ascontiguousarray(codes, dtype=np.int32).reshape(-1)
        cond = cond.to(device=self.device, dtype=torch.float32).contiguous().view(-1, self.ccfg.model_dim)
        out = torch.empty(1, c.shape[0], self.ccfg.model_dim, dtype=self.tdt, device=self.device)
        self._enter()
        self._ck(self.lib.itts_gpt_latent(self.h, cond.data_ptr(), t.ctypes.data_as(C.c_void_p), t.shape[0],
                                         c.ctypes.data_as(C.c_void_p), c.shape[0], out.data_ptr(), self._s()), "gpt_latent")
        self._exit()
        return out

    def latent_batch(self, cond: torch.Tensor, texts, codes) -> List[torch.Tensor]:
        """Several sentences in one pass -> list of latents [1, T_i, D] (views of one buffer)."""
        ts = [np.ascontiguousarray(t, dtype=np.int32).reshape(-1) for t in texts]
        cs = [np.ascontiguousarray(c, dtype=np.int32).reshape(-1) for c in codes]
        tl = np.asarray([t.shape[0] for t in ts], dtype=np.int32)
        cl = np.asarray([c.shape[0] for c in cs], dtype=np.int32)
        tcat, ccat = np.concatenate(ts), np.concatenate(cs)
        cond = cond.to(device=self.device, dtype=torch.float32).contiguous().view(-1, self.ccfg.model_dim)
        out = torch.empty(int(cl.sum()), self.ccfg.model_dim, dtype=self.tdt, device=self.device)
        self._enter()
        self._ck(self.lib.itts_gpt_latent_batch(self.h, cond.data_ptr(), tcat.ctypes.data_as(C.c_void_p),
                                               tl.ctypes.data_as(C.c_void_p), ccat.ctypes.data_as(C.c_void_p),
                                               cl.ctypes.data_as(C.c_void_p), len(ts), out.data_ptr(), self._s()),
                "gpt_latent_batch")
        self._exit()
        res, o = [], 0
        for n in cl:
            res.append(out[o:o + int(n)].unsqueeze(0))
            o += int(n)
        return res

    def bigvgan(self, latent: torch.Tensor, spk: torch.Tensor) -> torch.Tensor:
        """latent [B, T, D], spk [B, E] -> wav fp32 [B, 1, T*up]."""
        lat = self.to_act(latent)
        B, T, _ = lat.shape
        spk = spk.to(device=self.device, dtype=torch.float32).contiguous().view(B, -1)
        out = torch.empty(B, 1, T * self.up_total, dtype=torch.float32, device=self.device)
        self._enter()
        self._join_side()
        self._ck(self.lib.itts_bigvgan(self.h, lat.data_ptr(), spk.data_ptr(), B, T, out.data_ptr(), self._s()), "bigvgan")
        self._exit()
        lat.record_stream(self.stream)
        spk.record_stream(self.stream)
        return out

    def dvae_encode(self, mel_bct: torch.Tensor) -> np.ndarray:
        """DiscreteVAE.get_codebook_indices: mel [B, channels, T] (reference layout) -> codes int64 [B, T']."""
        mel = self.to_act(mel_bct.transpose(1, 2))
        B, T, _ = mel.shape
        Tc = _dvae_code_len(T, self.ccfg.dv_layers)
        codes = np.empty((B, Tc), dtype=np.int32)
        self._enter()
        self._ck(self.lib.itts_dvae_encode(self.h, mel.data_ptr(), B, T, codes.ctypes.data_as(C.c_void_p), self._s()), "dvae_encode")
        self._exit()
        return codes.astype(np.int64)

    def bigvgan_grouped(self, lats: List[torch.Tensor], spk: torch.Tensor) -> List[torch.Tensor]:
        """Vocoder over several sentences: latents [1, T_i, D] of EQUAL length share one batched launch sequence (rows of
        a batch are independent, so each result equals its batch-1 run); ragged lengths cannot be padded - the convs
        would see conv_pre(0) = bias instead of zero "same" padding - and go in groups of their own."""
        groups: Dict[int, List[int]] = {}
        for i, l in enumerate(lats):
            groups.setdefault(int(l.shape[1]), []).append(i)
        out: List[Optional[torch.Tensor]] = [None] * len(lats)
        cap = max(1, int(self.ccfg.max_batch))
        for T, idx in groups.items():
            for lo in range(0, len(idx), cap):
                sel = idx[lo:lo + cap]
                wav = self.bigvgan(torch.cat([lats[i] for i in sel], 0), spk.view(1, -1).expand(len(sel), -1).contiguous())
                for j, i in enumerate(sel):
                    out[i] = wav[j:j + 1]
        return out

    def dvae_decode(self, codes: np.ndarray) -> torch.Tensor:
        """codes [B, T] -> mel [B, channels, 4T] (reference layout), engine dtype."""
        c = np.ascontiguousarray(codes, dtype=np.int32)
        B, T = c.shape
        ch = self.ccfg.dv_channels
        up = 2 ** self.ccfg.dv_layers
        out = torch.empty(B, T * up, ch, dtype=self.tdt, device=self.device)
        self._enter()
        self._ck(self.lib.itts_dvae_decode(self.h, c.ctypes.data_as(C.c_void_p), B, T, out.data_ptr(), self._s()), "dvae_decode")
        self._exit()
        return out.transpose(1, 2)


def _warn_downgrade(err):
    import logging
    import warnings

    msg = ("itts_hip: the persistent decode engine timed out on a hand-off (one process per GPU is required, INTEGRATION.md); "
           f"this engine now decodes on the five-launches-per-block path and the utterance is generated again: {err}")
    logging.getLogger("itts_hip").warning(msg)
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _dvae_code_len(T: int, layers: int) -> int:
    for _ in range(layers):
        T = (T + 1) // 2
    return T


def build_engine(cfg, dtype: str = "bf16", device: str = "cuda:0", seed: int = 1234, parts=("gpt", "bigvgan", "dvae"),
                 state_dicts: Optional[dict] = None, max_batch: int = 64, gpt_fp8: str = "") -> Engine:
    """Engine with synthetic (PRNG) or supplied reference-layout state dicts.  gpt_fp8: "" = plain weights;
    "fp8" = GPT projections quantised to e4m3 (power-of-two row scales) with the fp8 bytes used by the decode GEMV;
    "dequant" = the same quantised model but every kernel reads its bf16 dequantisation (the fp8 path's reference)."""
    from . import pack, synth

    eng = Engine(cfg, dtype, device, max_batch)
    sds = state_dicts or {}
    if "gpt" in parts:
        packed = pack.pack_gpt(sds.get("gpt") or synth.gpt_state_dict(cfg, seed), cfg)
        if gpt_fp8:
            packed = pack.quantize_gpt_fp8(packed, keep_bytes=(gpt_fp8 == "fp8"))
        eng.load_packed(packed)
    if "bigvgan" in parts:
        eng.load_packed(pack.pack_bigvgan(sds.get("bigvgan") or synth.bigvgan_state_dict(cfg, seed), cfg))
    if "dvae" in parts:
        eng.load_packed(pack.pack_dvae(sds.get("dvae") or synth.dvae_state_dict(cfg, seed), cfg))
    eng.finalize()
    return eng
