"""Config-shaped synthetic checkpoints (same key names / shapes as the reference's state dicts).

The key layout is the one `UnifiedVoice`, `BigVGAN` (after `remove_weight_norm`, infer.py:66) and
`DiscreteVAE` expose (SURVEY.md section 5 "Checkpoint / resume"); `oracle/make_golden.py` asserts that
the key set and every shape equal the reference modules' `state_dict()`.

Initialisation is chosen so that the network is *exercised* (SURVEY.md 8c sensitivity warning): sharp-ish
attention (large Q/K gain), O(1) embeddings, variance-preserving projections, non-trivial LayerNorm /
BatchNorm statistics, SnakeBeta log-alpha/log-beta spread around 0.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from . import prng
from .config import ecapa_dims, perceiver_inner


class _B:
    """Collects tensor specs, then materialises them on a thread pool (numpy releases the GIL)."""

    def __init__(self, seed: int, prefix: str = ""):
        self.seed, self.prefix, self.sd = seed, prefix, {}
        self.jobs = []

    def t(self, name, shape, std=1.0, mean=0.0, post=None):
        self.jobs.append((name, tuple(shape), std, mean, post))

    def build(self):
        from concurrent.futures import ThreadPoolExecutor
        import os

        def run(job):
            name, shape, std, mean, post = job
            x = prng.tensor(self.prefix + name, self.seed, shape, std=std, mean=mean)
            return name, (post(x) if post else x)

        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:
            for name, x in ex.map(run, self.jobs):
                self.sd[name] = x
        self.jobs = []
        return self.sd

    def lin(self, name, out_f, in_f, gain=1.0, bias=True, extra=()):
        self.t(name + ".weight", (out_f, in_f, *extra), std=gain / math.sqrt(in_f * int(np.prod(extra or (1,)))))
        if bias:
            self.t(name + ".bias", (out_f,), std=0.05)

    def ln(self, name, d):
        self.t(name + ".weight", (d,), std=0.1, mean=1.0)
        self.t(name + ".bias", (d,), std=0.05)

    def bn(self, name, d):
        self.t(name + ".weight", (d,), std=0.1, mean=1.0)
        self.t(name + ".bias", (d,), std=0.05)
        self.t(name + ".running_mean", (d,), std=0.1)
        self.t(name + ".running_var", (d,), std=0.2, mean=1.0, post=lambda x: np.abs(x) + np.float32(0.05))
        self.sd[name + ".num_batches_tracked"] = np.array(1, dtype=np.int64)


def conformer_pe(max_len: int, d: int) -> np.ndarray:
    """Sinusoid table `pe` (embedding.py:36-44), computed in float32 like the reference buffer."""
    import torch

    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).numpy()


def gpt_state_dict(cfg, seed: int = 1234, profile: str = "sharp", stop_bias: float = 0.0) -> Dict[str, np.ndarray]:
    """profile "sharp" (default, every parity fixture): Q / K projections x 2.5, so attention is close to an arg-max and a
    one-token change flips greedy ids (SURVEY 8c sensitivity warning) - which also amplifies bf16 weight rounding to 0.1 - 0.3
    relative on logits / latents.  profile "smooth": Q / K gain 1 (score std ~ 1, soft attention), mel_head gain 1: the
    conditioning of a trained checkpoint - the bf16 ACCURACY tests use it, so their bounds can be tight enough to fail.
    stop_bias: added to mel_head.bias[stop_mel_token] - the random checkpoint otherwise (almost) never emits the stop token, and
    the eos bookkeeping of the decode loops (rows finishing at different steps, pad fill, beams finishing) would go untested."""
    assert profile in ("sharp", "smooth"), profile
    qk_gain = 2.5 if profile == "sharp" else 1.0
    head_gain = 3.0 if profile == "sharp" else 1.0
    g = cfg["gpt"]
    cm = g["condition_module"]
    D, NL, H = g["model_dim"], g["layers"], g["heads"]
    b = _B(seed, "gpt/")
    # --- conformer (conformer_encoder.py, subsampling.py:135-160, attention.py) ---
    od, lu, ah = cm["output_size"], cm["linear_units"], cm["attention_heads"]
    idim = 100
    ce = "conditioning_encoder."
    b.lin(ce + "embed.conv.0", od, 1, gain=1.0, extra=(3, 3))
    b.lin(ce + "embed.out.0", od, od * ((idim - 1) // 2), gain=1.0)
    b.sd[ce + "embed.pos_enc.pe"] = conformer_pe(5000, od)
    b.ln(ce + "after_norm", od)
    for i in range(cm["num_blocks"]):
        p = f"{ce}encoders.{i}."
        b.t(p + "self_attn.pos_bias_u", (ah, od // ah), std=0.3)
        b.t(p + "self_attn.pos_bias_v", (ah, od // ah), std=0.3)
        b.lin(p + "self_attn.linear_q", od, od, gain=2.0)
        b.lin(p + "self_attn.linear_k", od, od, gain=2.0)
        b.lin(p + "self_attn.linear_v", od, od)
        b.lin(p + "self_attn.linear_out", od, od, gain=0.7)
        b.lin(p + "self_attn.linear_pos", od, od, gain=1.0, bias=False)
        b.lin(p + "feed_forward.w_1", lu, od)
        b.lin(p + "feed_forward.w_2", od, lu, gain=0.7)
        b.lin(p + "conv_module.pointwise_conv1", 2 * od, od, extra=(1,))
        b.lin(p + "conv_module.depthwise_conv", od, 1, extra=(15,))
        b.ln(p + "conv_module.norm", od)
        b.lin(p + "conv_module.pointwise_conv2", od, od, gain=0.7, extra=(1,))
        for n in ("norm_ff", "norm_mha", "norm_conv", "norm_final"):
            b.ln(p + n, od)
    # --- perceiver (perceiver.py:223-274) ---
    inner = ah * 64
    ffi = perceiver_inner(g)
    pe_ = "perceiver_encoder."
    b.t(pe_ + "latents", (32, D), std=1.0)
    b.lin(pe_ + "proj_context", D, od)
    for j in range(2):
        p = f"{pe_}layers.{j}."
        b.lin(p + "0.to_q", inner, D, gain=2.0, bias=False)
        b.lin(p + "0.to_kv", 2 * inner, D, gain=1.5, bias=False)
        b.lin(p + "0.to_out", D, inner, gain=0.7, bias=False)
        b.lin(p + "1.0", 2 * ffi, D)
        b.lin(p + "1.2", D, ffi, gain=0.7)
    b.t(pe_ + "norm.gamma", (D,), std=0.1, mean=1.0)
    # --- embeddings / heads (model.py:362-379) ---
    b.t("text_embedding.weight", (g["number_text_tokens"] + 1, D), std=0.7)
    b.t("mel_embedding.weight", (g["number_mel_codes"], D), std=0.7)
    b.t("mel_pos_embedding.emb.weight", (g["max_mel_tokens"] + 3, D), std=0.5)
    b.t("text_pos_embedding.emb.weight", (g["max_text_tokens"] + 2, D), std=0.5)
    # --- GPT-2 blocks (HF Conv1D stores [in, out]) ---
    for i in range(NL):
        p = f"gpt.h.{i}."
        b.ln(p + "ln_1", D)
        def sharpen(w, D=D, gain=qk_gain):  # sharper attention: larger Q,K projections
            w[:, : 2 * D] *= np.float32(gain)
            return w

        b.t(p + "attn.c_attn.weight", (D, 3 * D), std=1.0 / math.sqrt(D), post=sharpen)
        b.t(p + "attn.c_attn.bias", (3 * D,), std=0.05)
        b.t(p + "attn.c_proj.weight", (D, D), std=0.5 / math.sqrt(D))
        b.t(p + "attn.c_proj.bias", (D,), std=0.05)
        b.ln(p + "ln_2", D)
        b.t(p + "mlp.c_fc.weight", (D, 4 * D), std=1.0 / math.sqrt(D))
        b.t(p + "mlp.c_fc.bias", (4 * D,), std=0.05)
        b.t(p + "mlp.c_proj.weight", (4 * D, D), std=0.5 / math.sqrt(4 * D))
        b.t(p + "mlp.c_proj.bias", (D,), std=0.05)
    b.ln("gpt.ln_f", D)
    b.ln("final_norm", D)
    b.lin("text_head", g["number_text_tokens"] + 1, D)
    b.lin("mel_head", g["number_mel_codes"], D, gain=head_gain)
    sd = b.build()
    if stop_bias:
        hb = np.array(sd["mel_head.bias"], dtype=np.float32, copy=True)
        hb[g["stop_mel_token"]] += np.float32(stop_bias)
        sd["mel_head.bias"] = hb
    return sd


def bigvgan_state_dict(cfg, seed: int = 1234) -> Dict[str, np.ndarray]:
    h = cfg["bigvgan"]
    b = _B(seed, "bigvgan/")
    C0 = h["upsample_initial_channel"]
    b.lin("conv_pre", C0, h["gpt_dim"], extra=(7,))
    ch = C0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        # ConvTranspose1d weight [in, out, k]; each output sample sees k/u taps
        b.t(f"ups.{i}.0.weight", (cin, cout, k), std=1.0 / math.sqrt(cin * k / u))
        b.t(f"ups.{i}.0.bias", (cout,), std=0.05)
        ch = cout
        for j, (ks, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            r = f"resblocks.{i * len(h['resblock_kernel_sizes']) + j}."
            for l in range(len(dil)):
                b.lin(f"{r}convs1.{l}", ch, ch, gain=0.9, extra=(ks,))
                b.lin(f"{r}convs2.{l}", ch, ch, gain=0.5, extra=(ks,))
            for m in range(2 * len(dil)):
                b.t(f"{r}activations.{m}.act.alpha", (ch,), std=0.3)
                b.t(f"{r}activations.{m}.act.beta", (ch,), std=0.3)
    b.t("activation_post.act.alpha", (ch,), std=0.3)
    b.t("activation_post.act.beta", (ch,), std=0.3)
    b.lin("conv_post", 1, ch, gain=0.7, extra=(7,))
    # --- ECAPA-TDNN speaker encoder (ECAPA_TDNN.py:429-581) ---
    e = ecapa_dims(h)
    chs, ks_, se, att, sc = e["channels"], e["kernel_sizes"], e["se_channels"], e["attention_channels"], e["res2net_scale"]
    s = "speaker_encoder."
    b.lin(s + "blocks.0.conv.conv", chs[0], e["input_size"], gain=0.3, extra=(ks_[0],))
    b.bn(s + "blocks.0.norm.norm", chs[0])
    for i in range(1, len(chs) - 1):
        p = f"{s}blocks.{i}."
        b.lin(p + "tdnn1.conv.conv", chs[i], chs[i - 1], extra=(1,))
        b.bn(p + "tdnn1.norm.norm", chs[i])
        hc = chs[i] // sc
        for q in range(sc - 1):
            b.lin(f"{p}res2net_block.blocks.{q}.conv.conv", hc, hc, extra=(ks_[i],))
            b.bn(f"{p}res2net_block.blocks.{q}.norm.norm", hc)
        b.lin(p + "tdnn2.conv.conv", chs[i], chs[i], extra=(1,))
        b.bn(p + "tdnn2.norm.norm", chs[i])
        b.lin(p + "se_block.conv1.conv", se, chs[i], extra=(1,))
        b.lin(p + "se_block.conv2.conv", chs[i], se, extra=(1,))
    cat = chs[-2] * (len(chs) - 2)
    b.lin(s + "mfa.conv.conv", chs[-1], cat, extra=(1,))
    b.bn(s + "mfa.norm.norm", chs[-1])
    b.lin(s + "asp.tdnn.conv.conv", att, chs[-1] * 3, extra=(1,))
    b.bn(s + "asp.tdnn.norm.norm", att)
    b.lin(s + "asp.conv.conv", chs[-1], att, gain=2.0, extra=(1,))
    b.bn(s + "asp_bn.norm", chs[-1] * 2)
    b.lin(s + "fc.conv", e["lin_neurons"], chs[-1] * 2, extra=(1,))
    b.lin("cond_layer", C0, h["speaker_embedding_dim"], gain=0.5, extra=(1,))
    for i in range(len(h["upsample_rates"])):
        b.lin(f"conds.{i}", C0 // (2 ** (i + 1)), h["speaker_embedding_dim"], gain=0.5, extra=(1,))
    return b.build()


def dvae_state_dict(cfg, seed: int = 1234) -> Dict[str, np.ndarray]:
    """DiscreteVAE tensors (xtts_dvae.py:251-291): decoder + codebook (Q1) and the encoder of get_codebook_indices
    (strided convs, ResBlocks, 1x1 to the codebook dim)."""
    v = cfg["vqvae"]
    b = _B(seed, "dvae/")
    hid, cb, nl = v["hidden_dim"], v["codebook_dim"], v["num_layers"]
    chans = [hid * 2 ** i for i in range(nl)]
    dec = list(reversed(chans))
    inner = dec[0]
    dec = [inner] + dec
    idx = 0
    b.lin(f"decoder.{idx}", inner, cb, extra=(1,))
    idx += 1
    for _ in range(v["num_resnet_blocks"]):
        b.lin(f"decoder.{idx}.net.0", inner, inner, extra=(3,))
        b.lin(f"decoder.{idx}.net.2", inner, inner, extra=(3,))
        b.lin(f"decoder.{idx}.net.4", inner, inner, gain=0.5, extra=(1,))
        idx += 1
    for ci, co in zip(dec[:-1], dec[1:]):
        b.lin(f"decoder.{idx}.0.conv", co, ci, extra=(v["kernel_size"],))
        idx += 1
    b.lin(f"decoder.{idx}", v["channels"], dec[-1], extra=(1,))
    b.t("codebook.embed", (cb, v["num_tokens"]), std=1.0)
    enc = [v["channels"]] + chans
    idx = 0
    for ci, co in zip(enc[:-1], enc[1:]):
        b.lin(f"encoder.{idx}.0", co, ci, extra=(v["kernel_size"],))
        idx += 1
    for _ in range(v["num_resnet_blocks"]):
        b.lin(f"encoder.{idx}.net.0", inner, inner, extra=(3,))
        b.lin(f"encoder.{idx}.net.2", inner, inner, extra=(3,))
        b.lin(f"encoder.{idx}.net.4", inner, inner, gain=0.5, extra=(1,))
        idx += 1
    b.lin(f"encoder.{idx}", cb, inner, gain=4.0, extra=(1,))  # spread the codes: |x| comparable to the codebook rows
    return b.build()


# ---- synthetic inputs ---------------------------------------------------------------

def prompt_mel(frames: int = 511, seed: int = 7, n_mels: int = 100) -> np.ndarray:
    """[1, n_mels, frames] ~ (-4, 2^2) clipped at ln(1e-7) (log-mel range of feature_extractors.py:49)."""
    x = prng.tensor("prompt_mel", seed, (1, n_mels, frames), std=2.0, mean=-4.0)
    return np.maximum(x, np.float32(math.log(1e-7))).astype(np.float32)


def text_ids(n: int, seed: int, vocab: int) -> np.ndarray:
    """n text ids uniform in [2, vocab) (0/1 are start/stop)."""
    return prng.randint("text_ids", seed, n, 2, vocab)
