"""Weight packer: reference-layout state dicts -> the engine's tensor set.

Ingests what `load_checkpoint(gpt, gpt.pth)` / `bigvgan_generator.pth["generator"]` / `dvae.pth` hold
(SURVEY.md section 5 key layout; infer.py:47-66) and produces the layouts the HIP kernels stream:

* every linear / conv weight as [N][taps][Cin] (k-contiguous rows; HF Conv1D [in,out] is transposed,
  torch Conv1d [Cout,Cin,k] is tap-major re-ordered, ConvTranspose1d [Cin,Cout,k] becomes `u` polyphase
  slabs [u][Cout][k/u][Cin]),
* weight-norm pairs (weight_g / weight_v) folded (what `remove_weight_norm` does, models.py:252-260),
* eval-mode BatchNorm folded to a per-channel (scale, shift) applied after the ReLU (ECAPA_TDNN.py:128),
* conformer q/k/v projections fused, the ASP context conv split into its x / (mean,std) column blocks,
* the perceiver FFN's odd inner width zero-padded to a multiple of 32,
* SnakeBeta alpha/beta kept LOG-scale fp32 (the native-op contract, cuda/activation1d.py:60-71).

Values tagged "w" are stored in the engine dtype (fp32 parity path or bf16), "f" always fp32.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .config import ecapa_dims, perceiver_inner

Packed = Dict[str, Tuple[str, np.ndarray]]


def _fold_weight_norm(sd: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """weight = g * v / ||v|| with the norm over all dims but 0 (torch weight_norm dim=0)."""
    out = dict(sd)
    for k in list(sd):
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            g, v = sd[k], sd[base + ".weight_v"]
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
            out[base + ".weight"] = (v * (g / nrm)).astype(np.float32)
            del out[k], out[base + ".weight_v"]
    return out


def conv_w(w: np.ndarray) -> np.ndarray:
    """torch Conv1d [Cout, Cin, k] -> [Cout, k*Cin] (tap-major)."""
    co, ci, k = w.shape
    return np.ascontiguousarray(w.transpose(0, 2, 1)).reshape(co, k * ci)


def convT_w(w: np.ndarray, u: int, p: int) -> np.ndarray:
    """torch ConvTranspose1d [Cin, Cout, k] -> [u, Cout, (k/u)*Cin]; phase ph tap m uses kernel index
    ((ph+p) % u) + u*m and input offset floor((ph+p)/u) - m."""
    ci, co, k = w.shape
    nt = k // u
    out = np.zeros((u, co, nt * ci), dtype=np.float32)
    for ph in range(u):
        for m in range(nt):
            kk = (ph + p) % u + u * m
            out[ph, :, m * ci:(m + 1) * ci] = w[:, :, kk].T
    return out


def fold_ln(w: np.ndarray, b: np.ndarray, gamma: np.ndarray, beta: np.ndarray):
    """(W, c) applied to LayerNorm output g*z + b  ->  (W diag(g), W b + c) applied to the plain normalised z."""
    w64 = w.astype(np.float64)
    return (w64 * gamma.astype(np.float64)[None, :]).astype(np.float32), (b.astype(np.float64) + w64 @ beta.astype(np.float64)).astype(np.float32)


def bn_fold(sd, name: str, eps: float = 1e-5):
    sc = sd[name + ".weight"] / np.sqrt(sd[name + ".running_var"] + eps)
    sh = sd[name + ".bias"] - sd[name + ".running_mean"] * sc
    return sc.astype(np.float32), sh.astype(np.float32)


def pack_gpt(sd: Dict[str, np.ndarray], cfg) -> Packed:
    g = cfg["gpt"]
    cm = g["condition_module"]
    P: Packed = {}
    ce = "conditioning_encoder."
    od = cm["output_size"]
    P["cond.embed.conv.weight"] = ("f", sd[ce + "embed.conv.0.weight"].reshape(od, 9))
    P["cond.embed.conv.bias"] = ("f", sd[ce + "embed.conv.0.bias"])
    P["cond.embed.out.weight"] = ("w", sd[ce + "embed.out.0.weight"])
    P["cond.embed.out.bias"] = ("f", sd[ce + "embed.out.0.bias"])
    P["cond.pe"] = ("w", sd[ce + "embed.pos_enc.pe"][0])
    for n in ("weight", "bias"):
        P[f"cond.after_norm.{n}"] = ("f", sd[f"{ce}after_norm.{n}"])
    for i in range(cm["num_blocks"]):
        s, d = f"{ce}encoders.{i}.", f"cond.{i}."
        a = s + "self_attn."
        P[d + "qkv.weight"] = ("w", np.concatenate([sd[a + f"linear_{x}.weight"] for x in "qkv"], 0))
        P[d + "qkv.bias"] = ("f", np.concatenate([sd[a + f"linear_{x}.bias"] for x in "qkv"], 0))
        P[d + "pos.weight"] = ("w", sd[a + "linear_pos.weight"])
        P[d + "pos_bias_u"] = ("f", sd[a + "pos_bias_u"].reshape(-1))
        P[d + "pos_bias_v"] = ("f", sd[a + "pos_bias_v"].reshape(-1))
        P[d + "out.weight"] = ("w", sd[a + "linear_out.weight"])
        P[d + "out.bias"] = ("f", sd[a + "linear_out.bias"])
        P[d + "ff.w1.weight"] = ("w", sd[s + "feed_forward.w_1.weight"])
        P[d + "ff.w1.bias"] = ("f", sd[s + "feed_forward.w_1.bias"])
        P[d + "ff.w2.weight"] = ("w", sd[s + "feed_forward.w_2.weight"])
        P[d + "ff.w2.bias"] = ("f", sd[s + "feed_forward.w_2.bias"])
        c = s + "conv_module."
        P[d + "conv.pw1.weight"] = ("w", sd[c + "pointwise_conv1.weight"][:, :, 0])
        P[d + "conv.pw1.bias"] = ("f", sd[c + "pointwise_conv1.bias"])
        P[d + "conv.dw.weight"] = ("f", sd[c + "depthwise_conv.weight"][:, 0, :])
        P[d + "conv.dw.bias"] = ("f", sd[c + "depthwise_conv.bias"])
        P[d + "conv.norm.weight"] = ("f", sd[c + "norm.weight"])
        P[d + "conv.norm.bias"] = ("f", sd[c + "norm.bias"])
        P[d + "conv.pw2.weight"] = ("w", sd[c + "pointwise_conv2.weight"][:, :, 0])
        P[d + "conv.pw2.bias"] = ("f", sd[c + "pointwise_conv2.bias"])
        for nn in ("norm_ff", "norm_mha", "norm_conv", "norm_final"):
            P[d + nn + ".weight"] = ("f", sd[s + nn + ".weight"])
            P[d + nn + ".bias"] = ("f", sd[s + nn + ".bias"])
    pe = "perceiver_encoder."
    P["perc.latents"] = ("f", sd[pe + "latents"])
    P["perc.proj.weight"] = ("w", sd[pe + "proj_context.weight"])
    P["perc.proj.bias"] = ("f", sd[pe + "proj_context.bias"])
    ffi = perceiver_inner(g)
    ffp = (ffi + 31) // 32 * 32
    for j in range(2):
        s, d = f"{pe}layers.{j}.", f"perc.{j}."
        P[d + "to_q.weight"] = ("w", sd[s + "0.to_q.weight"])
        P[d + "to_kv.weight"] = ("w", sd[s + "0.to_kv.weight"])
        P[d + "to_out.weight"] = ("w", sd[s + "0.to_out.weight"])
        P[d + "ff1.weight"] = ("w", sd[s + "1.0.weight"])
        P[d + "ff1.bias"] = ("f", sd[s + "1.0.bias"])
        w2 = np.zeros((sd[s + "1.2.weight"].shape[0], ffp), dtype=np.float32)
        w2[:, :ffi] = sd[s + "1.2.weight"]
        P[d + "ff2.weight"] = ("w", w2)
        P[d + "ff2.bias"] = ("f", sd[s + "1.2.bias"])
    P["perc.norm.gamma"] = ("f", sd[pe + "norm.gamma"])
    # GPT-2 stack (inference_model.* aliases of a post-init state dict are ignored: SURVEY 3.1 step 3).
    # ln_1 / ln_2 affine parameters are folded into c_attn / c_fc:  W (g*z + b) + c = (W diag(g)) z + (W b + c), so the
    # decode GEMV only normalises (no gamma/beta traffic in the per-token loop); same fold for final_norm -> mel_head.
    for i in range(g["layers"]):
        s = f"gpt.h.{i}."
        for nn, ln in (("attn.c_attn", "ln_1"), ("attn.c_proj", None), ("mlp.c_fc", "ln_2"), ("mlp.c_proj", None)):
            w = np.ascontiguousarray(sd[s + nn + ".weight"].T)  # [out, in]
            b = sd[s + nn + ".bias"]
            if ln is not None:
                w, b = fold_ln(w, b, sd[s + ln + ".weight"], sd[s + ln + ".bias"])
            P[s + nn + ".weight"] = ("w", w)
            P[s + nn + ".bias"] = ("f", b)
    for nn in ("weight", "bias"):
        P["gpt.ln_f." + nn] = ("f", sd["gpt.ln_f." + nn])
        P["gpt.final_norm." + nn] = ("f", sd["final_norm." + nn])
    hw, hb = fold_ln(sd["mel_head.weight"], sd["mel_head.bias"], sd["final_norm.weight"], sd["final_norm.bias"])
    P["gpt.mel_head.weight"] = ("w", hw)
    P["gpt.mel_head.bias"] = ("f", hb)
    P["gpt.text_embedding"] = ("w", sd["text_embedding.weight"])
    P["gpt.mel_embedding"] = ("w", sd["mel_embedding.weight"])
    P["gpt.mel_pos"] = ("w", sd["mel_pos_embedding.emb.weight"])
    P["gpt.text_pos"] = ("w", sd["text_pos_embedding.emb.weight"])
    return P


def _filter12() -> np.ndarray:
    """kaiser_sinc_filter1d(cutoff 0.25, half_width 0.3, 12 taps) (alias_free_torch/filter.py:29-58), computed
    in float64 numpy (np.i0 Kaiser window) and rounded once to fp32."""
    ks, half = 12, 6
    A = 2.285 * (half - 1) * np.pi * (4 * 0.3) + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    n = np.arange(ks, dtype=np.float64)
    window = np.i0(beta * np.sqrt(1 - ((n - (ks - 1) / 2) / ((ks - 1) / 2)) ** 2)) / np.i0(beta)
    time = np.arange(-half, half, dtype=np.float64) + 0.5
    f = 2 * 0.25 * window * np.sinc(2 * 0.25 * time)
    return (f / f.sum()).astype(np.float32)


def pack_bigvgan(sd: Dict[str, np.ndarray], cfg) -> Packed:
    h = cfg["bigvgan"]
    sd = _fold_weight_norm(sd)
    P: Packed = {}
    C0 = h["upsample_initial_channel"]
    P["bv.conv_pre.weight"] = ("w", conv_w(sd["conv_pre.weight"]))
    P["bv.conv_pre.bias"] = ("f", sd["conv_pre.bias"])
    P["bv.cond_layer.weight"] = ("f", sd["cond_layer.weight"][:, :, 0])
    P["bv.cond_layer.bias"] = ("f", sd["cond_layer.bias"])
    nk = len(h["resblock_kernel_sizes"])
    ch = C0
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        P[f"bv.ups.{i}.weight"] = ("w", convT_w(sd[f"ups.{i}.0.weight"], u, (k - u) // 2))
        P[f"bv.ups.{i}.bias"] = ("f", sd[f"ups.{i}.0.bias"])
        P[f"bv.conds.{i}.weight"] = ("f", sd[f"conds.{i}.weight"][:, :, 0])
        P[f"bv.conds.{i}.bias"] = ("f", sd[f"conds.{i}.bias"])
        ch = C0 // (2 ** (i + 1))
        for j in range(nk):
            n = i * nk + j
            nd = len(h["resblock_dilation_sizes"][j])
            for l in range(nd):
                P[f"bv.res.{n}.c1.{l}.weight"] = ("w", conv_w(sd[f"resblocks.{n}.convs1.{l}.weight"]))
                P[f"bv.res.{n}.c1.{l}.bias"] = ("f", sd[f"resblocks.{n}.convs1.{l}.bias"])
                P[f"bv.res.{n}.c2.{l}.weight"] = ("w", conv_w(sd[f"resblocks.{n}.convs2.{l}.weight"]))
                P[f"bv.res.{n}.c2.{l}.bias"] = ("f", sd[f"resblocks.{n}.convs2.{l}.bias"])
            for m in range(2 * nd):
                P[f"bv.res.{n}.act.{m}.alpha"] = ("f", sd[f"resblocks.{n}.activations.{m}.act.alpha"])
                P[f"bv.res.{n}.act.{m}.beta"] = ("f", sd[f"resblocks.{n}.activations.{m}.act.beta"])
    P["bv.act_post.alpha"] = ("f", sd["activation_post.act.alpha"])
    P["bv.act_post.beta"] = ("f", sd["activation_post.act.beta"])
    P["bv.conv_post.weight"] = ("w", conv_w(sd["conv_post.weight"]))
    P["bv.conv_post.bias"] = ("f", sd["conv_post.bias"])
    fk = "activation_post.upsample.filter"
    P["bv.filter"] = ("f", sd[fk].reshape(-1).astype(np.float32) if fk in sd else _filter12())
    # ---- ECAPA-TDNN ----
    e = ecapa_dims(h)
    chs = e["channels"]
    s = "speaker_encoder."

    def tdnn(dst, src):
        w = np.asarray(sd[src + ".conv.conv.weight"], dtype=np.float32)
        if w.shape[1] % 8:  # the first block reads the 100-bin mel: input channels zero-padded to a multiple of 8 so the
            wp = np.zeros((w.shape[0], (w.shape[1] + 7) // 8 * 8, w.shape[2]), dtype=np.float32)  # conv runs on MFMA
            wp[:, : w.shape[1]] = w
            w = wp
        P[dst + ".weight"] = ("w", conv_w(w))
        P[dst + ".bias"] = ("f", sd[src + ".conv.conv.bias"])
        sc, sh = bn_fold(sd, src + ".norm.norm")
        P[dst + ".bn_scale"], P[dst + ".bn_shift"] = ("f", sc), ("f", sh)

    tdnn("spk.b0", s + "blocks.0")
    for i in range(1, len(chs) - 1):
        p = f"{s}blocks.{i}."
        tdnn(f"spk.b{i}.tdnn1", p + "tdnn1")
        for q in range(e["res2net_scale"] - 1):
            tdnn(f"spk.b{i}.res.{q}", f"{p}res2net_block.blocks.{q}")
        tdnn(f"spk.b{i}.tdnn2", p + "tdnn2")
        P[f"spk.b{i}.se1.weight"] = ("f", sd[p + "se_block.conv1.conv.weight"][:, :, 0])
        P[f"spk.b{i}.se1.bias"] = ("f", sd[p + "se_block.conv1.conv.bias"])
        P[f"spk.b{i}.se2.weight"] = ("f", sd[p + "se_block.conv2.conv.weight"][:, :, 0])
        P[f"spk.b{i}.se2.bias"] = ("f", sd[p + "se_block.conv2.conv.bias"])
    tdnn("spk.mfa", s + "mfa")
    wa = sd[s + "asp.tdnn.conv.conv.weight"][:, :, 0]
    C4 = chs[-1]
    P["spk.asp.tdnn_x.weight"] = ("w", np.ascontiguousarray(wa[:, :C4]))
    P["spk.asp.tdnn_ms.weight"] = ("f", np.ascontiguousarray(wa[:, C4:]))
    P["spk.asp.tdnn_ms.bias"] = ("f", sd[s + "asp.tdnn.conv.conv.bias"])
    sc, sh = bn_fold(sd, s + "asp.tdnn.norm.norm")
    P["spk.asp.tdnn.bn_scale"], P["spk.asp.tdnn.bn_shift"] = ("f", sc), ("f", sh)
    P["spk.asp.conv.weight"] = ("w", sd[s + "asp.conv.conv.weight"][:, :, 0])
    P["spk.asp.conv.bias"] = ("f", sd[s + "asp.conv.conv.bias"])
    sc, sh = bn_fold(sd, s + "asp_bn.norm")
    P["spk.asp_bn.scale"], P["spk.asp_bn.shift"] = ("f", sc), ("f", sh)
    P["spk.fc.weight"] = ("f", sd[s + "fc.conv.weight"][:, :, 0])
    P["spk.fc.bias"] = ("f", sd[s + "fc.conv.bias"])
    return P


def pack_dvae(sd: Dict[str, np.ndarray], cfg) -> Packed:
    v = cfg["vqvae"]
    P: Packed = {}
    P["dvae.codebook"] = ("w", np.ascontiguousarray(sd["codebook.embed"].T))
    idx = 0
    P["dvae.in.weight"] = ("w", conv_w(sd[f"decoder.{idx}.weight"]))
    P["dvae.in.bias"] = ("f", sd[f"decoder.{idx}.bias"])
    idx += 1
    for i in range(v["num_resnet_blocks"]):
        for a, b in (("c0", 0), ("c2", 2), ("c4", 4)):
            P[f"dvae.rb{i}.{a}.weight"] = ("w", conv_w(sd[f"decoder.{idx}.net.{b}.weight"]))
            P[f"dvae.rb{i}.{a}.bias"] = ("f", sd[f"decoder.{idx}.net.{b}.bias"])
        idx += 1
    for i in range(v["num_layers"]):
        P[f"dvae.up{i}.weight"] = ("w", conv_w(sd[f"decoder.{idx}.0.conv.weight"]))
        P[f"dvae.up{i}.bias"] = ("f", sd[f"decoder.{idx}.0.conv.bias"])
        idx += 1
    P["dvae.out.weight"] = ("w", conv_w(sd[f"decoder.{idx}.weight"]))
    P["dvae.out.bias"] = ("f", sd[f"decoder.{idx}.bias"])
    if "encoder.0.0.weight" in sd:  # get_codebook_indices (xtts_dvae.py:325-330); absent from decoder-only checkpoints
        assert v["kernel_size"] == 3 and v.get("stride", 2) == 2, "DVAE encoder packer: kernel 3 / stride 2 only"
        idx = 0
        for i in range(v["num_layers"]):
            # Conv1d(k=3, stride=2, pad=1) == a 2-tap stride-1 conv over PAIRED input rows [x[2t], x[2t+1]] (2*Cin channels,
            # one zero pair of left padding): out[t] = w0 x[2t-1] + w1 x[2t] + w2 x[2t+1]
            w = sd[f"encoder.{idx}.0.weight"]
            co, ci, _ = w.shape
            w2 = np.zeros((co, 2, 2 * ci), dtype=np.float32)
            w2[:, 0, ci:] = w[:, :, 0]
            w2[:, 1, :ci] = w[:, :, 1]
            w2[:, 1, ci:] = w[:, :, 2]
            P[f"dvae.enc{i}.weight"] = ("w", w2.reshape(co, 4 * ci))
            P[f"dvae.enc{i}.bias"] = ("f", sd[f"encoder.{idx}.0.bias"])
            idx += 1
        for i in range(v["num_resnet_blocks"]):
            for a, b in (("c0", 0), ("c2", 2), ("c4", 4)):
                P[f"dvae.erb{i}.{a}.weight"] = ("w", conv_w(sd[f"encoder.{idx}.net.{b}.weight"]))
                P[f"dvae.erb{i}.{a}.bias"] = ("f", sd[f"encoder.{idx}.net.{b}.bias"])
            idx += 1
        P["dvae.eout.weight"] = ("w", conv_w(sd[f"encoder.{idx}.weight"]))
        P["dvae.eout.bias"] = ("f", sd[f"encoder.{idx}.bias"])
        P["dvae.codebook_sq"] = ("f", (sd["codebook.embed"].astype(np.float32) ** 2).sum(0))  # |e_n|^2 of Quantize.forward's distance
    return P


def quantize_gpt_fp8(packed, keep_bytes: bool = True):
    """GPT projection weights -> OCP fp8 e4m3 with one power-of-two scale per output row (BASELINE config 5,
    SURVEY 8d "fp8-e4m3 GPT weights (per-output-channel scale), bf16 KV/activations").

    Every `gpt.h.*.{attn.c_attn,attn.c_proj,mlp.c_fc,mlp.c_proj}.weight` and `gpt.mel_head.weight` [N, K] is replaced by
    its dequantisation q * 2^e (exactly representable in bf16, so prefill / latent pass and the fp8 decode GEMV see the
    same model), and - with keep_bytes - the fp8 bytes `<name>_fp8` (tag "q") + `<name>_scale` fp32 [N] are added for
    the decode step.  Rounding is torch's float8_e4m3fn cast (round-to-nearest-even, saturating at 448)."""
    import re

    import torch

    out = dict(packed)
    pat = re.compile(r"^gpt\.(h\.\d+\.(attn\.c_attn|attn\.c_proj|mlp\.c_fc|mlp\.c_proj)|mel_head)\.weight$")
    for name, (tag, arr) in packed.items():
        if not pat.match(name):
            continue
        w = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
        amax = w.abs().amax(dim=1).clamp_min(1e-30)
        e = torch.ceil(torch.log2(amax / 448.0))
        scale = torch.pow(2.0, e)
        q = (w / scale[:, None]).to(torch.float8_e4m3fn)
        out[name] = (tag, (q.float() * scale[:, None]).numpy())
        if keep_bytes:
            out[name + "_fp8"] = ("q", q.view(torch.uint8).numpy())
            out[name + "_scale"] = ("f", scale.numpy().astype(np.float32))
    return out
