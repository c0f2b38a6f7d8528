"""ORACLE (test infrastructure, not product): CPU fp32 restatement of BigVGAN2 + ECAPA-TDNN + DVAE decode.

Functional torch over `{reference state_dict key: tensor}`; cites /root/reference lines.  See oracle/gpt.py
header for the usage rules.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]


# ---------------------------------------------------------------------------------------------------
# anti-aliased SnakeBeta (Activation1d)
# ---------------------------------------------------------------------------------------------------

def kaiser_sinc_filter12() -> torch.Tensor:
    """alias_free_torch/filter.py:29-58 kaiser_sinc_filter1d(cutoff=0.25, half_width=0.3, kernel_size=12)
    (UpSample1d/DownSample1d with ratio 2, resample.py:17-19,40-43).  Returns [12]."""
    cutoff, half_width, ks = 0.25, 0.3, 12
    half = ks // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(ks, beta=beta, periodic=False)
    time = torch.arange(-half, half) + 0.5
    f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return f / f.sum()


def snakebeta(x, log_alpha, log_beta):
    """activations.py:63-122 SnakeBeta.forward with alpha_logscale: x + sin^2(x*e^a)/(e^b + 1e-9)."""
    a = torch.exp(log_alpha)[None, :, None]
    b = torch.exp(log_beta)[None, :, None]
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def activation1d(x, log_alpha, log_beta, filt=None):
    """alias_free_torch/act.py:24-29 + resample.py:24-33 (UpSample1d: replicate pad 5/5, 2*conv_transpose1d
    stride 2, crop 15/15) + SnakeBeta + filter.py:86-95 (LowPassFilter1d: replicate pad 5/6, stride-2 conv).
    x [B,C,T] -> [B,C,T]."""
    if filt is None:
        filt = kaiser_sinc_filter12()
    C = x.shape[1]
    fw = filt.view(1, 1, 12).expand(C, -1, -1)
    u = F.pad(x, (5, 5), mode="replicate")
    u = 2 * F.conv_transpose1d(u, fw, stride=2, groups=C)
    u = u[..., 15:-15]
    u = snakebeta(u, log_alpha, log_beta)
    u = F.pad(u, (5, 6), mode="replicate")
    return F.conv1d(u, fw, stride=2, groups=C)


# ---------------------------------------------------------------------------------------------------
# ECAPA-TDNN
# ---------------------------------------------------------------------------------------------------

def sb_conv1d(x, w: W, name: str, dilation: int = 1):
    """nnet/CNN.py:411-488 Conv1d.forward with padding='same', padding_mode='reflect' (skip_transpose);
    pad = floor(dilation*(k-1)/2) each side (get_padding_elem :519-546, stride 1)."""
    wt = w[name + ".weight"]
    k = wt.shape[-1]
    pad = (dilation * (k - 1)) // 2
    if pad > 0:
        x = F.pad(x, (pad, pad), mode="reflect")
    return F.conv1d(x, wt, w[name + ".bias"], dilation=dilation)


def bn_eval(x, w: W, name: str, eps: float = 1e-5):
    """nnet/normalization.py:75-108 BatchNorm1d in eval mode (running stats)."""
    return F.batch_norm(x, w[name + ".running_mean"], w[name + ".running_var"], w[name + ".weight"], w[name + ".bias"],
                        False, 0.0, eps)


def tdnn_block(x, w: W, p: str, dilation: int = 1):
    """ECAPA_TDNN.py:79-128 TDNNBlock: norm(relu(conv(x)))."""
    return bn_eval(F.relu(sb_conv1d(x, w, p + "conv.conv", dilation)), w, p + "norm.norm")


def ecapa_tdnn(mel_bfc, w: W, dims, p: str = "speaker_encoder."):
    """ECAPA_TDNN.py:545-581 forward (lengths=None): x [B,F,100] -> [B,1,lin]."""
    x = mel_bfc.transpose(1, 2)
    chs, dil, scale = dims["channels"], dims["dilations"], dims["res2net_scale"]
    xl = []
    x = tdnn_block(x, w, p + "blocks.0.", dil[0])
    xl.append(x)
    for i in range(1, len(chs) - 1):
        q = f"{p}blocks.{i}."
        res = x  # in == out channels: no shortcut conv (:397-402)
        y = tdnn_block(x, w, q + "tdnn1.")
        # Res2NetBlock :172-191
        ys = []
        yi = None
        for j, xi in enumerate(torch.chunk(y, scale, dim=1)):
            if j == 0:
                yi = xi
            elif j == 1:
                yi = tdnn_block(xi, w, f"{q}res2net_block.blocks.{j - 1}.", dil[i])
            else:
                yi = tdnn_block(xi + yi, w, f"{q}res2net_block.blocks.{j - 1}.", dil[i])
            ys.append(yi)
        y = torch.cat(ys, dim=1)
        y = tdnn_block(y, w, q + "tdnn2.")
        # SEBlock :223-242 (lengths None -> plain mean)
        s = y.mean(dim=2, keepdim=True)
        s = F.relu(sb_conv1d(s, w, q + "se_block.conv1.conv"))
        s = torch.sigmoid(sb_conv1d(s, w, q + "se_block.conv2.conv"))
        x = s * y + res
        xl.append(x)
    x = torch.cat(xl[1:], dim=1)
    x = tdnn_block(x, w, p + "mfa.")
    # AttentiveStatisticsPooling :283-338 (global context, full mask)
    L = x.shape[-1]
    eps = 1e-12
    mean = x.mean(dim=2)
    std = torch.sqrt(((x - mean.unsqueeze(2)).pow(2) / L).sum(2).clamp(eps))
    attn_in = torch.cat([x, mean.unsqueeze(2).expand(-1, -1, L), std.unsqueeze(2).expand(-1, -1, L)], dim=1)
    a = sb_conv1d(torch.tanh(tdnn_block(attn_in, w, p + "asp.tdnn.")), w, p + "asp.conv.conv")
    a = F.softmax(a, dim=2)
    mean = (a * x).sum(2)
    std = torch.sqrt((a * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
    pooled = torch.cat((mean, std), dim=1).unsqueeze(2)
    pooled = bn_eval(pooled, w, p + "asp_bn.norm")
    out = sb_conv1d(pooled, w, p + "fc.conv")
    return out.transpose(1, 2)


# ---------------------------------------------------------------------------------------------------
# BigVGAN generator
# ---------------------------------------------------------------------------------------------------

def amp_block1(x, w: W, p: str, ksize: int, dilations, filt):
    """BigVGAN/models.py:65-74 AMPBlock1.forward: 3x { a1 -> conv(dil d) -> a2 -> conv(dil 1) -> +x }."""
    for l, d in enumerate(dilations):
        xt = activation1d(x, w[f"{p}activations.{2 * l}.act.alpha"], w[f"{p}activations.{2 * l}.act.beta"], filt)
        xt = F.conv1d(xt, w[f"{p}convs1.{l}.weight"], w[f"{p}convs1.{l}.bias"], dilation=d, padding=(ksize * d - d) // 2)
        xt = activation1d(xt, w[f"{p}activations.{2 * l + 1}.act.alpha"], w[f"{p}activations.{2 * l + 1}.act.beta"], filt)
        xt = F.conv1d(xt, w[f"{p}convs2.{l}.weight"], w[f"{p}convs2.{l}.bias"], padding=(ksize - 1) // 2)
        x = xt + x
    return x


def bigvgan_generator(latent_btd, spk_b1e, w: W, h, taps: dict = None):
    """BigVGAN/models.py:201-250 BigVGAN.forward given the speaker embedding [B,1,E] (feat_upsample False,
    cond in each up layer): latent [B,T,D] -> wav [B,1,T*prod(up)]."""
    filt = kaiser_sinc_filter12()
    spk = spk_b1e.transpose(1, 2)
    x = latent_btd.transpose(1, 2)
    x = F.conv1d(x, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    x = x + F.conv1d(spk, w["cond_layer.weight"], w["cond_layer.bias"])
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.conv_transpose1d(x, w[f"ups.{i}.0.weight"], w[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        x = x + F.conv1d(spk, w[f"conds.{i}.weight"], w[f"conds.{i}.bias"])
        xs = None
        for j in range(nk):
            r = amp_block1(x, w, f"resblocks.{i * nk + j}.", h["resblock_kernel_sizes"][j],
                           h["resblock_dilation_sizes"][j], filt)
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps[f"stage{i}"] = x
    x = activation1d(x, w["activation_post.act.alpha"], w["activation_post.act.beta"], filt)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    return torch.tanh(x)


def bigvgan_forward(latent_btd, mel_ref_bfc, w: W, h, ecapa):
    """BigVGAN.forward(latent, mel_ref) as infer.py:204 calls it (speaker encoder included)."""
    spk = ecapa_tdnn(mel_ref_bfc, w, ecapa)
    return bigvgan_generator(latent_btd, spk, w, h)


# ---------------------------------------------------------------------------------------------------
# DVAE decode
# ---------------------------------------------------------------------------------------------------

def dvae_decode(codes, w: W, v):
    """vqvae/xtts_dvae.py:332-351 DiscreteVAE.decode (positional_dims 1, use_transposed_convs False):
    embed_code :128-129 -> 1x1 conv -> ResBlocks :171-183 -> UpsampledConv (nearest x2 + conv) :186-196 + ReLU
    -> 1x1 conv.  codes [B,T] -> mel [B,channels,4T]."""
    x = F.embedding(codes.long(), w["codebook.embed"].transpose(0, 1)).transpose(1, 2)
    idx = 0
    x = F.conv1d(x, w[f"decoder.{idx}.weight"], w[f"decoder.{idx}.bias"])
    idx += 1
    for _ in range(v["num_resnet_blocks"]):
        y = F.relu(F.conv1d(x, w[f"decoder.{idx}.net.0.weight"], w[f"decoder.{idx}.net.0.bias"], padding=1))
        y = F.relu(F.conv1d(y, w[f"decoder.{idx}.net.2.weight"], w[f"decoder.{idx}.net.2.bias"], padding=1))
        x = F.conv1d(y, w[f"decoder.{idx}.net.4.weight"], w[f"decoder.{idx}.net.4.bias"]) + x
        idx += 1
    pad = (v["kernel_size"] - 1) // 2
    for _ in range(v["num_layers"]):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.relu(F.conv1d(x, w[f"decoder.{idx}.0.conv.weight"], w[f"decoder.{idx}.0.conv.bias"], padding=pad))
        idx += 1
    return F.conv1d(x, w[f"decoder.{idx}.weight"], w[f"decoder.{idx}.bias"])


def dvae_encode(mel_bct, w: W, v):
    """vqvae/xtts_dvae.py:325-330 DiscreteVAE.get_codebook_indices (positional_dims 1, no normalization): encoder =
    num_layers x (Conv1d(k, stride 2, pad (k-1)//2) + ReLU) -> ResBlocks -> 1x1 conv (:251-291); Quantize.forward :86-89:
    dist = |x|^2 - 2 x E + |E|^2, codes = argmax(-dist).  mel [B,channels,T] -> codes [B,T']."""
    x = mel_bct
    pad = (v["kernel_size"] - 1) // 2
    idx = 0
    for _ in range(v["num_layers"]):
        x = F.relu(F.conv1d(x, w[f"encoder.{idx}.0.weight"], w[f"encoder.{idx}.0.bias"], stride=2, padding=pad))
        idx += 1
    for _ in range(v["num_resnet_blocks"]):
        y = F.relu(F.conv1d(x, w[f"encoder.{idx}.net.0.weight"], w[f"encoder.{idx}.net.0.bias"], padding=1))
        y = F.relu(F.conv1d(y, w[f"encoder.{idx}.net.2.weight"], w[f"encoder.{idx}.net.2.bias"], padding=1))
        x = F.conv1d(y, w[f"encoder.{idx}.net.4.weight"], w[f"encoder.{idx}.net.4.bias"]) + x
        idx += 1
    x = F.conv1d(x, w[f"encoder.{idx}.weight"], w[f"encoder.{idx}.bias"]).permute(0, 2, 1)
    flat = x.reshape(-1, x.shape[-1])
    E = w["codebook.embed"]
    dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ E + E.pow(2).sum(0, keepdim=True)
    return (-dist).max(1)[1].view(*x.shape[:-1])
