"""ORACLE (test infrastructure, not product): CPU fp32 restatement of the reference's acoustic-LM path.

Plain functional torch on CPU over a `{reference state_dict key: tensor}` mapping; no nn.Module of the
reference is used or copied.  Every function names the reference lines it follows (paths relative to
/root/reference).  Pinned against the real reference modules by `oracle/make_golden.py` ->
`tests/golden/*.npz` (see tests/test_oracle_golden.py).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this package.
"""
from __future__ import annotations

import math

import numpy as np
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

W = Dict[str, torch.Tensor]


def to_torch(sd) -> W:
    return {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in sd.items()}


def layer_norm(x, w: W, name: str, eps: float = 1e-5):
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], eps)


def linear(x, w: W, name: str):
    return F.linear(x, w[name + ".weight"], w.get(name + ".bias"))


# ---------------------------------------------------------------------------------------------------
# Conformer conditioning encoder
# ---------------------------------------------------------------------------------------------------

def conv2d_subsampling2(x, w: W, p: str):
    """indextts/gpt/conformer/subsampling.py:135-186 (Conv2dSubsampling2): Conv2d(1,odim,3,stride 2)+ReLU,
    flatten (c,f) per frame, Linear; then RelPositionalEncoding (embedding.py:117-143): x*sqrt(d), pe[:T']."""
    x = x.unsqueeze(1)  # [b,1,t,f]
    x = F.relu(F.conv2d(x, w[p + "conv.0.weight"], w[p + "conv.0.bias"], stride=2))
    b, c, t, f = x.shape
    x = linear(x.transpose(1, 2).contiguous().view(b, t, c * f), w, p + "out.0")
    d = x.shape[-1]
    x = x * math.sqrt(d)
    pos_emb = w[p + "pos_enc.pe"][:, :t]
    return x, pos_emb


def rel_pos_mha(x, pos_emb, w: W, p: str, heads: int):
    """attention.py:235-312 RelPositionMultiHeadedAttention.forward with forward_qkv :48-75 and
    forward_attention :77-120 (full mask): scores = ((q+u)k^T + (q+v)p^T)/sqrt(dk), NO rel-shift (:305-307)."""
    b, t, d = x.shape
    dk = d // heads
    q = linear(x, w, p + "linear_q").view(b, t, heads, dk)
    k = linear(x, w, p + "linear_k").view(b, t, heads, dk).transpose(1, 2)
    v = linear(x, w, p + "linear_v").view(b, t, heads, dk).transpose(1, 2)
    pp = F.linear(pos_emb, w[p + "linear_pos.weight"]).view(pos_emb.shape[0], -1, heads, dk).transpose(1, 2)
    qu = (q + w[p + "pos_bias_u"]).transpose(1, 2)
    qv = (q + w[p + "pos_bias_v"]).transpose(1, 2)
    scores = (qu @ k.transpose(-2, -1) + qv @ pp.transpose(-2, -1)) / math.sqrt(dk)
    attn = torch.softmax(scores, dim=-1)
    o = (attn @ v).transpose(1, 2).contiguous().view(b, t, d)
    return linear(o, w, p + "linear_out")


def conformer_conv_module(x, w: W, p: str, tail: int = 0):
    """conformer_encoder.py:112-167 ConvolutionModule.forward (non-causal, LayerNorm variant, SiLU).  tail > 0: the sequence is
    the valid part of a padded one; the module zero-fills the masked rows BEFORE its first pointwise convolution (:143-146), so
    its depthwise convolution sees GLU(pw1 bias) behind the end - `tail` such rows are appended and dropped again."""
    T = x.shape[1]
    if tail > 0:
        x = F.pad(x, (0, 0, 0, tail))
    x = x.transpose(1, 2)
    x = F.conv1d(x, w[p + "pointwise_conv1.weight"], w[p + "pointwise_conv1.bias"])
    x = F.glu(x, dim=1)
    c = x.shape[1]
    k = w[p + "depthwise_conv.weight"].shape[-1]
    x = F.conv1d(x, w[p + "depthwise_conv.weight"], w[p + "depthwise_conv.bias"], padding=(k - 1) // 2, groups=c)[:, :, :T]
    x = F.silu(layer_norm(x.transpose(1, 2), w, p + "norm")).transpose(1, 2)
    x = F.conv1d(x, w[p + "pointwise_conv2.weight"], w[p + "pointwise_conv2.bias"])
    return x.transpose(1, 2)


def conformer_layer(x, pos_emb, w: W, p: str, heads: int, tail: int = 0):
    """conformer_encoder.py:232-313 ConformerEncoderLayer.forward (normalize_before, no macaron, ff_scale 1)."""
    x = x + rel_pos_mha(layer_norm(x, w, p + "norm_mha"), pos_emb, w, p + "self_attn.", heads)
    x = x + conformer_conv_module(layer_norm(x, w, p + "norm_conv"), w, p + "conv_module.", tail)
    y = layer_norm(x, w, p + "norm_ff")
    y = linear(F.silu(linear(y, w, p + "feed_forward.w_1")), w, p + "feed_forward.w_2")  # :20-53
    x = x + y
    return layer_norm(x, w, p + "norm_final")


def conformer_encoder(mel_btf, w: W, cfg_gpt, total_frames: int = 0) -> torch.Tensor:
    """conformer_encoder.py:400-436 BaseEncoder.forward; input [b, F, 100] -> [b, F', od].  total_frames > F: the input is the
    valid part of a prompt padded to total_frames (xs_lens = F): the subsampled mask (subsampling.py:186) keeps exactly the F'
    rows computed here, masked keys are absent keys; only the convolution module sees the masked rows (conformer_conv_module)."""
    cm = cfg_gpt["condition_module"]
    p = "conditioning_encoder."
    x, pos = conv2d_subsampling2(mel_btf, w, p + "embed.")
    tail = min(7, ((total_frames - 3) // 2 + 1) - x.shape[1]) if total_frames > mel_btf.shape[1] else 0
    for i in range(cm["num_blocks"]):
        x = conformer_layer(x, pos, w, f"{p}encoders.{i}.", cm["attention_heads"], tail)
    return layer_norm(x, w, p + "after_norm")


# ---------------------------------------------------------------------------------------------------
# Perceiver resampler
# ---------------------------------------------------------------------------------------------------

def perceiver(ctx, w: W, cfg_gpt) -> torch.Tensor:
    """perceiver.py:263-274 PerceiverResampler.forward; Attention :277-317 with
    cross_attn_include_queries (kv = cat(latents, ctx)); Attend non-flash :107-150; GEGLU FFN :204-221;
    RMSNorm :167-186.  Mask is all-true for a single full-length prompt (model.py:343,501)."""
    p = "perceiver_encoder."
    heads = cfg_gpt["condition_module"]["attention_heads"]
    b = ctx.shape[0]
    x = linear(ctx, w, p + "proj_context")
    lat = w[p + "latents"].unsqueeze(0).expand(b, -1, -1)
    for j in range(2):
        q_ = f"{p}layers.{j}."
        kvsrc = torch.cat((lat, x), dim=-2)
        q = F.linear(lat, w[q_ + "0.to_q.weight"])
        k, v = F.linear(kvsrc, w[q_ + "0.to_kv.weight"]).chunk(2, dim=-1)
        dh = q.shape[-1] // heads

        def sp(t):
            return t.view(b, -1, heads, dh).transpose(1, 2)

        q, k, v = sp(q), sp(k), sp(v)
        sim = (q @ k.transpose(-2, -1)) * (dh ** -0.5)
        o = (sim.softmax(dim=-1) @ v).transpose(1, 2).reshape(b, -1, heads * dh)
        lat = F.linear(o, w[q_ + "0.to_out.weight"]) + lat
        hcat = linear(lat, w, q_ + "1.0")
        xx, gate = hcat.chunk(2, dim=-1)
        lat = linear(F.gelu(gate) * xx, w, q_ + "1.2") + lat
    d = lat.shape[-1]
    return F.normalize(lat, dim=-1) * (d ** 0.5) * w[p + "norm.gamma"]


def get_conditioning(mel_bcf, w: W, cfg_gpt, length: Optional[int] = None) -> torch.Tensor:
    """model.py:490-502 get_conditioning ('conformer_perceiver'): [b,100,F] -> [b,32,D]; length < F = cond_mel_lengths of a
    padded prompt (masked conformer + perceiver)."""
    F_ = mel_bcf.shape[-1]
    if length is not None and int(length) < F_:
        return perceiver(conformer_encoder(mel_bcf[..., : int(length)].transpose(1, 2), w, cfg_gpt, total_frames=F_), w, cfg_gpt)
    return perceiver(conformer_encoder(mel_bcf.transpose(1, 2), w, cfg_gpt), w, cfg_gpt)


# ---------------------------------------------------------------------------------------------------
# GPT-2 stack (transformers==4.36.2 GPT2Model eager math; see SURVEY 8c 'Third-party arithmetic')
# ---------------------------------------------------------------------------------------------------

def gelu_new(x):
    """HF NewGELUActivation: 0.5x(1+tanh(sqrt(2/pi)(x+0.044715x^3)))."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def gpt2_stack(emb, w: W, cfg_gpt, key_mask: Optional[torch.Tensor] = None,
               past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None):
    """GPT2Model.forward with inputs_embeds (wpe == 0: model.py:17-18,268-270), pre-LN blocks, causal mask
    + additive (1-mask)*finfo.min key mask, KV cache; returns (ln_f(hidden), new_past).
    emb [b,n,D]; key_mask [b, past+n] of 0/1 or None."""
    D, H, NL = cfg_gpt["model_dim"], cfg_gpt["heads"], cfg_gpt["layers"]
    dh = D // H
    b, n, _ = emb.shape
    pl = 0 if past is None else past[0][0].shape[2]
    S = pl + n
    neg = torch.finfo(emb.dtype).min
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))[pl:S, :S]
    bias = torch.zeros(b, 1, n, S, dtype=emb.dtype)
    if key_mask is not None:
        bias = bias + (1.0 - key_mask[:, None, None, :].to(emb.dtype)) * neg
    x = emb
    new_past = []
    for i in range(NL):
        p = f"gpt.h.{i}."
        hN = layer_norm(x, w, p + "ln_1")
        qkv = hN @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]  # HF Conv1D: [in,out]
        q, k, v = qkv.split(D, dim=2)

        def sp(t):
            return t.view(b, -1, H, dh).transpose(1, 2)

        q, k, v = sp(q), sp(k), sp(v)
        if past is not None:
            k = torch.cat((past[i][0], k), dim=2)
            v = torch.cat((past[i][1], v), dim=2)
        new_past.append((k, v))
        att = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
        att = torch.where(causal[None, None], att, torch.full_like(att, neg)) + bias
        att = torch.softmax(att, dim=-1)
        o = (att @ v).transpose(1, 2).contiguous().view(b, n, D)
        x = x + (o @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"])
        hN = layer_norm(x, w, p + "ln_2")
        m = gelu_new(hN @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
        x = x + (m @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"])
    return layer_norm(x, w, "gpt.ln_f"), new_past


def prepare_gpt_inputs(cond, text_inputs, w: W, cfg_gpt):
    """model.py:591-654: strip start/stop ids, wrap [0]+text+[1], text_emb+text_pos, LEFT-pad with zero
    vectors, mask 0 on pad; fake ids are all 1 with last = start_mel_token."""
    start_t, stop_t = cfg_gpt["start_text_token"], cfg_gpt["stop_text_token"]
    b, L = text_inputs.shape
    single = cond.shape[0] == 1
    target = cond.shape[1] + L + 2
    embs, masks = [], []
    for i in range(b):
        ti = text_inputs[i]
        ti = ti[(ti != stop_t) & (ti != start_t)]
        ti = F.pad(F.pad(ti, (1, 0), value=start_t), (0, 1), value=stop_t).long()
        te = w["text_embedding.weight"][ti] + w["text_pos_embedding.emb.weight"][: ti.shape[0]]
        parts = [cond[0] if single else cond[i], te]
        am = torch.ones(target + 1, dtype=torch.long)
        pad = L + 2 - ti.shape[0]
        if pad > 0:
            parts.insert(0, torch.zeros(pad, cond.shape[-1], dtype=te.dtype))
            am[:pad] = 0
        embs.append(torch.cat(parts))
        masks.append(am)
    emb = torch.stack(embs)
    fake = torch.ones(b, target + 1, dtype=torch.long)
    fake[:, -1] = cfg_gpt["start_mel_token"]
    return fake, emb, torch.stack(masks)


def lm_head(h, w: W):
    """GPT2InferenceModel.lm_head = Sequential(final_norm, mel_head) (model.py:48,180)."""
    return linear(layer_norm(h, w, "final_norm"), w, "mel_head")


def repetition_penalty_(scores, ids, penalty: float):
    """HF RepetitionPenaltyLogitsProcessor.__call__: gather at every id in input_ids, <0 -> *p, else /p, scatter."""
    sc = torch.gather(scores, 1, ids)
    sc = torch.where(sc < 0, sc * penalty, sc / penalty)
    return scores.scatter(1, ids, sc)


def sample_distribution(scores, top_k: int, top_p: float, temperature: float):
    """HF 4.36.2 logits warpers as GenerationMixin.sample applies them for infer.py:116-124 (TemperatureLogitsWarper ->
    TopKLogitsWarper(min_tokens_to_keep=1) -> TopPLogitsWarper(min_tokens_to_keep=1) -> softmax), restated on one fp32 row
    in the arithmetic order the HIP sampler uses.  Returns (token ids in descending-score order, un-normalised weights
    e_r = exp(s_r - s_0) of the kept tokens)."""
    import numpy as np

    s = np.asarray(scores, dtype=np.float32).copy()
    V = s.shape[0]
    if temperature != 1.0:
        s = (s / np.float32(temperature)).astype(np.float32)
    k = min(int(top_k), V)
    kth = np.partition(s, V - k)[V - k]  # TopK: keeps everything >= the k-th largest (ties stay)
    idx = np.nonzero(s >= kth)[0]
    order = np.lexsort((idx, -s[idx].astype(np.float64)))
    idx = idx[order][:128]
    v = s[idx]
    e = np.exp((v - v[0]).astype(np.float32)).astype(np.float32)
    Z = np.float32(0)
    for x in e:
        Z = np.float32(Z + x)
    R = len(idx)
    if top_p < 1.0:
        # TopP: ascending cumulative probability <= 1 - top_p is removed, the best token always stays
        tail, R = np.float32(0), 1
        lim = np.float32(1.0) - np.float32(top_p)
        for r in range(len(idx) - 1, 0, -1):
            tail = np.float32(tail + np.float32(e[r] / Z))
            if not tail <= lim:
                R = r + 1
                break
    return idx[:R], e[:R]


def sample_pick(scores, top_k: int, top_p: float, temperature: float, u: float) -> int:
    """One multinomial draw as an inverse-CDF lookup of the uniform u over the kept tokens (descending-score order)."""
    import numpy as np

    idx, e = sample_distribution(scores, top_k, top_p, temperature)
    total = np.float32(0)
    for x in e:
        total = np.float32(total + x)
    target = np.float32(np.float32(u) * total)
    c = np.float32(0)
    for r, x in enumerate(e):
        c = np.float32(c + x)
        if c >= target:
            return int(idx[r])
    return int(idx[-1])


def greedy_generate(cond, text_inputs, w: W, cfg_gpt, max_generate_length: int, repetition_penalty: float = 10.0,
                    suppress_eos: bool = False, trace: Optional[dict] = None, sampling: Optional[dict] = None,
                    input_tokens: Optional[torch.Tensor] = None, typical_mass: float = 0.0):
    """UnifiedVoice.inference_speech (model.py:655-708) with HF 4.36.2 `generate` greedy_search semantics
    (do_sample False, num_beams 1; eos=pad=stop_mel_token; MaxLengthCriteria): hand-rolled because the
    installed transformers 5.x `generate` skips the prefill (SURVEY 8c 'Critical caveat').

    Step 0 feeds cat(prefix, mel_emb(start)+mel_pos[0]) (model.py:139-150); step k>=1 feeds
    mel_emb(tok)+mel_pos[mask_len - s] -> positions 0,2,3,4,... (model.py:151-155).
    Returns codes [b, <=max_generate_length] (prefix stripped, model.py:704-705).

    sampling = {"top_k", "top_p", "temperature", "uniforms" [max_gen, b]} switches the pick to GenerationMixin.sample
    (do_sample=True, num_beams=1) with the draws supplied as uniforms (sample_pick).

    input_tokens [b or 1, n] (model.py:672-686): given mel tokens are appended to the fake ids, so the FIRST forward embeds
    [start_mel, t1..tn] with positions 0..n (model.py:141-144) and the first generated token is fed at position n + 2
    (model.py:151-155); the returned codes start after the given tokens (trunc_index, model.py:687,704).

    typical_mass > 0 (typical_sampling=True, model.py:690-697) with sampling None: greedy search over the typical-filtered
    scores - the warper sits in HF's logits_processor list, which greedy search runs as well."""
    import numpy as np

    stop = cfg_gpt["stop_mel_token"]
    fake, prefix, mask = prepare_gpt_inputs(cond, text_inputs, w, cfg_gpt)
    b, s, _ = prefix.shape
    ids = fake.clone()
    n_in = 0
    if input_tokens is not None:
        it = torch.as_tensor(input_tokens).long()
        it = it[None] if it.ndim == 1 else it
        it = it.repeat(b // it.shape[0], 1)
        n_in = it.shape[1]
        ids = torch.cat([ids, it], dim=1)
        mask = torch.cat([mask, torch.ones(b, n_in, dtype=mask.dtype)], dim=1)
    mel_emb, mel_pos = w["mel_embedding.weight"], w["mel_pos_embedding.emb.weight"]
    start_emb = mel_emb[ids[:, s:]] + mel_pos[: 1 + n_in]
    emb = torch.cat([prefix, start_emb], dim=1)
    h, past = gpt2_stack(emb, w, cfg_gpt, key_mask=mask)
    unfinished = torch.ones(b, dtype=torch.long)
    out_logits = []
    while True:
        logits = lm_head(h[:, -1:], w)[:, 0]
        if trace is not None:
            out_logits.append(logits.clone())
        scores = repetition_penalty_(logits.clone(), ids, repetition_penalty) if repetition_penalty != 1.0 else logits
        if suppress_eos:
            scores[:, stop] = -float("inf")
        if sampling is not None:
            k_step = ids.shape[1] - (s + 1 + n_in)
            if sampling.get("typical_mass"):
                from . import hf_beam

                scores = torch.from_numpy(np.stack([hf_beam.typical_filter(scores[r].numpy(), sampling["typical_mass"], 1) for r in range(b)]))
            nxt = torch.tensor([sample_pick(scores[r].numpy(), sampling["top_k"], sampling["top_p"], sampling["temperature"],
                                            float(sampling["uniforms"][k_step, r])) for r in range(b)], dtype=torch.long)
        else:
            if typical_mass:  # greedy search runs the logits_processor list too: RepetitionPenalty, TypicalLogitsWarper(min keep 1)
                from . import hf_beam

                scores = torch.from_numpy(np.stack([hf_beam.typical_filter(scores[r].numpy(), typical_mass, 1) for r in range(b)]))
            nxt = torch.argmax(scores, dim=-1)
        nxt = nxt * unfinished + stop * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        mask = torch.cat([mask, torch.ones(b, 1, dtype=torch.long)], dim=1)
        unfinished = unfinished * (nxt != stop).long()
        if unfinished.max() == 0 or ids.shape[1] >= s + 1 + n_in + max_generate_length:
            break
        e = mel_emb[nxt][:, None] + mel_pos[mask.shape[1] - s][None, None]
        h, past = gpt2_stack(e, w, cfg_gpt, key_mask=mask, past=past)
    if trace is not None:
        trace["logits"] = torch.stack(out_logits, dim=1)
    return ids[:, s + 1 + n_in:]


def latent_forward(cond, text_tokens, codes, w: W, cfg_gpt):
    """UnifiedVoice.forward(return_latent=True) for batch 1 as infer.py:194-200 calls it (model.py:521-589,
    get_logits :462-477): text -> [start,text,stop]+pos; codes -> [start,codes,stop]+pos;
    cat(cond32, text, mel) -> GPT2 (full causal, no cache) -> strip cond -> final_norm -> mel part [:, :-2]."""
    g = cfg_gpt
    t = F.pad(F.pad(text_tokens.long(), (0, 1), value=g["stop_text_token"]), (1, 0), value=g["start_text_token"])
    m = F.pad(F.pad(codes.long(), (0, 1), value=g["stop_mel_token"]), (1, 0), value=g["start_mel_token"])
    te = w["text_embedding.weight"][t] + w["text_pos_embedding.emb.weight"][: t.shape[1]]
    me = w["mel_embedding.weight"][m] + w["mel_pos_embedding.emb.weight"][: m.shape[1]]
    emb = torch.cat([cond.expand(t.shape[0], -1, -1), te, me], dim=1)
    h, _ = gpt2_stack(emb, w, g)
    enc = layer_norm(h[:, cond.shape[1]:], w, "final_norm")
    return enc[:, -m.shape[1]:][:, :-2]


def beam_sample_generate(cond, text_inputs, w: W, cfg_gpt, max_generate_length: int, num_beams: int = 3, top_k: int = 30,
                         top_p: float = 0.8, temperature: float = 1.0, repetition_penalty: float = 10.0,
                         length_penalty: float = 0.0, uniforms=None, trace: Optional[dict] = None, typical_mass: float = 0.0,
                         do_sample: bool = True, num_return_sequences: int = 1, input_tokens=None):
    """UnifiedVoice.inference_speech under the reference's DEFAULT kwargs (infer.py:116-124: do_sample=True, num_beams=3,
    top_k=30, top_p=0.8, length_penalty=0.0, repetition_penalty=10.0): HF 4.36.2 GenerationMixin.beam_sample +
    BeamSearchScorer, restated in oracle/hf_beam.py, over this module's GPT-2 stack with the KV cache re-ordered by
    beam_idx every step (GPT2InferenceModel._reorder_cache, model.py:194-207).  uniforms [max_gen, b, 2*num_beams] are
    the draws.  Returns codes [b * num_return_sequences, <= max_generate_length] (prefix stripped, model.py:704-705;
    num_return_sequences = BeamSearchScorer's num_beam_hyps_to_keep, model.py:655,698-703: the n best per text row).

    input_tokens [b or 1, n] (model.py:672-686): appended to the fake ids before the expansion, so they are part of the decoder
    prompt of every beam (first forward at positions 0 .. n, generated_len counts after them); the draws of generated step j
    are uniforms[n + j] (the device indexes its uniforms by the absolute step)."""
    import numpy as np

    from . import hf_beam

    stop = cfg_gpt["stop_mel_token"]
    V = cfg_gpt["number_mel_codes"]
    nb = num_beams
    fake, prefix, mask = prepare_gpt_inputs(cond, text_inputs, w, cfg_gpt)
    b, s, _ = prefix.shape
    n_in = 0
    if input_tokens is not None:
        it = torch.as_tensor(input_tokens).long()
        it = it[None] if it.ndim == 1 else it
        it = it.repeat(b // it.shape[0], 1)
        n_in = it.shape[1]
        fake = torch.cat([fake, it], dim=1)
        mask = torch.cat([mask, torch.ones(b, n_in, dtype=mask.dtype)], dim=1)
    # _expand_inputs_for_generation: repeat_interleave(num_beams) on ids / mask; store_mel_emb's prefix repeats likewise
    ids = fake.repeat_interleave(nb, 0).clone()
    mask = mask.repeat_interleave(nb, 0)
    prefix = prefix.repeat_interleave(nb, 0)
    mel_emb, mel_pos = w["mel_embedding.weight"], w["mel_pos_embedding.emb.weight"]
    emb = torch.cat([prefix, mel_emb[ids[:, s:]] + mel_pos[: 1 + n_in]], dim=1)
    h, past = gpt2_stack(emb, w, cfg_gpt, key_mask=mask)
    prompt_len = s + 1 + n_in
    scorer = hf_beam.BeamSearchScorer(b, nb, length_penalty=length_penalty, max_length=prompt_len + max_generate_length,
                                      num_beam_hyps_to_keep=num_return_sequences)
    beam_scores = np.zeros(b * nb, dtype=np.float32)
    if not do_sample:  # beam_search: `beam_scores[:, 1:] = -1e9`, so that step 0 expands beam 0 only
        beam_scores.reshape(b, nb)[:, 1:] = -1e9
    step = 0
    steps_log = []
    while True:
        logits = lm_head(h[:, -1:], w)[:, 0]
        lp = torch.log_softmax(logits, dim=-1)
        lp = repetition_penalty_(lp.clone(), ids, repetition_penalty) if repetition_penalty != 1.0 else lp
        lpn = lp.numpy()
        if not do_sample:
            # HF beam_search: processors only (no warpers; the reference's TypicalLogitsWarper IS a processor, model.py:690-697),
            # torch.topk over the flat [beams * V] scores
            if typical_mass:
                lp = torch.from_numpy(np.stack([hf_beam.typical_filter(lpn[r], typical_mass, 2) for r in range(lpn.shape[0])]))
            flat = (lp + torch.from_numpy(beam_scores)[:, None]).view(b, nb * V)
            top = torch.topk(flat, 2 * nb, dim=1, largest=True, sorted=True)
            ns, ni, nt = top.values.numpy(), (top.indices // V).numpy(), (top.indices % V).numpy()
            ids_np = ids.numpy()
            beam_scores, btok, bidx = scorer.process(ids_np, ns, nt, ni, stop, stop, prompt_len)
            bidx_t = torch.from_numpy(bidx)
            ids = torch.cat([ids[bidx_t], torch.from_numpy(btok)[:, None]], dim=1)
            mask = torch.cat([mask, torch.ones(b * nb, 1, dtype=torch.long)], dim=1)
            past = [(kk[bidx_t], vv[bidx_t]) for kk, vv in past]
            step += 1
            if scorer.is_done or ids.shape[1] >= prompt_len + max_generate_length:
                break
            e = mel_emb[ids[:, -1]][:, None] + mel_pos[mask.shape[1] - s][None, None]
            h, past = gpt2_stack(e, w, cfg_gpt, key_mask=mask, past=past)
            continue
        if typical_mass:  # logits_processor list: RepetitionPenalty, then the reference's TypicalLogitsWarper (model.py:690-697)
            lpn = np.stack([hf_beam.typical_filter(lpn[r], typical_mass, 2) for r in range(lpn.shape[0])])
        ns, nt, ni = [], [], []
        for bi in range(b):
            cands = [hf_beam.warp_row(lpn[bi * nb + r], top_k, top_p, temperature, 2) for r in range(nb)]
            sc, tk, bm = hf_beam.beam_sample_step(cands, beam_scores[bi * nb:(bi + 1) * nb], V, uniforms[n_in + step, bi])
            ns.append(sc)
            nt.append(tk)
            ni.append(bm)
        ns, nt, ni = np.stack(ns), np.stack(nt), np.stack(ni)
        ids_np = ids.numpy()
        beam_scores, btok, bidx = scorer.process(ids_np, ns, nt, ni, stop, stop, prompt_len)
        if trace is not None:
            steps_log.append(dict(scores=ns.copy(), tokens=nt.copy(), beams=ni.copy(), beam_idx=bidx.copy(), next_tokens=btok.copy(),
                                  beam_scores=beam_scores.copy(), done=list(scorer.done)))
        bidx_t = torch.from_numpy(bidx)
        ids = torch.cat([ids[bidx_t], torch.from_numpy(btok)[:, None]], dim=1)
        mask = torch.cat([mask, torch.ones(b * nb, 1, dtype=torch.long)], dim=1)
        past = [(k[bidx_t], v[bidx_t]) for k, v in past]
        step += 1
        if scorer.is_done or ids.shape[1] >= prompt_len + max_generate_length:
            break
        e = mel_emb[ids[:, -1]][:, None] + mel_pos[mask.shape[1] - s][None, None]
        h, past = gpt2_stack(e, w, cfg_gpt, key_mask=mask, past=past)
    if trace is not None:
        trace["steps"] = steps_log
    out = scorer.finalize(ids.numpy(), beam_scores, stop, stop, prompt_len)
    return torch.from_numpy(out[:, prompt_len:])
