"""Golden-fixture generator: runs the REAL reference modules (imported from /root/reference with the two
stubs of oracle/ref_import.py) on builder-PRNG weights/inputs and stores inputs + expected outputs under
tests/golden/.  Runs only in the authoring container; the fixtures are data (no reference source).

    python -m oracle.make_golden [--full]      # --full adds the IndexTTS-1.5-sized id/logit fixtures

The weights are NOT stored: they regenerate bit-identically from itts_hip.synth (integer PRNG).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "index-tts-ipex_amd"))

from itts_hip import config as icfg  # noqa: E402
from itts_hip import prng, synth  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class H(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def tt(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def build_ref_gpt(cfg, seed, profile="sharp", stop_bias=0.0):
    from indextts.gpt.model import UnifiedVoice

    m = UnifiedVoice(**cfg.gpt)
    sd = tt(synth.gpt_state_dict(cfg, seed, profile=profile, stop_bias=stop_bias))
    ref_sd = m.state_dict()
    assert set(sd) == set(ref_sd), (sorted(set(sd) ^ set(ref_sd))[:10])
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), (k, sd[k].shape, ref_sd[k].shape)
    m.load_state_dict(sd, strict=True)
    m.eval()
    m.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=False)
    return m


def build_ref_bigvgan(cfg, seed):
    from indextts.BigVGAN.models import BigVGAN

    m = BigVGAN(H(cfg.bigvgan), use_cuda_kernel=False)
    m.eval()
    m.remove_weight_norm()
    sd = tt(synth.bigvgan_state_dict(cfg, seed))
    ref_sd = m.state_dict()
    missing = [k for k in ref_sd if k not in sd]
    assert all(k.endswith("filter") for k in missing), [k for k in missing if not k.endswith("filter")][:10]
    assert not [k for k in sd if k not in ref_sd]
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), (k, sd[k].shape, ref_sd[k].shape)
    m.load_state_dict(sd, strict=False)
    return m


def build_ref_dvae(cfg, seed):
    from indextts.vqvae.xtts_dvae import DiscreteVAE

    m = DiscreteVAE(**{k: v for k, v in cfg.vqvae.items()})
    m.eval()
    sd = tt(synth.dvae_state_dict(cfg, seed))
    ref_sd = m.state_dict()
    assert not [k for k in sd if k not in ref_sd]
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), (k, sd[k].shape, ref_sd[k].shape)
    dec_keys = [k for k in ref_sd if k.startswith("decoder.") or k.startswith("encoder.") or k == "codebook.embed"]
    assert set(dec_keys) == set(sd), sorted(set(dec_keys) ^ set(sd))[:10]
    m.load_state_dict(sd, strict=False)
    return m


def ref_greedy(gpt, cond_mel, text, max_gen, rep=10.0, n_trace=None, suppress_eos=False, trace_steps=None, input_tokens=None,
               lens=None, typical_mass=0.0):
    """Hand-rolled HF-4.36.2-style greedy_search over the reference's own GPT2InferenceModel.forward
    (the installed transformers-5.x `generate` skips the prefill, so it is not used)."""
    from transformers import RepetitionPenaltyLogitsProcessor

    stop = gpt.stop_mel_token
    if lens is None:
        lens = torch.tensor([cond_mel.shape[-1]])
    conds = gpt.get_conditioning(cond_mel, lens)
    ids, emb, mask = gpt.prepare_gpt_inputs(conds, text)
    gpt.inference_model.store_mel_emb(emb)
    s = emb.shape[1]
    n_in = 0
    if input_tokens is not None:  # inference_speech, model.py:672-686
        it = input_tokens[None] if input_tokens.ndim == 1 else input_tokens
        it = it.repeat(ids.shape[0] // it.shape[0], 1)
        n_in = it.shape[1]
        ids = torch.cat([ids, it], dim=1)
        mask = torch.nn.functional.pad(mask, (0, n_in), value=1)
    proc = RepetitionPenaltyLogitsProcessor(rep)
    typical = None
    if typical_mass:  # typical_sampling=True: the reference's own subclass, appended to the logits_processor list (model.py:690-697)
        from indextts.utils.typical_sampling import TypicalLogitsWarper

        typical = TypicalLogitsWarper(mass=typical_mass, min_tokens_to_keep=1)
    past = None
    b = ids.shape[0]
    unfinished = torch.ones(b, dtype=torch.long)
    logits_trace = []
    margins = []
    stop_gaps = []
    step = 0
    while True:
        inp = ids if past is None else ids[:, -1:]
        out = gpt.inference_model(input_ids=inp, past_key_values=past, attention_mask=mask, use_cache=True,
                                  return_dict=True)
        past = out.past_key_values
        logits = out.logits[:, -1, :]
        if trace_steps is not None:
            if step in trace_steps:
                logits_trace.append(logits.clone())
        elif n_trace is None or len(logits_trace) < n_trace:
            logits_trace.append(logits.clone())
        scores = proc(ids, logits.clone())
        if suppress_eos:  # fixed-length decode: the engine's suppress_stop masks the stop score after the penalty
            scores[:, stop] = -float("inf")
        if typical is not None:
            scores = typical(ids, scores)
        top2 = torch.topk(scores, 2, dim=-1).values
        margins.append((top2[:, 0] - top2[:, 1]).clone())
        if suppress_eos:  # how far the stop logit is below the winner (calibrates synth's stop_bias for the eos fixtures)
            stop_gaps.append((top2[:, 0] - logits[:, stop]).clone())
        step += 1
        nxt = torch.argmax(scores, dim=-1)
        nxt = nxt * unfinished + stop * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        mask = torch.cat([mask, torch.ones(b, 1, dtype=mask.dtype)], dim=-1)
        unfinished = unfinished * (nxt != stop).long()
        if unfinished.max() == 0 or ids.shape[-1] >= s + 1 + n_in + max_gen:
            break
    ref_greedy.last_margins = torch.stack(margins, 1)
    ref_greedy.last_stop_gaps = torch.stack(stop_gaps, 1) if stop_gaps else None
    return ids[:, s + 1 + n_in:], torch.stack(logits_trace, 1), conds, emb, mask[:, : s + 1]


def ref_beam_sample(gpt, cond_mel, text, max_gen, uniforms, nb=3, top_k=30, top_p=0.8, temperature=1.0, rep=10.0,
                    length_penalty=0.0, typical_mass=0.0, do_sample=True, input_tokens=None, prebuilt=None, nret=1):
    """Hand-rolled HF-4.36.2 `beam_sample` over the reference's own GPT2InferenceModel.forward / _reorder_cache with the
    INSTALLED transformers logits processors / warpers (min_tokens_to_keep = 2 under beams) and the BeamSearchScorer
    restatement of oracle/hf_beam.py; torch.multinomial is replaced by the shared-uniform sequential draw."""
    from transformers import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                              TopPLogitsWarper)

    from oracle import hf_beam

    stop = gpt.stop_mel_token
    V = gpt.number_mel_codes
    n_in = 0
    if prebuilt is not None:
        # (ids, mask) exactly as the reference's inference_speech handed them to generate() (captured by a stub generate: the
        # row expansion of model.py:672-686 is the reference's own code); store_mel_emb has been called by inference_speech
        ids, mask = prebuilt
        s = ids.shape[1] - 1
        b = ids.shape[0]
    else:
        lens = torch.tensor([cond_mel.shape[-1]])
        conds = gpt.get_conditioning(cond_mel, lens)
        ids, emb, mask = gpt.prepare_gpt_inputs(conds, text)
        gpt.inference_model.store_mel_emb(emb)
        s = emb.shape[1]
        b = ids.shape[0]
    if input_tokens is not None:  # inference_speech, model.py:672-686: part of the decoder prompt of every beam
        it = input_tokens[None] if input_tokens.ndim == 1 else input_tokens
        it = it.repeat(b // it.shape[0], 1)
        n_in = it.shape[1]
        ids = torch.cat([ids, it], dim=1)
        mask = torch.nn.functional.pad(mask, (0, n_in), value=1)
    ids = ids.repeat_interleave(nb, 0)
    mask = mask.repeat_interleave(nb, 0)
    proc = RepetitionPenaltyLogitsProcessor(rep)
    typical = None
    if typical_mass:  # the reference's own subclass, placed after the default processors (model.py:690-697)
        from indextts.utils.typical_sampling import TypicalLogitsWarper

        typical = TypicalLogitsWarper(mass=typical_mass, min_tokens_to_keep=2 if nb > 1 else 1)
    warpers = []
    if temperature != 1.0:
        warpers.append(TemperatureLogitsWarper(temperature))
    if top_k:
        warpers.append(TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=2))
    if top_p is not None and top_p < 1.0:
        warpers.append(TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=2))
    prompt_len = s + 1 + n_in
    scorer = hf_beam.BeamSearchScorer(b, nb, length_penalty=length_penalty, max_length=prompt_len + max_gen, num_beam_hyps_to_keep=nret)
    beam_scores = np.zeros(b * nb, dtype=np.float32)
    if not do_sample:  # beam_search initialisation
        beam_scores.reshape(b, nb)[:, 1:] = -1e9
    past, step = None, 0
    while True:
        inp = ids if past is None else ids[:, -1:]
        out = gpt.inference_model(input_ids=inp, past_key_values=past, attention_mask=mask, use_cache=True, return_dict=True)
        past = out.past_key_values
        sc = torch.log_softmax(out.logits[:, -1, :], dim=-1)
        sc = proc(ids, sc.clone())
        if typical is not None:
            sc = typical(ids, sc)
        if do_sample:
            for wp in warpers:
                sc = wp(ids, sc)
        scn = sc.numpy()
        ns, nt, ni = [], [], []
        if not do_sample:  # beam_search: torch.topk over [beams * V]
            flat = (sc + torch.from_numpy(beam_scores)[:, None]).view(b, nb * V)
            top = torch.topk(flat, 2 * nb, dim=1, largest=True, sorted=True)
            ns, ni, nt = list(top.values.numpy()), list((top.indices // V).numpy()), list((top.indices % V).numpy())
        for bi in range(b if do_sample else 0):
            cands = []
            for r in range(nb):
                row = scn[bi * nb + r]
                keep = np.nonzero(np.isfinite(row))[0]
                cands.append((keep, row[keep]))
            a, t, m = hf_beam.beam_sample_step(cands, beam_scores[bi * nb:(bi + 1) * nb], V, uniforms[n_in + step, bi])
            ns.append(a)
            nt.append(t)
            ni.append(m)
        beam_scores, btok, bidx = scorer.process(ids.numpy(), np.stack(ns), np.stack(nt), np.stack(ni), stop, stop, prompt_len)
        bidx_t = torch.from_numpy(bidx)
        ids = torch.cat([ids[bidx_t], torch.from_numpy(btok)[:, None]], dim=-1)
        mask = torch.cat([mask, torch.ones(b * nb, 1, dtype=mask.dtype)], dim=-1)
        if isinstance(past, tuple):
            past = type(gpt.inference_model)._reorder_cache(past, bidx_t)
        else:
            past.reorder_cache(bidx_t)
        step += 1
        if scorer.is_done or ids.shape[-1] >= prompt_len + max_gen:
            break
    return scorer.finalize(ids.numpy(), beam_scores, stop, stop, prompt_len)[:, prompt_len:]


def save(name, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def rnd(name, shape, seed=3, std=1.0, mean=0.0):
    return torch.from_numpy(prng.tensor(name, seed, shape, std=std, mean=mean))


@torch.no_grad()
def micro_fixtures():
    cfg = icfg.micro()
    seed = 1234
    g = cfg.gpt
    print("[micro] activation1d / filter")
    from indextts.BigVGAN import activations
    from indextts.BigVGAN.alias_free_torch import Activation1d

    for tag, (B, C, T) in {"a": (2, 8, 37), "b": (1, 24, 5), "c": (1, 3, 1)}.items():
        act = Activation1d(activation=activations.SnakeBeta(C, alpha_logscale=True))
        al, be = rnd(f"act.{tag}.alpha", (C,), std=0.4), rnd(f"act.{tag}.beta", (C,), std=0.4)
        act.act.alpha.data.copy_(al)
        act.act.beta.data.copy_(be)
        x = rnd(f"act.{tag}.x", (B, C, T), std=1.5)
        save(f"micro_act1d_{tag}", x=x, alpha=al, beta=be, y=act(x), filt=act.upsample.filter.view(-1),
             filt_down=act.downsample.lowpass.filter.view(-1))

    print("[micro] GPT conditioning / prefix / decode / latent")
    gpt = build_ref_gpt(cfg, seed)
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    lens = torch.tensor([mel.shape[-1]])
    enc, mask = gpt.conditioning_encoder(mel.transpose(1, 2), lens)
    assert bool(mask.all())
    cond = gpt.get_conditioning(mel, lens)
    save("micro_conditioning", mel=mel, conformer_out=enc, cond=cond)

    L = 11
    text = torch.from_numpy(synth.text_ids(L, 11, g.number_text_tokens)).view(1, L).int()
    codes, logits, conds, emb, mask0 = ref_greedy(gpt, mel, text, max_gen=24)
    save("micro_decode_b1", text=text, codes=codes, logits=logits, prefix_emb=emb, prefix_mask=mask0)
    # sensitivity self-check data: one changed text id
    text2 = text.clone()
    text2[0, 3] = (int(text2[0, 3]) + 7) % (g.number_text_tokens - 2) + 2
    codes2, logits2, *_ = ref_greedy(gpt, mel, text2, max_gen=24)
    save("micro_decode_b1_alt", text=text2, codes=codes2, logits0=logits2[:, 0])

    # the padding/batch invariance of tests/padding_test.py:52-67 (five bos/eos padded variants as one batch)
    F_ = torch.nn.functional
    pads = [F_.pad(text, (8, 0), value=0), F_.pad(text, (0, 8), value=1),
            F_.pad(F_.pad(text, (4, 0), value=0), (0, 4), value=1),
            F_.pad(F_.pad(text, (6, 0), value=0), (0, 2), value=1),
            F_.pad(F_.pad(text, (0, 4), value=0), (0, 4), value=1)]
    btxt = torch.cat(pads, 0)
    bcodes, blogits, _, bemb, bmask = ref_greedy(gpt, mel, btxt, max_gen=24, n_trace=4)
    save("micro_decode_b5", text=btxt, codes=bcodes, logits=blogits, prefix_emb=bemb, prefix_mask=bmask)

    # ragged real batch (different sentences, right-padded with stop like infer_fast's pad_tokens_cat)
    t_a = torch.from_numpy(synth.text_ids(9, 21, g.number_text_tokens)).view(1, -1).int()
    t_b = torch.from_numpy(synth.text_ids(14, 22, g.number_text_tokens)).view(1, -1).int()
    rag = torch.nn.utils.rnn.pad_sequence([t_a[0], t_b[0]], batch_first=True, padding_value=1)
    rcodes, rlogits, *_ = ref_greedy(gpt, mel, rag, max_gen=20, n_trace=4)
    save("micro_decode_ragged", text=rag, codes=rcodes, logits=rlogits)

    T = codes.shape[1]
    lat_codes = codes.clone()
    lat_codes[lat_codes == g.stop_mel_token] = 5  # latent pass sees cleaned codes (no stop inside)
    latent = gpt(mel, text, torch.tensor([L]), lat_codes, torch.tensor([T * 1024]), cond_mel_lengths=lens,
                 return_latent=True, clip_inputs=False)
    save("micro_latent", text=text, codes=lat_codes, latent=latent)

    print("[micro] ECAPA / BigVGAN")
    bv = build_ref_bigvgan(cfg, seed)
    spk = bv.speaker_encoder(mel.transpose(1, 2), None)
    save("micro_ecapa", mel=mel, spk=spk)
    lat_in = rnd("bigvgan.latent", (1, 7, cfg.bigvgan.gpt_dim), std=1.0)
    wav, _ = bv(lat_in, mel.transpose(1, 2))
    # stage taps through the reference's own submodules
    x = bv.conv_pre(lat_in.transpose(1, 2)) + bv.cond_layer(spk.transpose(1, 2))
    x_up0 = bv.ups[0][0](x) + bv.conds[0](spk.transpose(1, 2))
    amp0 = bv.resblocks[0](x_up0)
    save("micro_bigvgan", latent=lat_in, mel=mel, wav=wav, pre=x, up0=x_up0, amp0=amp0)
    lat2 = rnd("bigvgan.latent2", (2, 5, cfg.bigvgan.gpt_dim), std=1.0)
    mel2 = torch.cat([mel, torch.from_numpy(synth.prompt_mel(61, seed=8))], 0)
    wav2, _ = bv(lat2, mel2.transpose(1, 2))
    save("micro_bigvgan_b2", latent=lat2, mel=mel2, wav=wav2)

    print("[micro] DVAE decode")
    dv = build_ref_dvae(cfg, seed)
    dcodes = torch.from_numpy(prng.randint("dvae.codes", 5, 2 * 9, 0, cfg.vqvae.num_tokens)).view(2, 9)
    out, _ = dv.decode(dcodes)
    save("micro_dvae", codes=dcodes, mel=out)
    # get_codebook_indices (encoder + Quantize arg-min): even, odd and tiny lengths
    enc = {}
    for tag, (B_, T_) in {"a": (2, 36), "b": (1, 37), "c": (1, 5)}.items():
        m = rnd(f"dvae.enc.mel.{tag}", (B_, cfg.vqvae.channels, T_), std=2.0, mean=-4.0)
        enc[f"mel_{tag}"] = m
        enc[f"codes_{tag}"] = dv.get_codebook_indices(m)
    save("micro_dvae_encode", **enc)

    print("[micro] beam-sample (the reference's default kwargs: 3 beams, top_k 30, top_p 0.8)")
    rng = np.random.default_rng(2024)
    for tag, (txt, nb, tk, tp, tmp, n) in {"a": (torch.cat([text, text2], 0), 3, 30, 0.8, 1.0, 24), "b": (text, 3, 10, 0.6, 0.9, 20),
                                            "c": (rag, 2, 30, 0.9, 1.0, 16)}.items():
        u = rng.random((n, txt.shape[0], 2 * nb), dtype=np.float32)
        codes_b = ref_beam_sample(gpt, mel, txt, n, u, nb=nb, top_k=tk, top_p=tp, temperature=tmp)
        save(f"micro_beam_{tag}", text=txt, codes=codes_b, uniforms=u, num_beams=nb, top_k=tk, top_p=tp, temperature=tmp, max_gen=n)
    u = rng.random((20, 2, 6), dtype=np.float32)
    txt = torch.cat([text, text2], 0)
    codes_t = ref_beam_sample(gpt, mel, txt, 20, u, nb=3, top_k=30, top_p=0.8, temperature=1.0, typical_mass=0.6)
    for tag, (nb_, ds_, lp_, tk_, n_) in {"search3": (3, False, 0.0, 30, 20), "search5_lp": (5, False, 1.0, 30, 16),
                                           "sample5_lp": (5, True, 0.7, 100, 16)}.items():
        u5 = rng.random((n_, 2, 2 * nb_), dtype=np.float32)
        cb = ref_beam_sample(gpt, mel, txt, n_, u5, nb=nb_, top_k=tk_, top_p=0.8, temperature=1.0, length_penalty=lp_, do_sample=ds_)
        save(f"micro_beam_{tag}", text=txt, codes=cb, uniforms=u5, num_beams=nb_, top_k=tk_, top_p=0.8, temperature=1.0, max_gen=n_,
             length_penalty=lp_, do_sample=int(ds_))
    save("micro_beam_typical", text=txt, codes=codes_t, uniforms=u, num_beams=3, top_k=30, top_p=0.8, temperature=1.0, max_gen=20,
         typical_mass=0.6)

    print("[int] remove_long_silence known answers")
    ref_import._stub("omegaconf", OmegaConf=object)
    from indextts.infer import IndexTTS

    class Dummy:
        stop_mel_token = g.stop_mel_token

    cases = []
    S = g.stop_mel_token
    rows = [
        [3, 4, 5, S, S, S],
        [3] + [52] * 40 + [7, 8, S],
        [52] * 12 + [9] + [52] * 25 + [4],
        [1, 2, 3, 4, 5, 6],
        [S, 1, 2],
    ]
    for i, r in enumerate(rows):
        c = torch.tensor([r], dtype=torch.long)
        oc, ol = IndexTTS.remove_long_silence(Dummy(), c.clone(), silent_token=52, max_consecutive=30)
        cases.append((c, oc, ol))
    # batched ragged case
    c = torch.tensor([[3] + [52] * 40 + [7, 8, S, S], [5, 6, 7, S] + [S] * 41], dtype=torch.long)
    oc, ol = IndexTTS.remove_long_silence(Dummy(), c.clone(), silent_token=52, max_consecutive=30)
    cases.append((c, oc, ol))
    save("silence_cases", **{f"in{i}": a for i, (a, _, _) in enumerate(cases)},
         **{f"out{i}": b for i, (_, b, _) in enumerate(cases)}, **{f"len{i}": c_ for i, (_, _, c_) in enumerate(cases)},
         n=len(cases), stop=S)


@torch.no_grad()
def full_fixtures():
    """IndexTTS-1.5-sized: greedy ids + top-8 logits per step + margins (ints / small floats only)."""
    cfg = icfg.indextts_1_5()
    g = cfg.gpt
    print("[full] building reference UnifiedVoice (583 M params) ...")
    t0 = time.time()
    gpt = build_ref_gpt(cfg, 1234)
    print(f"  built in {time.time() - t0:.1f}s")
    mel = torch.from_numpy(synth.prompt_mel(511, seed=7))
    L = 52
    text = torch.from_numpy(synth.text_ids(L, 11, g.number_text_tokens)).view(1, L).int()
    t0 = time.time()
    codes, logits, conds, emb, _ = ref_greedy(gpt, mel, text, max_gen=48)
    print(f"  48-step greedy in {time.time() - t0:.1f}s")
    top = torch.topk(logits[0], 8, dim=-1)
    lat_codes = codes.clone()
    lat_codes[lat_codes == g.stop_mel_token] = 5
    latent = gpt(mel, text, torch.tensor([L]), lat_codes, torch.tensor([codes.shape[1] * 1024]),
                 cond_mel_lengths=torch.tensor([511]), return_latent=True, clip_inputs=False)
    save("full_decode_b1", text=text, codes=codes, top_idx=top.indices, top_val=top.values,
         cond_sample=conds[0, :, :16], cond_rms=conds.pow(2).mean().sqrt(),
         latent_sample=latent[0, :, :16], latent_rms=latent.pow(2).mean().sqrt(), lat_codes=lat_codes)
    del gpt
    print("[full] BigVGAN generator (134 M params) ...")
    bv = build_ref_bigvgan(cfg, 1234)
    lat_in = rnd("bigvgan.latent.full", (1, 12, cfg.bigvgan.gpt_dim), std=1.0)
    t0 = time.time()
    wav, _ = bv(lat_in, mel.transpose(1, 2))
    spk = bv.speaker_encoder(mel.transpose(1, 2), None)
    print(f"  vocoder in {time.time() - t0:.1f}s")
    save("full_bigvgan", latent=lat_in, wav=wav, spk=spk)


@torch.no_grad()
def long_fixtures():
    """IndexTTS-1.5 sizes at the BENCHMARK shapes (L = 105, T = 480 latent, 64 vocoder frames): greedy ids for 660 steps
    with the stop token masked (the sequence passes S = 768, beyond every register window of the decode attention),
    top-8 logits every 32 steps and at S = 400 / 520 / 780, latent for T = 480, a 64-frame waveform."""
    cfg = icfg.indextts_1_5()
    g = cfg.gpt
    print("[long] building reference UnifiedVoice ...")
    gpt = build_ref_gpt(cfg, 1234)
    mel = torch.from_numpy(synth.prompt_mel(511, seed=7))
    L, NS = 105, 660
    text = torch.from_numpy(synth.text_ids(L, 11, g.number_text_tokens)).view(1, L).int()
    s0 = 32 + L + 2 + 1
    steps = sorted(set(range(0, NS, 32)) | {400 - s0, 520 - s0, 780 - s0, NS - 1})
    t0 = time.time()
    codes, logits, conds, emb, _ = ref_greedy(gpt, mel, text, max_gen=NS, suppress_eos=True, trace_steps=set(steps))
    print(f"  {NS}-step greedy in {time.time() - t0:.1f}s")
    assert codes.shape == (1, NS) and logits.shape[1] == len(steps)
    top = torch.topk(logits[0], 8, dim=-1)
    T = 480
    lat_codes = codes[:, :T].clone()
    t0 = time.time()
    latent = gpt(mel, text, torch.tensor([L]), lat_codes, torch.tensor([T * 1024]), cond_mel_lengths=torch.tensor([511]),
                 return_latent=True, clip_inputs=False)
    print(f"  latent T={T} in {time.time() - t0:.1f}s")
    rows = [0, 1, 239, 478, 479]
    save("long_decode_b1", text=text, codes=codes, trace_steps=np.asarray(steps), top_idx=top.indices, top_val=top.values,
         margins=ref_greedy.last_margins[0], latent_sample=latent[0, :, :16], latent_rows=latent[0, rows], latent_row_idx=np.asarray(rows),
         latent_rms=latent.pow(2).mean().sqrt())
    del gpt
    print("[long] BigVGAN 64 frames ...")
    bv = build_ref_bigvgan(cfg, 1234)
    lat_in = rnd("bigvgan.latent.long", (1, 64, cfg.bigvgan.gpt_dim), std=1.0)
    t0 = time.time()
    wav, _ = bv(lat_in, mel.transpose(1, 2))
    print(f"  vocoder in {time.time() - t0:.1f}s")
    save("long_bigvgan", wav=wav[0, 0])  # the latent regenerates from the PRNG (name bigvgan.latent.long, seed 3)


@torch.no_grad()
def smooth_fixtures():
    """The bf16 ACCURACY fixtures: IndexTTS-1.5 sizes on the "smooth" synthetic checkpoint (itts_hip.synth profile: Q / K gain
    1, soft attention - fp32 compute on bf16-rounded weights moves its logits / latents by < 1e-2, against 0.1 - 0.3 on the
    "sharp" parity checkpoint), so a bound of a few percent can tell a correct bf16 kernel from a wrong one.
      smooth_decode_b1: greedy ids for 480 steps (stop masked), top-8 logits at steps up to S = 520, margins, latent T = 480
      smooth_decode_b6: six different sentences as one batch (the MFMA decode path), 64 free-running greedy steps + margins"""
    cfg = icfg.indextts_1_5()
    g = cfg.gpt
    print("[smooth] building reference UnifiedVoice (smooth profile) ...")
    gpt = build_ref_gpt(cfg, 1234, profile="smooth")
    mel = torch.from_numpy(synth.prompt_mel(511, seed=7))
    L, NS = 105, 480
    text = torch.from_numpy(synth.text_ids(L, 11, g.number_text_tokens)).view(1, L).int()
    s0 = 32 + L + 2 + 1
    steps = sorted({0, 1, 2, 8, 32, 64, 128, 200, 400 - s0, 320, 520 - s0, NS - 1})
    t0 = time.time()
    codes, logits, *_ = ref_greedy(gpt, mel, text, max_gen=NS, suppress_eos=True, trace_steps=set(steps))
    print(f"  {NS}-step greedy in {time.time() - t0:.1f}s")
    top = torch.topk(logits[0], 8, dim=-1)
    T = 480
    latent = gpt(mel, text, torch.tensor([L]), codes[:, :T].clone(), torch.tensor([T * 1024]), cond_mel_lengths=torch.tensor([511]),
                 return_latent=True, clip_inputs=False)
    rows = [0, 1, 239, 478, 479]
    save("smooth_decode_b1", text=text, codes=codes, trace_steps=np.asarray(steps), top_idx=top.indices, top_val=top.values,
         logits_rms=logits[0].pow(2).mean(-1).sqrt(), margins=ref_greedy.last_margins[0], latent_sample=latent[0, :, :16],
         latent_rows=latent[0, rows], latent_row_idx=np.asarray(rows), latent_rms=latent.pow(2).mean().sqrt())
    t6 = torch.stack([torch.from_numpy(synth.text_ids(L, 60 + i, g.number_text_tokens)).int() for i in range(6)])
    t0 = time.time()
    c6, lg6, *_ = ref_greedy(gpt, mel, t6, max_gen=64, suppress_eos=True, n_trace=1)
    print(f"  6-row 64-step greedy in {time.time() - t0:.1f}s")
    top6 = torch.topk(lg6[:, 0], 8, dim=-1)
    save("smooth_decode_b6", text=t6, codes=c6, margins=ref_greedy.last_margins, top_idx0=top6.indices, top_val0=top6.values)


EOS_STOP_BIAS = None  # set by eos_fixtures(); the committed value lives in the fixture (stop_bias)
# (Reproducibility: ids / codes of every fixture are the same on any host; the FLOAT diagnostics - margins, the calibrated stop_bias -
#  follow torch's CPU summation order, i.e. the thread count: the committed files are what the default of this 8-CPU container gives,
#  torch.set_num_threads(6) moves stop_bias from 3.2128735 to 3.2128756 and the margins in the 6th digit, with the same ids.)


@torch.no_grad()
def eos_fixtures():
    """The SHIPPED decode loop: eos enabled, rows finishing at different steps (HF pads a finished row with the stop token,
    greedy_search's unfinished_sequences; beams finish into BeamHypotheses).  The synthetic checkpoint never emits the stop
    token on its own, so mel_head.bias[stop] is raised (synth stop_bias) by an amount calibrated HERE on the reference: the gap
    between the winning score and the stop logit over 64 stop-suppressed steps of three ragged rows; the bias is chosen so
    that the three rows stop at three different steps, each with >= 0.05 of head-room against a rounding flip.
      smooth_eos_b3:     3 ragged rows, greedy, eos enabled: ids [3, n] (pad = stop), per-row stop step, margins
      smooth_eos_beam:   2 sentences x 3 beams, beam_sample (the reference's default kwargs) AND beam_search, eos enabled:
                         finalized hypotheses [2, n]"""
    cfg = icfg.indextts_1_5()
    g = cfg.gpt
    mel = torch.from_numpy(synth.prompt_mel(511, seed=7))
    L = 105
    pool_lens = [105, 80, 60, 95, 70, 100, 50, 88]
    pool = torch.full((len(pool_lens), L), g.stop_text_token, dtype=torch.int32)  # pad_tokens_cat pads with the stop TEXT token (infer.py:316-318)
    for r, n in enumerate(pool_lens):
        pool[r, :n] = torch.from_numpy(synth.text_ids(n, 141 + r, g.number_text_tokens)).int()
    print("[eos] calibrating stop_bias on the reference (smooth profile, stop suppressed, a pool of 8 ragged rows) ...")
    gpt = build_ref_gpt(cfg, 1234, profile="smooth")
    NS = 64
    t0 = time.time()
    cache = os.environ.get("ITTS_EOS_GAPS_CACHE", "")  # authoring convenience: the calibration pass takes a minute
    if cache and os.path.exists(cache):
        gaps = np.load(cache)
    else:
        ref_greedy(gpt, mel, pool, max_gen=NS, suppress_eos=True, n_trace=1)
        gaps = ref_greedy.last_stop_gaps.numpy()  # [8, NS] winner - stop logit
        if cache:
            np.save(cache, gaps)
    print(f"  {NS} steps in {time.time() - t0:.1f}s; gap min per row {gaps.min(1)}, median {np.median(gaps):.2f}")
    # a bias d and three rows of the pool (the first two = the two sentences of the beam / drop-in tests, lengths 105 / 80 preferred)
    # whose first step with gap < d differs by >= 6 steps, lies in [6, 58], with >= `room` of head-room at every step up to it
    import itertools

    found = []
    room = 0.05
    for cand in np.sort(gaps.reshape(-1)):
        d = float(cand) + 2 * room
        first = {}
        for r in range(gaps.shape[0]):
            hit = np.nonzero(gaps[r] < d)[0]
            if len(hit) and 3 <= hit[0] <= 60 and not np.any(np.abs(gaps[r, : int(hit[0]) + 1] - d) < room):
                first[r] = int(hit[0])
        for tri in itertools.combinations(sorted(first), 3):
            st = [first[r] for r in tri]
            found.append((min(abs(x - y) for i, x in enumerate(st) for y in st[i + 1:]), -tri[0], d, list(tri), st))
    found.sort(reverse=True)  # the widest spacing between the three stop steps; ties: the pool's first row (105 text tokens) included
    best = found[0][2:] if found and found[0][0] >= 6 else None
    assert best is not None, "no stop_bias separates the rows"
    stop_bias, sel, first = float(np.float32(best[0])), best[1], best[2]
    t3 = pool[sel].clone()
    lens = [pool_lens[r] for r in sel]
    print(f"  stop_bias {stop_bias:.4f}: pool rows {sel} (text lengths {lens}) stop at steps {first}")
    del gpt
    gpt = build_ref_gpt(cfg, 1234, profile="smooth", stop_bias=stop_bias)
    codes, lg, *_ = ref_greedy(gpt, mel, t3, max_gen=NS, n_trace=1)
    stops = [int(np.nonzero(codes[r].numpy() == g.stop_mel_token)[0][0]) for r in range(3)]
    assert stops == first, (stops, first)
    assert codes.shape[1] == max(stops) + 1  # HF stops with the step in which the last row finishes
    save("smooth_eos_b3", text=t3, text_lens=np.asarray(lens), stop_bias=np.float32(stop_bias), codes=codes, stop_steps=np.asarray(stops),
         margins=ref_greedy.last_margins)
    # beams: the reference's default generate() kwargs (infer.py:116-124) on a two-sentence text = 6 rows, and beam search
    t2 = t3[:2].clone()
    for tag, sample in (("sample", True), ("search", False)):
        MG = 72
        u = np.random.default_rng(23).random((MG, 2, 6), dtype=np.float32)
        t0 = time.time()
        out = ref_beam_sample(gpt, mel, t2, MG, u, nb=3, top_k=30, top_p=0.8, temperature=1.0, rep=10.0, length_penalty=0.0,
                              do_sample=sample)
        print(f"  beam_{tag} 2 x 3: {out.shape} in {time.time() - t0:.1f}s; stop positions "
              f"{[int(np.argmax(out[r] == g.stop_mel_token)) if (out[r] == g.stop_mel_token).any() else -1 for r in range(2)]}")
        save(f"smooth_eos_beam_{tag}", text=t2, stop_bias=np.float32(stop_bias), uniforms=u, codes=out)


@torch.no_grad()
def nrs_input_token_fixtures():
    """`inference_speech(input_tokens=[2 rows], num_return_sequences=2)` (model.py:672-686): the reference repeats the text row
    and the given tokens to num_return_sequences rows BEFORE generate(), which expands them again (x num_beams under beams, with
    num_return_sequences hypotheses returned per row).  The expansion is the reference's own code: inference_speech is called
    for real with `inference_model.generate` replaced by a stub that records what it was handed; the recorded (ids, mask) then go
    through the hand-rolled beam search (the installed transformers generate() is not usable)."""
    cfg = icfg.micro()
    g = cfg.gpt
    gpt = build_ref_gpt(cfg, 1234)
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    text = torch.from_numpy(synth.text_ids(11, 11, g.number_text_tokens)).view(1, 11).int()
    base, *_ = ref_greedy(gpt, mel, text, max_gen=24)
    given = torch.stack([base[0, :4].clone(), base[0, 4:8].clone()])  # two different continuations of one text
    given[1, 1] = (int(given[1, 1]) + 3) % (g.number_mel_codes - 2)
    cap = {}

    def stub(inputs, **kw):
        cap["inputs"], cap["kw"] = inputs.clone(), kw
        return inputs

    real = gpt.inference_model.generate
    gpt.inference_model.generate = stub
    try:
        gpt.inference_speech(mel, text, input_tokens=given, num_return_sequences=2, max_generate_length=12, do_sample=False,
                             num_beams=3, length_penalty=1.0, repetition_penalty=10.0)
    finally:
        gpt.inference_model.generate = real
    ids, mask = cap["inputs"], cap["kw"]["attention_mask"]
    assert cap["kw"]["num_return_sequences"] == 2 and ids.shape[0] == 2 and cap["kw"]["max_length"] == ids.shape[1] + 12
    n_in = given.shape[1]
    rows_tokens = ids[:, -n_in:].clone()  # what each pre-expansion row continues from
    codes = ref_beam_sample(gpt, mel, text, 12, None, nb=3, do_sample=False, length_penalty=1.0, prebuilt=(ids, mask), nret=2)
    assert codes.shape[0] == 4
    save("micro_input_tokens_nrs", text=text, input_tokens=given, rows_tokens=rows_tokens, codes=codes[:, :], num_beams=3,
         num_return_sequences=2, length_penalty=1.0, max_gen=12)


@torch.no_grad()
def host_beam_fixtures():
    """beam_sample with the TopK warper OFF (`top_k = 0`, which infer.py:116-124 forwards verbatim and webui.py:393-402 offers) and
    with top_k = 200: more candidates per beam than the device sampler's 128 - IndexTTS-1.5 sizes (V = 8194), smooth checkpoint
    (logit std ~1: top_p 0.8 keeps thousands of tokens), 1 sentence x 3 beams, 20 steps; installed transformers warpers."""
    cfg = icfg.indextts_1_5()
    g = cfg.gpt
    gpt = build_ref_gpt(cfg, 1234, profile="smooth")
    mel = torch.from_numpy(synth.prompt_mel(511, seed=7))
    text = torch.from_numpy(synth.text_ids(105, 151, g.number_text_tokens)).view(1, 105).int()
    MG = 20
    # (top_k = 0 at temperature 1 keeps ~6000 tokens per beam on this checkpoint: a draw over 18000 flat candidates flips on the
    # last bits of the logits, and the GPU's fp32 logits differ from the CPU reference's in exactly those - the WIDE case is
    # therefore checked step by step against the oracle on the same box (tests/test_gpu_shipped_path.py); the fixture pins the
    # "warper off" semantics end to end at temperature 0.3, where top_p 0.8 keeps a few hundred tokens)
    for tag, tk, temp in (("topk0", 0, 0.3), ("topk200", 200, 1.0)):
        u = np.random.default_rng(31).random((MG, 1, 6), dtype=np.float32)
        t0 = time.time()
        out = ref_beam_sample(gpt, mel, text, MG, u, nb=3, top_k=tk, top_p=0.8, temperature=temp, rep=10.0, length_penalty=0.0)
        print(f"  beam_sample {tag}: {out.shape} in {time.time() - t0:.1f}s")
        save(f"smooth_beam_{tag}", text=text, uniforms=u, codes=out, top_k=tk, top_p=0.8, temperature=np.float32(temp), num_beams=3, max_gen=MG)


@torch.no_grad()
def fast_fixtures():
    """`infer_fast` (infer.py:332-537) on the micro config, greedy: 5 sentences, bucket size 2 -> length-sorted buckets,
    batched AR decode per bucket, per-sentence silence fix + latent, original order restored, BigVGAN over chunks of 2
    latents concatenated along time (:480-498), clamp to the int16 range.  Every arithmetic step runs the reference's
    own modules / methods; only the tokenizer (needs bpe.model) is replaced by pre-tokenised ids."""
    cfg = icfg.micro()
    g = cfg.gpt
    gpt = build_ref_gpt(cfg, 1234)
    bv = build_ref_bigvgan(cfg, 1234)
    ref_import._stub("omegaconf", OmegaConf=object)
    from indextts.infer import IndexTTS

    class Dummy:
        stop_mel_token = g.stop_mel_token
        device = "xpu"  # any non-"cpu" string: keeps bucket_max_size (infer.py:385-386)
        cfg = H(gpt=H(stop_text_token=g.stop_text_token))

    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    lens = [9, 14, 6, 11, 8]
    sents = [torch.from_numpy(synth.text_ids(n, 31 + i, g.number_text_tokens)).int() for i, n in enumerate(lens)]
    max_mel = 12
    buckets = IndexTTS.bucket_sentences(Dummy(), [s.tolist() for s in sents], bucket_max_size=2)
    all_codes, all_idx, all_lat = [], [], []
    for bk in buckets:
        toks = [torch.tensor(x["sent"], dtype=torch.int32).unsqueeze(0) for x in bk]
        batch = IndexTTS.pad_tokens_cat(Dummy(), toks) if len(toks) > 1 else toks[0]
        codes, *_ = ref_greedy(gpt, mel, batch, max_gen=max_mel)
        for i, x in enumerate(bk):
            c, cl = IndexTTS.remove_long_silence(Dummy(), codes[i:i + 1].clone(), silent_token=52, max_consecutive=30)
            lat = gpt(mel, toks[i], torch.tensor([toks[i].shape[-1]]), c, cl * gpt.mel_length_compression,
                      cond_mel_lengths=torch.tensor([mel.shape[-1]]), return_latent=True, clip_inputs=False)
            all_idx.append(x["idx"])
            all_lat.append(lat)
            all_codes.append(codes[i])
    all_lat = [all_lat[all_idx.index(i)] for i in range(len(all_lat))]
    wavs = []
    for lo in range(0, len(all_lat), 2):
        wav, _ = bv(torch.cat(all_lat[lo:lo + 2], dim=1), mel.transpose(1, 2))
        wavs.append(torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0))
    wav = torch.cat(wavs, dim=1)
    pad = torch.nn.utils.rnn.pad_sequence(sents, batch_first=True, padding_value=-1)
    cpad = torch.nn.utils.rnn.pad_sequence([all_codes[all_idx.index(i)] for i in range(len(sents))], batch_first=True, padding_value=-1)
    save("micro_infer_fast", text=pad, text_lens=np.asarray(lens), codes=cpad, wav_int16=wav.type(torch.int16),
         max_mel_tokens=max_mel, bucket_size=2, bucket_order=np.asarray([x["idx"] for bk in buckets for x in bk]))


@torch.no_grad()
def input_token_fixtures():
    """`inference_speech(..., input_tokens=...)` (model.py:672-686) through the reference's own forward: given tokens in the
    first forward at positions 0 .. n, generation continuing at position n + 2 - micro config, one row and a 2-row batch
    with one shared continuation prefix."""
    cfg = icfg.micro()
    g = cfg.gpt
    gpt = build_ref_gpt(cfg, 1234)
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    text = torch.from_numpy(synth.text_ids(11, 11, g.number_text_tokens)).view(1, 11).int()
    base, *_ = ref_greedy(gpt, mel, text, max_gen=24)
    given = base[:, :5].clone()
    given[0, 2] = (int(given[0, 2]) + 3) % (g.number_mel_codes - 2)  # not what greedy itself would have picked
    codes, logits, *_ = ref_greedy(gpt, mel, text, max_gen=16, input_tokens=given, n_trace=6)
    save("micro_input_tokens_b1", text=text, input_tokens=given, codes=codes, logits=logits)
    t2 = torch.stack([text[0], torch.from_numpy(synth.text_ids(11, 12, g.number_text_tokens)).int()])
    codes2, logits2, *_ = ref_greedy(gpt, mel, t2, max_gen=16, input_tokens=given, n_trace=3)
    save("micro_input_tokens_b2", text=t2, input_tokens=given, codes=codes2, logits=logits2)
    # the same continuation under beams (3-beam sample with a length penalty, 3-beam search): given tokens in every beam's prompt
    n = 14
    u = np.random.default_rng(77).random((5 + n, 2, 6), dtype=np.float32)
    cs = ref_beam_sample(gpt, mel, t2, n, u, nb=3, top_k=30, top_p=0.8, temperature=1.0, length_penalty=0.7, input_tokens=given)
    save("micro_input_tokens_beam_sample", text=t2, input_tokens=given, codes=cs, uniforms=u, num_beams=3, top_k=30, top_p=0.8,
         temperature=1.0, length_penalty=0.7, max_gen=n)
    cb = ref_beam_sample(gpt, mel, t2, n, u, nb=3, do_sample=False, length_penalty=1.0, input_tokens=given)
    save("micro_input_tokens_beam_search", text=t2, input_tokens=given, codes=cb, num_beams=3, length_penalty=1.0, max_gen=n)


@torch.no_grad()
def typical_fixtures():
    """typical_sampling=True WITHOUT sampling (model.py:690-697 appends the TypicalLogitsWarper to `logits_processor`, which
    greedy search and beam search run too): the reference's forward + its own warper class, greedy and 3-beam search."""
    cfg = icfg.micro()
    g = cfg.gpt
    gpt = build_ref_gpt(cfg, 1234)
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    text = torch.stack([torch.from_numpy(synth.text_ids(11, 11 + i, g.number_text_tokens)).int() for i in range(2)])
    plain, *_ = ref_greedy(gpt, mel, text, max_gen=24)
    codes, *_ = ref_greedy(gpt, mel, text, max_gen=24, typical_mass=0.3)
    assert not torch.equal(plain[:, : codes.shape[1]], codes[:, : plain.shape[1]]), "the filter never removed the arg-max: pick another mass"
    save("micro_greedy_typical", text=text, codes=codes, typical_mass=0.3, max_gen=24)
    u = np.zeros((20, 2, 6), dtype=np.float32)
    cb = ref_beam_sample(gpt, mel, text, 20, u, nb=3, do_sample=False, typical_mass=0.3)
    cb0 = ref_beam_sample(gpt, mel, text, 20, u, nb=3, do_sample=False)
    assert cb.shape != cb0.shape or not np.array_equal(cb, cb0), "beam search unchanged by the filter: pick another mass"
    save("micro_beam_search3_typical", text=text, codes=cb, num_beams=3, typical_mass=0.3, max_gen=20)


@torch.no_grad()
def cond_batch_fixtures():
    """A batch of prompts of different lengths (model.py:490-502 with cond_mel_lengths, :599-602 per-row conditioning): two
    prompts padded to 61 frames, lengths 61 and 45, the padding filled with noise (it must not matter); conditioning latents
    and the greedy ids of two sentences, each with its own prompt - all through the reference's own modules."""
    cfg = icfg.micro()
    g = cfg.gpt
    gpt = build_ref_gpt(cfg, 1234)
    m0 = torch.from_numpy(synth.prompt_mel(61, seed=7))
    m1 = torch.from_numpy(synth.prompt_mel(61, seed=8)).clone()
    m1[:, :, 45:] = rnd("cond_batch.pad", (1, 100, 16), std=3.0)
    mel = torch.cat([m0, m1], 0)
    lens = torch.tensor([61, 45])
    cond = gpt.get_conditioning(mel, lens)
    text = torch.stack([torch.from_numpy(synth.text_ids(11, 71 + i, g.number_text_tokens)).int() for i in range(2)])
    codes, logits, *_ = ref_greedy(gpt, mel, text, max_gen=16, n_trace=2, lens=lens)
    save("micro_cond_batch", mel=mel, lens=lens, cond=cond, text=text, codes=codes, logits=logits)


def front_fixtures():
    """Known answers of the reference's text front end (indextts/utils/front.py, utils/common.py): sentence splitting on
    token lists, CJK pre-tokenisation, and TextNormalizer.normalize with the third-party written-form normalisers replaced
    by identity stubs (WeTextProcessing is absent offline) - i.e. everything the reference itself implements."""
    import json
    import random

    from indextts.utils import common as rcommon
    from indextts.utils.front import TextNormalizer, TextTokenizer

    rnd_ = random.Random(20241004)
    alphabet = ["▁A", "B", "C", "▁D", "E", ",", "▁,", ".", "▁.", "!", "?", "▁?", "▁...", "-", "'", "▁'", "F", "G", "▁H"]
    weights = [6, 6, 6, 6, 6, 3, 1, 2, 1, 1, 1, 1, 1, 2, 1, 1, 6, 6, 6]
    splits = []
    for case in range(60):
        n = rnd_.choice([0, 1, 3, 7, 20, 45, 90, 200])
        toks = rnd_.choices(alphabet, weights=weights, k=n)
        cap = rnd_.choice([4, 8, 15, 30, 120])
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = TextTokenizer.split_sentences_by_token(list(toks), TextTokenizer.punctuation_marks_tokens, cap)
        splits.append({"tokens": toks, "cap": cap, "out": out})

    class Ident:
        def normalize(self, t):
            return t

    tn = TextNormalizer()
    tn.zh_normalizer, tn.en_normalizer = Ident(), Ident()
    texts = ["IndexTTS 正式发布1.0版本了，效果666", "晕XUAN4是一种GAN3觉", "我爱你！", "I love you!", "what's up? it's fine: ok; (yes) [no]",
             "“我爱你”的英语是“I love you”", "2.5平方电线", "共465篇，约315万字", "2002年的第一场雪，下在了2003年", "速度是10km/h",
             "现在是北京时间2025年01月11日 20:00", "他这条裤子是2012年买的，花了200块钱", "电话：135-4567-8900", "1键3连",
             "他这条视频点赞3000+，评论1000+，收藏500+", "这是1024元的手机，你要吗？", "受不liao3你了", "“衣裳”不读衣chang2，而是读衣shang5",
             "最zhong4要的是：不要chong2蹈覆辙", "不zuo1死就不会死", "See you at 8:00 AM", "8:00 AM 开会", "Couting down 3, 2, 1, go!",
             "数到3就开始：1、2、3", "This sales for 2.5% off, only $12.5.", "5G网络是4G网络的升级版，2G网络是3G网络的前身",
             "苹果于2030/1/2发布新 iPhone 2X 系列手机，最低售价仅 ¥12999", "这酒...里...有毒...", "只有,,,才是最好的", "babala2是什么？",
             "用beta1测试", "have you ever been to beta2?", "such as XTTS, CosyVoice2, Fish-Speech, and F5-TTS", "where's the money?",
             "who's there?", "which's the best?", "how's it going?", "今天是个好日子 it's a good day", "约瑟夫·高登-莱维特（Joseph Gordon-Levitt is an American actor）",
             "蒂莫西·唐纳德·库克（英文名：Timothy Donald Cook），通称蒂姆·库克（Tim Cook），美国商业经理、工业工程师和工业开发商，现任苹果公司首席执行官。",
             "《盗梦空间》是由美国华纳兄弟影片公司出品的电影，由克里斯托弗·诺兰执导并编剧", "jv2 que4 xün1 ju3", "test@example.com", "   ", "a"]
    norm = [{"text": t, "use_chinese": tn.use_chinese(t), "out": tn.normalize(t)} for t in texts]
    pin = [{"in": p, "out": tn.correct_pinyin(p)} for p in ["ju2", "que4", "xün1", "jUan3", "qu5", "xue2", "lv3", "nü3", "zhong4", "Ju1"]]
    cjk = [{"in": t, "tok": rcommon.tokenize_by_CJK_char(t), "tok_keep": rcommon.tokenize_by_CJK_char(t, do_upper_case=False)}
           for t in ["你好世界是 hello world 的中文", "  a b  ", "안녕 hello こんにちは", "abc", "", "３Ｄ打印 ｶﾀｶﾅ"]]
    detok = [{"in": t, "out": rcommon.de_tokenized_by_CJK_char(t), "out_lower": rcommon.de_tokenized_by_CJK_char(t, do_lower_case=True)}
             for t in ["你 好 世 界 是 HELLO WORLD 的 中 文", "SEE YOU!", "A-B 你 C D", ""]]
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, "front_cases.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"splits": splits, "normalize": norm, "correct_pinyin": pin, "cjk": cjk, "detok": detok}, f, ensure_ascii=False, indent=0)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--front", action="store_true")
    ap.add_argument("--long", action="store_true")
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--skip-micro", action="store_true")
    ap.add_argument("--input-tokens", action="store_true")
    ap.add_argument("--smooth", action="store_true")
    ap.add_argument("--cond-batch", action="store_true")
    ap.add_argument("--typical", action="store_true")
    ap.add_argument("--eos", action="store_true")
    ap.add_argument("--nrs", action="store_true")
    ap.add_argument("--host-beams", action="store_true")
    a = ap.parse_args()
    ref_import.install()
    torch.manual_seed(0)
    if not a.skip_micro:
        micro_fixtures()
    if a.input_tokens:
        input_token_fixtures()
    if a.smooth:
        smooth_fixtures()
    if a.cond_batch:
        cond_batch_fixtures()
    if a.typical:
        typical_fixtures()
    if a.eos:
        eos_fixtures()
    if a.nrs:
        nrs_input_token_fixtures()
    if a.host_beams:
        host_beam_fixtures()
    if a.full:
        full_fixtures()
    if a.front:
        front_fixtures()
    if a.fast:
        fast_fixtures()
    if a.long:
        long_fixtures()
