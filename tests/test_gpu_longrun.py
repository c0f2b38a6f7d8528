"""GPU: parity at the BENCHMARK shapes (L = 105 text tokens, 660 decode steps so S passes 768, latent T = 480, 64 vocoder
frames) against fixtures produced by the real reference modules (oracle/make_golden.py --long).  Every variant of the
cache-attention kernel is driven past its register window, so the streaming online-softmax loop
(csrc/decode2.hip `for (int cb = 2 * NIT; ...)`) executes under an oracle comparison:

  engine  rows  kernel                                   register window (keys)
  fp32    1, 2  decode_attn2<float, 1024 threads>        384
  fp32    26    decode_attn2<float, 256 threads, NIT 8>  256   (rows x heads >= 512)
  bf16    2     decode_attn2<bf16, 256, split 4>         768
  bf16    32    decode_attn2<bf16, 256 threads, NIT 8>   512

fp32: greedy ids bit-exact over the whole run (a divergence is accepted only at a step whose reference top-1/top-2
margin is below 1e-3, and the run is then re-checked with teacher forcing); bf16: teacher-forced logits against the
reference top-8 at S = 400 / 520 / 780 within the stated tolerance, measured values recorded in r02_accuracy.json."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import engine as ieng  # noqa: E402
from itts_hip import prng, synth  # noqa: E402

CFG = icfg.indextts_1_5()
NS = 660
S0 = 32 + 105 + 2 + 1


@pytest.fixture(scope="module")
def eng32():
    return ieng.build_engine(CFG, "fp32", parts=("gpt", "bigvgan"))


@pytest.fixture(scope="module")
def eng16():
    return ieng.build_engine(CFG, "bf16", parts=("gpt", "bigvgan"))


@pytest.fixture(scope="module")
def eng32r():
    """CONTROL for the bf16 numbers: the fp32 engine (fp32 activations, KV, accumulation) on weights rounded to bf16 -
    what ANY bf16-weight implementation of this synthetic checkpoint must lose (its init makes attention sharp on purpose,
    SURVEY 8c sensitivity warning, so rounding noise is amplified far beyond what a trained checkpoint shows)."""
    sd = synth.gpt_state_dict(CFG, 1234)
    rounded = {k: (torch.from_numpy(np.asarray(v)).to(torch.bfloat16).float().numpy() if np.asarray(v).ndim >= 2 else v) for k, v in sd.items()}
    return ieng.build_engine(CFG, "fp32", parts=("gpt",), state_dicts={"gpt": rounded})


@pytest.fixture(scope="module")
def eng32v():
    """CONTROL for the bf16 vocoder numbers: fp32 compute on exactly the values the bf16 arena holds (the packed "w" tensors
    rounded to bf16)."""
    from itts_hip import pack

    packed = pack.pack_bigvgan(synth.bigvgan_state_dict(CFG, 1234), CFG)
    rounded = {k: (t, torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy() if t == "w" else a)
               for k, (t, a) in packed.items()}
    eng = ieng.Engine(CFG, "fp32", "cuda:0")
    eng.load_packed(rounded)
    eng.finalize()
    return eng


@pytest.fixture(scope="module")
def mel():
    return torch.from_numpy(synth.prompt_mel(511, seed=7))


def rms_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-12))


def check_ids(codes_row, g, what):
    """ids equal to the reference's; a first divergence is tolerated only where the reference margin is < 1e-3."""
    ref = g["codes"][0, : len(codes_row)]
    same = codes_row == ref
    if same.all():
        return len(ref)
    k = int(np.argmin(same))
    assert g["margins"][k] < 1e-3, f"{what}: ids diverge at step {k} where the reference margin is {g['margins'][k]:.4g}"
    return k


def forced_trace(eng, cond, text, g, nrows):
    """Teacher-forced run with the reference ids: logits [len(trace_steps)][nrows, V] at the fixture's trace steps."""
    eng.set_forced(g["codes"][:, :NS])
    out = []
    try:
        eng.prefill(cond, np.repeat(text, nrows, 0) if text.shape[0] == 1 else text, NS, 10.0, True)
        done = 0  # logits of step k are available after k decode steps (step 0 comes from the prefill)
        for k in g["trace_steps"]:
            k = int(k)
            if k > done:
                eng.decode(k - done)
                done = k
            codes, lg = eng.fetch(logits=True)
            assert np.array_equal(codes[0, : k + 1], g["codes"][0, : k + 1]), k  # forcing took effect
            out.append(lg.copy())
        eng._exit()
    finally:
        eng.set_forced(None)
    return out


@pytest.mark.parametrize("nrows", [1, 2, 26])
def test_long_greedy_ids_fp32(eng32, mel, gold, nrows, accuracy):
    g = gold("long_decode_b1")
    cond = eng32.conditioning(mel)
    text = g["text"].astype(np.int32)
    if nrows == 2:  # a real second sentence next to the reference one
        text = np.concatenate([text, synth.text_ids(105, 77, CFG.gpt.number_text_tokens).reshape(1, -1).astype(np.int32)], 0)
    elif nrows > 2:
        text = np.repeat(text, nrows, 0)
    codes = eng32.generate(cond, text, NS, suppress_stop=True)
    assert codes.shape == (nrows, NS)
    agree = check_ids(codes[0], g, f"fp32 B={nrows}")
    if nrows > 2:  # replicated rows (the fp32 kernels are not bit-invariant to the row position: each row against the reference)
        agree = min([agree] + [check_ids(codes[r], g, f"fp32 B={nrows} row {r}") for r in range(1, nrows)])
    accuracy[f"fp32_long_greedy_rows{nrows}_steps_bit_exact"] = agree
    # teacher-forced logits at every trace step, S up to 799 (beyond every register window)
    lgs = forced_trace(eng32, cond, g["text"].astype(np.int32), g, nrows)
    worst = 0.0
    for i, lg in enumerate(lgs):
        for r in range(nrows):
            worst = max(worst, float(np.abs(lg[r, g["top_idx"][i]] - g["top_val"][i]).max()))
    accuracy[f"fp32_long_forced_rows{nrows}_top8_logits_max_abs_err"] = worst
    assert worst < 3e-3, worst
    assert agree == NS or agree >= 1  # divergence only at a near-tie (checked above)


def test_long_forced_logits_control_bf16_rounded_weights(eng32r, mel, gold, accuracy):
    g = gold("long_decode_b1")
    cond = eng32r.conditioning(mel)
    lgs = forced_trace(eng32r, cond, g["text"].astype(np.int32), g, 1)
    res = {int(k) + S0: rms_rel(lgs[i][0, g["top_idx"][i]], g["top_val"][i]) for i, k in enumerate(g["trace_steps"])}
    accuracy["control_fp32_compute_bf16_rounded_weights_top8_logits_rel_rms_by_S"] = res
    lat = eng32r.latent(cond, g["text"].astype(np.int32), g["codes"][0, :480]).float().cpu().numpy()[0]
    accuracy["control_fp32_compute_bf16_rounded_weights_latent_T480_rel_rms"] = rms_rel(lat[:, :16], g["latent_sample"])


@pytest.mark.parametrize("nrows", [2, 32])
def test_long_forced_logits_bf16(eng16, mel, gold, nrows, accuracy):
    """The benchmarked (bf16) engine against the reference at long S: teacher-forced, so every step sees the reference
    history.  Bound = 2x the worst value measured in round 2 (0.179 at S = 520; profiles/r02_accuracy.json) - the error
    does not grow with S, and the control above (fp32 compute on bf16-rounded weights) shows how much of it is the
    synthetic checkpoint's sensitivity to weight rounding."""
    g = gold("long_decode_b1")
    cond = eng16.conditioning(mel)
    lgs = forced_trace(eng16, cond, g["text"].astype(np.int32), g, nrows)
    res = {}
    for i, k in enumerate(g["trace_steps"]):
        lg = lgs[i]
        for r in range(1, nrows):
            assert np.array_equal(lg[r], lg[0]), (int(k), r)
        res[int(k) + S0] = rms_rel(lg[0, g["top_idx"][i]], g["top_val"][i])
        assert int(lg[0].argmax()) in set(int(x) for x in g["top_idx"][i]) or res[int(k) + S0] < 0.2
    accuracy[f"bf16_long_forced_rows{nrows}_top8_logits_rel_rms_by_S"] = res
    for S in (400, 520, 780):
        assert res[S] < 0.36, (S, res[S])
    assert max(res.values()) < 0.36, res
    early = np.mean([v for S, v in res.items() if S < 400])
    late = np.mean([v for S, v in res.items() if S >= 520])
    assert late < 1.5 * early + 0.02, (early, late)  # no degradation once the keys stream past the register windows


def test_long_latent_and_vocoder(eng32, eng16, eng32v, mel, gold, accuracy):
    g = gold("long_decode_b1")
    T = 480
    codes = g["codes"][0, :T]
    # bf16 bound = 2x measured in round 2 (0.367: see the control test for the share that is weight rounding)
    for name, eng, tol in (("fp32", eng32, 2e-3), ("bf16", eng16, 0.74)):
        cond = eng.conditioning(mel)
        lat = eng.latent(cond, g["text"].astype(np.int32), codes).float().cpu().numpy()[0]
        e1 = rms_rel(lat[:, :16], g["latent_sample"])
        e2 = rms_rel(lat[g["latent_row_idx"]], g["latent_rows"])
        accuracy[f"{name}_latent_T480_rel_rms"] = max(e1, e2)
        assert max(e1, e2) < tol, (name, e1, e2)
        assert abs(float(np.sqrt((lat.astype(np.float64) ** 2).mean())) - float(g["latent_rms"])) < 2e-2 * float(g["latent_rms"])
    w = gold("long_bigvgan")["wav"]
    lat_in = torch.from_numpy(prng.tensor("bigvgan.latent.long", 3, (1, 64, CFG.bigvgan.gpt_dim), std=1.0, mean=0.0))
    # CONTROL for the bf16 vocoder: fp32 compute on the bf16-rounded vocoder weights and the bf16-rounded latent - what any
    # bf16-weight implementation loses; the bf16 engine (bf16 activations between the layers on top) must stay within
    # 2 x control + 2e-3 and under an absolute 4e-2 (r03 measured 1.6e-2)
    errs = {}
    for name, eng in (("fp32", eng32), ("control", eng32v), ("bf16", eng16)):
        lat = lat_in if name == "fp32" else lat_in.to(torch.bfloat16).float()
        wav = eng.bigvgan(lat, eng.ecapa(mel.transpose(1, 2))).float().cpu().numpy()[0, 0]
        errs[name] = rms_rel(wav, w)
        assert wav.shape == w.shape
        accuracy[f"{name}_bigvgan_64frames_waveform_rel_rms"] = errs[name]
    assert errs["fp32"] < 1e-3, errs
    assert errs["bf16"] < 2.0 * errs["control"] + 2e-3 and errs["bf16"] < 4e-2, errs


def test_max_gen_change_on_one_engine(eng32, mel, gold):
    """max_mel_tokens is a per-request knob (webui / infer kwargs): the captured decode graph bakes max_gen in as the
    ids row stride, so it must be re-captured when max_gen changes with B / Smax unchanged."""
    g = gold("full_decode_b1")
    cond = eng32.conditioning(mel)
    text = np.concatenate([g["text"], g["text"]], 0)
    for n in (48, 24, 48, 40):
        codes = eng32.generate(cond, text, n, suppress_stop=False)
        for r in range(2):
            assert np.array_equal(codes[r], g["codes"][0, : codes.shape[1]]), (n, r)
