"""GPU: accuracy of the BENCHMARKED dtype (bf16) with bounds that can fail.

The parity fixtures use a synthetic checkpoint whose attention is sharp on purpose (one changed text id flips greedy ids:
SURVEY.md 8c); that init amplifies bf16 weight rounding to 0.1 - 0.3 relative, so on it a bf16 bound can only say "within a
factor of two of a noisy control".  These tests use the "smooth" profile of the same generator (Q / K gain 1, soft attention -
the conditioning of a trained checkpoint; one changed text id still flips ids) and fixtures produced from it by the REAL
reference modules (oracle/make_golden.py --smooth, IndexTTS-1.5 sizes):

  * CONTROL: fp32 engine on bf16-rounded weights - what any bf16-weight implementation must lose (measured < 1e-2);
  * the bf16 engine, teacher-forced, at 2 rows (persistent decode engine / GEMV path) and 32 rows (MFMA path): top-8 logits
    against the reference up to S = 520, bound 5e-2 relative RMS and a small multiple of the control;
  * latent pass T = 480: bound 5e-2;
  * six different sentences decoded FREE-RUNNING as one batch on the MFMA path against the reference ids: a row may leave
    the reference sequence only at a step whose reference top-1 / top-2 margin is small (tests/padding_test.py:35-96 is the
    reference's own batched greedy flow).
  * BASELINE config 5 (fp8-e4m3 GPT weights, power-of-two row scales; bf16 activations / KV): the same teacher-forced top-8
    logits and the latent against the fp32 reference - what the quantisation costs, with a bound that can fail.
Measured values go to gpurun_out/r04_accuracy.json (committed under profiles/)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import engine as ieng  # noqa: E402
from itts_hip import synth  # noqa: E402

CFG = icfg.indextts_1_5()
S0 = 32 + 105 + 2 + 1
# relative RMS against the reference; measured in r03 (profiles/r03_accuracy.json): logits 2.0e-3 .. 3.4e-3 (control 0.8e-3 ..
# 2.9e-3), latent 7.1e-3 (control 4.6e-3) - the bounds are ~3 x the measurement and, below, 2 x the control + 2e-3
BOUND = 1e-2         # top-8 logits
BOUND_LATENT = 2e-2  # latent T = 480


@pytest.fixture(scope="module")
def sd_smooth():
    return synth.gpt_state_dict(CFG, 1234, profile="smooth")


@pytest.fixture(scope="module")
def eng16s(sd_smooth):
    return ieng.build_engine(CFG, "bf16", parts=("gpt",), state_dicts={"gpt": sd_smooth})


@pytest.fixture(scope="module")
def eng32rs(sd_smooth):
    rounded = {k: (torch.from_numpy(np.asarray(v)).to(torch.bfloat16).float().numpy() if np.asarray(v).ndim >= 2 else v)
               for k, v in sd_smooth.items()}
    return ieng.build_engine(CFG, "fp32", parts=("gpt",), state_dicts={"gpt": rounded})


@pytest.fixture(scope="module")
def mel():
    return torch.from_numpy(synth.prompt_mel(511, seed=7))


def rms_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-12))


def forced_trace(eng, cond, g, nrows):
    """Teacher-forced run with the reference ids: {step: logits [nrows, V]} at the fixture's trace steps."""
    ns = g["codes"].shape[1]
    eng.set_forced(g["codes"][:, :ns])
    out = {}
    try:
        eng.prefill(cond, np.repeat(g["text"].astype(np.int32), nrows, 0), ns, 10.0, True)
        done = 0
        for k in g["trace_steps"]:
            k = int(k)
            if k > done:
                eng.decode(k - done)
                done = k
            codes, lg = eng.fetch(logits=True)
            assert np.array_equal(codes[0, : k + 1], g["codes"][0, : k + 1]), k  # forcing took effect
            out[k] = lg.copy()
        eng._exit()
    finally:
        eng.set_forced(None)
    return out


def top8_err(lgs, g, row=0):
    return {int(k) + S0: rms_rel(lgs[int(k)][row, g["top_idx"][i]], g["top_val"][i]) for i, k in enumerate(g["trace_steps"])}


@pytest.fixture(scope="module")
def control(eng32rs, mel, gold):
    g = gold("smooth_decode_b1")
    cond = eng32rs.conditioning(mel)
    lg = top8_err(forced_trace(eng32rs, cond, g, 1), g)
    lat = eng32rs.latent(cond, g["text"].astype(np.int32), g["codes"][0, :480]).float().cpu().numpy()[0]
    return {"logits": lg, "latent": rms_rel(lat[:, :16], g["latent_sample"])}


def test_control_is_small(control, accuracy):
    """The premise of this file: on the smooth checkpoint weight rounding alone costs < 2e-2."""
    accuracy["smooth_control_fp32_compute_bf16_rounded_weights_top8_logits_rel_rms_by_S"] = control["logits"]
    accuracy["smooth_control_fp32_compute_bf16_rounded_weights_latent_T480_rel_rms"] = control["latent"]
    assert max(control["logits"].values()) < 2e-2 and control["latent"] < 2e-2, control


@pytest.mark.parametrize("nrows", [2, 32])
def test_bf16_forced_logits(eng16s, mel, gold, control, accuracy, nrows):
    g = gold("smooth_decode_b1")
    cond = eng16s.conditioning(mel)
    lgs = forced_trace(eng16s, cond, g, nrows)
    for k, lg in lgs.items():
        for r in range(1, nrows):
            assert np.array_equal(lg[r], lg[0]), (k, r)  # identical rows of a batch: identical logits
    res = top8_err(lgs, g)
    accuracy[f"smooth_bf16_forced_rows{nrows}_top8_logits_rel_rms_by_S"] = res
    worst_c = max(control["logits"].values())
    assert max(res.values()) < BOUND, res
    assert max(res.values()) < 2.0 * worst_c + 2e-3, (res, worst_c)  # bf16 activations / cache on top of the rounded weights
    for i, k in enumerate(g["trace_steps"]):  # the arg-max stays the reference's wherever the reference is not in a near-tie
        if g["top_val"][i][0] - g["top_val"][i][1] > 0.25:
            assert int(lgs[int(k)][0].argmax()) == int(g["top_idx"][i][0]), (int(k), g["top_val"][i][:2])


def test_bf16_latent(eng16s, mel, gold, control, accuracy):
    g = gold("smooth_decode_b1")
    cond = eng16s.conditioning(mel)
    lat = eng16s.latent(cond, g["text"].astype(np.int32), g["codes"][0, :480]).float().cpu().numpy()[0]
    e = max(rms_rel(lat[:, :16], g["latent_sample"]), rms_rel(lat[g["latent_row_idx"]], g["latent_rows"]))
    accuracy["smooth_bf16_latent_T480_rel_rms"] = e
    assert e < BOUND_LATENT, e
    assert e < 2.0 * control["latent"] + 2e-3, (e, control["latent"])
    assert abs(float(np.sqrt((lat.astype(np.float64) ** 2).mean())) - float(g["latent_rms"])) < 1e-2 * float(g["latent_rms"])


def test_ksplit_changes_only_the_summation_order(eng16s, mel, gold, control, accuracy, monkeypatch):
    """The few-tile GEMMs of small batches (one-sentence prefill mlp.c_proj, batch-1 latent pass) split K over workgroups and add
    the shares in split order (c_api.cpp gemm_ksplit_plan): a different fp32 summation order than the unsplit kernels of large
    batches, so bit-identity ACROSS batch sizes holds only with ITTS_GEMM_KSPLIT=0 (tested in test_gpu_configs.py /
    test_gpu_fullsize.py).  Here: what the order change costs - prefill logits and the T = 480 latent with and without the split
    differ by less than the bf16 engine differs from the reference (control), and both modes meet the reference bound."""
    g = gold("smooth_decode_b1")
    cond = eng16s.conditioning(mel)
    got = {}
    for mode in ("auto", "0"):
        if mode == "0":
            monkeypatch.setenv("ITTS_GEMM_KSPLIT", "0")
        eng16s.prefill(cond, np.repeat(g["text"].astype(np.int32), 2, 0), 4, 10.0, True)
        lg = eng16s.fetch(logits=True)[1].copy()
        eng16s._exit()
        lat = eng16s.latent(cond, g["text"].astype(np.int32), g["codes"][0, :480]).float().cpu().numpy()[0]
        got[mode] = (lg, lat)
    monkeypatch.delenv("ITTS_GEMM_KSPLIT")
    d_lg = rms_rel(got["auto"][0][0], got["0"][0][0])
    d_lat = rms_rel(got["auto"][1], got["0"][1])
    accuracy["smooth_bf16_ksplit_vs_unsplit_prefill_logits_rel_rms"] = d_lg
    accuracy["smooth_bf16_ksplit_vs_unsplit_latent_T480_rel_rms"] = d_lat
    assert 0.0 < d_lat, "the split did not engage on the batch-1 latent pass"
    assert d_lg < BOUND / 2 and d_lat < BOUND_LATENT / 2, (d_lg, d_lat)
    for mode in got:
        e = rms_rel(got[mode][1][:, :16], g["latent_sample"])
        assert e < BOUND_LATENT and e < 2.0 * control["latent"] + 2e-3, (mode, e)


@pytest.mark.parametrize("path", ["mfma", "engine"])
def test_bf16_free_running_six_rows(eng16s, mel, gold, accuracy, path):
    """Six different sentences as ONE decode batch, greedy, free-running, on both six-row paths - "mfma": the launch path
    (skinny MFMA projections, 256-thread cache attention), "engine": the persistent decode engine (the default at <= 6 rows
    from r03 on; GEMV arithmetic) - against the reference's batched greedy ids: a row may part from the reference only where the reference's
    own top-1 / top-2 margin is below 0.04 (logit std 1.0, |top logit| ~ 4: the bf16 logits carry 3e-3 relative = ~0.01 - 0.02
    absolute error; r03 measured 0.0005 .. 0.018 at the parting steps: profiles/r03_accuracy.json) - a wrong kernel parts at a
    step with a comfortable margin."""
    g = gold("smooth_decode_b6")
    cond = eng16s.conditioning(mel)
    ns = g["codes"].shape[1]
    eng16s.debug(no_engine=path == "mfma", engine=path == "engine")
    try:
        eng16s.prefill(cond, g["text"].astype(np.int32), ns, 10.0, True)
        _, lg0 = eng16s.fetch(logits=True)
        eng16s.decode(ns - 1)
        codes = eng16s.fetch()
        eng16s._exit()
        assert eng16s.decode_mode() == (1 if path == "engine" else 0)
    finally:
        eng16s.debug()
    tag = "" if path == "mfma" else "_engine"
    e0 = [rms_rel(lg0[r, g["top_idx0"][r]], g["top_val0"][r]) for r in range(6)]
    accuracy[f"smooth_bf16_rows6{tag}_first_step_top8_logits_rel_rms"] = max(e0)
    assert max(e0) < BOUND, e0
    agree, at = [], []
    for r in range(6):
        same = codes[r] == g["codes"][r]
        k = ns if same.all() else int(np.argmin(same))
        agree.append(k)
        at.append(float(g["margins"][r, k]) if k < ns else None)
    accuracy[f"smooth_bf16_rows6{tag}_free_running_steps_equal_to_reference"] = agree
    accuracy[f"smooth_bf16_rows6{tag}_reference_margin_at_the_parting_step"] = at
    for r in range(6):
        assert at[r] is None or at[r] < 0.04, f"row {r}: ids part at step {agree[r]} where the reference margin is {at[r]:.3f}"
    assert sum(agree) >= 6 * 8 and sorted(agree)[-2] >= 16, agree  # most rows follow the reference for a while


# ---- BASELINE config 5: fp8-e4m3 GPT weights (per-row power-of-two scales), bf16 activations / KV ----
# e4m3 keeps 3 mantissa bits: a weight moves by up to 2^-4 relative (bf16: 2^-9).  Measured in r04 (profiles/r04_accuracy.json):
# top-8 logits 1.2e-2 .. 2.2e-2 relative RMS at 2 and at 20 rows (bf16 weights: 2.0e-3 .. 3.4e-3), flat in S, the arg-max equal to
# the reference's at 10 of the 12 traced steps; latent T = 480 5.8e-2 (bf16: 7.1e-3).  The bounds are 2 x the measurement.
BOUND_FP8 = 0.045
BOUND_FP8_LATENT = 0.12


@pytest.fixture(scope="module")
def eng8s(sd_smooth):
    return ieng.build_engine(CFG, "bf16", parts=("gpt",), state_dicts={"gpt": sd_smooth}, gpt_fp8="fp8")


@pytest.mark.parametrize("nrows", [2, 20])
def test_fp8_forced_logits(eng8s, mel, gold, control, accuracy, nrows):
    """Teacher-forced on the reference ids, 2 rows (fp8 GEMV reader) and 20 rows (config 5's batch: fp8 fragment tiles on the
    matrix cores): top-8 logits of the fp8-weight engine against the fp32 reference, up to S = 619."""
    g = gold("smooth_decode_b1")
    cond = eng8s.conditioning(mel)
    lgs = forced_trace(eng8s, cond, g, nrows)
    assert eng8s.decode_mode() == 0  # the persistent engine streams bf16 weights only
    for k, lg in lgs.items():
        for r in range(1, nrows):
            assert np.array_equal(lg[r], lg[0]), (k, r)
    res = top8_err(lgs, g)
    accuracy[f"smooth_fp8_weights_forced_rows{nrows}_top8_logits_rel_rms_by_S"] = res
    agree = [int(lgs[int(k)][0].argmax()) == int(g["top_idx"][i][0]) for i, k in enumerate(g["trace_steps"])]
    accuracy[f"smooth_fp8_weights_forced_rows{nrows}_argmax_equal_to_reference"] = f"{sum(agree)} of {len(agree)} traced steps"
    assert max(res.values()) < BOUND_FP8, res
    early = np.mean([v for S, v in res.items() if S < 300])
    late = np.mean([v for S, v in res.items() if S >= 400])
    assert late < 1.5 * early + 0.02, (early, late)  # no growth with the sequence length


def test_fp8_latent(eng8s, mel, gold, accuracy):
    g = gold("smooth_decode_b1")
    cond = eng8s.conditioning(mel)
    lat = eng8s.latent(cond, g["text"].astype(np.int32), g["codes"][0, :480]).float().cpu().numpy()[0]
    e = max(rms_rel(lat[:, :16], g["latent_sample"]), rms_rel(lat[g["latent_row_idx"]], g["latent_rows"]))
    accuracy["smooth_fp8_weights_latent_T480_rel_rms"] = e
    assert e < BOUND_FP8_LATENT, e
