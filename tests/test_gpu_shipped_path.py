"""GPU: what ships by default - the eos-enabled decode loop on the persistent engine and the bf16 drop-in at IndexTTS-1.5
sizes.

The bit-equality tests of tests/test_gpu_engine_persistent.py run with the stop token suppressed (fixed length); the product
(`IndexTTS.infer()` -> `Engine.generate`) runs with eos ENABLED: the in-launch greedy sampler's bookkeeping (a row finishing
while the others run on, pad = stop token, the unfinished count `status()` reports) and, in the reference's default mode, beams
finishing into hypotheses at different steps.  The synthetic checkpoint never emits the stop token by itself, so the fixtures
`smooth_eos_*` (oracle/make_golden.py --eos: the REAL reference on the "smooth" checkpoint with mel_head.bias[stop] raised by a
bias calibrated on the reference) make three ragged rows stop at three different steps.

  * fp32 engine vs the reference fixtures: ids, pad fill and length bit-exact (greedy 3 rows; beam-sample / beam search 2 x 3);
  * bf16: persistent engine vs the launch path with eos enabled at 2 and 3 rows - ids, pad fill, the (steps, unfinished)
    sequence of status() - and 2 x 3 beam rows on the engine against each sentence's own 3-row run;
  * `IndexTTS(is_fp16=True)` at 1.5 dims: `infer()` greedy and with the reference's default kwargs on the persistent engine,
    waveform against the oracle (fp32 CPU restatement) run on the same codes, bound from a control.
Reference: indextts/infer.py:101-241, indextts/gpt/model.py:655-708."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import engine as ieng  # noqa: E402
from itts_hip import infer_core, pack, synth  # noqa: E402

CFG = icfg.indextts_1_5()
STOP = CFG.gpt.stop_mel_token


def rms_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-12))


@pytest.fixture(scope="module")
def g3(gold):
    return gold("smooth_eos_b3")


@pytest.fixture(scope="module")
def sd_eos(g3):
    return synth.gpt_state_dict(CFG, 1234, profile="smooth", stop_bias=float(g3["stop_bias"]))


@pytest.fixture(scope="module")
def mel():
    return torch.from_numpy(synth.prompt_mel(511, seed=7))


@pytest.fixture(scope="module")
def eng32e(sd_eos):
    return ieng.build_engine(CFG, "fp32", parts=("gpt",), state_dicts={"gpt": sd_eos})


@pytest.fixture(scope="module")
def eng16e(sd_eos):
    return ieng.build_engine(CFG, "bf16", parts=("gpt",), state_dicts={"gpt": sd_eos})


# ---------------------------------------------------------------- fp32 vs the reference
def test_eos_greedy_matches_reference_fp32(eng32e, mel, g3):
    """HF greedy_search with eos enabled, three ragged rows stopping at three different steps: the reference's ids (finished
    rows padded with the stop token, generation ending with the step in which the last row stops) bit for bit."""
    cond = eng32e.conditioning(mel)
    codes = eng32e.generate(cond, g3["text"].astype(np.int32), 64)
    assert len(set(int(x) for x in g3["stop_steps"])) == 3
    assert codes.shape == g3["codes"].shape and np.array_equal(codes, g3["codes"]), (codes.shape, g3["codes"].shape)
    for r, k in enumerate(g3["stop_steps"]):
        assert (codes[r, int(k):] == STOP).all() and (codes[r, : int(k)] != STOP).all()


@pytest.mark.parametrize("tag", ["sample", "search"])
def test_eos_beams_match_reference_fp32(eng32e, mel, gold, tag):
    """The reference's default generate() kwargs (beam_sample, 3 beams) and beam search on a two-sentence text, eos enabled:
    beams finish into BeamHypotheses at different steps, `done` per sentence; finalized ids bit-exact."""
    g = gold(f"smooth_eos_beam_{tag}")
    cond = eng32e.conditioning(mel)
    mg = g["uniforms"].shape[0]
    out = eng32e.generate(cond, g["text"].astype(np.int32), mg, do_sample=tag == "sample", num_beams=3, top_k=30, top_p=0.8,
                          temperature=1.0, uniforms=g["uniforms"], length_penalty=0.0)
    want = g["codes"]
    n = min(out.shape[1], want.shape[1])
    assert np.array_equal(out[:, :n], want[:, :n]), (out.shape, want.shape)
    assert (out[:, n:] == STOP).all() and (want[:, n:] == STOP).all()


# ---------------------------------------------------------------- bf16: engine vs launch path, eos enabled
def product_loop(eng, cond, text, max_gen, no_engine, chunk=8, **modes):
    """Engine.generate's own loop (status() in front of every chunk), returning what the product sees: the full id buffer
    (pad included), the status() sequence and the decode mode."""
    eng.debug(no_engine=no_engine, engine=not no_engine)
    seq = []
    try:
        eng.prefill(cond, text, max_gen, 10.0, False)
        done = 1
        while done < max_gen:
            step, unf = eng.status()
            seq.append((step, unf))
            if unf == 0:
                break
            n = min(chunk, max_gen - done)
            eng.decode(n)
            done += n
        seq.append(eng.status())
        codes = eng.fetch()
        mode = eng.decode_mode()
        eng._exit()
    finally:
        eng.debug()
    return codes, seq, mode


@pytest.mark.parametrize("rows", [2, 3])
def test_eos_engine_equals_launch_path_bf16(eng16e, mel, g3, rows):  # noqa: D401
    """suppress_stop=False on both paths: same ids, same pad fill over the WHOLE id buffer, same (steps, unfinished) read-backs.
    The rows stop at different steps (asserted), so a row is padded while another still decodes - on the engine that is the
    in-launch sampler's `unf ? choice : stop` with workgroup b owning row b."""
    cond = eng16e.conditioning(mel)
    text = g3["text"][:rows].astype(np.int32)
    want, seq0, m0 = product_loop(eng16e, cond, text, 64, no_engine=True)
    got, seq1, m1 = product_loop(eng16e, cond, text, 64, no_engine=False)
    assert (m0, m1) == (0, 1)
    assert np.array_equal(got, want)
    assert seq0 == seq1, (seq0, seq1)
    stops = [int(np.argmax(want[r] == STOP)) if (want[r] == STOP).any() else -1 for r in range(rows)]
    assert min(stops) >= 0 and len(set(stops)) == rows, stops  # every row stops, each at its own step
    nsteps = seq1[-1][0]  # steps run before the loop saw "no row unfinished" (the buffer behind them is never written)
    assert max(stops) < nsteps <= max(stops) + 1 + 8
    for r in range(rows):
        assert (want[r, stops[r]:nsteps] == STOP).all()  # a finished row is padded with the stop token while the others run on
    # (bf16 free-running ids leave the fp32 reference's at its first near-tie, so a late stop step may differ from the
    # fixture's - r04: 55 against 57; the early ones, before any parting, are the reference's)
    ref = [int(x) for x in g3["stop_steps"][:rows]]
    assert all(a == b for a, b in zip(stops, ref) if b < 16), (stops, ref)
    assert seq1[-1][1] == 0 and [u for _, u in seq1] == sorted((u for _, u in seq1), reverse=True)


@pytest.mark.parametrize("tag", ["sample", "search"])
def test_eos_beam_rows_engine_bf16(eng16e, mel, gold, tag):
    """Reference-default mode with eos enabled on the persistent engine: 1 x 3 beam rows bit-identical to the launch path;
    2 x 3 = 6 rows (the launch path runs on the matrix cores there) equal to each sentence's own 3-row engine run."""
    g = gold(f"smooth_eos_beam_{tag}")
    cond = eng16e.conditioning(mel)
    text, u = g["text"].astype(np.int32), g["uniforms"]
    mg = u.shape[0]
    kw = dict(do_sample=tag == "sample", num_beams=3, top_k=30, top_p=0.8, temperature=1.0, length_penalty=0.0)
    single = []
    for i in range(2):
        res = []
        for no_engine in (True, False):
            eng16e.debug(no_engine=no_engine, engine=not no_engine)
            try:
                res.append(eng16e.generate(cond, text[i:i + 1], mg, uniforms=np.ascontiguousarray(u[:, i:i + 1]), **kw))
                assert eng16e.decode_mode() == (0 if no_engine else 1)
            finally:
                eng16e.debug()
        assert res[0].shape == res[1].shape and np.array_equal(res[0], res[1]), i
        assert (res[1][0] == STOP).any(), "the hypothesis ends with the stop token"
        single.append(res[1][0])
    eng16e.debug(engine=True)
    try:
        both = eng16e.generate(cond, text, mg, uniforms=u, **kw)
        assert eng16e.decode_mode() == 1
    finally:
        eng16e.debug()
    for i in range(2):
        n = single[i].shape[0]
        assert np.array_equal(both[i, :n], single[i]) and (both[i, n:] == STOP).all(), i


# ---------------------------------------------------------------- the drop-in at 1.5 dims, is_fp16=True
def round_packed(packed):
    """bf16-round what the bf16 arena rounds ("w" tensors); the fp32 engine then computes in fp32 on those values: the CONTROL."""
    out = {}
    for k, (tag, arr) in packed.items():
        if tag == "w":
            arr = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(torch.bfloat16).float().numpy()
        out[k] = (tag, arr)
    return out


@pytest.fixture(scope="module")
def sd_voc():
    return synth.bigvgan_state_dict(CFG, 1234)


@pytest.fixture(scope="module")
def tts16(sd_eos, sd_voc):
    from indextts.infer import IndexTTS

    return IndexTTS(cfg=CFG, model_dir="/nonexistent", is_fp16=True, state_dicts={"gpt": sd_eos, "bigvgan": sd_voc})


@pytest.fixture(scope="module")
def control(sd_eos, sd_voc):
    """fp32 compute on bf16-rounded weights (GPT + vocoder): what any bf16-weight implementation loses."""
    eng = ieng.Engine(CFG, "fp32", "cuda:0")
    eng.load_packed(round_packed(pack.pack_gpt(sd_eos, CFG)))
    eng.load_packed(round_packed(pack.pack_bigvgan(sd_voc, CFG)))
    eng.finalize()
    return eng


def oracle_wave(sd_eos, sd_voc, mel, sents, code_rows):
    """fp32 CPU restatement (the checker): latent pass + ECAPA + BigVGAN on the given codes, per sentence as infer.py:134-212."""
    from oracle import gpt as ogpt
    from oracle import vocoder as ovoc

    wg, wb = ogpt.to_torch(sd_eos), ogpt.to_torch(sd_voc)
    with torch.no_grad():
        cond = ogpt.get_conditioning(mel, wg, CFG.gpt)
        wavs = []
        for ids, codes in zip(sents, code_rows):
            lat = ogpt.latent_forward(cond, torch.from_numpy(np.asarray(ids)).view(1, -1), torch.from_numpy(np.asarray(codes)).view(1, -1), wg, CFG.gpt)
            wavs.append(ovoc.bigvgan_forward(lat, mel.transpose(1, 2), wb, CFG.bigvgan, icfg.ecapa_dims(CFG.bigvgan))[0, 0].numpy())
    return wavs


def clean_rows(codes):
    out = []
    for r in np.asarray(codes):
        c, n = infer_core.remove_long_silence(r[None], STOP)
        out.append(c[0, : int(n[0])])
    return out


def test_indextts_fp16_default_path_greedy(tts16, control, sd_eos, sd_voc, mel, g3, accuracy):
    """`IndexTTS(is_fp16=True)` -> bf16 engine -> persistent decode engine -> eos-enabled product loop -> remove_long_silence
    -> latent -> vocoder, at 1.5 dims, two sentences (= 2 decode rows).  Ids: the fp32 reference's (a row may part from them only
    at a reference near-tie).  Waveform: against the oracle on the product's own codes, bound 2 x control + 2e-3."""
    sents = [g3["text"][r, : int(g3["text_lens"][r])].astype(np.int32) for r in range(2)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        sr, wav = tts16.infer(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=64, do_sample=False, num_beams=1)
        assert tts16.engine.decode_mode() == 1, "the product default must run on the persistent decode engine"
        codes = tts16.gpt.inference_speech(mel.cuda(), torch.from_numpy(g3["text"][:2].astype(np.int32)), do_sample=False, num_beams=1,
                                           repetition_penalty=10.0, max_generate_length=64).cpu().numpy()
    assert sr == 24000 and wav.dtype == np.int16 and wav.ndim == 2 and wav.shape[1] == 1
    ref = g3["codes"][:2]
    n = min(codes.shape[1], ref.shape[1])
    for r in range(2):
        same = codes[r, :n] == ref[r, :n]
        if not same.all():
            k = int(np.argmin(same))
            assert float(g3["margins"][r, k]) < 0.04, f"row {r} parts from the reference ids at step {k}, margin {g3['margins'][r, k]:.3f}"
    rows = clean_rows(codes)
    assert sum(len(r) for r in rows) * 1024 == wav.shape[0]
    want = np.concatenate(oracle_wave(sd_eos, sd_voc, mel, sents, rows))
    # control: fp32 compute on the bf16-rounded weights, same codes
    cond_c = control.conditioning(mel)
    spk_c = control.ecapa(mel.transpose(1, 2))
    ctl = np.concatenate([control.bigvgan(control.latent(cond_c, s, c), spk_c).float().cpu().numpy()[0, 0] for s, c in zip(sents, rows)])
    got = wav[:, 0].astype(np.float64) / 32767.0
    e, ec = rms_rel(got, want), rms_rel(ctl, want)
    accuracy["dropin_fp16_greedy_waveform_rel_rms_vs_oracle"] = e
    accuracy["dropin_fp16_greedy_waveform_rel_rms_control"] = ec
    print(f"IndexTTS(is_fp16=True).infer greedy: waveform rel-RMS {e:.4f} (control {ec:.4f})")
    assert e < 2.0 * ec + 2e-3 + 1.0 / 32767 / max(float(np.sqrt((want ** 2).mean())), 1e-6), (e, ec)


def test_indextts_fp16_default_kwargs(tts16, mel, g3):
    """infer() with NO generation kwargs = the reference's defaults (infer.py:116-124: do_sample, 3 beams, top_k 30, top_p 0.8,
    repetition_penalty 10): 2 sentences x 3 beams = 6 rows on the persistent engine; torch.manual_seed makes it reproducible."""
    sents = [g3["text"][r, : int(g3["text_lens"][r])].astype(np.int32) for r in range(2)]
    outs = []
    for _ in range(2):
        torch.manual_seed(1234)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            sr, wav = tts16.infer(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=72)
        assert tts16.engine.decode_mode() == 1
        outs.append(wav)
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])
    assert outs[0].shape[0] % 1024 == 0 and 2 * 1024 <= outs[0].shape[0] <= 2 * 72 * 1024
    assert np.abs(outs[0]).max() > 0


# ---------------------------------------------------------------- beams with the TopK warper off / wider than the device sampler
@pytest.fixture(scope="module")
def eng32s():
    return ieng.build_engine(CFG, "fp32", parts=("gpt",), state_dicts={"gpt": synth.gpt_state_dict(CFG, 1234, profile="smooth")})


@pytest.mark.parametrize("tag", ["topk0", "topk200"])
def test_beam_sample_any_top_k_matches_reference_fp32(eng32s, mel, gold, tag):
    """beam_sample with `top_k = 0` (HF: TopK warper off; infer.py:116-124 forwards the kwarg verbatim, webui.py:393-402 offers 0)
    and top_k = 200 (> the device sampler's 128 candidates per beam): the warpers and the draws run on the host
    (infer_core.host_beam_step), BeamSearchScorer.process / the beam re-ordering / finalize on the device
    (itts_gpt_commit_beams).  Ids bit-exact against the reference fixture (installed transformers warpers, shared uniforms)."""
    g = gold(f"smooth_beam_{tag}")
    cond = eng32s.conditioning(mel)
    mg = int(g["max_gen"])
    out = eng32s.generate(cond, g["text"].astype(np.int32), mg, do_sample=True, num_beams=int(g["num_beams"]), top_k=int(g["top_k"]),
                          top_p=float(g["top_p"]), temperature=float(g["temperature"]), uniforms=g["uniforms"], length_penalty=0.0)
    want = g["codes"]
    n = min(out.shape[1], want.shape[1])
    assert np.array_equal(out[:, :n], want[:, :n]), (out, want)
    # the drop-in forwards the kwarg un-clamped and without a warning
    kw = infer_core.sampling_kwargs(True, 3, int(g["top_k"]), 0.8, 1.0)
    assert kw["top_k"] == int(g["top_k"]) and kw["num_beams"] == 3


def test_beam_sample_wide_top_k0_equals_the_oracle_step_by_step(eng32s, mel):
    """The WIDE case: top_k = 0 at temperature 1 keeps ~6000 tokens per beam on this checkpoint (18000 flat candidates per draw) -
    a draw then depends on the last bits of the logits, which differ between the GPU's fp32 forward and the CPU reference's, so no
    stored id sequence can pin it.  Instead the product's whole generation (host_beam_step + device scorer) must equal a loop the
    TEST drives through the C ABI with picks computed by the oracle on the same logits: torch.log_softmax -> the installed
    transformers RepetitionPenalty / TopP(min_tokens_to_keep 2) classes -> oracle.hf_beam.beam_sample_step - what
    oracle/make_golden.ref_beam_sample runs per step over the reference's forward.  2 sentences x 3 beams, 12 steps."""
    import ctypes as C

    from itts_hip import lib as L
    from oracle import hf_beam

    tlp = pytest.importorskip("transformers.generation.logits_process")
    eng = eng32s
    cond = eng.conditioning(mel)
    items, nb, mg = 2, 3, 12
    V, stop, start = CFG.gpt.number_mel_codes, STOP, CFG.gpt.start_mel_token
    text = np.stack([synth.text_ids(105, 161 + i, CFG.gpt.number_text_tokens) for i in range(items)]).astype(np.int32)
    u = np.random.default_rng(41).random((mg, items, 2 * nb), dtype=np.float32)
    got = eng.generate(cond, text, mg, do_sample=True, num_beams=nb, top_k=0, top_p=0.8, temperature=1.0, uniforms=u, length_penalty=0.0)
    # the same generation, driven from here with the ORACLE's picks
    widths = []
    L.check(eng.lib.itts_gpt_set_host_sampling(eng.h, 1))
    try:
        eng.set_beam_sample(nb, 0, 0.8, 1.0, None, do_sample=True, length_penalty=0.0, host=True)
        eng.prefill(cond, text, mg, 10.0, False)
        hist = np.empty((items * nb, mg), dtype=np.int32)
        scores = np.empty(items * nb, dtype=np.float32)
        done = np.empty(items, dtype=np.int32)
        step = C.c_int()
        while True:
            L.check(eng.lib.itts_gpt_beam_state(eng.h, hist.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p),
                                                done.ctypes.data_as(C.c_void_p), C.byref(step), eng._s()))
            k = step.value
            if k >= mg or done.all():
                break
            lg = np.empty((items * nb, V), dtype=np.float32)
            L.check(eng.lib.itts_gpt_fetch(eng.h, None, lg.ctypes.data_as(C.c_void_p), eng._s()))
            psc = np.zeros((items, 2 * nb), np.float32)
            ptok = np.full((items, 2 * nb), stop, np.int32)
            pbeam = np.zeros((items, 2 * nb), np.int32)
            for bi in range(items):
                if done[bi]:
                    continue
                cands = []
                for r in range(nb):
                    row = bi * nb + r
                    ids = torch.tensor([[1, start] + [int(t) for t in hist[row, :k]]])
                    sc = torch.log_softmax(torch.from_numpy(lg[row])[None], dim=-1)
                    sc = tlp.RepetitionPenaltyLogitsProcessor(10.0)(ids, sc.clone())
                    sc = tlp.TopPLogitsWarper(top_p=0.8, min_tokens_to_keep=2)(ids, sc)[0].numpy()
                    keep = np.nonzero(np.isfinite(sc))[0]
                    cands.append((keep, sc[keep]))
                    widths.append(len(keep))
                ws, wt, wb = hf_beam.beam_sample_step(cands, scores[bi * nb:(bi + 1) * nb], V, u[k, bi])
                psc[bi], ptok[bi], pbeam[bi] = ws, wt, wb  # (already sorted: the device's stable sort leaves them as they are)
            L.check(eng.lib.itts_gpt_commit_beams(eng.h, psc.ctypes.data_as(C.c_void_p), ptok.ctypes.data_as(C.c_void_p),
                                                  pbeam.ctypes.data_as(C.c_void_p), eng._s()))
            if k + 1 >= mg:
                break
            eng.decode(1)
        nstep, _ = eng.status()
        want = eng.fetch()[:, :nstep].astype(np.int64)
        eng._exit()
    finally:
        eng.set_beam_sample(1)
        L.check(eng.lib.itts_gpt_set_host_sampling(eng.h, 0))
    assert min(widths) > 128, f"the case is meant to exceed the device sampler's 128 candidates per beam (kept {min(widths)} .. {max(widths)})"
    n = min(got.shape[1], want.shape[1])
    assert got.shape[0] == items and np.array_equal(got[:, :n], want[:, :n]), (got, want)


def test_speaker_encoder_on_the_side_stream_is_the_same_embedding(mel):
    """Engine.ecapa(overlap=True) (IndexTTS.infer: the speaker encoder on the engine's side stream, with its own scratch arena, beside
    the conditioning encoder and the prefill): the embedding, the conditioning latents computed beside it, the prefill logits and the
    waveform of a vocoder call that consumes the overlapped embedding are bit-identical to the serial order - over several rounds,
    so that a race between the two arenas would have its chances."""
    cfg = icfg.indextts_1_5()
    eng = ieng.build_engine(cfg, "bf16", parts=("gpt", "bigvgan"))
    texts = np.stack([synth.text_ids(40, 5 + i, cfg.gpt.number_text_tokens) for i in range(2)]).astype(np.int32)
    lat = torch.randn(1, 24, cfg.bigvgan.gpt_dim, generator=torch.Generator().manual_seed(3))
    spk0 = eng.ecapa(mel.transpose(1, 2))
    cond0 = eng.conditioning(mel)
    eng.prefill(cond0, texts, 4, 10.0, True)
    lg0 = eng.fetch(logits=True)[1].copy()
    eng._exit()
    wav0 = eng.bigvgan(lat, spk0)
    torch.cuda.synchronize()
    for _ in range(4):
        spk = eng.ecapa(mel.transpose(1, 2), overlap=True)
        cond = eng.conditioning(mel)
        eng.prefill(cond, texts, 4, 10.0, True)
        lg = eng.fetch(logits=True)[1].copy()
        eng._exit()
        wav = eng.bigvgan(lat, spk)
        eng.join_side()
        torch.cuda.synchronize()
        assert torch.equal(spk, spk0) and torch.equal(cond, cond0)
        assert np.array_equal(lg, lg0) and torch.equal(wav, wav0)
