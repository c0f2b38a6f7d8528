"""GPU: engine-level parity (through the C ABI) against the golden fixtures of the real reference and the
oracle.  fp32 engine = parity path (bit-exact greedy ids, fp32 tolerances); bf16 engine = throughput path
(stated tolerances, ids compared with a divergence report)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import engine as ieng  # noqa: E402
from itts_hip import synth  # noqa: E402
from oracle import gpt as ogpt  # noqa: E402
from oracle import vocoder as ovoc  # noqa: E402

CFG = icfg.micro()


@pytest.fixture(scope="module")
def eng32():
    return ieng.build_engine(CFG, "fp32")


@pytest.fixture(scope="module")
def eng16():
    return ieng.build_engine(CFG, "bf16")


def relerr(a, b):
    a = torch.as_tensor(np.asarray(a.float().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rms_rel(a, b):
    a = torch.as_tensor(np.asarray(a.float().cpu() if isinstance(a, torch.Tensor) else a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float(((a - b) ** 2).mean().sqrt() / ((b ** 2).mean().sqrt() + 1e-12))


def test_conditioning_fp32(eng32, gold):
    g = gold("micro_conditioning")
    eng32.debug(taps=True)
    cond = eng32.conditioning(torch.from_numpy(g["mel"]))
    conf = eng32.fetch_tap("conformer_out").reshape(g["conformer_out"].shape)
    eng32.debug(taps=False)
    assert relerr(conf, g["conformer_out"]) < 1e-4
    assert relerr(cond, g["cond"]) < 1e-4


def test_conditioning_bf16(eng16, gold):
    g = gold("micro_conditioning")
    cond = eng16.conditioning(torch.from_numpy(g["mel"]))
    assert rms_rel(cond, g["cond"]) < 3e-2


@pytest.mark.parametrize("name,max_gen", [("micro_decode_b1", 24), ("micro_decode_b5", 24), ("micro_decode_ragged", 20)])
def test_greedy_ids_bit_exact_fp32(eng32, gold, name, max_gen):
    """Emitted token ids must equal the reference's (golden) under greedy decode; first-step logits to fp32 tol."""
    c, g = gold("micro_conditioning"), gold(name)
    cond = torch.from_numpy(c["cond"])
    eng32.debug(taps=True)
    eng32.prefill(cond, g["text"], max_gen)
    lg0 = eng32.fetch_tap("logits0").reshape(g["text"].shape[0], -1)
    if "prefix_emb" in g:
        pe = eng32.fetch_tap("prefix_emb").reshape(g["prefix_emb"].shape[0], -1, g["prefix_emb"].shape[2])
        assert relerr(pe[:, :-1], g["prefix_emb"]) < 1e-5
    eng32.debug(taps=False)
    assert relerr(lg0, g["logits"][:, 0]) < 2e-4
    codes = eng32.generate(cond, g["text"], max_gen)
    assert codes.shape == g["codes"].shape, (codes.shape, g["codes"].shape)
    assert np.array_equal(codes, g["codes"])


def test_greedy_graph_equals_eager(eng32, gold):
    c, g = gold("micro_conditioning"), gold("micro_decode_b5")
    cond = torch.from_numpy(c["cond"])
    a = eng32.generate(cond, g["text"], 24)
    eng32.debug(no_graph=True)
    b = eng32.generate(cond, g["text"], 24)
    eng32.debug()
    assert np.array_equal(a, b)


def test_decode_logits_trace_fp32(eng32, gold):
    """Per-step logits (not only ids) against the reference trace: catches position/mask bugs that ids hide."""
    c, g = gold("micro_conditioning"), gold("micro_decode_b1")
    cond = torch.from_numpy(c["cond"])
    eng32.prefill(cond, g["text"], 24)
    n = g["logits"].shape[1]
    for k in range(n):
        _, lg = eng32.fetch(logits=True)
        assert relerr(lg, g["logits"][:, k]) < 3e-4, k
        if k + 1 < n:
            eng32.decode(1)
    eng32._exit()


def test_greedy_bf16_report(eng16, gold):
    c, g = gold("micro_conditioning"), gold("micro_decode_b1")
    codes = eng16.generate(torch.from_numpy(c["cond"]), g["text"], 24)
    n = min(codes.shape[1], g["codes"].shape[1])
    same = (codes[0, :n] == g["codes"][0, :n])
    first_div = int(np.argmin(same)) if not same.all() else n
    print(f"bf16 greedy: first divergence at step {first_div} of {n}")
    assert first_div >= 1  # the first token must survive bf16 rounding on this fixture


def test_decode_logits_trace_bf16(eng16, gold):
    """bf16 throughput path (dot2 GEMV, bf16 KV cache): per-step logits within a stated tolerance of the fp32
    reference trace for as long as the greedy ids agree."""
    c, g = gold("micro_conditioning"), gold("micro_decode_b1")
    cond = torch.from_numpy(c["cond"])
    eng16.prefill(cond, g["text"], 24)
    n = g["logits"].shape[1]
    worst = 0.0
    for k in range(n):
        codes, lg = eng16.fetch(logits=True)
        worst = max(worst, rms_rel(lg, g["logits"][:, k]))
        if codes[0, k] != g["codes"][0, k]:
            break
        if k + 1 < n:
            eng16.decode(1)
    eng16._exit()
    print(f"bf16 decode: {k + 1} steps compared, worst logits rel-RMS {worst:.4f}")
    assert k >= 3 and worst < 5e-2


def test_latent(eng32, eng16, gold):
    c, g = gold("micro_conditioning"), gold("micro_latent")
    cond = torch.from_numpy(c["cond"])
    lat = eng32.latent(cond, g["text"], g["codes"])
    assert relerr(lat, g["latent"]) < 1e-4
    lat16 = eng16.latent(cond, g["text"], g["codes"])
    assert rms_rel(lat16, g["latent"]) < 3e-2


def test_latent_batch_equals_single(eng32, gold):
    """Stacked (left-padded, masked) sentences give each sentence its batch-1 latent."""
    c, g = gold("micro_conditioning"), gold("micro_latent")
    cond = torch.from_numpy(c["cond"])
    t2 = synth.text_ids(7, 3, CFG.gpt.number_text_tokens).astype(np.int32)
    c2 = synth.text_ids(13, 4, CFG.gpt.stop_mel_token - 1).astype(np.int32)
    one = eng32.latent(cond, g["text"], g["codes"])
    two = eng32.latent(cond, t2, c2)
    both = eng32.latent_batch(cond, [g["text"], t2], [g["codes"], c2])
    assert relerr(both[0], one.cpu().numpy()) < 2e-5 and relerr(both[1], two.cpu().numpy()) < 2e-5
    assert relerr(both[0], g["latent"]) < 1e-4


def test_ecapa(eng32, eng16, gold):
    g = gold("micro_ecapa")
    mel = torch.from_numpy(g["mel"]).transpose(1, 2)
    assert relerr(eng32.ecapa(mel), g["spk"][:, 0]) < 1e-4
    assert rms_rel(eng16.ecapa(mel), g["spk"][:, 0]) < 3e-2


def test_bigvgan_fp32(eng32, gold):
    g = gold("micro_bigvgan")
    mel = torch.from_numpy(g["mel"]).transpose(1, 2)
    spk = eng32.ecapa(mel)
    eng32.debug(taps=True)
    wav = eng32.bigvgan(torch.from_numpy(g["latent"]), spk)
    pre = eng32.fetch_tap("bv_pre").reshape(1, -1, g["pre"].shape[1])
    up0 = eng32.fetch_tap("bv_up0").reshape(1, -1, g["up0"].shape[1])
    eng32.debug(taps=False)
    assert relerr(pre.transpose(0, 2, 1), g["pre"]) < 1e-4
    assert relerr(up0.transpose(0, 2, 1), g["up0"]) < 1e-4
    assert relerr(wav, g["wav"]) < 1e-3
    assert rms_rel(wav, g["wav"]) < 1e-4  # stated fp32 waveform RMS tolerance
    g2 = gold("micro_bigvgan_b2")
    mel2 = torch.from_numpy(g2["mel"]).transpose(1, 2)
    wav2 = eng32.bigvgan(torch.from_numpy(g2["latent"]), eng32.ecapa(mel2))
    assert rms_rel(wav2, g2["wav"]) < 1e-4


def test_bigvgan_bf16(eng16, gold):
    g = gold("micro_bigvgan")
    mel = torch.from_numpy(g["mel"]).transpose(1, 2)
    wav = eng16.bigvgan(torch.from_numpy(g["latent"]), eng16.ecapa(mel))
    r = rms_rel(wav, g["wav"])
    print(f"bf16 vocoder waveform RMS error / RMS signal = {r:.4f}")
    assert r < 6e-2


def test_dvae(eng32, eng16, gold):
    g = gold("micro_dvae")
    assert relerr(eng32.dvae_decode(g["codes"]), g["mel"]) < 1e-4
    assert rms_rel(eng16.dvae_decode(g["codes"]), g["mel"]) < 3e-2


def test_error_paths(eng32):
    with pytest.raises(RuntimeError):
        eng32.prefill(torch.zeros(1, 32, CFG.gpt.model_dim), np.full((1, 5), 10 ** 6, dtype=np.int32), 8)
    with pytest.raises(RuntimeError):
        eng32.dvae_decode(np.full((1, 4), 10 ** 6, dtype=np.int32))


def test_decode_batched_mfma_path(eng16, gold):
    """Decode batches > 4 run LayerNorm as a row kernel and the projections on MFMA (weights streamed once).  The same
    sentence replicated 12x must give 12 identical rows (batch-invariant kernels) and per-step logits within the bf16
    tolerance of the fp32 reference trace for as long as the greedy ids agree."""
    c, g = gold("micro_conditioning"), gold("micro_decode_b1")
    cond = torch.from_numpy(c["cond"])
    text = np.repeat(g["text"], 12, 0)
    eng16.prefill(cond, text, 24)
    n = g["logits"].shape[1]
    worst = 0.0
    for k in range(n):
        codes, lg = eng16.fetch(logits=True)
        assert all(np.array_equal(lg[r], lg[0]) for r in range(1, 12))
        worst = max(worst, rms_rel(lg[:1], g["logits"][:, k]))
        if codes[0, k] != g["codes"][0, k]:
            break
        if k + 1 < n:
            eng16.decode(1)
    eng16._exit()
    print(f"bf16 batched decode: {k + 1} steps compared, worst logits rel-RMS {worst:.4f}")
    assert k >= 3 and worst < 5e-2


@pytest.mark.parametrize("name,max_gen", [("micro_decode_b1", 24), ("micro_decode_b5", 24)])
@pytest.mark.parametrize("cfgs", [(30, 0.8, 1.0), (8, 0.6, 0.7)])
def test_sampling_ids_match_oracle_fp32(eng32, gold, name, max_gen, cfgs):
    """do_sample=True (HF GenerationMixin.sample, num_beams=1: repetition penalty -> temperature -> top-k -> top-p ->
    draw): with the same uniforms the HIP sampler and the oracle emit identical ids (fp32 engine, micro config)."""
    top_k, top_p, temp = cfgs
    c, g = gold("micro_conditioning"), gold(name)
    cond = torch.from_numpy(c["cond"])
    B = g["text"].shape[0]
    u = np.random.default_rng(11).random((max_gen, B), dtype=np.float32)
    got = eng32.generate(cond, g["text"], max_gen, do_sample=True, top_k=top_k, top_p=top_p, temperature=temp, uniforms=u)
    w = ogpt.to_torch(synth.gpt_state_dict(CFG, 1234))
    ref = ogpt.greedy_generate(cond, torch.from_numpy(g["text"]), w, CFG.gpt, max_gen,
                               sampling=dict(top_k=top_k, top_p=top_p, temperature=temp, uniforms=u)).numpy()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(got, ref)
    assert not np.array_equal(got, g["codes"][:, : got.shape[1]]) or top_k == 1  # it really sampled
    # greedy still works afterwards (sampling state is reset)
    again = eng32.generate(cond, g["text"], max_gen)
    assert np.array_equal(again, g["codes"])


def test_fp8_decode_weights_equal_their_dequantisation(gold):
    """BASELINE config 5 storage: GPT projections as fp8 e4m3 + power-of-two row scales.  The decode GEMV reading the
    fp8 bytes must reproduce, bit for bit, the engine that reads the bf16 dequantisation of the same quantised model
    (same products, power-of-two scaling commutes with fp32 rounding) - logits and greedy ids over a whole generation."""
    c, g = gold("micro_conditioning"), gold("micro_decode_b5")
    cond = torch.from_numpy(c["cond"])
    e8 = ieng.build_engine(CFG, "bf16", parts=("gpt",), gpt_fp8="fp8")
    ed = ieng.build_engine(CFG, "bf16", parts=("gpt",), gpt_fp8="dequant")
    text = g["text"][:4]
    out = []
    for e in (e8, ed):
        e.prefill(cond, text, 24, 10.0, True)
        trace = []
        for k in range(24):
            trace.append(e.fetch(logits=True))
            if k < 23:
                e.decode(1)
        e._exit()
        out.append(trace)
    for k in range(24):
        assert np.array_equal(out[0][k][1], out[1][k][1]), k
        assert np.array_equal(out[0][k][0][:, :k + 1], out[1][k][0][:, :k + 1]), k  # ids emitted so far (the rest of the buffer is unwritten)
    # quantisation moves the model: report how far the first-step logits are from the unquantised bf16 engine
    e0 = ieng.build_engine(CFG, "bf16", parts=("gpt",))
    e0.prefill(cond, text, 24, 10.0, True)
    _, lg0 = e0.fetch(logits=True)
    e0._exit()
    print(f"fp8-e4m3 weights vs bf16 weights, first-step logits rel-RMS {rms_rel(out[0][0][1], lg0):.4f}")


# ---- beam-sample: the reference's DEFAULT generate() mode (num_beams = 3) --------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c", "typical", "search3", "search5_lp", "sample5_lp"])
def test_beam_sample_ids_match_reference_fixture_fp32(eng32, gold, tag):
    """Engine beam-sample (device sampler + BeamSearchScorer + cache ancestry + host finalize) against the fixture the
    reference's GPT2InferenceModel / _reorder_cache produced with the same uniforms (make_golden.ref_beam_sample): the
    finalized best hypotheses are bit-exact, graph replay == eager launches."""
    c, g = gold("micro_conditioning"), gold(f"micro_beam_{tag}")
    cond = torch.from_numpy(c["cond"])
    kw = dict(do_sample=bool(int(g["do_sample"])) if "do_sample" in g else True,  # False: HF beam_search (deterministic)
              num_beams=int(g["num_beams"]), top_k=int(g["top_k"]), top_p=float(g["top_p"]),
              temperature=float(g["temperature"]), uniforms=g["uniforms"],
              length_penalty=float(g["length_penalty"]) if "length_penalty" in g else 0.0,
              typical_mass=float(g["typical_mass"]) if "typical_mass" in g else 0.0)  # the reference's TypicalLogitsWarper
    codes = eng32.generate(cond, g["text"], int(g["max_gen"]), **kw)
    assert codes.shape == g["codes"].shape, (codes.shape, g["codes"].shape)
    assert np.array_equal(codes, g["codes"]), (codes, g["codes"])
    eng32.debug(no_graph=True)
    eager = eng32.generate(cond, g["text"], int(g["max_gen"]), **kw)
    eng32.debug()
    assert np.array_equal(eager, codes)
    # a following greedy run on the same engine is unaffected (the decode graph is re-captured per mode)
    g1 = gold("micro_decode_b1")
    assert np.array_equal(eng32.generate(cond, g1["text"], 24), g1["codes"])


def test_beam_sample_matches_oracle_more_rows_fp32(eng32, gold):
    """4 batch items x 3 beams = 12 rows (the batched decode path) and 2 beams on one item, against the oracle."""
    c = gold("micro_conditioning")
    cond = torch.from_numpy(c["cond"])
    wg = ogpt.to_torch(synth.gpt_state_dict(CFG, 1234))
    text = np.stack([synth.text_ids(9, 300 + i, CFG.gpt.number_text_tokens) for i in range(4)]).astype(np.int32)
    for nb, txt, n in ((3, text, 20), (2, text[:1], 16), (4, text[:2], 12)):
        u = np.random.default_rng(nb).random((n, txt.shape[0], 2 * nb), dtype=np.float32)
        got = eng32.generate(cond, txt, n, do_sample=True, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0, uniforms=u)
        with torch.no_grad():
            want = ogpt.beam_sample_generate(cond, torch.from_numpy(txt).long(), wg, CFG.gpt, n, num_beams=nb, top_k=30, top_p=0.8,
                                             temperature=1.0, uniforms=u).numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (nb, got, want)


@pytest.mark.parametrize("nb,nret,sample", [(3, 2, True), (3, 3, True), (4, 2, False)])
def test_beam_n_best_matches_oracle_fp32(eng32, gold, nb, nret, sample):
    """generate()'s num_return_sequences under beams (model.py:655,698-703 -> BeamSearchScorer num_beam_hyps_to_keep): the n
    best hypotheses of every text row, best first, against the oracle; n > num_beams is HF's error."""
    c = gold("micro_conditioning")
    cond = torch.from_numpy(c["cond"])
    wg = ogpt.to_torch(synth.gpt_state_dict(CFG, 1234))
    txt = np.stack([synth.text_ids(9, 400 + i, CFG.gpt.number_text_tokens) for i in range(2)]).astype(np.int32)
    n = 18
    u = np.random.default_rng(nb + nret).random((n, 2, 2 * nb), dtype=np.float32)
    got = eng32.generate(cond, txt, n, do_sample=sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0, uniforms=u,
                         num_return_sequences=nret)
    with torch.no_grad():
        want = ogpt.beam_sample_generate(cond, torch.from_numpy(txt).long(), wg, CFG.gpt, n, num_beams=nb, top_k=30, top_p=0.8,
                                         temperature=1.0, uniforms=u, do_sample=sample, num_return_sequences=nret).numpy()
    assert got.shape[0] == 2 * nret
    m = min(got.shape[1], want.shape[1])
    stop = CFG.gpt.stop_mel_token
    assert np.array_equal(got[:, :m], want[:, :m]) and (got[:, m:] == stop).all() and (want[:, m:] == stop).all(), (got, want)
    # the next generation is single-best again
    one = eng32.generate(cond, txt, n, do_sample=sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0, uniforms=u)
    assert one.shape[0] == 2 and np.array_equal(one[:, :m][0], got[0, :one.shape[1]][:m])
    with pytest.raises(RuntimeError, match="num_return_sequences"):
        eng32.generate(cond, txt, n, do_sample=sample, num_beams=nb, top_k=30, top_p=0.8, temperature=1.0, uniforms=u,
                       num_return_sequences=nb + 1)


@pytest.mark.parametrize("mode", ["sample", "search"])
def test_input_tokens_under_beams_match_reference_fp32(eng32, gold, mode):
    """`input_tokens` continuation with 3 beams (model.py:672-686, 698-703): given tokens forced into every beam row at
    positions 1 .. n with the beam scores, ancestry and hypotheses untouched, generated_len counted after them - ids against
    the fixtures made from the reference's forward (beam-sample with length_penalty 0.7, beam search with 1.0)."""
    c = gold("micro_conditioning")
    cond = torch.from_numpy(c["cond"])
    g = gold(f"micro_input_tokens_beam_{mode}")
    given = g["input_tokens"].astype(np.int32)
    n_in, n = given.shape[1], int(g["max_gen"])
    eng32.set_input_tokens(given)
    try:
        got = eng32.generate(cond, g["text"], n_in + n, do_sample=mode == "sample", num_beams=3, top_k=30, top_p=0.8, temperature=1.0,
                             uniforms=g["uniforms"] if mode == "sample" else None, length_penalty=float(g["length_penalty"]))
    finally:
        eng32.set_input_tokens(None)
    assert np.array_equal(got[:, :n_in], np.repeat(given, got.shape[0] // given.shape[0], 0))
    m = min(got.shape[1] - n_in, g["codes"].shape[1])
    stop = CFG.gpt.stop_mel_token
    assert np.array_equal(got[:, n_in:n_in + m], g["codes"][:, :m]) and (got[:, n_in + m:] == stop).all() and (g["codes"][:, m:] == stop).all(), (got, g["codes"])
    # the next beam generation starts from the plain prompt again
    g0 = gold("micro_beam_search3")
    assert np.array_equal(eng32.generate(cond, g0["text"], int(g0["max_gen"]), do_sample=False, num_beams=3)[:, :g0["codes"].shape[1]], g0["codes"])


def test_beam_sample_bf16_runs_and_is_deterministic(eng16, gold):
    c, g = gold("micro_conditioning"), gold("micro_beam_a")
    cond = torch.from_numpy(c["cond"])
    kw = dict(do_sample=True, num_beams=3, top_k=30, top_p=0.8, temperature=1.0, uniforms=g["uniforms"])
    a = eng16.generate(cond, g["text"], int(g["max_gen"]), **kw)
    b = eng16.generate(cond, g["text"], int(g["max_gen"]), **kw)
    assert np.array_equal(a, b) and a.shape[0] == g["text"].shape[0] and a.shape[1] <= int(g["max_gen"])


def test_typical_filter_in_greedy_and_beam_search_matches_reference_fp32(eng32, gold):
    """typical_sampling=True with do_sample=False (a logits PROCESSOR in the reference, model.py:690-697): the device runs
    the typical pre-pass in front of the greedy arg-max / the beam-search candidates - ids against the reference fixtures."""
    c = gold("micro_conditioning")
    cond = torch.from_numpy(c["cond"])
    g = gold("micro_greedy_typical")
    got = eng32.generate(cond, g["text"], int(g["max_gen"]), typical_mass=float(g["typical_mass"]))
    m = min(got.shape[1], g["codes"].shape[1])
    assert np.array_equal(got[:, :m], g["codes"][:, :m]), (got, g["codes"])
    gb = gold("micro_beam_search3_typical")
    gotb = eng32.generate(cond, gb["text"], int(gb["max_gen"]), do_sample=False, num_beams=3, typical_mass=float(gb["typical_mass"]))
    m = min(gotb.shape[1], gb["codes"].shape[1])
    assert np.array_equal(gotb[:, :m], gb["codes"][:, :m]), (gotb, gb["codes"])
    # and the next plain greedy generation is unfiltered again
    g1 = gold("micro_decode_b1")
    assert np.array_equal(eng32.generate(cond, g1["text"], 24), g1["codes"])


def test_typical_sampling_single_beam_matches_oracle_fp32(eng32, gold):
    """typical_sampling=True with num_beams=1 (GenerationMixin.sample: RepetitionPenalty -> Typical(min keep 1) ->
    Temperature -> TopK -> TopP), micro and through a long vocabulary sort at IndexTTS-1.5 size in test_gpu_fullsize."""
    c, g = gold("micro_conditioning"), gold("micro_decode_b5")
    cond = torch.from_numpy(c["cond"])
    wg = ogpt.to_torch(synth.gpt_state_dict(CFG, 1234))
    n, B = 20, g["text"].shape[0]
    u = np.random.default_rng(9).random((n, B), dtype=np.float32)
    got = eng32.generate(cond, g["text"], n, do_sample=True, top_k=30, top_p=0.8, temperature=0.9, uniforms=u, typical_mass=0.7)
    with torch.no_grad():
        want = ogpt.greedy_generate(cond, torch.from_numpy(g["text"]), wg, CFG.gpt, n,
                                    sampling=dict(top_k=30, top_p=0.8, temperature=0.9, uniforms=u, typical_mass=0.7)).numpy()
    assert got.shape == want.shape and np.array_equal(got, want), (got, want)
    plain = eng32.generate(cond, g["text"], n, do_sample=True, top_k=30, top_p=0.8, temperature=0.9, uniforms=u)
    assert not np.array_equal(plain, got)  # the filter changes the distribution


def test_sampling_with_topk_off_runs_on_the_host_exactly(gold):
    """HF: top_k = 0 / None switches the TopK warper off (webui.py:393-402 offers 0, infer.py:116-124 forwards it): the token
    choice then runs on the host over the whole vocabulary (Engine._generate_host_sampled -> itts_gpt_commit).  Micro config,
    fp32: ids equal to the oracle's sample() with top_k = V (the oracle's warper restatement is pinned to transformers)."""
    cfg = icfg.micro()
    eng = ieng.build_engine(cfg, "fp32", parts=("gpt",))
    g = gold("micro_decode_b1")
    mel = torch.from_numpy(gold("micro_conditioning")["mel"])
    cond = eng.conditioning(mel)
    text = np.concatenate([g["text"], gold("micro_decode_b1_alt")["text"]], 0).astype(np.int32)
    V = cfg.gpt.number_mel_codes
    u = np.random.default_rng(17).random((20, 2), dtype=np.float32)
    got = eng.generate(cond, text, 20, do_sample=True, top_k=0, top_p=0.8, temperature=0.9, uniforms=u)
    wg = ogpt.to_torch(synth.gpt_state_dict(cfg, 1234))
    ocond = ogpt.get_conditioning(mel, wg, cfg.gpt)
    want = ogpt.greedy_generate(ocond, torch.from_numpy(text), wg, cfg.gpt, 20,
                                sampling={"top_k": V, "top_p": 0.8, "temperature": 0.9, "uniforms": u}).numpy()
    assert np.array_equal(got, want[:, : got.shape[1]]) and got.shape[1] >= 1
    # and the device sampler still serves top_k <= 128
    got2 = eng.generate(cond, text, 12, do_sample=True, top_k=30, top_p=0.8, temperature=0.9, uniforms=u)
    want2 = ogpt.greedy_generate(ocond, torch.from_numpy(text), wg, cfg.gpt, 12,
                                 sampling={"top_k": 30, "top_p": 0.8, "temperature": 0.9, "uniforms": u}).numpy()
    assert np.array_equal(got2, want2[:, : got2.shape[1]])
