"""GPU: the drop-in surface - `anti_alias_activation_cuda.forward` replacement, `Activation1d(fused)`, and
`IndexTTS.infer` / `tts.gpt.inference_speech` on a synthetic micro checkpoint, checked against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import synth  # noqa: E402
from oracle import gpt as ogpt  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402
from oracle import vocoder as ovoc  # noqa: E402

CFG = icfg.micro()


def test_native_op_dropin(gold):
    from indextts.BigVGAN.alias_free_activation.cuda import load

    op = load.load()
    g = gold("micro_act1d_a")
    x = torch.from_numpy(g["x"]).cuda()
    y = op.forward(x, torch.from_numpy(g["filt"]).view(1, 1, 12).cuda(), torch.from_numpy(g["filt_down"]).view(1, 1, 12).cuda(),
                   torch.from_numpy(g["alpha"]).cuda(), torch.from_numpy(g["beta"]).cuda())
    assert y.shape == x.shape and y.dtype == x.dtype
    assert float((y.cpu() - torch.from_numpy(g["y"])).abs().max()) < 2e-5 * float(np.abs(g["y"]).max())
    # half and bfloat16 inputs (the reference dispatches float / half / bf16, type_shim.h:20-43; its GPU default is half):
    # fp32 arithmetic inside, output rounded once -> within 1 ulp of the fp32 result on the rounded input
    from oracle import vocoder as ovoc_

    for dt, ulp in ((torch.float16, 2.0 ** -10), (torch.bfloat16, 2.0 ** -7)):
        xh = x.to(dt)
        yh = op.forward(xh, torch.from_numpy(g["filt"]).cuda(), torch.from_numpy(g["filt_down"]).cuda(),
                        torch.from_numpy(g["alpha"]).cuda(), torch.from_numpy(g["beta"]).cuda())
        assert yh.dtype == dt and yh.shape == x.shape
        want = ovoc_.activation1d(xh.float().cpu(), torch.from_numpy(g["alpha"]), torch.from_numpy(g["beta"]))
        err = (yh.float().cpu() - want).abs()
        assert bool((err <= ulp * want.abs() + 1e-3 * ulp).all() or float(err.max()) < 1.01 * ulp * float(want.abs().max())), dt
        assert float((yh.float().cpu() - torch.from_numpy(g["y"])).abs().max()) < 8 * ulp * float(np.abs(g["y"]).max())
    with pytest.raises(RuntimeError):
        op.forward(x.double(), torch.zeros(12), torch.zeros(12), torch.zeros(8), torch.zeros(8))
    with pytest.raises(RuntimeError):
        op.forward(x.cpu(), torch.zeros(12), torch.zeros(12), torch.zeros(8), torch.zeros(8))
    with pytest.raises(RuntimeError):
        op.forward(x.transpose(1, 2), torch.zeros(12), torch.zeros(12), torch.zeros(8), torch.zeros(8))


def test_activation1d_module(gold):
    from indextts.BigVGAN.alias_free_activation.cuda.activation1d import Activation1d

    class SnakeBeta(torch.nn.Module):  # parameter holder with the reference's attribute names (activations.py:63-122)
        def __init__(self, a, b):
            super().__init__()
            self.alpha, self.beta, self.alpha_logscale = torch.nn.Parameter(a), torch.nn.Parameter(b), True

    g = gold("micro_act1d_b")
    act = Activation1d(SnakeBeta(torch.from_numpy(g["alpha"]), torch.from_numpy(g["beta"]))).cuda()
    y = act(torch.from_numpy(g["x"]).cuda())
    assert float((y.cpu() - torch.from_numpy(g["y"])).abs().max()) < 2e-5 * float(np.abs(g["y"]).max())


@pytest.fixture(scope="module")
def tts():
    from indextts.infer import IndexTTS

    sds = {"gpt": synth.gpt_state_dict(CFG, 1234), "bigvgan": synth.bigvgan_state_dict(CFG, 1234),
           "dvae": synth.dvae_state_dict(CFG, 1234)}
    return IndexTTS(cfg=CFG, model_dir="/nonexistent", is_fp16=False, state_dicts=sds)


def test_indextts_infer_matches_oracle(tts):
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    sents = [synth.text_ids(11, 11, CFG.gpt.number_text_tokens).astype(np.int32),
             synth.text_ids(7, 12, CFG.gpt.number_text_tokens).astype(np.int32)]
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # "generation stopped due to exceeding max_mel_tokens"
        sr, wav = tts.infer(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=24, do_sample=False, num_beams=1)
    assert sr == 24000 and wav.dtype == np.int16 and wav.ndim == 2 and wav.shape[1] == 1
    wg = ogpt.to_torch(synth.gpt_state_dict(CFG, 1234))
    wb = ogpt.to_torch(synth.bigvgan_state_dict(CFG, 1234))
    parts = []
    for s in sents:
        _, _, w = opipe.infer_sentence(mel, torch.from_numpy(s).view(1, -1), wg, wb, CFG, max_mel_tokens=24)
        parts.append(w)
    ref = torch.cat(parts, dim=1).numpy().T
    assert ref.shape == wav.shape, (ref.shape, wav.shape)
    err = np.sqrt(((wav.astype(np.float64) - ref) ** 2).mean()) / np.sqrt((ref.astype(np.float64) ** 2).mean())
    assert err < 2e-3, err  # int16 quantisation + fp32 tolerance


def test_indextts_default_kwargs_sample_reproducibly(tts):
    """infer()'s defaults are do_sample=True, num_beams=3, top_k=30, top_p=0.8 (infer.py:116-124): beam-sample with 3
    beams runs on the device and torch.manual_seed fixes the draws."""
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    sents = [synth.text_ids(11, 11, CFG.gpt.number_text_tokens).astype(np.int32)]
    outs = []
    for seed in (3, 3, 4):
        torch.manual_seed(seed)
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)  # max_mel_tokens reached
            _, wav = tts.infer(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=24)
        outs.append(wav)
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])
    assert outs[0].shape != outs[2].shape or not np.array_equal(outs[0], outs[2])
    # UnifiedVoice.inference_speech with HF kwargs
    torch.manual_seed(5)
    a = tts.gpt.inference_speech(mel, torch.from_numpy(sents[0])[None], do_sample=True, top_k=30, top_p=0.8, temperature=1.0,
                                 num_beams=1, repetition_penalty=10.0, max_generate_length=16)
    torch.manual_seed(5)
    b = tts.gpt.inference_speech(mel, torch.from_numpy(sents[0])[None], do_sample=True, top_k=30, top_p=0.8, temperature=1.0,
                                 num_beams=1, repetition_penalty=10.0, max_generate_length=16)
    assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["micro_input_tokens_b1", "micro_input_tokens_b2"])
def test_input_tokens_continuation_matches_reference(tts, gold, name):
    """`inference_speech(input_tokens=...)` (model.py:672-686) against the reference's own forward: the given tokens sit at
    mel positions 1 .. n (they are part of the reference's first forward), the first generated token is fed at n + 2; the
    returned codes start after the given ones."""
    g = gold(name)
    mel = torch.from_numpy(gold("micro_conditioning")["mel"]).cuda()
    out = tts.gpt.inference_speech(mel, torch.from_numpy(g["text"]), input_tokens=torch.from_numpy(g["input_tokens"]),
                                   do_sample=False, num_beams=1, repetition_penalty=10.0, max_generate_length=16)
    assert np.array_equal(out.cpu().numpy(), g["codes"])


def test_input_tokens_with_beam_search_through_dropin(tts, gold):
    """`inference_speech(input_tokens=..., num_beams=3, do_sample=False)`: the continuation under beam search, returned codes
    start after the given tokens - against the reference fixture."""
    g = gold("micro_input_tokens_beam_search")
    mel = torch.from_numpy(gold("micro_conditioning")["mel"]).cuda()
    out = tts.gpt.inference_speech(mel, torch.from_numpy(g["text"]), input_tokens=torch.from_numpy(g["input_tokens"]), do_sample=False,
                                   num_beams=3, repetition_penalty=10.0, length_penalty=float(g["length_penalty"]),
                                   max_generate_length=int(g["max_gen"])).cpu().numpy()
    m = min(out.shape[1], g["codes"].shape[1])
    assert np.array_equal(out[:, :m], g["codes"][:, :m]), (out, g["codes"])


def test_input_tokens_with_num_return_sequences_matches_reference(tts, gold):
    """`inference_speech(input_tokens=[2 rows], num_return_sequences=2, num_beams=3)` (model.py:672-686): the reference repeats
    text and tokens to nrs rows before generate(), which returns nrs hypotheses for each of them - 4 sequences.  The fixture's
    row expansion was recorded from the reference's own inference_speech (stubbed generate), the ids come from the hand-rolled
    beam search over exactly those inputs."""
    g = gold("micro_input_tokens_nrs")
    mel = torch.from_numpy(gold("micro_conditioning")["mel"]).cuda()
    out = tts.gpt.inference_speech(mel, torch.from_numpy(g["text"]), input_tokens=torch.from_numpy(g["input_tokens"]),
                                   num_return_sequences=int(g["num_return_sequences"]), do_sample=False, num_beams=int(g["num_beams"]),
                                   repetition_penalty=10.0, length_penalty=float(g["length_penalty"]),
                                   max_generate_length=int(g["max_gen"])).cpu().numpy()
    assert out.shape[0] == 4
    m = min(out.shape[1], g["codes"].shape[1])
    assert np.array_equal(out[:, :m], g["codes"][:, :m]), (out, g["codes"])
    stop = CFG.gpt.stop_mel_token
    assert (out[:, m:] == stop).all() and (g["codes"][:, m:] == stop).all()
    # one beam, sampling: nrs pre-expansion rows (tokens row j % 2) x nrs sampled copies each = 4 rows - equal to the explicit
    # call on the expanded rows (same seed -> same uniforms), i.e. the expansion is the reference's (rows_tokens, recorded from it)
    kw = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, num_beams=1, repetition_penalty=10.0, max_generate_length=10)
    torch.manual_seed(5)
    a = tts.gpt.inference_speech(mel, torch.from_numpy(g["text"]), input_tokens=torch.from_numpy(g["input_tokens"]),
                                 num_return_sequences=2, **kw).cpu().numpy()
    rows_tok = np.repeat(g["rows_tokens"], 2, axis=0)
    torch.manual_seed(5)
    b = tts.gpt.inference_speech(mel, torch.from_numpy(np.repeat(g["text"], 4, axis=0)), input_tokens=torch.from_numpy(rows_tok), **kw).cpu().numpy()
    assert a.shape[0] == 4 and np.array_equal(a, b)
    with pytest.raises(AssertionError, match="divisible"):
        tts.gpt.inference_speech(mel, torch.from_numpy(g["text"]), input_tokens=torch.from_numpy(g["input_tokens"]), num_return_sequences=3, **kw)


def test_padding_test_through_dropin(tts, gold):
    """tests/padding_test.py flow through `tts.gpt.inference_speech` with its kwargs."""
    g1, g5 = gold("micro_decode_b1"), gold("micro_decode_b5")
    mel = torch.from_numpy(gold("micro_conditioning")["mel"]).cuda()
    kwargs = dict(cond_mel_lengths=torch.tensor([mel.shape[-1]]), do_sample=False, top_p=0.8, top_k=None, temperature=1.0,
                  num_return_sequences=1, length_penalty=0.0, num_beams=1, repetition_penalty=10.0, max_generate_length=24)
    base = tts.gpt.inference_speech(mel, torch.from_numpy(g1["text"]).cuda(), **kwargs)
    assert np.array_equal(base.cpu().numpy(), g1["codes"])
    batch = tts.gpt.inference_speech(mel, torch.from_numpy(g5["text"]).cuda(), **kwargs)
    assert np.array_equal(batch.cpu().numpy(), g5["codes"])
    codes, lens = tts.remove_long_silence(base)
    assert codes.shape[1] == int(lens[0])


def test_dropin_misc(tts, gold, tmp_path):
    from indextts.vqvae.xtts_dvae import DiscreteVAE

    g = gold("micro_dvae")
    mel, _ = DiscreteVAE(tts.engine).decode(torch.from_numpy(g["codes"]))
    assert float((mel.float().cpu() - torch.from_numpy(g["mel"])).abs().max()) < 1e-4 * float(np.abs(g["mel"]).max())
    ge = gold("micro_dvae_encode")
    for tag in "abc":  # get_codebook_indices: integer-exact against the reference (even / odd / tiny lengths)
        codes = DiscreteVAE(tts.engine).get_codebook_indices(torch.from_numpy(ge[f"mel_{tag}"]))
        assert codes.dtype == torch.int64 and np.array_equal(codes.cpu().numpy(), ge[f"codes_{tag}"]), tag
    calls = []
    tts.set_gr_progress_callback(lambda v, d: calls.append(v))
    out = tts.infer_fast(prompt_mel=torch.from_numpy(synth.prompt_mel(61, seed=7)), text=[[5, 6, 7, 8, 9]],
                         output_path=str(tmp_path / "o" / "x.wav"), do_sample=False, num_beams=1, max_mel_tokens=8)
    assert out.endswith("x.wav") and calls
    from scipy.io import wavfile

    sr, data = wavfile.read(out)
    assert sr == 24000 and data.dtype == np.int16 and data.shape[0] % 1024 == 0
    with pytest.raises(TypeError):
        tts.infer(text=[[5, 6]])


def test_infer_fast_matches_reference_fixture(tts, gold):
    """`infer_fast` against the reference's own flow (oracle/make_golden.py fast_fixtures: reference modules, 5 sentences,
    bucket size 2 -> length-sorted buckets, BigVGAN over time-concatenated chunks of 2 latents, infer.py:480-498)."""
    import warnings

    g = gold("micro_infer_fast")
    sents = [g["text"][i, : int(n)].astype(np.int32) for i, n in enumerate(g["text_lens"])]
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        sr, wav = tts.infer_fast(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=int(g["max_mel_tokens"]),
                                 sentences_bucket_max_size=int(g["bucket_size"]), do_sample=False, num_beams=1)
    ref = g["wav_int16"].T
    assert sr == 24000 and wav.shape == ref.shape, (wav.shape, ref.shape)
    err = np.sqrt(((wav.astype(np.float64) - ref) ** 2).mean()) / np.sqrt((ref.astype(np.float64) ** 2).mean())
    assert err < 2e-3, err
    # plain `infer` vocodes per sentence: same codes, different samples around every chunk seam
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        _, wav2 = tts.infer(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=int(g["max_mel_tokens"]),
                            do_sample=False, num_beams=1)
    assert wav2.shape == wav.shape and not np.array_equal(wav2, wav)


def test_prompt_cache_is_not_aliased_and_many_sentences(tts):
    """Two DIFFERENT same-shape prompts back to back must not share conditioning latents (the r01 cache was keyed on the
    tensor address, which the caching allocator recycles); more sentences than the engine's max_batch are decoded in
    groups; an empty text raises like the reference (torch.cat of nothing -> RuntimeError)."""
    import warnings

    sents = [synth.text_ids(6, 40, CFG.gpt.number_text_tokens).astype(np.int32)]
    outs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for seed in (7, 8, 7):
            mel = torch.from_numpy(synth.prompt_mel(61, seed=seed))  # fresh host tensor -> fresh device copy inside infer
            outs.append(tts.infer(prompt_mel=mel, text=sents, output_path=None, max_mel_tokens=6, do_sample=False, num_beams=1)[1])
            del mel
        assert np.array_equal(outs[0], outs[2]) and not np.array_equal(outs[0], outs[1])
        many = [synth.text_ids(5, 100 + i, CFG.gpt.number_text_tokens).astype(np.int32) for i in range(70)]
        _, wav = tts.infer(prompt_mel=torch.from_numpy(synth.prompt_mel(61, seed=7)), text=many, output_path=None,
                           max_mel_tokens=3, do_sample=False, num_beams=1)
        assert wav.shape[0] > 0 and wav.shape[0] % 1024 == 0
    with pytest.raises(RuntimeError):
        tts.infer(prompt_mel=torch.from_numpy(synth.prompt_mel(61, seed=7)), text=[], output_path=None)


def test_engine_is_serialised_across_threads(tts):
    """The reference web UI starts a worker thread per request into ONE IndexTTS (webui.py:441-452): concurrent calls
    must serialise on the engine lock and return what the same calls return one after the other."""
    import threading
    import warnings

    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    texts = [[synth.text_ids(5 + i, 60 + i, CFG.gpt.number_text_tokens).astype(np.int32)] for i in range(4)]

    def run(i, out):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            out[i] = tts.infer(prompt_mel=mel, text=texts[i], output_path=None, max_mel_tokens=8, do_sample=False, num_beams=1)[1]

    serial = {}
    for i in range(4):
        run(i, serial)
    par = {}
    ths = [threading.Thread(target=run, args=(i, par)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(4):
        assert np.array_equal(serial[i], par[i]), i


def test_infer_batch_equals_per_utterance_infer(tts):
    import warnings

    mel = torch.from_numpy(synth.prompt_mel(61, seed=7)).cuda()
    utts = [[synth.text_ids(5 + (i + k) % 4, 200 + 10 * i + k, CFG.gpt.number_text_tokens).astype(np.int32) for k in range(1 + i % 3)]
            for i in range(5)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        batch = tts.infer_batch(mel, utts, max_mel_tokens=6, do_sample=False, num_beams=1)
        for i, u in enumerate(utts):
            sr, w = tts.infer(prompt_mel=mel, text=u, output_path=None, max_mel_tokens=6, do_sample=False, num_beams=1)
            assert batch[i][0] == sr and np.array_equal(batch[i][1], w), i


def test_indextts_from_files(tts, tmp_path):
    """The reference's construction path end to end (infer.py:42-76, utils/checkpoint.py:25-34): `config.yaml`,
    `gpt.pth` = {"model": state_dict}, `bigvgan_generator.pth` = {"generator": state_dict with weight_g / weight_v pairs,
    folded after load}, `bpe.model` - and `infer(text=<str>)` through the normaliser + SentencePiece front end."""
    spm = pytest.importorskip("sentencepiece")
    from indextts.infer import IndexTTS

    gpt_sd = synth.gpt_state_dict(CFG, 1234)
    bv_sd = synth.bigvgan_state_dict(CFG, 1234)
    # un-fold weight norm the way torch.nn.utils.weight_norm stores it (dim 0): weight = g * v / ||v||
    raw = {}
    folded = 0
    for k, v in bv_sd.items():
        is_wn = k.endswith(".weight") and v.ndim == 3 and not k.startswith(("cond_layer", "conds.", "speaker_encoder"))
        if is_wn:
            scale = 0.5 + (np.arange(v.shape[0], dtype=np.float32) % 7)[:, None, None] / 4.0
            vv = (v * scale).astype(np.float32)
            raw[k[:-len("weight")] + "weight_v"] = torch.from_numpy(vv)
            raw[k[:-len("weight")] + "weight_g"] = torch.from_numpy(
                np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True)).astype(np.float32))
            folded += 1
        else:
            raw[k] = torch.from_numpy(np.asarray(v))
    assert folded > 100
    torch.save({"model": {k: torch.from_numpy(np.asarray(v)) for k, v in gpt_sd.items()}}, tmp_path / "gpt.pth")
    torch.save({"generator": raw}, tmp_path / "bigvgan_generator.pth")
    corpus = tmp_path / "corpus.txt"
    corpus.write_text("\n".join(["HELLO WORLD THIS IS A TEST .", "你 好 世 界 , 今 天 天 气 很 好 .", "GOOD MORNING , HOW ARE YOU ?"] * 20))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "bpe"), vocab_size=60, model_type="bpe",
                                   character_coverage=1.0, bos_id=0, eos_id=1, unk_id=2, pad_id=-1, hard_vocab_limit=False)
    cfg = icfg.micro()
    cfg["gpt_checkpoint"], cfg["bigvgan_checkpoint"] = "gpt.pth", "bigvgan_generator.pth"
    cfg["dataset"] = {"bpe_model": "bpe.model"}
    icfg.dump_yaml(cfg, str(tmp_path / "config.yaml"))
    with pytest.warns(RuntimeWarning):  # WeTextProcessing is not installed here: the documented degraded mode
        ftts = IndexTTS(str(tmp_path / "config.yaml"), str(tmp_path), is_fp16=False)
    assert ftts.normalizer is not None and ftts.tokenizer.normalizer is ftts.normalizer
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    sents = [synth.text_ids(11, 11, CFG.gpt.number_text_tokens).astype(np.int32)]
    kw = dict(do_sample=False, num_beams=1, max_mel_tokens=24)
    sr, a = ftts.infer(mel, sents, None, **kw)
    _, b = tts.infer(mel, sents, None, **kw)
    assert sr == 24000 and a.shape == b.shape
    d = (a.astype(np.float64) - b.astype(np.float64))
    assert np.sqrt((d ** 2).mean()) <= 2e-3 * np.sqrt((b.astype(np.float64) ** 2).mean()) + 1.0  # weight-norm fold: fp32 rounding only
    # a string goes through TextNormalizer (punctuation folding, "'s" expansion) and the BPE model of the checkpoint dir
    text = "Hello world； this is a test。"
    pieces = ftts.tokenizer.tokenize(text)
    assert pieces and "；" not in "".join(pieces) and "。" not in "".join(pieces)
    ids = [np.asarray(ftts.tokenizer.convert_tokens_to_ids(s), dtype=np.int32) for s in ftts.tokenizer.split_sentences(pieces, 120)]
    _, w_str = ftts.infer(mel, text, None, **kw)
    _, w_ids = ftts.infer(mel, ids, None, **kw)
    assert np.array_equal(w_str, w_ids)


def test_num_return_sequences_sampling(tts):
    """`inference_speech(num_return_sequences=n, do_sample=True)` (model.py:655,698-703): every input row expanded n times
    (repeat_interleave), the copies sampled independently; greedy with n > 1 is an error as in HF."""
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7)).cuda()
    t = torch.from_numpy(np.stack([synth.text_ids(9, 51, CFG.gpt.number_text_tokens), synth.text_ids(9, 52, CFG.gpt.number_text_tokens)]).astype(np.int32))
    torch.manual_seed(3)
    out = tts.gpt.inference_speech(mel, t, do_sample=True, top_k=30, top_p=0.9, temperature=1.0, num_beams=1, num_return_sequences=3,
                                   repetition_penalty=10.0, max_generate_length=12)
    assert out.shape[0] == 6
    a = out.cpu().numpy()
    assert not (np.array_equal(a[0], a[1]) and np.array_equal(a[1], a[2]))  # independent draws
    torch.manual_seed(3)
    one = tts.gpt.inference_speech(mel, t.repeat_interleave(3, 0), do_sample=True, top_k=30, top_p=0.9, temperature=1.0, num_beams=1,
                                   repetition_penalty=10.0, max_generate_length=12)
    assert torch.equal(out, one)  # = the expanded batch, row order item0 x 3, item1 x 3
    with pytest.raises(ValueError):
        tts.gpt.inference_speech(mel, t, do_sample=False, num_beams=1, num_return_sequences=2, max_generate_length=4)
    # beams: the n best hypotheses per row (BeamSearchScorer num_beam_hyps_to_keep), row 0 of each item = the single best
    kw = dict(do_sample=False, num_beams=3, repetition_penalty=10.0, length_penalty=0.0, max_generate_length=12)
    best = tts.gpt.inference_speech(mel, t, **kw).cpu().numpy()
    nbest = tts.gpt.inference_speech(mel, t, num_return_sequences=3, **kw).cpu().numpy()
    assert nbest.shape[0] == 6
    for b in range(2):
        m = min(best.shape[1], nbest.shape[1])
        assert np.array_equal(nbest[3 * b][:m], best[b][:m])
    with pytest.raises(ValueError, match="num_return_sequences"):
        tts.gpt.inference_speech(mel, t, num_return_sequences=4, **kw)


def test_batched_prompts_with_lengths_match_reference(tts, gold):
    """A batch of prompts of different lengths (get_conditioning with cond_mel_lengths, model.py:490-502; one set of latents
    per row, model.py:599-602): latents and greedy ids against the reference's own modules (fixture micro_cond_batch; the
    padding of the shorter prompt is noise and must not matter)."""
    g = gold("micro_cond_batch")
    mel = torch.from_numpy(g["mel"]).cuda()
    lens = torch.from_numpy(g["lens"])
    cond = tts.gpt.get_conditioning(mel, lens)
    assert tuple(cond.shape) == tuple(g["cond"].shape)
    assert float((cond.cpu() - torch.from_numpy(g["cond"])).abs().max()) < 1e-4 * float(np.abs(g["cond"]).max())
    out = tts.gpt.inference_speech(mel, torch.from_numpy(g["text"]), cond_mel_lengths=lens, do_sample=False, num_beams=1,
                                   repetition_penalty=10.0, max_generate_length=16)
    assert np.array_equal(out.cpu().numpy(), g["codes"])
