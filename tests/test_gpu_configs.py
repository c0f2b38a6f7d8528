"""GPU: the BASELINE configurations that are not the default bench line, through the C ABI.

config 3  32 utterances x 2 sentences = 64 RAGGED decode rows (L in 60..120 "mixed zh/en", T = round(4.57 L), stop times
          mixed, eos enabled): prefill -> decode -> stacked latent pass -> vocoder batched over equal-length groups; every
          row is compared with its own batch-1 run (fp32: ids equal, waveform rel-RMS <= 1e-3).
          + the batched vocoder at the bench size (64 rows x 480 frames, 31 M output samples) against batch-1 runs.
config 5  see test_config5_* (fp8 GPT weights, 20 sentences sequential and batched)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from itts_hip import config as icfg  # noqa: E402
from itts_hip import engine as ieng  # noqa: E402
from itts_hip import infer_core, prng, synth  # noqa: E402

CFG = icfg.indextts_1_5()
STOP = CFG.gpt.stop_mel_token


def rms_rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-12))


@pytest.fixture(scope="module")
def mel():
    return torch.from_numpy(synth.prompt_mel(511, seed=7))


def test_config3_ragged_64_rows_fp32(mel, accuracy):
    eng = ieng.build_engine(CFG, "fp32", parts=("gpt", "bigvgan"), max_batch=64)
    cond = eng.conditioning(mel)
    spk = eng.ecapa(mel.transpose(1, 2))
    Ls = [60, 68, 77, 85, 94, 103, 111, 120]
    rows = []  # 32 utterances x 2 sentences; 8 rows per length so equal-T groups reach the batched vocoder
    for u in range(32):
        for k in range(2):
            L = Ls[(2 * u + k) % 8]
            rows.append(synth.text_ids(L, 500 + 2 * u + k, CFG.gpt.number_text_tokens).astype(np.int32))
    Ts = [int(round(4.57 * len(r))) for r in rows]
    max_gen = max(Ts) + 1
    ids = infer_core.pad_tokens_cat(rows, CFG.gpt.stop_text_token)
    assert ids.shape == (64, 120)
    # mixed stop times with eos ENABLED: row b is made to emit the stop token at step T_b (the product's forced-token
    # table, -1 elsewhere); after that HF greedy pads the row with stop while the others run on
    forced = np.full((64, max_gen), -1, dtype=np.int32)
    for b, T in enumerate(Ts):
        forced[b, T] = STOP
    eng.set_forced(forced)
    codes = eng.generate(cond, ids, max_gen, suppress_stop=False)
    eng.set_forced(None)
    assert codes.shape == (64, max_gen)
    clean = []
    for b in range(64):
        c, n = infer_core.remove_long_silence(codes[b:b + 1], STOP)
        assert int(n[0]) == Ts[b] and (codes[b, Ts[b]:] == STOP).all(), b
        clean.append(c[0, : int(n[0])])
    lats = eng.latent_batch(cond, rows, clean)
    wavs = eng.bigvgan_grouped(lats, spk)
    # each row on its own (batch 1 everywhere)
    exact, worst_wav, worst_lat = 0, 0.0, 0.0
    for b in range(64):
        f1 = np.full((1, max_gen), -1, dtype=np.int32)
        f1[0, Ts[b]] = STOP
        eng.set_forced(f1)
        c1 = eng.generate(cond, rows[b][None], max_gen, suppress_stop=False)
        eng.set_forced(None)
        same = np.array_equal(c1[0], codes[b, : c1.shape[1]])
        exact += int(same)
        if not same:
            continue
        lat1 = eng.latent(cond, rows[b], clean[b])
        worst_lat = max(worst_lat, rms_rel(lats[b].float().cpu().numpy(), lat1.float().cpu().numpy()))
        if b % 4 == 0:  # 16 rows through the batch-1 vocoder (2 per equal-length group)
            w1 = eng.bigvgan(lat1, spk)
            worst_wav = max(worst_wav, rms_rel(wavs[b].cpu().numpy(), w1.cpu().numpy()))
    accuracy["config3_fp32_rows_ids_equal_to_batch1"] = exact
    accuracy["config3_fp32_latent_batch_vs_batch1_rel_rms"] = worst_lat
    accuracy["config3_fp32_waveform_batch_vs_batch1_rel_rms"] = worst_wav
    assert exact >= 62, exact  # fp32 summation order differs between the 64-row and the 1-row kernels: near-ties may flip
    assert worst_lat < 1e-3 and worst_wav < 1e-3, (worst_lat, worst_wav)


def test_batched_vocoder_at_bench_size_bf16(mel, accuracy, monkeypatch):
    """itts_bigvgan with B = 64 x 480 frames (what `bench.py --batch 32` launches: 31 M output samples per call) against
    batch-1 runs of the same latents: bit-identical with ITTS_GEMM_KSPLIT=0 (every kernel accumulates K in the same order at
    every batch size); with the default K split of the batch-1 conv_pre / stage-0 convolutions, within the bf16 bound."""
    eng = ieng.build_engine(CFG, "bf16", parts=("bigvgan",), max_batch=64)
    spk = eng.ecapa(mel.transpose(1, 2))
    lat = torch.from_numpy(prng.tensor("bigvgan.latent.b64", 5, (64, 480, CFG.bigvgan.gpt_dim), std=1.0, mean=0.0))
    wav = eng.bigvgan(lat, spk.expand(64, -1).contiguous())
    assert wav.shape == (64, 1, 480 * 1024) and torch.isfinite(wav).all() and float(wav.abs().max()) <= 1.0
    worst = 0.0
    for b in (0, 17, 63):
        w1 = eng.bigvgan(lat[b:b + 1], spk)
        worst = max(worst, rms_rel(wav[b].cpu().numpy(), w1[0].cpu().numpy()))
    accuracy["bf16_bigvgan_b64x480_vs_batch1_ksplit_rel_rms"] = worst
    assert worst < 4e-2, worst  # (PRNG vocoder weights: the bf16 waveform itself is 1.6e-2 from the fp32 reference, test_gpu_longrun.py)
    monkeypatch.setenv("ITTS_GEMM_KSPLIT", "0")
    worst0 = 0.0
    for b in (0, 63):
        w1 = eng.bigvgan(lat[b:b + 1], spk)
        worst0 = max(worst0, rms_rel(wav[b].cpu().numpy(), w1[0].cpu().numpy()))
    accuracy["bf16_bigvgan_b64x480_vs_batch1_rel_rms"] = worst0
    assert worst0 == 0.0, worst0


def test_config5_longform_fp8_weights_batched_and_sequential(mel, accuracy, monkeypatch):
    """(ITTS_GEMM_KSPLIT=0: the K split of few-tile GEMMs would give the one-sentence prefill a different fp32 summation order than the
    20-sentence one - tests/test_gpu_bf16_accuracy.py::test_ksplit_changes_only_the_summation_order measures that; the subject here is
    the fp8 decode kernels.)
    BASELINE config 5: a 2000-char text = 20 sentences, GPT projection weights stored as fp8 e4m3 + row scales.
    (a) all 20 sentences as ONE decode batch (MFMA path, fp8 bytes read by skinny_mfma_kernel<W8>): codes and logits are
        bit-identical to an engine that reads the bf16 DEQUANTISATION of the same weights - the fp8 bytes really are what
        is streamed, and the conversion is exact;
    (b) the same sentences one at a time ("sequential chunks, fresh KV per chunk": GEMV path, gemv_bf16_kernel<W8>) agree
        with their row of the batch within the bf16 tolerance of the two kernel families (ids until the first near-tie)."""
    monkeypatch.setenv("ITTS_GEMM_KSPLIT", "0")
    texts = np.stack([synth.text_ids(105, 900 + i, CFG.gpt.number_text_tokens) for i in range(20)]).astype(np.int32)
    n = 24
    res = {}
    cond = None
    for mode in ("fp8", "dequant"):
        eng = ieng.build_engine(CFG, "bf16", parts=("gpt",), gpt_fp8=mode, max_batch=32)
        cond = eng.conditioning(mel) if cond is None else cond
        eng.prefill(cond, texts, n, 10.0, True)
        eng.decode(n - 1)
        res[mode] = eng.fetch(logits=True)
        eng._exit()
        if mode == "fp8":
            seq = []
            for i in range(4):  # sequential chunks
                eng.prefill(cond, texts[i:i + 1], n, 10.0, True)
                first = eng.fetch(logits=True)[1].copy()
                eng.decode(n - 1)
                seq.append((eng.fetch()[0].copy(), first))
                eng._exit()
            eng.prefill(cond, texts, n, 10.0, True)
            first20 = eng.fetch(logits=True)[1].copy()
            eng._exit()
        del eng
        torch.cuda.empty_cache()
    assert np.array_equal(res["fp8"][0], res["dequant"][0]) and np.array_equal(res["fp8"][1], res["dequant"][1])
    worst, agree = 0.0, []
    for i, (codes1, lg1) in enumerate(seq):
        worst = max(worst, rms_rel(first20[i], lg1[0]))
        same = codes1 == res["fp8"][0][i, :n]
        agree.append(int(np.argmin(same)) if not same.all() else n)
    accuracy["config5_fp8_batched20_vs_sequential_first_logits_rel_rms"] = worst
    accuracy["config5_fp8_batched20_vs_sequential_ids_agree_steps"] = agree
    assert worst < 3e-2, worst


def test_rows_5_to_16_layernorm_in_projection_and_half_tiles(mel, accuracy, monkeypatch):
    """(ITTS_GEMM_KSPLIT=0, as above: (c) compares decode kernel families behind the SAME prefill arithmetic.)
    The reference's default mode decodes 3 beams per sentence, 2-4 sentences per bucket: 6-12 rows.  At 5-16 rows the
    engine folds the LayerNorms into c_attn / c_fc and runs the residual projections as half tiles (decode_mfma.hip
    LNP / HALF), on the fragment-tiled weight copies.
    (a) a row's codes and logits do not depend on how many other rows share the batch (5, 9 or 16 rows: bit-identical);
    (b) fp8 weights (config 5's format) give the bits of their bf16 dequantisation on this path too;
    (c) against the 2-row GEMV path the first-step logits agree within the bf16 tolerance of the two kernel families."""
    monkeypatch.setenv("ITTS_GEMM_KSPLIT", "0")
    texts = np.stack([synth.text_ids(105, 700 + i, CFG.gpt.number_text_tokens) for i in range(16)]).astype(np.int32)
    n = 20
    eng = ieng.build_engine(CFG, "bf16", parts=("gpt",), max_batch=16)
    eng.debug(no_engine=True)  # the launch path's MFMA kernels are the subject (5 - 6 rows default to the persistent engine)
    cond = eng.conditioning(mel)
    out = {}
    for rows in (16, 9, 5, 2):
        eng.prefill(cond, texts[:rows], n, 10.0, True)
        first = eng.fetch(logits=True)[1].copy()
        eng.decode(n - 1)
        codes, lg = eng.fetch(logits=True)
        out[rows] = (codes.copy(), lg.copy(), first)
        eng._exit()
    for rows in (9, 5):
        assert np.array_equal(out[rows][0], out[16][0][:rows]) and np.array_equal(out[rows][1], out[16][1][:rows]), rows
    worst = max(rms_rel(out[16][2][i], out[2][2][i]) for i in range(2))
    accuracy["bf16_rows16_vs_rows2_first_logits_rel_rms"] = worst
    assert worst < 3e-2, worst
    del eng
    torch.cuda.empty_cache()
    res = {}
    for mode in ("fp8", "dequant"):
        e8 = ieng.build_engine(CFG, "bf16", parts=("gpt",), gpt_fp8=mode, max_batch=16)
        e8.prefill(cond, texts[:8], n, 10.0, True)
        e8.decode(n - 1)
        res[mode] = e8.fetch(logits=True)
        e8._exit()
        del e8
        torch.cuda.empty_cache()
    assert np.array_equal(res["fp8"][0], res["dequant"][0]) and np.array_equal(res["fp8"][1], res["dequant"][1])
