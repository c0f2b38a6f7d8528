"""CPU: the oracle's restatement of HF's sampling warpers (oracle/gpt.py sample_distribution / sample_pick) pinned
against the transformers implementation installed here (the reference's third-party dependency for this path:
transformers==4.36.2 GenerationMixin.sample, call site indextts/gpt/model.py:690-703 with infer.py:116-124 kwargs)."""
import numpy as np
import pytest
import torch

from oracle import gpt as ogpt

tlp = pytest.importorskip("transformers.generation.logits_process")


def hf_probs(scores, top_k, top_p, temperature):
    s = torch.from_numpy(scores)[None].clone()
    ids = torch.zeros(1, 1, dtype=torch.long)
    if temperature != 1.0:
        s = tlp.TemperatureLogitsWarper(temperature)(ids, s)
    s = tlp.TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(ids, s)
    if top_p < 1.0:
        s = tlp.TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(ids, s)
    return torch.softmax(s, -1)[0].numpy()


@pytest.mark.parametrize("V", [66, 8194])
@pytest.mark.parametrize("cfg", [(30, 0.8, 1.0), (30, 0.8, 0.7), (5, 0.5, 1.3), (1, 0.8, 1.0), (50, 1.0, 1.0), (64, 0.95, 0.9)])
def test_sample_distribution_matches_hf_warpers(V, cfg):
    top_k, top_p, temp = cfg
    rng = np.random.default_rng(V * 1000 + top_k)
    for trial in range(8):
        scores = (rng.standard_normal(V) * (1.0 + trial)).astype(np.float32)
        if trial == 3:
            scores[7] = -np.inf  # suppressed stop token
        idx, e = ogpt.sample_distribution(scores, top_k, top_p, temp)
        p = hf_probs(scores, top_k, top_p, temp)
        support = np.nonzero(p > 0)[0]
        assert set(idx.tolist()) == set(support.tolist()), (trial, cfg)
        mine = e / e.sum()
        assert np.abs(mine - p[idx]).max() < 2e-6
        assert np.all(np.diff(p[idx]) <= 1e-9)  # descending order


def test_sample_pick_is_inverse_cdf_and_respects_repetition_penalty():
    rng = np.random.default_rng(3)
    scores = rng.standard_normal(200).astype(np.float32) * 3
    idx, e = ogpt.sample_distribution(scores, 30, 0.8, 1.0)
    cdf = np.cumsum(e.astype(np.float64)) / e.astype(np.float64).sum()
    for u in [0.0, 1e-6, 0.2, 0.5, 0.79, 0.999999]:
        want = idx[min(int(np.searchsorted(cdf, u, side="left")), len(idx) - 1)]
        assert ogpt.sample_pick(scores, 30, 0.8, 1.0, u) == want
    counts = np.zeros(200)
    us = rng.random(4000)
    for u in us:
        counts[ogpt.sample_pick(scores, 30, 0.8, 1.0, float(u))] += 1
    emp = counts[idx] / counts.sum()
    assert np.abs(emp - e / e.sum()).max() < 0.03
    # HF RepetitionPenaltyLogitsProcessor on the oracle side (oracle.repetition_penalty_) then the warpers
    s = torch.from_numpy(scores)[None].clone()
    seen = torch.tensor([[int(idx[0]), int(idx[1])]])
    pen = ogpt.repetition_penalty_(s.clone(), seen, 10.0)
    hf = tlp.RepetitionPenaltyLogitsProcessor(10.0)(seen, s.clone())
    assert torch.equal(pen, hf)


# ---- beam-sample (the reference's default: num_beams = 3) -----------------------------------------------------------
from itts_hip import config as icfg  # noqa: E402
from itts_hip import synth  # noqa: E402
from oracle import hf_beam  # noqa: E402


@pytest.mark.parametrize("V", [66, 8194])
@pytest.mark.parametrize("cfg", [(30, 0.8, 1.0), (10, 0.6, 0.9), (1, 0.3, 1.0), (64, 0.95, 1.2)])
def test_beam_warp_row_matches_hf_warpers_min_keep_2(V, cfg):
    """hf_beam.warp_row (Temperature -> TopK -> TopP with min_tokens_to_keep = 2, as _get_logits_warper builds them for
    num_beams > 1) against the installed transformers classes on log-softmaxed, penalised rows."""
    top_k, top_p, temp = cfg
    rng = np.random.default_rng(V + top_k)
    for trial in range(6):
        lp = torch.log_softmax(torch.from_numpy((rng.standard_normal(V) * (1.0 + trial)).astype(np.float32)), -1)[None]
        ids = torch.from_numpy(rng.integers(0, V, (1, 9)))
        lp = tlp.RepetitionPenaltyLogitsProcessor(10.0)(ids, lp.clone())
        s = lp.clone()
        if temp != 1.0:
            s = tlp.TemperatureLogitsWarper(temp)(ids, s)
        s = tlp.TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=2)(ids, s)
        s = tlp.TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=2)(ids, s)
        keep = np.nonzero(np.isfinite(s[0].numpy()))[0]
        got_ids, got_sc = hf_beam.warp_row(lp[0].numpy(), top_k, top_p, temp, 2)
        assert np.array_equal(got_ids, keep), (trial, cfg)
        assert np.abs(got_sc - s[0].numpy()[keep]).max() < 1e-6


def test_draw_without_replacement_distribution():
    """The shared-uniform sequential draw samples what torch.multinomial(replacement=False) samples: first-draw
    frequencies follow the weights, all draws distinct."""
    rng = np.random.default_rng(0)
    sc = np.log(np.asarray([0.5, 0.25, 0.125, 0.0625, 0.0625], dtype=np.float32))
    first = np.zeros(5)
    for _ in range(4000):
        picks = hf_beam.draw_without_replacement(sc, rng.random(4, dtype=np.float32))
        assert len(set(picks)) == 4
        first[picks[0]] += 1
    assert np.abs(first / first.sum() - np.exp(sc)).max() < 0.03


def test_beam_hypotheses_bookkeeping():
    h = hf_beam.BeamHypotheses(2, 0.0)
    h.add(np.arange(3), -1.0, 3)
    assert not h.is_done(-0.5, 5, 2)
    h.add(np.arange(4), -3.0, 4)
    assert h.worst_score == -3.0 and h.is_done(-3.5, 5, 2) and not h.is_done(-2.0, 5, 2)
    h.add(np.arange(5), -2.0, 5)  # evicts the -3.0 hypothesis
    assert sorted(s for s, _ in h.beams) == [-2.0, -1.0] and h.worst_score == -2.0
    h.add(np.arange(6), -9.0, 6)  # worse than the worst: ignored
    assert len(h) == 2


def test_typical_filter_without_sampling_matches_reference_fixtures(gold):
    """typical_sampling=True with do_sample=False: the reference appends its TypicalLogitsWarper to `logits_processor`
    (model.py:690-697), so greedy search and beam search run it too (and may lose their arg-max to it).  Oracle against the
    fixtures made from the reference's forward + its own warper class (make_golden.typical_fixtures)."""
    cfg = icfg.micro()
    w = ogpt.to_torch(synth.gpt_state_dict(cfg, 1234))
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    g = gold("micro_greedy_typical")
    with torch.no_grad():
        cond = ogpt.get_conditioning(mel, w, cfg.gpt)
        out = ogpt.greedy_generate(cond, torch.from_numpy(g["text"]).long(), w, cfg.gpt, int(g["max_gen"]), typical_mass=float(g["typical_mass"]))
        plain = ogpt.greedy_generate(cond, torch.from_numpy(g["text"]).long(), w, cfg.gpt, int(g["max_gen"]))
    assert np.array_equal(out.numpy(), g["codes"])
    assert not np.array_equal(plain.numpy()[:, : out.shape[1]], out.numpy()[:, : plain.shape[1]])  # the filter does something
    gb = gold("micro_beam_search3_typical")
    with torch.no_grad():
        outb = ogpt.beam_sample_generate(cond, torch.from_numpy(gb["text"]).long(), w, cfg.gpt, int(gb["max_gen"]), num_beams=3,
                                         do_sample=False, typical_mass=float(gb["typical_mass"]))
    assert np.array_equal(outb.numpy(), gb["codes"]), (outb.numpy(), gb["codes"])


def test_beam_scorer_keeps_n_best_in_descending_order():
    """BeamSearchScorer.finalize with num_beam_hyps_to_keep = n (generate()'s num_return_sequences under beams): rows
    n * b .. n * b + n - 1 are batch item b's hypotheses by descending score, the later insertion first on equal scores
    (sorted() is stable and finalize pops from the end); n > num_beams is HF's ValueError."""
    sc = hf_beam.BeamSearchScorer(2, 3, length_penalty=0.0, max_length=12, num_beam_hyps_to_keep=2)
    for b, scores in enumerate(([-3.0, -1.0, -1.0], [-0.5, -4.0, -2.0])):
        for j, s in enumerate(scores):
            sc.hyps[b].add(np.full(4 + j, 10 * b + j), s, 1)
    sc.done = [True, True]
    out = sc.finalize(np.zeros((6, 3), dtype=np.int64), np.zeros(6, np.float32), pad=99, eos=98, prompt_len=3)
    assert out.shape == (4, 7)
    assert [int(r[0]) for r in out] == [2, 1, 10, 12]  # item 0: the two -1.0s, later first; item 1: -0.5 then -2.0
    assert list(out[1][:7]) == [1] * 5 + [98, 99]
    with pytest.raises(ValueError):
        hf_beam.BeamSearchScorer(1, 3, num_beam_hyps_to_keep=4)


def test_beam_generate_n_best_first_row_is_the_single_best(gold):
    """num_return_sequences = 3 of 3 beams: row 0 of every item is what num_return_sequences = 1 returns (the reference
    fixture micro_beam_a), all three rows differ."""
    cfg = icfg.micro()
    g = gold("micro_beam_a")
    w = ogpt.to_torch(synth.gpt_state_dict(cfg, 1234))
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    with torch.no_grad():
        cond = ogpt.get_conditioning(mel, w, cfg.gpt)
        out = ogpt.beam_sample_generate(cond, torch.from_numpy(g["text"]).long(), w, cfg.gpt, int(g["max_gen"]), num_beams=3,
                                        top_k=int(g["top_k"]), top_p=float(g["top_p"]), temperature=float(g["temperature"]),
                                        uniforms=g["uniforms"], num_return_sequences=3).numpy()
    B = g["text"].shape[0]
    assert out.shape[0] == 3 * B
    for b in range(B):
        n = min(out.shape[1], g["codes"].shape[1])
        assert np.array_equal(out[3 * b][:n], g["codes"][b][:n])
        assert len({tuple(r) for r in out[3 * b:3 * b + 3]}) == 3


@pytest.mark.parametrize("tag", ["a", "b", "c", "typical", "search3", "search5_lp", "sample5_lp"])
def test_beam_sample_generate_matches_reference_fixture(gold, tag):
    """oracle.gpt.beam_sample_generate (own GPT-2 stack + own warper restatement) against the fixture produced by the
    reference's GPT2InferenceModel.forward / _reorder_cache + the installed transformers warpers (make_golden.ref_beam_sample),
    same uniforms: token ids bit-exact."""
    cfg = icfg.micro()
    g = gold(f"micro_beam_{tag}")
    w = ogpt.to_torch(synth.gpt_state_dict(cfg, 1234))
    mel = torch.from_numpy(synth.prompt_mel(61, seed=7))
    with torch.no_grad():
        cond = ogpt.get_conditioning(mel, w, cfg.gpt)
        out = ogpt.beam_sample_generate(cond, torch.from_numpy(g["text"]).long(), w, cfg.gpt, int(g["max_gen"]), num_beams=int(g["num_beams"]),
                                        top_k=int(g["top_k"]), top_p=float(g["top_p"]), temperature=float(g["temperature"]),
                                        uniforms=g["uniforms"], typical_mass=float(g["typical_mass"]) if "typical_mass" in g else 0.0,
                                        length_penalty=float(g["length_penalty"]) if "length_penalty" in g else 0.0,
                                        do_sample=bool(int(g["do_sample"])) if "do_sample" in g else True)
    assert np.array_equal(out.numpy(), g["codes"]), (out.numpy(), g["codes"])


@pytest.mark.parametrize("V", [66, 8194])
@pytest.mark.parametrize("mass,min_keep", [(0.9, 1), (0.5, 2), (0.2, 1), (0.97, 2)])
def test_typical_filter_matches_reference_class(V, mass, min_keep):
    """hf_beam.typical_filter against the reference's own TypicalLogitsWarper subclass (indextts/utils/typical_sampling.py,
    imported from /root/reference when present; the installed transformers base class otherwise - same algorithm)."""
    from oracle import ref_import

    if ref_import.available():
        import importlib.util
        import os

        spec = importlib.util.spec_from_file_location("_ref_typical", os.path.join(ref_import.REF_ROOT, "indextts", "utils", "typical_sampling.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        warper = mod.TypicalLogitsWarper(mass=mass, min_tokens_to_keep=min_keep)
    else:
        warper = tlp.TypicalLogitsWarper(mass=mass, min_tokens_to_keep=min_keep)
    rng = np.random.default_rng(V)
    for trial in range(6):
        s = (rng.standard_normal(V) * (0.5 + trial)).astype(np.float32)
        if trial == 4:
            s[5] = -np.inf
        want = warper(torch.zeros(1, 1, dtype=torch.long), torch.from_numpy(s)[None].clone())[0].numpy()
        got = hf_beam.typical_filter(s, mass, min_keep)
        kg, kw = np.isfinite(got), np.isfinite(want)
        # the cut sits where an fp32 running sum over thousands of terms crosses `mass`: torch.cumsum and a sequential sum
        # may disagree by a few tail tokens there (measure: the probability mass of the disagreement)
        p = np.exp(s - s.max()) / np.exp(s - s.max()).sum()
        assert (kg != kw).sum() <= 4 and float(p[kg != kw].sum()) < 1e-5, (trial, int(kg.sum()), int(kw.sum()))
        both = kg & kw
        assert np.array_equal(got[both], want[both])


@pytest.mark.parametrize("V", [66, 8194])
@pytest.mark.parametrize("cfg", [(0, 0.8, 1.0), (0, 0.95, 0.7), (0, 1.0, 1.3), (300, 0.9, 1.0), (0, 0.3, 1.0)])
def test_host_sampler_matches_hf_processors_over_the_whole_vocabulary(V, cfg):
    """The product's HOST token choice (itts_hip.infer_core.host_distribution: generate() modes the device samplers do not
    cover - top_k = 0 / None = HF's TopK warper off, or top_k > 128) against the installed transformers processors:
    RepetitionPenalty -> Temperature -> [TopK] -> TopP, kept set and probabilities."""
    from itts_hip import infer_core

    top_k, top_p, temp = cfg
    rng = np.random.default_rng(V * 7 + top_k + int(top_p * 100))
    for trial in range(6):
        scores = (rng.standard_normal(V) * (0.5 + trial)).astype(np.float32)
        seen = set(int(x) for x in rng.integers(0, V, 5))
        stop = V - 1
        idx, e = infer_core.host_distribution(scores, seen, 10.0, temp, top_k, top_p, 0.0, stop, trial % 2 == 1)
        s = torch.from_numpy(scores)[None].clone()
        ids = torch.tensor([sorted(seen)])
        s = tlp.RepetitionPenaltyLogitsProcessor(10.0)(ids, s)
        if trial % 2 == 1:
            s[0, stop] = -float("inf")
        if temp != 1.0:
            s = tlp.TemperatureLogitsWarper(temp)(ids, s)
        if top_k >= 1:
            s = tlp.TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=1)(ids, s)
        if top_p < 1.0:
            s = tlp.TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=1)(ids, s)
        p = torch.softmax(s, -1)[0].numpy()
        support = np.nonzero(p > 0)[0]
        assert set(idx.tolist()) == set(support.tolist()), (trial, cfg, len(idx), len(support))
        assert np.abs(e / e.sum() - p[idx]).max() < 2e-6
        assert np.all(np.diff(p[idx]) <= 1e-9)
    # the draw: inverse CDF over the kept tokens in descending order
    toks = infer_core.host_sample_step(scores[None], [seen], 10.0, temp, top_k, top_p, 0.0, np.asarray([0.0], np.float32), stop, False)
    idx, e = infer_core.host_distribution(scores, seen, 10.0, temp, top_k, top_p, 0.0, stop, False)
    assert toks[0] == idx[0]


def test_host_typical_filter_matches_reference_subclass():
    from itts_hip import infer_core

    rng = np.random.default_rng(11)
    base = getattr(tlp, "TypicalLogitsWarper", None)
    if base is None:
        pytest.skip("installed transformers has no TypicalLogitsWarper")
    for V, mass in ((66, 0.9), (8194, 0.9), (8194, 0.3)):
        scores = (rng.standard_normal(V) * 2).astype(np.float32)
        got = infer_core._typical_filter(scores, mass)
        want = base(mass=mass, min_tokens_to_keep=1)(torch.zeros(1, 1, dtype=torch.long), torch.from_numpy(scores)[None].clone())[0].numpy()
        assert np.array_equal(np.isfinite(got), np.isfinite(want)), (V, mass)


@pytest.mark.parametrize("V", [66, 8194])
@pytest.mark.parametrize("cfg", [(0, 0.8, 1.0), (200, 0.9, 0.7), (0, 1.0, 1.0), (30, 0.8, 1.0)])
def test_host_beam_step_matches_hf_processors_and_the_oracle_draw(V, cfg):
    """The product's HOST beam_sample step (itts_hip.infer_core.host_beam_step: beams with `top_k = 0 / None` or > 128) against
    what oracle/make_golden.ref_beam_sample runs per step: torch.log_softmax -> the INSTALLED RepetitionPenalty / Temperature /
    TopK / TopP classes (min_tokens_to_keep = 2) -> oracle.hf_beam.beam_sample_step (flat order, sequential draws without
    replacement, stable sort by score).  Two batch items x 3 beams, one item finished (its picks are ignored by the scorer)."""
    from itts_hip import infer_core
    from oracle import hf_beam

    top_k, top_p, temp = cfg
    nb, items, k = 3, 2, 6
    rng = np.random.default_rng(V * 13 + top_k + int(top_p * 100))
    logits = (rng.standard_normal((items * nb, V)) * 1.5).astype(np.float32)
    hist = rng.integers(2, V - 1, (items * nb, 12)).astype(np.int32)
    beam_scores = (-rng.random(items * nb) * 3).astype(np.float32)
    u = rng.random((items, 2 * nb)).astype(np.float32)
    stop, start = V - 1, V - 2
    done = np.zeros(items, dtype=np.int32)
    psc, ptok, pbeam = infer_core.host_beam_step(logits, hist, k, beam_scores, done, nb, 10.0, temp, top_k, top_p, 0.0, u, stop, False, start)
    for bi in range(items):
        cands = []
        for r in range(nb):
            row = bi * nb + r
            ids = torch.tensor([[1, start] + [int(t) for t in hist[row, :k]]])
            s = torch.log_softmax(torch.from_numpy(logits[row])[None], dim=-1)
            s = tlp.RepetitionPenaltyLogitsProcessor(10.0)(ids, s.clone())
            if temp != 1.0:
                s = tlp.TemperatureLogitsWarper(temp)(ids, s)
            if top_k:
                s = tlp.TopKLogitsWarper(top_k=top_k, min_tokens_to_keep=2)(ids, s)
            if top_p < 1.0:
                s = tlp.TopPLogitsWarper(top_p=top_p, min_tokens_to_keep=2)(ids, s)
            rowv = s[0].numpy()
            keep = np.nonzero(np.isfinite(rowv))[0]
            cands.append((keep, rowv[keep]))
        ws, wt, wb = hf_beam.beam_sample_step(cands, beam_scores[bi * nb:(bi + 1) * nb], V, u[bi])
        # the product hands the picks over in DRAW order; the scorer kernel sorts them (stable, descending) as the oracle does
        order = sorted(range(2 * nb), key=lambda j: -float(psc[bi, j]))
        assert np.array_equal(ptok[bi][order], wt) and np.array_equal(pbeam[bi][order], wb), (bi, cfg)
        assert np.abs(psc[bi][order] - ws).max() < 2e-5
    done[1] = 1
    psc2, ptok2, _ = infer_core.host_beam_step(logits, hist, k, beam_scores, done, nb, 10.0, temp, top_k, top_p, 0.0, u, stop, False, start)
    assert np.array_equal(ptok2[0], ptok[0]) and (ptok2[1] == stop).all()


def test_host_typical_filter_min_keep_2_matches_reference_subclass():
    from itts_hip import infer_core
    from oracle import hf_beam

    rng = np.random.default_rng(12)
    for V, mass in ((66, 0.9), (8194, 0.2)):
        scores = (rng.standard_normal(V) * 2).astype(np.float32)
        got = infer_core._typical_filter(scores, mass, 2)
        want = hf_beam.typical_filter(scores, mass, 2)  # pinned to the reference's class by test_typical_filter_matches_reference_class
        assert np.array_equal(np.isfinite(got), np.isfinite(want)) and np.isfinite(got).sum() >= 2
