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
