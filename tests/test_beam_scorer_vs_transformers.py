"""CPU: the beam bookkeeping restated in oracle/hf_beam.py (BeamSearchScorer.process / finalize, BeamHypotheses.add / is_done of
transformers 4.36.2, which the reference pins and which is not installable here) against an INDEPENDENT implementation: the beam
search of the installed transformers (5.x `GenerationMixin._beam_search`, a vectorised rewrite that ships no BeamSearchScorer) on a
tiny random GPT-2.  Same model, same prompts, same processors: the 4.36.2 loop of `beam_search` driven through the restated scorer
must return the sequences (n best per item, eos placement, padding) and the sequence scores `generate()` returns.

Where the versions differ - and what is therefore left out: at `max_length` 4.36.2's `finalize` lets the RUNNING beams of an item that
is not done compete with its finished hypotheses; 5.x treats that corner differently (black-box: sometimes the running beam is taken,
sometimes not).  A case is compared only when the running beams do not matter for the restated `finalize` (its result is the same with
and without them for every item that owns a finished hypothesis); everything before `finalize` - `process`, `BeamHypotheses.add`,
`is_done`, the eos / pad handling - and the selection, ordering and padding of `finalize` are exercised by every compared case."""
import copy

import numpy as np
import pytest
import torch

from oracle import hf_beam

transformers = pytest.importorskip("transformers")

V, B, P, NEW, NB = 40, 2, 4, 14, 3
PAD = V - 1


def _model(seed, eos):
    from transformers import GPT2Config, GPT2LMHeadModel

    torch.manual_seed(seed)
    cfg = GPT2Config(vocab_size=V, n_positions=64, n_embd=32, n_layer=2, n_head=2, bos_token_id=0, eos_token_id=eos, pad_token_id=PAD)
    m = GPT2LMHeadModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(4.0)  # sharper distributions: beams finish at different steps, some items never finish
    return m


def _restated(m, ids, eos, lp, nret, rep):
    scorer = hf_beam.BeamSearchScorer(B, NB, length_penalty=lp, max_length=P + NEW, num_beam_hyps_to_keep=nret)
    x = ids.repeat_interleave(NB, 0).numpy()
    bsc = np.zeros((B, NB), np.float32)
    bsc[:, 1:] = -1e9
    bsc = bsc.reshape(-1)
    while True:  # generation/utils.py beam_search (4.36.2): log_softmax -> processors -> + beam scores -> top 2 * beams -> process
        with torch.no_grad():
            s = torch.log_softmax(m(torch.from_numpy(x)).logits[:, -1, :].float(), -1)
        if rep != 1.0:  # RepetitionPenaltyLogitsProcessor
            g = torch.gather(s, 1, torch.from_numpy(x))
            s = s.scatter(1, torch.from_numpy(x), torch.where(g < 0, g * rep, g / rep))
        flat = (s + torch.from_numpy(bsc)[:, None]).view(B, NB * V)
        top, idx = torch.topk(flat, 2 * NB, dim=1, largest=True, sorted=True)
        bsc, ntok, nidx = scorer.process(x, top.numpy(), (idx % V).numpy(), (idx // V).numpy(), PAD, eos, P)
        x = np.concatenate([x[nidx], ntok[:, None]], 1)
        if scorer.is_done or x.shape[1] >= P + NEW:
            break
    return scorer, x, bsc


CASES = [(seed, eos, lp, nret, rep) for seed in range(24) for eos in (3, 7) for lp, nret, rep in ((1.0, 1, 1.0), (0.0, 3, 1.0), (1.0, 2, 1.3))]


def test_restated_scorer_equals_the_installed_beam_search():
    compared = skipped = 0
    for seed, eos, lp, nret, rep in CASES:
        m = _model(seed, eos)
        ids = torch.randint(1, V - 2, (B, P), generator=torch.Generator().manual_seed(seed + 1))
        with torch.no_grad():
            out = m.generate(ids, attention_mask=torch.ones_like(ids), num_beams=NB, do_sample=False, max_new_tokens=NEW,
                             num_return_sequences=nret, length_penalty=lp, early_stopping=False, repetition_penalty=rep,
                             return_dict_in_generate=True, output_scores=True, pad_token_id=PAD, eos_token_id=eos)
        seq5 = out.sequences.numpy()
        scorer, x, bsc = _restated(m, ids, eos, lp, nret, rep)
        finished = [len(h) for h in scorer.hyps]
        nodone = copy.deepcopy(scorer)
        nodone.done = [d or f > 0 for d, f in zip(nodone.done, finished)]
        got4 = scorer.finalize(x, bsc, PAD, eos, P)
        if any(0 < f < nret for f in finished):
            skipped += 1  # fewer finished hypotheses than requested rows: only the running beams can fill them
            continue
        gotn = nodone.finalize(x, bsc, PAD, eos, P)
        if got4.shape != gotn.shape or not np.array_equal(got4, gotn):
            skipped += 1  # the running beams matter: the corner in which the versions differ
            continue
        L = max(seq5.shape[1], got4.shape[1])
        a = np.full((seq5.shape[0], L), PAD)
        a[:, : seq5.shape[1]] = seq5
        b = np.full((got4.shape[0], L), PAD)
        b[:, : got4.shape[1]] = got4
        assert np.array_equal(a, b), (seed, eos, lp, nret, rep)
        compared += 1
    assert compared + skipped == len(CASES) and compared >= (2 * len(CASES)) // 3, (compared, skipped)
