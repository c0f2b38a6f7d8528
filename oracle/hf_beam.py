"""ORACLE (test infrastructure, not product): restatement of the beam bookkeeping of `transformers==4.36.2`
(`generation/beam_search.py`: BeamHypotheses, BeamSearchScorer.process / finalize) and of the per-step score pipeline of
`GenerationMixin.beam_sample` (`generation/utils.py`), which is what `UnifiedVoice.inference_speech` runs for the
reference's DEFAULT kwargs (do_sample=True, num_beams=3, top_k=30, top_p=0.8, length_penalty=0.0:
/root/reference/indextts/infer.py:116-124, indextts/gpt/model.py:690-703).

transformers 4.36.2 is a pinned, un-vendored dependency (/root/reference/setup.py:49) and the installed 5.x no longer
ships `beam_sample` / `BeamSearchScorer`, so the published algorithm is restated here; the logits warpers it calls are
pinned against the installed `TopKLogitsWarper` / `TopPLogitsWarper` / `TemperatureLogitsWarper` classes
(min_tokens_to_keep = 2 under beams) and the reference's own `TypicalLogitsWarper` in tests/test_sampling_oracle.py.

Order of one step (4.36.2 beam_sample): log_softmax(logits) -> logits_processor (RepetitionPenalty [-> Typical]) ->
logits_warper (Temperature -> TopK -> TopP) -> + beam_scores -> view(batch, beams * V) -> softmax -> multinomial(2 * beams,
without replacement) -> gather -> sort descending -> BeamSearchScorer.process -> reorder.
`torch.multinomial` cannot be reproduced on a device, so the draw is DEFINED from caller uniforms: draw j picks, by inverse
CDF of u_j over the not-yet-drawn candidates in flat index order (beam-major, token ascending), i.e. sequential sampling
without replacement - the same distribution torch.multinomial(replacement=False) samples from."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

f32 = np.float32


class BeamHypotheses:
    """beam_search.py BeamHypotheses (4.36.2): n-best list of finished hypotheses of one batch item."""

    def __init__(self, num_beams: int, length_penalty: float, early_stopping=False):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams: List[Tuple[float, np.ndarray]] = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: np.ndarray, sum_logprobs: float, generated_len: int):
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                srt = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[srt[0][1]]
                self.worst_score = srt[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int, decoder_prompt_len: int) -> bool:
        if len(self) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        highest = best_sum_logprobs / (cur_len - decoder_prompt_len) ** self.length_penalty  # early_stopping False
        return self.worst_score >= highest


class BeamSearchScorer:
    """beam_search.py BeamSearchScorer (4.36.2), one beam group; num_beam_hyps_to_keep = generate()'s num_return_sequences."""

    def __init__(self, batch_size: int, num_beams: int, length_penalty: float = 1.0, max_length: Optional[int] = None,
                 num_beam_hyps_to_keep: int = 1):
        if not 1 <= num_beam_hyps_to_keep <= num_beams:
            raise ValueError("`num_return_sequences` has to be smaller or equal to `num_beams`.")
        self.num_beams, self.max_length, self.keep = num_beams, max_length, num_beam_hyps_to_keep
        self.hyps = [BeamHypotheses(num_beams, length_penalty) for _ in range(batch_size)]
        self.done = [False] * batch_size

    @property
    def is_done(self):
        return all(self.done)

    def process(self, input_ids: np.ndarray, next_scores, next_tokens, next_indices, pad: int, eos: int, prompt_len: int):
        """input_ids [batch*beams, cur]; next_* [batch, 2*beams] sorted by descending score.
        -> (next_beam_scores, next_beam_tokens, next_beam_indices) each [batch*beams]."""
        nb = self.num_beams
        cur_len = input_ids.shape[-1] + 1
        B = len(self.hyps)
        bs = np.zeros((B, nb), dtype=np.float32)
        bt = np.zeros((B, nb), dtype=np.int64)
        bi = np.zeros((B, nb), dtype=np.int64)
        for b in range(B):
            if self.done[b]:
                bs[b], bt[b], bi[b] = 0, pad, 0
                continue
            k = 0
            for rank in range(next_tokens.shape[1]):
                tok, sc, idx = int(next_tokens[b, rank]), next_scores[b, rank], int(next_indices[b, rank])
                row = b * nb + idx
                if tok == eos:
                    if rank >= nb:
                        continue
                    self.hyps[b].add(input_ids[row].copy(), float(sc), cur_len - prompt_len)
                else:
                    bs[b, k], bt[b, k], bi[b, k] = sc, tok, row
                    k += 1
                if k == nb:
                    break
            if k < nb:
                raise ValueError("fewer than num_beams non-eos candidates")
            self.done[b] = self.done[b] or self.hyps[b].is_done(float(next_scores[b].max()), cur_len, prompt_len)
        return bs.reshape(-1), bt.reshape(-1), bi.reshape(-1)

    def finalize(self, input_ids: np.ndarray, final_scores, pad: int, eos: int, prompt_len: int) -> np.ndarray:
        nb = self.num_beams
        for b, h in enumerate(self.hyps):
            if self.done[b]:
                continue
            for k in range(nb):
                row = b * nb + k
                h.add(input_ids[row].copy(), float(final_scores[row]), input_ids.shape[-1] - prompt_len)
        best = []
        for h in self.hyps:
            srt = sorted(h.beams, key=lambda x: x[0])  # stable: the later of two equal scores is popped
            for _ in range(self.keep):  # rows keep * b .. keep * b + keep - 1: descending score
                best.append(srt.pop()[1])
        lens = np.asarray([len(x) for x in best])
        smax = int(lens.max()) + 1
        if self.max_length is not None:
            smax = min(smax, self.max_length)
        out = np.full((len(best), smax), pad, dtype=np.int64)
        for i, hyp in enumerate(best):
            out[i, : lens[i]] = hyp
            if lens[i] < smax:
                out[i, lens[i]] = eos
        return out


def warp_row(lp: np.ndarray, top_k: int, top_p: float, temperature: float, min_keep: int = 2):
    """One row of processed log-probs through Temperature -> TopK(min_tokens_to_keep) -> TopP(min_tokens_to_keep), in the
    arithmetic order the HIP beam sampler uses.  Returns (token ids ascending, scores) of the kept tokens."""
    s = np.asarray(lp, dtype=np.float32).copy()
    V = s.shape[0]
    if temperature != 1.0:
        s = (s / f32(temperature)).astype(np.float32)
    k = min(max(int(top_k), min_keep), V) if top_k else V
    kth = np.partition(s, V - k)[V - k]
    idx = np.nonzero(s >= kth)[0]
    order = np.lexsort((idx, -s[idx].astype(np.float64)))  # descending score, ascending index on ties
    idx = idx[order][:128]
    v = s[idx]
    R = len(idx)
    if top_p is not None and top_p < 1.0:
        e = np.exp((v - v[0]).astype(np.float32)).astype(np.float32)
        Z = f32(0)
        for x in e:
            Z = f32(Z + x)
        tail, R = f32(0), 1
        lim = f32(1.0) - f32(top_p)
        for r in range(len(idx) - 1, 0, -1):
            tail = f32(tail + f32(e[r] / Z))
            if not tail <= lim:
                R = r + 1
                break
        R = min(max(R, min_keep), len(idx))
    keep = np.sort(idx[:R])
    return keep, s[keep]


def draw_without_replacement(scores_flat: np.ndarray, uniforms: np.ndarray) -> List[int]:
    """Sequential inverse-CDF draws without replacement over candidates given in flat index order (their scores are
    log-weights); returns the drawn candidate positions in draw order."""
    m = scores_flat.max()
    e = np.exp((scores_flat - m).astype(np.float32)).astype(np.float32)
    alive = np.ones(len(e), dtype=bool)
    picks = []
    for u in uniforms:
        total = f32(0)
        for i in range(len(e)):
            if alive[i]:
                total = f32(total + e[i])
        target = f32(f32(u) * total)
        c, pick = f32(0), -1
        last = -1
        for i in range(len(e)):
            if not alive[i]:
                continue
            last = i
            c = f32(c + e[i])
            if c >= target:
                pick = i
                break
        if pick < 0:
            pick = last
        alive[pick] = False
        picks.append(pick)
    return picks


def beam_sample_step(cands: List[Tuple[np.ndarray, np.ndarray]], beam_scores: np.ndarray, V: int, uniforms: np.ndarray):
    """One batch item: cands[r] = (kept token ids ascending, warped scores) of beam r; adds the running beam scores, draws
    2 * beams candidates, sorts them by descending score (stable).  -> (scores, tokens, beam indices) each [2 * beams]."""
    flat_s, flat_t, flat_b = [], [], []
    for r, (ids, sc) in enumerate(cands):
        flat_s.append((sc + f32(beam_scores[r])).astype(np.float32))
        flat_t.append(ids)
        flat_b.append(np.full(len(ids), r))
    fs, ft, fb = np.concatenate(flat_s), np.concatenate(flat_t), np.concatenate(flat_b)
    picks = draw_without_replacement(fs, uniforms)
    order = sorted(range(len(picks)), key=lambda j: -float(fs[picks[j]]))  # stable: equal scores keep draw order
    sel = [picks[j] for j in order]
    return fs[sel], ft[sel], fb[sel]


def typical_filter(scores: np.ndarray, mass: float, min_keep: int = 1) -> np.ndarray:
    """The reference's TypicalLogitsWarper (/root/reference/indextts/utils/typical_sampling.py:9-30) on one fp32 row:
    entropy of softmax(scores), tokens ordered by |(-log p) - H| ascending, the shortest prefix whose probability mass
    reaches `mass` is kept (plus at least min_keep), everything else -> -inf."""
    s = np.asarray(scores, dtype=np.float32)
    m = s.max()
    z = np.exp((s - m).astype(np.float32)).astype(np.float32)
    logz = np.float32(np.log(z.sum(dtype=np.float32)))
    normalized = ((s - m) - logz).astype(np.float32)
    p = np.exp(normalized).astype(np.float32)
    with np.errstate(invalid="ignore"):
        prod = normalized * p
    ent = -np.nansum(prod, dtype=np.float32)  # -inf * 0 -> nan -> skipped, as torch.nansum
    shifted = np.abs((-normalized) - ent).astype(np.float32)
    order = np.argsort(shifted, kind="stable")
    sp = p[order]  # (the reference re-normalises: sorted_logits.softmax)
    del sp
    cum = np.cumsum(_softmax32(s[order]), dtype=np.float32)
    last = int((cum < np.float32(mass)).sum())
    last = min(last, len(s) - 1)
    remove_sorted = shifted[order] > shifted[order][last]
    if min_keep > 1:
        remove_sorted[:min_keep] = False
    out = s.copy()
    out[order[remove_sorted]] = -np.inf
    return out


def _softmax32(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    e = np.exp((x - x.max()).astype(np.float32)).astype(np.float32)
    return (e / e.sum(dtype=np.float32)).astype(np.float32)
