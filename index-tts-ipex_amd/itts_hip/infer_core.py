"""Host-side integer logic of the `IndexTTS.infer` / `infer_fast` orchestration (product code, numpy only).

Mirrors /root/reference/indextts/infer.py: `remove_long_silence` (:244-298), `bucket_sentences` (:303-315),
`pad_tokens_cat` (:316-318).  Parity with the reference's own function is pinned by
tests/golden/silence_cases.npz (tests/test_host_logic.py)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def remove_long_silence(codes: np.ndarray, stop_mel_token: int, silent_token: int = 52,
                        max_consecutive: int = 30) -> Tuple[np.ndarray, np.ndarray]:
    """codes int [B, T] -> (codes', code_lens[B]).  Each row is cut at its first stop token; a row holding more
    than `max_consecutive` silent tokens in total keeps at most 10 consecutive ones (infer.py:262-280); ragged
    rows are right-padded with the stop token (:287) and the batch is clipped to the longest kept row (:294-296)."""
    codes = np.asarray(codes)
    assert codes.ndim == 2
    lens: List[int] = []
    rows: List[np.ndarray] = []
    fixed = False
    for code in codes:
        hit = np.nonzero(code == stop_mel_token)[0]
        n = int(hit[0]) if len(hit) else code.shape[0]
        if int((code == silent_token).sum()) > max_consecutive:
            body = code[:n]
            is_sil = body == silent_token
            # run position of each silent token inside its run (0-based); non-silent tokens reset the run
            idx = np.arange(n)
            last_non = np.maximum.accumulate(np.where(~is_sil, idx, -1))
            run_pos = idx - last_non - 1
            keep = (~is_sil) | (run_pos < 10)
            rows.append(body[keep])
            n = int(keep.sum())
            fixed = True
        else:
            rows.append(code[:n])
        lens.append(n)
    if fixed:
        width = max(r.shape[0] for r in rows)
        out = np.full((len(rows), width), stop_mel_token, dtype=codes.dtype)
        for i, r in enumerate(rows):
            out[i, : r.shape[0]] = r
        codes = out
    m = max(lens)
    if m < codes.shape[1]:
        codes = codes[:, :m]
    return codes, np.asarray(lens, dtype=np.int64)


def bucket_sentences(sentences: Sequence[Sequence], bucket_max_size: int = 4) -> List[List[Dict]]:
    """infer.py:303-315: keep order if everything fits one bucket, else sort by length into buckets."""
    outputs = [{"idx": i, "sent": s, "len": len(s)} for i, s in enumerate(sentences)]
    if len(outputs) <= bucket_max_size:
        return [outputs]
    buckets: List[List[Dict]] = []
    for item in sorted(outputs, key=lambda x: x["len"]):
        if not buckets or len(buckets[-1]) >= bucket_max_size:
            buckets.append([item])
        else:
            buckets[-1].append(item)
    return buckets


def pad_tokens_cat(tokens: Sequence[np.ndarray], stop_text_token: int) -> np.ndarray:
    """infer.py:316-318: right-pad id rows with the stop text token into [B, Lmax]."""
    rows = [np.asarray(t).reshape(-1) for t in tokens]
    width = max(r.shape[0] for r in rows)
    out = np.full((len(rows), width), stop_text_token, dtype=np.int32)
    for i, r in enumerate(rows):
        out[i, : r.shape[0]] = r
    return out


def sampling_kwargs(do_sample, num_beams, top_k, top_p, temperature, typical_sampling=False, typical_mass=0.9,
                    length_penalty=0.0) -> dict:
    """HF `generate` kwargs (infer.py:116-124) -> Engine.generate keywords; every mode runs on the device:
      do_sample, num_beams > 1       beam_sample (the reference default: 3 beams)
      do_sample, num_beams == 1      multinomial sampling
      not do_sample, num_beams > 1   beam_search (deterministic)
      not do_sample, num_beams == 1  greedy
      typical_sampling               the reference's TypicalLogitsWarper(typical_mass) behind the repetition penalty - a logits
                                     PROCESSOR in the reference (model.py:690-697), so greedy and beam search run it too
    Limits of the device samplers (the web UI offers num_beams 1..10 and top_k 0..100): num_beams <= 10; at most 128 kept
    candidates per row.  top_k = 0 / None (HF: TopK warper off) or > 128 is exact all the same: the token choice then runs
    on the host over the whole vocabulary (host_sample_step / host_beam_step below, one logits read-back per token) - with
    one beam or several.  The seed is drawn from torch's global RNG so that torch.manual_seed governs the run as it does for
    the reference's torch.multinomial."""
    import warnings

    import torch

    nb = 1 if num_beams is None else max(1, int(num_beams))
    if typical_sampling and not (0.0 < float(typical_mass) < 1.0):
        raise ValueError(f"`typical_mass` has to be a float > 0 and < 1, but is {typical_mass}")  # model.py:692-693
    if nb > 10:
        warnings.warn(f"itts_hip: num_beams={nb} > 10 is not supported; using 10", RuntimeWarning)
        nb = 10
    lp = float(length_penalty or 0.0)
    tm = float(typical_mass) if typical_sampling else 0.0
    if not do_sample:
        kw = dict(num_beams=nb, length_penalty=lp) if nb > 1 else {}
        if tm:
            kw["typical_mass"] = tm
        return kw
    k = int(top_k) if top_k else 0
    if k < 1:
        k = 0  # HF: TopK warper off - exact on the host (Engine.generate takes the host-sampling path, with or without beams)
    p = 1.0 if top_p is None else float(top_p)
    return dict(do_sample=True, top_k=k, top_p=min(max(p, 1e-6), 1.0), temperature=float(temperature or 1.0), num_beams=nb,
                typical_mass=tm, length_penalty=lp,
                seed=int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))


# ---- host-side token choice (HF GenerationMixin.sample over the whole vocabulary) ----
def host_distribution(scores: np.ndarray, seen_ids, penalty: float, temperature: float, top_k: int, top_p: float,
                      typical_mass: float, stop: int, suppress_stop: bool, min_keep: int = 1, log_softmax_first: bool = False,
                      return_scores: bool = False):
    """One row of HF 4.36.2 `sample()`'s score pipeline, in the order generate() builds it for infer.py:116-124 /
    model.py:688-703: RepetitionPenaltyLogitsProcessor over every id seen so far (fake prefix id, start token, generated
    codes) -> [TypicalLogitsWarper] -> TemperatureLogitsWarper -> TopKLogitsWarper (off when top_k < 1) -> TopPLogitsWarper.
    Returns (kept token ids in descending-score order - ties: lower id first -, un-normalised weights exp(s - s_max)).  fp32.
    Beams (beam_sample): log_softmax_first (the processors see log-probabilities), min_keep = 2 (min_tokens_to_keep of every
    warper and of the typical filter, model.py:693-694); return_scores: the warped scores instead of the weights."""
    # The kept SET is computed with torch's CPU ops in the order and form of the HF 4.36.2 classes (sort -> softmax -> cumsum,
    # topk's k-th value, log_softmax): with thousands of kept tokens the top-p boundary depends on the last bits of a
    # cumulative sum, and the reference's arithmetic IS torch's - a numpy restatement parts from it once in a few hundred steps.
    import torch

    s = torch.from_numpy(np.asarray(scores, dtype=np.float32).copy())
    V = s.shape[0]
    if log_softmax_first:
        s = torch.log_softmax(s, dim=-1)
    ids = torch.from_numpy(np.fromiter(seen_ids, dtype=np.int64))
    if penalty != 1.0 and ids.numel():
        v = s[ids]
        s[ids] = torch.where(v < 0, v * float(penalty), v / float(penalty))  # RepetitionPenaltyLogitsProcessor
    if suppress_stop:
        s[stop] = -float("inf")
    if typical_mass and 0.0 < typical_mass < 1.0:
        s = torch.from_numpy(_typical_filter(s.numpy(), float(typical_mass), min_keep))
    if temperature != 1.0:
        s = s / float(temperature)  # TemperatureLogitsWarper
    if top_k and top_k >= 1:  # TopKLogitsWarper: top_k = max(top_k, min_tokens_to_keep), ties with the k-th value stay
        kk = min(max(int(top_k), min_keep), V)
        s = s.masked_fill(s < torch.topk(s, kk)[0][-1], -float("inf"))
    if top_p is not None and top_p < 1.0:  # TopPLogitsWarper
        sorted_logits, sorted_idx = torch.sort(s, descending=False)
        remove = sorted_logits.softmax(dim=-1).cumsum(dim=-1) <= (1 - float(top_p))
        remove[-min_keep:] = False
        s[sorted_idx[remove]] = -float("inf")
    s = s.numpy()
    keep = np.nonzero(np.isfinite(s))[0]
    order = np.lexsort((keep, -s[keep].astype(np.float64)))
    idx = keep[order]
    e = np.exp((s[idx] - s[idx[0]]).astype(np.float32)).astype(np.float32)
    n = idx.size
    return (idx[:n], s[idx[:n]]) if return_scores else (idx[:n], e[:n])


def host_beam_step(logits: np.ndarray, ids_hist: np.ndarray, k: int, beam_scores: np.ndarray, done: np.ndarray, nb: int,
                   penalty: float, temperature: float, top_k: int, top_p: float, typical_mass: float, u: np.ndarray, stop: int,
                   suppress_stop: bool, start_tok: int, fake_id: int = 1):
    """One step of HF 4.36.2 `beam_sample` up to (not including) BeamSearchScorer.process, for every batch item, on the host:
    per beam log_softmax -> RepetitionPenalty over the beam's own ids (the fake prompt id, the start token, its k generated
    codes) -> [Typical] -> Temperature -> TopK -> TopP (min_tokens_to_keep = 2) -> + running beam score; the kept candidates
    of an item's beams in flat (beam-major, token-ascending) order = next_token_scores.view(batch, beams * vocab); 2 * nb
    draws without replacement as sequential inverse-CDF look-ups of the caller's uniforms u [items, 2 * nb] (the convention
    of the device sampler, beam.hip beam_select_kernel: fp32 running sums in flat order).
    logits [items * nb, V], ids_hist [items * nb, >= k], beam_scores [items * nb], done [items].
    -> (scores, tokens, beams) each [items, 2 * nb], in DRAW order (itts_gpt_commit_beams sorts them and runs the scorer)."""
    lg = np.asarray(logits, dtype=np.float32)
    items = lg.shape[0] // nb
    nd = 2 * nb
    psc = np.zeros((items, nd), dtype=np.float32)
    ptok = np.full((items, nd), stop, dtype=np.int32)
    pbeam = np.zeros((items, nd), dtype=np.int32)
    for bi in range(items):
        if done[bi]:
            continue
        fs, ft, fb = [], [], []
        for r in range(nb):
            row = bi * nb + r
            seen = {int(fake_id), int(start_tok)} | {int(t) for t in ids_hist[row, :k]}
            idx, sc = host_distribution(lg[row], seen, penalty, temperature, top_k, top_p, typical_mass, stop, suppress_stop,
                                        min_keep=2, log_softmax_first=True, return_scores=True)
            o = np.argsort(idx, kind="stable")
            fs.append((sc[o] + np.float32(beam_scores[row])).astype(np.float32))
            ft.append(idx[o])
            fb.append(np.full(idx.size, r, dtype=np.int32))
        fs, ft, fb = np.concatenate(fs), np.concatenate(ft), np.concatenate(fb)
        e = np.exp((fs - fs.max()).astype(np.float32)).astype(np.float32)
        alive = np.ones(e.size, dtype=bool)
        for j in range(nd):
            c = np.cumsum(np.where(alive, e, np.float32(0.0)), dtype=np.float32)  # sequential fp32 sums; a dead entry adds 0.0 (exact)
            target = np.float32(np.float32(u[bi, j]) * c[-1])
            hit = np.nonzero(alive & (c >= target))[0]
            live = np.nonzero(alive)[0]
            if live.size == 0:
                break  # fewer live candidates than picks (cannot happen with min_tokens_to_keep = 2): the stop-token defaults stay
            pick = int(hit[0]) if hit.size else int(live[-1])
            alive[pick] = False
            psc[bi, j], ptok[bi, j], pbeam[bi, j] = fs[pick], ft[pick], fb[pick]
        if live.size == 0:
            psc[bi, j:] = -np.inf
    return psc, ptok, pbeam


def host_sample_step(logits: np.ndarray, seen: Sequence[set], penalty: float, temperature: float, top_k: int, top_p: float,
                     typical_mass: float, u: np.ndarray, stop: int, suppress_stop: bool) -> np.ndarray:
    """One multinomial draw per row over host_distribution, taken as the inverse-CDF lookup of the caller's uniform u[row]
    over the kept tokens in descending-score order - the convention of the device samplers."""
    lg = np.asarray(logits, dtype=np.float32)
    out = np.empty(lg.shape[0], dtype=np.int32)
    for r in range(lg.shape[0]):
        idx, e = host_distribution(lg[r], seen[r], penalty, temperature, top_k, top_p, typical_mass, stop, suppress_stop)
        c = np.cumsum(e, dtype=np.float32)
        target = np.float32(np.float32(u[r]) * c[-1])
        out[r] = int(idx[min(int(np.searchsorted(c, target, side="left")), idx.size - 1)])
    return out


def _typical_filter(s: np.ndarray, mass: float, min_keep: int = 1) -> np.ndarray:
    """The reference's TypicalLogitsWarper (indextts/utils/typical_sampling.py:9-30) on one fp32 row; min_keep =
    min_tokens_to_keep (2 under beams, model.py:693-694)."""
    m = s.max()
    z = np.exp((s - m).astype(np.float32)).astype(np.float32)
    normalized = ((s - m) - np.float32(np.log(z.sum(dtype=np.float32)))).astype(np.float32)
    p = np.exp(normalized).astype(np.float32)
    with np.errstate(invalid="ignore"):
        ent = -np.nansum(normalized * p, dtype=np.float32)
    shifted = np.abs((-normalized) - ent).astype(np.float32)
    order = np.argsort(shifted, kind="stable")
    so = s[order]
    eo = np.exp((so - so.max()).astype(np.float32)).astype(np.float32)
    cum = np.cumsum(eo / eo.sum(dtype=np.float32), dtype=np.float32)
    last = min(int((cum < np.float32(mass)).sum()), s.size - 1)
    out = s.copy()
    remove = shifted[order] > shifted[order][last]
    if min_keep > 1:
        remove[:min_keep] = False
    out[order[remove]] = -np.inf
    return out
