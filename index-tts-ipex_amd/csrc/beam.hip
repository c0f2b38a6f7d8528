// Beam-sample on the device: one step of HF transformers 4.36.2 `GenerationMixin.beam_sample` + `BeamSearchScorer.process`
// (the generate() mode the reference's DEFAULT kwargs select: do_sample=True, num_beams=3, top_k=30, top_p=0.8,
// length_penalty=0.0, repetition_penalty=10.0 - indextts/infer.py:116-124, indextts/gpt/model.py:690-703), for every
// batch item, without a host round trip per token.  Restated as oracle/hf_beam.py; pinned by tests/golden/micro_beam_*.
//
// One 1024-thread workgroup per batch item, its nb beams one after the other:
//   log_softmax(logits) -> RepetitionPenalty over the beam's own id history (a bitmap rebuilt in LDS from that history:
//   beams swap histories every step) -> Temperature -> TopK (4-pass radix select, min_tokens_to_keep = 2) -> TopP
//   (min_tokens_to_keep = 2) -> + running beam score.
// Then one thread: softmax over the kept candidates of all beams in flat (beam-major, token-ascending) order, 2 * nb
// draws WITHOUT replacement by inverse CDF of caller uniforms, sort by score, the BeamSearchScorer bookkeeping
// (finished hypotheses, worst score, done test).  Then all threads re-order what the beams own: id histories and the
// KV-cache ancestry rows (see decode_attn2_kernel ANC - the cache itself is never copied), and prepare the next
// step's input embeddings.
#include "itts_decode.h"

namespace itts {
namespace {

constexpr int MAXB = 10;   // beams per batch item (the reference web UI offers 1..10)
constexpr int MAXC = 128;  // kept candidates per beam (top_k <= 128; the UI offers 0..100)

__device__ __forceinline__ unsigned okey(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < 16; ++i) r = fmaxf(r, red[i]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += red[i];
  __syncthreads();
  return r;
}

// bitonic sort of MAXC (value, index) pairs in LDS by the whole block (every thread reaches the barriers).
// BY_SCORE: descending value, ascending index on ties; else ascending index.
template <bool BY_SCORE>
__device__ __forceinline__ void sort_cands(float* cv, int* ci, int tid) {
  for (int kq = 2; kq <= MAXC; kq <<= 1)
    for (int j = kq >> 1; j > 0; j >>= 1) {
      if (tid < MAXC / 2) {
        const int lo = ((tid & ~(j - 1)) << 1) | (tid & (j - 1)), hi = lo | j;
        const bool up = (lo & kq) == 0;
        const float v0 = cv[lo], v1 = cv[hi];
        const int i0 = ci[lo], i1 = ci[hi];
        const bool second_first = BY_SCORE ? (v1 > v0 || (v1 == v0 && i1 < i0)) : (i1 < i0);
        if (second_first == up) {
          cv[lo] = v1;
          cv[hi] = v0;
          ci[lo] = i1;
          ci[hi] = i0;
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void beam_sample_kernel(BeamArgs a) {
  extern __shared__ float ssc[];  // [V] processed scores of the beam being worked on
  __shared__ unsigned seenw[512];  // V <= 16384 bits
  __shared__ unsigned hist[256];
  __shared__ float red[16];
  __shared__ int s_bin, s_k, s_cnt;
  __shared__ float cval[MAXC];
  __shared__ int cidx[MAXC];
  __shared__ float cand_sc[MAXB][MAXC];  // kept candidates per beam, token-ascending, running beam score included
  __shared__ int cand_tok[MAXB][MAXC];
  __shared__ float cand_e[MAXB][MAXC];   // exp(score - max) and the not-yet-drawn flag of the draw loop
  __shared__ unsigned char cand_alive[MAXB][MAXC];
  __shared__ int cand_n[MAXB];
  __shared__ int nxt_src[MAXB], nxt_tok[MAXB];  // new beam k continues physical row nxt_src[k] with token nxt_tok[k]
  __shared__ float nxt_score[MAXB];
  __shared__ int add_slot[MAXB], add_src[MAXB], n_add;  // hypotheses finished this step: copy history of add_src into slot
  __shared__ int s_done;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int nb = a.nb, V = a.V, Beff = a.B * nb, mg = a.max_gen;
  const int k = a.len[bi * nb];  // tokens generated so far: the same for every beam (they step together)
  if (k >= mg) return;           // graph replays past the end are no-ops
  const int par = k & 1;
  const int* ids_old = a.ids + (size_t)par * Beff * mg;
  int* ids_new = a.ids + (size_t)(par ^ 1) * Beff * mg;
  const uint8_t* anc_old = a.anc + (size_t)par * Beff * a.Smax;
  uint8_t* anc_new = a.anc + (size_t)(par ^ 1) * Beff * a.Smax;
  const int was_done = a.done[bi];
  const int nd = 2 * nb;
  if (!was_done) {
    for (int r = 0; r < nb; ++r) {
      const int row = bi * nb + r;
      const float* __restrict__ lg = a.logits + (size_t)row * V;
      // ---- log_softmax (a.preprocessed: the typical pre-pass already did this, the penalty and the suppression) ----
      float lse = 0.f;
      if (!a.preprocessed) {
        float mx = -INFINITY;
        for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, lg[i]);
        mx = block_max(mx, red, tid);
        float se = 0.f;
        for (int i = tid; i < V; i += 1024) se += expf(lg[i] - mx);
        se = block_sum(se, red, tid);
        lse = mx + logf(se);
      }
      // ---- ids this beam has seen: the fake prompt ids (all 1, then start_mel: model.py:644-653) + its history ----
      for (int i = tid; i < (V + 31) / 32; i += 1024) seenw[i] = 0u;
      __syncthreads();
      if (tid == 0) {
        atomicOr(&seenw[a.fake_id >> 5], 1u << (a.fake_id & 31));
        atomicOr(&seenw[a.start_tok >> 5], 1u << (a.start_tok & 31));
      }
      for (int i = tid; i < k; i += 1024) {
        const int t = ids_old[(size_t)row * mg + i];
        atomicOr(&seenw[t >> 5], 1u << (t & 31));
      }
      __syncthreads();
      for (int i = tid; i < V; i += 1024) {
        float v = lg[i] - lse;
        if (!a.preprocessed) {
          if (a.penalty != 1.f && ((seenw[i >> 5] >> (i & 31)) & 1u)) v = v < 0.f ? v * a.penalty : v / a.penalty;
          if (a.suppress_stop && i == a.stop) v = -INFINITY;
        }
        if (a.do_sample && a.temperature != 1.f) v = v / a.temperature;  // warpers only exist in beam_sample
        ssc[i] = v;
      }
      // ---- the kk largest scores by radix select: beam_sample TopK (kk = max(top_k, min_tokens_to_keep = 2));
      //      beam_search (no warpers): the row can contribute at most the 2 * nb best of the batch item's 2 * nb ----
      unsigned prefix = 0;
      int kk = a.do_sample ? min(max(a.top_k, 2), V) : min(nd, V);
      for (int pass = 3; pass >= 0; --pass) {
        const int shift = pass * 8;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (int i = tid; i < V; i += 1024) {
          const unsigned key = okey(ssc[i]);
          if (pass == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
          const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
          const unsigned own = h0 + h1 + h2 + h3;
          unsigned x = own;
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_down(x, off, 64);
            if (lane + off < 64) x += t;
          }
          const unsigned above = x - own;
          if (above < (unsigned)kk && (unsigned)kk <= x) {
            unsigned acc = above;
            int bin = 4 * lane + 3;
            const unsigned hb[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int j = 3; j >= 0; --j) {
              if (acc + hb[j] >= (unsigned)kk) {
                bin = 4 * lane + j;
                break;
              }
              acc += hb[j];
            }
            s_bin = bin;
            s_k = kk - (int)acc;
          }
        }
        __syncthreads();
        prefix |= (unsigned)s_bin << shift;
        kk = s_k;
      }
      if (tid == 0) s_cnt = 0;
      if (tid < MAXC) {
        cval[tid] = -INFINITY;
        cidx[tid] = 0x7fffffff;
      }
      __syncthreads();
      for (int i = tid; i < V; i += 1024) {
        const float v = ssc[i];
        // (-inf scores - filtered by the typical pre-pass, the suppressed stop - never count: when fewer than top_k finite
        // scores exist the k-th largest is -inf and HF's `scores < kth` removes nothing, i.e. keeps exactly the finite ones)
        if (okey(v) >= prefix && v > -INFINITY) {
          const int p = atomicAdd(&s_cnt, 1);
          if (p < MAXC) {
            cval[p] = v;
            cidx[p] = i;
          }
        }
      }
      __syncthreads();
      const int n = min(s_cnt, MAXC);
      sort_cands<true>(cval, cidx, tid);  // descending score, ascending index on ties
      if (tid == 0) {
        // TopP (ascending cumulative probability <= 1 - top_p goes; the best min_tokens_to_keep = 2 always stay)
        int R = n;
        if (a.do_sample && a.top_p < 1.f) {
          const float m = cval[0];
          float Z = 0.f;
          for (int q = 0; q < n; ++q) Z += expf(cval[q] - m);
          float tail = 0.f;
          R = 1;
          for (int q = n - 1; q >= 1; --q) {
            tail += expf(cval[q] - m) / Z;
            if (!(tail <= 1.f - a.top_p)) {
              R = q + 1;
              break;
            }
          }
          R = min(max(R, 2), n);
        }
        s_cnt = R;
      }
      __syncthreads();
      const int R = s_cnt;
      if (tid < MAXC && tid >= R) {  // dropped by top-p (or never filled): out of the token-order sort
        cval[tid] = 0.f;
        cidx[tid] = 0x7fffffff;
      }
      __syncthreads();
      // the kept ones in token order (the flat index order of next_token_scores.view(batch, beams * vocab))
      sort_cands<false>(cval, cidx, tid);
      if (tid < MAXC) {
        cand_sc[r][tid] = cval[tid] + a.beam_scores[row];
        cand_tok[r][tid] = cidx[tid];
        if (tid == 0) cand_n[r] = R;
      }
      __syncthreads();
    }
  }
  // ---- one thread: draws (or top-2nb), sort, BeamSearchScorer.process ----
  if (tid == 0) {
    n_add = 0;
    if (was_done) {
      for (int q = 0; q < nb; ++q) {
        nxt_src[q] = bi * nb + q;  // (HF points done batches at row 0; nothing of a done batch is read again)
        nxt_tok[q] = a.stop;
        nxt_score[q] = 0.f;
      }
      s_done = 1;
    } else {
      float psc[2 * MAXB];
      int ptok[2 * MAXB], pbeam[2 * MAXB];
      if (a.do_sample) {
        float m = -INFINITY;
        for (int r = 0; r < nb; ++r)
          for (int q = 0; q < cand_n[r]; ++q) m = fmaxf(m, cand_sc[r][q]);
        for (int r = 0; r < nb; ++r)
          for (int q = 0; q < cand_n[r]; ++q) {
            cand_alive[r][q] = 1;
            cand_e[r][q] = expf(cand_sc[r][q] - m);
          }
        const float* u = a.uniforms + ((size_t)k * a.B + bi) * nd;
        for (int j = 0; j < nd; ++j) {
          float total = 0.f;
          for (int r = 0; r < nb; ++r)
            for (int q = 0; q < cand_n[r]; ++q)
              if (cand_alive[r][q]) total += cand_e[r][q];
          const float target = u[j] * total;
          float c = 0.f;
          int pr = -1, pq = -1, lr = -1, lq = -1;
          for (int r = 0; r < nb && pr < 0; ++r)
            for (int q = 0; q < cand_n[r]; ++q) {
              if (!cand_alive[r][q]) continue;
              lr = r;
              lq = q;
              c += cand_e[r][q];
              if (c >= target) {
                pr = r;
                pq = q;
                break;
              }
            }
          if (pr < 0) {
            pr = lr;
            pq = lq;
          }
          if (pr < 0) {  // fewer live candidates than draws (cannot happen with min_tokens_to_keep = 2): repeat a stop
            psc[j] = -INFINITY;
            ptok[j] = a.stop;
            pbeam[j] = 0;
            continue;
          }
          cand_alive[pr][pq] = 0;
          psc[j] = cand_sc[pr][pq];
          ptok[j] = cand_tok[pr][pq];
          pbeam[j] = pr;
        }
      } else {
        // beam_search: torch.topk(next_token_scores.view(batch, beams * vocab), 2 * beams): repeatedly the best remaining
        // candidate, the lower flat index on ties (each row holds its own best 2 * nb in token order)
        for (int r = 0; r < nb; ++r)
          for (int q = 0; q < cand_n[r]; ++q) cand_alive[r][q] = 1;
        for (int j = 0; j < nd; ++j) {
          int pr = -1, pq = -1;
          for (int r = 0; r < nb; ++r)
            for (int q = 0; q < cand_n[r]; ++q)
              if (cand_alive[r][q] && (pr < 0 || cand_sc[r][q] > cand_sc[pr][pq])) {
                pr = r;
                pq = q;
              }
          if (pr < 0) {
            psc[j] = -INFINITY;
            ptok[j] = a.stop;
            pbeam[j] = 0;
            continue;
          }
          cand_alive[pr][pq] = 0;
          psc[j] = cand_sc[pr][pq];
          ptok[j] = cand_tok[pr][pq];
          pbeam[j] = pr;
        }
      }
      // torch.sort(descending): stable insertion sort (equal scores keep draw order)
      for (int i = 1; i < nd; ++i) {
        const float s0 = psc[i];
        const int t0 = ptok[i], b0 = pbeam[i];
        int j = i - 1;
        while (j >= 0 && psc[j] < s0) {
          psc[j + 1] = psc[j];
          ptok[j + 1] = ptok[j];
          pbeam[j + 1] = pbeam[j];
          --j;
        }
        psc[j + 1] = s0;
        ptok[j + 1] = t0;
        pbeam[j + 1] = b0;
      }
      // BeamSearchScorer.process (beam_search.py, 4.36.2); generated_len = cur_len - decoder_prompt_len = k + 1
      float* hs = a.hyp_score + (size_t)bi * (nb + 1);
      int* hl = a.hyp_len + (size_t)bi * (nb + 1);
      int* ho = a.hyp_order + (size_t)bi * (nb + 1);  // insertion counter, -1 = free slot
      int hn = a.hyp_n[bi];
      float worst = a.hyp_worst[bi];
      int counter = a.hyp_counter[bi];
      int filled = 0;
      const float lpdiv = a.length_penalty == 0.f ? 1.f : powf((float)(k + 1), a.length_penalty);
      for (int rank = 0; rank < nd && filled < nb; ++rank) {
        if (ptok[rank] == a.stop) {
          if (rank >= nb) continue;
          const float score = psc[rank] / lpdiv;  // sum_logprobs / generated_len ** length_penalty
          if (hn < nb || score > worst) {
            int slot = 0;
            while (ho[slot] >= 0) ++slot;  // nb + 1 slots, at most nb in use here
            hs[slot] = score;
            hl[slot] = k;  // the hypothesis is the history WITHOUT the stop token
            ho[slot] = counter++;
            add_slot[n_add] = slot;
            add_src[n_add] = bi * nb + pbeam[rank];
            ++n_add;
            ++hn;
            if (hn > nb) {  // drop the lowest (score, insertion order)
              int lo = -1;
              for (int q = 0; q <= nb; ++q)
                if (ho[q] >= 0 && (lo < 0 || hs[q] < hs[lo] || (hs[q] == hs[lo] && ho[q] < ho[lo]))) lo = q;
              ho[lo] = -1;
              --hn;
              worst = INFINITY;
              for (int q = 0; q <= nb; ++q)
                if (ho[q] >= 0) worst = fminf(worst, hs[q]);
            } else {
              worst = fminf(score, worst);
            }
          }
        } else {
          nxt_score[filled] = psc[rank];
          nxt_tok[filled] = ptok[rank];
          nxt_src[filled] = bi * nb + pbeam[rank];
          ++filled;
        }
      }
      for (; filled < nb; ++filled) {  // HF raises here; keep the state well-formed
        nxt_score[filled] = -INFINITY;
        nxt_tok[filled] = a.stop;
        nxt_src[filled] = bi * nb;
      }
      a.hyp_n[bi] = hn;
      a.hyp_worst[bi] = worst;
      a.hyp_counter[bi] = counter;
      // is_done(best_sum_logprobs = the best of the 2 * nb candidates, cur_len = k + 1 generated; early_stopping False)
      s_done = (hn >= nb && worst >= psc[0] / lpdiv) ? 1 : 0;
      a.done[bi] = s_done;
    }
  }
  __syncthreads();
  // ---- all threads: finished hypotheses keep a copy of their history ----
  for (int q = 0; q < n_add; ++q) {
    int* dst = a.hyp_tok + ((size_t)bi * (nb + 1) + add_slot[q]) * mg;
    const int* src = ids_old + (size_t)add_src[q] * mg;
    for (int i = tid; i < k; i += 1024) dst[i] = src[i];
  }
  // ---- beams swap histories: ids and cache ancestry of new beam q come from physical row nxt_src[q] ----
  const int pos_next = a.prefix_dev[0] + k + 1;  // where the next step appends (own physical row)
  for (int q = 0; q < nb; ++q) {
    const int dst = bi * nb + q, src = nxt_src[q];
    for (int i = tid; i < k; i += 1024) ids_new[(size_t)dst * mg + i] = ids_old[(size_t)src * mg + i];
    for (int i = tid; i < a.Smax; i += 1024)
      anc_new[(size_t)dst * a.Smax + i] = i == pos_next ? (uint8_t)q : anc_old[(size_t)src * a.Smax + i];
    if (tid == 0) {
      ids_new[(size_t)dst * mg + k] = nxt_tok[q];
      a.cur_tok[dst] = nxt_tok[q];
      a.len[dst] = k + 1;
      a.beam_scores[dst] = nxt_score[q];
      a.unfinished[dst] = !s_done;
    }
    if (a.h_next) {  // next step's input row: mel_emb[tok] + mel_pos[k + 2] (positions 0, 2, 3, ...: model.py:153-155)
      const int tok = nxt_tok[q], p = min(k + 2, a.pos_rows - 1);
      for (int i = tid; i < a.D; i += 1024) {
        float v;
        if (a.emb_bf16)
          v = (float)((const bf16_t*)a.emb)[(size_t)tok * a.D + i] + (float)((const bf16_t*)a.pos)[(size_t)p * a.D + i];
        else
          v = ((const float*)a.emb)[(size_t)tok * a.D + i] + ((const float*)a.pos)[(size_t)p * a.D + i];
        a.h_next[(size_t)dst * a.D + i] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// TypicalLogitsWarper as a pre-pass (the reference's optional `typical_sampling=True`, gpt/model.py:690-697 +
// utils/typical_sampling.py:9-30): HF places it in the logits_processor list right after RepetitionPenalty, i.e. before
// the Temperature / TopK / TopP warpers.  One 1024-thread workgroup per row: processed scores (log_softmax first under
// beams, repetition penalty, stop suppression) -> entropy H of their softmax -> tokens sorted by |(-log p) - H| (block
// bitonic sort in LDS) -> running probability mass in that order (block scan) -> everything behind the point where
// the mass reaches `mass` is set to -inf (the first min_keep always stay).  The samplers then run on `out` with
// `preprocessed` set (no second log_softmax / penalty).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void typical_filter_kernel(TypicalArgs a) {
  extern __shared__ unsigned char tsm[];
  const int NP = a.npad;
  float* keys = reinterpret_cast<float*>(tsm);                       // [NP]
  unsigned short* idx = reinterpret_cast<unsigned short*>(tsm + (size_t)NP * 4);  // [NP]
  __shared__ unsigned seenw[512];
  __shared__ float red[16];
  __shared__ float wsum[16];
  __shared__ int s_last;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, V = a.V;
  const float* __restrict__ lg = a.logits + (size_t)row * V;
  float* __restrict__ out = a.out + (size_t)row * V;
  // ---- seen set: the byte bitmap of the single-beam samplers, or (beams) the row's id history ----
  if (a.beam_ids) {
    const int k = a.len[row], mg = a.max_gen;
    const int* hist = a.beam_ids + ((size_t)(k & 1) * gridDim.x + row) * mg;
    for (int i = tid; i < (V + 31) / 32; i += 1024) seenw[i] = 0u;
    __syncthreads();
    if (tid == 0) {
      atomicOr(&seenw[a.fake_id >> 5], 1u << (a.fake_id & 31));
      atomicOr(&seenw[a.start_tok >> 5], 1u << (a.start_tok & 31));
    }
    for (int i = tid; i < k; i += 1024) {
      const int t = hist[i];
      atomicOr(&seenw[t >> 5], 1u << (t & 31));
    }
    __syncthreads();
  }
  float lse0 = 0.f;
  if (a.log_softmax_first) {
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, lg[i]);
    mx = block_max(mx, red, tid);
    float se = 0.f;
    for (int i = tid; i < V; i += 1024) se += expf(lg[i] - mx);
    se = block_sum(se, red, tid);
    lse0 = mx + logf(se);
  }
  // ---- processed scores -> out; their log-sum-exp ----
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 1024) {
    float v = lg[i] - lse0;
    const bool seen = a.beam_ids ? ((seenw[i >> 5] >> (i & 31)) & 1u) != 0 : (a.seen && a.seen[(size_t)row * V + i]);
    if (a.penalty != 1.f && seen) v = v < 0.f ? v * a.penalty : v / a.penalty;
    if (a.suppress_stop && i == a.stop) v = -INFINITY;
    out[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, red, tid);
  __syncthreads();  // out[] written by this block is read back below
  float se = 0.f;
  for (int i = tid; i < V; i += 1024) se += expf(out[i] - mx);
  se = block_sum(se, red, tid);
  const float lse = mx + logf(se);
  float en = 0.f;
  for (int i = tid; i < V; i += 1024) {
    const float nl = out[i] - lse;
    if (nl > -INFINITY) en += nl * expf(nl);  // nansum: (-inf) * 0 is skipped
  }
  const float ent = -block_sum(en, red, tid);
  for (int i = tid; i < NP; i += 1024) {
    float key = INFINITY;
    if (i < V) {
      const float nl = out[i] - lse;
      key = fabsf((-nl) - ent);  // +inf for removed scores: they sort last
    }
    keys[i] = key;
    idx[i] = (unsigned short)(i < V ? i : 0xFFFF);
  }
  __syncthreads();
  // ---- bitonic sort, ascending (key, index) ----
  for (int kk = 2; kk <= NP; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < NP / 2; t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const bool up = (lo & kk) == 0;
        const float k0 = keys[lo], k1 = keys[hi];
        const unsigned short i0 = idx[lo], i1 = idx[hi];
        const bool gt = k0 > k1 || (k0 == k1 && i0 > i1);
        if (gt == up) {
          keys[lo] = k1;
          keys[hi] = k0;
          idx[lo] = i1;
          idx[hi] = i0;
        }
      }
      __syncthreads();
    }
  }
  // ---- running mass in sorted order: per-thread runs of NP / 1024 consecutive entries, block scan of the run sums ----
  const int per = NP / 1024;
  float loc[16];
  float mine = 0.f;
  for (int e = 0; e < per; ++e) {
    const int j = tid * per + e;
    const unsigned short ix = idx[j];
    const float p = (j < V && ix != 0xFFFF) ? expf(out[ix] - lse) : 0.f;
    loc[e] = p;
    mine += p;
  }
  float inc = mine;  // inclusive scan over the wave, then over the waves
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  if (tid == 0) s_last = 0;
  __syncthreads();
  float base = inc - mine;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int cnt = 0;
  float c = base;
  for (int e = 0; e < per; ++e) {
    c += loc[e];
    if (tid * per + e < V && c < a.mass) ++cnt;
  }
  if (cnt) atomicAdd(&s_last, cnt);
  __syncthreads();
  const int last = min(s_last, V - 1);
  const float thr = keys[last];
  for (int j = tid; j < V; j += 1024)
    if (keys[j] > thr && j >= a.min_keep && idx[j] != 0xFFFF) out[idx[j]] = -INFINITY;
}

}  // namespace

int typical_filter(const TypicalArgs& a, int rows, hipStream_t s) {
  ITTS_REQUIRE(a.logits && a.out && a.V > 1 && a.V <= 16384 && a.mass > 0.f && a.mass < 1.f, "typical_filter: bad arguments (V <= 16384, 0 < mass < 1)");
  TypicalArgs t = a;
  t.npad = 1024;
  while (t.npad < a.V) t.npad <<= 1;
  const size_t lds = (size_t)t.npad * 6;
  static bool attr_done = false;
  if (!attr_done) {
    ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)typical_filter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(typical_filter_kernel, dim3(rows), dim3(1024), lds, s, t);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int beam_sample_step(const BeamArgs& a, hipStream_t s) {
  ITTS_REQUIRE(a.nb >= 2 && a.nb <= MAXB, "beam_sample: 2 <= num_beams <= 10");
  ITTS_REQUIRE(!a.do_sample || (a.top_k >= 1 && a.top_k <= MAXC && a.top_p > 0.f && a.temperature > 0.f),
               "beam_sample: 1 <= top_k <= 128, top_p > 0, temperature > 0");
  ITTS_REQUIRE(a.V <= 15000, "beam_sample: vocabulary too large for the LDS-resident sampler");
  ITTS_REQUIRE(a.logits && (a.uniforms || !a.do_sample) && a.ids && a.anc && a.len && a.hyp_tok && a.done, "beam_sample: null state");
  hipLaunchKernelGGL(beam_sample_kernel, dim3(a.B), dim3(1024), (size_t)a.V * 4, s, a);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
