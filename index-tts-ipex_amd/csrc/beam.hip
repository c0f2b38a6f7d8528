// Beam-sample on the device: one step of HF transformers 4.36.2 `GenerationMixin.beam_sample` + `BeamSearchScorer.process`
// (the generate() mode the reference's DEFAULT kwargs select: do_sample=True, num_beams=3, top_k=30, top_p=0.8,
// length_penalty=0.0, repetition_penalty=10.0 - indextts/infer.py:116-124, indextts/gpt/model.py:690-703), for every
// batch item, without a host round trip per token.  Restated as oracle/hf_beam.py; pinned by tests/golden/micro_beam_*.
//
// Two launches per step.  beam_cand_kernel, one 1024-thread workgroup per (batch item, beam) - the beams of an item in
// parallel:
//   log_softmax(logits) -> RepetitionPenalty over the beam's own id history (a bitmap rebuilt in LDS from that history:
//   beams swap histories every step) -> Temperature -> TopK (4-pass radix select, min_tokens_to_keep = 2) -> TopP
//   (min_tokens_to_keep = 2) -> + running beam score -> the beam's kept candidates (token-ascending) to a scratch row.
// beam_select_kernel, one workgroup per batch item: softmax over the kept candidates of all beams in flat (beam-major, token-ascending) order, 2 * nb
// draws WITHOUT replacement by inverse CDF of caller uniforms, sort by score, the BeamSearchScorer bookkeeping
// (finished hypotheses, worst score, done test).  Then all threads re-order what the beams own: id histories and the
// KV-cache ancestry rows (see decode_attn2_kernel ANC - the cache itself is never copied), and prepare the next
// step's input embeddings.
#include "itts_decode.h"

namespace itts {
namespace {

constexpr int MAXB = 10;   // beams per batch item (the reference web UI offers 1..10)
constexpr int MAXC = BEAM_MAX_CAND;  // kept candidates per beam (top_k <= 128; the UI offers 0..100)

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

// bitonic sort of MAXC (value, index) pairs in LDS by ONE wave (lanes 0..63 = the MAXC / 2 comparators of a stage): a
// wave's LDS operations execute in program order, so the 28 stages need no workgroup barrier (at 16 waves each barrier
// costs ~0.4 us and the block form spent 11 us per sort); the fence only pins the compiler's order.
// BY_SCORE: descending value, ascending index on ties; else ascending index.
template <bool BY_SCORE>
__device__ __forceinline__ void sort_cands_wave(float* cv, int* ci, int lane) {
  static_assert(MAXC == 128, "one comparator per lane");
  for (int kq = 2; kq <= MAXC; kq <<= 1)
    for (int j = kq >> 1; j > 0; j >>= 1) {
      const int lo = ((lane & ~(j - 1)) << 1) | (lane & (j - 1)), hi = lo | j;
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
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// value of lane l (wave-uniform l) in every lane: v_readlane_b32, no LDS crossbar round trip
__device__ __forceinline__ float lane_val(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// histogram increment aggregated over the wave: the digits of log-probabilities crowd into a handful of bins (the top
// byte is sign + high exponent bits), and 64 lanes adding to one LDS word serialise; here each distinct digit of the
// wave costs one atomic.  Every lane of the wave must call it (act = does this lane contribute).
__device__ __forceinline__ void hist_add_wave(unsigned* hist, unsigned digit, bool act, int lane) {
  unsigned long long m = __ballot(act);
  while (m) {  // wave-uniform
    const int leader = __ffsll((long long)m) - 1;
    const unsigned dl = (unsigned)__shfl((int)digit, leader, 64);
    const unsigned long long same = __ballot(act && digit == dl);
    if (lane == leader) atomicAdd(&hist[dl], (unsigned)__popcll(same));
    m &= ~same;
  }
}

__global__ __launch_bounds__(1024) void beam_cand_kernel(BeamArgs a) {
  extern __shared__ float ssc[];  // [V] processed scores of this beam
  __shared__ unsigned seenw[512];  // V <= 16384 bits
  __shared__ unsigned hist[256];
  __shared__ float red[16];
  __shared__ int s_bin, s_k, s_cnt;
  __shared__ float cval[MAXC];
  __shared__ int cidx[MAXC];
  const int r = blockIdx.x, bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int nb = a.nb, V = a.V, Beff = a.B * nb, mg = a.max_gen;
  const int k = a.len[bi * nb];  // tokens generated so far: the same for every beam (they step together)
  if (k >= mg || a.done[bi]) return;  // graph replays past the end / finished batch items are no-ops
  const int par = k & 1;
  const int* ids_old = a.ids + (size_t)par * Beff * mg;
  const int nd = 2 * nb;
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
    for (int i0 = 0; i0 < V; i0 += 1024) {  // every lane takes part in every round (wave-aggregated atomics)
      const int i = i0 + tid;
      const unsigned key = i < V ? okey(ssc[i]) : 0u;
      const bool act = i < V && (pass == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8)));
      hist_add_wave(hist, (key >> shift) & 255u, act, lane);
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
  if (tid < 64) sort_cands_wave<true>(cval, cidx, lane);  // descending score, ascending index on ties
  __syncthreads();
  if (tid < 64) {
    // TopP (ascending cumulative probability <= 1 - top_p goes; the best min_tokens_to_keep = 2 always stay).  The two
    // running sums are the sequential fp32 sums of the restatement; the exponentials and quotients are computed by the
    // lanes in parallel (lane q and q + 64 of the sorted candidates) and broadcast one by one
    int R = n;
    if (a.do_sample && a.top_p < 1.f) {
      const float m = cval[0];
      const float e0 = lane < n ? expf(cval[lane] - m) : 0.f, e1 = lane + 64 < n ? expf(cval[lane + 64] - m) : 0.f;
      float Z = 0.f;
      for (int q = 0; q < n; ++q) Z += q < 64 ? lane_val(e0, q) : lane_val(e1, q - 64);
      const float t0 = e0 / Z, t1 = e1 / Z;
      float tail = 0.f;
      R = 1;
      for (int q = n - 1; q >= 1; --q) {
        tail += q < 64 ? lane_val(t0, q) : lane_val(t1, q - 64);
        if (!(tail <= 1.f - a.top_p)) {
          R = q + 1;
          break;
        }
      }
      R = min(max(R, 2), n);
    }
    if (lane == 0) s_cnt = R;
  }
  __syncthreads();
  const int R = s_cnt;
  if (tid < MAXC && tid >= R) {  // dropped by top-p (or never filled): out of the token-order sort
    cval[tid] = 0.f;
    cidx[tid] = 0x7fffffff;
  }
  __syncthreads();
  // the kept ones in token order (the flat index order of next_token_scores.view(batch, beams * vocab))
  if (tid < 64) sort_cands_wave<false>(cval, cidx, lane);
  __syncthreads();
  if (tid < MAXC) {
    a.cand_sc[(size_t)row * MAXC + tid] = cval[tid] + a.beam_scores[row];
    a.cand_tok[(size_t)row * MAXC + tid] = cidx[tid];
    if (tid == 0) a.cand_n[row] = R;
  }
}

__global__ __launch_bounds__(1024) void beam_select_kernel(BeamArgs a) {
  // the kept candidates of the item's beams in flat (beam-major, token-ascending) order, running beam score included
  __shared__ float fsc[MAXB * MAXC];
  __shared__ int ftok[MAXB * MAXC];
  __shared__ float fe[MAXB * MAXC];  // exp(score - max)
  __shared__ unsigned char fbeam[MAXB * MAXC], falive[MAXB * MAXC];
  __shared__ int cand_n[MAXB], cand_off[MAXB + 1];
  __shared__ float red[16];
  __shared__ float p_sc[2 * MAXB];  // the 2 * nb picks in draw order
  __shared__ int p_tok[2 * MAXB], p_beam[2 * MAXB];
  __shared__ int nxt_src[MAXB], nxt_tok[MAXB];  // new beam k continues physical row nxt_src[k] with token nxt_tok[k]
  __shared__ float nxt_score[MAXB];
  __shared__ int add_slot[MAXB], add_src[MAXB], n_add;  // hypotheses finished this step: copy history of add_src into slot
  __shared__ int s_done;
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int nb = a.nb, Beff = a.B * nb, mg = a.max_gen;
  const int k = a.len[bi * nb];
  if (k >= mg) return;
  const bool given = a.forced && k < a.input_n;  // an `input_tokens` step: every beam takes the given token
  const int par = k & 1;
  const int* ids_old = a.ids + (size_t)par * Beff * mg;
  int* ids_new = a.ids + (size_t)(par ^ 1) * Beff * mg;
  const uint8_t* anc_old = a.anc + (size_t)par * Beff * a.Smax;
  uint8_t* anc_new = a.anc + (size_t)(par ^ 1) * Beff * a.Smax;
  const int was_done = a.done[bi];
  const int pref0 = a.prefix_dev[0];
  const int nd = 2 * nb;
  // the scorer's per-item state, requested now by the thread that will use it after the picks
  int hn_pre = 0, counter_pre = 0;
  float worst_pre = 0.f;
  if (tid == 0) {
    hn_pre = a.hyp_n[bi];
    worst_pre = a.hyp_worst[bi];
    counter_pre = a.hyp_counter[bi];
  }
  if (!was_done && !given && a.host_sc) {  // the caller warped and drew (gpt_commit_beams): picks in draw order
    if (tid < nd) {
      p_sc[tid] = a.host_sc[(size_t)bi * nd + tid];
      p_tok[tid] = a.host_tok[(size_t)bi * nd + tid];
      p_beam[tid] = a.host_beam[(size_t)bi * nd + tid];
    }
    __syncthreads();
  } else if (!was_done && !given) {  // block-uniform
    if (tid == 0) {
      int o = 0;
      for (int r = 0; r < nb; ++r) {
        cand_off[r] = o;
        cand_n[r] = a.cand_n[bi * nb + r];
        o += cand_n[r];
      }
      cand_off[nb] = o;
    }
    __syncthreads();
    const int T = cand_off[nb];
    float mloc = -INFINITY;
    for (int i = tid; i < nb * MAXC; i += 1024) {
      const int r = i / MAXC, q = i - r * MAXC;
      if (q < cand_n[r]) {
        const int f = cand_off[r] + q;
        const float v = a.cand_sc[(size_t)(bi * nb + r) * MAXC + q];
        fsc[f] = v;
        ftok[f] = a.cand_tok[(size_t)(bi * nb + r) * MAXC + q];
        fbeam[f] = (unsigned char)r;
        falive[f] = 1;
        mloc = fmaxf(mloc, v);
      }
    }
    const float m = block_max(mloc, red, tid);  // (barriers inside: the flat arrays are complete after it)
    if (a.do_sample)
      for (int f = tid; f < T; f += 1024) fe[f] = expf(fsc[f] - m);
    __syncthreads();
    // ---- wave 0: the 2 * nb picks.  beam_sample: draws WITHOUT replacement by inverse CDF over the live candidates in
    //      flat order.  The running sums are the sequential fp32 sums of the restatement (oracle/hf_beam.py
    //      draw_without_replacement) - same order, same roundings - but every lane carries them: a chunk of 64 values is
    //      loaded by the wave at once and broadcast lane by lane (v_readlane), dead entries add 0.0f (exact).  One thread
    //      walking LDS, as before, paid an LDS round trip per element: 66 us per step at 3 beams x 30 candidates.
    if (tid < 64) {
      for (int j = 0; j < nd; ++j) {
        int pick = -1;
        if (a.do_sample) {
          float total = 0.f;
          for (int c0 = 0; c0 < T; c0 += 64) {
            const int f = c0 + lane;
            const float ev = (f < T && falive[f]) ? fe[f] : 0.f;
#pragma unroll
            for (int l = 0; l < 64; ++l) total += lane_val(ev, l);
          }
          const float target = a.uniforms[((size_t)k * a.B + bi) * nd + j] * total;
          float c = 0.f;
          int last = -1;
          for (int c0 = 0; c0 < T && pick < 0; c0 += 64) {
            const int f = c0 + lane;
            const bool al = f < T && falive[f];
            const float ev = al ? fe[f] : 0.f;
            const unsigned long long am = __ballot(al);
#pragma unroll
            for (int l = 0; l < 64; ++l) {
              const bool a_l = (am >> l) & 1ull;
              c += lane_val(ev, l);
              last = a_l ? c0 + l : last;
              pick = (a_l && pick < 0 && c >= target) ? c0 + l : pick;
            }
          }
          if (pick < 0) pick = last;
        } else {
          // beam_search: torch.topk(next_token_scores.view(batch, beams * vocab), 2 * beams): the best remaining candidate,
          // the lower flat index on ties
          float bv = -INFINITY;
          int bf = 0x7fffffff;
          for (int f = lane; f < T; f += 64)
            if (falive[f] && (fsc[f] > bv || bf == 0x7fffffff)) {
              bv = fsc[f];
              bf = f;
            }
          for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int of = __shfl_xor(bf, o, 64);
            if (of != 0x7fffffff && (bf == 0x7fffffff || ov > bv || (ov == bv && of < bf))) {
              bv = ov;
              bf = of;
            }
          }
          pick = bf == 0x7fffffff ? -1 : bf;
        }
        if (lane == 0) {
          if (pick < 0) {  // fewer live candidates than picks (cannot happen with min_tokens_to_keep = 2): repeat a stop
            p_sc[j] = -INFINITY;
            p_tok[j] = a.stop;
            p_beam[j] = 0;
          } else {
            falive[pick] = 0;
            p_sc[j] = fsc[pick];
            p_tok[j] = ftok[pick];
            p_beam[j] = fbeam[pick];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the next pick reads falive
      }
    }
    __syncthreads();
  }
  // ---- one thread: sort the picks, BeamSearchScorer.process ----
  if (tid == 0) {
    n_add = 0;
    if (given) {
      for (int q = 0; q < nb; ++q) {
        nxt_src[q] = bi * nb + q;
        nxt_tok[q] = a.forced[(size_t)(bi * nb + q) * mg + k];
        nxt_score[q] = a.beam_scores[bi * nb + q];  // untouched: beam_sample zeros / beam_search [0, -1e9, ..]
      }
      s_done = was_done;
    } else if (was_done) {
      for (int q = 0; q < nb; ++q) {
        nxt_src[q] = bi * nb + q;  // (HF points done batches at row 0; nothing of a done batch is read again)
        nxt_tok[q] = a.stop;
        nxt_score[q] = 0.f;
      }
      s_done = 1;
    } else {
      // (the picks stay in LDS: dynamically indexed local arrays would live in scratch memory)
      float* psc = p_sc;
      int *ptok = p_tok, *pbeam = p_beam;
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
      int hn = hn_pre;
      float worst = worst_pre;
      int counter = counter_pre;
      int filled = 0;
      const float lpdiv = a.length_penalty == 0.f ? 1.f : powf((float)(k + 1 - a.input_n), a.length_penalty);  // generated_len
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
  // ---- beams swap histories: ids and cache ancestry of new beam q come from physical row nxt_src[q]; next step's input
  //      rows mel_emb[tok] + mel_pos[k + 2] (positions 0, 2, 3, ...: model.py:153-155).  The three copies are flattened
  //      over the beams and staged through registers - every load of a round is issued before its first store, so a round
  //      costs one memory round trip (a loop per beam and array paid nine) ----
  const int pos_next = pref0 + k + 1;  // where the next step appends (own physical row)
  const int n_ids = nb * k, n_anc = nb * a.Smax, n_h = a.h_next ? nb * a.D : 0;
  const int pp = min(k < a.input_n ? k + 1 : k + 2, a.pos_rows - 1);  // a given token k was in the first forward, at position k + 1
  for (int rd = 0;; ++rd) {
    const int b_ids = rd * 2048, b_anc = rd * 4096, b_h = rd * 4096;
    if (b_ids >= n_ids && b_anc >= n_anc && b_h >= n_h) break;
    int vi[2];
    uint8_t va[4];
    float vh[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = b_ids + j * 1024 + tid;
      if (i < n_ids) {
        const int q = i / k, e = i - q * k;
        vi[j] = ids_old[(size_t)nxt_src[q] * mg + e];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = b_anc + j * 1024 + tid;
      if (i < n_anc) {
        const int q = i / a.Smax, e = i - q * a.Smax;
        va[j] = e == pos_next ? (uint8_t)q : anc_old[(size_t)nxt_src[q] * a.Smax + e];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = b_h + j * 1024 + tid;
      if (i < n_h) {
        const int q = i / a.D, e = i - q * a.D, tok = nxt_tok[q];
        if (a.emb_bf16)
          vh[j] = (float)((const bf16_t*)a.emb)[(size_t)tok * a.D + e] + (float)((const bf16_t*)a.pos)[(size_t)pp * a.D + e];
        else
          vh[j] = ((const float*)a.emb)[(size_t)tok * a.D + e] + ((const float*)a.pos)[(size_t)pp * a.D + e];
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = b_ids + j * 1024 + tid;
      if (i < n_ids) {
        const int q = i / k, e = i - q * k;
        ids_new[(size_t)(bi * nb + q) * mg + e] = vi[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = b_anc + j * 1024 + tid;
      if (i < n_anc) {
        const int q = i / a.Smax, e = i - q * a.Smax;
        anc_new[(size_t)(bi * nb + q) * a.Smax + e] = va[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = b_h + j * 1024 + tid;
      if (i < n_h) a.h_next[(size_t)bi * nb * a.D + i] = vh[j];
    }
  }
  if (tid < nb) {
    const int q = tid, dst = bi * nb + q;
    ids_new[(size_t)dst * mg + k] = nxt_tok[q];
    a.cur_tok[dst] = nxt_tok[q];
    a.len[dst] = k + 1;
    a.beam_scores[dst] = nxt_score[q];
    a.unfinished[dst] = !s_done;
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
  ITTS_REQUIRE(a.cand_sc && a.cand_tok && a.cand_n, "beam_sample: candidate scratch missing");
  hipLaunchKernelGGL(beam_cand_kernel, dim3(a.nb, a.B), dim3(1024), (size_t)a.V * 4, s, a);
  hipLaunchKernelGGL(beam_select_kernel, dim3(a.B), dim3(1024), 0, s, a);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int beam_commit_step(const BeamArgs& a, hipStream_t s) {
  ITTS_REQUIRE(a.nb >= 2 && a.nb <= MAXB, "beam_commit: 2 <= num_beams <= 10");
  ITTS_REQUIRE(a.host_sc && a.host_tok && a.host_beam, "beam_commit: picks missing");
  ITTS_REQUIRE(a.ids && a.anc && a.len && a.hyp_tok && a.done, "beam_commit: null state");
  hipLaunchKernelGGL(beam_select_kernel, dim3(a.B), dim3(1024), 0, s, a);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
