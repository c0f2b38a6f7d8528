// Device-side bookkeeping shared by the sampler kernels (decode2.hip) and the persistent decode engine's in-launch greedy
// sampler (decode_engine.hip): HF greedy search / sample() commit of one token per row + the next step's input embedding.
#pragma once
#include "itts_decode.h"

namespace itts {

// repetition penalty (RepetitionPenaltyLogitsProcessor over the row's id history, kept as a byte map) + stop suppression
__device__ __forceinline__ float sampler_score(const SamplerArgs& a, const uint8_t* seen_row, float v, int i) {
  if (a.preprocessed) return v;
  if (a.penalty != 1.f && seen_row[i]) v = v < 0.f ? v * a.penalty : v / a.penalty;
  if (a.suppress_stop && i == a.stop) v = -INFINITY;
  return v;
}

// bookkeeping shared by the greedy and the sampling kernels (thread 0 of the row's block)
__device__ __forceinline__ void sampler_commit(const SamplerArgs& a, int b, int choice, int* si, int k, int unf) {
  si[1] = -1;
  if (k < a.max_gen) {  // graph replays past the end are no-ops
    if (a.forced) {
      const int f = a.forced[(size_t)b * a.max_gen + k];
      choice = f >= 0 ? f : choice;
    }
    int tok = unf ? choice : a.stop;
    tok = tok < 0 ? 0 : (tok >= a.V ? a.V - 1 : tok);  // never index seen[] / emb[] outside the vocabulary
    a.ids[(size_t)b * a.max_gen + k] = tok;
    a.cur_tok[b] = tok;
    a.seen[(size_t)b * a.V + tok] = 1;
    a.unfinished[b] = unf && tok != a.stop;
    a.step[b] = k + 1;
    si[0] = tok;
    // position of the token fed at the next step: 0, 2, 3, ... (model.py:153-155); a given `input_tokens` token k was part of
    // the reference's first forward, at position k + 1 (model.py:141-144)
    si[1] = k < a.input_n ? k + 1 : k + 2;
  }
}

// next step's input row h[b] = mel_emb[tok] + mel_pos[k + 2], fused here (one launch less per token)
__device__ __forceinline__ void sampler_next_embedding(const SamplerArgs& a, int b, const int* si, int tid) {
  if (a.h_next && si[1] > 0) {
    const int tok = si[0], p = min(si[1], a.pos_rows - 1);
    for (int i = tid; i < a.D; i += 1024) {
      float v;
      if (a.emb_bf16)
        v = (float)((const bf16_t*)a.emb)[(size_t)tok * a.D + i] + (float)((const bf16_t*)a.pos)[(size_t)p * a.D + i];
      else
        v = ((const float*)a.emb)[(size_t)tok * a.D + i] + ((const float*)a.pos)[(size_t)p * a.D + i];
      a.h_next[(size_t)b * a.D + i] = v;
    }
  }
}

}  // namespace itts
