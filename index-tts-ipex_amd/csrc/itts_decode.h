// Decode-step kernels (decode.hip).
#pragma once
#include "itts_common.h"

namespace itts {

struct GemvArgs {
  const float* X = nullptr;  // [B, K] fp32
  const void* W = nullptr;   // [N, K] (fp32 or bf16)
  float* Y = nullptr;        // [B, ldy] fp32
  const float* bias = nullptr;
  int B = 0, N = 0, K = 0, ldy = 0;
  int act = ACT_NONE;
  int accumulate = 0;                // Y += result (residual stream)
  const float* ln_gamma = nullptr;   // fused LayerNorm of X (over K) when non-null
  const float* ln_beta = nullptr;
  float ln_eps = 1e-5f;
};

struct SamplerArgs {
  const float* logits = nullptr;  // [B, V]
  uint8_t* seen = nullptr;        // [B, V]
  int* ids = nullptr;             // [B, max_gen]
  int* cur_tok = nullptr;         // [B]
  int* unfinished = nullptr;      // [B]
  int* step = nullptr;            // [1] tokens generated so far
  int* n_unfinished = nullptr;    // [1] rows still running after the last completed step
  int* n_unfinished_next = nullptr;
  int V = 0, max_gen = 0, stop = 0, suppress_stop = 0;
  float penalty = 1.f;
};

int decode_embed(float* h, const void* emb, const void* pos, const int* tok, const int* step, int B, int D, int tw,
                 hipStream_t s);
int gemv(const GemvArgs& g, int tw, hipStream_t s);
int decode_attn(float* ctx, const float* qkv, void* kc, void* vc, const int* step, const int* kv_start,
                const int* prefix_dev, int B, int H, int dh, int Smax, int tc, hipStream_t s);
int double_ln(float* y, const float* x, const float* g1, const float* b1, const float* g2, const float* b2, int rows,
              int D, float eps, hipStream_t s);
int sampler_step(const SamplerArgs& a, int B, hipStream_t s);
int kv_scatter(void* kc, void* vc, const void* qkv, int B, int S, int H, int dh, int Smax, int tq, int tc, hipStream_t s);

}  // namespace itts
