// Decode-step kernels (decode.hip).
#pragma once
#include "itts_common.h"

namespace itts {

struct Lin;

struct GemvArgs {
  const float* X = nullptr;  // [B, K] fp32
  const void* W = nullptr;   // [N, K] (fp32 or bf16)
  float* Y = nullptr;        // [B, ldy] fp32
  const float* bias = nullptr;
  int B = 0, N = 0, K = 0, ldy = 0;
  int act = ACT_NONE;
  int accumulate = 0;                // Y += result (residual stream)
  int prologue = 0;                  // 0 plain, 1 LayerNorm(X), 2 LN2(LN(X)), 3 merge split-attention partials
  int x_bf16 = 0;                    // X is bf16 [B, K] (gemv_bf16 only)
  int y_bf16 = 0;                    // Y is bf16 [B, ldy] (gemv_bf16 only; no accumulate)
  const float* ln_gamma = nullptr;
  const float* ln_beta = nullptr;
  const float* ln2_gamma = nullptr;
  const float* ln2_beta = nullptr;
  float ln_eps = 1e-5f;
  int ksplit = 1;            // skinny_mfma: K split across workgroups; > 1 writes raw sums to partial[split][B][ldy]
  float* partial = nullptr;
  const struct Lin* w8src = nullptr;  // engine-internal: projection whose fp8 copy may replace W (decode GEMV)
  // prologue 3 (gemv_bf16): x = the attention output merged from ATTN_NSPLIT partials of decode_attn2 (K = heads * 64)
  const float* attn_o = nullptr;   // [B][heads][ATTN_NSPLIT][64] un-normalised weighted V
  const float* attn_ml = nullptr;  // [B][heads][2][ATTN_NSPLIT] partial max / sum
  const void* W8 = nullptr;      // gemv_bf16: [N, K] OCP fp8 e4m3 bytes (instead of W) ...
  const float* wscale = nullptr; // ... with one scale per output row: y = scale[n] * (x . w8[n]) + bias[n]
  unsigned long long* stamp = nullptr;  // -DITTS_GEMV_STAMPS builds only (tools/ubench_gemv2.hip): s_memtime per phase
  int x_tiled = 0, y_tiled = 0;  // skinny_mfma: bf16 X / Y in MFMA-fragment tiles (tile_off) instead of row-major
  const void* Wt = nullptr;      // skinny_mfma: W as bf16 fragment tiles (wtile_off) - used instead of W when set
  const void* W8t = nullptr;     // skinny_mfma: the fp8 bytes in the same tile order - used instead of W8 when set
  int half_tiles = 0;            // skinny_mfma, <= 16 rows: 8 features per workgroup (narrow projections, no K split)
};

// bf16 activations of the batched decode step live in the operand order of v_mfma_f32_16x16x32_bf16: element (b, k) of a
// [B, K] matrix at ((k/32) * BT + b/16) * 512 + ((k%32)/8 * 16 + b%16) * 8 + k%8, BT = ceil(B/16) - every (k-step, batch
// tile) fragment is one contiguous KiB, which the L2 -> CU path streams ~2.4x faster than 16 rows x 64 B.
__host__ __device__ inline size_t tile_off(int b, int k, int BT) {
  return ((size_t)(k >> 5) * BT + (b >> 4)) * 512 + ((((k & 31) >> 3) << 4) + (b & 15)) * 8 + (k & 7);
}

// bf16 weights of the batched decode step in the same operand order: element (n, k) of W[N, K] at
// ((n/16) * (K/32) + k/32) * 512 + ((k%32)/8 * 16 + n%16) * 8 + k%8 - every wave-load of a (16 features x 32 k) fragment is
// one contiguous KiB (8 whole 128-B lines) instead of 16 half lines 2*K bytes apart, and the K/32 fragments of a feature
// tile are contiguous: the HBM weight stream of the skinny GEMMs runs 1.3x faster (tools/ubench_skinny.hip).
// Rows N .. 16*ceil(N/16) are zero.
__host__ __device__ inline size_t wtile_off(int n, int k, int K) {
  return ((size_t)(n >> 4) * (K >> 5) + (k >> 5)) * 512 + ((((k & 31) >> 3) << 4) + (n & 15)) * 8 + (k & 7);
}
int retile_weights_bf16(void* dst, const void* src, int N, int K, hipStream_t s);
int retile_weights_fp8(void* dst, const void* src, int N, int K, hipStream_t s);  // same order, one byte per element

struct SamplerArgs {
  const float* logits = nullptr;  // [B, V]
  uint8_t* seen = nullptr;        // [B, V]
  int* ids = nullptr;             // [B, max_gen]
  int* cur_tok = nullptr;         // [B]
  int* unfinished = nullptr;      // [B]
  int* step = nullptr;            // [B] tokens generated so far, per row
  int V = 0, max_gen = 0, stop = 0, suppress_stop = 0;
  float penalty = 1.f;
  // fused input embedding of the next step: h_next[b] = emb[tok] + pos[k + 2]
  float* h_next = nullptr;
  const void* emb = nullptr;
  const void* pos = nullptr;
  int D = 0, pos_rows = 0, emb_bf16 = 0;
  // multinomial sampling (HF GenerationMixin.sample: processors -> Temperature -> TopK -> TopP -> softmax -> draw);
  // the draw is an inverse-CDF lookup of uniforms[k * B + b] over the kept tokens in descending-score order
  int do_sample = 0, top_k = 0, B = 0;
  float top_p = 1.f, temperature = 1.f;
  const float* uniforms = nullptr;  // [max_gen][B]
  // forced tokens (HF `input_tokens` continuation, model.py:672-686 / teacher forcing): forced[b][k] >= 0 replaces the
  // choice of step k for row b; nullptr = nothing forced (the table is only read when it exists)
  const int* forced = nullptr;  // [B][max_gen]
  int input_n = 0;  // the first input_n forced tokens are HF `input_tokens`: token k is fed at mel position k + 1, not k + 2
  int preprocessed = 0;  // logits already went through typical_filter (penalty, stop suppression done)
};

// one beam-sample step for every batch item (beam.hip): HF 4.36.2 beam_sample + BeamSearchScorer.process on the device
struct BeamArgs {
  const float* logits = nullptr;  // [B * nb, V]
  int V = 0, max_gen = 0, stop = 0, suppress_stop = 0, nb = 0, B = 0;  // B = batch items (rows = B * nb)
  int top_k = 0, start_tok = 0, fake_id = 1, Smax = 0;
  float penalty = 1.f, top_p = 1.f, temperature = 1.f;
  const float* uniforms = nullptr;  // [max_gen][B][2 * nb]
  int *len = nullptr, *cur_tok = nullptr, *unfinished = nullptr;  // per row
  int* ids = nullptr;          // [2][B * nb][max_gen] id histories, ping-pong by the parity of the step count
  uint8_t* anc = nullptr;      // [2][B * nb][Smax] cache ancestry rows (decode_attn2 ANC), same ping-pong
  const int* prefix_dev = nullptr;
  float* beam_scores = nullptr;  // [B * nb]
  // finished hypotheses per batch item: nb + 1 slots (one spare while the worst is being replaced)
  int* hyp_tok = nullptr;      // [B][nb + 1][max_gen]
  float* hyp_score = nullptr;  // [B][nb + 1]
  int* hyp_len = nullptr;      // [B][nb + 1]
  int* hyp_order = nullptr;    // [B][nb + 1] insertion counter, -1 = free
  int* hyp_n = nullptr;        // [B]
  float* hyp_worst = nullptr;  // [B]
  int* hyp_counter = nullptr;  // [B]
  int* done = nullptr;         // [B]
  float* h_next = nullptr;     // next step's input rows [B * nb][D]
  const void* emb = nullptr;
  const void* pos = nullptr;
  int D = 0, pos_rows = 0, emb_bf16 = 0;
  int preprocessed = 0;  // logits already went through typical_filter (log_softmax, penalty, stop suppression done)
  int do_sample = 1;     // 1: beam_sample (warpers + draws from uniforms); 0: beam_search (top 2 * nb, no warpers)
  float length_penalty = 0.f;  // BeamHypotheses score = sum_logprobs / generated_len ** length_penalty
  // scratch between the two launches of a step: every beam's kept candidates (token-ascending, beam score included)
  float* cand_sc = nullptr;  // [B * nb][BEAM_MAX_CAND]
  int* cand_tok = nullptr;   // [B * nb][BEAM_MAX_CAND]
  int* cand_n = nullptr;     // [B * nb]
  // HF `input_tokens` under beams (model.py:672-686): the first input_n steps of every beam row take forced[row][k] (all >= 0)
  // with the beam scores, the ancestry and the hypotheses untouched; they belong to the decoder prompt, so generated_len
  // (length penalty, is_done) counts from input_n and token k < input_n is fed at mel position k + 1
  const int* forced = nullptr;  // [B * nb][max_gen]
  int input_n = 0;
  // host-side warpers + draws (top_k outside [1, 128] under beams): the 2 * nb picks of every batch item IN DRAW ORDER
  // [B][2 * nb] - score (running beam score included), token, beam; beam_select_kernel then only sorts them and runs the
  // BeamSearchScorer bookkeeping + the history / ancestry swap
  const float* host_sc = nullptr;
  const int* host_tok = nullptr;
  const int* host_beam = nullptr;
};
constexpr int BEAM_MAX_CAND = 128;
int beam_sample_step(const BeamArgs& a, hipStream_t s);
int beam_commit_step(const BeamArgs& a, hipStream_t s);  // host picks -> scorer bookkeeping (beam_select_kernel only)

// TypicalLogitsWarper pre-pass (beam.hip): processed scores of every row -> out [rows, V] with the filtered ones at -inf
struct TypicalArgs {
  const float* logits = nullptr;
  float* out = nullptr;
  int V = 0, stop = 0, suppress_stop = 0, min_keep = 1, log_softmax_first = 0, npad = 0;
  float penalty = 1.f, mass = 0.9f;
  const uint8_t* seen = nullptr;   // [rows, V] byte bitmap (single-beam modes) ...
  const int* beam_ids = nullptr;   // ... or the beam id histories [2][rows][max_gen] with len[] (beam-sample)
  const int* len = nullptr;
  int max_gen = 0, start_tok = 0, fake_id = 1;
};
int typical_filter(const TypicalArgs& a, int rows, hipStream_t s);

int gemv(const GemvArgs& g, int tw, hipStream_t s);
int double_ln(float* y, const float* x, const float* g1, const float* b1, const float* g2, const float* b2, int rows,
              int D, float eps, hipStream_t s);
int kv_scatter(void* kc, void* vc, const void* qkv, int B, int S, int H, int dh, int Smax, int tq, int tc, hipStream_t s);

// second generation (decode2.hip)
bool gemv2_supported(const GemvArgs& g);
int gemv2(const GemvArgs& g, int tw, hipStream_t s);
int decode_attn2(void* ctx, int to, const float* qkv, void* kc, void* vc, const int* len, const int* kv_start,
                 const int* prefix_dev, int B, int H, int dh, int Smax, int tc, hipStream_t s, int ctx_tiled = 0,
                 float* part_o = nullptr, float* part_ml = nullptr, const uint8_t* anc = nullptr, int nb = 1);
// LN + c_attn projection and the cache attention of one layer in one launch (decode2.hip qkv_attn_fused_kernel)
bool qkv_attn_fused_supported(const GemvArgs& g, int H, int dh);
int qkv_attn_fused(const GemvArgs& g, unsigned long long* gran, int* err, void* kc, void* vc, const int* len,
                   const int* kv_start, const int* prefix_dev, int H, int dh, int Smax, float* part_o, float* part_ml,
                   const uint8_t* anc, int nb, hipStream_t s);
constexpr int ATTN_NSPLIT = 4;  // workgroups per (row, head) in the split form of decode_attn2
bool gemv_bf16_supported(const GemvArgs& g);
int gemv_bf16(const GemvArgs& g, hipStream_t s);
int sampler2_step(const SamplerArgs& a, int B, hipStream_t s);
int decode_embed2(float* h, const void* emb, const void* pos, const int* tok, const int* len, int B, int D, int tw,
                  hipStream_t s);

// larger decode batches (decode_mfma.hip): X bf16 [B,K], W bf16 [N,K]; Y fp32 (store / accumulate) or bf16
bool skinny_mfma_supported(const GemvArgs& g);
int skinny_mfma(const GemvArgs& g, hipStream_t s);
int ln_rows_bf16(void* y, float* x, const float* g1, const float* b1, int rows, int D, float eps, int passes,
                 const float* partial, int nsplit, const float* pbias, int y_tiled, hipStream_t s);

}  // namespace itts
