// Persistent decode engine (decode_engine.hip): one launch for the 24 GPT-2 blocks of a token step, decode batches <= 4.
#pragma once
#include "itts_common.h"
#include "itts_decode.h"

namespace itts {

constexpr int ENG_D = 1280, ENG_H = 20, ENG_NCU = 256, ENG_MAX_ROWS = 6;
constexpr int ENG_MAX_LAYERS = 24;
constexpr bool ENG_DEFAULT_ON = true;   // ITTS_ENGINE=0 keeps the five-launches-per-block path  // IndexTTS-1.5 GPT on the 256 CUs of an MI355X

struct EngLayerW {  // one GPT-2 block: bf16 [N][K] projections (LayerNorm affine folded in by the packer), fp32 biases
  const bf16_t *wa, *wp, *wf, *w2;
  const float *ba, *bp, *bf, *b2;
};

struct EngArgs {
  EngLayerW L[ENG_MAX_LAYERS];        // in the kernel-argument segment: block l's pointers are scalar loads (a table in global
                                      // memory is re-read with VECTOR loads wherever hipcc cannot prove it unaliased - a dependent
                                      // memory latency, and a vmcnt(0) in front of the weight-prefetch issue, in every phase)
  unsigned long long* gran = nullptr; // hand-off granules: eng_gran_count(NL) words, zeroed once at allocation
  float* h = nullptr;                 // residual stream [B][D]: this step's input embedding in, last block's output out
  bf16_t* kc = nullptr;               // KV cache [NL][B][H][Smax][64]
  bf16_t* vc = nullptr;
  const int* len = nullptr;           // [B] tokens generated so far
  const int* kv_start = nullptr;      // [B] first valid (not left-padded) cache row
  const int* prefix = nullptr;        // [0] prefix length
  unsigned* ctr = nullptr;            // [0] step counter = hand-off tag (starts at 1, never 0), [1] abort word
  const uint8_t* anc = nullptr;       // beam rows: ancestry [2][B][Smax] (ping-pong by the parity of len), nb rows per batch item
  int nb = 1;
  int NL = 0, B = 0, Smax = 0;
  float scale = 0.125f, eps = 1e-5f;
  unsigned timeout_ticks = 2000000;   // wall-clock bound of every wait, 100 MHz ticks (20 ms)
  int first_delay = 14, pass_sleep = 1; // gather pacing (s_sleep units of 64 clocks): before the first pass / between passes
  int ctx_delay = 0, act_delay = 16;  // the same for the context edge behind its sentinel poll / the gelu(fc) edge
  // head inside the launch (null head_w: the last block writes h and the head GEMV is its own launch)
  const bf16_t* head_w = nullptr;     // mel_head [V][D] (final_norm's affine folded in)
  const float* head_b = nullptr;      // [V]
  const float* lnf_g = nullptr;       // ln_f gamma / beta [D]
  const float* lnf_b = nullptr;
  float* logits = nullptr;            // [B][V]
  int V = 0;
  // greedy sampler inside the launch (needs the head inside it): every workgroup publishes its best (score, id) per row,
  // workgroup b picks row b's token, does sampler2_kernel's bookkeeping and writes the next step's input embedding
  int fold_sampler = 0;
  SamplerArgs samp;
  unsigned long long* cand = nullptr;  // [B][256][2] hand-off granules of the candidates (tail of the granule buffer)
  int early_fc = 0;                   // > 0: c_fc requested this many x 64 clocks behind c_attn on the workgroups without attention
  int early_kv = 1;                   // the attention workgroups request their cache rows at the head of the block (0: behind c_attn, the r03 placement)
  int fake_div = 1;                   // TIMING PROBE ONLY (ITTS_ENG_FAKE_DIV): gather 1 / fake_div of the h and gelu(fc) edges - wrong results
  int thin_fc = 0;                    // the loader keeps at most 12 c_fc requests in flight (they run beside a gather)
  float* dbg = nullptr;               // debugging aid (ITTS_TAP_LAYER): qkv [B][3D], h1 [B][D], act [B][4D], h2 [B][D] of block dbg_layer
  int dbg_layer = -1;
  unsigned* stamp = nullptr;          // debugging aid (ITTS_ENGINE_STAMPS): [256][NL][12] wall-clock stamps (100 MHz) of one step
};

size_t eng_gran_count(int layers);  // 8-byte words of the granule buffer (sized for ENG_MAX_ROWS rows; the last ENG_CAND_WORDS: sampler candidates)
constexpr size_t ENG_CAND_WORDS = (size_t)ENG_MAX_ROWS * ENG_NCU * 2;
int decode_engine_layers(const EngArgs& a, hipStream_t s);

}  // namespace itts
