// The inference engine behind the C ABI: owns nothing but scratch/state memory; weights stay in the
// caller's arena and are bound by name (itts_engine_bind_tensor).
#pragma once
#include <map>
#include <utility>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/itts_hip.h"
#include "itts_decode.h"
#include "itts_engine_kernel.h"
#include "itts_kernels.h"

namespace itts {

struct Tensor {
  const void* p = nullptr;
  int dt = 0, nd = 0;
  int64_t d[4] = {0, 0, 0, 0};
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < nd; ++i) n *= d[i];
    return n;
  }
};

// a linear / conv layer: W [nphase][N][taps*Cin] in `dt`, bias fp32 [N] (optional), BN-eval affine (optional)
struct Lin {
  const void* w = nullptr;
  const float* b = nullptr;
  const float* bn_scale = nullptr;
  const float* bn_shift = nullptr;
  int N = 0, Cin = 0, taps = 1, nphase = 1, dt = 0;
  const void* w8 = nullptr;       // optional fp8 e4m3 copy of w (decode GEMV of the GPT), one scale per output row
  const float* wscale = nullptr;
  void* wt = nullptr;             // bf16 copy in MFMA-fragment tiles (wtile_off) for the batched decode step, made on first use
  void* wt8 = nullptr;            // the same for the fp8 copy (8 bytes per lane: half-KiB fragments)
};
struct Norm {
  const float* g = nullptr;
  const float* b = nullptr;
};

struct ConformerLayerW {
  Norm norm_mha, norm_conv, norm_ff, norm_final, conv_norm;
  Lin qkv, pos, out, pw1, pw2, w1, w2;
  const float *bu = nullptr, *bv = nullptr, *dw_w = nullptr, *dw_b = nullptr;
};
struct CondW {
  bool ok = false;
  const float *conv_w = nullptr, *conv_b = nullptr;
  Lin embed_out;
  const void* pe = nullptr;  // [max_len, od] activation dtype
  int pe_len = 0;
  std::vector<ConformerLayerW> layers;
  Norm after_norm;
  // perceiver
  const float* latents = nullptr;
  Lin proj;
  struct PL {
    Lin to_q, to_kv, to_out, ff1, ff2;
  };
  std::vector<PL> pl;
  const float* gamma = nullptr;
  int ffi = 0, ffi_pad = 0, inner = 0;
};
struct GptLayerW {
  Norm ln1, ln2;
  Lin attn, proj, fc, proj2;
};
struct GptW {
  bool ok = false;
  std::vector<GptLayerW> layers;
  Norm ln_f, final_norm;
  Lin head;
  const void *text_emb = nullptr, *mel_emb = nullptr, *mel_pos = nullptr, *text_pos = nullptr;
};
struct AmpW {
  Lin c1[4], c2[4];
  const float *a1[4], *b1[4], *a2[4], *b2[4];
};
struct BigvganW {
  bool ok = false;
  Lin conv_pre, cond_layer, conv_post;
  std::vector<Lin> ups, conds;
  std::vector<AmpW> res;
  const float *post_alpha = nullptr, *post_beta = nullptr, *filter = nullptr;
};
struct EcapaW {
  bool ok = false;
  Lin b0;
  struct Blk {
    Lin tdnn1, tdnn2, se1, se2;
    std::vector<Lin> res;
  };
  std::vector<Blk> blks;
  Lin mfa, asp_x, asp_ms, asp_conv, fc;
  const float *aspbn_scale = nullptr, *aspbn_shift = nullptr;
};
struct DvaeW {
  bool ok = false;
  const void* codebook = nullptr;
  Lin in_conv, out_conv;
  struct RB {
    Lin c0, c2, c4;
  };
  std::vector<RB> rbs;
  std::vector<Lin> ups;
  // encoder of get_codebook_indices (optional: only when the checkpoint carries it)
  bool enc_ok = false;
  std::vector<Lin> enc;   // stride-2 convs as 2-tap convs over paired rows
  std::vector<RB> erbs;
  Lin eout, quant;        // 1x1 to the codebook dim; quant = the codebook as a [tokens, dim] projection
  const float* codebook_sq = nullptr;
};

struct DecodeState {
  int B = 0, Smax = 0, prefix = 0, max_gen = 0;
  size_t cache_bytes = 0;
  void *kc = nullptr, *vc = nullptr;  // [layers][B][H][Smax][dh]
  float *h = nullptr, *qkv = nullptr, *ctx = nullptr, *act = nullptr, *hn = nullptr, *logits = nullptr;
  float *attn_o = nullptr, *attn_ml = nullptr;  // split decode attention partials (decode2.hip ATTN_NSPLIT), small batches
  float* partial = nullptr;           // [4][B][D] split-K partial sums of the residual projections (batched decode)
  int pend_split = 0;                 // launch-time bookkeeping: partials waiting to be absorbed by the next LayerNorm
  const float* pend_bias = nullptr;
  int *len = nullptr, *prefix_dev = nullptr;  // len[b]: tokens generated so far by row b
  int *kv_start = nullptr, *cur_tok = nullptr, *ids = nullptr, *unfinished = nullptr;
  uint8_t* seen = nullptr;
  int cap_B = 0, cap_gen = 0;
  float penalty = 1.f;
  int suppress_stop = 0;
  hipGraphExec_t graph = nullptr, graphK = nullptr;  // one decode step / ITTS_GRAPH_STEPS (8) steps per launch
  int graph_B = 0, graph_Smax = 0, graph_suppress = 0, graph_max_gen = 0;
  // multinomial sampling (itts_gpt_set_sampling): parameters baked into the captured step, uniforms [max_gen][B]
  int do_sample = 0, top_k = 0, graph_sample = 0, graph_top_k = 0;
  float top_p = 1.f, temperature = 1.f, graph_top_p = 1.f, graph_temperature = 1.f;
  float* uniforms = nullptr;
  size_t uniforms_cap = 0;
  // beam-sample (itts_gpt_set_beam_sample): B = batch items * nb rows; state of beam.hip
  int nb = 1, graph_nb = 1;          // beams per batch item (1 = off)
  int beam_sample = 1, graph_beam_sample = 1;  // 1: beam_sample (draws), 0: beam_search (deterministic top-2nb)
  float length_penalty = 0.f, graph_length_penalty = 0.f;
  int* beam_ids = nullptr;           // [2][B][max_gen]
  uint8_t* anc = nullptr;            // [2][B][Smax]
  float* beam_scores = nullptr;      // [B]
  float* cand_sc = nullptr;          // beam sampler scratch [rows][BEAM_MAX_CAND]
  int *cand_tok = nullptr, *cand_n = nullptr;
  int *hyp_tok = nullptr, *hyp_len = nullptr, *hyp_order = nullptr, *hyp_n = nullptr, *hyp_counter = nullptr, *beam_done = nullptr;
  float *hyp_score = nullptr, *hyp_worst = nullptr;
  size_t beam_cap = 0;               // bytes-independent capacity key: rows * max_gen * Smax the beam buffers were sized for
  int beam_rows = 0, beam_gen = 0, beam_smax = 0;
  // LN + c_attn + cache attention in one launch (decode2.hip qkv_attn_fused): per-layer granule buffers + error flag
  unsigned long long* gran = nullptr;  // [layers][cap_B <= 4][3 * D]
  int* fuse_err = nullptr;
  int fuse_failed = 0;                 // a hand-off timed out: two launches from then on
  int fuse = 0, graph_fuse = 0;        // opt-in (ITTS_FUSE_QKV_ATTN=1 / debug bit 3): measured 1.5 % slower than two launches; 0 again after a hand-off timeout
  // persistent decode engine (decode_engine.hip): the step as one launch, <= 6 rows (beam rows included), bf16, no fp8
  unsigned long long* eng_gran = nullptr;
  unsigned* eng_ctr = nullptr;           // [0] step counter, [1] abort word
  int eng_off = 0;                       // debug bit 4 / ITTS_ENGINE=0: keep the five-launches-per-layer path (A/B)
  int eng_force = 0;                     // debug bit 5: use the engine whatever ITTS_ENGINE / the default says
  int eng_failed = 0;                    // a hand-off timed out: launch path from then on
  int graph_eng = 0;
  int graph_mode = 0;                    // last_mode of the captured step
  float typical_mass = 0.f, graph_typical = 0.f;  // TypicalLogitsWarper pre-pass (0 = off)
  float* scores2 = nullptr;                        // [cap_B][V] its output
  int* forced = nullptr;  // [cap_B][cap_gen] forced token per (row, step) or -1; allocated with ids
  int use_forced = 0, graph_forced = 0;
  int host_sample = 0, graph_host_sample = 0;  // the caller picks every token (itts_gpt_commit): the step stops behind the head
  int input_n = 0, graph_input_n = 0;  // forced steps that are HF input_tokens (positions k + 1 instead of k + 2)
  int last_mode = 0;                   // 1: the last captured / launched decode step ran on the persistent engine
  float graph_penalty = 0.f;
  bool active = false;
};

struct Engine {
  itts_config cfg;
  int adt = 0;  // activation / weight dtype
  size_t es = 4;
  std::unordered_map<std::string, Tensor> tensors;
  bool finalized = false;
  CondW cond;
  GptW gpt;
  BigvganW bv;
  EcapaW ec;
  DvaeW dv;
  DecodeState ds;
  bool use_graph = true;
  bool force_simple = false;

  // scratch workspace (bump allocator, two-pass: dry run sizes it, real run uses it)
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  bool dry = false;
  void ws_reset() { ws_off = 0; }
  void* alloc(size_t bytes) {
    const size_t a = (ws_off + 255) & ~size_t(255);
    ws_off = a + bytes;
    return dry ? (void*)(uintptr_t)(0x1000 + a) : (void*)(ws + a);
  }
  int ws_reserve(size_t bytes, hipStream_t s);
  // A second scratch arena for the speaker encoder: Engine::ecapa swaps it in for the duration of the call, so that the caller may
  // run it on a side stream beside the conditioning encoder / the prefill (both are chains of small kernels that leave most of the
  // chip idle; itts_hip/engine.py ecapa(overlap=True)).  K-split GEMMs are off inside the swap (one split workspace per engine).
  char* ws_b = nullptr;
  size_t ws_b_cap = 0;
  bool ksplit_off = false;
  struct ArenaSwap {
    Engine& e;
    explicit ArenaSwap(Engine& en) : e(en) { flip(); e.ksplit_off = true; }
    ~ArenaSwap() { flip(); e.ksplit_off = false; }
    void flip() {
      std::swap(e.ws, e.ws_b);
      std::swap(e.ws_cap, e.ws_b_cap);
    }
  };
  // raw partial sums of K-split GEMMs (Engine::conv): [split][M][N] fp32, allocated at the first split launch
  static constexpr size_t KSPLIT_WS_BYTES = size_t(64) << 20;
  float* ksplit_ws = nullptr;

  // debug taps
  bool debug = false;
  std::map<std::string, std::vector<float>> taps;
  int tap(const char* name, const void* p, int dt, int64_t n, hipStream_t s);

  void* gpt_tiles = nullptr;  // one allocation: Lin::wt of every GPT projection + head (batched decode, bf16)
  int ensure_decode_tiles(hipStream_t s);
  ~Engine();

  // op wrappers (skip launches in dry mode)
  int lin(void* C, int tc, const void* A, int ta, int lda, const Lin& w, int M, int ldc, hipStream_t s, int act = ACT_NONE,
          const void* R = nullptr, int ldr = 0, float alpha = 1.f);
  int conv(GemmArgs& g, int ta, int tw, int tc, hipStream_t s);
  int ln(void* y, int ty, const void* x, int tx, const Norm& n, int rows, int D, hipStream_t s, int act = ACT_NONE,
         float eps = 1e-5f);

  // model entry points
  int finalize();
  int conditioning(const void* mel, int F, float* cond_out, hipStream_t s, int F_total = 0);
  int ecapa(const void* mel, int B, int F, float* spk_out, hipStream_t s);
  int gpt_prefill(const float* cond, const int32_t* text_ids, int B, int L, int max_gen, float penalty, int suppress,
                  hipStream_t s);
  int gpt_decode(int nsteps, hipStream_t s);
  int gpt_decode_steps(int nsteps, hipStream_t s);
  int gpt_set_sampling(int do_sample, int top_k, float top_p, float temperature, const float* uniforms_host, long n);
  std::vector<float> sample_uniforms;  // host copy, uploaded by the next prefill
  int gpt_set_forced(const int32_t* ids_host, int B, int n);
  int gpt_set_input_tokens(const int32_t* ids_host, int B, int n);
  int gpt_set_host_sampling(int on);
  int cond_per_row = 0;  // itts_gpt_set_cond_per_row: the cond passed to prefill holds one [latents, D] block per batch item
  int gpt_commit(const int32_t* tokens_host, hipStream_t s);
  int gpt_beam_state(int32_t* ids_host, float* scores_host, int32_t* done_host, int* step_host, hipStream_t s);
  int gpt_commit_beams(const float* pick_score_host, const int32_t* pick_tok_host, const int32_t* pick_beam_host, hipStream_t s);
  BeamArgs beam_args(const float* lg_in, bool typical) const;
  int forced_input = 0;  // the forced tokens are HF `input_tokens`: token k sits at mel position k + 1 (model.py:141-144)
  int gpt_set_typical(float mass);
  int gpt_set_beam_sample(int num_beams, int top_k, float top_p, float temperature, const float* uniforms_host, long n);
  int gpt_set_beams(int num_beams, int do_sample, int top_k, float top_p, float temperature, float length_penalty,
                    const float* uniforms_host, long n);
  int gpt_set_beam_returns(int n);
  int beam_do_sample = 1;
  float beam_length_penalty = 0.f;
  int gen_epoch = 0;   // generation counter: high bits of the hand-off tags (never 0)
  int beam_beams = 1;  // requested beams for the following generations (1 = off)
  int beam_returns = 1;  // hypotheses returned per batch item (generate()'s num_return_sequences = num_beam_hyps_to_keep)
  int ensure_beam_state(int rows, int max_gen, int Smax, hipStream_t s);
  int beam_finalize(int32_t* codes_host, hipStream_t s);
  std::vector<int32_t> forced_host;  // [forced_B][forced_n], uploaded by the next prefill
  int forced_B = 0, forced_n = 0;
  int gpt_status(int* steps, int* n_unf, hipStream_t s);
  int gpt_fetch(int32_t* codes, float* logits, hipStream_t s);
  int gpt_latent(const float* cond, const int32_t* text_ids, int L, const int32_t* codes, int T, void* latent_out,
                 hipStream_t s);
  int gpt_latent_batch(const float* cond, const int32_t* text_ids, const int* Ls, const int32_t* codes, const int* Ts,
                       int nseq, void* latent_out, hipStream_t s);
  int bigvgan(const void* latent, const float* spk, int B, int T, float* wav, hipStream_t s);
  int dvae_decode(const int32_t* codes, int B, int T, void* mel_out, hipStream_t s);
  int dvae_encode(const void* mel, int B, int T, int32_t* codes_host, hipStream_t s);

  // internals
  int gpt_layers_full(float* h, int B, int S, const int* kv_start_dev, bool write_cache, hipStream_t s);
  int decode_step_launch(hipStream_t s);
  bool engine_usable() const;
  int ensure_engine_state(hipStream_t s);
  int engine_check(hipStream_t s);
  int head_and_sample(hipStream_t s, bool have_logits = false, bool sampled = false);
  int sample_from_logits(hipStream_t s, bool sampled = false);
  SamplerArgs greedy_sampler_args(const float* lg_in, bool typical) const;
  int ensure_decode_state(int B, int Smax, int max_gen, hipStream_t s);
  template <typename F>
  int two_pass(F&& body, hipStream_t s) {
    dry = true;
    ws_reset();
    int st = body();
    dry = false;
    if (st != OK) return st;
    ITTS_TRY(ws_reserve(ws_off + 4096, s));
    ws_reset();
    return body();
  }
};

}  // namespace itts
