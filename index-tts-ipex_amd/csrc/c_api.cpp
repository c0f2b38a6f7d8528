// extern "C" boundary of libitts_hip (declared in include/itts_hip.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#include "engine.h"

using namespace itts;

struct itts_engine {
  Engine e;
};

namespace itts {
// which kernel family the dispatcher takes for a shape: 0 vector ALU, 1 register-staged MFMA (gemm_mfma), 2 LDS-DMA staged 128-wide
// tiles (gemm_glds), 3 256 x 256 eight-phase (gemm_p8), 4 LDS-tiled narrow conv (conv_lds)
// K split for the few-tile, deep-K shapes (batch-1 latent pass: 1242 rows x 1280 features over K = 5120 is 25 tiles of 256 x 256 - a
// tenth of the CUs, each MFMA-bound for 70 us; BigVGAN conv_pre and stage 0 likewise): the tile's K-tiles go to S workgroups that
// write raw fp32 sums to the caller's workspace, a second launch adds them in split order (deterministic) and runs the epilogue.
// Returns S (1 = no split).  ITTS_GEMM_KSPLIT=0 turns it off, =n forces n where the shape is eligible (A/B, read per call).
int gemm_ksplit_plan(const GemmArgs& g, int ta, int tw, int tc, size_t ws_bytes) {
  const char* e = getenv("ITTS_GEMM_KSPLIT");
  const int forced = e ? atoi(e) : -1;
  if (forced == 0 || g.nphase != 1 || getenv("ITTS_GEMM_FORCE_OLD") || getenv("ITTS_NO_GEMM_GLDS")) return 1;
  const char* p8e = getenv("ITTS_GEMM_P8");
  if (p8e && atoi(p8e) == 0) return 1;
  if (!getenv("ITTS_NO_CONV_LDS") && conv_lds_supported(g, ta, tw, tc)) return 1;
  const long tiles = gemm_p8_tiles(g, ta, tw, tc);
  const long nk = (long)g.taps * (g.Cin / 64);
  // measured (tools/bench_gemm.py --batch 1 --ksplit, profiles/r04_gemm_ksplit_b1.txt): pays below 64 tiles with K >= 2048 (conv_pre
  // 126 -> 50 us, stage-0 k = 11 conv 146 -> 67, latent mlp.c_proj 70 -> 41); at 75 - 100 tiles or K = 1280 the reduction launch
  // costs more than the idle CUs did (c_attn 34 -> 36, c_proj 23 -> 27)
  if (tiles <= 0 || tiles >= (forced > 0 ? 128 : 64) || nk < (forced > 0 ? 16 : 32)) return 1;
  long S = forced > 0 ? forced : 224 / tiles;
  S = std::min(S, 8L);
  S = std::min(S, nk / 4);                                            // at least four K-tiles per split (the pipeline's fill)
  S = std::min(S, (long)(ws_bytes / ((size_t)g.M * g.N * 4)));
  while (S > 1 && (S - 1) * ((nk + S - 1) / S) >= nk) --S;            // every split owns at least one K-tile
  return S < 2 ? 1 : (int)S;
}

int gemm_which(const GemmArgs& g, int ta, int tw, int tc) {
  if (g.ksplit > 1) return 3;  // planned by gemm_ksplit_plan (Engine::conv): gemm_p8 with its reduction launch
  static const bool no_conv_lds = getenv("ITTS_NO_CONV_LDS") != nullptr;
  if (!no_conv_lds && conv_lds_supported(g, ta, tw, tc)) return 4;
  static const bool no_glds = getenv("ITTS_NO_GEMM_GLDS") != nullptr;  // A/B switch: the register-staged kernel everywhere
  const bool old = no_glds || getenv("ITTS_GEMM_FORCE_OLD") != nullptr;  // (per call: the parity test runs both on one shape)
  // ITTS_GEMM_P8=0: without the 256 x 256 eight-phase kernel (A/B, read per call)
  const char* p8e = getenv("ITTS_GEMM_P8");
  const long p8_tiles = (!old && !(p8e && atoi(p8e) == 0)) ? gemm_p8_tiles(g, ta, tw, tc) : 0;
  if (p8_tiles >= 200) return 3;
  if (!old && gemm_glds_supported(g, ta, tw, tc)) return 2;
  if (p8_tiles >= 64) return 3;  // a quarter of the CUs busy with the deep pipeline still beats the register-staged kernel
  if (gemm_mfma_supported(g, ta, tw, tc)) return 1;
  return 0;
}
int gemm(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s) {
  switch (gemm_which(g, ta, tw, tc)) {
    case 4: return conv_lds(g, s);
    case 3: return gemm_p8(g, ta, tw, tc, s);
    case 2: return gemm_glds(g, ta, tw, tc, s);
    case 1: return gemm_mfma(g, ta, tw, tc, s);
    default: return gemm_simple(g, ta, tw, tc, s);
  }
}
}  // namespace itts

extern "C" {

const char* itts_last_error(void) { return last_error(); }
int itts_abi_version(void) { return 4; }
int itts_half_is_f16(void) {
#ifdef ITTS_HALF_F16
  return 1;
#else
  return 0;
#endif
}

int itts_snake_aa_fwd(void* dst, const void* src, const float* up12, const float* down12, const float* log_alpha,
                      const float* log_beta, int B, int C, int T, int dtype, int layout, itts_stream stream) {
  (void)hipGetLastError();  // drop stale errors left by other HIP users (torch)
  if (layout == 1) return snake_aa(dst, src, log_alpha, log_beta, up12, down12, B, T, C, dtype, (hipStream_t)stream);
  if (layout == 0) return snake_aa_bct(dst, src, log_alpha, log_beta, up12, down12, B, C, T, dtype, (hipStream_t)stream);
  set_error("itts_snake_aa_fwd: layout must be 0 ([B,C,T]) or 1 ([B,T,C])");
  return E_INVALID;
}

static void to_gemm_args(const itts_gemm_args* a, GemmArgs& g) {
  g.A = a->A; g.W = a->W; g.C = a->C;
  g.M = a->M; g.N = a->N; g.Cin = a->Cin; g.taps = a->taps; g.lda = a->lda; g.ldc = a->ldc; g.T = a->T;
  g.dil = a->dil; g.pad_left = a->pad_left; g.pad_mode = a->pad_mode; g.in_up = a->in_up < 1 ? 1 : a->in_up;
  g.nphase = a->nphase < 1 ? 1 : a->nphase;
  for (int i = 0; i < 8; ++i) g.phase_shift[i] = a->phase_shift[i];
  g.bias = a->bias; g.bias_bstride = a->bias_bstride; g.act = a->act; g.scale = a->scale; g.shift = a->shift;
  g.act2 = a->act2; g.R = a->R; g.ldr = a->ldr; g.alpha = a->alpha; g.ADD = a->ADD; g.ldadd = a->ldadd; g.beta = a->beta;
}

int itts_gemm(const itts_gemm_args* a, itts_stream stream) {
  (void)hipGetLastError();  // drop stale errors left by other HIP users (torch)
  if (!a) {
    set_error("itts_gemm: null args");
    return E_INVALID;
  }
  GemmArgs g;
  to_gemm_args(a, g);
  if (a->force_simple) return gemm_simple(g, a->dtype_a, a->dtype_w, a->dtype_c, (hipStream_t)stream);
  return gemm(g, a->dtype_a, a->dtype_w, a->dtype_c, (hipStream_t)stream);
}

int itts_gemm_ws(const itts_gemm_args* a, void* ws, size_t ws_bytes, itts_stream stream) {
  (void)hipGetLastError();
  if (!a) {
    set_error("itts_gemm_ws: null args");
    return E_INVALID;
  }
  GemmArgs g;
  to_gemm_args(a, g);
  if (a->force_simple) return gemm_simple(g, a->dtype_a, a->dtype_w, a->dtype_c, (hipStream_t)stream);
  if (ws && !((uintptr_t)ws & 15)) {
    const int S = gemm_ksplit_plan(g, a->dtype_a, a->dtype_w, a->dtype_c, ws_bytes);
    if (S > 1) {
      g.ws = (float*)ws;
      g.ksplit = S;
    }
  }
  return gemm(g, a->dtype_a, a->dtype_w, a->dtype_c, (hipStream_t)stream);
}

int itts_gemm_ksplit(const itts_gemm_args* a, size_t ws_bytes) {
  if (!a) return E_INVALID;
  GemmArgs g;
  to_gemm_args(a, g);
  return a->force_simple ? 1 : gemm_ksplit_plan(g, a->dtype_a, a->dtype_w, a->dtype_c, ws_bytes);
}

int itts_gemm_which(const itts_gemm_args* a) {
  if (!a) return E_INVALID;
  GemmArgs g;
  to_gemm_args(a, g);
  return a->force_simple ? 0 : gemm_which(g, a->dtype_a, a->dtype_w, a->dtype_c);
}

int itts_layernorm(void* y, int dtype_y, const void* x, int dtype_x, const float* gamma, const float* beta, int rows,
                   int D, float eps, itts_stream stream) {
  (void)hipGetLastError();  // drop stale errors left by other HIP users (torch)
  return layernorm(y, dtype_y, x, dtype_x, gamma, beta, rows, D, D, D, eps, ACT_NONE, (hipStream_t)stream);
}

int itts_attention(void* o, const void* q, const void* k, const void* v, int B, int H, int Sq, int Sk, int dqk, int dv,
                   int ldq, int ldk, int ldv, int ldo, float scale, int causal, const int* kv_start, int dtype,
                   itts_stream stream) {
  (void)hipGetLastError();  // drop stale errors left by other HIP users (torch)
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = o; a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.dqk = dqk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.scale = scale; a.causal = causal; a.kv_start = kv_start;
  return attention(a, dtype, (hipStream_t)stream);
}

int itts_gemv(float* Y, const float* X, const void* W, const float* bias, int B, int N, int K, int act, int accumulate,
              int prologue, const float* ln_gamma, const float* ln_beta, const float* ln2_gamma, const float* ln2_beta,
              int dtype_w, int version, itts_stream stream) {
  (void)hipGetLastError();
  GemvArgs g;
  g.X = X; g.W = W; g.Y = Y; g.bias = bias; g.B = B; g.N = N; g.K = K; g.ldy = N; g.act = act; g.accumulate = accumulate;
  g.prologue = prologue; g.ln_gamma = ln_gamma; g.ln_beta = ln_beta; g.ln2_gamma = ln2_gamma; g.ln2_beta = ln2_beta;
  if (version == 2 || (version == 0 && gemv2_supported(g))) return gemv2(g, dtype_w, (hipStream_t)stream);
  return gemv(g, dtype_w, (hipStream_t)stream);
}

int itts_skinny_gemm(void* Y, int y_bf16, const void* X, const void* W, const float* bias, int B, int N, int K, int act,
                     int accumulate, int ksplit, float* partial, int layout, itts_stream stream) {
  (void)hipGetLastError();
  GemvArgs g;
  g.X = (const float*)X; g.x_bf16 = 1; g.W = W; g.Y = (float*)Y; g.y_bf16 = y_bf16; g.bias = bias; g.B = B; g.N = N; g.K = K;
  g.ldy = N; g.act = act; g.accumulate = accumulate; g.ksplit = ksplit; g.partial = partial;
  g.x_tiled = layout & 1; g.y_tiled = (layout >> 1) & 1;
  if (layout & 4) {
    g.Wt = W;
    g.W = nullptr;
  }
  if (layout & 8) {  // X is fp32 [B, K]: LayerNorm (eps 1e-5, no affine) in the prologue, B <= 16
    g.x_bf16 = 0;
    g.prologue = 1;
  }
  g.half_tiles = (layout >> 4) & 1;
  return skinny_mfma(g, (hipStream_t)stream);
}

int itts_retile_weights(void* dst, const void* src, int N, int K, itts_stream stream) {
  (void)hipGetLastError();
  return retile_weights_bf16(dst, src, N, K, (hipStream_t)stream);
}

int itts_ln_rows_bf16(void* y, float* x, const float* gamma, const float* beta, int rows, int D, float eps, int passes,
                      const float* partial, int nsplit, const float* partial_bias, int y_tiled, itts_stream stream) {
  (void)hipGetLastError();
  if (!y || !x || rows <= 0) {
    set_error("itts_ln_rows_bf16: bad arguments");
    return E_INVALID;
  }
  return ln_rows_bf16(y, x, gamma, beta, rows, D, eps, passes, partial, nsplit, partial_bias, y_tiled, (hipStream_t)stream);
}

int itts_transpose(void* y, const void* x, int B, int R, int C, int dtype, itts_stream stream) {
  (void)hipGetLastError();  // drop stale errors left by other HIP users (torch)
  return transpose_brc(y, x, B, R, C, dtype, (hipStream_t)stream);
}

int itts_engine_create(const itts_config* cfg, itts_engine** out) {
  if (!cfg || !out) {
    set_error("itts_engine_create: null argument");
    return E_INVALID;
  }
  if (cfg->dtype != F32 && cfg->dtype != BF16) {
    set_error("itts_engine_create: dtype must be ITTS_F32 or ITTS_BF16");
    return E_INVALID;
  }
  if (cfg->model_dim <= 0 || cfg->heads <= 0 || cfg->model_dim % cfg->heads != 0 || cfg->model_dim / cfg->heads != 64) {
    set_error("itts_engine_create: model_dim/heads must give head_dim 64");
    return E_INVALID;
  }
  if (cfg->bv_num_up > 8 || cfg->bv_num_res > 4 || cfg->bv_num_dil > 4) {
    set_error("itts_engine_create: vocoder topology out of range");
    return E_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    (void)hipGetLastError();
    set_error("itts_engine_create: no HIP device (this library has no CPU fallback)");
    return E_HIP;
  }
  itts_engine* e = new (std::nothrow) itts_engine();
  if (!e) return E_NOMEM;
  e->e.cfg = *cfg;
  e->e.adt = cfg->dtype;
  e->e.es = dtype_size(cfg->dtype);
  *out = e;
  return OK;
}

void itts_engine_destroy(itts_engine* e) { delete e; }

int itts_engine_bind_tensor(itts_engine* e, const char* name, const void* ptr, int dtype, int ndim, const int64_t* dims) {
  if (!e || !name || !ptr || ndim < 1 || ndim > 4 || !dims) {
    set_error("itts_engine_bind_tensor: bad argument");
    return E_INVALID;
  }
#ifdef ITTS_HALF_F16
  if (dtype == FP8) {  // the fp8 readers expand to bf16 pairs (v_cvt_scalef32_pk_bf16_fp8): BASELINE config 5 is a bf16-build mode
    set_error("itts_engine_bind_tensor: fp8 weight copies are not supported by the f16 build of the library");
    return E_INVALID;
  }
#endif
  Tensor t;
  t.p = ptr;
  t.dt = dtype;
  t.nd = ndim;
  for (int i = 0; i < ndim; ++i) t.d[i] = dims[i];
  e->e.tensors[name] = t;
  e->e.finalized = false;
  return OK;
}

int itts_engine_finalize(itts_engine* e) { return e ? e->e.finalize() : E_INVALID; }

#define ENG(e)                           \
  (void)hipGetLastError();               \
  if (!(e)) {                            \
    set_error("null engine");            \
    return E_INVALID;                    \
  }

int itts_conditioning(itts_engine* e, const void* mel, int F, float* cond_out, itts_stream s) {
  ENG(e);
  return e->e.conditioning(mel, F, cond_out, (hipStream_t)s);
}
int itts_conditioning_padded(itts_engine* e, const void* mel, int F, int F_total, float* cond_out, itts_stream s) {
  ENG(e);
  return e->e.conditioning(mel, F, cond_out, (hipStream_t)s, F_total);
}
int itts_ecapa(itts_engine* e, const void* mel, int B, int F, float* spk_out, itts_stream s) {
  ENG(e);
  return e->e.ecapa(mel, B, F, spk_out, (hipStream_t)s);
}
int itts_gpt_prefill(itts_engine* e, const float* cond, const int32_t* text_ids_host, int B, int L, int max_gen,
                     float repetition_penalty, int suppress_stop, itts_stream s) {
  ENG(e);
  return e->e.gpt_prefill(cond, text_ids_host, B, L, max_gen, repetition_penalty, suppress_stop, (hipStream_t)s);
}
int itts_gpt_set_sampling(itts_engine* e, int do_sample, int top_k, float top_p, float temperature, const float* uniforms_host,
                          int64_t n_uniforms) {
  ENG(e);
  return e->e.gpt_set_sampling(do_sample, top_k, top_p, temperature, uniforms_host, (long)n_uniforms);
}
int itts_gpt_set_beam_sample(itts_engine* e, int num_beams, int top_k, float top_p, float temperature, const float* uniforms_host,
                             int64_t n_uniforms) {
  ENG(e);
  return e->e.gpt_set_beam_sample(num_beams, top_k, top_p, temperature, uniforms_host, (long)n_uniforms);
}
int itts_gpt_set_beams(itts_engine* e, int num_beams, int do_sample, int top_k, float top_p, float temperature, float length_penalty,
                       const float* uniforms_host, int64_t n_uniforms) {
  ENG(e);
  return e->e.gpt_set_beams(num_beams, do_sample, top_k, top_p, temperature, length_penalty, uniforms_host, (long)n_uniforms);
}
int itts_gpt_set_beam_returns(itts_engine* e, int num_return_sequences) {
  ENG(e);
  return e->e.gpt_set_beam_returns(num_return_sequences);
}
int itts_gpt_set_typical(itts_engine* e, float mass) {
  ENG(e);
  return e->e.gpt_set_typical(mass);
}
int itts_gpt_set_input_tokens(itts_engine* e, const int32_t* ids_host, int B, int n) {
  ENG(e);
  return e->e.gpt_set_input_tokens(ids_host, B, n);
}
int itts_gpt_decode_mode(itts_engine* e) { return e ? e->e.ds.last_mode : -1; }
int itts_gpt_set_cond_per_row(itts_engine* e, int on) {
  ENG(e);
  e->e.cond_per_row = on ? 1 : 0;
  return OK;
}
int itts_gpt_set_host_sampling(itts_engine* e, int on) {
  ENG(e);
  return e->e.gpt_set_host_sampling(on);
}
int itts_gpt_commit(itts_engine* e, const int32_t* tokens_host, itts_stream s) {
  ENG(e);
  return e->e.gpt_commit(tokens_host, (hipStream_t)s);
}
int itts_gpt_beam_state(itts_engine* e, int32_t* ids_host, float* scores_host, int32_t* done_host, int* step_host, itts_stream s) {
  ENG(e);
  return e->e.gpt_beam_state(ids_host, scores_host, done_host, step_host, (hipStream_t)s);
}
int itts_gpt_commit_beams(itts_engine* e, const float* pick_score_host, const int32_t* pick_tok_host, const int32_t* pick_beam_host,
                          itts_stream s) {
  ENG(e);
  return e->e.gpt_commit_beams(pick_score_host, pick_tok_host, pick_beam_host, (hipStream_t)s);
}
int itts_gpt_set_forced(itts_engine* e, const int32_t* ids_host, int B, int n) {
  ENG(e);
  return e->e.gpt_set_forced(ids_host, B, n);
}
int itts_gpt_decode(itts_engine* e, int nsteps, itts_stream s) {
  ENG(e);
  return e->e.gpt_decode(nsteps, (hipStream_t)s);
}
int itts_gpt_status(itts_engine* e, int* steps, int* n_unf, itts_stream s) {
  ENG(e);
  return e->e.gpt_status(steps, n_unf, (hipStream_t)s);
}
int itts_gpt_fetch(itts_engine* e, int32_t* codes, float* logits, itts_stream s) {
  ENG(e);
  return e->e.gpt_fetch(codes, logits, (hipStream_t)s);
}
int itts_gpt_latent(itts_engine* e, const float* cond, const int32_t* text_ids_host, int L, const int32_t* codes_host,
                    int T, void* latent_out, itts_stream s) {
  ENG(e);
  return e->e.gpt_latent(cond, text_ids_host, L, codes_host, T, latent_out, (hipStream_t)s);
}
int itts_gpt_latent_batch(itts_engine* e, const float* cond, const int32_t* text_ids_host, const int32_t* text_lens_host,
                          const int32_t* codes_host, const int32_t* code_lens_host, int nseq, void* latent_out,
                          itts_stream s) {
  ENG(e);
  return e->e.gpt_latent_batch(cond, text_ids_host, text_lens_host, codes_host, code_lens_host, nseq, latent_out,
                               (hipStream_t)s);
}
int itts_bigvgan(itts_engine* e, const void* latent, const float* spk, int B, int T, float* wav_out, itts_stream s) {
  ENG(e);
  return e->e.bigvgan(latent, spk, B, T, wav_out, (hipStream_t)s);
}
int itts_dvae_decode(itts_engine* e, const int32_t* codes_host, int B, int T, void* mel_out, itts_stream s) {
  ENG(e);
  return e->e.dvae_decode(codes_host, B, T, mel_out, (hipStream_t)s);
}

int itts_dvae_encode(itts_engine* e, const void* mel_btc, int B, int T, int32_t* codes_host, itts_stream s) {
  ENG(e);
  return e->e.dvae_encode(mel_btc, B, T, codes_host, (hipStream_t)s);
}

int itts_debug_enable(itts_engine* e, int on) {
  ENG(e);
  e->e.debug = on & 1;
  e->e.force_simple = (on & 2) != 0;
  e->e.use_graph = (on & 4) == 0;
  e->e.ds.fuse = (on & 8) != 0;  // bit 3: the fused projection + attention launch instead of two launches (A/B, parity tests)
  e->e.ds.eng_off = (on & 16) != 0;  // bit 4: the launch path instead of the persistent decode engine (A/B, parity tests)
  e->e.ds.eng_force = (on & 32) != 0;  // bit 5: the persistent decode engine whatever ITTS_ENGINE says
  return OK;
}

int64_t itts_debug_fetch(itts_engine* e, const char* name, float* out_host, int64_t max_elems) {
  if (!e || !name) return -1;
  auto it = e->e.taps.find(name);
  if (it == e->e.taps.end()) return -1;
  const int64_t n = (int64_t)it->second.size();
  if (out_host) std::memcpy(out_host, it->second.data(), (size_t)std::min(n, max_elems) * 4);
  return n;
}

}  // extern "C"
