// GPT-2 acoustic LM: prefix assembly + prefill (step 0), per-token decode (hipGraph-captured), latent pass.
// Reference: UnifiedVoice.inference_speech / prepare_gpt_inputs / forward(return_latent) and
// GPT2InferenceModel.forward (indextts/gpt/model.py:115-192,521-708); HF transformers 4.36.2 GPT2Model /
// greedy_search semantics (not vendored; restated in oracle/gpt.py and pinned by tests/golden).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <mutex>

#include <cstdlib>

#include "engine.h"

namespace itts {

namespace {

// row descriptor for the embedding assembly: kind 0 zero, 1 cond row, 2 text (emb+pos), 3 mel (emb+pos)
struct RowDesc {
  int kind, a, b, pad;
};

template <typename TW>
__global__ void gpt_embed_rows_kernel(float* __restrict__ h, const RowDesc* __restrict__ rd, const float* __restrict__ cond,
                                      const TW* __restrict__ text_emb, const TW* __restrict__ text_pos,
                                      const TW* __restrict__ mel_emb, const TW* __restrict__ mel_pos, int D) {
  const int r = blockIdx.x;
  const RowDesc d = rd[r];
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float v = 0.f;
    if (d.kind == 1) v = cond[(size_t)d.a * D + i];
    else if (d.kind == 2) v = ldf(text_emb + (size_t)d.a * D + i) + ldf(text_pos + (size_t)d.b * D + i);
    else if (d.kind == 3) v = ldf(mel_emb + (size_t)d.a * D + i) + ldf(mel_pos + (size_t)d.b * D + i);
    h[(size_t)r * D + i] = v;
  }
}

__global__ void init_decode_state_kernel(uint8_t* seen, int* unfinished, int* cur_tok, int* len, int* prefix_dev,
                                         int B, int V, int fake_id, int start_tok, int prefix, int epoch) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)B * V) {
    const int v = (int)(i % V);
    seen[i] = (v == fake_id || v == start_tok) ? 1 : 0;
  }
  if (i < B) {
    unfinished[i] = 1;
    cur_tok[i] = start_tok;
    len[i] = 0;
  }
  if (i == 0) {
    prefix_dev[0] = prefix;
    prefix_dev[1] = epoch;  // generation epoch: the high bits of the in-launch hand-off tags (qkv_attn_fused)
  }
}

// beam-sample state of a fresh generation: identity cache ancestry (every beam row reads its own prefix copy), zero
// running scores (beam_sample starts all beams at 0, unlike beam_search), no finished hypotheses
__global__ void beam_init_kernel(uint8_t* anc, float* beam_scores, int* hyp_order, int* hyp_n, float* hyp_worst,
                                 int* hyp_counter, int* done, int rows, int nb, int Smax, int do_sample) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)rows * Smax;
  if (i < 2 * n) anc[i] = (uint8_t)(((i % n) / Smax) % nb);
  // beam_sample starts every beam at 0; beam_search starts beams 1.. at -1e9 so that step 0 expands beam 0 only
  if (i < rows) beam_scores[i] = (do_sample || i % nb == 0) ? 0.f : -1e9f;
  const int B = rows / nb;
  if (i < (long)B * (nb + 1)) hyp_order[i] = -1;
  if (i < B) {
    hyp_n[i] = 0;
    hyp_worst[i] = 1e9f;
    hyp_counter[i] = 0;
    done[i] = 0;
  }
}

}  // namespace

#define K(call)               \
  do {                        \
    if (!dry) ITTS_TRY(call); \
  } while (0)

// full-sequence pass of the 24 blocks over h fp32 [B*S, D] (in place); optional KV-cache fill
int Engine::gpt_layers_full(float* h, int B, int S, const int* kv_start_dev, bool write_cache, hipStream_t s) {
  const itts_config& c = cfg;
  const int D = c.model_dim, H = c.heads, dh = D / H, M = B * S;
  void* xn = alloc((size_t)M * D * es);
  void* qkv = alloc((size_t)M * 3 * D * es);
  void* ctx = alloc((size_t)M * D * es);
  void* act = alloc((size_t)M * 4 * D * es);
  for (int l = 0; l < c.layers; ++l) {
    const GptLayerW& L = gpt.layers[l];
    ITTS_TRY(ln(xn, adt, h, F32, L.ln1, M, D, s));
    ITTS_TRY(lin(qkv, adt, xn, adt, D, L.attn, M, 3 * D, s));
    if (write_cache) {
      const size_t lo = (size_t)l * ds.B * H * ds.Smax * dh * es;
      K(kv_scatter((char*)ds.kc + lo, (char*)ds.vc + lo, qkv, B, S, H, dh, ds.Smax, adt, adt, s));
    }
    AttnArgs a;
    a.q = qkv;
    a.k = (const char*)qkv + (size_t)D * es;
    a.v = (const char*)qkv + (size_t)2 * D * es;
    a.o = ctx;
    a.B = B;
    a.H = H;
    a.Sq = a.Sk = S;
    a.dqk = a.dv = dh;
    a.ldq = a.ldk = a.ldv = 3 * D;
    a.ldo = D;
    a.scale = 1.f / std::sqrt((float)dh);
    a.causal = 1;
    a.kv_start = kv_start_dev;
    K(force_simple ? attention_simple(a, adt, s) : attention(a, adt, s));
    ITTS_TRY(lin(h, F32, ctx, adt, D, L.proj, M, D, s, ACT_NONE, h, D));
    ITTS_TRY(ln(xn, adt, h, F32, L.ln2, M, D, s));
    ITTS_TRY(lin(act, adt, xn, adt, D, L.fc, M, 4 * D, s, ACT_GELU_NEW));
    ITTS_TRY(lin(h, F32, act, adt, 4 * D, L.proj2, M, D, s, ACT_NONE, h, D));
  }
  return OK;
}

static int dev_alloc(void** p, size_t bytes) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  if (hipMalloc(p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("device allocation of " + std::to_string(bytes) + " bytes failed");
    return E_NOMEM;
  }
  return OK;
}

// The batched decode step (B > 4, bf16) streams its weights from fragment-tiled copies (wtile_off): made once per loaded
// checkpoint, on the first batched generation (+ 2 bytes per GPT weight; ITTS_NO_TILED_W=1 keeps the row-major stream).
int Engine::ensure_decode_tiles(hipStream_t s) {
  static const bool off = getenv("ITTS_NO_TILED_W") != nullptr;
  if (gpt_tiles || off || !gpt.ok) return OK;
  std::vector<Lin*> lins;
  for (GptLayerW& L : gpt.layers)
    for (Lin* l : {&L.attn, &L.proj, &L.fc, &L.proj2}) lins.push_back(l);
  lins.push_back(&gpt.head);
  size_t total = 0;
  for (Lin* l : lins) {
    if (l->dt != BF16 || l->Cin % 32 != 0 || l->taps != 1) return OK;  // not a bf16 checkpoint: nothing to tile
    total += (size_t)((l->N + 15) / 16) * 16 * l->Cin * (l->w8 ? 3 : 2);
  }
  ITTS_TRY(dev_alloc(&gpt_tiles, total));
  size_t o = 0;
  for (Lin* l : lins) {
    const size_t elems = (size_t)((l->N + 15) / 16) * 16 * l->Cin;
    l->wt = (char*)gpt_tiles + o;
    ITTS_TRY(retile_weights_bf16(l->wt, l->w, l->N, l->Cin, s));
    o += elems * 2;
    if (l->w8) {  // BASELINE config 5: the fp8 bytes in the same order
      l->wt8 = (char*)gpt_tiles + o;
      ITTS_TRY(retile_weights_fp8(l->wt8, l->w8, l->N, l->Cin, s));
      o += elems;
    }
  }
  return OK;
}

// The persistent decode engine (decode_engine.hip) replaces the 120 per-layer launches of a step when the step is the
// launch-bound small-batch bf16 one it was built for: IndexTTS-1.5 GPT dims, <= 6 rows (beam rows included: cache ancestry), no fp8
// copies, a whole MI355X (256 CUs) to itself.  ITTS_ENGINE=0 (or debug bit 4) keeps the launch path.
bool Engine::engine_usable() const {
  const char* ev = getenv("ITTS_ENGINE");  // read per call: tests flip it inside one process
  const bool env_off = ev ? atoi(ev) == 0 : !ENG_DEFAULT_ON;
  // CU count of the device this call runs on (a process may drive engines on several ordinals, and a partitioned MI355X
  // exposes fewer CUs on some of them): cached per ordinal, the same key EngineGate uses
  static std::atomic<int> ncu_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  int ncu = ncu_of[dev & 63].load(std::memory_order_relaxed);
  if (ncu == 0) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = -1;
    ncu_of[dev & 63].store(ncu, std::memory_order_relaxed);
  }
  const itts_config& c = cfg;
  if ((env_off && !ds.eng_force) || ds.eng_off || ds.eng_failed || ncu != ENG_NCU) return false;
  if (ds.fuse && !ds.fuse_failed) return false;  // A/B switch of the fused projection + attention launch: a launch-path variant
  if (adt != BF16 || ds.B < 1 || ds.B > ENG_MAX_ROWS || ds.nb < 1 || ds.B % ds.nb != 0 || (ds.nb > 1 && !ds.anc)) return false;
  if (c.model_dim != ENG_D || c.heads != ENG_H || c.layers < 1 || c.layers > ENG_MAX_LAYERS || ds.Smax > 2048 || ds.Smax % 256 != 0) return false;
  for (const GptLayerW& L : gpt.layers)
    for (const Lin* l : {&L.attn, &L.proj, &L.fc, &L.proj2})
      if (l->dt != BF16 || l->w8 || !l->b || l->taps != 1) return false;
  return true;
}

int Engine::ensure_engine_state(hipStream_t s) {
  DecodeState& d = ds;
  if (d.eng_gran) return OK;
  const itts_config& c = cfg;
  for (int l = 0; l < c.layers; ++l) {
    const GptLayerW& L = gpt.layers[l];
    ITTS_REQUIRE(L.attn.N == 3 * ENG_D && L.attn.Cin == ENG_D && L.proj.N == ENG_D && L.proj.Cin == ENG_D &&
                     L.fc.N == 4 * ENG_D && L.fc.Cin == ENG_D && L.proj2.N == ENG_D && L.proj2.Cin == 4 * ENG_D,
                 "decode engine: projection shapes");
  }
  const size_t gb = eng_gran_count(c.layers) * 8;
  ITTS_TRY(dev_alloc((void**)&d.eng_gran, gb));
  ITTS_TRY(dev_alloc((void**)&d.eng_ctr, 64));
  const unsigned ctr0[16] = {1u, 0u};
  ITTS_HIP_CHECK(hipMemsetAsync(d.eng_gran, 0, gb, s));  // tag 0 is never issued
  ITTS_HIP_CHECK(hipMemcpyAsync(d.eng_ctr, ctr0, 64, hipMemcpyHostToDevice, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));  // ctr0 is a host stack buffer
  return OK;
}

int Engine::ensure_decode_state(int B, int Smax, int max_gen, hipStream_t s) {
  const itts_config& c = cfg;
  const int D = c.model_dim, H = c.heads, dh = D / H, V = c.number_mel_codes;
  DecodeState& d = ds;
  if (adt == BF16 && B > 4 && D % 32 == 0 && D <= 2048) ITTS_TRY(ensure_decode_tiles(s));
  const size_t need = (size_t)c.layers * B * H * Smax * dh * es;
  const bool regrow = B > d.cap_B || max_gen > d.cap_gen;
  if (need > d.cache_bytes || regrow || B != d.B || Smax != d.Smax) {
    ITTS_HIP_CHECK(hipStreamSynchronize(s));
    for (hipGraphExec_t* ge : {&d.graph, &d.graphK})
      if (*ge) {
        (void)hipGraphExecDestroy(*ge);
        *ge = nullptr;
      }
  }
  if (need > d.cache_bytes) {
    ITTS_TRY(dev_alloc(&d.kc, need));
    ITTS_TRY(dev_alloc(&d.vc, need));
    // zero-filled once: the decode attention multiplies rows past the sequence end by p = 0 instead of selecting them
    // away, so whatever such a row holds has to be finite
    ITTS_HIP_CHECK(hipMemsetAsync(d.kc, 0, need, s));
    ITTS_HIP_CHECK(hipMemsetAsync(d.vc, 0, need, s));
    d.cache_bytes = need;
  }
  if (regrow) {
    // rows rounded up to 16: the batched decode keeps its bf16 activations in 16-row MFMA tiles
    const int cb = (std::max(B, d.cap_B) + 15) / 16 * 16, cg = std::max(max_gen, d.cap_gen);
    ITTS_TRY(dev_alloc((void**)&d.h, (size_t)cb * D * 4));
    ITTS_TRY(dev_alloc((void**)&d.qkv, (size_t)cb * 3 * D * 4));
    ITTS_TRY(dev_alloc((void**)&d.ctx, (size_t)cb * D * 4));
    ITTS_TRY(dev_alloc((void**)&d.act, (size_t)cb * 4 * D * 4));
    ITTS_TRY(dev_alloc((void**)&d.hn, (size_t)cb * D * 4));
    ITTS_TRY(dev_alloc((void**)&d.partial, (size_t)8 * cb * D * 4));
    ITTS_TRY(dev_alloc((void**)&d.attn_o, (size_t)cb * D * ATTN_NSPLIT * 4));
    ITTS_TRY(dev_alloc((void**)&d.attn_ml, (size_t)cb * H * 2 * ATTN_NSPLIT * 4));
    ITTS_TRY(dev_alloc((void**)&d.logits, (size_t)cb * V * 4));
    ITTS_TRY(dev_alloc((void**)&d.scores2, (size_t)cb * V * 4));
    ITTS_TRY(dev_alloc((void**)&d.kv_start, (size_t)cb * 4));
    ITTS_TRY(dev_alloc((void**)&d.cur_tok, (size_t)cb * 4));
    ITTS_TRY(dev_alloc((void**)&d.unfinished, (size_t)cb * 4));
    ITTS_TRY(dev_alloc((void**)&d.ids, (size_t)cb * cg * 4));
    ITTS_TRY(dev_alloc((void**)&d.forced, (size_t)cb * cg * 4));
    {
      const size_t gb = (size_t)c.layers * 4 * 3 * D * 8;
      ITTS_TRY(dev_alloc((void**)&d.gran, gb));
      ITTS_HIP_CHECK(hipMemsetAsync(d.gran, 0, gb, s));  // tag 0 is never issued
      ITTS_TRY(dev_alloc((void**)&d.fuse_err, 64));
      ITTS_HIP_CHECK(hipMemsetAsync(d.fuse_err, 0, 64, s));
    }
    ITTS_TRY(dev_alloc((void**)&d.seen, (size_t)cb * V));
    ITTS_TRY(dev_alloc((void**)&d.len, (size_t)cb * 4));
    if (!d.prefix_dev) ITTS_TRY(dev_alloc((void**)&d.prefix_dev, 64));
    d.cap_B = cb;
    d.cap_gen = cg;
  }
  d.B = B;
  d.Smax = Smax;
  d.max_gen = max_gen;
  return OK;
}

int Engine::ensure_beam_state(int rows, int max_gen, int Smax, hipStream_t s) {
  DecodeState& d = ds;
  if (rows <= d.beam_rows && max_gen <= d.beam_gen && Smax <= d.beam_smax && d.beam_ids) return OK;
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  for (hipGraphExec_t* ge : {&d.graph, &d.graphK})  // captured nodes hold the old pointers
    if (*ge) {
      (void)hipGraphExecDestroy(*ge);
      *ge = nullptr;
    }
  const int r = std::max(rows, d.beam_rows), g = std::max(max_gen, d.beam_gen), sm = std::max(Smax, d.beam_smax);
  ITTS_TRY(dev_alloc((void**)&d.beam_ids, (size_t)2 * r * g * 4));
  ITTS_TRY(dev_alloc((void**)&d.anc, (size_t)2 * r * sm));
  ITTS_TRY(dev_alloc((void**)&d.beam_scores, (size_t)r * 4));
  ITTS_TRY(dev_alloc((void**)&d.cand_sc, (size_t)r * BEAM_MAX_CAND * 4));
  ITTS_TRY(dev_alloc((void**)&d.cand_tok, (size_t)r * BEAM_MAX_CAND * 4));
  ITTS_TRY(dev_alloc((void**)&d.cand_n, (size_t)r * 4));
  ITTS_TRY(dev_alloc((void**)&d.hyp_tok, (size_t)r * 2 * g * 4));  // [B][nb + 1][g] <= rows * 2 * g
  ITTS_TRY(dev_alloc((void**)&d.hyp_score, (size_t)r * 2 * 4));
  ITTS_TRY(dev_alloc((void**)&d.hyp_len, (size_t)r * 2 * 4));
  ITTS_TRY(dev_alloc((void**)&d.hyp_order, (size_t)r * 2 * 4));
  ITTS_TRY(dev_alloc((void**)&d.hyp_n, (size_t)r * 4));
  ITTS_TRY(dev_alloc((void**)&d.hyp_worst, (size_t)r * 4));
  ITTS_TRY(dev_alloc((void**)&d.hyp_counter, (size_t)r * 4));
  ITTS_TRY(dev_alloc((void**)&d.beam_done, (size_t)r * 4));
  d.beam_rows = r;
  d.beam_gen = g;
  d.beam_smax = sm;
  return OK;
}

// HF beam_sample configuration for the following generations (num_beams <= 1 switches it off): the generate() mode of
// the reference's default kwargs (infer.py:116-124).  uniforms_host: row-major [max_gen][B][2 * num_beams] draws in [0, 1).
// generate()'s num_return_sequences under beams: the n best hypotheses of every batch item (1 <= n <= num_beams, checked
// against the beam count when the generation starts)
int Engine::gpt_set_beam_returns(int n) {
  ITTS_REQUIRE(n >= 1 && n <= 10, "gpt_set_beam_returns: num_return_sequences must be in [1, 10]");
  beam_returns = n;
  return OK;
}

int Engine::gpt_set_beam_sample(int num_beams, int top_k, float top_p, float temperature, const float* uniforms_host, long n) {
  return gpt_set_beams(num_beams, 1, top_k, top_p, temperature, 0.f, uniforms_host, n);
}

// do_sample = 1: beam_sample as above; do_sample = 0: HF beam_search (deterministic: per step the 2 * num_beams best of
// log_softmax + repetition penalty + running beam score, no warpers, no uniforms).  length_penalty as BeamHypotheses
// uses it (score = sum_logprobs / generated_len ** length_penalty; the reference passes 0.0).
int Engine::gpt_set_beams(int num_beams, int do_sample, int top_k, float top_p, float temperature, float length_penalty,
                          const float* uniforms_host, long n) {
  if (num_beams <= 1) {
    beam_beams = 1;
    if (!ds.do_sample) sample_uniforms.clear();
    return OK;
  }
  ITTS_REQUIRE(num_beams <= 10, "gpt_set_beams: num_beams must be in [2, 10]");
  if (do_sample && !ds.host_sample) {  // (host sampling: the caller warps and draws itself, any top_k - gpt_commit_beams)
    ITTS_REQUIRE(top_k >= 1 && top_k <= 128, "gpt_set_beams: top_k must be in [1, 128] (wider: itts_gpt_set_host_sampling first)");
    ITTS_REQUIRE(top_p > 0.f && top_p <= 1.f && temperature > 0.f, "gpt_set_beams: need 0 < top_p <= 1 and temperature > 0");
    ITTS_REQUIRE(uniforms_host && n > 0, "gpt_set_beams: uniforms missing");
    sample_uniforms.assign(uniforms_host, uniforms_host + n);
  }
  beam_beams = num_beams;
  beam_do_sample = do_sample ? 1 : 0;
  beam_length_penalty = length_penalty;
  ds.do_sample = 0;
  ds.top_k = top_k;
  ds.top_p = top_p;
  ds.temperature = temperature;
  return OK;
}

// BeamSearchScorer.finalize (beam_search.py, 4.36.2) on the host: open beams of unfinished batch items join their
// hypotheses with the running scores, the beam_returns best hypotheses (num_beam_hyps_to_keep = generate()'s
// num_return_sequences, model.py:655,698-703; highest score first, the later one on ties) are returned per batch item as
// codes [B * beam_returns][max_gen] padded with the stop token (eos = pad = stop_mel_token, model.py:698-700).
int Engine::beam_finalize(int32_t* codes, hipStream_t s) {
  DecodeState& d = ds;
  const int nb = d.nb, rows = d.B, B = rows / nb, mg = d.max_gen;
  std::vector<int> len(rows), done(B), hn(B), hlen((size_t)B * (nb + 1)), hord((size_t)B * (nb + 1));
  std::vector<float> bscore(rows), hscore((size_t)B * (nb + 1));
  ITTS_HIP_CHECK(hipMemcpyAsync(len.data(), d.len, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(done.data(), d.beam_done, (size_t)B * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(bscore.data(), d.beam_scores, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(hscore.data(), d.hyp_score, hscore.size() * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(hlen.data(), d.hyp_len, hlen.size() * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(hord.data(), d.hyp_order, hord.size() * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  const int k = len[0];
  std::vector<int32_t> ids((size_t)rows * mg), htok((size_t)B * (nb + 1) * mg);
  ITTS_HIP_CHECK(hipMemcpyAsync(ids.data(), d.beam_ids + (size_t)(k & 1) * rows * mg, ids.size() * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(htok.data(), d.hyp_tok, htok.size() * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  for (int b = 0; b < B; ++b) {
    struct Hyp {
      float score;
      int order, len;
      const int32_t* tok;
    };
    std::vector<Hyp> hy;
    for (int q = 0; q <= nb; ++q) {
      const size_t i = (size_t)b * (nb + 1) + q;
      if (hord[i] >= 0) hy.push_back({hscore[i], hord[i], hlen[i], htok.data() + i * mg});
    }
    if (!done[b]) {
      // beam_hyp.add(final_tokens, final_score, generated_len) for every open beam, with BeamHypotheses.add's eviction
      std::sort(hy.begin(), hy.end(), [](const Hyp& x, const Hyp& y) { return x.order < y.order; });
      float worst = 1e9f;
      for (const Hyp& h : hy) worst = std::min(worst, h.score);
      int counter = 1 << 20;
      const float lpdiv = d.length_penalty == 0.f ? 1.f : std::pow((float)(k - d.input_n), d.length_penalty);  // generated_len = k - given tokens
      for (int q = 0; q < nb; ++q) {
        const int row = b * nb + q;
        const float score = bscore[row] / lpdiv;
        if ((int)hy.size() < nb || score > worst) {
          hy.push_back({score, counter++, k, ids.data() + (size_t)row * mg});
          if ((int)hy.size() > nb) {
            size_t lo = 0;
            for (size_t i = 1; i < hy.size(); ++i)
              if (hy[i].score < hy[lo].score || (hy[i].score == hy[lo].score && hy[i].order < hy[lo].order)) lo = i;
            hy.erase(hy.begin() + lo);
            worst = 1e9f;
            for (const Hyp& h : hy) worst = std::min(worst, h.score);
          } else {
            worst = std::min(worst, score);
          }
        }
      }
    }
    const int keep = beam_returns;
    ITTS_REQUIRE((int)hy.size() >= keep, "beam_finalize: fewer hypotheses than num_return_sequences for a batch item");
    for (int r = 0; r < keep; ++r) {
      size_t best = 0;  // sorted(key=score).pop(): highest score, the later insertion on ties
      for (size_t i = 1; i < hy.size(); ++i)
        if (hy[i].score > hy[best].score || (hy[i].score == hy[best].score && hy[i].order > hy[best].order)) best = i;
      int32_t* out = codes + ((size_t)b * keep + r) * mg;
      for (int i = 0; i < mg; ++i) out[i] = i < hy[best].len ? hy[best].tok[i] : cfg.stop_mel_token;
      hy.erase(hy.begin() + best);
    }
  }
  return OK;
}

int Engine::gpt_prefill(const float* cond_dev, const int32_t* text_ids_in, int B_items, int L, int max_gen, float penalty,
                        int suppress, hipStream_t s) {
  // beam-sample: every batch item becomes nb rows (HF _expand_inputs_for_generation: repeat_interleave)
  const int nbeam = beam_beams > 1 ? beam_beams : 1;
  const int B = B_items * nbeam;
  std::vector<int32_t> text_exp;
  const int32_t* text_ids = text_ids_in;
  if (nbeam > 1 && text_ids_in && B_items > 0 && L > 0) {
    text_exp.resize((size_t)B * L);
    for (int b = 0; b < B; ++b) std::memcpy(text_exp.data() + (size_t)b * L, text_ids_in + (size_t)(b / nbeam) * L, (size_t)L * 4);
    text_ids = text_exp.data();
  }
  if (!finalized || !gpt.ok) {
    set_error("gpt_prefill: GPT weights not bound");
    return E_STATE;
  }
  const itts_config& c = cfg;
  ITTS_REQUIRE(cond_dev && text_ids && B > 0 && L > 0 && max_gen > 0, "gpt_prefill: bad arguments");
  ITTS_REQUIRE(B <= c.max_batch, "gpt_prefill: batch larger than max_batch");
  ITTS_REQUIRE(L + 2 <= c.max_text_tokens + 2, "gpt_prefill: text longer than max_text_tokens");
  ITTS_REQUIRE(max_gen + 2 <= c.max_mel_tokens + 3, "gpt_prefill: max_gen exceeds the mel position table");
  const int D = c.model_dim, nl = c.cond_latents, V = c.number_mel_codes;
  const int sp = nl + L + 2;  // prefix length s (model.py:614)
  const int S0 = sp + 1;      // + start_mel_token
  const int Smax = (S0 + max_gen + 255) / 256 * 256;  // coarse, so one captured graph serves many L
  ITTS_REQUIRE(Smax <= 2048, "gpt_prefill: sequence too long for the decode attention kernel");
  ITTS_REQUIRE(max_gen < 4095, "gpt_prefill: max_gen must stay below 4095 (12-bit step field of the fused hand-off tags)");
  ITTS_TRY(ensure_decode_state(B, Smax, max_gen, s));
  ds.prefix = sp;
  ds.penalty = penalty;
  ds.suppress_stop = suppress;
  gen_epoch = (gen_epoch % 0xFFFFF) + 1;  // 1 .. 2^20 - 1
  ds.nb = nbeam;
  ds.beam_sample = beam_do_sample;
  ds.length_penalty = beam_length_penalty;
  if (nbeam > 1) {
    ITTS_REQUIRE(forced_n == 0 || forced_input, "gpt_prefill: teacher forcing (gpt_set_forced) is not supported together with beams; `input_tokens` is");
    for (size_t i = 0; i < forced_host.size() && forced_n > 0; ++i)
      ITTS_REQUIRE(forced_host[i] >= 0, "gpt_prefill: `input_tokens` under beams must not hold free (-1) entries");
    ITTS_REQUIRE(beam_returns <= nbeam, "`num_return_sequences` has to be smaller or equal to `num_beams`.");
    ITTS_TRY(ensure_beam_state(B, max_gen, Smax, s));
    const long n_init = std::max<long>(2L * B * Smax, 64);
    hipLaunchKernelGGL(beam_init_kernel, dim3((unsigned)((n_init + 255) / 256)), dim3(256), 0, s, ds.anc, ds.beam_scores,
                       ds.hyp_order, ds.hyp_n, ds.hyp_worst, ds.hyp_counter, ds.beam_done, B, nbeam, Smax, beam_do_sample);
    ITTS_HIP_CHECK(hipGetLastError());
  }
  if (engine_usable()) ITTS_TRY(ensure_engine_state(s));  // allocations must not happen inside the graph capture of the step
  if (ds.do_sample || (nbeam > 1 && beam_do_sample && !ds.host_sample)) {
    const size_t need_u = nbeam > 1 ? (size_t)max_gen * B_items * 2 * nbeam : (size_t)max_gen * B;
    ITTS_REQUIRE(sample_uniforms.size() >= need_u,
                 "gpt_prefill: sampling enabled but fewer uniforms than max_gen * B (* 2 * num_beams) were supplied");
    if (need_u > ds.uniforms_cap) {
      ITTS_HIP_CHECK(hipStreamSynchronize(s));
      for (hipGraphExec_t* ge : {&ds.graph, &ds.graphK})  // the captured samplers hold the old pointer
        if (*ge) {
          (void)hipGraphExecDestroy(*ge);
          *ge = nullptr;
        }
      ITTS_TRY(dev_alloc((void**)&ds.uniforms, need_u * 4));
      ds.uniforms_cap = need_u;
    }
    ITTS_HIP_CHECK(hipMemcpyAsync(ds.uniforms, sample_uniforms.data(), need_u * 4, hipMemcpyHostToDevice, s));
    ITTS_HIP_CHECK(hipStreamSynchronize(s));
  }
  ds.use_forced = forced_n > 0;
  ds.input_n = forced_input ? forced_n : 0;
  if (ds.host_sample) {
    ITTS_REQUIRE(forced_n == 0 && !ds.do_sample, "gpt_prefill: host sampling excludes forced tokens and the device sampler");
    if (nbeam == 1) ITTS_HIP_CHECK(hipMemsetAsync(ds.forced, 0xFF, (size_t)B * max_gen * 4, s));  // -1: nothing forced yet
  }
  if (ds.use_forced) {
    ITTS_REQUIRE(forced_B == B_items || forced_B == 1, "gpt_prefill: forced tokens were set for a different batch size");
    ITTS_REQUIRE(forced_n <= max_gen, "gpt_prefill: more forced tokens than max_gen");
    std::vector<int32_t> tab((size_t)B * max_gen, -1);
    for (int b = 0; b < B; ++b)
      for (int k = 0; k < forced_n; ++k) tab[(size_t)b * max_gen + k] = forced_host[(size_t)(forced_B == 1 ? 0 : b / nbeam) * forced_n + k];
    ITTS_HIP_CHECK(hipMemcpyAsync(ds.forced, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    ITTS_HIP_CHECK(hipStreamSynchronize(s));
  }
  // host: row descriptors (prepare_gpt_inputs, model.py:615-639)
  std::vector<RowDesc> rd((size_t)B * S0);
  std::vector<int> kvs(B);
  for (int b = 0; b < B; ++b) {
    std::vector<int> ids;
    for (int i = 0; i < L; ++i) {
      const int t = text_ids[(size_t)b * L + i];
      if (t != c.start_text_token && t != c.stop_text_token) {
        ITTS_REQUIRE(t >= 0 && t <= c.number_text_tokens, "gpt_prefill: text id out of range");
        ids.push_back(t);
      }
    }
    const int n = (int)ids.size();
    const int pad = L - n;
    kvs[b] = pad;
    RowDesc* r = rd.data() + (size_t)b * S0;
    int k = 0;
    for (int i = 0; i < pad; ++i) r[k++] = {0, 0, 0, 0};
    // conditioning latents: one set for every row (infer.py), or one per batch item (model.py:599-602 takes [b, 32, D])
    for (int i = 0; i < nl; ++i) r[k++] = {1, (cond_per_row ? (b / nbeam) * nl : 0) + i, 0, 0};
    r[k++] = {2, c.start_text_token, 0, 0};
    for (int i = 0; i < n; ++i) r[k++] = {2, ids[i], i + 1, 0};
    r[k++] = {2, c.stop_text_token, n + 1, 0};
    r[k++] = {3, c.start_mel_token, 0, 0};
  }
  auto body = [&]() -> int {
    RowDesc* rd_dev = (RowDesc*)alloc(rd.size() * sizeof(RowDesc));
    float* h = (float*)alloc((size_t)B * S0 * D * 4);
    if (!dry) {
      ITTS_HIP_CHECK(hipMemcpyAsync(rd_dev, rd.data(), rd.size() * sizeof(RowDesc), hipMemcpyHostToDevice, s));
      ITTS_HIP_CHECK(hipMemcpyAsync(ds.kv_start, kvs.data(), B * 4, hipMemcpyHostToDevice, s));
      const long n = (long)B * V;
      hipLaunchKernelGGL(init_decode_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ds.seen,
                         ds.unfinished, ds.cur_tok, ds.len, ds.prefix_dev, B, V, 1, c.start_mel_token, sp, gen_epoch);
      if (adt == F32)
        hipLaunchKernelGGL(gpt_embed_rows_kernel<float>, dim3(B * S0), dim3(256), 0, s, h, rd_dev, cond_dev,
                           (const float*)gpt.text_emb, (const float*)gpt.text_pos, (const float*)gpt.mel_emb,
                           (const float*)gpt.mel_pos, D);
      else
        hipLaunchKernelGGL(gpt_embed_rows_kernel<bf16_t>, dim3(B * S0), dim3(256), 0, s, h, rd_dev, cond_dev,
                           (const bf16_t*)gpt.text_emb, (const bf16_t*)gpt.text_pos, (const bf16_t*)gpt.mel_emb,
                           (const bf16_t*)gpt.mel_pos, D);
      ITTS_HIP_CHECK(hipGetLastError());
    }
    ITTS_TRY(tap("prefix_emb", h, F32, (int64_t)B * S0 * D, s));
    ITTS_TRY(gpt_layers_full(h, B, S0, ds.kv_start, true, s));
    // logits of the last position only (generate takes logits[:, -1, :])
    K(copy_rows(ds.h, D, h + (size_t)(S0 - 1) * D, S0 * D, B, D, F32, s));
    if (!dry) ITTS_TRY(head_and_sample(s));
    return OK;
  };
  ITTS_TRY(two_pass(body, s));
  // rd/kvs are host stack buffers read by async copies
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  ds.active = true;
  return OK;
}

// ln_f -> final_norm -> mel_head -> repetition penalty / argmax / bookkeeping (lm_head, model.py:48,180)
int Engine::head_and_sample(hipStream_t s, bool have_logits, bool sampled) {
  const itts_config& c = cfg;
  const int D = c.model_dim, V = c.number_mel_codes, B = ds.B;
  if (have_logits) return sample_from_logits(s, sampled);  // the persistent engine ran ln_f / final_norm / mel_head itself
  GemvArgs g;
  g.W = gpt.head.w;
  g.Y = ds.logits;
  g.bias = gpt.head.b;
  g.B = B;
  g.N = V;
  g.K = D;
  g.ldy = V;
  g.X = ds.h;
  g.prologue = 2;
  g.ln_gamma = gpt.ln_f.g;
  g.ln_beta = gpt.ln_f.b;
  g.ln2_gamma = nullptr;  // final_norm's affine is folded into mel_head by the packer
  g.ln2_beta = nullptr;
  if (adt == BF16 && B > 4 && D % 32 == 0 && D <= 2048) {
    // batched decode: ln_f -> final_norm as a row kernel (bf16), head on the matrix cores
    ITTS_TRY(ln_rows_bf16(ds.hn, ds.h, gpt.ln_f.g, gpt.ln_f.b, B, D, 1e-5f, 2, ds.partial, ds.pend_split, ds.pend_bias, 1, s));
    ds.pend_split = 0;
    g.X = ds.hn;
    g.x_bf16 = 1;
    g.x_tiled = 1;
    g.prologue = 0;
    g.W8 = gpt.head.w8;
    g.wscale = gpt.head.wscale;
    g.Wt = gpt.head.wt;
    g.W8t = gpt.head.wt8;
    ITTS_TRY(skinny_mfma(g, s));
  } else if (adt == BF16 && gemv_bf16_supported(g)) {
    g.W8 = gpt.head.w8;
    g.wscale = gpt.head.wscale;
    ITTS_TRY(gemv_bf16(g, s));
  } else if (gemv2_supported(g)) {
    ITTS_TRY(gemv2(g, gpt.head.dt, s));
  } else {
    ITTS_TRY(double_ln(ds.hn, ds.h, gpt.ln_f.g, gpt.ln_f.b, nullptr, nullptr, B, D, 1e-5f, s));
    g.X = ds.hn;
    g.prologue = 0;
    ITTS_TRY(gemv(g, gpt.head.dt, s));
  }
  return sample_from_logits(s);
}

// arguments of the beam kernels (beam_sample_step / beam_commit_step) from the decode state
BeamArgs Engine::beam_args(const float* lg_in, bool typical) const {
  const itts_config& c = cfg;
  const int D = c.model_dim, V = c.number_mel_codes, B = ds.B;
  BeamArgs ba;
  ba.logits = lg_in;
  ba.preprocessed = typical;
  ba.do_sample = ds.beam_sample;
  ba.length_penalty = ds.length_penalty;
  ba.cand_sc = ds.cand_sc;
  ba.cand_tok = ds.cand_tok;
  ba.cand_n = ds.cand_n;
  ba.V = V;
  ba.max_gen = ds.max_gen;
  ba.stop = c.stop_mel_token;
  ba.suppress_stop = ds.suppress_stop;
  ba.nb = ds.nb;
  ba.B = B / ds.nb;
  ba.top_k = ds.top_k;
  ba.top_p = ds.top_p;
  ba.temperature = ds.temperature;
  ba.penalty = ds.penalty;
  ba.start_tok = c.start_mel_token;
  ba.fake_id = 1;  // prepare_gpt_inputs: fake ids are all 1 with last = start_mel_token (model.py:644-653)
  ba.Smax = ds.Smax;
  ba.uniforms = ds.uniforms;
  ba.len = ds.len;
  ba.cur_tok = ds.cur_tok;
  ba.unfinished = ds.unfinished;
  ba.ids = ds.beam_ids;
  ba.anc = ds.anc;
  ba.prefix_dev = ds.prefix_dev;
  ba.beam_scores = ds.beam_scores;
  ba.hyp_tok = ds.hyp_tok;
  ba.hyp_score = ds.hyp_score;
  ba.hyp_len = ds.hyp_len;
  ba.hyp_order = ds.hyp_order;
  ba.hyp_n = ds.hyp_n;
  ba.hyp_worst = ds.hyp_worst;
  ba.hyp_counter = ds.hyp_counter;
  ba.done = ds.beam_done;
  ba.h_next = ds.h;
  ba.emb = gpt.mel_emb;
  ba.pos = gpt.mel_pos;
  ba.D = D;
  ba.pos_rows = c.max_mel_tokens + 3;
  ba.emb_bf16 = adt == BF16;
  ba.forced = ds.use_forced ? ds.forced : nullptr;
  ba.input_n = ds.use_forced ? ds.input_n : 0;
  return ba;
}

// repetition penalty / argmax or the sampling / beam modes / bookkeeping on ds.logits [B][V]
int Engine::sample_from_logits(hipStream_t s, bool sampled) {
  const itts_config& c = cfg;
  const int V = c.number_mel_codes, B = ds.B;
  ITTS_TRY(tap("logits0", ds.logits, F32, (int64_t)B * V, s));
  if (sampled) return OK;  // the persistent engine's in-launch greedy sampler committed this step's tokens
  if (ds.host_sample) return OK;  // the caller reads the logits, picks the tokens and commits them (gpt_commit)
  const float* lg_in = ds.logits;
  // (a logits PROCESSOR in the reference, model.py:690-697: greedy search and beam search run it too, not only the sampling modes)
  const bool typical = ds.typical_mass > 0.f;
  if (typical) {  // TypicalLogitsWarper sits in HF's logits_processor list, right after the repetition penalty
    TypicalArgs ta;
    ta.logits = ds.logits;
    ta.out = ds.scores2;
    ta.V = V;
    ta.stop = c.stop_mel_token;
    ta.suppress_stop = ds.suppress_stop;
    ta.penalty = ds.penalty;
    ta.mass = ds.typical_mass;
    ta.min_keep = ds.nb > 1 ? 2 : 1;  // model.py:693-694
    ta.log_softmax_first = ds.nb > 1;  // beam_sample processes log-probs, sample() raw logits
    if (ds.nb > 1) {
      ta.beam_ids = ds.beam_ids;
      ta.len = ds.len;
      ta.max_gen = ds.max_gen;
      ta.start_tok = c.start_mel_token;
      ta.fake_id = 1;
    } else {
      ta.seen = ds.seen;
    }
    ITTS_TRY(typical_filter(ta, B, s));
    ITTS_TRY(tap("typical0", ds.scores2, F32, (int64_t)B * V, s));
    lg_in = ds.scores2;
  }
  if (ds.nb > 1) {  // beam-sample: one workgroup per batch item over its nb rows
    const BeamArgs ba = beam_args(lg_in, typical);
    return beam_sample_step(ba, s);
  }
  SamplerArgs sa = greedy_sampler_args(lg_in, typical);
  return sampler2_step(sa, B, s);
}

// arguments of the greedy / sampling kernels (sampler2_step) - also what the persistent engine's in-launch sampler runs on
SamplerArgs Engine::greedy_sampler_args(const float* lg_in, bool typical) const {
  const itts_config& c = cfg;
  const int D = c.model_dim, V = c.number_mel_codes, B = ds.B;
  SamplerArgs sa;
  sa.logits = lg_in;
  sa.preprocessed = typical;
  sa.seen = ds.seen;
  sa.ids = ds.ids;
  sa.cur_tok = ds.cur_tok;
  sa.unfinished = ds.unfinished;
  sa.step = ds.len;
  sa.V = V;
  sa.max_gen = ds.max_gen;
  sa.stop = c.stop_mel_token;
  sa.suppress_stop = ds.suppress_stop;
  sa.penalty = ds.penalty;
  sa.h_next = ds.h;  // the sampler also prepares the next step's input embedding
  sa.emb = gpt.mel_emb;
  sa.pos = gpt.mel_pos;
  sa.D = D;
  sa.pos_rows = c.max_mel_tokens + 3;
  sa.emb_bf16 = adt == BF16;
  sa.do_sample = ds.do_sample;
  sa.top_k = ds.top_k;
  sa.top_p = ds.top_p;
  sa.temperature = ds.temperature;
  sa.uniforms = ds.uniforms;
  sa.forced = ds.use_forced ? ds.forced : nullptr;
  sa.input_n = ds.use_forced ? ds.input_n : 0;
  sa.B = B;
  return sa;
}

// one decode step: 24 x {LN+QKV gemv, cache attention (+append), proj gemv (+res),
// LN+FC gemv (gelu), proj2 gemv (+res)} -> head.  Every length / position is read from device memory, so the
// captured graph is identical for every step and every prefix length.
int Engine::decode_step_launch(hipStream_t s) {
  const itts_config& c = cfg;
  const int D = c.model_dim, H = c.heads, dh = D / H, B = ds.B;
  // ds.h already holds mel_emb[tok] + mel_pos[...] of this step (written by the previous step's sampler)
  const bool fast = adt == BF16 && B <= 4;  // bf16 activations between the decode kernels (ctx, act)
  const bool skinny = adt == BF16 && B > 4 && D % 32 == 0 && D <= 2048;  // weights once, batch on MFMA
  ds.pend_split = 0;
  ds.last_mode = 0;
  int eng_first = 0;
  if (engine_usable()) {  // <= 6 rows: the 24 blocks (+ head, + greedy sampler) as ONE persistent launch
    ITTS_REQUIRE(ds.eng_gran && ds.eng_ctr, "decode engine: state not allocated (prefill first)");
    EngArgs ea;
    for (int l = 0; l < c.layers; ++l) {
      const GptLayerW& L = gpt.layers[l];
      ea.L[l] = {(const bf16_t*)L.attn.w, (const bf16_t*)L.proj.w, (const bf16_t*)L.fc.w, (const bf16_t*)L.proj2.w,
                 L.attn.b, L.proj.b, L.fc.b, L.proj2.b};
    }
    ea.gran = ds.eng_gran;
    ea.h = ds.h;
    ea.kc = (bf16_t*)ds.kc;
    ea.vc = (bf16_t*)ds.vc;
    ea.len = ds.len;
    ea.kv_start = ds.kv_start;
    ea.prefix = ds.prefix_dev;
    ea.ctr = ds.eng_ctr;
    ea.anc = ds.nb > 1 ? ds.anc : nullptr;  // beam rows gather their keys through the ancestry table (no cache re-ordering)
    ea.nb = ds.nb;
    // ITTS_ENGINE_LAYERS=n (debugging aid): blocks [0, n) on the engine, the rest as launches
    static const int e_nl = getenv("ITTS_ENGINE_LAYERS") ? atoi(getenv("ITTS_ENGINE_LAYERS")) : -1;
    eng_first = e_nl >= 0 && e_nl < c.layers ? e_nl : c.layers;
    ea.NL = eng_first;
    ea.B = B;
    ea.Smax = ds.Smax;
    ea.scale = 1.f / std::sqrt((float)dh);
    static const int e_tap = getenv("ITTS_TAP_LAYER") ? atoi(getenv("ITTS_TAP_LAYER")) : -1;
    if (debug && e_tap >= 0 && e_tap < eng_first && !dry) {  // debugging aid: the engine's view of one block's edges
      ea.dbg = ds.act;  // [16][4D] fp32 scratch of the launch path, unused by the engine: room for the 9 B D floats of the dump
      ea.dbg_layer = e_tap;
    }
    // gather pacing (s_sleep units of 64 clocks): a publish needs ~0.4 us to become visible; earlier passes fail AND slow the
    // stores down.  Defaults from tools/eng_pacing_rows.sh (profiles/r03_engine_pacing_sweep.txt): 14 / 16 at <= 2 rows, 12 / 8
    // at 3 - 4 rows, 12 / 4 at 5 - 6 (more granules per pass: the first pass itself takes longer)
    static const int e_fd_env = getenv("ITTS_ENGINE_FIRST_DELAY") ? atoi(getenv("ITTS_ENGINE_FIRST_DELAY")) : -1;
    static const int e_ad_env = getenv("ITTS_ENGINE_ACT_DELAY") ? atoi(getenv("ITTS_ENGINE_ACT_DELAY")) : -1;
    const int e_fd = e_fd_env >= 0 ? e_fd_env : (B <= 2 ? 14 : 12);
    const int e_ad = e_ad_env >= 0 ? e_ad_env : (B <= 2 ? 16 : B <= 4 ? 8 : 4);
    static const int e_ps = getenv("ITTS_ENGINE_PASS_SLEEP") ? atoi(getenv("ITTS_ENGINE_PASS_SLEEP")) : 1;
    static const int e_thin = getenv("ITTS_ENGINE_THIN_FC") ? atoi(getenv("ITTS_ENGINE_THIN_FC")) : 0;
    ea.thin_fc = e_thin;
    static const int e_early = getenv("ITTS_ENGINE_EARLY_FC") ? atoi(getenv("ITTS_ENGINE_EARLY_FC")) : -1;
    ea.early_fc = e_early >= 0 ? e_early : 74;  // tools/ab_early.sh, tools/eng_pacing_rows.sh (profiles/r03_engine_early_fc.txt)
    ea.first_delay = e_fd;
    static const int e_fake = getenv("ITTS_ENG_FAKE_DIV") ? atoi(getenv("ITTS_ENG_FAKE_DIV")) : 1;
    ea.fake_div = e_fake < 1 ? 1 : e_fake;
    static const int e_ekv = getenv("ITTS_ENGINE_EARLY_KV") ? atoi(getenv("ITTS_ENGINE_EARLY_KV")) : 1;
    ea.early_kv = e_ekv;
    static const int e_cd = getenv("ITTS_ENGINE_CTX_DELAY") ? atoi(getenv("ITTS_ENGINE_CTX_DELAY")) : 0;
    ea.ctx_delay = e_cd;
    ea.act_delay = e_ad;
    ea.pass_sleep = e_ps;
    if (const char* tt = getenv("ITTS_ENGINE_TIMEOUT_TICKS")) ea.timeout_ticks = (unsigned)atol(tt);  // tests: force the give-up path (read per call)
    static const bool e_stamps = getenv("ITTS_ENGINE_STAMPS") != nullptr;
    if (debug && e_stamps && !dry) ea.stamp = (unsigned*)ds.scores2;  // [16][V] fp32 scratch of the typical filter (off in this mode) >= 256 * 24 * 16 words
    ds.last_mode = eng_first > 0;
    // the head (ln_f -> final_norm -> mel_head) inside the same launch when the engine runs every block (ITTS_ENGINE_HEAD=0: own launch)
    static const bool e_head = !(getenv("ITTS_ENGINE_HEAD") && atoi(getenv("ITTS_ENGINE_HEAD")) == 0);
    const bool fold_head = e_head && eng_first == c.layers && gpt.head.dt == BF16 && !gpt.head.w8 && gpt.head.b && gpt.head.Cin == ENG_D &&
                           gpt.ln_f.g && gpt.ln_f.b && (c.number_mel_codes + ENG_NCU - 1) / ENG_NCU <= 33;
    // ... and the greedy sampler behind it (ITTS_ENGINE_SAMPLER=0: own launch): plain greedy search only - sampling, beams,
    // typical filtering and host-side sampling keep their kernels
    static const bool e_samp = !(getenv("ITTS_ENGINE_SAMPLER") && atoi(getenv("ITTS_ENGINE_SAMPLER")) == 0);
    const bool fold_samp = fold_head && e_samp && !ds.do_sample && ds.nb == 1 && !ds.host_sample && !(ds.typical_mass > 0.f);
    if (fold_samp) {
      ea.fold_sampler = 1;
      ea.samp = greedy_sampler_args(ds.logits, false);
      ea.cand = ds.eng_gran + eng_gran_count(c.layers) - ENG_CAND_WORDS;
    }
    if (fold_head) {
      ea.head_w = (const bf16_t*)gpt.head.w;
      ea.head_b = gpt.head.b;
      ea.lnf_g = gpt.ln_f.g;
      ea.lnf_b = gpt.ln_f.b;
      ea.logits = ds.logits;
      ea.V = c.number_mel_codes;
    }
    if (eng_first > 0) ITTS_TRY(decode_engine_layers(ea, s));
    if (ea.stamp) ITTS_TRY(tap("eng_stamps", ea.stamp, F32, (int64_t)ENG_NCU * eng_first * 16, s));
    if (ea.dbg) {
      ITTS_TRY(tap("eng_qkv", ea.dbg, F32, (int64_t)B * 3 * D, s));
      ITTS_TRY(tap("eng_h1", ea.dbg + (size_t)B * 3 * D, F32, (int64_t)B * D, s));
      ITTS_TRY(tap("eng_act", ea.dbg + (size_t)B * 4 * D, F32, (int64_t)B * 4 * D, s));
      ITTS_TRY(tap("eng_h2", ea.dbg + (size_t)B * 8 * D, F32, (int64_t)B * D, s));
    }
    if (eng_first == c.layers) return head_and_sample(s, fold_head, fold_samp);
  }
  // 5-16 rows: the step is launch-bound, so the two LayerNorm launches of a layer fold into the projections they feed
  // (skinny_mfma_kernel<LNP>) and the residual projections run as half tiles without a K split (<HALF>): 5 launches a
  // layer instead of 7 (ITTS_NO_SKINNY16=1 keeps the 7-launch form)
  static const bool no_s16 = getenv("ITTS_NO_SKINNY16") != nullptr;
  const bool small16 = skinny && B <= 16 && D <= 1280 && !no_s16;
  auto run = [&](GemvArgs& g, int dt) -> int {
    if (small16 && ((g.prologue == 1 && !g.ln_gamma) || (g.prologue == 0 && g.accumulate && g.Y == ds.h && g.N == D))) {
      if (g.prologue != 1) {
        g.half_tiles = 1;
        g.x_tiled = 1;
      }
      g.y_tiled = g.y_bf16;
      if (g.w8src) {
        g.W8 = g.w8src->w8;
        g.wscale = g.w8src->wscale;
        g.W8t = g.w8src->wt8;
      }
      return skinny_mfma(g, s);
    }
    if (skinny) {
      if (g.prologue == 1) {  // LayerNorm as a row kernel (its affine lives in the projection); it also absorbs the
                              // split-K partial sums of the projection that fed the residual stream
        ITTS_TRY(ln_rows_bf16(ds.hn, ds.h, g.ln_gamma, g.ln_beta, B, g.K, g.ln_eps, 1, ds.partial, ds.pend_split, ds.pend_bias, 1, s));
        ds.pend_split = 0;
        g.X = ds.hn;
        g.x_bf16 = 1;
        g.prologue = 0;
      } else if (g.accumulate && g.Y == ds.h && g.N == D) {
        // residual projections have few 16-feature tiles (D/16 = 80 on 256 CUs): split K over workgroups, reduced by
        // the LayerNorm that follows.  3 / 5 splits (240 / 200 workgroups): best of tools/split_sweep.sh
        static const int e_s1 = getenv("ITTS_SKINNY_SPLIT_PROJ") ? atoi(getenv("ITTS_SKINNY_SPLIT_PROJ")) : 3;
        static const int e_s2 = getenv("ITTS_SKINNY_SPLIT_PROJ2") ? atoi(getenv("ITTS_SKINNY_SPLIT_PROJ2")) : 5;
        g.ksplit = g.K >= 4 * D ? e_s2 : e_s1;
        g.partial = ds.partial;
        ds.pend_split = g.ksplit;
        ds.pend_bias = g.bias;
      }
      g.x_tiled = 1;           // every bf16 activation of this path (hn, ctx, act) is fragment-tiled
      g.y_tiled = g.y_bf16;
      if (g.w8src) {  // fp8 copy of this projection (BASELINE config 5): the same weight bytes the GEMV path streams
        g.W8 = g.w8src->w8;
        g.wscale = g.w8src->wscale;
        g.W8t = g.w8src->wt8;
      }
      return skinny_mfma(g, s);
    }
    if (fast && gemv_bf16_supported(g)) {
      if (g.w8src) {  // fp8 copy of this projection (decode only; prefill / latent use the bf16 dequantisation of it)
        g.W8 = g.w8src->w8;
        g.wscale = g.w8src->wscale;
      }
      return gemv_bf16(g, s);
    }
    ITTS_REQUIRE(!g.x_bf16 && !g.y_bf16, "decode: bf16 activation without the bf16 GEMV");
    return gemv2_supported(g) ? gemv2(g, dt, s) : gemv(g, dt, s);
  };
  GemvArgs probe;
  probe.B = B;
  probe.K = 4 * D;
  probe.x_bf16 = 1;
  probe.N = D;
  const bool bf_act = skinny || (fast && gemv_bf16_supported(probe));
  probe.K = D;
  const bool bf_ctx = skinny || (fast && gemv_bf16_supported(probe));
  for (int l = eng_first; l < c.layers; ++l) {
    const GptLayerW& L = gpt.layers[l];
    GemvArgs g;  // qkv = LN1(h) Wqkv + b
    g.B = B;
    g.X = ds.h;
    g.W = L.attn.w;
    g.Y = ds.qkv;
    g.bias = L.attn.b;
    g.N = 3 * D;
    g.K = D;
    g.ldy = 3 * D;
    g.prologue = 1;
    g.ln_gamma = L.ln1.g;
    g.ln_beta = L.ln1.b;
    g.w8src = L.attn.w8 ? &L.attn : nullptr;
    g.Wt = L.attn.wt;
    const size_t lo = (size_t)l * B * H * ds.Smax * dh * es;
    // few (row, head) pairs: the keys of each pair go to ATTN_NSPLIT workgroups and the projection merges the partials
    static const bool no_split = getenv("ITTS_ATTN_NOSPLIT") != nullptr;
    static const bool env_fuse = getenv("ITTS_FUSE_QKV_ATTN") != nullptr && atoi(getenv("ITTS_FUSE_QKV_ATTN")) != 0;
    const bool split = fast && bf_ctx && !no_split && (long)B * H <= 128 && D % 64 == 0;
    // projection + attention of the layer as ONE launch (the attention workgroups poll for q / k / v): bf16 weights only
    const bool fused = split && (ds.fuse || env_fuse) && !ds.fuse_failed && !g.w8src && gemv_bf16_supported(g) && qkv_attn_fused_supported(g, H, dh);
    if (fused) {
      ITTS_TRY(qkv_attn_fused(g, ds.gran + (size_t)l * 4 * 3 * D, ds.fuse_err, (char*)ds.kc + lo, (char*)ds.vc + lo, ds.len,
                              ds.kv_start, ds.prefix_dev, H, dh, ds.Smax, ds.attn_o, ds.attn_ml, ds.nb > 1 ? ds.anc : nullptr,
                              ds.nb, s));
    } else {
      ITTS_TRY(run(g, L.attn.dt));
    }
    static const int l_tap = getenv("ITTS_TAP_LAYER") ? atoi(getenv("ITTS_TAP_LAYER")) : -1;
    if (l == l_tap) ITTS_TRY(tap("lp_qkv", ds.qkv, F32, (int64_t)B * 3 * D, s));
    if (fused) {
    } else if (split)
      ITTS_TRY(decode_attn2(nullptr, BF16, ds.qkv, (char*)ds.kc + lo, (char*)ds.vc + lo, ds.len, ds.kv_start, ds.prefix_dev, B, H,
                            dh, ds.Smax, adt, s, 0, ds.attn_o, ds.attn_ml, ds.nb > 1 ? ds.anc : nullptr, ds.nb));
    else
      ITTS_TRY(decode_attn2(ds.ctx, bf_ctx ? BF16 : F32, ds.qkv, (char*)ds.kc + lo, (char*)ds.vc + lo, ds.len, ds.kv_start,
                            ds.prefix_dev, B, H, dh, ds.Smax, adt, s, skinny ? 1 : 0, nullptr, nullptr,
                            ds.nb > 1 ? ds.anc : nullptr, ds.nb));
    GemvArgs p;  // h += ctx Wproj + b
    p.B = B;
    p.X = ds.ctx;
    p.x_bf16 = bf_ctx;
    p.W = L.proj.w;
    p.Y = ds.h;
    p.bias = L.proj.b;
    p.N = D;
    p.K = D;
    p.ldy = D;
    p.accumulate = 1;
    if (split) {
      p.prologue = 3;
      p.attn_o = ds.attn_o;
      p.attn_ml = ds.attn_ml;
      ITTS_REQUIRE(gemv_bf16_supported(p), "decode: split attention needs the bf16 GEMV with prologue 3");
    }
    p.w8src = L.proj.w8 ? &L.proj : nullptr;
    p.Wt = L.proj.wt;
    ITTS_TRY(run(p, L.proj.dt));
    if (l == l_tap) ITTS_TRY(tap("lp_h1", ds.h, F32, (int64_t)B * D, s));
    GemvArgs f;  // act = gelu_new(LN2(h) Wfc + b)
    f.B = B;
    f.X = ds.h;
    f.W = L.fc.w;
    f.Y = ds.act;
    f.y_bf16 = bf_act;
    f.bias = L.fc.b;
    f.N = 4 * D;
    f.K = D;
    f.ldy = 4 * D;
    f.act = ACT_GELU_NEW;
    f.prologue = 1;
    f.ln_gamma = L.ln2.g;
    f.ln_beta = L.ln2.b;
    f.w8src = L.fc.w8 ? &L.fc : nullptr;
    f.Wt = L.fc.wt;
    ITTS_TRY(run(f, L.fc.dt));
    if (l == l_tap) ITTS_TRY(tap("lp_act", ds.act, bf_act ? BF16 : F32, (int64_t)B * 4 * D, s));
    GemvArgs q;  // h += act Wproj2 + b
    q.B = B;
    q.X = ds.act;
    q.x_bf16 = bf_act;
    q.W = L.proj2.w;
    q.Y = ds.h;
    q.bias = L.proj2.b;
    q.N = D;
    q.K = 4 * D;
    q.ldy = D;
    q.accumulate = 1;
    q.w8src = L.proj2.w8 ? &L.proj2 : nullptr;
    q.Wt = L.proj2.wt;
    ITTS_TRY(run(q, L.proj2.dt));
    if (l == l_tap) ITTS_TRY(tap("lp_h2", ds.h, F32, (int64_t)B * D, s));
  }
  return head_and_sample(s);
}

// HF GenerationMixin.sample configuration of infer.py:116-124 (do_sample, top_k, top_p, temperature); the random
// draws are supplied by the caller as uniforms in [0, 1), one per (step, row): row-major [max_gen][B]
int Engine::gpt_set_sampling(int do_sample, int top_k, float top_p, float temperature, const float* uniforms_host, long n) {
  if (!do_sample) {
    ds.do_sample = 0;
    sample_uniforms.clear();
    return OK;
  }
  ITTS_REQUIRE(top_k >= 1 && top_k <= 128, "gpt_set_sampling: top_k must be in [1, 128]");
  ITTS_REQUIRE(top_p > 0.f && top_p <= 1.f && temperature > 0.f, "gpt_set_sampling: need 0 < top_p <= 1 and temperature > 0");
  ITTS_REQUIRE(uniforms_host && n > 0, "gpt_set_sampling: uniforms missing");
  ds.do_sample = 1;
  ds.top_k = top_k;
  ds.top_p = top_p;
  ds.temperature = temperature;
  sample_uniforms.assign(uniforms_host, uniforms_host + n);
  return OK;
}

// TypicalLogitsWarper(mass) in front of the warpers of the sampling modes (typical_sampling=True, model.py:690-697); 0 = off
int Engine::gpt_set_typical(float mass) {
  ITTS_REQUIRE(mass == 0.f || (mass > 0.f && mass < 1.f), "gpt_set_typical: `typical_mass` has to be a float > 0 and < 1");
  ds.typical_mass = mass;
  return OK;
}

// Tokens that replace the sampler's choice for the first n steps of every following generation (n = 0 clears):
// the HF `input_tokens` continuation of inference_speech (model.py:672-686) and teacher forcing for parity tests.
// HF `input_tokens` (inference_speech, model.py:672-686): the given mel tokens are appended to the fake input ids, so the
// reference's FIRST forward embeds [start_mel, t1 .. tn] with mel positions 0 .. n (model.py:141-144) and the first
// generated token is fed at position n + 2 (model.py:151-155).  Same forcing as gpt_set_forced, but token k (0-based) is fed
// at position k + 1 instead of k + 2.
int Engine::gpt_set_input_tokens(const int32_t* ids_host, int B, int n) {
  ITTS_TRY(gpt_set_forced(ids_host, B, n));
  forced_input = forced_n > 0;
  return OK;
}

// Host-side token choice for the generate() modes the device samplers do not cover (HF warpers over the FULL vocabulary:
// top_k = 0 / None): prefill / decode(1) stop behind the head GEMV, the caller fetches the logits, applies the processors and
// warpers itself and hands the chosen token of every row back with gpt_commit, which runs the sampler's bookkeeping (ids,
// repetition bitmap, eos state, step counter, next step's input embedding) with those tokens.
int Engine::gpt_set_host_sampling(int on) {
  ds.host_sample = on ? 1 : 0;
  return OK;
}

int Engine::gpt_commit(const int32_t* tokens_host, hipStream_t s) {
  if (!ds.active || !ds.host_sample) {
    set_error("gpt_commit: needs an active generation in host-sampling mode");
    return E_STATE;
  }
  ITTS_REQUIRE(tokens_host && ds.nb == 1, "gpt_commit: tokens missing (or beams active)");
  std::vector<int> lens(ds.B);
  ITTS_HIP_CHECK(hipMemcpyAsync(lens.data(), ds.len, (size_t)ds.B * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  for (int b = 0; b < ds.B; ++b) {
    ITTS_REQUIRE(tokens_host[b] >= 0 && tokens_host[b] < cfg.number_mel_codes, "gpt_commit: token id out of range");
    ITTS_REQUIRE(lens[b] < ds.max_gen, "gpt_commit: generation already at max length");
    ITTS_HIP_CHECK(hipMemcpyAsync(ds.forced + (size_t)b * ds.max_gen + lens[b], tokens_host + b, 4, hipMemcpyHostToDevice, s));
  }
  ITTS_HIP_CHECK(hipStreamSynchronize(s));  // tokens_host is the caller's buffer
  const int keep = ds.host_sample, forced = ds.use_forced;
  ds.host_sample = 0;
  ds.use_forced = 1;
  const float* lg = ds.logits;
  (void)lg;
  // the sampler's commit path with the forced table: argmax is computed and overridden, bookkeeping as in every other mode
  SamplerArgs sa;
  sa.logits = ds.logits;
  sa.seen = ds.seen;
  sa.ids = ds.ids;
  sa.cur_tok = ds.cur_tok;
  sa.unfinished = ds.unfinished;
  sa.step = ds.len;
  sa.V = cfg.number_mel_codes;
  sa.max_gen = ds.max_gen;
  sa.stop = cfg.stop_mel_token;
  sa.suppress_stop = ds.suppress_stop;
  sa.penalty = ds.penalty;
  sa.h_next = ds.h;
  sa.emb = gpt.mel_emb;
  sa.pos = gpt.mel_pos;
  sa.D = cfg.model_dim;
  sa.pos_rows = cfg.max_mel_tokens + 3;
  sa.emb_bf16 = adt == BF16;
  sa.forced = ds.forced;
  sa.input_n = 0;
  sa.B = ds.B;
  const int st = sampler2_step(sa, ds.B, s);
  ds.host_sample = keep;
  ds.use_forced = forced;
  return st;
}

// Host-side beam_sample (the generate() kwargs the device beam sampler does not cover: top_k = 0 / None or > 128 with several
// beams, infer.py:116-124 + webui.py:393-402).  The step stops behind the head GEMV (host sampling mode); the caller reads the
// logits and gpt_beam_state (every beam's id history, running scores, the done flags), does log_softmax -> processors ->
// warpers -> + beam scores -> the 2 * num_beams draws itself and hands the picks IN DRAW ORDER to gpt_commit_beams, which runs
// BeamSearchScorer.process and the history / cache-ancestry swap on the device exactly as the device-sampled mode does
// (beam_select_kernel with the draw skipped), so hypotheses, finalize and the engine's beam rows are shared.
int Engine::gpt_beam_state(int32_t* ids_host, float* scores_host, int32_t* done_host, int* step_host, hipStream_t s) {
  if (!ds.active || ds.nb <= 1) {
    set_error("gpt_beam_state: needs an active beam generation");
    return E_STATE;
  }
  int k = 0;
  ITTS_HIP_CHECK(hipMemcpyAsync(&k, ds.len, 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  const size_t rows = (size_t)ds.B, mg = (size_t)ds.max_gen;
  if (ids_host) ITTS_HIP_CHECK(hipMemcpyAsync(ids_host, ds.beam_ids + (size_t)(k & 1) * rows * mg, rows * mg * 4, hipMemcpyDeviceToHost, s));
  if (scores_host) ITTS_HIP_CHECK(hipMemcpyAsync(scores_host, ds.beam_scores, rows * 4, hipMemcpyDeviceToHost, s));
  if (done_host) ITTS_HIP_CHECK(hipMemcpyAsync(done_host, ds.beam_done, (size_t)(ds.B / ds.nb) * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  if (step_host) *step_host = k;
  return OK;
}

int Engine::gpt_commit_beams(const float* pick_score_host, const int32_t* pick_tok_host, const int32_t* pick_beam_host, hipStream_t s) {
  if (!ds.active || !ds.host_sample || ds.nb <= 1) {
    set_error("gpt_commit_beams: needs an active beam generation in host-sampling mode");
    return E_STATE;
  }
  ITTS_REQUIRE(pick_score_host && pick_tok_host && pick_beam_host, "gpt_commit_beams: picks missing");
  const int items = ds.B / ds.nb, nd = 2 * ds.nb;
  for (int i = 0; i < items * nd; ++i) {
    ITTS_REQUIRE(pick_tok_host[i] >= 0 && pick_tok_host[i] < cfg.number_mel_codes, "gpt_commit_beams: token id out of range");
    ITTS_REQUIRE(pick_beam_host[i] >= 0 && pick_beam_host[i] < ds.nb, "gpt_commit_beams: beam index out of range");
  }
  // the candidate scratch of the device sampler is idle in this mode: [0, items * nd) of cand_sc / cand_tok, beams behind the tokens
  static_assert(BEAM_MAX_CAND >= 4, "pick scratch fits the candidate scratch");
  ITTS_HIP_CHECK(hipMemcpyAsync(ds.cand_sc, pick_score_host, (size_t)items * nd * 4, hipMemcpyHostToDevice, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(ds.cand_tok, pick_tok_host, (size_t)items * nd * 4, hipMemcpyHostToDevice, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(ds.cand_tok + (size_t)items * nd, pick_beam_host, (size_t)items * nd * 4, hipMemcpyHostToDevice, s));
  BeamArgs ba = beam_args(ds.logits, false);
  ba.host_sc = ds.cand_sc;
  ba.host_tok = ds.cand_tok;
  ba.host_beam = ds.cand_tok + (size_t)items * nd;
  const int st = beam_commit_step(ba, s);
  ITTS_HIP_CHECK(hipStreamSynchronize(s));  // the picks are the caller's buffers
  return st;
}

int Engine::gpt_set_forced(const int32_t* ids_host, int B, int n) {
  forced_input = 0;
  if (n <= 0 || !ids_host) {
    forced_host.clear();
    forced_B = forced_n = 0;
    return OK;
  }
  ITTS_REQUIRE(B >= 1, "gpt_set_forced: B < 1");
  for (long i = 0; i < (long)B * n; ++i)
    ITTS_REQUIRE(ids_host[i] >= -1 && ids_host[i] < cfg.number_mel_codes, "gpt_set_forced: token id out of range");
  forced_host.assign(ids_host, ids_host + (size_t)B * n);
  forced_B = B;
  forced_n = n;
  return OK;
}

// The persistent decode engine keeps one workgroup on EVERY CU for a whole token step and its workgroups wait for each other:
// two such launches in flight together (two Engine objects decoding on two streams of one device) could each hold part of
// the CUs and wait for workgroups that cannot become resident - until the 20 ms bound of the waits fires.  So engine launches
// of one device are chained across streams: a decode call first makes its stream wait for the event behind the last engine
// launch of the device (a no-op on the same stream), then records its own.  Other kernels may share the device freely: they
// finish on their own and the engine's workgroups simply start later.
namespace {
struct EngineGate {
  std::mutex mu;
  hipEvent_t ev = nullptr;
};
EngineGate& engine_gate(int dev) {
  static EngineGate gates[64];
  return gates[dev & 63];
}
}  // namespace

int Engine::gpt_decode(int nsteps, hipStream_t s) {
  if (!ds.active) {
    set_error("gpt_decode: call itts_gpt_prefill first");
    return E_STATE;
  }
  static const bool no_gate = getenv("ITTS_ENGINE_NO_GATE") != nullptr;  // debugging aid: show what the chaining prevents
  if (!engine_usable() || dry || no_gate) return gpt_decode_steps(nsteps, s);
  int dev = 0;
  ITTS_HIP_CHECK(hipGetDevice(&dev));
  EngineGate& g = engine_gate(dev);
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.ev) ITTS_HIP_CHECK(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
  else ITTS_HIP_CHECK(hipStreamWaitEvent(s, g.ev, 0));
  const int rc = gpt_decode_steps(nsteps, s);
  ITTS_HIP_CHECK(hipEventRecord(g.ev, s));
  return rc;
}

int Engine::gpt_decode_steps(int nsteps, hipStream_t s) {
  ITTS_REQUIRE(nsteps >= 0, "gpt_decode: nsteps < 0");
  if (use_graph && s != nullptr) {
    DecodeState& d = ds;
    // everything SamplerArgs carries by value is baked into the captured nodes: max_gen is the ids row stride and the
    // `k < max_gen` bound, so a per-request max_mel_tokens must re-capture (same B / Smax notwithstanding)
    const bool stale = !d.graph || d.graph_B != d.B || d.graph_Smax != d.Smax || d.graph_max_gen != d.max_gen ||
                       d.graph_forced != d.use_forced || d.graph_input_n != d.input_n || d.graph_host_sample != d.host_sample || d.graph_fuse != (d.fuse && !d.fuse_failed) || d.graph_eng != (int)engine_usable() || d.graph_nb != d.nb || d.graph_beam_sample != d.beam_sample ||
                       d.graph_length_penalty != d.length_penalty || d.graph_typical != d.typical_mass || d.graph_penalty != d.penalty || d.graph_suppress != d.suppress_stop || d.graph_sample != d.do_sample ||
                       d.graph_top_k != d.top_k || d.graph_top_p != d.top_p || d.graph_temperature != d.temperature;
    // two executables: one step, and GK steps back to back (one launch per GK tokens: the gap between consecutive graph
    // launches is paid once per GK steps; every step reads its lengths from device memory, so any mix is valid)
    static const int GK = [] {
      const char* e = getenv("ITTS_GRAPH_STEPS");
      const int v = e ? atoi(e) : 8;
      return v < 1 ? 1 : (v > 32 ? 32 : v);
    }();
    if (stale && nsteps > 0) {
      for (hipGraphExec_t* ge : {&d.graph, &d.graphK})
        if (*ge) {
          (void)hipGraphExecDestroy(*ge);
          *ge = nullptr;
        }
      for (int which = 0; which < (GK > 1 ? 2 : 1); ++which) {
        hipGraph_t g = nullptr;
        ITTS_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int st = OK;
        for (int k = 0; k < (which ? GK : 1) && st == OK; ++k) st = decode_step_launch(s);
        const hipError_t ee = hipStreamEndCapture(s, &g);
        if (st != OK) {
          if (g) (void)hipGraphDestroy(g);
          return st;
        }
        ITTS_HIP_CHECK(ee);
        ITTS_HIP_CHECK(hipGraphInstantiate(which ? &d.graphK : &d.graph, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
      }
      d.graph_B = d.B;
      d.graph_Smax = d.Smax;
      d.graph_max_gen = d.max_gen;
      d.graph_forced = d.use_forced;
      d.graph_input_n = d.input_n;
      d.graph_host_sample = d.host_sample;
      d.graph_fuse = d.fuse && !d.fuse_failed;
      d.graph_eng = (int)engine_usable();
      d.graph_mode = d.last_mode;  // what the captured step runs on
      d.graph_nb = d.nb;
      d.graph_beam_sample = d.beam_sample;
      d.graph_length_penalty = d.length_penalty;
      d.graph_typical = d.typical_mass;
      d.graph_penalty = d.penalty;
      d.graph_suppress = d.suppress_stop;
      d.graph_sample = d.do_sample;
      d.graph_top_k = d.top_k;
      d.graph_top_p = d.top_p;
      d.graph_temperature = d.temperature;
    }
    d.last_mode = d.graph_mode;  // a replayed graph: the mode it was captured with
    int left = nsteps;
    if (d.graphK)
      for (; left >= GK; left -= GK) ITTS_HIP_CHECK(hipGraphLaunch(d.graphK, s));
    for (; left > 0; --left) ITTS_HIP_CHECK(hipGraphLaunch(d.graph, s));
    return OK;
  }
  for (int i = 0; i < nsteps; ++i) ITTS_TRY(decode_step_launch(s));
  return OK;
}

// abort word of the persistent decode engine: a hand-off wait gave up, so the codes of this generation are not valid.
// The engine is switched off for this Engine (launch path from the next generation on) and its state re-initialised.
int Engine::engine_check(hipStream_t s) {
  if (ds.fuse_err && (ds.fuse || ds.graph_fuse)) {  // the opt-in fused projection + attention launch: checked at fetch too
    int ferr = 0;
    ITTS_HIP_CHECK(hipMemcpyAsync(&ferr, ds.fuse_err, 4, hipMemcpyDeviceToHost, s));
    ITTS_HIP_CHECK(hipStreamSynchronize(s));
    if (ferr) {  // an attention workgroup gave up waiting for its q / k / v: the codes of this generation are not valid
      ds.fuse_failed = 1;  // later generations take the two-launch path
      ITTS_HIP_CHECK(hipMemsetAsync(ds.fuse_err, 0, 4, s));
      set_error("in-launch q/k/v hand-off timed out (fused projection + attention disabled for this engine; the codes of this generation are not valid)");
      return E_HANDOFF;
    }
  }
  if (!ds.eng_ctr) return OK;
  unsigned host[2] = {0, 0};
  ITTS_HIP_CHECK(hipMemcpyAsync(host, ds.eng_ctr, 8, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  if (!host[1]) return OK;
  ds.eng_failed = 1;
  // The drained launches computed on hand-offs that never arrived: what they appended to the K/V cache is arbitrary bits, and the
  // decode attention multiplies rows past the sequence end by p = 0 instead of selecting them away (ensure_decode_state) - a NaN
  // or infinity left there would poison the next generation's first steps.  Back to the zero fill of a fresh allocation.
  if (ds.kc && ds.vc && ds.cache_bytes) {
    ITTS_HIP_CHECK(hipMemsetAsync(ds.kc, 0, ds.cache_bytes, s));
    ITTS_HIP_CHECK(hipMemsetAsync(ds.vc, 0, ds.cache_bytes, s));
  }
  const unsigned ctr0[2] = {host[0] ? host[0] : 1u, 0u};
  ITTS_HIP_CHECK(hipMemcpyAsync(ds.eng_ctr, ctr0, 8, hipMemcpyHostToDevice, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  set_error("decode engine: an in-launch hand-off timed out (persistent engine disabled for this engine; the codes of this generation are not valid)");
  fprintf(stderr, "[itts_hip] persistent decode engine: a hand-off wait gave up (another process or kernel held CUs of this GPU?); "
                  "this engine object uses the five-launches-per-block decode path from now on\n");
  return E_HANDOFF;
}

int Engine::gpt_status(int* steps, int* n_unf, hipStream_t s) {
  if (!ds.active) {
    set_error("gpt_status: no active generation");
    return E_STATE;
  }
  std::vector<int> host(ds.B + 1);
  ITTS_HIP_CHECK(hipMemcpyAsync(host.data(), ds.len, 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipMemcpyAsync(host.data() + 1, ds.unfinished, (size_t)ds.B * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  ITTS_TRY(engine_check(s));
  if (steps) *steps = host[0];
  if (n_unf) {
    int n = 0;
    for (int b = 0; b < ds.B; ++b) n += host[1 + b] != 0;
    *n_unf = n;
  }
  return OK;
}

int Engine::gpt_fetch(int32_t* codes, float* logits, hipStream_t s) {
  if (!ds.active) {
    set_error("gpt_fetch: no active generation");
    return E_STATE;
  }
  ITTS_TRY(engine_check(s));
  if (codes && ds.nb > 1)
    ITTS_TRY(beam_finalize(codes, s));  // [B / nb * beam_returns][max_gen]
  else if (codes)
    ITTS_HIP_CHECK(hipMemcpyAsync(codes, ds.ids, (size_t)ds.B * ds.max_gen * 4, hipMemcpyDeviceToHost, s));
  if (logits)
    ITTS_HIP_CHECK(hipMemcpyAsync(logits, ds.logits, (size_t)ds.B * cfg.number_mel_codes * 4, hipMemcpyDeviceToHost, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  return OK;
}

// UnifiedVoice.forward(return_latent=True) for one or more sentences in ONE pass.  The reference runs it at batch 1
// per sentence (infer.py:194-200); sentences are independent, so several are stacked here as left-padded rows
// ([pad.., cond, text, mel] with the pad keys masked exactly like prepare_gpt_inputs' left padding), which fills the
// MFMA tiles better without changing any row's result.  out = concatenated [sum T_i, D] latents in the engine dtype.
int Engine::gpt_latent_batch(const float* cond_dev, const int32_t* text_ids, const int* Ls, const int32_t* codes,
                             const int* Ts, int nseq, void* latent_out, hipStream_t s) {
  if (!finalized || !gpt.ok) {
    set_error("gpt_latent: GPT weights not bound");
    return E_STATE;
  }
  const itts_config& c = cfg;
  ITTS_REQUIRE(cond_dev && text_ids && codes && latent_out && Ls && Ts && nseq > 0, "gpt_latent: bad arguments");
  const int D = c.model_dim, nl = c.cond_latents;
  int Smax = 0;
  long Ttot = 0;
  for (int i = 0; i < nseq; ++i) {
    ITTS_REQUIRE(Ls[i] > 0 && Ts[i] > 0, "gpt_latent: empty sentence");
    ITTS_REQUIRE(Ls[i] + 2 <= c.max_text_tokens + 2 && Ts[i] + 2 <= c.max_mel_tokens + 3, "gpt_latent: sequence too long");
    Smax = std::max(Smax, nl + Ls[i] + 2 + Ts[i] + 2);
    Ttot += Ts[i];
  }
  std::vector<RowDesc> rd((size_t)nseq * Smax);
  std::vector<int> kvs(nseq), mel_row(nseq);
  long to = 0, co = 0;
  for (int b = 0; b < nseq; ++b) {
    const int L = Ls[b], T = Ts[b];
    const int S = nl + L + 2 + T + 2, pad = Smax - S;
    kvs[b] = pad;
    RowDesc* r = rd.data() + (size_t)b * Smax;
    int k = 0;
    for (int i = 0; i < pad; ++i) r[k++] = {0, 0, 0, 0};
    for (int i = 0; i < nl; ++i) r[k++] = {1, i, 0, 0};
    r[k++] = {2, c.start_text_token, 0, 0};
    for (int i = 0; i < L; ++i) {
      const int t = text_ids[to + i];
      ITTS_REQUIRE(t >= 0 && t <= c.number_text_tokens, "gpt_latent: text id out of range");
      r[k++] = {2, t, i + 1, 0};
    }
    r[k++] = {2, c.stop_text_token, L + 1, 0};
    mel_row[b] = b * Smax + k;  // first mel row (the start token); model.py:578 keeps rows [0, T) of the mel part
    r[k++] = {3, c.start_mel_token, 0, 0};
    for (int i = 0; i < T; ++i) {
      const int m = codes[co + i];
      ITTS_REQUIRE(m >= 0 && m < c.number_mel_codes, "gpt_latent: mel code out of range");
      r[k++] = {3, m, i + 1, 0};
    }
    r[k++] = {3, c.stop_mel_token, T + 1, 0};
    to += L;
    co += T;
  }
  auto body = [&]() -> int {
    RowDesc* rd_dev = (RowDesc*)alloc(rd.size() * sizeof(RowDesc));
    int* kvs_dev = (int*)alloc((size_t)nseq * 4);
    float* h = (float*)alloc((size_t)nseq * Smax * D * 4);
    float* hn = (float*)alloc((size_t)Ttot * D * 4);
    if (!dry) {
      ITTS_HIP_CHECK(hipMemcpyAsync(rd_dev, rd.data(), rd.size() * sizeof(RowDesc), hipMemcpyHostToDevice, s));
      ITTS_HIP_CHECK(hipMemcpyAsync(kvs_dev, kvs.data(), (size_t)nseq * 4, hipMemcpyHostToDevice, s));
      if (adt == F32)
        hipLaunchKernelGGL(gpt_embed_rows_kernel<float>, dim3(nseq * Smax), dim3(256), 0, s, h, rd_dev, cond_dev,
                           (const float*)gpt.text_emb, (const float*)gpt.text_pos, (const float*)gpt.mel_emb,
                           (const float*)gpt.mel_pos, D);
      else
        hipLaunchKernelGGL(gpt_embed_rows_kernel<bf16_t>, dim3(nseq * Smax), dim3(256), 0, s, h, rd_dev, cond_dev,
                           (const bf16_t*)gpt.text_emb, (const bf16_t*)gpt.text_pos, (const bf16_t*)gpt.mel_emb,
                           (const bf16_t*)gpt.mel_pos, D);
      ITTS_HIP_CHECK(hipGetLastError());
    }
    ITTS_TRY(gpt_layers_full(h, nseq, Smax, nseq > 1 ? kvs_dev : nullptr, false, s));
    long off = 0;
    for (int b = 0; b < nseq; ++b) {
      K(double_ln(hn + off * D, h + (size_t)mel_row[b] * D, gpt.ln_f.g, gpt.ln_f.b, gpt.final_norm.g, gpt.final_norm.b,
                  Ts[b], D, 1e-5f, s));
      off += Ts[b];
    }
    K(cast_copy(latent_out, adt, hn, F32, Ttot * D, s));
    return OK;
  };
  ITTS_TRY(two_pass(body, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));  // rd / kvs are host buffers
  return OK;
}

int Engine::gpt_latent(const float* cond_dev, const int32_t* text_ids, int L, const int32_t* codes, int T,
                       void* latent_out, hipStream_t s) {
  return gpt_latent_batch(cond_dev, text_ids, &L, codes, &T, 1, latent_out, s);
}

}  // namespace itts
