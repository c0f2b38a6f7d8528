/* libitts_hip - C ABI of the MI355X-native IndexTTS hot path.
 *
 * Conventions: plain pointers and sizes, no torch types; every pointer is a DEVICE pointer
 * unless the parameter name ends in `_host`; the caller allocates outputs; calls are asynchronous on the
 * given HIP stream unless stated; the return value is 0 or a negative error code (ITTS_E_*) and
 * `itts_last_error()` returns the message (thread-local).  One engine per host thread (the reference's
 * web UI shares one engine between threads without a lock, webui.py:441-452: callers must serialise).
 *
 * dtype codes: 0 = f32, 1 = bf16, 5 = f16 (itts_snake_aa_fwd only).  Activations are channels-last ([B, T, C]) everywhere.
 */
#ifndef ITTS_HIP_H
#define ITTS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* itts_stream; /* hipStream_t */
typedef struct itts_engine itts_engine;

#define ITTS_F32 0
#define ITTS_BF16 1
#define ITTS_F16 5 /* IEEE half: accepted by itts_snake_aa_fwd only (the reference op dispatches float/half/bf16) */

/* status codes */
#define ITTS_OK 0
#define ITTS_E_INVALID (-1)
#define ITTS_E_HIP (-2)
#define ITTS_E_NOMEM (-3)
#define ITTS_E_STATE (-4)
#define ITTS_E_MISSING (-5)
/* itts_gpt_status / itts_gpt_fetch: a hand-off wait inside the persistent decode engine gave up (20 ms wall clock; the engine
 * needs the 256 CUs of its GPU to itself - ONE process per GPU).  The codes of this generation are not valid;
 * the engine object has switched itself to the five-launches-per-block path, so the caller simply generates again (the Python
 * shim does, once, with a logged warning). */
#define ITTS_E_HANDOFF (-6)

const char* itts_last_error(void);
int itts_abi_version(void);
/* The 16-bit storage type behind dtype code ITTS_BF16 in THIS build of the library: 0 = bfloat16 (libitts_hip.so), 1 = IEEE
 * binary16 (libitts_hip_f16.so: the same sources with -DITTS_HALF_F16 - the reference's GPU precision, `is_fp16=True` =
 * fp16 autocast / .half(), indextts/infer.py:39,44,52).  Weights, activations and the K/V cache are stored in it; accumulation is
 * fp32 in both.  The f16 build has no fp8 weight copies (BASELINE config 5 is a bf16-build mode). */
int itts_half_is_f16(void);

/* ---- operator level -------------------------------------------------------------------------------- */

/* Fused anti-aliased SnakeBeta.  Replaces the reference's only native entry point
 *   anti_alias_activation_cuda.forward(input, up_filter, down_filter, alpha, beta) -> Tensor
 *   (indextts/BigVGAN/alias_free_activation/cuda/anti_alias_activation.cpp:19-23, fwd_cuda .cu:214-256)
 * Same contract: alpha/beta are LOG-scale fp32 [C]; filters fp32 [12]; dst has the shape/dtype of src;
 * src is not modified.  layout 0 = [B, C, T] (the reference's), 1 = [B, T, C] (engine-native). */
int itts_snake_aa_fwd(void* dst, const void* src, const float* up12, const float* down12, const float* log_alpha,
                      const float* log_beta, int B, int C, int T, int dtype, int layout, itts_stream stream);

/* Generic linear / conv1d / transposed conv1d on channels-last activations (replaces nn.Linear, nn.Conv1d,
 * nn.ConvTranspose1d calls of BigVGAN/models.py:149-161,184 and the conformer/GPT projections).
 * W is [nphase][N][taps*Cin].  See csrc/itts_common.h GemmArgs for the epilogue definition. */
typedef struct {
  const void* A; const void* W; void* C;
  int M, N, Cin, taps, lda, ldc, T, dil, pad_left, pad_mode, in_up, nphase;
  int phase_shift[8];
  const float* bias; int bias_bstride; int act;
  const float* scale; const float* shift; int act2;
  const void* R; int ldr; float alpha;
  const void* ADD; int ldadd; float beta;
  int dtype_a, dtype_w, dtype_c;
  int force_simple; /* 1 = vector-ALU kernel even when the MFMA kernel supports the shape */
} itts_gemm_args;
int itts_gemm(const itts_gemm_args* args, itts_stream stream);
/* Which kernel family itts_gemm takes for these arguments (no launch): 0 vector ALU, 1 register-staged MFMA, 2 LDS-DMA staged
 * 128-wide tiles, 3 the 256 x 256 eight-phase kernel, 4 the LDS-tiled narrow conv.  Tests and tools/bench_gemm.py use it. */
int itts_gemm_which(const itts_gemm_args* args);
/* itts_gemm with a caller-owned fp32 workspace (16-byte aligned, ws_bytes long): few-tile, deep-K shapes (M x N in fewer than 128
 * tiles of 256 x 256, K >= 1024) split K over up to 8 workgroups per tile - raw sums into ws[split][M][N], a second launch adds
 * them in split order (deterministic) and runs the epilogue.  Other shapes run exactly as itts_gemm.  itts_gemm_ksplit returns the
 * split count the call would use (1 = none) without launching.  The engine's own GEMMs use a 64 MiB workspace of the engine. */
int itts_gemm_ws(const itts_gemm_args* args, void* ws, size_t ws_bytes, itts_stream stream);
int itts_gemm_ksplit(const itts_gemm_args* args, size_t ws_bytes);

int itts_layernorm(void* y, int dtype_y, const void* x, int dtype_x, const float* gamma, const float* beta, int rows,
                   int D, float eps, itts_stream stream);

/* attention over strided q/k/v (see csrc/itts_kernels.h AttnArgs) */
int itts_attention(void* o, const void* q, const void* k, const void* v, int B, int H, int Sq, int Sk, int dqk, int dv,
                   int ldq, int ldk, int ldv, int ldo, float scale, int causal, const int* kv_start, int dtype,
                   itts_stream stream);

/* Decode-step batched GEMV (GPT2 Conv1D on one token per row, HF GPT2Attention/GPT2MLP):
 * Y[b, n] (+)= act(prologue(X)[b, :] . W[n, :] + bias[n]); X, Y fp32; prologue 0 none, 1 LayerNorm, 2 LN o LN.
 * version 0 = auto, 1 = generic kernel, 2 = register-resident kernel (B <= 4). */
int itts_gemv(float* Y, const float* X, const void* W, const float* bias, int B, int N, int K, int act, int accumulate,
              int prologue, const float* ln_gamma, const float* ln_beta, const float* ln2_gamma, const float* ln2_beta,
              int dtype_w, int version, itts_stream stream);

/* Decode-step projections at batch > 4 (same Conv1D call sites): X bf16 [B, K], W bf16 [N, K], weights streamed once,
 * batch on MFMA; Y fp32 [B, N] (store, or += when accumulate) or bf16 when y_bf16.  K % 32 == 0, B <= 128.
 * ksplit > 1 splits K over workgroups: raw sums go to partial[ksplit][B][N] (no bias / act / Y), to be absorbed by
 * itts_ln_rows_bf16 (deterministic two-stage reduction, no atomics).
 * layout bit 0: X is MFMA-fragment tiled, bit 1: bf16 Y is written tiled - element (b, k) at
 * ((k/32) * ceil(B/16) + b/16) * 512 + ((k%32)/8 * 16 + b%16) * 8 + k%8 (csrc/itts_decode.h tile_off).
 * layout bit 2: W is the fragment-tiled copy made by itts_retile_weights (same bits out, 1.3x faster weight stream).
 * B <= 16 only - bit 3: X is the fp32 residual stream and LayerNorm (eps 1e-5, affine folded into W) runs in the
 * prologue (K <= 1280, ksplit 1); bit 4: 8 features per workgroup (narrow residual projections, ksplit 1). */
int itts_skinny_gemm(void* Y, int y_bf16, const void* X, const void* W, const float* bias, int B, int N, int K, int act,
                     int accumulate, int ksplit, float* partial, int layout, itts_stream stream);

/* dst[16*ceil(N/16)*K] <- bf16 W[N, K] in MFMA-fragment tiles: element (n, k) at
 * ((n/16) * (K/32) + k/32) * 512 + ((k%32)/8 * 16 + n%16) * 8 + k%8, rows past N zero (csrc/itts_decode.h wtile_off).
 * The engine makes these copies of the GPT projections itself on the first batched generation. */
int itts_retile_weights(void* dst, const void* src, int N, int K, itts_stream stream);

/* y (bf16) = LayerNorm(x fp32) [passes == 2: LayerNorm again without affine], GPT-2 ln_1 / ln_2 / ln_f o final_norm.
 * nsplit > 0: first x += partial_bias + sum_s partial[s] (the residual add of a split-K projection), written back. */
int itts_ln_rows_bf16(void* y, float* x, const float* gamma, const float* beta, int rows, int D, float eps, int passes,
                      const float* partial, int nsplit, const float* partial_bias, int y_tiled, itts_stream stream);

int itts_transpose(void* y, const void* x, int B, int R, int C, int dtype, itts_stream stream);

/* ---- engine level ----------------------------------------------------------------------------------- */

typedef struct {
  int dtype; /* weight + activation dtype of the engine (ITTS_F32 parity path / ITTS_BF16 throughput path) */
  /* gpt (UnifiedVoice ctor, gpt/model.py:301-379) */
  int model_dim, layers, heads, max_mel_tokens, max_text_tokens, number_text_tokens, number_mel_codes;
  int start_mel_token, stop_mel_token, start_text_token, stop_text_token, cond_latents;
  /* conformer + perceiver (condition_module) */
  int cond_dim, cond_ff, cond_heads, cond_blocks, cond_idim, perc_inner, perc_layers;
  /* bigvgan (BigVGAN.__init__, BigVGAN/models.py:132-197) */
  int bv_gpt_dim, bv_init_ch, bv_num_up, bv_up_rates[8], bv_up_kernels[8];
  int bv_num_res, bv_res_kernels[4], bv_res_dils[4][4], bv_num_dil, bv_spk_dim, bv_num_mels;
  /* ECAPA-TDNN (ECAPA_TDNN.py:429-449 defaults) */
  int ec_channels[5], ec_kernels[5], ec_dils[5], ec_att, ec_scale, ec_se;
  /* DVAE decoder (vqvae/xtts_dvae.py:202-291) */
  int dv_channels, dv_tokens, dv_hidden, dv_resblocks, dv_codebook, dv_layers, dv_kernel;
  /* limits */
  int max_batch;
} itts_config;

int itts_engine_create(const itts_config* cfg, itts_engine** out);
void itts_engine_destroy(itts_engine* e);
/* Register a packed tensor that lives in caller-owned device memory (the weight arena). */
int itts_engine_bind_tensor(itts_engine* e, const char* name, const void* ptr, int dtype, int ndim, const int64_t* dims);
/* Validate that every tensor the configured model needs is bound with the right shape/dtype. */
int itts_engine_finalize(itts_engine* e);

/* G2/C1/P1  UnifiedVoice.get_conditioning (gpt/model.py:490-502): mel [1, F, idim] -> cond fp32 [latents, D] */
int itts_conditioning(itts_engine* e, const void* mel_bfc, int F, float* cond_out, itts_stream stream);
/* The same for a prompt that is the first F frames of a tensor padded to F_total frames (get_conditioning with
 * cond_mel_lengths, gpt/model.py:490-502: masked conformer + perceiver; mel_bfc holds at least F frames). */
int itts_conditioning_padded(itts_engine* e, const void* mel_bfc, int F, int F_total, float* cond_out, itts_stream stream);
/* V6  ECAPA_TDNN.forward (BigVGAN/ECAPA_TDNN.py:545-581): mel [B, F, num_mels] -> spk fp32 [B, spk_dim] */
int itts_ecapa(itts_engine* e, const void* mel_bfc, int B, int F, float* spk_out, itts_stream stream);

/* Sampling mode of the next itts_gpt_prefill / itts_gpt_decode calls: HF 4.36.2 GenerationMixin.sample as
 * indextts/infer.py:116-124 + gpt/model.py:690-703 configure it with num_beams = 1 (RepetitionPenaltyLogitsProcessor ->
 * TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax -> one draw).  The draw is the inverse CDF
 * of uniforms_host[k * B + b] (step k, row b; host array of n_uniforms >= max_gen * B floats in [0, 1)) over the kept
 * tokens in descending-score order, so a caller-side RNG fixes the sequence.  do_sample = 0 returns to greedy.
 * 1 <= top_k <= 128, 0 < top_p <= 1, temperature > 0. */
int itts_gpt_set_sampling(itts_engine* e, int do_sample, int top_k, float top_p, float temperature, const float* uniforms_host,
                          int64_t n_uniforms);

/* Beam-sample mode of the next generations: HF 4.36.2 GenerationMixin.beam_sample + BeamSearchScorer, the generate() mode
 * the reference's DEFAULT kwargs select (infer.py:116-124: do_sample=True, num_beams=3, top_k=30, top_p=0.8,
 * length_penalty=0.0; call site gpt/model.py:698-703; cache re-ordering gpt/model.py:194-207).  Per step and batch item:
 * log_softmax -> RepetitionPenalty -> Temperature -> TopK -> TopP (min_tokens_to_keep 2) -> + beam scores -> 2 * num_beams
 * draws without replacement -> BeamSearchScorer.process, all on the device; the KV cache is not copied when beams swap
 * (per-beam ancestry rows).  itts_gpt_prefill then takes B batch items and runs B * num_beams rows (<= max_batch);
 * itts_gpt_fetch returns the finalized best hypothesis per batch item, codes [B, max_gen] padded with the stop token.
 * uniforms_host: [max_gen][B][2 * num_beams] floats in [0, 1): draw j of (step, item) is the inverse CDF of its uniform over
 * the not-yet-drawn candidates in flat (beam-major, token-ascending) order.  2 <= num_beams <= 10; num_beams <= 1 = off. */
int itts_gpt_set_beam_sample(itts_engine* e, int num_beams, int top_k, float top_p, float temperature, const float* uniforms_host,
                             int64_t n_uniforms);

/* The general beam entry point (itts_gpt_set_beam_sample = do_sample 1, length_penalty 0): do_sample = 0 selects HF
 * beam_search - deterministic, per step the 2 * num_beams best of log_softmax + repetition penalty + beam score, no warpers,
 * no uniforms - which is what `num_beams > 1, do_sample=False` means to generate(); length_penalty enters
 * BeamHypotheses' score = sum_logprobs / generated_len ** length_penalty (infer.py:121 passes 0.0).  2 <= num_beams <= 10,
 * top_k <= 128 (the web UI offers num_beams 1..10, top_k 0..100). */
int itts_gpt_set_beams(itts_engine* e, int num_beams, int do_sample, int top_k, float top_p, float temperature, float length_penalty,
                       const float* uniforms_host, int64_t n_uniforms);

/* `num_return_sequences` of generate() under beams (gpt/model.py:655,698-703 forwards it; HF's BeamSearchScorer keeps
 * num_beam_hyps_to_keep = num_return_sequences hypotheses per batch item): itts_gpt_fetch then returns the n best
 * hypotheses of every batch item, best first, as codes [B * n][max_gen].  1 <= n <= num_beams (checked by
 * itts_gpt_prefill, with HF's message); stays in force until changed.  (Without beams the caller repeats the rows, as HF
 * does: indextts/gpt/model.py.)  ABI 4. */
int itts_gpt_set_beam_returns(itts_engine* e, int num_return_sequences);

/* `typical_sampling=True` of UnifiedVoice.inference_speech (gpt/model.py:690-697): the reference's TypicalLogitsWarper
 * (utils/typical_sampling.py:9-30, mass in (0, 1); min_tokens_to_keep 2 under beams, else 1) runs right after the repetition
 * penalty and before Temperature / TopK / TopP in the sampling and beam-sample modes.  mass = 0 switches it off. */
int itts_gpt_set_typical(itts_engine* e, float mass);

/* Forced tokens for the first n steps of every following generation (n = 0 clears): ids_host int32 [B, n] (B = 1 is
 * broadcast to every row; -1 = leave that step free).  This is the `input_tokens` continuation of
 * UnifiedVoice.inference_speech (gpt/model.py:672-686: given mel tokens are appended to the prompt and generation
 * continues after them) and the teacher forcing the parity tests use; the forced tokens are returned by
 * itts_gpt_fetch as steps 0..n-1 (the reference strips them with trunc_index, model.py:687,704). */
int itts_gpt_set_forced(itts_engine* e, const int32_t* ids_host, int B, int n);

/* The same, with the positions of the reference's `input_tokens` path (gpt/model.py:141-155,672-686): the given tokens are
 * part of the reference's FIRST forward, embedded with mel positions 0 .. n ([start_mel, t1 .. tn]), and the first generated
 * token is fed at position n + 2 - so given token k (0-based) is fed at position k + 1 here, not k + 2 as a generated
 * (or teacher-forced) token would be.  n = 0 clears.  Works with beams too (itts_gpt_set_beams; B = batch items, every id
 * >= 0): the given tokens go into every beam row with the beam scores, the cache ancestry and the hypotheses untouched, and
 * HF's generated_len (length penalty, is_done) counts after them, as they belong to the decoder prompt there. */
int itts_gpt_set_input_tokens(itts_engine* e, const int32_t* ids_host, int B, int n);

/* How the last captured / launched decode step ran: 1 = the persistent decode engine (ONE launch per token step: the GPT
 * blocks, the head and - greedy search - the sampler; 1 - 6 decode rows, beam rows included (cache ancestry); IndexTTS-1.5
 * dims, a whole MI355X), 0 = five launches per block. */
int itts_gpt_decode_mode(itts_engine* e);

/* Host-side token choice, for generate() modes outside the device samplers' limits (HF warpers over the whole vocabulary:
 * `top_k = 0 / None`, infer.py:116-124 forwards it verbatim and webui.py:393-402 offers 0): with on = 1, itts_gpt_prefill and
 * itts_gpt_decode(e, 1, ..) stop behind the head GEMV; the caller reads the logits (itts_gpt_fetch), applies the logits
 * processors / warpers and the draw itself, and itts_gpt_commit hands the chosen token of every row (host int32 [B]) to the
 * sampler's bookkeeping (ids, repetition bitmap, eos state, step counter, next input embedding).  One stream sync per token. */
int itts_gpt_set_host_sampling(itts_engine* e, int on);

/* Conditioning latents per batch item: with on = 1 the `cond` of the following itts_gpt_prefill calls is [B][latents, D] (one
 * prompt per row, UnifiedVoice.inference_speech with a batch of prompts: gpt/model.py:599-602,670) instead of one [latents, D]
 * block shared by every row (what infer.py passes). */
int itts_gpt_set_cond_per_row(itts_engine* e, int on);
int itts_gpt_commit(itts_engine* e, const int32_t* tokens_host, itts_stream stream);

/* The same under beams (itts_gpt_set_host_sampling(e, 1) BEFORE itts_gpt_set_beams, which then accepts any top_k and no
 * uniforms): HF beam_sample with `top_k = 0 / None` or > 128 (infer.py:116-124 forwards the kwargs verbatim; the device beam
 * sampler keeps at most 128 candidates per beam).  Per step the caller reads the logits [B * num_beams, V] (itts_gpt_fetch) and
 * itts_gpt_beam_state - ids_host int32 [B * num_beams][max_gen]: every beam's id history (the first *step_host entries are
 * valid), scores_host [B * num_beams]: the running beam scores, done_host [B]: BeamSearchScorer._done - applies log_softmax ->
 * logits processors -> warpers -> + beam scores and draws 2 * num_beams candidates per batch item, then hands them IN DRAW
 * ORDER ([B][2 * num_beams]: score with the beam score included, token, beam index within the item) to
 * itts_gpt_commit_beams, which runs BeamSearchScorer.process (hypotheses, done) and the beam re-ordering (id histories, cache
 * ancestry = _reorder_cache, gpt/model.py:194-207) on the device as the device-sampled mode does.  itts_gpt_fetch finalizes. */
int itts_gpt_beam_state(itts_engine* e, int32_t* ids_host, float* scores_host, int32_t* done_host, int* step_host, itts_stream stream);
int itts_gpt_commit_beams(itts_engine* e, const float* pick_score_host, const int32_t* pick_tok_host, const int32_t* pick_beam_host,
                          itts_stream stream);

/* G1/G3/G4 step 0: prepare_gpt_inputs (model.py:591-654) + prefill + first greedy token.
 * cond fp32 [latents, D]; text ids host int32 [B, L] (may hold start/stop padding ids, stripped per row).
 * Synchronises the stream once (uploads the row descriptors). */
int itts_gpt_prefill(itts_engine* e, const float* cond, const int32_t* text_ids_host, int B, int L, int max_gen,
                     float repetition_penalty, int suppress_stop, itts_stream stream);
/* G4/G5/G6: run up to nsteps further greedy steps (hipGraph replay of one captured step). */
int itts_gpt_decode(itts_engine* e, int nsteps, itts_stream stream);
/* Host-visible progress (synchronises): tokens generated per row so far, rows still unfinished. */
int itts_gpt_status(itts_engine* e, int* steps_done_host, int* n_unfinished_host, itts_stream stream);
/* Copy generated codes (int32 [B, max_gen], pad = stop token) and optionally the last logits fp32 [B, V]. */
int itts_gpt_fetch(itts_engine* e, int32_t* codes_host, float* logits_host, itts_stream stream);

/* G8  UnifiedVoice.forward(return_latent=True) for one sentence (model.py:521-589):
 * text ids host [L], codes host [T] -> latent [T, D] in the engine dtype. */
int itts_gpt_latent(itts_engine* e, const float* cond, const int32_t* text_ids_host, int L, const int32_t* codes_host,
                    int T, void* latent_out, itts_stream stream);

/* Same for several sentences at once (concatenated ids / codes with per-sentence lengths): rows are stacked with
 * left padding and masked keys, so each sentence's latent equals the batch-1 result.  latent_out = [sum T_i, D]. */
int itts_gpt_latent_batch(itts_engine* e, const float* cond, const int32_t* text_ids_host, const int32_t* text_lens_host,
                          const int32_t* codes_host, const int32_t* code_lens_host, int nseq, void* latent_out,
                          itts_stream stream);

/* V1-V5  BigVGAN.forward given the speaker embedding (BigVGAN/models.py:201-250):
 * latent [B, T, gpt_dim] (engine dtype), spk fp32 [B, spk_dim] -> wav fp32 [B, T * prod(up_rates)] */
int itts_bigvgan(itts_engine* e, const void* latent, const float* spk, int B, int T, float* wav_out, itts_stream stream);

/* Q1  DiscreteVAE.decode (vqvae/xtts_dvae.py:332-351): codes host int32 [B, T] -> mel [B, 4T, channels] engine dtype */
int itts_dvae_decode(itts_engine* e, const int32_t* codes_host, int B, int T, void* mel_out, itts_stream stream);

/* DiscreteVAE.get_codebook_indices (vqvae/xtts_dvae.py:325-330; Quantize.forward distance arg-min :86-92):
 * mel [B, T, channels] engine dtype -> codes host int32 [B, T'], T' = T halved (rounding up) once per stride-2 layer.
 * Needs the encoder tensors of the checkpoint (dvae.enc* / dvae.erb* / dvae.eout / dvae.codebook_sq). Synchronises. */
int itts_dvae_encode(itts_engine* e, const void* mel_btc, int B, int T, int32_t* codes_host, itts_stream stream);

/* Debug/testing: copy a named intermediate of the LAST call into host memory (fp32), returns element count. */
int64_t itts_debug_fetch(itts_engine* e, const char* name, float* out_host, int64_t max_elems);
int itts_debug_enable(itts_engine* e, int on);

#ifdef __cplusplus
}
#endif
#endif
