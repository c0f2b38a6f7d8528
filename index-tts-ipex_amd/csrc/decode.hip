// Kernels of the autoregressive decode step (GPT2InferenceModel.forward single-token path,
// /root/reference/indextts/gpt/model.py:151-155 + HF GPT2Block + greedy_search bookkeeping).
// The per-token loop is HBM-bound on the weight stream: every kernel here reads its
// weights exactly once per step with 16-byte lane loads, keeps all per-sequence state on the device
// (step counter, ids, seen-bitmap, finished flags) so a step needs no host round trip and the whole
// step replays from a hipGraph.
#include "itts_decode.h"

namespace itts {
namespace {

// ---------------------------------------------------------------------------------------------
// h[b] = mel_emb[tok[b]] + mel_pos[step + 1]      (positions 0, 2, 3, ... : model.py:153-155)
// ---------------------------------------------------------------------------------------------
template <typename TW>
__global__ void decode_embed_kernel(float* __restrict__ h, const TW* __restrict__ emb, const TW* __restrict__ pos,
                                    const int* __restrict__ tok, const int* __restrict__ step, int D) {
  const int b = blockIdx.x;
  const int t = tok[b];
  const int p = step[0] + 1;
  for (int i = threadIdx.x; i < D; i += blockDim.x)
    h[(size_t)b * D + i] = ldf(emb + (size_t)t * D + i) + ldf(pos + (size_t)p * D + i);
}

// ---------------------------------------------------------------------------------------------
// Batched GEMV:  Y[b, n] (+)= act( LN?(X[b, :]) . W[n, :] + bias[n] ),  X fp32 [B, K], W [N, K]
// One wave per RPW output rows; lanes stride K with 16-byte weight loads; optional fused LayerNorm of X
// (stats recomputed per block from L2-resident X: B*K*4 bytes), optional accumulate into Y (residual).
// ---------------------------------------------------------------------------------------------
template <typename TW> struct Vec8;
template <> struct Vec8<bf16_t> {
  uint4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ float get(int i) const {
    const uint32_t w = (&raw.x)[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xFFFF0000u) : (w << 16));
  }
};
template <> struct Vec8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ float get(int i) const { return i < 4 ? (&a.x)[i] : (&b.x)[i - 4]; }
};

constexpr int GEMV_RPW = 2;     // output rows per wave
constexpr int GEMV_WAVES = 4;   // waves per block

template <typename TW, int NB>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs g) {
  __shared__ float s_mean[NB], s_rstd[NB];
  __shared__ float s_red[GEMV_WAVES][NB][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ X = g.X;
  const TW* __restrict__ W = (const TW*)g.W;
  const int K = g.K;
  const bool ln = g.ln_gamma != nullptr;
  if (ln) {
    // block-wide LayerNorm statistics of each of the NB rows
    float sum[NB], sq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) sum[b] = sq[b] = 0.f;
    for (int i = tid; i < K; i += 256)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (b < g.B) sum[b] += X[(size_t)b * K + i];
#pragma unroll
    for (int b = 0; b < NB; ++b) sum[b] = wave_sum(sum[b]);
    if (lane == 0)
#pragma unroll
      for (int b = 0; b < NB; ++b) s_red[wave][b][0] = sum[b];
    __syncthreads();
    float mean[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) mean[b] = (s_red[0][b][0] + s_red[1][b][0] + s_red[2][b][0] + s_red[3][b][0]) / K;
    for (int i = tid; i < K; i += 256)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (b < g.B) {
          const float d = X[(size_t)b * K + i] - mean[b];
          sq[b] = fmaf(d, d, sq[b]);
        }
#pragma unroll
    for (int b = 0; b < NB; ++b) sq[b] = wave_sum(sq[b]);
    if (lane == 0)
#pragma unroll
      for (int b = 0; b < NB; ++b) s_red[wave][b][1] = sq[b];
    __syncthreads();
    if (tid < NB) {
      s_mean[tid] = (s_red[0][tid][0] + s_red[1][tid][0] + s_red[2][tid][0] + s_red[3][tid][0]) / K;
      s_rstd[tid] = rsqrtf((s_red[0][tid][1] + s_red[1][tid][1] + s_red[2][tid][1] + s_red[3][tid][1]) / K + g.ln_eps);
    }
    __syncthreads();
  }
  const int n0 = (blockIdx.x * GEMV_WAVES + wave) * GEMV_RPW;
  if (n0 >= g.N) return;
  float acc[GEMV_RPW][NB];
#pragma unroll
  for (int r = 0; r < GEMV_RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
    Vec8<TW> w[GEMV_RPW];
#pragma unroll
    for (int r = 0; r < GEMV_RPW; ++r) {
      const int n = min(n0 + r, g.N - 1);
      w[r].load(W + (size_t)n * K + k0);
    }
    float gam[8], bet[8];
    if (ln) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        gam[i] = g.ln_gamma[k0 + i];
        bet[i] = g.ln_beta[k0 + i];
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= g.B) break;
      Vec8<float> x;
      x.load(X + (size_t)b * K + k0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float xv = x.get(i);
        if (ln) xv = (xv - s_mean[b]) * s_rstd[b] * gam[i] + bet[i];
#pragma unroll
        for (int r = 0; r < GEMV_RPW; ++r) acc[r][b] = fmaf(xv, w[r].get(i), acc[r][b]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < GEMV_RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < GEMV_RPW; ++r) {
      const int n = n0 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= g.B) break;
        float v = acc[r][b] + (g.bias ? g.bias[n] : 0.f);
        v = act_apply(g.act, v);
        float* y = g.Y + (size_t)b * g.ldy + n;
        *y = g.accumulate ? (*y + v) : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Single-query attention over the KV cache, with the append of this step's K/V fused in.
// grid (H, B), 256 threads.  qkv fp32 [B, 3D]; cache [B][H][Smax][dh] (dh == 64).
// ---------------------------------------------------------------------------------------------
template <typename TC>
__global__ __launch_bounds__(256) void decode_attn_kernel(float* __restrict__ ctx, const float* __restrict__ qkv,
                                                          TC* __restrict__ kc, TC* __restrict__ vc,
                                                          const int* __restrict__ step, const int* __restrict__ kv_start,
                                                          const int* __restrict__ prefix, int H, int Smax, float scale) {
  constexpr int DH = 64;
  __shared__ float sq[DH];
  __shared__ float sp[2048];          // scores / probabilities (Smax <= 2048, enforced on host)
  __shared__ float sred[4];
  __shared__ float so[4][DH];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * DH;
  const int pos = prefix[0] + step[0];  // index of the key appended now (prefix rows 0..s-1, start token at s)
  const int S = pos + 1;
  TC* kb = kc + ((size_t)b * H + h) * Smax * DH;
  TC* vb = vc + ((size_t)b * H + h) * Smax * DH;
  const float* qv = qkv + (size_t)b * 3 * D + h * DH;
  if (tid < DH) {
    sq[tid] = qv[tid] * scale;
    stf(kb + (size_t)pos * DH + tid, qv[D + tid]);
    stf(vb + (size_t)pos * DH + tid, qv[2 * D + tid]);
  }
  __syncthreads();
  const int ks = kv_start[b];
  float mx = -INFINITY;
  for (int j = tid; j < S; j += 256) {
    float sc = -INFINITY;
    if (j >= ks) {
      sc = 0.f;
      if (j == pos) {
#pragma unroll 8
        for (int d = 0; d < DH; ++d) sc = fmaf(sq[d], (float)(TC)qv[D + d], sc);  // same rounding as the cache
      } else {
        const TC* kr = kb + (size_t)j * DH;
#pragma unroll 8
        for (int d = 0; d < DH; ++d) sc = fmaf(sq[d], ldf(kr + d), sc);
      }
    }
    sp[j] = sc;
    mx = fmaxf(mx, sc);
  }
  mx = wave_max(mx);
  if (lane == 0) sred[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < S; j += 256) {
    const float p = sp[j] > -INFINITY ? __expf(sp[j] - mx) : 0.f;
    sp[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if (lane == 0) sred[wave] = sum;
  __syncthreads();
  const float inv = 1.f / (sred[0] + sred[1] + sred[2] + sred[3]);
  // o[d] = sum_j p_j v_j[d]; lane <-> d, waves split the keys
  float o = 0.f;
  for (int j = wave; j < S; j += 4) {
    const float vv = (j == pos) ? (float)(TC)qv[2 * D + lane] : ldf(vb + (size_t)j * DH + lane);
    o = fmaf(sp[j], vv, o);
  }
  so[wave][lane] = o;
  __syncthreads();
  if (tid < DH) ctx[(size_t)b * D + h * DH + tid] = (so[0][tid] + so[1][tid] + so[2][tid] + so[3][tid]) * inv;
}

// ---------------------------------------------------------------------------------------------
// y = LN_b(LN_a(x)) per row (gpt.ln_f followed by final_norm: model.py:473-474 / lm_head :48)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void double_ln_kernel(float* __restrict__ y, const float* __restrict__ x,
                                                        const float* __restrict__ g1, const float* __restrict__ b1,
                                                        const float* __restrict__ g2, const float* __restrict__ b2,
                                                        int D, float eps) {
  __shared__ float buf[2048];
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xr = x + (size_t)row * D;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  for (int pass = 0; pass < 2; ++pass) {
    const float* g = pass ? g2 : g1;
    const float* bb = pass ? b2 : b1;
    float s = 0.f;
    for (int i = tid; i < D; i += 256) s += pass ? buf[i] : xr[i];
    const float mean = block_sum(s) / D;
    float v = 0.f;
    for (int i = tid; i < D; i += 256) {
      const float d = (pass ? buf[i] : xr[i]) - mean;
      v = fmaf(d, d, v);
    }
    const float rstd = rsqrtf(block_sum(v) / D + eps);
    for (int i = tid; i < D; i += 256) {
      const float o = ((pass ? buf[i] : xr[i]) - mean) * rstd * g[i] + bb[i];
      if (pass) y[(size_t)row * D + i] = o; else buf[i] = o;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// greedy_search bookkeeping on device (HF 4.36.2 semantics): RepetitionPenaltyLogitsProcessor over every id
// seen so far (fake prefix id 1, start token, generated codes), argmax (lowest index wins ties, like
// torch.argmax), finished rows emit pad(=stop), append, update unfinished; thread 0 of block 0 bumps the
// step counter LAST via a second tiny kernel (so every block of this kernel reads the same step).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sampler_kernel(SamplerArgs a) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lg = a.logits + (size_t)b * a.V;
  uint8_t* seen = a.seen + (size_t)b * a.V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < a.V; i += 256) {
    float v = lg[i];
    if (a.penalty != 1.f && seen[i]) v = v < 0.f ? v * a.penalty : v / a.penalty;
    if (a.suppress_stop && i == a.stop) v = -INFINITY;
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    const int k = a.step[0];
    const int unf = a.unfinished[b];
    int tok = unf ? bi : a.stop;
    if (k >= a.max_gen) return;  // graph replays past the end are no-ops
    a.ids[(size_t)b * a.max_gen + k] = tok;
    a.cur_tok[b] = tok;
    seen[tok] = 1;
    const int nu = unf && tok != a.stop;
    a.unfinished[b] = nu;
    if (nu) atomicAdd(a.n_unfinished_next, 1);
  }
}

__global__ void step_advance_kernel(int* step, int* n_unf, int* n_unf_next, int max_gen) {
  if (step[0] < max_gen) step[0] += 1;
  n_unf[0] = n_unf_next[0];
  n_unf_next[0] = 0;
}

// prefill K/V (rows of the fused qkv buffer) -> cache layout [B][H][Smax][dh]
template <typename TQ, typename TC>
__global__ void kv_scatter_kernel(TC* __restrict__ kc, TC* __restrict__ vc, const TQ* __restrict__ qkv, int B, int S,
                                  int H, int dh, int Smax) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int D = H * dh;
  if (i >= (long)B * S * D) return;
  const int d = (int)(i % dh);
  const int h = (int)((i / dh) % H);
  const int s = (int)((i / D) % S);
  const long b = i / ((long)D * S);
  const TQ* row = qkv + (b * S + s) * 3 * D;
  const size_t o = ((b * H + h) * (size_t)Smax + s) * dh + d;
  stf(kc + o, ldf(row + D + h * dh + d));
  stf(vc + o, ldf(row + 2 * D + h * dh + d));
}

}  // namespace

int decode_embed(float* h, const void* emb, const void* pos, const int* tok, const int* step, int B, int D, int tw,
                 hipStream_t s) {
  if (tw == F32)
    hipLaunchKernelGGL(decode_embed_kernel<float>, dim3(B), dim3(256), 0, s, h, (const float*)emb, (const float*)pos, tok, step, D);
  else
    hipLaunchKernelGGL(decode_embed_kernel<bf16_t>, dim3(B), dim3(256), 0, s, h, (const bf16_t*)emb, (const bf16_t*)pos, tok, step, D);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

template <typename TW>
static int gemv_launch(const GemvArgs& g, hipStream_t s) {
  const int rows_per_block = GEMV_RPW * GEMV_WAVES;
  dim3 grid((g.N + rows_per_block - 1) / rows_per_block), blk(256);
  // batches larger than 8 are processed in chunks of 8 rows (weights re-streamed per chunk; the MFMA
  // skinny-GEMM path takes over for large batches)
  for (int b0 = 0; b0 < g.B; b0 += 8) {
    GemvArgs c = g;
    c.B = g.B - b0 < 8 ? g.B - b0 : 8;
    c.X = g.X + (size_t)b0 * g.K;
    c.Y = g.Y + (size_t)b0 * g.ldy;
    if (c.B == 1) hipLaunchKernelGGL((gemv_kernel<TW, 1>), grid, blk, 0, s, c);
    else if (c.B == 2) hipLaunchKernelGGL((gemv_kernel<TW, 2>), grid, blk, 0, s, c);
    else if (c.B <= 4) hipLaunchKernelGGL((gemv_kernel<TW, 4>), grid, blk, 0, s, c);
    else hipLaunchKernelGGL((gemv_kernel<TW, 8>), grid, blk, 0, s, c);
  }
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int gemv(const GemvArgs& g, int tw, hipStream_t s) {
  ITTS_REQUIRE(g.X && g.W && g.Y && g.B > 0 && g.N > 0, "gemv: bad args");
  ITTS_REQUIRE(g.K % 8 == 0, "gemv: K must be a multiple of 8");
  return tw == F32 ? gemv_launch<float>(g, s) : gemv_launch<bf16_t>(g, s);
}

int decode_attn(float* ctx, const float* qkv, void* kc, void* vc, const int* step, const int* kv_start,
                const int* prefix, int B, int H, int dh, int Smax, int tc, hipStream_t s) {
  ITTS_REQUIRE(dh == 64, "decode_attn: head dim must be 64");
  ITTS_REQUIRE(Smax <= 2048, "decode_attn: Smax > 2048");
  const float scale = 1.f / sqrtf((float)dh);
  if (tc == F32)
    hipLaunchKernelGGL(decode_attn_kernel<float>, dim3(H, B), dim3(256), 0, s, ctx, qkv, (float*)kc, (float*)vc, step, kv_start, prefix, H, Smax, scale);
  else
    hipLaunchKernelGGL(decode_attn_kernel<bf16_t>, dim3(H, B), dim3(256), 0, s, ctx, qkv, (bf16_t*)kc, (bf16_t*)vc, step, kv_start, prefix, H, Smax, scale);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int double_ln(float* y, const float* x, const float* g1, const float* b1, const float* g2, const float* b2, int rows,
              int D, float eps, hipStream_t s) {
  ITTS_REQUIRE(D <= 2048, "double_ln: D > 2048");
  hipLaunchKernelGGL(double_ln_kernel, dim3(rows), dim3(256), 0, s, y, x, g1, b1, g2, b2, D, eps);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int sampler_step(const SamplerArgs& a, int B, hipStream_t s) {
  hipLaunchKernelGGL(sampler_kernel, dim3(B), dim3(256), 0, s, a);
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, a.step, a.n_unfinished, a.n_unfinished_next, a.max_gen);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int kv_scatter(void* kc, void* vc, const void* qkv, int B, int S, int H, int dh, int Smax, int tq, int tc, hipStream_t s) {
  const long n = (long)B * S * H * dh;
  dim3 grid((unsigned)((n + 255) / 256)), blk(256);
  if (tq == F32 && tc == F32)
    hipLaunchKernelGGL((kv_scatter_kernel<float, float>), grid, blk, 0, s, (float*)kc, (float*)vc, (const float*)qkv, B, S, H, dh, Smax);
  else if (tq == BF16 && tc == BF16)
    hipLaunchKernelGGL((kv_scatter_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (bf16_t*)kc, (bf16_t*)vc, (const bf16_t*)qkv, B, S, H, dh, Smax);
  else {
    set_error("kv_scatter: dtype");
    return E_INVALID;
  }
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
