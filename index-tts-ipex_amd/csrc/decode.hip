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
    return (i & 1) ? half_hi(w) : half_lo(w);
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
  const bool ln = g.prologue == 1;
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
        gam[i] = g.ln_gamma ? g.ln_gamma[k0 + i] : 1.f;
        bet[i] = g.ln_beta ? g.ln_beta[k0 + i] : 0.f;
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
      float o = ((pass ? buf[i] : xr[i]) - mean) * rstd;
      if (g) o = o * g[i] + bb[i];
      if (pass) y[(size_t)row * D + i] = o; else buf[i] = o;
    }
    __syncthreads();
  }
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
  ITTS_REQUIRE(g.prologue == 0 || g.prologue == 1, "gemv (v1): only plain / LayerNorm prologues");
  ITTS_REQUIRE(g.K % 8 == 0, "gemv: K must be a multiple of 8");
  return tw == F32 ? gemv_launch<float>(g, s) : gemv_launch<bf16_t>(g, s);
}

int double_ln(float* y, const float* x, const float* g1, const float* b1, const float* g2, const float* b2, int rows,
              int D, float eps, hipStream_t s) {
  ITTS_REQUIRE(D <= 2048, "double_ln: D > 2048");
  hipLaunchKernelGGL(double_ln_kernel, dim3(rows), dim3(256), 0, s, y, x, g1, b1, g2, b2, D, eps);
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
