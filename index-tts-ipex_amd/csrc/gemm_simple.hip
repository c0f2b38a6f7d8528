// Generic shift-GEMM on the vector ALU (exact fp32 FMA chain).  This is the fp32 parity path and the
// fallback for shapes the MFMA kernel does not take (odd channel counts); see gemm_mfma.hip for the
// bf16 matrix-core path.  One kernel covers nn.Linear, Conv1d (zero/reflect pad, dilation), the
// polyphase form of ConvTranspose1d and the nearest-upsample+conv of the DVAE decoder.
#include "itts_kernels.h"

namespace itts {

namespace {

constexpr int BM = 64, BN = 64, BK = 16;

__device__ __forceinline__ int reflect_idx(int t, int T) {
  // torch 'reflect' padding (no edge repeat); valid for |overhang| < T
  if (t < 0) t = -t;
  if (t >= T) t = 2 * (T - 1) - t;
  return t;
}

template <typename TA, typename TW, typename TC>
__global__ __launch_bounds__(256) void gemm_simple_kernel(GemmArgs g) {
  __shared__ float As[BK][BM + 4];
  __shared__ float Ws[BK][BN + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int phase = blockIdx.z;
  const TA* __restrict__ A = (const TA*)g.A;
  const int K = g.taps * g.Cin;
  const TW* __restrict__ W = (const TW*)g.W + (size_t)phase * g.N * K;
  const int T = g.T > 0 ? g.T : g.M;
  const int Tin = T / g.in_up;  // source rows per batch item

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  // loader mapping: 16 consecutive threads read 16 consecutive channels (coalesced), 16 row groups
  const int lc = tid & 15, lr = tid >> 4;

  for (int tap = 0; tap < g.taps; ++tap) {
    const int off = g.phase_shift[phase] + tap * g.dil - g.pad_left;
    // source row of each of this thread's 4 loader rows (independent of the channel block)
    long arow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = m0 + lr + 16 * p;
      long r = -1;
      if (m < g.M) {
        const int b = m / T, t = m - b * T;
        int ts = t + off;
        if (g.pad_mode == PAD_REFLECT) ts = reflect_idx(ts, T);
        if (ts >= 0 && ts < T) r = (long)b * Tin + ts / g.in_up;
      }
      arow[p] = r;
    }
    for (int c0 = 0; c0 < g.Cin; c0 += BK) {
      const int c = c0 + lc;
      const bool cok = c < g.Cin;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float v = 0.f;
        if (cok && arow[p] >= 0) v = ldf(A + arow[p] * g.lda + c);
        As[lc][lr + 16 * p] = v;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int n = n0 + lr + 16 * p;
        float v = 0.f;
        if (cok && n < g.N) v = ldf(W + (size_t)n * K + (size_t)tap * g.Cin + c);
        Ws[lc][lr + 16 * p] = v;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) {
        const float4 a4 = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
        const float4 w4 = *reinterpret_cast<const float4*>(&Ws[kk][tx * 4]);
        const float a[4] = {a4.x, a4.y, a4.z, a4.w};
        const float w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
      }
      __syncthreads();
    }
  }

  TC* __restrict__ C = (TC*)g.C;
  const TC* __restrict__ R = (const TC*)g.R;
  const TC* __restrict__ ADD = (const TC*)g.ADD;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= g.M) continue;
    const int b = m / T;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= g.N) continue;
      const int col = phase * g.N + n;
      float v = acc[i][j];
      if (g.bias) v += g.bias[(size_t)b * g.bias_bstride + n];
      v = act_apply(g.act, v);
      if (g.scale) v *= g.scale[n];
      if (g.shift) v += g.shift[n];
      v = act_apply(g.act2, v);
      if (R) v += ldf(R + (size_t)m * g.ldr + col);
      v *= g.alpha;
      if (ADD) v += g.beta * ldf(ADD + (size_t)m * g.ldadd + col);
      stf(C + (size_t)m * g.ldc + col, v);
    }
  }
}

template <typename TA, typename TW, typename TC>
int launch(const GemmArgs& g, hipStream_t s) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nphase);
  hipLaunchKernelGGL((gemm_simple_kernel<TA, TW, TC>), grid, dim3(256), 0, s, g);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace

int gemm_simple(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s) {
  ITTS_REQUIRE(g.A && g.W && g.C, "gemm: null pointer");
  ITTS_REQUIRE(g.M > 0 && g.N > 0 && g.Cin > 0 && g.taps > 0, "gemm: bad dims");
  ITTS_REQUIRE(g.nphase >= 1 && g.nphase <= 8 && g.in_up >= 1, "gemm: bad phase/upsample");
  ITTS_REQUIRE(g.lda >= g.Cin && g.ldc >= g.N * g.nphase, "gemm: bad leading dims");
  const int T = g.T > 0 ? g.T : g.M;
  ITTS_REQUIRE(g.M % T == 0 && T % g.in_up == 0, "gemm: M must be a multiple of T");
  if (ta == F32 && tw == F32 && tc == F32) return launch<float, float, float>(g, s);
  if (ta == BF16 && tw == BF16 && tc == BF16) return launch<bf16_t, bf16_t, bf16_t>(g, s);
  if (ta == BF16 && tw == BF16 && tc == F32) return launch<bf16_t, bf16_t, float>(g, s);
  if (ta == F32 && tw == F32 && tc == BF16) return launch<float, float, bf16_t>(g, s);
  if (ta == F32 && tw == BF16 && tc == F32) return launch<float, bf16_t, float>(g, s);
  if (ta == F32 && tw == BF16 && tc == BF16) return launch<float, bf16_t, bf16_t>(g, s);
  set_error("gemm_simple: unsupported dtype combination");
  return E_INVALID;
}

}  // namespace itts
