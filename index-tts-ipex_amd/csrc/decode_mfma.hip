// Decode step at larger batches (B > 4 rows): the projections are "skinny" GEMMs Y[B,N] = X[B,K] W[N,K]^T that are
// still bound by the weight stream, so the weights are read exactly once and the batch rides on the matrix cores.
// One workgroup = 16 output features x all batch rows; its 8 waves split K (interleaved 32-wide k-steps, so the
// 8 waves together read 512 contiguous bytes of every weight row), each wave runs v_mfma_f32_16x16x32_bf16 with
// A = X tile (16 batch rows), B = W tile (16 features), and the 8 partial tiles are summed through LDS in a fixed
// order (deterministic, independent of the other rows in the batch).  LayerNorm runs as its own row kernel here
// (bf16 output; the affine is folded into the projection by the packer) - at B >= 8 recomputing it per workgroup
// as the GEMV path does would cost more than the launch.
#include "itts_decode.h"
#include "itts_kernels.h"

namespace itts {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int SW = 8;  // waves per workgroup (K split)
constexpr int SU = 5;  // k-steps in flight per wave

template <int BT>
__global__ __launch_bounds__(512) void skinny_mfma_kernel(GemvArgs g) {
  __shared__ float red[SW][BT][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W;
  const bf16_t* __restrict__ X = (const bf16_t*)g.X;
  const int K = g.K, nks = K >> 5;
  const bf16_t* wp = W + (size_t)min(n0 + fr, g.N - 1) * K + fg * 8;
  const bf16_t* xp[BT];
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) xp[bt] = X + (size_t)min(bt * 16 + fr, g.B - 1) * K + fg * 8;
  f32x4v acc[BT];
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) acc[bt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (int k0 = wave; k0 < nks; k0 += SW * SU) {
    bf16x8 wf[SU], xf[SU][BT];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = min(k0 + u * SW, nks - 1);  // clamped loads, masked below: keeps the loads branch-free
      wf[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp + (size_t)ks * 32));
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = min(k0 + u * SW, nks - 1);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) xf[u][bt] = *reinterpret_cast<const bf16x8*>(xp[bt] + (size_t)ks * 32);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      if (k0 + u * SW < nks) {
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) acc[bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[u][bt], wf[u], acc[bt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int bt = 0; bt < BT; ++bt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][bt][r * 64 + lane] = acc[bt][r];
  __syncthreads();
  for (int idx = tid; idx < BT * 256; idx += 512) {
    const int bt = idx >> 8, e = idx & 255, r = e >> 6, l = e & 63;
    const int b = bt * 16 + (l >> 4) * 4 + r, n = n0 + (l & 15);
    if (b >= g.B || n >= g.N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) v += red[w][bt][e];
    if (g.bias) v += g.bias[n];
    v = act_apply(g.act, v);
    const size_t o = (size_t)b * g.ldy + n;
    if (g.y_bf16)
      reinterpret_cast<bf16_t*>(g.Y)[o] = (bf16_t)v;
    else if (g.accumulate)
      g.Y[o] += v;
    else
      g.Y[o] = v;
  }
}

// y(bf16) = LN(x) [-> LN again for the head]; optional affine on the first pass
__global__ __launch_bounds__(256) void ln_rows_bf16_kernel(bf16_t* __restrict__ y, const float* __restrict__ x,
                                                           const float* __restrict__ g1, const float* __restrict__ b1,
                                                           int D, float eps, int passes) {
  __shared__ float red[2][4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int MAXE = 8;  // D <= 2048
  float v[MAXE];
  const float* xr = x + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) v[i] = tid + i * 256 < D ? xr[tid + i * 256] : 0.f;
  for (int pass = 0; pass < passes; ++pass) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) s += v[i];
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mean = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const float d = tid + i * 256 < D ? v[i] - mean : 0.f;
      q = fmaf(d, d, q);
    }
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    const float rstd = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / D + eps);
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int c = tid + i * 256;
      if (c < D) {
        float o = (v[i] - mean) * rstd;
        if (pass == 0 && g1) o = o * g1[c] + b1[c];
        v[i] = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXE; ++i)
    if (tid + i * 256 < D) y[(size_t)row * D + tid + i * 256] = (bf16_t)v[i];
}

}  // namespace

bool skinny_mfma_supported(const GemvArgs& g) {
  return g.B >= 1 && g.B <= 128 && g.K % 32 == 0 && g.x_bf16 && g.prologue == 0 && !(g.accumulate && g.y_bf16) &&
         !(((uintptr_t)g.W | (uintptr_t)g.X) & 15);
}

int skinny_mfma(const GemvArgs& g, hipStream_t s) {
  ITTS_REQUIRE(g.X && g.W && g.Y && g.N > 0, "skinny_mfma: bad args");
  ITTS_REQUIRE(skinny_mfma_supported(g), "skinny_mfma: unsupported shape");
  const int bt = (g.B + 15) / 16;
  dim3 grid((g.N + 15) / 16), blk(512);
  if (bt <= 1)
    hipLaunchKernelGGL(skinny_mfma_kernel<1>, grid, blk, 0, s, g);
  else if (bt <= 2)
    hipLaunchKernelGGL(skinny_mfma_kernel<2>, grid, blk, 0, s, g);
  else if (bt <= 4)
    hipLaunchKernelGGL(skinny_mfma_kernel<4>, grid, blk, 0, s, g);
  else
    hipLaunchKernelGGL(skinny_mfma_kernel<8>, grid, blk, 0, s, g);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int ln_rows_bf16(void* y, const float* x, const float* g1, const float* b1, int rows, int D, float eps, int passes,
                 hipStream_t s) {
  ITTS_REQUIRE(D <= 2048 && passes >= 1 && passes <= 2, "ln_rows_bf16: D > 2048 or bad pass count");
  hipLaunchKernelGGL(ln_rows_bf16_kernel, dim3(rows), dim3(256), 0, s, (bf16_t*)y, x, g1, b1, D, eps, passes);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
