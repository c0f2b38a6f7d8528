// Tiled online-softmax attention on the vector ALU (fp32 math), any head dim <= 128 / value dim <= 64.
// Used by the conformer (rel-pos, dqk = 2*dk), the perceiver cross-attention, and the GPT prefill /
// latent pass (causal + left-padding mask).  K/V tiles of 64 keys are staged once in LDS per 16 queries.
#include "itts_kernels.h"

namespace itts {
namespace {

constexpr int QB = 16;   // queries per block (4 per wave)
constexpr int KT = 64;   // keys per tile (one per lane)
constexpr int MAXDQK = 128, MAXDV = 64;

template <typename T>
__global__ __launch_bounds__(256) void attn_simple_kernel(AttnArgs a) {
  __shared__ float Ks[KT][MAXDQK + 1];
  __shared__ float Vs[KT][MAXDV];
  __shared__ float Qs[QB][MAXDQK];
  __shared__ float Ps[QB][KT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
  const T* __restrict__ q = (const T*)a.q;
  const T* __restrict__ k = (const T*)a.k;
  const T* __restrict__ v = (const T*)a.v;
  T* __restrict__ o = (T*)a.o;
  const int kvs = a.kv_start ? a.kv_start[b] : 0;
  const int shift = a.Sk - a.Sq;

  for (int i = tid; i < QB * a.dqk; i += 256) {
    const int r = i / a.dqk, d = i - r * a.dqk;
    const int qi = q0 + r;
    Qs[r][d] = qi < a.Sq ? ldf(q + ((size_t)b * a.Sq + qi) * a.ldq + h * a.dqk + d) * a.scale : 0.f;
  }
  float m[4], l[4], acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
    acc[r] = 0.f;
  }
  int kend = a.Sk;
  if (a.causal) kend = min(a.Sk, q0 + QB - 1 + shift + 1);
  for (int j0 = 0; j0 < kend; j0 += KT) {
    __syncthreads();
    for (int i = tid; i < KT * a.dqk; i += 256) {
      const int r = i / a.dqk, d = i - r * a.dqk;
      const int j = j0 + r;
      Ks[r][d] = j < a.Sk ? ldf(k + ((size_t)b * a.Sk + j) * a.ldk + h * a.dqk + d) : 0.f;
    }
    for (int i = tid; i < KT * a.dv; i += 256) {
      const int r = i / a.dv, d = i - r * a.dv;
      const int j = j0 + r;
      Vs[r][d] = j < a.Sk ? ldf(v + ((size_t)b * a.Sk + j) * a.ldv + h * a.dv + d) : 0.f;
    }
    __syncthreads();
    const int j = j0 + lane;
    float corr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 4 + r;
      const int qi = q0 + row;
      float sc = 0.f;
      for (int d = 0; d < a.dqk; ++d) sc = fmaf(Qs[row][d], Ks[lane][d], sc);
      bool ok = j < a.Sk && j >= kvs && qi < a.Sq;
      if (a.causal) ok = ok && (j <= qi + shift);
      sc = ok ? sc : -INFINITY;
      const float mn = fmaxf(m[r], wave_max(sc));
      float p = 0.f;
      corr[r] = 1.f;
      if (mn > -INFINITY) {
        p = ok ? __expf(sc - mn) : 0.f;
        corr[r] = m[r] > -INFINITY ? __expf(m[r] - mn) : 0.f;
      }
      l[r] = l[r] * corr[r] + wave_sum(p);
      m[r] = mn;
      Ps[row][lane] = p;
    }
    __syncthreads();
    if (lane < a.dv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 4 + r;
        float s = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < KT; ++jj) s = fmaf(Ps[row][jj], Vs[jj][lane], s);
        acc[r] = acc[r] * corr[r] + s;
      }
    }
  }
  if (lane < a.dv) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + wave * 4 + r;
      if (qi < a.Sq) stf(o + ((size_t)b * a.Sq + qi) * a.ldo + h * a.dv + lane, l[r] > 0.f ? acc[r] / l[r] : 0.f);
    }
  }
}

}  // namespace

int attention_simple(const AttnArgs& a, int dt, hipStream_t s) {
  ITTS_REQUIRE(a.q && a.k && a.v && a.o, "attention: null pointer");
  ITTS_REQUIRE(a.dqk > 0 && a.dqk <= MAXDQK && a.dv > 0 && a.dv <= MAXDV, "attention: head dims out of range");
  ITTS_REQUIRE(a.Sq > 0 && a.Sk > 0 && a.B > 0 && a.H > 0, "attention: bad dims");
  dim3 grid((a.Sq + QB - 1) / QB, a.H, a.B);
  if (dt == F32)
    hipLaunchKernelGGL(attn_simple_kernel<float>, grid, dim3(256), 0, s, a);
  else if (dt == BF16)
    hipLaunchKernelGGL(attn_simple_kernel<bf16_t>, grid, dim3(256), 0, s, a);
  else {
    set_error("attention: dtype");
    return E_INVALID;
  }
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
