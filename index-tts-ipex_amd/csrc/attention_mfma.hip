// bf16 flash attention on the matrix cores (head dim 64): GPT prefill / latent pass (causal, left-padding mask)
// and the perceiver cross-attention.  One workgroup = 4 waves = 64 queries (16 per wave); keys/values are consumed in
// tiles of 64 through LDS: K row-major (B operand of S = Q K^T is k-contiguous), V transposed on the way in (B operand
// of O = P V needs 8 consecutive keys per lane).  Softmax is the online form kept in the MFMA accumulator layout:
// a lane owns rows (lane>>4)*4+r, so row max / sum are DPP reductions over the 16-lane row - no LDS.  P goes through
// a wave-private LDS tile to become the A operand of the second MFMA.  fp32 accumulation, bf16 P (flash practice).
#include <cstdlib>

#include "itts_kernels.h"

namespace itts {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 64, QT = 64, KT = 64, KROW = 72;  // LDS row stride in bf16 (144 B: conflict-free for ds_read_b128)
// DQK = query/key width: 64 (GPT, perceiver) or 128 (conformer rel-pos attention as [q+u | q+v] . [k | p]); values are 64 wide

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

template <int DQK>
__global__ __launch_bounds__(256) void attn_mfma_kernel(AttnArgs a) {
  constexpr int KS = DQK / 32, KQROW = DQK + 8;  // k-steps of S = Q K^T; LDS row stride of the key tile
  __shared__ __attribute__((aligned(16))) bf16_t sK[KT][KQROW];     // [key][dim]
  __shared__ __attribute__((aligned(16))) bf16_t sVt[DH][KROW];     // [dim][key]
  __shared__ __attribute__((aligned(16))) bf16_t sP[4][16][KROW];   // per wave [query][key]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QT;
  const bf16_t* __restrict__ q = (const bf16_t*)a.q;
  const bf16_t* __restrict__ k = (const bf16_t*)a.k;
  const bf16_t* __restrict__ v = (const bf16_t*)a.v;
  bf16_t* __restrict__ o = (bf16_t*)a.o;
  const int kvs = a.kv_start ? a.kv_start[b] : 0;
  const int shift = a.Sk - a.Sq;
  const int fr = lane & 15, fg = lane >> 4;
  // Q fragments of this wave's 16 rows (A operand: row fr, dims 32*ks + 8*fg .. +8), pre-scaled later in fp32
  const int qrow = min(q0 + wave * 16 + fr, a.Sq - 1);
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = *reinterpret_cast<const bf16x8*>(q + ((size_t)b * a.Sq + qrow) * a.ldq + h * DQK + ks * 32 + fg * 8);
  f32x4v oacc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) oacc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    m[r] = -INFINITY;
    l[r] = 0.f;
  }
  int kend = a.Sk;
  if (a.causal) kend = min(a.Sk, q0 + QT + shift);
  // loader mapping: thread -> key row tid/4 (64 keys), 16 dims (tid&3)*16 .. +16 (two 16-byte loads)
  const int lk = tid >> 2, ld = (tid & 3) * 16, ldk4 = (tid & 3) * (DQK / 4);
  for (int j0 = 0; j0 < kend; j0 += KT) {
    const int jr = min(j0 + lk, a.Sk - 1);
    const bf16_t* kp = k + ((size_t)b * a.Sk + jr) * a.ldk + h * DQK + ldk4;
    const bf16_t* vp = v + ((size_t)b * a.Sk + jr) * a.ldv + h * DH + ld;
    u32x4 kq[DQK / 32];
#pragma unroll
    for (int i = 0; i < DQK / 32; ++i) kq[i] = *reinterpret_cast<const u32x4*>(kp + 8 * i);
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(vp), v1 = *reinterpret_cast<const u32x4*>(vp + 8);
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < DQK / 32; ++i) *reinterpret_cast<u32x4*>(&sK[lk][ldk4 + 8 * i]) = kq[i];
    {
      const unsigned short* e0 = reinterpret_cast<const unsigned short*>(&v0);
      const unsigned short* e1 = reinterpret_cast<const unsigned short*>(&v1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        reinterpret_cast<unsigned short*>(&sVt[ld + i][0])[lk] = e0[i];
        reinterpret_cast<unsigned short*>(&sVt[ld + 8 + i][0])[lk] = e1[i];
      }
    }
    __syncthreads();
    // ---- S = Q K^T : 4 key tiles of 16 x 2 k-steps ----
    f32x4v s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&sK[nt * 16 + fr][ks * 32 + fg * 8]);
        s[nt] = half_mfma16(qf[ks], kf, s[nt]);
      }
    }
    // ---- mask + online softmax (rows fg*4 + r, columns fr + 16 nt) ----
    float p[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qi = q0 + wave * 16 + fg * 4 + r;
      float mx = -INFINITY;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int j = j0 + nt * 16 + fr;
        bool ok = j < a.Sk && j >= kvs;
        if (a.causal) ok = ok && j <= qi + shift;
        const float sc = ok ? s[nt][r] * a.scale : -INFINITY;
        p[nt][r] = sc;
        mx = fmaxf(mx, sc);
      }
      mx = row16_max(mx);
      const float mn = fmaxf(m[r], mx);
      const float corr = mn > -INFINITY ? __expf(m[r] - mn) : 1.f;
      float sum = 0.f;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float e = p[nt][r] > -INFINITY ? __expf(p[nt][r] - mn) : 0.f;
        p[nt][r] = e;
        sum += e;
      }
      sum = row16_sum(sum);
      l[r] = l[r] * corr + sum;
      m[r] = mn;
#pragma unroll
      for (int t = 0; t < 4; ++t) oacc[t][r] *= corr;
    }
    // ---- P -> wave-private LDS tile [query][key] (bf16), then as A operand ----
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sP[wave][fg * 4 + r][nt * 16 + fr] = (bf16_t)p[nt][r];
    __builtin_amdgcn_wave_barrier();
    bf16x8 pf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) pf[ks] = *reinterpret_cast<const bf16x8*>(&sP[wave][fr][ks * 32 + fg * 8]);
    // ---- O += P V : 4 dim tiles of 16 x 2 k-steps (keys) ----
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(&sVt[t * 16 + fr][ks * 32 + fg * 8]);
        oacc[t] = half_mfma16(pf[ks], vf, oacc[t]);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = q0 + wave * 16 + fg * 4 + r;
    if (qi >= a.Sq) continue;
    const float inv = l[r] > 0.f ? 1.f / l[r] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) o[((size_t)b * a.Sq + qi) * a.ldo + h * DH + t * 16 + fr] = (bf16_t)(oacc[t][r] * inv);
  }
}

}  // namespace

bool attention_mfma_supported(const AttnArgs& a, int dt) {
  return dt == BF16 && (a.dqk == 64 || a.dqk == 128) && a.dv == 64 && a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 &&
         !(((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v) & 15) && a.Sq >= 16;
}

int attention_mfma(const AttnArgs& a, int dt, hipStream_t s) {
  ITTS_REQUIRE(a.q && a.k && a.v && a.o, "attention_mfma: null pointer");
  ITTS_REQUIRE(attention_mfma_supported(a, dt), "attention_mfma: unsupported shape/dtype");
  dim3 grid((a.Sq + QT - 1) / QT, a.H, a.B);
  if (a.dqk == 128)
    hipLaunchKernelGGL(attn_mfma_kernel<128>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(attn_mfma_kernel<64>, grid, dim3(256), 0, s, a);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int attention(const AttnArgs& a, int dt, hipStream_t s) {
  static const bool no_mfma = getenv("ITTS_ATTN_SIMPLE") != nullptr;
  if (!no_mfma && attention_mfma_supported(a, dt)) return attention_mfma(a, dt, s);
  return attention_simple(a, dt, s);
}

}  // namespace itts
