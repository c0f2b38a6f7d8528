// Decode-step kernels, second generation (the per-token loop is launch/latency bound at B <= 8:
// SURVEY.md 8d, MI355X_MICROARCH "launches-baseline"), so each kernel here is built to
//   * touch every weight byte exactly once with 16-byte lane loads that are ALL issued before the first use
//     (deep memory-level parallelism, no LDS round trip for streamed weights - cdna_hip_programming "GEMV" row),
//   * keep the tiny activation vectors in LDS (fused LayerNorm / ln_f+final_norm / split-KV combine prologues),
//   * spread over >= 256 workgroups so every CU pulls on HBM.
#include <cstdlib>

#include "itts_decode.h"
#include "itts_sampler_dev.h"
#include "decode_pinned.h"

namespace itts {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// weights are read exactly once per step: non-temporal 16-byte loads (MI355X_MICROARCH "nt-weights")
template <typename TW> struct V8;
template <> struct V8<bf16_t> {
  u32x4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
  __device__ __forceinline__ float get(int i) const {
    const uint32_t w = raw[i >> 1];
    return (i & 1) ? half_hi(w) : half_lo(w);
  }
};
template <> struct V8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
  }
  __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};

// ---------------------------------------------------------------------------------------------
// gemv2: Y[b, n] (+)= act( prologue(X)[b, :] . W[n, :] + bias[n] )
//   prologue: 0 plain, 1 LayerNorm, 2 LayerNorm o LayerNorm (ln_f then final_norm)
// block = 4 waves; wave w owns RPW rows; each lane holds RPW x NCH 16-byte weight fragments in registers.
// Single-latency structure: the weight fragments, the activation rows, gamma and beta are all requested
// before anything is consumed; LayerNorm statistics are one shifted-moment pass (pivot = x[0]) reduced
// with ONE barrier; the normalised rows go to LDS for the dot-product phase.
// ---------------------------------------------------------------------------------------------
template <int NB>
struct RowStats {
  float mean[NB], rstd[NB];
};

// sum[b], sq[b] (shifted moments of this thread's elements) -> per-row mean / rstd, one barrier
template <int NB>
__device__ __forceinline__ RowStats<NB> reduce_stats(const float (&sum)[NB], const float (&sq)[NB],
                                                     const float (&pivot)[NB], int K, float eps,
                                                     float (*red)[2 * NB], int lane, int wave) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float s = wave_sum(sum[b]), q = wave_sum(sq[b]);
    if (lane == 0) {
      red[wave][2 * b] = s;
      red[wave][2 * b + 1] = q;
    }
  }
  __syncthreads();
  RowStats<NB> st;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float s = red[0][2 * b] + red[1][2 * b] + red[2][2 * b] + red[3][2 * b];
    const float q = red[0][2 * b + 1] + red[1][2 * b + 1] + red[2][2 * b + 1] + red[3][2 * b + 1];
    const float md = s / K;
    st.mean[b] = pivot[b] + md;
    st.rstd[b] = rsqrtf(fmaxf(q / K - md * md, 0.f) + eps);
  }
  return st;
}

template <typename TW, int NB, int RPW, int NCH>
__global__ __launch_bounds__(256) void gemv2_kernel(GemvArgs g) {
  constexpr int XCH = (NB * NCH + 1) / 2;  // float4 chunks of X per thread (NB*K <= NB*512*NCH floats)
  extern __shared__ __attribute__((aligned(16))) float sx[];  // [NB][K]
  __shared__ float red[2][4][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = g.K, B = g.B, BK = B * K;
  const TW* __restrict__ W = (const TW*)g.W;
  const int n0 = (blockIdx.x * 4 + wave) * RPW;
  // 1. request everything: weight fragments, activation rows, LayerNorm parameters, pivots
  V8<TW> w[RPW][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < K) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int n = min(n0 + r, g.N - 1);
        w[r][c].load(W + (size_t)n * K + k);
      }
    }
  }
  float4 x[XCH], gm[XCH], bt[XCH], gm2[XCH], bt2[XCH];
  const bool ln = g.prologue >= 1, ln2 = g.prologue == 2;
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int i = tid * 4 + j * 1024;
    if (i < BK) {
      x[j] = *reinterpret_cast<const float4*>(g.X + i);
      const int col = i % K;
      gm[j] = gm2[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      bt[j] = bt2[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ln && g.ln_gamma) {  // null = plain normalisation (affine folded into W by the packer)
        gm[j] = *reinterpret_cast<const float4*>(g.ln_gamma + col);
        bt[j] = *reinterpret_cast<const float4*>(g.ln_beta + col);
      }
      if (ln2 && g.ln2_gamma) {
        gm2[j] = *reinterpret_cast<const float4*>(g.ln2_gamma + col);
        bt2[j] = *reinterpret_cast<const float4*>(g.ln2_beta + col);
      }
    }
  }
  float pivot[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) pivot[b] = (ln && b < B) ? g.X[(size_t)b * K] : 0.f;
  // 2. LayerNorm(s) in registers, result to LDS
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0 ? !ln : !ln2) break;
    float sum[NB], sq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) sum[b] = sq[b] = 0.f;
    if (pass == 1) {
#pragma unroll
      for (int b = 0; b < NB; ++b) pivot[b] = 0.f;  // LayerNorm output: mean ~ beta, well conditioned
    }
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const int i = tid * 4 + j * 1024;
      if (i < BK) {
        const int b = i / K;
        const float v[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
        for (int bb = 0; bb < NB; ++bb)
          if (bb == b) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = v[e] - pivot[bb];
              sum[bb] += d;
              sq[bb] = fmaf(d, d, sq[bb]);
            }
          }
      }
    }
    const RowStats<NB> st = reduce_stats<NB>(sum, sq, pivot, K, g.ln_eps, red[pass], lane, wave);
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const int i = tid * 4 + j * 1024;
      if (i < BK) {
        const int b = i / K;
        float m = 0.f, r = 1.f;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb)
          if (bb == b) {
            m = st.mean[bb];
            r = st.rstd[bb];
          }
        const float4 G = pass ? gm2[j] : gm[j], Bt = pass ? bt2[j] : bt[j];
        x[j].x = (x[j].x - m) * r * G.x + Bt.x;
        x[j].y = (x[j].y - m) * r * G.y + Bt.y;
        x[j].z = (x[j].z - m) * r * G.z + Bt.z;
        x[j].w = (x[j].w - m) * r * G.w + Bt.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int i = tid * 4 + j * 1024;
    if (i < BK) *reinterpret_cast<float4*>(sx + i) = x[j];
  }
  __syncthreads();
  if (n0 >= g.N) return;
  // 3. dot products
  float acc[RPW][NB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < K) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= B) break;
        const float4 x0 = *reinterpret_cast<const float4*>(sx + b * K + k);
        const float4 x1 = *reinterpret_cast<const float4*>(sx + b * K + k + 4);
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[r][b] = fmaf(xv[i], w[r][c].get(i), acc[r][b]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int n = n0 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= B) break;
        float v = acc[r][b] + (g.bias ? g.bias[n] : 0.f);
        v = act_apply(g.act, v);
        float* y = g.Y + (size_t)b * g.ldy + n;
        *y = g.accumulate ? (*y + v) : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gemv_bf16: the throughput-path GEMV (bf16 weights).  Same contract as gemv2, built from the measured
// anatomy of these 3-10 us kernels (tools/ubench_gemv.hip; DESIGN.md "decode GEMV"):
//   * branch-free: every load has a clamped address and is unconditional, so hipcc's in-order vmcnt
//     bookkeeping is exact - the activations (requested FIRST) are consumed while the weight fragments
//     (requested second) are still streaming;
//   * LayerNorm statistics with DPP row reductions (4 DPP + 2 bpermute instead of 6 bpermute);
//   * activations are kept in LDS as bf16 pairs and multiplied with v_dot2c_f32_bf16: 4 VALU ops per
//     16-byte weight fragment instead of 8 cvt + 8 fma (the kernels are short enough to be issue-bound);
//   * the output can be written as bf16 (gelu(fc) feeding proj2) to halve the next kernel's LDS fill.
// ---------------------------------------------------------------------------------------------
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror: every lane now holds the sum of its 16-lane row
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// full-wave sum, result uniform in every lane: DPP inside the 16-lane rows, then one v_readlane per row - no ds_bpermute
// (an LDS-crossbar round trip with an lgkmcnt wait) on the dependent chain
__device__ __forceinline__ float wave_sum_rl(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  const int iv = __float_as_int(v);
  const float a = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), b = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), d = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return (a + b) + (c + d);
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  bf16x2_t v = {(bf16_t)a, (bf16_t)b};
  return __builtin_bit_cast(uint32_t, v);
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return half_dot2(a, b, c);
}

// PRO: 0 plain, 1 LayerNorm without affine (gamma/beta are folded into W by the packer), 2 LayerNorm(affine) then
// LayerNorm without affine (ln_f, then final_norm folded into mel_head).  XBF: X is bf16 [B, K].  YBF: Y is bf16.
// (A wave-specialised variant - dedicated activation waves - was measured slower.)
// Every batch row uses the SAME thread <-> element mapping, so a row's result does not depend on its position in
// the batch (the padding/batch invariance the reference's tests/padding_test.py checks).
// W8: weights stored as OCP fp8 e4m3 bytes with one power-of-two scale per output row (BASELINE config 5): half the
// weight stream; two v_cvt_scalef32_pk_bf16_fp8 per 4 weights feed the same v_dot2c, the row scale multiplies the sum.
#ifdef ITTS_GEMV_STAMPS
#define GEMV_STAMP(i)                                                                   \
  {                                                                                     \
    unsigned long long t_;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
    if (threadIdx.x == 0 && g.stamp) g.stamp[(size_t)blockIdx.x * 8 + (i)] = t_;        \
  }
#else
#define GEMV_STAMP(i)
#endif

// WAVES per workgroup: 4, or 5 so that the per-layer projections (3840 / 1280 / 5120 rows) split into exactly 256
// workgroups - one per CU, every CU streaming the same share of the weights and loading x once.
template <int NB, int RPW, int NCH, int PRO, bool XBF, bool YBF, bool W8 = false, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64) void gemv_bf16_kernel(GemvArgs g) {
  constexpr int NTHR = WAVES * 64;
  GEMV_STAMP(0)
  constexpr int EPC = XBF ? 8 : 4;                                  // elements per 16-byte chunk
  constexpr int KCH = (NCH * 512 + NTHR * EPC - 1) / (NTHR * EPC);  // chunks per row per thread
  extern __shared__ __attribute__((aligned(16))) uint32_t sxb[];    // [NB][K/2] bf16 pairs
  __shared__ float red[2][WAVES][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = g.K;
  const float invK = 1.f / (float)K;  // off the critical path: the LayerNorm chain multiplies instead of dividing
  const int n0 = (blockIdx.x * WAVES + wave) * RPW;
  // ---- 1. activations (+ LayerNorm parameters) first, weights second; all unconditional ----
  u32x4 xr[NB][KCH];  // XBF: 8 bf16; else 4 floats
  f32x4 pm[PRO == 3 ? NB : 1][KCH], pl[PRO == 3 ? NB : 1][KCH], po[PRO == 3 ? NB : 1][KCH][ATTN_NSPLIT][2];
  static_assert(PRO != 3 || (XBF && ATTN_NSPLIT == 4), "prologue 3 feeds bf16 pairs and reads 4 partials as one float4");
  f32x4 gm[PRO == 2 ? KCH : 1], bt[PRO == 2 ? KCH : 1];
  bool xok[KCH];
#pragma unroll
  for (int j = 0; j < KCH; ++j) {
    const int i = (tid + j * NTHR) * EPC;
    xok[j] = i < K;
    const int ic = xok[j] ? i : K - EPC;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const size_t ro = (size_t)min(b, g.B - 1) * K + ic;
      if constexpr (PRO == 3) {
        // x is the attention output, still in ATTN_NSPLIT partials: this thread's 8 dims of head ic / 64
        const size_t bh = (size_t)min(b, g.B - 1) * (K >> 6) + (ic >> 6);
        const float* ml = g.attn_ml + bh * 2 * ATTN_NSPLIT;
        pm[b][j] = *reinterpret_cast<const f32x4*>(ml);
        pl[b][j] = *reinterpret_cast<const f32x4*>(ml + ATTN_NSPLIT);
#pragma unroll
        for (int sp = 0; sp < ATTN_NSPLIT; ++sp) {
          const float* po_ = g.attn_o + (bh * ATTN_NSPLIT + sp) * 64 + (ic & 63);
          po[b][j][sp][0] = *reinterpret_cast<const f32x4*>(po_);
          po[b][j][sp][1] = *reinterpret_cast<const f32x4*>(po_ + 4);
        }
      } else if (XBF) {
        xr[b][j] = *reinterpret_cast<const u32x4*>((const bf16_t*)g.X + ro);
      } else {
        xr[b][j] = *reinterpret_cast<const u32x4*>(g.X + ro);
      }
    }
    if (PRO == 2) {
      gm[j] = *reinterpret_cast<const f32x4*>(g.ln_gamma + ic);
      bt[j] = *reinterpret_cast<const f32x4*>(g.ln_beta + ic);
    }
  }
  float pivot[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) pivot[b] = (PRO == 1 || PRO == 2) ? g.X[(size_t)min(b, g.B - 1) * K] : 0.f;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W;
  const uint8_t* __restrict__ Wq = (const uint8_t*)g.W8;
  u32x4 w[W8 ? 1 : RPW][W8 ? 1 : NCH];
  u32x2 w8[W8 ? RPW : 1][W8 ? NCH : 1];  // 8 fp8 weights per lane and chunk
  const int klast = (NCH - 1) * 512 + lane * 8;
  const bool kok = klast < K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if constexpr (W8)
        w8[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(Wq + (size_t)min(n0 + r, g.N - 1) * K + k));
      else
        w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + r, g.N - 1) * K + k));
    }
  }
  // epilogue operands of the output this lane will finish, (row lane / NB, batch lane % NB): bias, fp8 row scale and the
  // residual-stream value it accumulates into are requested now (youngest loads, unconditional), so the epilogue has no
  // dependent memory latency of its own
  const int er = min(lane / NB, RPW - 1), eb = lane % NB;
  const int en = min(n0 + er, g.N - 1);
  const float* bp = g.bias ? g.bias : reinterpret_cast<const float*>(W8 ? g.W8 : g.W);  // any readable address when there is no bias
  const float bpre = bp[en];
  const float spre = W8 ? g.wscale[en] : 1.f;
  const float ypre = YBF ? 0.f : g.Y[(size_t)min(eb, g.B - 1) * g.ldy + en];
  // every request of this kernel is now in flight.  The fence keeps it that way: without it the machine scheduler sinks
  // most of the weight loads below the first wait on X (fewer live registers), i.e. two thirds of the weight stream
  // would be requested one memory latency late
  __builtin_amdgcn_sched_barrier(0);
  GEMV_STAMP(1)
#ifdef ITTS_GEMV_STAMPS
  { unsigned pr_ = xr[0][0][0]; asm volatile("" ::"v"(pr_)); }
  GEMV_STAMP(2)
#endif
  // ---- 2. LayerNorm(s) in registers (one barrier each), bf16 pairs to LDS ----
  if (!XBF) {
    float xv[NB][KCH][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[b][j][e] = __uint_as_float(xr[b][j][e]);
#pragma unroll
    for (int pass = 0; pass < PRO; ++pass) {
      float s[NB], q[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        s[b] = q[b] = 0.f;
        const float pv = pass == 0 ? pivot[b] : 0.f;
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = xok[j] ? xv[b][j][e] - pv : 0.f;
            s[b] += d;
            q[b] = fmaf(d, d, q[b]);
          }
        s[b] = wave_sum_rl(s[b]);
        q[b] = wave_sum_rl(q[b]);
      }
      if (lane == 0)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          red[pass][wave][2 * b] = s[b];
          red[pass][wave][2 * b + 1] = q[b];
        }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
          S += red[pass][ww][2 * b];
          Q += red[pass][ww][2 * b + 1];
        }
        // contraction pinned (decode_pinned.h): the persistent engine repeats these operations bit for bit
        const float md = __fmul_rn(S, invK);
        const float mean = __fadd_rn(pass == 0 ? pivot[b] : 0.f, md);
        const float rstd = __builtin_amdgcn_rsqf(__fadd_rn(fmaxf(ln_var_rn(Q, invK, md), 0.f), g.ln_eps));
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = (xv[b][j][e] - mean) * rstd;
            if (PRO == 2 && pass == 0) v = ln_affine_rn(v, gm[j][e], bt[j][e]);  // pinned: the persistent engine's head repeats it
            xv[b][j][e] = v;
          }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
        if (xok[j]) {
          const int i = (tid + j * NTHR) * 4;
          uint2 p;
          p.x = pack_bf16(xv[b][j][0], xv[b][j][1]);
          p.y = pack_bf16(xv[b][j][2], xv[b][j][3]);
          *reinterpret_cast<uint2*>(sxb + (b * K + i) / 2) = p;
        }
  } else {
    if constexpr (PRO == 3) {
      // merge the split-attention partials: weights exp(max_p - max), one division per head
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
          const float M = fmaxf(fmaxf(pm[b][j][0], pm[b][j][1]), fmaxf(pm[b][j][2], pm[b][j][3]));
          float wgt[ATTN_NSPLIT], L = 0.f;
#pragma unroll
          for (int sp = 0; sp < ATTN_NSPLIT; ++sp) {
            wgt[sp] = pm[b][j][sp] > -INFINITY ? __expf(pm[b][j][sp] - M) : 0.f;
            L = fmaf(wgt[sp], pl[b][j][sp], L);
          }
          const float inv = 1.f / L;
          float xm[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int sp = 0; sp < ATTN_NSPLIT; ++sp) a = fmaf(wgt[sp], po[b][j][sp][e >> 2][e & 3], a);
            xm[e] = a * inv;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[b][j][e] = pack_bf16(xm[2 * e], xm[2 * e + 1]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
        if (xok[j]) *reinterpret_cast<u32x4*>(sxb + (b * K + (tid + j * NTHR) * 8) / 2) = xr[b][j];
  }
  __syncthreads();
  GEMV_STAMP(3)
  // ---- 3. dot products: 4 x v_dot2c per weight fragment and batch row ----
  float acc[RPW][NB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      u32x4 xq = *reinterpret_cast<const u32x4*>(sxb + (b * K + k) / 2);
      if (c == NCH - 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) xq[e] = kok ? xq[e] : 0u;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        if constexpr (W8) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const uint32_t q = w8[r][c][h2];  // 4 fp8: bytes 0,1 -> pair 2*h2, bytes 2,3 -> pair 2*h2 + 1
            acc[r][b] = dot2(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, false)), xq[2 * h2], acc[r][b]);
            acc[r][b] = dot2(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, true)), xq[2 * h2 + 1], acc[r][b]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][b] = dot2(w[r][c][e], xq[e], acc[r][b]);
        }
      }
    }
  }
  // wave reduction, then one lane per output: lane l < RPW * NB stores (row l / NB, batch l % NB)
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float t = wave_sum_rl(acc[r][b]);
      mine = lane == r * NB + b ? t : mine;
    }
  GEMV_STAMP(4)
  if (lane < RPW * NB && n0 + er < g.N && eb < g.B) {
    float v = mine * spre + (g.bias ? bpre : 0.f);
    v = g.act == ACT_GELU_NEW ? gelu_new_rn(v) : act_apply(g.act, v);
    const size_t o = (size_t)eb * g.ldy + en;
    if (YBF)
      ((bf16_t*)g.Y)[o] = (bf16_t)v;
    else
      g.Y[o] = g.accumulate ? ypre + v : v;
  }
  GEMV_STAMP(5)
}

// ---------------------------------------------------------------------------------------------
// gemv_wave_kernel: the same projection, but every WAVE is autonomous.  A wave needs, for its RPW weight rows, exactly
// the activations x[c*512 + lane*8 .. +8] that multiply its lane's weight fragments - so each lane loads those itself
// (L2 hits: every wave of the grid reads the same 5-20 KB), LayerNorm statistics are two DPP wave reductions per row on
// registers, and the normalised bf16 pairs feed v_dot2c directly.  No LDS, no barrier, no inter-wave dependency: the
// phase timeline of the block-cooperative kernel above (tools/ubench_gemv2.hip) showed its LayerNorm + 2 barriers on the
// critical path for 1.4 us while the weight stream had already landed, and a serial lane-0 epilogue of 0.5 us.
// The epilogue is spread over lanes: lane l < RPW*NB finishes output (row l / NB, batch l % NB) with ONE store.
// ---------------------------------------------------------------------------------------------
template <int NB, int RPW, int NCH, int PRO, bool XBF, bool YBF, bool W8>
__global__ __launch_bounds__(256) void gemv_wave_kernel(GemvArgs g) {
  GEMV_STAMP(0)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = g.K;
  const int n0 = (blockIdx.x * 4 + wave) * RPW;
  int kc[NCH];
  bool kok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 512 + lane * 8;
    kok[c] = k < K;
    kc[c] = kok[c] ? k : K - 8;  // clamped, in bounds; masked below
  }
  // ---- 1. every request of the wave: activations first (LayerNorm starts on them), then weights, then the epilogue operands
  u32x4 xraw[NB][NCH][XBF ? 1 : 2];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const size_t ro = (size_t)min(b, g.B - 1) * K + kc[c];
      if constexpr (XBF) {
        xraw[b][c][0] = *reinterpret_cast<const u32x4*>((const bf16_t*)g.X + ro);
      } else {
        xraw[b][c][0] = *reinterpret_cast<const u32x4*>(g.X + ro);
        xraw[b][c][1] = *reinterpret_cast<const u32x4*>(g.X + ro + 4);
      }
    }
  f32x4 gm[PRO == 2 ? NCH : 1][2], bt[PRO == 2 ? NCH : 1][2];
  if constexpr (PRO == 2) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        gm[c][hh] = *reinterpret_cast<const f32x4*>(g.ln_gamma + kc[c] + 4 * hh);
        bt[c][hh] = *reinterpret_cast<const f32x4*>(g.ln_beta + kc[c] + 4 * hh);
      }
  }
  const bf16_t* __restrict__ W = (const bf16_t*)g.W;
  const uint8_t* __restrict__ Wq = (const uint8_t*)g.W8;
  u32x4 w[W8 ? 1 : RPW][W8 ? 1 : NCH];
  u32x2 w8[W8 ? RPW : 1][W8 ? NCH : 1];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const size_t wo = (size_t)min(n0 + r, g.N - 1) * K + kc[c];
      if constexpr (W8)
        w8[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(Wq + wo));
      else
        w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + wo));
    }
  // epilogue operands of the output this lane will finish: (row lane / NB, batch lane % NB)
  const int er = min(lane / NB, RPW - 1), eb = lane % NB;
  const int en = min(n0 + er, g.N - 1);
  const float* bp = g.bias ? g.bias : reinterpret_cast<const float*>(W8 ? g.W8 : g.W);  // any readable address without a bias
  const float bpre = bp[en];
  const float spre = W8 ? g.wscale[en] : 1.f;
  const float ypre = YBF ? 0.f : g.Y[(size_t)min(eb, g.B - 1) * g.ldy + en];
  __builtin_amdgcn_sched_barrier(0);  // keep all of the above in flight before the first wait (see gemv_bf16_kernel)
  GEMV_STAMP(1)
  // ---- 2. LayerNorm(s) in registers: two wave reductions per row and pass, then bf16 pairs ----
  uint32_t xq[NB][NCH][4];
  if constexpr (!XBF) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float xv[NCH][8];
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[c][e] = __uint_as_float(xraw[b][c][e >> 2][e & 3]);
#pragma unroll
      for (int pass = 0; pass < PRO; ++pass) {
        // one pass: moments about a pivot (the row's first element for the raw residual stream, 0 for a LayerNorm output),
        // both sums go through the wave reduction together
        const float pv = pass == 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv[0][0]), 0)) : 0.f;
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = kok[c] ? xv[c][e] - pv : 0.f;
            sm += d;
            sq = fmaf(d, d, sq);
          }
        sm = wave_sum_rl(sm);
        sq = wave_sum_rl(sq);
        const float md = sm / K;
        const float mean = pv + md;
        const float rstd = rsqrtf(fmaxf(sq / K - md * md, 0.f) + g.ln_eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = (xv[c][e] - mean) * rstd;
            if (PRO == 2 && pass == 0) v = v * gm[c][e >> 2][e & 3] + bt[c][e >> 2][e & 3];
            xv[c][e] = v;
          }
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) xq[b][c][e] = kok[c] ? pack_bf16(xv[c][2 * e], xv[c][2 * e + 1]) : 0u;
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) xq[b][c][e] = kok[c] ? xraw[b][c][0][e] : 0u;
  }
  GEMV_STAMP(3)
  // ---- 3. dot products ----
  float acc[RPW][NB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if constexpr (W8) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const uint32_t q = w8[r][c][h2];
          const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, false));
          const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, true));
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[r][b] = dot2(lo, xq[b][c][2 * h2], acc[r][b]);
            acc[r][b] = dot2(hi, xq[b][c][2 * h2 + 1], acc[r][b]);
          }
        }
      } else {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][b] = dot2(w[r][c][e], xq[b][c][e], acc[r][b]);
      }
    }
  // ---- 4. wave reduction, then one lane per output ----
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float t = wave_sum_rl(acc[r][b]);
      mine = lane == r * NB + b ? t : mine;
    }
  GEMV_STAMP(4)
  if (lane < RPW * NB && n0 + er < g.N && eb < g.B) {
    float v = mine * spre + (g.bias ? bpre : 0.f);
    v = g.act == ACT_GELU_NEW ? gelu_new_rn(v) : act_apply(g.act, v);
    const size_t o = (size_t)eb * g.ldy + en;
    if (YBF)
      ((bf16_t*)g.Y)[o] = (bf16_t)v;
    else
      g.Y[o] = g.accumulate ? ypre + v : v;
  }
  GEMV_STAMP(5)
}

// ---------------------------------------------------------------------------------------------
// decode_attn2: single-query attention over the KV cache with this step's K/V append fused in.
// One 1024-thread workgroup per (head, row).  LPK lanes share one key row with 16-byte loads (a wave reads
// 1 KiB of contiguous K and 1 KiB of V per step); every key slot runs an online softmax (running max, sum,
// partial context) so K and V are read in ONE pass with both loads of a step in flight together; slots are
// merged with max-rescaling through shuffles and LDS.  Output type TO: fp32 or bf16 (feeds the proj GEMV).
// ---------------------------------------------------------------------------------------------
template <typename TC> struct CacheVec;
template <> struct CacheVec<bf16_t> {
  static constexpr int VEC = 8, LPK = 8;
  uint4 raw;
#ifdef ITTS_KV_PLAIN_LOADS
  __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
#else
  // the cache is read once per step and never again before it has left every cache: nontemporal (streaming) loads
  __device__ __forceinline__ void load(const bf16_t* p) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p));
    raw = make_uint4(t[0], t[1], t[2], t[3]);
  }
#endif
  __device__ __forceinline__ float get(int i) const {
    const uint32_t w = (&raw.x)[i >> 1];
    return (i & 1) ? half_hi(w) : half_lo(w);
  }
};
template <> struct CacheVec<float> {
  static constexpr int VEC = 4, LPK = 16;
  float4 raw;
  __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ float get(int i) const { return (&raw.x)[i]; }
};

// NIT = key pairs per slot held in registers (NIT * 2 * SLOTS keys: 768 with bf16 at NIT = 3).  The first pair is
// requested before any device scalar is read, the rest as soon as the sequence length is known, all before the first
// use: the whole cache read costs two overlapped memory latencies instead of one per iteration.
#ifdef ITTS_GEMV_STAMPS
__device__ unsigned long long* g_attn_stamp = nullptr;
#define ATTN_STAMP(i)                                                                                  \
  {                                                                                                    \
    unsigned long long t_;                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                        \
    if (threadIdx.x == 0 && g_attn_stamp) g_attn_stamp[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = t_; \
  }
#else
#define ATTN_STAMP(i)
#endif

// NT = threads per (row, head): 1024 for the latency-bound small batches (one workgroup per CU), 256 once there are
// enough (row, head) pairs to fill the CUs several times over - 5 workgroups per CU overlap their load / softmax /
// merge phases, where the 86-VGPR 1024-thread form runs its 5 rounds per CU back to back.
// NSPLIT > 1: the keys of one (row, head) are dealt round-robin, SLOTS rows at a time, to NSPLIT workgroups
// (blockIdx.z); each writes an un-normalised partial (max, sum, weighted V) and the consumer - the attention-projection
// GEMV, prologue 3 - merges them while it loads its activations.  At 2 rows x 20 heads this turns 40 workgroups of 16
// waves (4 waves per SIMD: the softmax phase is issue-bound and the last wave trails the first by 1.4 us on the phase
// timeline) into 160 workgroups of 4 waves, one per SIMD, on 160 CUs.
// ANC (beam-sample): the cache is NOT re-ordered when the beams are (HF `_reorder_cache` copies every layer's K/V by
// beam_idx each step, model.py:194-207).  Instead every beam row b carries an ancestry row anc[b][j] = which of the nb
// physical rows of its batch item holds position j of ITS history; the beam sampler rewrites those few KB per step
// (ping-pong by the parity of the step count) and this kernel gathers K/V rows through it.  The row appended by this
// step always goes to the beam's own physical row.
template <typename TC, typename TO, int NIT, int NT, int NSPLIT = 1, bool ANC = false>
__global__ __launch_bounds__(NT) void decode_attn2_kernel(TO* __restrict__ ctx, const float* __restrict__ qkv,
                                                          TC* __restrict__ kc, TC* __restrict__ vc,
                                                          const int* __restrict__ len, const int* __restrict__ kv_start,
                                                          const int* __restrict__ prefix, int H, int Smax, float scale,
                                                          int ctx_bt, float* __restrict__ part_o = nullptr,
                                                          float* __restrict__ part_ml = nullptr,
                                                          const uint8_t* __restrict__ anc = nullptr, int nb = 1) {
  constexpr int DH = 64, VEC = CacheVec<TC>::VEC, LPK = CacheVec<TC>::LPK, NW = NT / 64, SLOTS = NT / LPK;
  constexpr int SD = NT >= 1024 ? 2 : 4;  // rows per slot in flight beyond the register window
  ATTN_STAMP(0)
  __shared__ float sm[NW], sl[NW];
  __shared__ float so[NW][DH];
  const int h = blockIdx.x, b = blockIdx.y, sp = NSPLIT > 1 ? blockIdx.z : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * DH;
  TC* kb = kc + ((size_t)b * H + h) * Smax * DH;
  TC* vb = vc + ((size_t)b * H + h) * Smax * DH;
  const int slot = tid / LPK, sub = tid % LPK;
  // cache row of key j: own row, or (ANC) the physical row the beam's ancestry names for that position
  const uint8_t* arow = nullptr;
  int rowbase = 0;
  if constexpr (ANC) {
    arow = anc + ((size_t)(len[b] & 1) * gridDim.y + b) * Smax;
    rowbase = (b / nb) * nb;
  }
  auto krow = [&](int j) -> const TC* {
    if constexpr (ANC) return kc + ((size_t)(rowbase + min((int)arow[j], nb - 1)) * H + h) * Smax * DH + (size_t)j * DH;
    return kb + (size_t)j * DH;
  };
  auto vrow = [&](int j) -> const TC* {
    if constexpr (ANC) return vc + ((size_t)(rowbase + min((int)arow[j], nb - 1)) * H + h) * Smax * DH + (size_t)j * DH;
    return vb + (size_t)j * DH;
  };
  // (a) the first pair of key/value rows of this slot, requested before ANYTHING else: their addresses depend on no
  //     device scalar (rows are clamped to the cache capacity; rows >= S are masked out below)
  CacheVec<TC> kr[2 * NIT], vr[2 * NIT];
  // UNC pairs are requested blind (rows past S cost their bytes but nothing waits for the length): 2 pairs cover every
  // prefix; with 1024 threads 4 pairs = 512 rows cover the sequence for most of a generation, so the device-scalar
  // -> load chain (a second full memory latency) only remains for the late steps
  constexpr int UNC = 2;  // 4 blind pairs (512 rows) measured slower: 0.602 vs 0.593 ms per step - the extra bytes cost more than the chain
#pragma unroll
  for (int u = 0; u < UNC; ++u) {
    const int j = min((u * NSPLIT + sp) * SLOTS + slot, Smax - 1);
    kr[u].load(krow(j) + sub * VEC);
    vr[u].load(vrow(j) + sub * VEC);
  }
  // (a') this thread's slice of the step's q / k / v (addresses depend on no device scalar either): straight to
  //      registers - no LDS round trip, no barrier between the query and the cache reads
  const float* qv = qkv + (size_t)b * 3 * D + h * DH + sub * VEC;
  float4 qraw[VEC / 4], kraw[VEC / 4], vraw[VEC / 4];
#pragma unroll
  for (int i = 0; i < VEC / 4; ++i) {
    qraw[i] = *reinterpret_cast<const float4*>(qv + 4 * i);
    kraw[i] = *reinterpret_cast<const float4*>(qv + D + 4 * i);
    vraw[i] = *reinterpret_cast<const float4*>(qv + 2 * D + 4 * i);
  }
  // (b) per-row scalars and the K/V append of this step
  const int pos = prefix[0] + len[b];
  const int S = pos + 1;
  const int ks = kv_start[b];
  // (c) now that S is known: request every remaining row of the sequence at once (one more memory latency in total).
  //     This comes BEFORE anything that consumes the q/k/v slice - vmcnt is in-order, and the K/V append below would
  //     otherwise make the wave sit out the first loads' latency before these are even issued
#pragma unroll
  for (int u = UNC; u < 2 * NIT; ++u)
    if ((u * NSPLIT + sp) * SLOTS < S) {  // block-uniform
      const int j = min((u * NSPLIT + sp) * SLOTS + slot, Smax - 1);
      kr[u].load(krow(j) + sub * VEC);
      vr[u].load(vrow(j) + sub * VEC);
    }
  ATTN_STAMP(1)
  // (d) the step's own q / k / v: scale, round as the cache does, append
  float qr[VEC], kown[VEC], vown[VEC];  // the appended row with the cache's rounding, never read back from HBM
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    qr[i] = (&qraw[i >> 2].x)[i & 3] * scale;
    kown[i] = (float)(TC)(&kraw[i >> 2].x)[i & 3];
    vown[i] = (float)(TC)(&vraw[i >> 2].x)[i & 3];
  }
  if (tid < LPK && sp == 0) {  // slot 0 (of split 0): its LPK lanes cover the 64 dims
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      stf(kb + (size_t)pos * DH + sub * VEC + i, kown[i]);
      stf(vb + (size_t)pos * DH + sub * VEC + i, vown[i]);
    }
  }
  ATTN_STAMP(2)
  float m = -INFINITY, l = 0.f, acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  // score of one cached key row for this slot (the LPK lanes of the key hold VEC dims each; DPP sums them: quad swaps,
  // half-row mirror, row mirror - no LDS crossbar trips)
  auto score = [&](const CacheVec<TC>& kk) {
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) sc = fmaf(qr[i], kk.get(i), sc);
    sc = dpp_add<0xB1>(sc);
    sc = dpp_add<0x4E>(sc);
    sc = dpp_add<0x141>(sc);
    if (LPK == 16) sc = dpp_add<0x140>(sc);
    return sc;
  };
  // (e) the register window in two phases, as torch.softmax does it: all scores, their maximum, then one exp per key and
  //     the weighted sum - half the VALU work of a per-key online update (no rescale of the accumulator per key), and the
  //     16 waves of a workgroup share 4 SIMDs, so this phase is issue-bound.  The row appended by this step (j == pos)
  //     is masked out of the window and enters as one extra key of slot 0, from registers.  Rows past S multiply by
  //     p = 0: the cache is zero-filled at allocation, so whatever they hold is finite.
  {
    float sc[2 * NIT + 1];
#pragma unroll
    for (int u = 0; u < 2 * NIT; ++u) {
      const int j = (u * NSPLIT + sp) * SLOTS + slot;
      const bool live = u < UNC || (u * NSPLIT + sp) * SLOTS < S;  // block-uniform: was this pair requested
      const float t = live ? score(kr[u]) : 0.f;
      sc[u] = (live && j < S && j >= ks && j != pos) ? t : -INFINITY;
    }
    {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) t = fmaf(qr[i], kown[i], t);
      t = dpp_add<0xB1>(t);
      t = dpp_add<0x4E>(t);
      t = dpp_add<0x141>(t);
      if (LPK == 16) t = dpp_add<0x140>(t);
      sc[2 * NIT] = (slot == 0 && sp == 0) ? t : -INFINITY;
    }
    float mw = sc[0];
#pragma unroll
    for (int u = 1; u <= 2 * NIT; ++u) mw = fmaxf(mw, sc[u]);
    if (mw > -INFINITY) {
#pragma unroll
      for (int u = 0; u < 2 * NIT; ++u)
        if (u < UNC || (u * NSPLIT + sp) * SLOTS < S) {  // block-uniform: pairs that were never requested hold no data at all
          const float p = __expf(sc[u] - mw);  // exp(-inf) = 0 for masked rows
          l += p;
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vr[u].get(i), acc[i]);
        }
      const float p = __expf(sc[2 * NIT] - mw);
      l += p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vown[i], acc[i]);
      m = mw;
    }
  }
  // online update for rows beyond the window (never the appended row when it lies inside the window)
  auto consume = [&](const CacheVec<TC>& kk, const CacheVec<TC>& vv, int j) {
    const bool ok = j < S && j >= ks && j != pos;
    float sc = score(kk);
    sc = ok ? sc : -INFINITY;  // also discards whatever an out-of-range row produced
    const float mn = fmaxf(m, sc);
    const float corr = mn > -INFINITY ? __expf(m - mn) : 1.f;
    const float p = ok ? __expf(sc - mn) : 0.f;
    l = fmaf(l, corr, p);  // contraction pinned: same operation in every build of this loop (decode_pinned.h)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, ok ? vv.get(i) : 0.f, acc[i] * corr);
    m = mn;
  };
  // sequences longer than the register-resident window: stream the rest two rows at a time
  for (int cb = 2 * NIT; (cb * NSPLIT + sp) * SLOTS < S; cb += SD) {  // chunk cb of this split = rows (cb*NSPLIT+sp)*SLOTS ..
    CacheVec<TC> k2[SD], v2[SD];
#pragma unroll
    for (int u = 0; u < SD; ++u) {
      const int j = min(((cb + u) * NSPLIT + sp) * SLOTS + slot, Smax - 1);
      k2[u].load(krow(j) + sub * VEC);
      v2[u].load(vrow(j) + sub * VEC);
    }
#pragma unroll
    for (int u = 0; u < SD; ++u) consume(k2[u], v2[u], ((cb + u) * NSPLIT + sp) * SLOTS + slot);
  }
  ATTN_STAMP(3)
  // merge the 64/LPK key slots of this wave (lanes with equal `sub`)
  // merge across the wave without the LDS crossbar: lane ^ 8 is a DPP rotate inside the 16-lane row; lane ^ 16 and
  // lane ^ 32 are v_permlane16_swap / v_permlane32_swap (CDNA4), which hand every lane BOTH partners' values
  // (tools/probe_permlane.hip prints the lane maps) - a sum or max of the two results is the butterfly step
  auto bfly_max = [&](float x, int o) {
    if (o == 8) return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true)));
    const u32x2 r = o == 16 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)
                            : __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  auto bfly_sum = [&](float x, int o) {
    if (o == 8) return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
    const u32x2 r = o == 16 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)
                            : __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  };
  float M = m;
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) M = bfly_max(M, o);
  const float sc0 = M > -INFINITY ? __expf(m - M) : 0.f;
  l *= sc0;
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] *= sc0;
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    l = bfly_sum(l, o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = bfly_sum(acc[i], o);
  }
  if (lane < LPK)
#pragma unroll
    for (int i = 0; i < VEC; ++i) so[wave][lane * VEC + i] = acc[i];
  if (lane == 0) {
    sm[wave] = M;
    sl[wave] = l;
  }
  ATTN_STAMP(4)
  __syncthreads();
  ATTN_STAMP(5)
  if (tid < DH) {
    float MM = sm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) MM = fmaxf(MM, sm[i]);
    float o = 0.f, L = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float e = sm[i] > -INFINITY ? __expf(sm[i] - MM) : 0.f;
      o = fmaf(e, so[i][tid], o);
      L = fmaf(e, sl[i], L);
    }
    if constexpr (NSPLIT > 1) {
      // partial of this split: [row][head][split][64] un-normalised, and [row][head][2][NSPLIT] = (max..., sum...)
      part_o[(((size_t)b * H + h) * NSPLIT + sp) * DH + tid] = o;
      if (tid == 0) {
        part_ml[((size_t)b * H + h) * 2 * NSPLIT + sp] = MM;
        part_ml[((size_t)b * H + h) * 2 * NSPLIT + NSPLIT + sp] = L;
      }
    } else {
      stf(ctx + (ctx_bt ? tile_off(b, h * DH + tid, ctx_bt) : (size_t)b * D + h * DH + tid), o / L);
    }
  }
  ATTN_STAMP(6)
}

// ---------------------------------------------------------------------------------------------
// qkv_attn_fused: LN + c_attn projection AND the cache attention of the same layer in ONE launch (decode batches <= 4).
// Of the five phase boundaries of a layer this is the only one that is not an all-to-all: an attention workgroup
// (row, head, key split) needs 192 of the 3 * D projection outputs, written by 24 of the 480 projection workgroups.
// The grid is [projection workgroups | attention workgroups]; the attention workgroups request their K/V cache rows at
// once (blind pairs + the rest once the length is known) and then POLL for q / k / v, which the projection workgroups
// publish as 8-byte {value, tag} granules (one relaxed agent-scope store each: written through, L2-served on any XCD, the
// tag travels with the value so no fence is needed - MI355X_MICROARCH "R2's granule"; tools/ubench_handoff.hip measures
// the hop).  tag = (generation epoch << 12) | (step + 1), read from device memory, so the captured graph never sees a
// stale match.  All workgroups of the launch are co-resident (640 x 256 threads at 2 rows), so the spin cannot starve
// its producers; it is bounded anyway and raises an error flag instead of hanging.  Saves the kernel boundary and
// overlaps the cache stream with the projection: 5.2 + 4.9 us -> see DESIGN.md section 5.
// ---------------------------------------------------------------------------------------------
template <int NB, int RPW, int NCH>
__device__ __forceinline__ void fused_gemv_part(const GemvArgs& g, int blk, unsigned long long* __restrict__ gran,
                                                const int* __restrict__ len, const int* __restrict__ prefix) {
  constexpr int PRO = 1, WAVES = 4;
  constexpr bool XBF = false, W8 = false;
  constexpr int NTHR = WAVES * 64;
  constexpr int EPC = XBF ? 8 : 4;                                  // elements per 16-byte chunk
  constexpr int KCH = (NCH * 512 + NTHR * EPC - 1) / (NTHR * EPC);  // chunks per row per thread
  extern __shared__ __attribute__((aligned(16))) uint32_t sxb[];    // [NB][K/2] bf16 pairs
  __shared__ float red[2][WAVES][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = g.K;
  const float invK = 1.f / (float)K;  // off the critical path: the LayerNorm chain multiplies instead of dividing
  const int n0 = (blk * WAVES + wave) * RPW;
  // ---- 1. activations (+ LayerNorm parameters) first, weights second; all unconditional ----
  u32x4 xr[NB][KCH];  // XBF: 8 bf16; else 4 floats
  f32x4 pm[PRO == 3 ? NB : 1][KCH], pl[PRO == 3 ? NB : 1][KCH], po[PRO == 3 ? NB : 1][KCH][ATTN_NSPLIT][2];
  static_assert(PRO != 3 || (XBF && ATTN_NSPLIT == 4), "prologue 3 feeds bf16 pairs and reads 4 partials as one float4");
  f32x4 gm[PRO == 2 ? KCH : 1], bt[PRO == 2 ? KCH : 1];
  bool xok[KCH];
#pragma unroll
  for (int j = 0; j < KCH; ++j) {
    const int i = (tid + j * NTHR) * EPC;
    xok[j] = i < K;
    const int ic = xok[j] ? i : K - EPC;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const size_t ro = (size_t)min(b, g.B - 1) * K + ic;
      if constexpr (PRO == 3) {
        // x is the attention output, still in ATTN_NSPLIT partials: this thread's 8 dims of head ic / 64
        const size_t bh = (size_t)min(b, g.B - 1) * (K >> 6) + (ic >> 6);
        const float* ml = g.attn_ml + bh * 2 * ATTN_NSPLIT;
        pm[b][j] = *reinterpret_cast<const f32x4*>(ml);
        pl[b][j] = *reinterpret_cast<const f32x4*>(ml + ATTN_NSPLIT);
#pragma unroll
        for (int sp = 0; sp < ATTN_NSPLIT; ++sp) {
          const float* po_ = g.attn_o + (bh * ATTN_NSPLIT + sp) * 64 + (ic & 63);
          po[b][j][sp][0] = *reinterpret_cast<const f32x4*>(po_);
          po[b][j][sp][1] = *reinterpret_cast<const f32x4*>(po_ + 4);
        }
      } else if (XBF) {
        xr[b][j] = *reinterpret_cast<const u32x4*>((const bf16_t*)g.X + ro);
      } else {
        xr[b][j] = *reinterpret_cast<const u32x4*>(g.X + ro);
      }
    }
    if (PRO == 2) {
      gm[j] = *reinterpret_cast<const f32x4*>(g.ln_gamma + ic);
      bt[j] = *reinterpret_cast<const f32x4*>(g.ln_beta + ic);
    }
  }
  float pivot[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) pivot[b] = (PRO == 1 || PRO == 2) ? g.X[(size_t)min(b, g.B - 1) * K] : 0.f;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W;
  const uint8_t* __restrict__ Wq = (const uint8_t*)g.W8;
  u32x4 w[W8 ? 1 : RPW][W8 ? 1 : NCH];
  u32x2 w8[W8 ? RPW : 1][W8 ? NCH : 1];  // 8 fp8 weights per lane and chunk
  const int klast = (NCH - 1) * 512 + lane * 8;
  const bool kok = klast < K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if constexpr (W8)
        w8[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(Wq + (size_t)min(n0 + r, g.N - 1) * K + k));
      else
        w[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + r, g.N - 1) * K + k));
    }
  }
  // epilogue operands of the output this lane will finish, (row lane / NB, batch lane % NB): bias, fp8 row scale and the
  // residual-stream value it accumulates into are requested now (youngest loads, unconditional), so the epilogue has no
  // dependent memory latency of its own
  const int er = min(lane / NB, RPW - 1), eb = lane % NB;
  const int en = min(n0 + er, g.N - 1);
  const float* bp = g.bias ? g.bias : reinterpret_cast<const float*>(W8 ? g.W8 : g.W);  // any readable address when there is no bias
  const float bpre = bp[en];
  const float spre = W8 ? g.wscale[en] : 1.f;
  // (no residual read: the q/k/v outputs are published, not accumulated)
  // every request of this kernel is now in flight.  The fence keeps it that way: without it the machine scheduler sinks
  // most of the weight loads below the first wait on X (fewer live registers), i.e. two thirds of the weight stream
  // would be requested one memory latency late
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. LayerNorm(s) in registers (one barrier each), bf16 pairs to LDS ----
  if (!XBF) {
    float xv[NB][KCH][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[b][j][e] = __uint_as_float(xr[b][j][e]);
#pragma unroll
    for (int pass = 0; pass < PRO; ++pass) {
      float s[NB], q[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        s[b] = q[b] = 0.f;
        const float pv = pass == 0 ? pivot[b] : 0.f;
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float d = xok[j] ? xv[b][j][e] - pv : 0.f;
            s[b] += d;
            q[b] = fmaf(d, d, q[b]);
          }
        s[b] = wave_sum_rl(s[b]);
        q[b] = wave_sum_rl(q[b]);
      }
      if (lane == 0)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          red[pass][wave][2 * b] = s[b];
          red[pass][wave][2 * b + 1] = q[b];
        }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
          S += red[pass][ww][2 * b];
          Q += red[pass][ww][2 * b + 1];
        }
        // contraction pinned (decode_pinned.h): the persistent engine repeats these operations bit for bit
        const float md = __fmul_rn(S, invK);
        const float mean = __fadd_rn(pass == 0 ? pivot[b] : 0.f, md);
        const float rstd = __builtin_amdgcn_rsqf(__fadd_rn(fmaxf(ln_var_rn(Q, invK, md), 0.f), g.ln_eps));
#pragma unroll
        for (int j = 0; j < KCH; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = (xv[b][j][e] - mean) * rstd;
            if (PRO == 2 && pass == 0) v = v * gm[j][e] + bt[j][e];
            xv[b][j][e] = v;
          }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
        if (xok[j]) {
          const int i = (tid + j * NTHR) * 4;
          uint2 p;
          p.x = pack_bf16(xv[b][j][0], xv[b][j][1]);
          p.y = pack_bf16(xv[b][j][2], xv[b][j][3]);
          *reinterpret_cast<uint2*>(sxb + (b * K + i) / 2) = p;
        }
  } else {
    if constexpr (PRO == 3) {
      // merge the split-attention partials: weights exp(max_p - max), one division per head
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < KCH; ++j) {
          const float M = fmaxf(fmaxf(pm[b][j][0], pm[b][j][1]), fmaxf(pm[b][j][2], pm[b][j][3]));
          float wgt[ATTN_NSPLIT], L = 0.f;
#pragma unroll
          for (int sp = 0; sp < ATTN_NSPLIT; ++sp) {
            wgt[sp] = pm[b][j][sp] > -INFINITY ? __expf(pm[b][j][sp] - M) : 0.f;
            L = fmaf(wgt[sp], pl[b][j][sp], L);
          }
          const float inv = 1.f / L;
          float xm[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float a = 0.f;
#pragma unroll
            for (int sp = 0; sp < ATTN_NSPLIT; ++sp) a = fmaf(wgt[sp], po[b][j][sp][e >> 2][e & 3], a);
            xm[e] = a * inv;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) xr[b][j][e] = pack_bf16(xm[2 * e], xm[2 * e + 1]);
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < KCH; ++j)
        if (xok[j]) *reinterpret_cast<u32x4*>(sxb + (b * K + (tid + j * NTHR) * 8) / 2) = xr[b][j];
  }
  __syncthreads();
  // ---- 3. dot products: 4 x v_dot2c per weight fragment and batch row ----
  float acc[RPW][NB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      u32x4 xq = *reinterpret_cast<const u32x4*>(sxb + (b * K + k) / 2);
      if (c == NCH - 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) xq[e] = kok ? xq[e] : 0u;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        if constexpr (W8) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const uint32_t q = w8[r][c][h2];  // 4 fp8: bytes 0,1 -> pair 2*h2, bytes 2,3 -> pair 2*h2 + 1
            acc[r][b] = dot2(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, false)), xq[2 * h2], acc[r][b]);
            acc[r][b] = dot2(__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q, 1.0f, true)), xq[2 * h2 + 1], acc[r][b]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][b] = dot2(w[r][c][e], xq[e], acc[r][b]);
        }
      }
    }
  }
  // wave reduction, then one lane per output: lane l < RPW * NB stores (row l / NB, batch l % NB)
  float mine = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float t = wave_sum_rl(acc[r][b]);
      mine = lane == r * NB + b ? t : mine;
    }
  if (lane < RPW * NB && n0 + er < g.N && eb < g.B) {
    float v = mine * spre + (g.bias ? bpre : 0.f);
    v = g.act == ACT_GELU_NEW ? gelu_new_rn(v) : act_apply(g.act, v);
    // publish as one 8-byte {value, tag} granule (sc1: write-through, L2-served for the polling consumer on any XCD); the
    // tag's two scalars are read here, at the end: nothing in front of the weight requests waits for them
    const unsigned tag = ((unsigned)prefix[1] << 12) | (unsigned)(len[0] + 1);
    const unsigned long long gv = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    __hip_atomic_store(gran + (size_t)eb * g.ldy + en, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <bool ANC>
__device__ __forceinline__ void fused_attn_part(const unsigned long long* __restrict__ gran, unsigned tag, int* __restrict__ err,
                                                bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, const int* __restrict__ len,
                                                const int* __restrict__ kv_start, const int* __restrict__ prefix, int H, int B,
                                                int Smax, float scale, float* __restrict__ part_o, float* __restrict__ part_ml,
                                                const uint8_t* __restrict__ anc, int nb, int h, int b, int sp, int sleep0, int sleep1) {
  typedef bf16_t TC;
  constexpr int NIT = 3, NT = 256, NSPLIT = ATTN_NSPLIT;
  constexpr int DH = 64, VEC = CacheVec<TC>::VEC, LPK = CacheVec<TC>::LPK, NW = NT / 64, SLOTS = NT / LPK;
  constexpr int SD = NT >= 1024 ? 2 : 4;  // rows per slot in flight beyond the register window
  __shared__ float sm[NW], sl[NW];
  __shared__ float so[NW][DH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = H * DH;
  TC* kb = kc + ((size_t)b * H + h) * Smax * DH;
  TC* vb = vc + ((size_t)b * H + h) * Smax * DH;
  const int slot = tid / LPK, sub = tid % LPK;
  // cache row of key j: own row, or (ANC) the physical row the beam's ancestry names for that position
  const uint8_t* arow = nullptr;
  int rowbase = 0;
  if constexpr (ANC) {
    arow = anc + ((size_t)(len[b] & 1) * B + b) * Smax;
    rowbase = (b / nb) * nb;
  }
  auto krow = [&](int j) -> const TC* {
    if constexpr (ANC) return kc + ((size_t)(rowbase + min((int)arow[j], nb - 1)) * H + h) * Smax * DH + (size_t)j * DH;
    return kb + (size_t)j * DH;
  };
  auto vrow = [&](int j) -> const TC* {
    if constexpr (ANC) return vc + ((size_t)(rowbase + min((int)arow[j], nb - 1)) * H + h) * Smax * DH + (size_t)j * DH;
    return vb + (size_t)j * DH;
  };
  // (a) the first pair of key/value rows of this slot, requested before ANYTHING else: their addresses depend on no
  //     device scalar (rows are clamped to the cache capacity; rows >= S are masked out below)
  CacheVec<TC> kr[2 * NIT], vr[2 * NIT];
  // UNC pairs are requested blind (rows past S cost their bytes but nothing waits for the length): 2 pairs cover every
  // prefix; with 1024 threads 4 pairs = 512 rows cover the sequence for most of a generation, so the device-scalar
  // -> load chain (a second full memory latency) only remains for the late steps
  constexpr int UNC = 2;  // 4 blind pairs (512 rows) measured slower: 0.602 vs 0.593 ms per step - the extra bytes cost more than the chain
#pragma unroll
  for (int u = 0; u < UNC; ++u) {
    const int j = min((u * NSPLIT + sp) * SLOTS + slot, Smax - 1);
    kr[u].load(krow(j) + sub * VEC);
    vr[u].load(vrow(j) + sub * VEC);
  }
  // (a') the step's q / k / v come from the projection workgroups of THIS launch: polled below, after every cache load is in flight
  // (b) per-row scalars and the K/V append of this step
  const int pos = prefix[0] + len[b];
  const int S = pos + 1;
  const int ks = kv_start[b];
  // (c) now that S is known: request every remaining row of the sequence at once (one more memory latency in total).
  //     This comes BEFORE anything that consumes the q/k/v slice - vmcnt is in-order, and the K/V append below would
  //     otherwise make the wave sit out the first loads' latency before these are even issued
#pragma unroll
  for (int u = UNC; u < 2 * NIT; ++u)
    if ((u * NSPLIT + sp) * SLOTS < S) {  // block-uniform
      const int j = min((u * NSPLIT + sp) * SLOTS + slot, Smax - 1);
      kr[u].load(krow(j) + sub * VEC);
      vr[u].load(vrow(j) + sub * VEC);
    }
  // (d) the step's own q / k / v: 8-byte {value, tag} granules published by the projection part; every thread needs its
  //     8 q values, slot 0 of split 0 also k and v (append + own key).  One sweep = all loads in flight (sc1: L2-served,
  //     never from this CU's L1), then the tags are checked; bounded, with an error flag instead of a hang.
  float qr[VEC], kown[VEC], vown[VEC];
  {
    const unsigned long long* gq = gran + (size_t)b * 3 * D + h * DH + sub * VEC;
    const bool need_kv = tid < LPK && sp == 0;
    unsigned long long gv[3][VEC];
    int spins = 0;
    // the projection part needs >= 2 us for its weight stream: do not poll (and load the L2 it streams through) before
    // that; afterwards sweep with pauses (MI355X_MICROARCH "polling-cost": pollers next to a weight stream slow it down)
    for (int z = 0; z < sleep0; ++z) __builtin_amdgcn_s_sleep(4);
    for (;;) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) gv[0][i] = __hip_atomic_load(gq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (need_kv) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          gv[1][i] = __hip_atomic_load(gq + D + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gv[2][i] = __hip_atomic_load(gq + 2 * D + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      bool ok = true;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        ok = ok && (unsigned)(gv[0][i] >> 32) == tag;
        if (need_kv) ok = ok && (unsigned)(gv[1][i] >> 32) == tag && (unsigned)(gv[2][i] >> 32) == tag;
      }
      if (ok) break;
      if (++spins > (1 << 18)) {  // ~0.3 s: the producers never take that long; flag it and go on with what is there
        *err = 1;
        break;
      }
      for (int z = 0; z < sleep1; ++z) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      qr[i] = __uint_as_float((unsigned)gv[0][i]) * scale;
      kown[i] = need_kv ? (float)(TC)__uint_as_float((unsigned)gv[1][i]) : 0.f;
      vown[i] = need_kv ? (float)(TC)__uint_as_float((unsigned)gv[2][i]) : 0.f;
    }
  }
  if (tid < LPK && sp == 0) {  // slot 0 (of split 0): its LPK lanes cover the 64 dims
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      stf(kb + (size_t)pos * DH + sub * VEC + i, kown[i]);
      stf(vb + (size_t)pos * DH + sub * VEC + i, vown[i]);
    }
  }
  float m = -INFINITY, l = 0.f, acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  // score of one cached key row for this slot (the LPK lanes of the key hold VEC dims each; DPP sums them: quad swaps,
  // half-row mirror, row mirror - no LDS crossbar trips)
  auto score = [&](const CacheVec<TC>& kk) {
    float sc = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) sc = fmaf(qr[i], kk.get(i), sc);
    sc = dpp_add<0xB1>(sc);
    sc = dpp_add<0x4E>(sc);
    sc = dpp_add<0x141>(sc);
    if (LPK == 16) sc = dpp_add<0x140>(sc);
    return sc;
  };
  // (e) the register window in two phases, as torch.softmax does it: all scores, their maximum, then one exp per key and
  //     the weighted sum - half the VALU work of a per-key online update (no rescale of the accumulator per key), and the
  //     16 waves of a workgroup share 4 SIMDs, so this phase is issue-bound.  The row appended by this step (j == pos)
  //     is masked out of the window and enters as one extra key of slot 0, from registers.  Rows past S multiply by
  //     p = 0: the cache is zero-filled at allocation, so whatever they hold is finite.
  {
    float sc[2 * NIT + 1];
#pragma unroll
    for (int u = 0; u < 2 * NIT; ++u) {
      const int j = (u * NSPLIT + sp) * SLOTS + slot;
      const bool live = u < UNC || (u * NSPLIT + sp) * SLOTS < S;  // block-uniform: was this pair requested
      const float t = live ? score(kr[u]) : 0.f;
      sc[u] = (live && j < S && j >= ks && j != pos) ? t : -INFINITY;
    }
    {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) t = fmaf(qr[i], kown[i], t);
      t = dpp_add<0xB1>(t);
      t = dpp_add<0x4E>(t);
      t = dpp_add<0x141>(t);
      if (LPK == 16) t = dpp_add<0x140>(t);
      sc[2 * NIT] = (slot == 0 && sp == 0) ? t : -INFINITY;
    }
    float mw = sc[0];
#pragma unroll
    for (int u = 1; u <= 2 * NIT; ++u) mw = fmaxf(mw, sc[u]);
    if (mw > -INFINITY) {
#pragma unroll
      for (int u = 0; u < 2 * NIT; ++u)
        if (u < UNC || (u * NSPLIT + sp) * SLOTS < S) {  // block-uniform: pairs that were never requested hold no data at all
          const float p = __expf(sc[u] - mw);  // exp(-inf) = 0 for masked rows
          l += p;
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vr[u].get(i), acc[i]);
        }
      const float p = __expf(sc[2 * NIT] - mw);
      l += p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vown[i], acc[i]);
      m = mw;
    }
  }
  // online update for rows beyond the window (never the appended row when it lies inside the window)
  auto consume = [&](const CacheVec<TC>& kk, const CacheVec<TC>& vv, int j) {
    const bool ok = j < S && j >= ks && j != pos;
    float sc = score(kk);
    sc = ok ? sc : -INFINITY;  // also discards whatever an out-of-range row produced
    const float mn = fmaxf(m, sc);
    const float corr = mn > -INFINITY ? __expf(m - mn) : 1.f;
    const float p = ok ? __expf(sc - mn) : 0.f;
    l = fmaf(l, corr, p);  // contraction pinned: same operation in every build of this loop (decode_pinned.h)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, ok ? vv.get(i) : 0.f, acc[i] * corr);
    m = mn;
  };
  // sequences longer than the register-resident window: stream the rest two rows at a time
  for (int cb = 2 * NIT; (cb * NSPLIT + sp) * SLOTS < S; cb += SD) {  // chunk cb of this split = rows (cb*NSPLIT+sp)*SLOTS ..
    CacheVec<TC> k2[SD], v2[SD];
#pragma unroll
    for (int u = 0; u < SD; ++u) {
      const int j = min(((cb + u) * NSPLIT + sp) * SLOTS + slot, Smax - 1);
      k2[u].load(krow(j) + sub * VEC);
      v2[u].load(vrow(j) + sub * VEC);
    }
#pragma unroll
    for (int u = 0; u < SD; ++u) consume(k2[u], v2[u], ((cb + u) * NSPLIT + sp) * SLOTS + slot);
  }
  // merge the 64/LPK key slots of this wave (lanes with equal `sub`)
  // merge across the wave without the LDS crossbar: lane ^ 8 is a DPP rotate inside the 16-lane row; lane ^ 16 and
  // lane ^ 32 are v_permlane16_swap / v_permlane32_swap (CDNA4), which hand every lane BOTH partners' values
  // (tools/probe_permlane.hip prints the lane maps) - a sum or max of the two results is the butterfly step
  auto bfly_max = [&](float x, int o) {
    if (o == 8) return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true)));
    const u32x2 r = o == 16 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)
                            : __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  auto bfly_sum = [&](float x, int o) {
    if (o == 8) return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, true));
    const u32x2 r = o == 16 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)
                            : __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  };
  float M = m;
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) M = bfly_max(M, o);
  const float sc0 = M > -INFINITY ? __expf(m - M) : 0.f;
  l *= sc0;
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] *= sc0;
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    l = bfly_sum(l, o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = bfly_sum(acc[i], o);
  }
  if (lane < LPK)
#pragma unroll
    for (int i = 0; i < VEC; ++i) so[wave][lane * VEC + i] = acc[i];
  if (lane == 0) {
    sm[wave] = M;
    sl[wave] = l;
  }
  __syncthreads();
  if (tid < DH) {
    float MM = sm[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) MM = fmaxf(MM, sm[i]);
    float o = 0.f, L = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float e = sm[i] > -INFINITY ? __expf(sm[i] - MM) : 0.f;
      o = fmaf(e, so[i][tid], o);
      L = fmaf(e, sl[i], L);
    }
    if constexpr (NSPLIT > 1) {
      // partial of this split: [row][head][split][64] un-normalised, and [row][head][2][NSPLIT] = (max..., sum...)
      part_o[(((size_t)b * H + h) * NSPLIT + sp) * DH + tid] = o;
      if (tid == 0) {
        part_ml[((size_t)b * H + h) * 2 * NSPLIT + sp] = MM;
        part_ml[((size_t)b * H + h) * 2 * NSPLIT + NSPLIT + sp] = L;
      }
    }
  }
}
struct FusedQkvAttn {
  GemvArgs g;
  unsigned long long* gran;  // [B][3 * D] granules of this layer
  int* err;
  bf16_t *kc, *vc;
  const int *len, *kv_start, *prefix;  // prefix[0] = prefix length, prefix[1] = generation epoch
  int H, B, Smax, n_gemv, nb, sleep0, sleep1;
  float scale;
  float *part_o, *part_ml;
  const uint8_t* anc;
};

template <int NB, int RPW, int NCH, bool ANC>
__global__ __launch_bounds__(256) void qkv_attn_fused_kernel(FusedQkvAttn f) {
  // projection workgroups first (dispatching the attention workgroups first was measured worse: they occupy the CUs the
  // projection needs)
  if ((int)blockIdx.x < f.n_gemv) {
    fused_gemv_part<NB, RPW, NCH>(f.g, blockIdx.x, f.gran, f.len, f.prefix);
  } else {
    const unsigned tag = ((unsigned)f.prefix[1] << 12) | (unsigned)(f.len[0] + 1);
    const int idx = (int)blockIdx.x - f.n_gemv;
    const int h = idx % f.H, b = (idx / f.H) % f.B, sp = idx / (f.H * f.B);
    fused_attn_part<ANC>(f.gran, tag, f.err, f.kc, f.vc, f.len, f.kv_start, f.prefix, f.H, f.B, f.Smax, f.scale, f.part_o,
                         f.part_ml, f.anc, f.nb, h, b, sp, f.sleep0, f.sleep1);
  }
}

// ---------------------------------------------------------------------------------------------
// sampler2: repetition penalty + argmax + bookkeeping, one 1024-thread block per row; the per-row length
// counter is advanced by the row's own block (no cross-block step counter, no extra launch).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sampler2_kernel(SamplerArgs a) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ lg = a.logits + (size_t)b * a.V;
  uint8_t* seen = a.seen + (size_t)b * a.V;
  const int k_pre = a.step[b], unf_pre = a.unfinished[b];  // requested with the logits: nothing to wait for after the argmax
  float best = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  };
  auto score = [&](float v, int i) { return sampler_score(a, seen, v, i); };
  // 8 consecutive logits per thread as two 16-byte loads (rows are dword aligned), the tail scalar
  const int nvec = a.V >> 3;
  for (int c = tid; c < nvec; c += 1024) {
    const float4 q0 = *reinterpret_cast<const float4*>(lg + c * 8), q1 = *reinterpret_cast<const float4*>(lg + c * 8 + 4);
    const float q[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) take(score(q[e], c * 8 + e), c * 8 + e);
  }
  for (int i = nvec * 8 + tid; i < a.V; i += 1024) take(score(lg[i], i), i);
  // wave argmax without the LDS crossbar: DPP inside the 16-lane rows, v_permlane16/32_swap across them
  auto merge = [&](float ov, int oi) {
    if (ov > best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  };
#define SAMPLER_DPP(CTRL)                                                                                  \
  merge(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xf, 0xf, true)),       \
        __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, true))
  SAMPLER_DPP(0xB1);
  SAMPLER_DPP(0x4E);
  SAMPLER_DPP(0x141);
  SAMPLER_DPP(0x140);
#undef SAMPLER_DPP
  {
    const u32x2 v16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const u32x2 i16 = __builtin_amdgcn_permlane16_swap((unsigned)bi, (unsigned)bi, false, false);
    best = __uint_as_float(v16[0]);
    bi = (int)i16[0];
    merge(__uint_as_float(v16[1]), (int)i16[1]);
    const u32x2 v32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const u32x2 i32 = __builtin_amdgcn_permlane32_swap((unsigned)bi, (unsigned)bi, false, false);
    best = __uint_as_float(v32[0]);
    bi = (int)i32[0];
    merge(__uint_as_float(v32[1]), (int)i32[1]);
  }
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    sampler_commit(a, b, bi, si, k_pre, unf_pre);
  }
  __syncthreads();
  sampler_next_embedding(a, b, si, tid);
}

// ---------------------------------------------------------------------------------------------
// sampler_sample: the do_sample=True path of HF 4.36.2 GenerationMixin.sample as infer.py:116-124 configures it
// (RepetitionPenaltyLogitsProcessor -> TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> softmax ->
// multinomial), one 1024-thread block per row.  Scores live in LDS; the k-th largest score is found by a 4-pass
// radix select on order-preserving keys (no sort of the vocabulary), the <= 64 survivors are bitonic-sorted by one
// wave, top-p and the draw run serially over them in the order torch.cumsum uses.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned order_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int SMAXC = 128;  // kept candidates (top_k <= 128: the reference web UI offers 0..100)

// value of lane l (wave-uniform l) in every lane: v_readlane_b32, no LDS crossbar round trip
__device__ __forceinline__ float lane_val(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// histogram increment aggregated over the wave (as beam.hip): the digits of scores crowd into a few bins and 64 lanes
// adding to one LDS word serialise; each distinct digit of the wave costs one atomic.  Every lane of the wave calls it.
__device__ __forceinline__ void hist_add_wave(unsigned* hist, unsigned digit, bool act, int lane) {
  unsigned long long m = __ballot(act);
  while (m) {  // wave-uniform
    const int leader = __ffsll((long long)m) - 1;
    const unsigned dl = (unsigned)__shfl((int)digit, leader, 64);
    const unsigned long long same = __ballot(act && digit == dl);
    if (lane == leader) atomicAdd(&hist[dl], (unsigned)__popcll(same));
    m &= ~same;
  }
}

__global__ __launch_bounds__(1024) void sampler_sample_kernel(SamplerArgs a) {
  extern __shared__ float ssc[];  // [V] processed scores
  __shared__ unsigned hist[256];
  __shared__ int s_bin, s_k, s_cnt;
  __shared__ float cval[SMAXC];
  __shared__ int cidx[SMAXC];
  __shared__ int si[2];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float* __restrict__ lg = a.logits + (size_t)b * a.V;
  const uint8_t* seen = a.seen + (size_t)b * a.V;
  const int k_pre = a.step[b], unf_pre = a.unfinished[b];
  for (int i = tid; i < a.V; i += 1024) {
    float v = lg[i];
    if (!a.preprocessed) {
      if (a.penalty != 1.f && seen[i]) v = v < 0.f ? v * a.penalty : v / a.penalty;
      if (a.suppress_stop && i == a.stop) v = -INFINITY;
    }
    if (a.temperature != 1.f) v = v / a.temperature;  // TemperatureLogitsWarper: scores / temperature
    ssc[i] = v;
  }
  // ---- radix select: key of the top_k-th largest score ----
  unsigned prefix = 0;
  int kk = min(a.top_k, a.V);
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = pass * 8;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < a.V; i0 += 1024) {  // every lane takes part in every round (wave-aggregated atomics)
      const int i = i0 + tid;
      const unsigned key = i < a.V ? order_key(ssc[i]) : 0u;
      const bool act = i < a.V && (pass == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8)));
      hist_add_wave(hist, (key >> shift) & 255u, act, lane);
    }
    __syncthreads();
    if (tid < 64) {
      const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const unsigned own = h0 + h1 + h2 + h3;
      unsigned x = own;  // inclusive suffix sum over lanes (higher lanes = larger keys)
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_down(x, off, 64);
        if (lane + off < 64) x += t;
      }
      const unsigned above = x - own;
      if (above < (unsigned)kk && (unsigned)kk <= x) {  // exactly one lane
        unsigned acc = above;
        int bin = 4 * lane + 3;
        const unsigned hb[4] = {h0, h1, h2, h3};
#pragma unroll
        for (int j = 3; j >= 0; --j) {
          if (acc + hb[j] >= (unsigned)kk) {
            bin = 4 * lane + j;
            break;
          }
          acc += hb[j];
        }
        s_bin = bin;
        s_k = kk - (int)acc;
      }
    }
    __syncthreads();
    prefix |= (unsigned)s_bin << shift;
    kk = s_k;
  }
  // ---- gather the survivors (score >= k-th largest; ties kept as HF's `scores < kth` mask keeps them, up to SMAXC) ----
  if (tid == 0) s_cnt = 0;
  if (tid < SMAXC) {
    cval[tid] = -INFINITY;
    cidx[tid] = 0x7fffffff;
  }
  __syncthreads();
  for (int i = tid; i < a.V; i += 1024) {
    const float v = ssc[i];
    // -inf scores never count: with fewer than top_k finite scores HF's `scores < kth` (kth = -inf) keeps exactly the finite ones
    if (order_key(v) >= prefix && v > -INFINITY) {
      const int pos = atomicAdd(&s_cnt, 1);
      if (pos < SMAXC) {
        cval[pos] = v;
        cidx[pos] = i;
      }
    }
  }
  __syncthreads();
  // bitonic sort in LDS by ONE wave (64 lanes = the SMAXC / 2 comparators of a stage; a wave's LDS operations execute in
  // program order, so the 28 stages need no workgroup barrier): descending score, ascending index on ties
  static_assert(SMAXC == 128, "one comparator per lane");
  if (tid < 64) {
    for (int kq = 2; kq <= SMAXC; kq <<= 1)
      for (int j = kq >> 1; j > 0; j >>= 1) {
        const int lo = ((lane & ~(j - 1)) << 1) | (lane & (j - 1)), hi = lo | j;
        const bool up = (lo & kq) == 0;
        const float v0 = cval[lo], v1 = cval[hi];
        const int i0 = cidx[lo], i1 = cidx[hi];
        const bool second_first = v1 > v0 || (v1 == v0 && i1 < i0);
        if (second_first == up) {
          cval[lo] = v1;
          cval[hi] = v0;
          cidx[lo] = i1;
          cidx[hi] = i0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
  }
  __syncthreads();
  if (tid < 64) {
    // wave 0: the exponentials and quotients in parallel (lane r and r + 64 of the sorted candidates), the running sums
    // sequentially in the restatement's order, every lane carrying them (values broadcast lane by lane)
    const int n = min(s_cnt, SMAXC);
    const float m = cval[0];
    const float e0 = lane < n ? expf(cval[lane] - m) : 0.f, e1 = lane + 64 < n ? expf(cval[lane + 64] - m) : 0.f;
    auto ev = [&](int r) { return r < 64 ? lane_val(e0, r) : lane_val(e1, r - 64); };
    float Z = 0.f;
    for (int r = 0; r < n; ++r) Z += ev(r);
    int R = n;
    if (a.top_p < 1.f) {
      // TopPLogitsWarper: ascending cumulative probability <= 1 - top_p is removed; the best token always stays
      const float t0 = e0 / Z, t1 = e1 / Z;
      float tail = 0.f;
      R = 1;
      for (int r = n - 1; r >= 1; --r) {
        tail += r < 64 ? lane_val(t0, r) : lane_val(t1, r - 64);
        if (!(tail <= 1.f - a.top_p)) {
          R = r + 1;
          break;
        }
      }
    }
    float total = 0.f;
    for (int r = 0; r < R; ++r) total += ev(r);
    const int k = k_pre;
    const float u = a.uniforms[(size_t)min(k, a.max_gen - 1) * a.B + b];
    const float target = u * total;
    int pick = R - 1;
    float c = 0.f;
    for (int r = 0; r < R; ++r) {
      c += ev(r);
      if (c >= target) {
        pick = r;
        break;
      }
    }
    if (lane == 0) sampler_commit(a, b, cidx[pick], si, k_pre, unf_pre);
  }
  __syncthreads();
  sampler_next_embedding(a, b, si, tid);
}

template <typename TW>
__global__ void decode_embed2_kernel(float* __restrict__ h, const TW* __restrict__ emb, const TW* __restrict__ pos,
                                     const int* __restrict__ tok, const int* __restrict__ len, int D) {
  const int b = blockIdx.x;
  const int t = tok[b];
  const int p = len[b] + 1;  // positions 0, 2, 3, ... (model.py:153-155)
  for (int i = threadIdx.x; i < D; i += blockDim.x)
    h[(size_t)b * D + i] = ldf(emb + (size_t)t * D + i) + ldf(pos + (size_t)p * D + i);
}

template <typename TW, int NB, int RPW, int NCH>
int launch_gemv2(const GemvArgs& g, hipStream_t s) {
  const int rows_per_block = 4 * RPW;
  dim3 grid((g.N + rows_per_block - 1) / rows_per_block), blk(256);
  const size_t lds = (size_t)NB * g.K * 4;
  hipLaunchKernelGGL((gemv2_kernel<TW, NB, RPW, NCH>), grid, blk, lds, s, g);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

template <typename TW, int NB>
int dispatch_gemv2(const GemvArgs& g, hipStream_t s) {
  // rows per wave chosen so the grid stays >= ~300 workgroups
  const int nch = (g.K + 511) / 512;
  if (nch <= 1) return g.N >= 4096 ? launch_gemv2<TW, NB, 4, 1>(g, s) : launch_gemv2<TW, NB, 1, 1>(g, s);
  if (nch <= 3) return g.N >= 3072 ? launch_gemv2<TW, NB, 2, 3>(g, s) : launch_gemv2<TW, NB, 1, 3>(g, s);
  if (nch <= 10) return launch_gemv2<TW, NB, 1, 10>(g, s);
  set_error("gemv2: K too large");
  return E_INVALID;
}

}  // namespace

bool gemv2_supported(const GemvArgs& g) {
  const int nb = g.B <= 2 ? g.B : 4;
  return g.B >= 1 && g.B <= 4 && g.K % 8 == 0 && g.K <= 5120 && (size_t)nb * g.K * 4 <= 64 * 1024 && g.prologue <= 2;
}

int gemv2(const GemvArgs& g, int tw, hipStream_t s) {
  ITTS_REQUIRE(g.X && g.W && g.Y && g.N > 0, "gemv2: bad args");
  ITTS_REQUIRE(gemv2_supported(g), "gemv2: unsupported shape");
#define GO(TW)                                                  \
  if (g.B == 1) return dispatch_gemv2<TW, 1>(g, s);             \
  if (g.B == 2) return dispatch_gemv2<TW, 2>(g, s);             \
  return dispatch_gemv2<TW, 4>(g, s);
  if (tw == F32) {
    GO(float)
  }
  GO(bf16_t)
#undef GO
}

static bool g_gemv_w5 = false;  // ITTS_GEMV_W5=1: 5-wave workgroups, exactly 256 of them per projection
static int g_gemv_mode = 0;  // ITTS_GEMV_MODE: 0 / 2 block-cooperative kernel (default, measured fastest), 1 wave-autonomous kernel
                             // everywhere, 3 wave-autonomous only for the short bf16-x projection

template <int NB, int RPW, int NCH, int PRO, bool XBF, bool YBF, int WAVES = 4>
static int launch_gemv_bf16(const GemvArgs& g, hipStream_t s) {
  dim3 grid((g.N + WAVES * RPW - 1) / (WAVES * RPW)), blk(WAVES * 64);
  // measured (bench, 2 rows): all block-cooperative 0.602 ms per step, all wave-autonomous 0.615 ms - the per-wave copies
  // of x cost more address-pipeline time (16 clk per KiB wave-load per CU) than the LDS hand-over and its barriers
  const bool block = WAVES != 4 || PRO == 3 || g_gemv_mode == 2 || g_gemv_mode == 0 || (g_gemv_mode == 3 && !(XBF && NCH < 4));
  const size_t lds = (size_t)NB * g.K * 2;
  if (block) {
    if (g.W8)
      hipLaunchKernelGGL((gemv_bf16_kernel<NB, RPW, NCH, PRO, XBF, YBF, true, WAVES>), grid, blk, lds, s, g);
    else
      hipLaunchKernelGGL((gemv_bf16_kernel<NB, RPW, NCH, PRO, XBF, YBF, false, WAVES>), grid, blk, lds, s, g);
  } else {
    if (g.W8)
      hipLaunchKernelGGL((gemv_wave_kernel<NB, RPW, NCH, PRO, XBF, YBF, true>), grid, blk, 0, s, g);
    else
      hipLaunchKernelGGL((gemv_wave_kernel<NB, RPW, NCH, PRO, XBF, YBF, false>), grid, blk, 0, s, g);
  }
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

template <int NB>
static int dispatch_gemv_bf16(const GemvArgs& g, hipStream_t s) {
  const int nch = (g.K + 511) / 512;
  // the shapes of the decode step; rows per wave from the tools/ubench_gemv.hip sweep
  if (nch <= 1) {  // micro configs
    if (g.prologue == 1 && !g.x_bf16 && g.y_bf16) return launch_gemv_bf16<NB, 1, 1, 1, false, true>(g, s);
    if (g.prologue == 1 && !g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 1, 1, 1, false, false>(g, s);
    if (g.prologue == 2 && !g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 1, 1, 2, false, false>(g, s);
    if (g.prologue == 0 && g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 1, 1, 0, true, false>(g, s);
    if (g.prologue == 3 && g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 1, 1, 3, true, false>(g, s);
  } else if (nch <= 3) {
    // one workgroup per CU: 5 waves x RPW rows x 256 workgroups = N (3840 -> RPW 3, 5120 -> RPW 4, 1280 -> RPW 1)
    if (g_gemv_w5 && NB <= 2) {
      if (g.prologue == 1 && !g.x_bf16 && g.y_bf16 && g.N == 5120) return launch_gemv_bf16<NB, 4, 3, 1, false, true, 5>(g, s);
      if (g.prologue == 1 && !g.x_bf16 && !g.y_bf16 && g.N == 3840) return launch_gemv_bf16<NB, 3, 3, 1, false, false, 5>(g, s);
      if (g.prologue == 0 && g.x_bf16 && !g.y_bf16 && g.N == 1280) return launch_gemv_bf16<NB, 1, 3, 0, true, false, 5>(g, s);
      if (g.prologue == 3 && g.x_bf16 && !g.y_bf16 && g.N == 1280) return launch_gemv_bf16<NB, 1, 3, 3, true, false, 5>(g, s);
    }
    if (g.prologue == 1 && !g.x_bf16 && g.y_bf16) return launch_gemv_bf16<NB, 2, 3, 1, false, true>(g, s);    // fc
    if (g.prologue == 1 && !g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 2, 3, 1, false, false>(g, s);  // qkv
    if (g.prologue == 2 && !g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 4, 3, 2, false, false>(g, s);  // head
    if (g.prologue == 0 && g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 2, 3, 0, true, false>(g, s);    // proj
    if (g.prologue == 3 && g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 2, 3, 3, true, false>(g, s);    // proj fed by split attention
  } else if (nch <= 4) {
    if (g.prologue == 0 && g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 2, 4, 0, true, false>(g, s);
  } else if (nch <= 10) {
    if (g_gemv_w5 && NB <= 2 && g.prologue == 0 && g.x_bf16 && !g.y_bf16 && g.N == 1280 && nch == 10)
      return launch_gemv_bf16<NB, 1, 10, 0, true, false, 5>(g, s);
    if (g.prologue == 0 && g.x_bf16 && !g.y_bf16) return launch_gemv_bf16<NB, 2, 10, 0, true, false>(g, s);   // proj2
  }
  set_error("gemv_bf16: no instantiation for this shape");
  return E_INVALID;
}

bool gemv_bf16_supported(const GemvArgs& g) {
  const int nch = (g.K + 511) / 512;
  if (!(g.B >= 1 && g.B <= 4 && g.K % 8 == 0 && g.K >= 64 && nch <= 10)) return false;
  if (g.x_bf16 && g.prologue == 3) return !g.y_bf16 && g.K % 64 == 0 && (nch <= 1 || nch == 3) && g.attn_o && g.attn_ml;
  if (g.x_bf16) return g.prologue == 0 && !g.y_bf16 && (nch <= 1 || nch == 3 || nch == 4 || (nch > 4 && nch <= 10));
  if (g.prologue == 0 || nch > 3) return false;
  return !(g.prologue == 2 && g.y_bf16);
}

int gemv_bf16(const GemvArgs& g, hipStream_t s) {
  static const bool once = [] {
    const char* m = getenv("ITTS_GEMV_MODE");
    g_gemv_mode = m ? atoi(m) : 0;
    g_gemv_w5 = getenv("ITTS_GEMV_W5") != nullptr;
    return true;
  }();
  (void)once;
  ITTS_REQUIRE((g.X || g.prologue == 3) && (g.W || g.W8) && g.Y && g.N > 0, "gemv_bf16: bad args");
  ITTS_REQUIRE(!g.W8 || g.wscale, "gemv_bf16: fp8 weights need their row scales");
  ITTS_REQUIRE(gemv_bf16_supported(g), "gemv_bf16: unsupported shape");
  ITTS_REQUIRE(!(g.accumulate && g.y_bf16), "gemv_bf16: accumulate needs an fp32 output");
  if (g.B == 1) return dispatch_gemv_bf16<1>(g, s);
  if (g.B == 2) return dispatch_gemv_bf16<2>(g, s);
  if (g.B == 3) return dispatch_gemv_bf16<3>(g, s);  // one sentence x 3 beams, the reference's default generate() mode
  return dispatch_gemv_bf16<4>(g, s);
}

int decode_attn2(void* ctx, int to, const float* qkv, void* kc, void* vc, const int* len, const int* kv_start,
                 const int* prefix_dev, int B, int H, int dh, int Smax, int tc, hipStream_t s, int ctx_tiled, float* part_o,
                 float* part_ml, const uint8_t* anc, int nb) {
  ITTS_REQUIRE(dh == 64, "decode_attn2: head dim must be 64");
  ITTS_REQUIRE(!anc || (nb >= 1 && nb <= 16 && B % nb == 0), "decode_attn2: beam ancestry needs B to be a multiple of 1 <= nb <= 16");
  if (part_o) {  // split form: 4 workgroups of 256 threads per (row, head), partials merged by the projection GEMV
    ITTS_REQUIRE(part_ml && tc == BF16, "decode_attn2: split form needs both partial buffers and a bf16 cache");
    const float scale = 1.f / sqrtf((float)dh);
    if (anc)
      hipLaunchKernelGGL((decode_attn2_kernel<bf16_t, bf16_t, 3, 256, ATTN_NSPLIT, true>), dim3(H, B, ATTN_NSPLIT), dim3(256), 0, s,
                         (bf16_t*)nullptr, qkv, (bf16_t*)kc, (bf16_t*)vc, len, kv_start, prefix_dev, H, Smax, scale, 0, part_o,
                         part_ml, anc, nb);
    else
      hipLaunchKernelGGL((decode_attn2_kernel<bf16_t, bf16_t, 3, 256, ATTN_NSPLIT>), dim3(H, B, ATTN_NSPLIT), dim3(256), 0, s, (bf16_t*)nullptr,
                         qkv, (bf16_t*)kc, (bf16_t*)vc, len, kv_start, prefix_dev, H, Smax, scale, 0, part_o, part_ml);
    ITTS_HIP_CHECK(hipGetLastError());
    return OK;
  }
  ITTS_REQUIRE(!ctx_tiled || to == BF16, "decode_attn2: tiled ctx is bf16 only");
  const float scale = 1.f / sqrtf((float)dh);
  dim3 grid(H, B);
  const int bt = ctx_tiled ? (B + 15) / 16 : 0;
  const bool many = (long)B * H >= 512;
#ifndef ATTN_NIT_MANY
#define ATTN_NIT_MANY 8
#endif
#define LAUNCH(TCT, TOT)                                                                                                  \
  if (anc && many)                                                                                                        \
    hipLaunchKernelGGL((decode_attn2_kernel<TCT, TOT, ATTN_NIT_MANY, 256, 1, true>), grid, dim3(256), 0, s, (TOT*)ctx, qkv, (TCT*)kc,   \
                       (TCT*)vc, len, kv_start, prefix_dev, H, Smax, scale, bt, nullptr, nullptr, anc, nb);               \
  else if (anc)                                                                                                           \
    hipLaunchKernelGGL((decode_attn2_kernel<TCT, TOT, 3, 1024, 1, true>), grid, dim3(1024), 0, s, (TOT*)ctx, qkv, (TCT*)kc, (TCT*)vc,  \
                       len, kv_start, prefix_dev, H, Smax, scale, bt, nullptr, nullptr, anc, nb);                         \
  else if (many)                                                                                                          \
    hipLaunchKernelGGL((decode_attn2_kernel<TCT, TOT, ATTN_NIT_MANY, 256>), grid, dim3(256), 0, s, (TOT*)ctx, qkv, (TCT*)kc, (TCT*)vc, len, \
                       kv_start, prefix_dev, H, Smax, scale, bt);                                                         \
  else                                                                                                                    \
    hipLaunchKernelGGL((decode_attn2_kernel<TCT, TOT, 3, 1024>), grid, dim3(1024), 0, s, (TOT*)ctx, qkv, (TCT*)kc, (TCT*)vc,    \
                       len, kv_start, prefix_dev, H, Smax, scale, bt);
  if (tc == F32 && to == F32) {
    LAUNCH(float, float)
  } else if (tc == BF16 && to == BF16) {
    LAUNCH(bf16_t, bf16_t)
  } else if (tc == BF16 && to == F32) {
    LAUNCH(bf16_t, float)
  } else {
    set_error("decode_attn2: dtype combination");
    return E_INVALID;
  }
#undef LAUNCH
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

bool qkv_attn_fused_supported(const GemvArgs& g, int H, int dh) {
  const int nch = (g.K + 511) / 512;
  return g.B >= 1 && g.B <= 4 && dh == 64 && g.N == 3 * H * dh && g.K % 8 == 0 && g.K >= 64 && (nch <= 1 || nch == 3) &&
         g.prologue == 1 && !g.x_bf16 && !g.y_bf16 && !g.accumulate && !g.W8 && g.act == ACT_NONE;
}

int qkv_attn_fused(const GemvArgs& g, unsigned long long* gran, int* err, void* kc, void* vc, const int* len,
                   const int* kv_start, const int* prefix_dev, int H, int dh, int Smax, float* part_o, float* part_ml,
                   const uint8_t* anc, int nb, hipStream_t s) {
  ITTS_REQUIRE(qkv_attn_fused_supported(g, H, dh) && gran && err && part_o && part_ml, "qkv_attn_fused: unsupported call");
  ITTS_REQUIRE(!anc || (nb >= 1 && nb <= 16 && g.B % nb == 0), "qkv_attn_fused: beam ancestry needs B to be a multiple of nb");
  const int nch = (g.K + 511) / 512, rpw = nch <= 1 ? 1 : 2;
  FusedQkvAttn f;
  f.g = g;
  f.gran = gran;
  f.err = err;
  f.kc = (bf16_t*)kc;
  f.vc = (bf16_t*)vc;
  f.len = len;
  f.kv_start = kv_start;
  f.prefix = prefix_dev;
  f.H = H;
  f.B = g.B;
  f.Smax = Smax;
  f.n_gemv = (g.N + 4 * rpw - 1) / (4 * rpw);
  f.nb = nb;
  f.scale = 1.f / sqrtf((float)dh);
  f.part_o = part_o;
  f.part_ml = part_ml;
  f.anc = anc;
  static const int e_s0 = getenv("ITTS_FUSE_SLEEP0") ? atoi(getenv("ITTS_FUSE_SLEEP0")) : 24;  // x 256 clocks before the first poll (best of tools/fuse_sweep.sh)
  static const int e_s1 = getenv("ITTS_FUSE_SLEEP1") ? atoi(getenv("ITTS_FUSE_SLEEP1")) : 4;   // x 64 clocks between sweeps
  f.sleep0 = e_s0;
  f.sleep1 = e_s1;
  const dim3 grid(f.n_gemv + H * g.B * ATTN_NSPLIT), blk(256);
  const size_t lds = (size_t)(g.B <= 2 ? g.B : 4) * g.K * 2;
#define FUSED_GO(NB, RPW, NCH)                                                                            \
  if (anc) hipLaunchKernelGGL((qkv_attn_fused_kernel<NB, RPW, NCH, true>), grid, blk, lds, s, f);          \
  else hipLaunchKernelGGL((qkv_attn_fused_kernel<NB, RPW, NCH, false>), grid, blk, lds, s, f);
#define FUSED_NB(RPW, NCH)            \
  if (g.B == 1) { FUSED_GO(1, RPW, NCH) } \
  else if (g.B == 2) { FUSED_GO(2, RPW, NCH) } \
  else { FUSED_GO(4, RPW, NCH) }
  if (nch <= 1) { FUSED_NB(1, 1) } else { FUSED_NB(2, 3) }
#undef FUSED_NB
#undef FUSED_GO
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int sampler2_step(const SamplerArgs& a, int B, hipStream_t s) {
  if (a.do_sample) {
    ITTS_REQUIRE(a.uniforms && a.top_k >= 1 && a.top_k <= 128 && a.temperature > 0.f && a.top_p > 0.f && a.B == B,
                 "sampler: sampling needs uniforms, 1 <= top_k <= 128, temperature > 0, top_p > 0");
    ITTS_REQUIRE((size_t)a.V * 4 <= 60 * 1024, "sampler: vocabulary too large for the LDS-resident sampler");
    hipLaunchKernelGGL(sampler_sample_kernel, dim3(B), dim3(1024), (size_t)a.V * 4, s, a);
    ITTS_HIP_CHECK(hipGetLastError());
    return OK;
  }
  hipLaunchKernelGGL(sampler2_kernel, dim3(B), dim3(1024), 0, s, a);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int decode_embed2(float* h, const void* emb, const void* pos, const int* tok, const int* len, int B, int D, int tw,
                  hipStream_t s) {
  if (tw == F32)
    hipLaunchKernelGGL(decode_embed2_kernel<float>, dim3(B), dim3(256), 0, s, h, (const float*)emb, (const float*)pos, tok, len, D);
  else
    hipLaunchKernelGGL(decode_embed2_kernel<bf16_t>, dim3(B), dim3(256), 0, s, h, (const bf16_t*)emb, (const bf16_t*)pos, tok, len, D);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
