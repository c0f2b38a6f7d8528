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

constexpr int SW = 8;  // waves per workgroup (K split inside the workgroup)


// NT = 16-feature tiles per workgroup (each wave multiplies the same X fragments into NT weight tiles: X is the larger
// L2->CU stream at B = 64, so two tiles per workgroup halve it per output); gridDim.y = K split across workgroups
// (ksplit > 1: raw partial sums go to g.partial[split][b][n]; bias / residual are applied by ln_rows_bf16).
// W8: the weights are OCP fp8 e4m3 bytes with one power-of-two scale per output row (BASELINE config 5: half the weight
// stream); a lane's 8 bytes become the same bf16x8 MFMA operand through v_cvt_scalef32_pk_bf16_fp8, the row scale
// multiplies the fp32 sum in the epilogue - bit-identical to running the bf16 dequantisation of the same weights.
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// LNP (B <= 16 only, BT = 1): X is the fp32 residual stream [B][K]; LayerNorm without affine (folded into W) runs in the
// prologue - wave w owns rows 2w and 2w + 1, moments by DPP wave sums, the normalised bf16 rows go to LDS and every wave
// reads its k-steps' fragments from there.  At 5-16 rows the step is launch-bound (7 launches a layer at ~5 us), so
// folding the two LayerNorm launches of a layer into the projections they feed is worth more than their redundant
// compute (80 KB of h per workgroup, L2 hits).
// HALF (BT = 1, NT = 1): a workgroup owns 8 of a tile's 16 features (lanes of the other 8 load nothing and store nothing):
// twice the workgroups for the narrow residual projections (N = D: 160 instead of 80), so they need no K split across
// workgroups and accumulate straight into the residual stream - no partial sums for a LayerNorm launch to absorb.
template <int BT, int NT, int SU, bool W8 = false, bool WT = false, bool LNP = false, bool HALF = false>
__global__ __launch_bounds__(512) void skinny_mfma_kernel(GemvArgs g) {
  static_assert(!(LNP || HALF) || BT == 1, "LayerNorm prologue / half tiles are the <= 16-row forms");
  static_assert(!HALF || NT == 1, "half tiles: one tile per workgroup");
  __shared__ float red[SW][NT * BT][256];
  extern __shared__ __attribute__((aligned(16))) unsigned short xs[];  // LNP: [16][K + 8] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int n0 = HALF ? (blockIdx.x >> 1) * 16 : blockIdx.x * (16 * NT);
  const bool wlive = !HALF || (fr >> 3) == (int)(blockIdx.x & 1);  // HALF: this lane's feature belongs to this workgroup
  const bf16_t* __restrict__ W = (const bf16_t*)(WT ? g.Wt : g.W);
  const bf16_t* __restrict__ X = (const bf16_t*)g.X;
  const int K = g.K, nks_all = K >> 5;
  constexpr int WSTEP = WT ? 512 : 32;  // elements from one k-step's fragment to the next
  const int S = gridDim.y, split = blockIdx.y;
  const int per = (nks_all + S - 1) / S;
  const int ks0 = split * per, nks = min(per, nks_all - ks0);  // this workgroup's k-steps [ks0, ks0 + nks)
  const bf16_t* wp[NT];
  const uint8_t* wq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const size_t off = (size_t)min(n0 + t * 16 + fr, g.N - 1) * K + fg * 8 + (size_t)ks0 * 32;
    // tiled: tile (n0/16 + t) clamped to the last one (a workgroup's surplus tile re-reads it; its outputs are masked)
    wp[t] = WT ? W + (((size_t)min(n0 / 16 + t, (g.N + 15) / 16 - 1) * nks_all + ks0) * 64 + lane) * 8 : W + off;
    wq[t] = WT ? (const uint8_t*)g.W8t + (((size_t)min(n0 / 16 + t, (g.N + 15) / 16 - 1) * nks_all + ks0) * 64 + lane) * 8
               : (const uint8_t*)g.W8 + off;
  }
  // X either row-major [B][K] or fragment-tiled [K/32][BT][64 lanes][8] (tile_off): one MFMA operand = 1 KiB contiguous
  const int btr = (g.B + 15) >> 4;  // batch tiles that exist (the template rounds up to 1 / 2 / 4 / 8)
  const int xstep = g.x_tiled ? btr * 512 : 32;
  const bf16_t* xp[BT];
#pragma unroll
  for (int bt = 0; bt < BT; ++bt)
    xp[bt] = g.x_tiled ? X + ((size_t)ks0 * btr + min(bt, btr - 1)) * 512 + lane * 8
                       : X + (size_t)min(bt * 16 + fr, g.B - 1) * K + fg * 8 + (size_t)ks0 * 32;
  // LNP: this wave's two rows of the residual stream, requested before the weights (L2 hits: they return while the HBM
  // weight requests behind them are in flight - loads return in issue order)
  constexpr int LNE = 8;  // float4 per lane and row: K <= 2048
  f32x4v hx[LNP ? 2 : 1][LNP ? LNE : 1];
  float hpiv[2] = {0.f, 0.f};
  if constexpr (LNP) {
    const float* Hs = g.X;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float* hr = Hs + (size_t)min(wave * 2 + r, g.B - 1) * K;
      hpiv[r] = hr[0];
#pragma unroll
      for (int i = 0; i < LNE; ++i)
        if (i * 256 < K) hx[r][i] = *reinterpret_cast<const f32x4v*>(hr + min(i * 256 + lane * 4, K - 4));
    }
  }
  f32x4v acc[NT][BT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) acc[t][bt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // epilogue operands (bias of the outputs this thread will finish) requested up front: the tail then has no
  // dependent memory latency
  constexpr int EPT = (NT * BT * 256 + 511) / 512;
  float bpre[EPT], spre[EPT], ypre[HALF ? EPT : 1];
  {
    const float* bp = g.bias ? g.bias : (W8 ? g.wscale : reinterpret_cast<const float*>(W));  // any readable address without a bias
    const bool rmw = HALF && g.accumulate && !g.y_bf16;  // the residual-stream value this thread will add into (HALF: the
                                                         // launch-bound <= 16-row form, where the late read-modify-write
                                                         // of the epilogue is 1.5 us of dependent latency)
#pragma unroll
    for (int it = 0; it < EPT; ++it) {
      const int idx = tid + it * 512, tb = idx >> 8, t = tb / BT, bt = tb - t * BT, e = idx & 255;
      const int nn = min(n0 + t * 16 + (idx & 15), g.N - 1);
      bpre[it] = bp[nn];
      spre[it] = W8 ? g.wscale[nn] : 1.f;
      if constexpr (HALF) {
        const int bb = min(bt * 16 + ((e & 63) >> 4) * 4 + (e >> 6), g.B - 1);
        ypre[it] = rmw ? g.Y[(size_t)bb * g.ldy + nn] : 0.f;
      }
    }
  }
  // LNP: every wave passes through the body exactly once (nks <= SW * SU, host-checked), also one without a k-step of its
  // own - it still normalises its two rows and joins the barrier
  for (int k0 = wave; k0 < nks || (LNP && k0 == wave); k0 += SW * SU) {
    bf16x8 wf[SU][NT], xf[SU][BT];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = k0 + u * SW;
      if (ks >= nks) break;  // wave-uniform: a wave with fewer k-steps than SU requests only its own (no duplicate ingest)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (HALF && !wlive) {
          wf[u][t] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        } else if constexpr (W8) {
          const u32x2v q = __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(wq[t] + (size_t)ks * WSTEP));
          u32x4v w4;  // bytes 0,1 | 2,3 of each dword -> one bf16 pair each (same pairing as gemv_bf16_kernel<W8>)
          w4[0] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[0], 1.0f, false));
          w4[1] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[0], 1.0f, true));
          w4[2] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[1], 1.0f, false));
          w4[3] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[1], 1.0f, true));
          wf[u][t] = __builtin_bit_cast(bf16x8, w4);
        } else {
          wf[u][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * WSTEP));
        }
      }
    }
    if constexpr (LNP) {
      const int xld = K + 8;  // bf16 elements per LDS row
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = wave * 2 + r;
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < LNE; ++i)
          if (i * 256 < K) {
            const bool ok = i * 256 + lane * 4 < K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float d = ok ? hx[r][i][e] - hpiv[r] : 0.f;  // moments about the row's first element, as ln_rows_bf16
              sm += d;
              sq = fmaf(d, d, sq);
            }
          }
        sm = wave_sum(sm);
        sq = wave_sum(sq);
        const float invK = 1.f / (float)K, md = sm * invK, mean = hpiv[r] + md;
        const float rstd = rsqrtf(fmaxf(sq * invK - md * md, 0.f) + g.ln_eps);
#pragma unroll
        for (int i = 0; i < LNE; ++i)
          if (i * 256 < K && i * 256 + lane * 4 < K) {
            bf16_t o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = row < g.B ? (bf16_t)((hx[r][i][e] - mean) * rstd) : (bf16_t)0.f;
            *reinterpret_cast<uint2*>(xs + (size_t)row * xld + i * 256 + lane * 4) = *reinterpret_cast<const uint2*>(o4);
          }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int ks = k0 + u * SW;
        if (ks >= nks) break;
        xf[u][0] = *reinterpret_cast<const bf16x8*>(xs + (size_t)fr * xld + ((size_t)ks0 + ks) * 32 + fg * 8);
      }
    } else {
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int ks = k0 + u * SW;
        if (ks >= nks) break;
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) {
          if (HALF && fr >= g.B)  // rows past the batch: nothing to fetch (their outputs are never stored)
            xf[u][bt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
          else
            xf[u][bt] = *reinterpret_cast<const bf16x8*>(xp[bt] + (size_t)ks * xstep);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      if (k0 + u * SW < nks) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int bt = 0; bt < BT; ++bt)
            acc[t][bt] = half_mfma16(xf[u][bt], wf[u][t], acc[t][bt]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t * BT + bt][r * 64 + lane] = acc[t][bt][r];
  __syncthreads();
#pragma unroll
  for (int it = 0; it < EPT; ++it) {
    const int idx = tid + it * 512;
    if (idx >= NT * BT * 256) break;
    const int tb = idx >> 8, t = tb / BT, bt = tb - t * BT, e = idx & 255, r = e >> 6, l = e & 63;
    const int b = bt * 16 + (l >> 4) * 4 + r, n = n0 + t * 16 + (l & 15);
    if (b >= g.B || n >= g.N) continue;
    if (HALF && ((l & 15) >> 3) != (int)(blockIdx.x & 1)) continue;  // the other workgroup of the pair owns this feature
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) v += red[w][tb][e];
    if (W8) v *= spre[it];
    const size_t o = (size_t)b * g.ldy + n;
    if (S > 1) {
      g.partial[((size_t)split * g.B + b) * g.ldy + n] = v;
      continue;
    }
    if (g.bias) v += bpre[it];
    v = act_apply(g.act, v);
    if (g.y_bf16)
      reinterpret_cast<bf16_t*>(g.Y)[g.y_tiled ? tile_off(b, n, btr) : o] = (bf16_t)v;
    else if (g.accumulate)
      g.Y[o] = (HALF ? ypre[HALF ? it : 0] : g.Y[o]) + v;
    else
      g.Y[o] = v;
  }
}

// y(bf16) = LN(x) [-> LN again for the head]; optional affine on the first pass.  With nsplit > 0 the row first absorbs a
// split-K projection: x += bias + sum_s partial[s] (fixed order), written back to the fp32 residual stream.
template <int MAXE>
__global__ __launch_bounds__(256) void ln_rows_bf16_kernel(bf16_t* __restrict__ y, float* __restrict__ x,
                                                           const float* __restrict__ g1, const float* __restrict__ b1,
                                                           int D, float eps, int passes, const float* __restrict__ partial,
                                                           int nsplit, const float* __restrict__ pbias, int rows, int y_tiled) {
  __shared__ float red[2][2][4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float v[MAXE];
  float* xr = x + (size_t)row * D;
  const float pv0 = xr[0];
#pragma unroll
  for (int i = 0; i < MAXE; ++i) v[i] = tid + i * 256 < D ? xr[tid + i * 256] : 0.f;
  if (nsplit > 0) {
    // all partial loads of a thread are issued together (up to 8 splits x MAXE elements), then summed in split order
    float pv[8][MAXE], bv[MAXE];
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int c = min(tid + i * 256, D - 1);
      bv[i] = pbias ? pbias[c] : 0.f;
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) pv[sp][i] = sp < nsplit ? partial[((size_t)sp * rows + row) * D + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int c = tid + i * 256;
      if (c < D) {
        float a = bv[i];
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) a += pv[sp][i];
        v[i] += a;
        xr[c] = v[i];
      }
    }
  }
  for (int pass = 0; pass < passes; ++pass) {
    // one reduction round per pass: moments about a pivot (the row's first element as it was before this kernel's
    // residual add - any value near the mean will do - for the raw residual stream, zero for a LayerNorm output), sum
    // and sum of squares travel together - one barrier instead of two
    const float pv = pass == 0 ? pv0 : 0.f;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const float d = tid + i * 256 < D ? v[i] - pv : 0.f;
      s += d;
      q = fmaf(d, d, q);
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) {
      red[pass][0][wave] = s;
      red[pass][1][wave] = q;
    }
    __syncthreads();
    const float invD = 1.f / D;
    const float md = (red[pass][0][0] + red[pass][0][1] + red[pass][0][2] + red[pass][0][3]) * invD;
    const float mean = pv + md;
    const float var = fmaxf((red[pass][1][0] + red[pass][1][1] + red[pass][1][2] + red[pass][1][3]) * invD - md * md, 0.f);
    const float rstd = rsqrtf(var + eps);
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
    if (tid + i * 256 < D) {
      const int c = tid + i * 256;
      y[y_tiled ? tile_off(row, c, (rows + 15) >> 4) : (size_t)row * D + c] = (bf16_t)v[i];
    }
}

// one thread per 16-byte group of the tiled copy: (tile, k-step, lane) <- W[tile*16 + lane%16][k-step*32 + lane/16*8 .. +8]
__global__ __launch_bounds__(256) void retile_weights_kernel(u32x4v* __restrict__ dst, const bf16_t* __restrict__ src, int N, int K,
                                                             size_t groups) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups) return;
  const int lane = (int)(i & 63), nks = K >> 5;
  const size_t tk = i >> 6;
  const int ks = (int)(tk % nks), tile = (int)(tk / nks);
  const int n = tile * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
  dst[i] = n < N ? *reinterpret_cast<const u32x4v*>(src + (size_t)n * K + k) : u32x4v{0u, 0u, 0u, 0u};
}

// fp8: one thread per 8-byte group
__global__ __launch_bounds__(256) void retile_weights8_kernel(u32x2v* __restrict__ dst, const uint8_t* __restrict__ src, int N, int K,
                                                              size_t groups) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= groups) return;
  const int lane = (int)(i & 63), nks = K >> 5;
  const size_t tk = i >> 6;
  const int ks = (int)(tk % nks), tile = (int)(tk / nks);
  const int n = tile * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8;
  dst[i] = n < N ? *reinterpret_cast<const u32x2v*>(src + (size_t)n * K + k) : u32x2v{0u, 0u};
}

}  // namespace

int retile_weights_fp8(void* dst, const void* src, int N, int K, hipStream_t s) {
  ITTS_REQUIRE(dst && src && N > 0 && K > 0 && K % 32 == 0, "retile_weights_fp8: K must be a multiple of 32");
  ITTS_REQUIRE(!(((uintptr_t)dst | (uintptr_t)src) & 7), "retile_weights_fp8: pointers must be 8-byte aligned");
  const size_t groups = (size_t)((N + 15) / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(retile_weights8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (u32x2v*)dst, (const uint8_t*)src, N, K,
                     groups);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int retile_weights_bf16(void* dst, const void* src, int N, int K, hipStream_t s) {
  ITTS_REQUIRE(dst && src && N > 0 && K > 0 && K % 32 == 0, "retile_weights_bf16: K must be a multiple of 32");
  ITTS_REQUIRE(!(((uintptr_t)dst | (uintptr_t)src) & 15), "retile_weights_bf16: pointers must be 16-byte aligned");
  const size_t groups = (size_t)((N + 15) / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(retile_weights_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, (u32x4v*)dst, (const bf16_t*)src, N, K,
                     groups);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

bool skinny_mfma_supported(const GemvArgs& g) {
  if (g.prologue == 1) {
    if (!(g.B <= 16 && !g.x_bf16 && !g.half_tiles && g.K <= 1280 && g.ksplit == 1)) return false;
  } else if (!(g.x_bf16 && g.prologue == 0)) {
    return false;
  }
  if (g.half_tiles && !(g.B <= 16 && g.ksplit == 1)) return false;
  return g.B >= 1 && g.B <= 128 && g.K % 32 == 0 && !(g.accumulate && g.y_bf16) &&
         !(((uintptr_t)g.W | (uintptr_t)g.X) & 15) && !((uintptr_t)g.W8 & 7) && (!g.W8 || g.wscale) && g.ksplit >= 1 &&
         (g.ksplit == 1 || (g.partial && g.act == ACT_NONE));
}

// <= 16 rows: LayerNorm prologue (prologue == 1, X = fp32 residual stream) or half-tile residual projection (half_tiles)
template <bool LNP, bool HALF>
static int launch_skinny16(const GemvArgs& g, hipStream_t s) {
  const int tiles = (g.N + 15) / 16;
  const bool nt2 = !HALF && tiles >= 300;
  dim3 grid(HALF ? tiles * 2 : nt2 ? (tiles + 1) / 2 : tiles), blk(512);
  const size_t lds = LNP ? (size_t)16 * (g.K + 8) * 2 : 0;
  // HALF has no K split: a deep projection (K = 4 D: 160 k-steps, 20 per wave) keeps all of them in flight at once
  // (SU = 20: 160 VGPRs of operands at one tile per wave) instead of four dependent rounds of 5
  const bool deep = HALF && (g.K >> 5) > SW * 5;
#define GO(W8_, WT_)                                                                                           \
  {                                                                                                            \
    if constexpr (HALF) {                                                                                      \
      if (deep)                                                                                                \
        hipLaunchKernelGGL((skinny_mfma_kernel<1, 1, 20, W8_, WT_, LNP, true>), grid, blk, lds, s, g);         \
      else                                                                                                     \
        hipLaunchKernelGGL((skinny_mfma_kernel<1, 1, 5, W8_, WT_, LNP, true>), grid, blk, lds, s, g);          \
    } else if (nt2)                                                                                              \
      hipLaunchKernelGGL((skinny_mfma_kernel<1, 2, 5, W8_, WT_, LNP, false>), grid, blk, lds, s, g);           \
    else                                                                                                       \
      hipLaunchKernelGGL((skinny_mfma_kernel<1, 1, 5, W8_, WT_, LNP, false>), grid, blk, lds, s, g);           \
  }
  if (g.W8 && g.W8t) GO(true, true)
  else if (g.W8) GO(true, false)
  else if (g.Wt) GO(false, true)
  else GO(false, false)
#undef GO
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

template <int BT>
static int launch_skinny(const GemvArgs& g, hipStream_t s) {
  // two feature tiles per workgroup once there are plenty of tiles (halves the X stream per output); otherwise one
  const int tiles = (g.N + 15) / 16;
  const bool nt2 = tiles * g.ksplit >= 300 && BT <= 4;
  dim3 grid(nt2 ? (tiles + 1) / 2 : tiles, g.ksplit), blk(512);
  if (g.W8 && g.W8t) {
    if (nt2)
      hipLaunchKernelGGL((skinny_mfma_kernel<BT, 2, 5, true, true>), grid, blk, 0, s, g);
    else
      hipLaunchKernelGGL((skinny_mfma_kernel<BT, 1, 5, true, true>), grid, blk, 0, s, g);
  } else if (g.W8) {
    if (nt2)
      hipLaunchKernelGGL((skinny_mfma_kernel<BT, 2, 5, true>), grid, blk, 0, s, g);
    else
      hipLaunchKernelGGL((skinny_mfma_kernel<BT, 1, 5, true>), grid, blk, 0, s, g);
  } else if (g.Wt) {
    if (nt2)
      hipLaunchKernelGGL((skinny_mfma_kernel<BT, 2, 5, false, true>), grid, blk, 0, s, g);
    else
      hipLaunchKernelGGL((skinny_mfma_kernel<BT, 1, 5, false, true>), grid, blk, 0, s, g);
  } else if (nt2)
    hipLaunchKernelGGL((skinny_mfma_kernel<BT, 2, 5>), grid, blk, 0, s, g);
  else
    hipLaunchKernelGGL((skinny_mfma_kernel<BT, 1, 5>), grid, blk, 0, s, g);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int skinny_mfma(const GemvArgs& g, hipStream_t s) {
  ITTS_REQUIRE(g.X && (g.W || g.W8 || g.Wt) && (g.Y || g.ksplit > 1) && g.N > 0, "skinny_mfma: bad args");
  ITTS_REQUIRE(!g.Wt || !((uintptr_t)g.Wt & 15), "skinny_mfma: tiled weights must be 16-byte aligned");
  ITTS_REQUIRE(!g.W8t || (g.W8 && !((uintptr_t)g.W8t & 7)), "skinny_mfma: tiled fp8 weights come with the row-major ones, 8-byte aligned");
  ITTS_REQUIRE(skinny_mfma_supported(g), "skinny_mfma: unsupported shape");
  ITTS_REQUIRE(g.ksplit <= (g.K >> 5), "skinny_mfma: ksplit larger than the number of k-steps");
  const int bt = (g.B + 15) / 16;
  if (g.prologue == 1 || g.half_tiles) {
    ITTS_REQUIRE(bt == 1 && g.ksplit == 1, "skinny_mfma: LayerNorm prologue / half tiles are for <= 16 rows, no K split");
    if (g.prologue == 1) {
      ITTS_REQUIRE(!g.x_bf16 && !g.half_tiles && g.K <= 1280 && !((uintptr_t)g.X & 15),
                   "skinny_mfma: LayerNorm prologue needs fp32 X [B, K], K <= 1280");
      return launch_skinny16<true, false>(g, s);
    }
    return launch_skinny16<false, true>(g, s);
  }
  if (bt <= 1) return launch_skinny<1>(g, s);
  if (bt <= 2) return launch_skinny<2>(g, s);
  if (bt <= 4) return launch_skinny<4>(g, s);
  return launch_skinny<8>(g, s);
}

int ln_rows_bf16(void* y, float* x, const float* g1, const float* b1, int rows, int D, float eps, int passes,
                 const float* partial, int nsplit, const float* pbias, int y_tiled, hipStream_t s) {
  ITTS_REQUIRE(D <= 2048 && passes >= 1 && passes <= 2, "ln_rows_bf16: D > 2048 or bad pass count");
  ITTS_REQUIRE(nsplit == 0 || (partial && nsplit <= 8), "ln_rows_bf16: partial sums missing or more than 8 splits");
  ITTS_REQUIRE(!y_tiled || D % 32 == 0, "ln_rows_bf16: tiled output needs D % 32 == 0");
  if (D <= 1280)
    hipLaunchKernelGGL(ln_rows_bf16_kernel<5>, dim3(rows), dim3(256), 0, s, (bf16_t*)y, x, g1, b1, D, eps, passes, partial,
                       nsplit, pbias, rows, y_tiled);
  else
    hipLaunchKernelGGL(ln_rows_bf16_kernel<8>, dim3(rows), dim3(256), 0, s, (bf16_t*)y, x, g1, b1, D, eps, passes, partial,
                       nsplit, pbias, rows, y_tiled);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
