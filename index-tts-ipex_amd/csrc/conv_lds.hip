// LDS-tiled 1-D convolution on the matrix cores for the narrow BigVGAN stages (AMPBlock1 convs with C = 24 / 48 / 96
// channels, k = 3 / 7 / 11, dilation 1 / 3 / 5; indextts/BigVGAN/models.py:20-81).
//
// The generic shift-GEMM (gemm_mfma.hip) re-fetches the activation rows once per tap; with K = k*C this small the
// kernel is bound by that global->LDS traffic, not by MFMA.  Here one workgroup owns a time tile of one batch item:
// the BM + (k-1)*dil input rows it needs (halo included, zero padded) are brought into LDS ONCE with contiguous
// 16-byte loads (channels-last rows are adjacent in memory), every tap then reads its MFMA A-fragments from that
// resident tile at a row offset.  Only the small weight slab streams (32-channel chunks, register-staged pairs).
// The epilogue goes back through LDS so residual / accumulate reads and the output store are 16-byte row-contiguous
// (C = 24 rows are 48 bytes: per-lane scalar stores would waste most of every HBM burst).
// v_mfma_f32_16x16x32_bf16; fp32 accumulation; bf16 in / out.
//
// LDS image (r04): one PLANE per 32-channel k-step, rows of 64 bytes (four 16-byte slots), slot g of row r at physical slot
// g ^ ((r >> 1) & 2).  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH
// LDS table): eight fragment rows at k-slot g beside the eight others at g + 1 - with any LINEAR row pitch two of those sixteen
// 16-byte reads always share a bank quartet (r03 counters: 6.9 M conflict cycles on 3.8 M active ones; the "odd multiple of
// 16 bytes" pitch was derived for contiguous 16-lane groups), with this key the sixteen land on sixteen different quartets for
// every row offset a tap adds.  The staged weight rows use the same 64-byte image.
//
// ACT (r04): Activation1d - the anti-aliased SnakeBeta that precedes EVERY one of these convolutions (a -> conv, models.py:65-74) -
// runs inside this kernel: the raw rows (tile + halo + 6 on either side for the FIR windows, replicate-clamped) are staged once,
// every lane slides down one channel (itts_snake_dev.h: the stand-alone kernel's loop, same operations) and writes the bf16
// result straight into the MFMA planes.  The stand-alone pass read and wrote the whole tensor once more per convolution: on these
// stages (HBM-bound convolutions, VALU-bound activation) that was 4 of the 10 tensor passes of an AMP iteration.
#include <cstdlib>

#include "itts_kernels.h"
#include "itts_snake_dev.h"

namespace itts {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WROW = 32;  // bf16 per staged weight row: 64 bytes = four 16-byte slots, slot q of row r at q ^ ((r >> 1) & 2)
constexpr int PROWE = 32;  // bf16 per activation-plane row (64 bytes)
__device__ __forceinline__ int swz4(int row) { return (row >> 1) & 2; }

// MT x NT 16x16 tiles per wave, WAVES_M x WAVES_N waves (= 4)
template <int MT, int NT, int WAVES_M, int WAVES_N, bool ACT>
__global__ __launch_bounds__(256) void conv_lds_kernel(GemmArgs g, int tiles_per_item, int HR, int CP) {
  constexpr int BM = WAVES_M * MT * 16, NP = WAVES_N * NT * 16;
  constexpr int WROWS = (NP + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: [planes][staged weights], or with ACT [planes][raw tile] with the staged weights on top of the raw tile once it is dead
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                              // [CP / 32 planes][HR][32] (CP = channels rounded up to 32)
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem + (size_t)HR * CP * 2);        // [2][NP][WROW]
  bf16_t* sX = sW;                                                           // ACT: raw rows [HR + 12][C], row j = x[clamp(tlo - 6 + j)]
  float* sC = reinterpret_cast<float*>(smem);                                // epilogue: [BM][NP + 4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int b = blockIdx.x / tiles_per_item, t0 = (blockIdx.x - b * tiles_per_item) * BM;
  const int T = g.T, C = g.Cin, K = g.taps * C;
  const bf16_t* __restrict__ A = (const bf16_t*)g.A + (size_t)b * T * g.lda;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W;

  // ---- resident input tile: rows t0 - pad_left + i, zero outside [0, T); pad columns zero ----
  if constexpr (!ACT) {
    // (row, 16-byte column) of vector v = tid + 256 * step, stepped without a division per vector
    const int vpr = CP >> 3, cv = C >> 3, nvec = HR * vpr;
    const int di = 256 / vpr, dq = 256 - di * vpr;
    int i = tid / vpr, q = tid - i * vpr;
    for (int v = tid; v < nvec; v += 256, i += di, q += dq) {
      if (q >= vpr) {
        q -= vpr;
        ++i;
      }
      const int ts = t0 - g.pad_left + i;
      u32x4 val = u32x4{0u, 0u, 0u, 0u};
      if (q < cv && ts >= 0 && ts < T) val = *reinterpret_cast<const u32x4*>(A + (size_t)ts * g.lda + q * 8);
      *reinterpret_cast<u32x4*>(sA + ((size_t)(q >> 2) * HR + i) * PROWE + (((q & 3) ^ swz4(i)) << 3)) = val;
    }
  } else {
    const int tlo = t0 - g.pad_left;  // time of plane row 0
    {  // raw rows, replicate-clamped (Activation1d pads its input and its up-sampled signal by replication, resample.py:24-33)
      const int cv = C >> 3, nvec = (HR + 12) * cv;
      const int di = 256 / cv, dq = 256 - di * cv;
      int i = tid / cv, q = tid - i * cv;
      for (int v = tid; v < nvec; v += 256, i += di, q += dq) {
        if (q >= cv) {
          q -= cv;
          ++i;
        }
        int ts = tlo - 6 + i;
        ts = ts < 0 ? 0 : (ts >= T ? T - 1 : ts);
        *reinterpret_cast<u32x4*>(sX + (size_t)i * C + q * 8) = *reinterpret_cast<const u32x4*>(A + (size_t)ts * g.lda + q * 8);
      }
    }
    // the channels behind C of the last plane multiply as zero whatever they hold (cok), but must be finite: clear them
    if ((C & 31) != 0) {
      const int padv = (CP - C) >> 3, cvv = C >> 3;  // 16-byte slots per row to clear
      for (int v = tid; v < HR * padv; v += 256) {
        const int i = v / padv, q = cvv + (v - i * padv);
        *reinterpret_cast<u32x4*>(sA + ((size_t)(q >> 2) * HR + i) * PROWE + (((q & 3) ^ swz4(i)) << 3)) = u32x4{0u, 0u, 0u, 0u};
      }
    }
    __syncthreads();
    const int runs = 256 / C, c = tid % C, run = tid / C;
    if (run < runs) {
      const int per = (HR + runs - 1) / runs;
      const int i0 = run * per, i1 = min(i0 + per, HR);  // plane rows [i0, i1) = times tlo + i
      const float ea = expf(g.pre_alpha[c]);
      const float inv_b = 1.f / (expf(g.pre_beta[c]) + 1e-9f);
      float fu[12], fd[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) fu[i] = fd[i] = g.pre_filt[i];
      const int pl = c >> 5, sl = (c & 31) >> 3, el = c & 7;
      auto put = [&](int i, float y) {
        sA[((size_t)pl * HR + i) * PROWE + ((sl ^ swz4(i)) << 3) + el] = (bf16_t)y;
      };
      // rows whose time lies outside [0, T) are the convolution's ZERO padding (of the activated signal)
      const int ta = max(tlo + i0, 0), tb = min(tlo + i1, T);
      for (int i = i0; i < min(i1, -tlo); ++i) put(i, 0.f);
      for (int i = max(i0, T - tlo); i < i1; ++i) put(i, 0.f);
      if (ta < tb) snake_run<bf16_t, true>(sX + c, C, tlo - 6, ta, tb, T, ea, inv_b, fu, fd, [&](int t, float y) { put(t - tlo, y); });
    }
    __syncthreads();  // the planes are complete and the raw tile is dead: the staged weights may take its place
  }
  const int cpt = (C + 31) >> 5, nchunk = g.taps * cpt, npair = (nchunk + 1) >> 1;
  const int lr = tid >> 2, lq = tid & 3;
  const bf16_t* w_row[WROWS];
#pragma unroll
  for (int p = 0; p < WROWS; ++p) w_row[p] = W + (size_t)min(lr + 64 * p, g.N - 1) * K;
  u32x4 rw[2][WROWS];
  unsigned wok = 0;
  int nx_ch = 0, nx_tap = 0, nx_c0 = 0;  // weight chunks are requested in order: running (tap, channel) counters, no division
  auto load_w = [&]() {
    wok = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool live = nx_ch < nchunk;
      const int tap = live ? nx_tap : 0, c0 = live ? nx_c0 : 0;
      const bool cok = live && lq * 8 < C - c0;
      const int cq = cok ? lq * 8 : 0;
#pragma unroll
      for (int p = 0; p < WROWS; ++p) {
        rw[c][p] = *reinterpret_cast<const u32x4*>(w_row[p] + (size_t)tap * C + c0 + cq);
        wok |= ((cok && lr + 64 * p < g.N) ? 1u : 0u) << (c * 8 + p);  // rows >= N and channels >= C multiply as zero
      }
      ++nx_ch;
      nx_c0 += 32;
      if (nx_c0 >= C) {
        nx_c0 = 0;
        ++nx_tap;
      }
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int p = 0; p < WROWS; ++p)
        if (lr + 64 * p < NP)
          *reinterpret_cast<u32x4*>(sW + ((size_t)c * NP + lr + 64 * p) * WROW + ((lq ^ swz4(lr + 64 * p)) << 3)) =
              ((wok >> (c * 8 + p)) & 1u) ? rw[c][p] : u32x4{0u, 0u, 0u, 0u};
  };
  f32x4v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  load_w();
  store_w();
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4, fk = fg * 8;
  const int a_row0 = wm * MT * 16 + fr;  // fragment row of tile 0 before the tap shift (tiles i: + 16 i, which leaves the swizzle key alone)
  const bf16_t* w_lane = sW + (size_t)(wn * NT * 16 + fr) * WROW + ((fg ^ swz4(fr)) << 3);  // (rows + multiples of 16: same key)
  int tap = 0, c0 = 0;  // chunk consumed by the MFMAs, same running-counter scheme
  for (int pr = 0; pr < npair; ++pr) {
    load_w();  // pair pr + 1 (past the end: clamped addresses, stored as zeros and never read)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (2 * pr + c < nchunk) {  // block-uniform
        const bool cok = fk < C - c0;  // lanes past the last channel of a partial chunk read zero
        const int arow = a_row0 + tap * g.dil;
        const bf16_t* ap = sA + ((size_t)(c0 >> 5) * HR + arow) * PROWE + ((fg ^ swz4(arow)) << 3);
        bf16x8 af[MT], wf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(ap + (size_t)i * 16 * PROWE);
          af[i] = cok ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(w_lane + ((size_t)c * NP + j * 16) * WROW);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = half_mfma16(af[i], wf[j], acc[i][j]);
        c0 += 32;
        if (c0 >= C) {
          c0 = 0;
          ++tap;
        }
      }
    }
    __syncthreads();  // every wave is done with this weight pair (and, on the last pass, with the input tile)
    if (pr + 1 < npair) {
      store_w();
      __syncthreads();
    }
  }
  // ---- epilogue through LDS: fp32 tile [BM][NP + 4], then row-contiguous 16-byte traffic ----
  constexpr int CS = NP + 4;
  const int cr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(size_t)(wm * MT * 16 + i * 16 + cr + r) * CS + wn * NT * 16 + j * 16 + cc] = acc[i][j][r];
  __syncthreads();
  const int nv = g.N >> 3;  // 8-column vectors per row
  bf16_t* __restrict__ Cc = (bf16_t*)g.C + (size_t)b * T * g.ldc;
  const bf16_t* __restrict__ R = g.R ? (const bf16_t*)g.R + (size_t)b * T * g.ldr : nullptr;
  const bf16_t* __restrict__ ADD = g.ADD ? (const bf16_t*)g.ADD + (size_t)b * T * g.ldadd : nullptr;
  const float* bias = g.bias ? g.bias + (size_t)b * g.bias_bstride : nullptr;
  const bool plain = g.act == ACT_NONE && g.act2 == ACT_NONE && !g.scale && !g.shift && (!bias || ((uintptr_t)bias & 15) == 0);
  const int dm = 256 / nv, dq2 = 256 - dm * nv;
  int m = tid / nv, q = tid - m * nv;
  for (int v = tid; v < BM * nv; v += 256, m += dm, q += dq2) {
    if (q >= nv) {
      q -= nv;
      ++m;
    }
    const int t = t0 + m;
    if (t >= T) continue;
    const int n = q * 8;
    float o[8];
    const float4 s0 = *reinterpret_cast<const float4*>(sC + (size_t)m * CS + n);
    const float4 s1 = *reinterpret_cast<const float4*>(sC + (size_t)m * CS + n + 4);
    o[0] = s0.x; o[1] = s0.y; o[2] = s0.z; o[3] = s0.w; o[4] = s1.x; o[5] = s1.y; o[6] = s1.z; o[7] = s1.w;
    u32x4 rv = u32x4{0u, 0u, 0u, 0u}, av = u32x4{0u, 0u, 0u, 0u};
    if (R) rv = *reinterpret_cast<const u32x4*>(R + (size_t)t * g.ldr + n);
    if (ADD) av = *reinterpret_cast<const u32x4*>(ADD + (size_t)t * g.ldadd + n);
    const bf16_t* rb = reinterpret_cast<const bf16_t*>(&rv);
    const bf16_t* ab = reinterpret_cast<const bf16_t*>(&av);
    u32x4 ov;
    bf16_t* ob = reinterpret_cast<bf16_t*>(&ov);
    if (plain) {
      // the AMP-block form (bias, residual, alpha, beta * running sum; no activation / BN affine): straight-line code -
      // on the narrow stages this epilogue is issue-bound
      float bv[8];
      if (bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + n), b1 = *reinterpret_cast<const float4*>(bias + n + 4);
        bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = o[e] + bv[e] + (float)rb[e];        // rv / av are zero vectors when R / ADD are absent
        x = fmaf(g.beta, (float)ab[e], x * g.alpha);
        ob[e] = (bf16_t)x;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = o[e];
        if (bias) x += bias[n + e];
        x = act_apply(g.act, x);
        x = x * (g.scale ? g.scale[n + e] : 1.f) + (g.shift ? g.shift[n + e] : 0.f);
        x = act_apply(g.act2, x);
        if (R) x += (float)rb[e];
        x *= g.alpha;
        if (ADD) x += g.beta * (float)ab[e];
        ob[e] = (bf16_t)x;
      }
    }
    *reinterpret_cast<u32x4*>(Cc + (size_t)t * g.ldc + n) = ov;
  }
}

// LDS bytes per activation row over all planes (bf16 elements): channels rounded up to whole 32-channel k-steps
inline int row_pitch(int C) { return (C + 31) / 32 * 32; }

template <int MT, int NT, int WAVES_M, int WAVES_N, bool ACT>
int launch(const GemmArgs& g, hipStream_t s) {
  constexpr int BM = WAVES_M * MT * 16, NP = WAVES_N * NT * 16;
  const int HR = BM + (g.taps - 1) * g.dil, CP = row_pitch(g.Cin);
  const size_t lds_w = (size_t)2 * NP * WROW * 2, lds_raw = ACT ? (size_t)(HR + 12) * g.Cin * 2 : 0;
  const size_t lds_main = (size_t)HR * CP * 2 + (lds_w > lds_raw ? lds_w : lds_raw);
  const size_t lds_epi = (size_t)BM * (NP + 4) * 4;
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  const int B = g.M / g.T, tiles = (g.T + BM - 1) / BM;
  static bool attr_done = false;
  if (!attr_done) {
    ITTS_HIP_CHECK(hipFuncSetAttribute((const void*)conv_lds_kernel<MT, NT, WAVES_M, WAVES_N, ACT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_lds_kernel<MT, NT, WAVES_M, WAVES_N, ACT>), dim3(B * tiles), dim3(256), lds, s, g, tiles, HR, CP);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace

bool conv_lds_supported(const GemmArgs& g, int ta, int tw, int tc) {
  if (ta != BF16 || tw != BF16 || tc != BF16) return false;
  if (g.nphase != 1 || g.in_up != 1 || g.pad_mode != PAD_ZERO || g.taps < 3) return false;
  if (g.Cin % 8 != 0 || g.Cin > 96 || g.Cin < 16 || g.N % 8 != 0 || g.N > 96) return false;
  if (g.T <= 0 || g.M % g.T != 0 || g.T < 256) return false;
  if (g.lda % 8 || g.ldc % 8 || (g.R && g.ldr % 8) || (g.ADD && g.ldadd % 8)) return false;
  if (((uintptr_t)g.A | (uintptr_t)g.W | (uintptr_t)g.C | (uintptr_t)g.R | (uintptr_t)g.ADD) & 15) return false;
  if ((g.taps - 1) * g.dil > 64) return false;
  return true;
}

// Activation1d inside the kernel: one lane per channel and run, at least two runs (Cin <= 96 of 256 threads)
bool conv_lds_act_supported(const GemmArgs& g, int ta, int tw, int tc) {
  const bool off = getenv("ITTS_NO_CONV_ACT") != nullptr;  // A/B switch (read per call): Activation1d as its own launch
  return !off && g.pre_alpha && g.pre_beta && g.pre_filt && conv_lds_supported(g, ta, tw, tc) && g.lda == g.Cin;
}

int conv_lds(const GemmArgs& g, hipStream_t s) {
  ITTS_REQUIRE(g.A && g.W && g.C, "conv_lds: null pointer");
  ITTS_REQUIRE(conv_lds_supported(g, BF16, BF16, BF16), "conv_lds: unsupported shape");
  if (g.pre_alpha) {
    ITTS_REQUIRE(conv_lds_act_supported(g, BF16, BF16, BF16), "conv_lds: fused activation unsupported for this shape");
    // Tile heights measured on 64 x 480 frames (r04): C = 48 gains from 128-row tiles (3 workgroups per CU instead of 2: the
    // activation phase of one overlaps the matrix phase of another; 248.9 -> 242.5 ms per vocoder pass), C = 24 loses with them
    // (257.5) and C = 96 is indifferent to 64-row tiles (244.1 vs 243.4)
    if (g.N <= 32) return launch<4, 2, 4, 1, true>(g, s);
    if (g.N <= 48) return launch<2, 3, 4, 1, true>(g, s);
    return launch<4, 3, 2, 2, true>(g, s);
  }
  if (g.N <= 32) return launch<4, 2, 4, 1, false>(g, s);
  if (g.N <= 48) return launch<4, 3, 4, 1, false>(g, s);
  return launch<4, 3, 2, 2, false>(g, s);
}

}  // namespace itts
