// bf16 matrix-core shift-GEMM for gfx950:  C[m, n] = epilogue( sum_{tap, c} A[row(m, tap), c] * W[n, tap*Cin + c] )
// (nn.Linear / Conv1d / polyphase ConvTranspose1d / upsampled conv on channels-last activations - the dense
// contractions of the GPT prefill + latent pass, the conformer/perceiver and BigVGAN stages with C >= 96).
//
// v_mfma_f32_16x16x32_bf16.  Both operands are k-contiguous in memory (A rows are channels-last activations,
// W rows are [tap][Cin]), which is exactly the MFMA operand map of cdna_hip_programming.md section 3:
// lane l holds A[row l&15][k = 8(l>>4) .. +8] and W[n = l&15][k = 8(l>>4) .. +8] as one 16-byte fragment.
// Workgroup = 4 waves (2 x 2), tile BM x BN, K consumed in 32-channel chunks per tap (Cin % 8 == 0; the tail chunk of
// a tap is zero-filled on the A side, so Cin = 24 / 48 of the last BigVGAN stages also run here), two chunks per barrier.  Global -> register -> LDS staging with the next
// chunk pair requested before the MFMAs of the current one (register double buffering, one barrier per
// pair).  LDS rows are padded to 80 bytes: bank(20*r) is a conflict-free pattern for ds_read_b128.
// The conv "shift" is folded into the A-row index (zero / reflect padding, dilation, nearest-upsampled
// source rows), so no im2col buffer ever exists in HBM.
#include <cstdlib>

#include "itts_kernels.h"

namespace itts {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;        // channels per chunk
constexpr int LDSROW = 40;    // bf16 elements per LDS row (80 bytes)

__device__ __forceinline__ int reflect_idx(int t, int T) {
  if (t < 0) t = -t;
  if (t >= T) t = 2 * (T - 1) - t;
  return t;
}

// DEPTH = chunk pairs in flight per thread (register ring).  The K loop of one workgroup costs one memory latency per
// refill; with DEPTH pairs requested ahead the refill rate is DEPTH x higher, which is what the few-tile / long-K shapes
// (GPT latent pass: 1242 rows, K up to 5120; BigVGAN stage 1-3 convs) need - they run 1-2 workgroups per CU, so no
// other wave hides the latency.
template <int BM, int BN, typename TC, int NBUF, int DEPTH>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2;      // wave tile
  constexpr int MT = WM / 16, NT = WN / 16;    // 16x16 MFMA tiles per wave
  constexpr int AROWS = BM / 64;               // A rows per thread per chunk (256 threads: 64 rows x 4 x 16 B)
  constexpr int WROWS = (BN + 63) / 64;
  // NBUF = 2: one barrier per chunk pair; NBUF = 1: two barriers but half the LDS -> more workgroups per CU, whose
  // interleaving hides the global-load latency of this register-staged loop (cdna_hip_programming 'step-3 structure')
  __shared__ __attribute__((aligned(16))) bf16_t sA[NBUF][2][BM][LDSROW];  // [buffer][chunk][row][k]
  __shared__ __attribute__((aligned(16))) bf16_t sW[NBUF][2][BN][LDSROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN, phase = blockIdx.z;
  const bf16_t* __restrict__ A = (const bf16_t*)g.A;
  const int K = g.taps * g.Cin;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W + (size_t)phase * g.N * K;
  const int T = g.T > 0 ? g.T : g.M;
  const int Tin = T / g.in_up;
  const int cpt = (g.Cin + CK - 1) / CK;  // chunks per tap (the last one may be partial)
  const int nchunk = g.taps * cpt;
  const int npair = (nchunk + 1) / 2;

  // loader mapping: thread -> (row lr + 64*p, 16-byte column group lq)
  const int lr = tid >> 2, lq = tid & 3;
  int a_t[AROWS];
  long a_base[AROWS];
#pragma unroll
  for (int p = 0; p < AROWS; ++p) {
    const int m = m0 + lr + 64 * p;
    if (m < g.M) {
      const int b = m / T;
      a_t[p] = m - b * T;
      a_base[p] = (long)b * Tin;
    } else {
      a_t[p] = -(1 << 28);  // always out of range -> zero rows
      a_base[p] = 0;
    }
  }
  const bf16_t* w_row[WROWS];
#pragma unroll
  for (int p = 0; p < WROWS; ++p) w_row[p] = W + (size_t)min(n0 + lr + 64 * p, g.N - 1) * K;

  struct Stage {
    u32x4 a[2][AROWS], w[2][WROWS];
    unsigned ok;  // validity bits (A: bit c*8+p, W: bit 16+c), applied when the stage is written to LDS - a select
                  // next to the load would make the wave wait for the data right where it was requested
  };
  Stage st[DEPTH];
  // every load is unconditional (clamped, in-bounds addresses; dead lanes are zeroed by a select afterwards): vmcnt is
  // in-order and hipcc's wait accounting turns conservative across predicated loads
  // loads are requested in increasing chunk order, so (tap, channel offset) of the next chunk is a running counter -
  // no integer division in the K loop
  int nx_ch = 0, nx_tap = 0, nx_c0 = 0;
  const bool up1 = g.in_up == 1;
  auto load_pair = [&](Stage& r) {
    unsigned okbits = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool live = nx_ch < nchunk;
      const int tap = live ? nx_tap : 0, c0 = live ? nx_c0 : 0;
      const bool cok = lq * 8 < g.Cin - c0;   // this lane's 8 channels exist in the tap
      const int cq = cok ? lq * 8 : 0;        // clamped (in-bounds) column for masked lanes
      const int off = g.phase_shift[phase] + tap * g.dil - g.pad_left;
#pragma unroll
      for (int p = 0; p < AROWS; ++p) {
        int ts = a_t[p] + off;
        if (g.pad_mode == PAD_REFLECT && a_t[p] >= 0) ts = reflect_idx(ts, T);
        const bool ok = live && cok && ts >= 0 && ts < T;
        const int tsc = ok ? ts : 0;
        const long rr = a_base[p] + (up1 ? tsc : tsc / g.in_up);
        r.a[c][p] = *reinterpret_cast<const u32x4*>(A + rr * g.lda + c0 + cq);
        okbits |= (ok ? 1u : 0u) << (c * 8 + p);
      }
#pragma unroll
      for (int p = 0; p < WROWS; ++p) {
        // masked lanes multiply zeros from A: their W fragment only has to be finite and in bounds
        r.w[c][p] = *reinterpret_cast<const u32x4*>(w_row[p] + (size_t)tap * g.Cin + c0 + cq);
      }
      okbits |= ((live && cok) ? 1u : 0u) << (16 + c);
      ++nx_ch;
      nx_c0 += CK;
      if (nx_c0 >= g.Cin) {
        nx_c0 = 0;
        ++nx_tap;
      }
    }
    r.ok = okbits;
  };
  auto store_pair = [&](int buf, const Stage& r) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int p = 0; p < AROWS; ++p)
        *reinterpret_cast<u32x4*>(&sA[buf][c][lr + 64 * p][lq * 8]) = ((r.ok >> (c * 8 + p)) & 1u) ? r.a[c][p] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int p = 0; p < WROWS; ++p)
        if (lr + 64 * p < BN)
          *reinterpret_cast<u32x4*>(&sW[buf][c][lr + 64 * p][lq * 8]) = ((r.ok >> (16 + c)) & 1u) ? r.w[c][p] : u32x4{0u, 0u, 0u, 0u};
    }
  };

  f32x4v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load_pair(st[d]);  // pairs past the end load clamped addresses and are never stored
  store_pair(0, st[0]);
  __syncthreads();
  const int fr = lane & 15, fk = (lane >> 4) * 8;
  // the trip count is rounded up to a multiple of DEPTH: pairs past the end were loaded from clamped addresses with all
  // validity bits clear, so they reach LDS as zeros and add nothing - and the loop body stays branch-free, which keeps
  // the ring slots in fixed registers
  const int npair_pad = (npair + DEPTH - 1) / DEPTH * DEPTH;
  for (int p0 = 0; p0 < npair_pad; p0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int pr = p0 + d;
      const int buf = NBUF == 2 ? (pr & 1) : 0;
      load_pair(st[d]);  // slot d went to LDS one step ago; refill it with the pair DEPTH ahead
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bf16x8 af[MT], bfr[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&sA[buf][c][wm * WM + i * 16 + fr][fk]);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(&sW[buf][c][wn * WN + j * 16 + fr][fk]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = half_mfma16(af[i], bfr[j], acc[i][j]);
      }
      if (NBUF == 1) __syncthreads();  // every wave has consumed the buffer before it is overwritten
      store_pair(NBUF == 2 ? (buf ^ 1) : 0, st[(d + 1) % DEPTH]);
      __syncthreads();
    }
  }

  // epilogue: lane holds rows (lane>>4)*4 + r, column lane&15 of each 16x16 tile
  TC* __restrict__ C = (TC*)g.C;
  const TC* __restrict__ R = (const TC*)g.R;
  const TC* __restrict__ ADD = (const TC*)g.ADD;
  const int cr = (lane >> 4) * 4, cc = lane & 15;
  // per-column operands once per column tile, per-row operands (batch item of the row -> bias row) once per row: the
  // m / T division used to run per element
  const bool plain = g.act == ACT_NONE && g.act2 == ACT_NONE && !g.scale && !g.shift;
  float sc[NT], sh[NT];
  int ncol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * WN + j * 16 + cc;
    ncol[j] = n;
    const int nc = min(n, g.N - 1);
    sc[j] = g.scale ? g.scale[nc] : 1.f;
    sh[j] = g.shift ? g.shift[nc] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * WM + i * 16 + cr + r;
      if (m >= g.M) continue;
      const float* brow = g.bias ? g.bias + (g.bias_bstride ? (size_t)(m / T) * g.bias_bstride : 0) : nullptr;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = ncol[j];
        if (n >= g.N) continue;
        const int col = phase * g.N + n;
        float v = acc[i][j][r];
        if (brow) v += brow[n];
        if (!plain) {
          v = act_apply_fast(g.act, v);
          v = v * sc[j] + sh[j];
          v = act_apply(g.act2, v);
        }
        if (R) v += ldf(R + (size_t)m * g.ldr + col);
        v *= g.alpha;
        if (ADD) v += g.beta * ldf(ADD + (size_t)m * g.ldadd + col);
        stf(C + (size_t)m * g.ldc + col, v);
      }
    }
  }
}

static int g_nbuf = 1;  // 1 = single LDS buffer (more workgroups per CU; measured faster end to end), ITTS_GEMM_NBUF=2 to A/B

static int g_depth = 0;  // 0 = per-tile default; ITTS_GEMM_DEPTH=1/2/4 to A/B

template <int BM, int BN, typename TC, int DEPTH>
int launch_d(const GemmArgs& g, hipStream_t s) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nphase);
  if (g_nbuf == 1) hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, TC, 1, DEPTH>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, TC, 2, DEPTH>), grid, dim3(256), 0, s, g);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

template <int BM, int BN, typename TC, int DDEF>
int launch(const GemmArgs& g, hipStream_t s) {
  const int d = g_depth ? g_depth : DDEF;
  if (d <= 1) return launch_d<BM, BN, TC, 1>(g, s);
  if (d == 2) return launch_d<BM, BN, TC, 2>(g, s);
  return launch_d<BM, BN, TC, 4>(g, s);
}

template <typename TC>
int dispatch(const GemmArgs& g, hipStream_t s) {
  // tile choice: the K loop of one workgroup is latency-bound (register-staged), so what matters first is having several
  // workgroups per CU (256 CUs); among shapes that do, prefer the larger tile (fewer LDS bytes per MFMA).  Small tiles
  // carry a deeper register ring (their stages are cheap: 16 VGPRs per pair at 64x64).
  auto tiles = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.nphase; };
  // ring depth: deep where a CU holds 1-2 workgroups (nothing else hides the refill latency), 1 where there are many
  // tiles per CU (the extra registers would cost more occupancy than the ring buys)
  static const int force = [] {  // ITTS_GEMM_TILE=1 (128x128) / 2 (128x64) / 3 (64x64): experiments only
    const char* e = getenv("ITTS_GEMM_TILE");
    return e ? atoi(e) : 0;
  }();
  if (force == 1 && g.N >= 64) return launch<128, 128, TC, 2>(g, s);
  if (force == 2 && g.N >= 64) return launch<128, 64, TC, 2>(g, s);
  if (force == 3 && g.N >= 64) return launch<64, 64, TC, 4>(g, s);
  if (g.N <= 32) return launch<128, 32, TC, 1>(g, s);
  // N = 192 (BigVGAN stage 3): two 128-wide tiles would waste a quarter of the MFMA work, three 64-wide tiles none
  const bool ragged128 = g.N % 128 != 0 && g.N % 64 == 0 && g.N < 256;
  if (g.N >= 128 && tiles(128, 128) >= 768 && !ragged128)
    return tiles(128, 128) >= 1536 ? launch<128, 128, TC, 1>(g, s) : launch<128, 128, TC, 2>(g, s);
  if (tiles(128, 64) >= 512 || g.M <= 64) {
    if (g.M <= 64) return launch<64, 64, TC, 4>(g, s);
    return tiles(128, 64) >= 768 ? launch<128, 64, TC, 1>(g, s) : launch<128, 64, TC, 2>(g, s);
  }
  return launch<64, 64, TC, 4>(g, s);
}

}  // namespace

bool gemm_mfma_supported(const GemmArgs& g, int ta, int tw, int tc) {
  if (ta != BF16 || tw != BF16 || (tc != BF16 && tc != F32)) return false;
  if (g.Cin % 8 != 0 || g.lda % 8 != 0) return false;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return false;
  if (g.M < 16) return false;  // tiny M (speaker-conditioning 1x1 convs): the vector kernel is fine
  return true;
}

int gemm_mfma(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s) {
  static const bool once = [] {
    const char* e = getenv("ITTS_GEMM_NBUF");
    if (e) g_nbuf = atoi(e) == 2 ? 2 : 1;
    const char* dd = getenv("ITTS_GEMM_DEPTH");
    if (dd) g_depth = atoi(dd);
    return true;
  }();
  (void)once;
  ITTS_REQUIRE(g.A && g.W && g.C, "gemm_mfma: null pointer");
  ITTS_REQUIRE(gemm_mfma_supported(g, ta, tw, tc), "gemm_mfma: unsupported shape/dtype");
  ITTS_REQUIRE(g.nphase >= 1 && g.nphase <= 8 && g.in_up >= 1, "gemm_mfma: bad phase/upsample");
  ITTS_REQUIRE(g.lda >= g.Cin && g.ldc >= g.N * g.nphase, "gemm_mfma: bad leading dims");
  const int T = g.T > 0 ? g.T : g.M;
  ITTS_REQUIRE(g.M % T == 0 && T % g.in_up == 0, "gemm_mfma: M must be a multiple of T");
  return tc == BF16 ? dispatch<bf16_t>(g, s) : dispatch<float>(g, s);
}

}  // namespace itts
