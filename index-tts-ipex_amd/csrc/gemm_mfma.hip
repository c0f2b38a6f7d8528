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

template <int BM, int BN, typename TC, int NBUF>
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

  u32x4 ra[2][AROWS], rw[2][WROWS];
  auto load_pair = [&](int pr) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int ch = 2 * pr + c;
      const bool live = ch < nchunk;
      const int chc = live ? ch : 0;
      const int tap = chc / cpt, c0 = (chc - tap * cpt) * CK;
      const bool cok = lq * 8 < g.Cin - c0;   // this lane's 8 channels exist in the tap
      const int cq = cok ? lq * 8 : 0;        // clamped (in-bounds) column for masked lanes
      const int off = g.phase_shift[phase] + tap * g.dil - g.pad_left;
#pragma unroll
      for (int p = 0; p < AROWS; ++p) {
        int ts = a_t[p] + off;
        if (g.pad_mode == PAD_REFLECT && a_t[p] >= 0) ts = reflect_idx(ts, T);
        const bool ok = live && cok && ts >= 0 && ts < T;
        const long r = a_base[p] + (ok ? ts / g.in_up : 0);
        const u32x4 v = *reinterpret_cast<const u32x4*>(A + r * g.lda + c0 + cq);
        ra[c][p] = ok ? v : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int p = 0; p < WROWS; ++p) {
        // masked lanes multiply zeros from A: their W fragment only has to be finite and in bounds
        const u32x4 v = *reinterpret_cast<const u32x4*>(w_row[p] + (size_t)tap * g.Cin + c0 + cq);
        rw[c][p] = (live && cok) ? v : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto store_pair = [&](int buf) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int p = 0; p < AROWS; ++p) *reinterpret_cast<u32x4*>(&sA[buf][c][lr + 64 * p][lq * 8]) = ra[c][p];
#pragma unroll
      for (int p = 0; p < WROWS; ++p)
        if (lr + 64 * p < BN) *reinterpret_cast<u32x4*>(&sW[buf][c][lr + 64 * p][lq * 8]) = rw[c][p];
    }
  };

  f32x4v acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  load_pair(0);
  store_pair(0);
  __syncthreads();
  const int fr = lane & 15, fk = (lane >> 4) * 8;
  for (int pr = 0; pr < npair; ++pr) {
    const int buf = NBUF == 2 ? (pr & 1) : 0;
    if (pr + 1 < npair) load_pair(pr + 1);
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
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (NBUF == 1) __syncthreads();  // every wave has consumed the buffer before it is overwritten
    if (pr + 1 < npair) store_pair(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  // epilogue: lane holds rows (lane>>4)*4 + r, column lane&15 of each 16x16 tile
  TC* __restrict__ C = (TC*)g.C;
  const TC* __restrict__ R = (const TC*)g.R;
  const TC* __restrict__ ADD = (const TC*)g.ADD;
  const int cr = (lane >> 4) * 4, cc = lane & 15;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * WN + j * 16 + cc;
    if (n >= g.N) continue;
    const int col = phase * g.N + n;
    const float sc = g.scale ? g.scale[n] : 1.f;
    const float sh = g.shift ? g.shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * WM + i * 16 + cr + r;
        if (m >= g.M) continue;
        float v = acc[i][j][r];
        if (g.bias) v += g.bias[(size_t)(m / T) * g.bias_bstride + n];
        v = act_apply(g.act, v);
        v = v * sc + sh;
        v = act_apply(g.act2, v);
        if (R) v += ldf(R + (size_t)m * g.ldr + col);
        v *= g.alpha;
        if (ADD) v += g.beta * ldf(ADD + (size_t)m * g.ldadd + col);
        stf(C + (size_t)m * g.ldc + col, v);
      }
    }
  }
}

static int g_nbuf = 1;  // 1 = single LDS buffer (more workgroups per CU; measured faster end to end), ITTS_GEMM_NBUF=2 to A/B

template <int BM, int BN, typename TC>
int launch(const GemmArgs& g, hipStream_t s) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, g.nphase);
  if (g_nbuf == 1) hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, TC, 1>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_mfma_kernel<BM, BN, TC, 2>), grid, dim3(256), 0, s, g);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

template <typename TC>
int dispatch(const GemmArgs& g, hipStream_t s) {
  // tile choice: the K loop of one workgroup is latency-bound (register-staged, one chunk pair ahead), so what matters
  // first is having several workgroups per CU (256 CUs); among shapes that do, prefer the larger tile (fewer LDS
  // bytes per MFMA).
  auto tiles = [&](int bm, int bn) { return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.nphase; };
  if (g.N <= 32) return launch<128, 32, TC>(g, s);
  if (g.N >= 128 && tiles(128, 128) >= 768) return launch<128, 128, TC>(g, s);
  if (tiles(128, 64) >= 512 || g.M <= 64) return g.M <= 64 ? launch<64, 64, TC>(g, s) : launch<128, 64, TC>(g, s);
  return launch<64, 64, TC>(g, s);
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
