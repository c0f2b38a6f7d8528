// bf16 shift-GEMM, LDS-DMA staged (second generation of gemm_mfma.hip for the large, regular shapes: BigVGAN stage 1-3
// AMPBlock convs / conv_pre / the transposed-conv phases, BigVGAN/models.py:20-81,149-161; GPT prefill + latent-pass
// projections).  Same contract as gemm_mfma_kernel: C[m, n] = epilogue(sum_{tap, c} A[row(m, tap), c] * W[n, tap*Cin + c]).
//
// What changed, following cdna_hip_programming.md section 5 (the "step-3 structure" with two LDS buffers):
//   * 128 x BN tile (BN = 128 or 64), K consumed 64 channels per stage (Cin % 64 == 0), v_mfma_f32_16x16x32_bf16, 4 waves
//     as 2 x 2, 4 x (BN/32) accumulator tiles per wave;
//   * operands go global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass): a wave instruction fills
//     8 rows x 128 bytes; rows are linear 128-byte lines (what the DMA needs) and the bank swizzle lives on the SOURCE
//     side: LDS 16-byte slot p of row r holds logical k-chunk p ^ ((r >> 1) & 7), which makes every ds_read_b128 lane
//     group hit 16 distinct slots of the 256-byte bank row;
//   * two LDS stages (64 KiB, two workgroups per CU): the DMA of stage s + 1 is in flight while stage s is multiplied;
//   * rows outside [0, T) of a batch item (conv zero padding) are fetched from a zero page - the DMA has no per-lane
//     select - reflect padding is resolved in the source address;
//   * workgroup ids are remapped so that the column tiles of one row tile share an XCD (their A rows hit that L2).
#include <cstdlib>

#include "itts_kernels.h"

namespace itts {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero_page[16];  // 256 bytes of zeros: the source of out-of-range conv rows

__device__ __forceinline__ int reflect_i(int t, int T) {
  if (t < 0) t = -t;
  if (t >= T) t = 2 * (T - 1) - t;
  return t;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

constexpr int BM = 128, BK = 64, ROWB = BK * 2;  // 128-byte LDS rows

template <int BN, typename TC>
__global__ __launch_bounds__(256, 2) void gemm_glds_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  constexpr int NT = BN / 32;              // 16-wide accumulator tiles per wave along N (wave tile 64 x BN/2)
  constexpr int WI = BN / 32;              // W DMA instructions per wave and stage (BN rows / 4 waves / 8 rows)
  constexpr int STAGE = (BM + BN) * ROWB;  // bytes per LDS stage
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];  // ONE LDS object (see the guide: a second one de-pipelines)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- XCD-aware, bijective remap: consecutive logical ids (= the column tiles of one row tile) land on one XCD ----
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int per_phase = tiles_m * tiles_n;
  const int phase = wgid / per_phase, rem = wgid - phase * per_phase;
  const int tm = rem / tiles_n, tn = rem - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const bf16_t* __restrict__ A = (const bf16_t*)g.A;
  const int K = g.taps * g.Cin;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W + (size_t)phase * g.N * K;
  const int T = g.T > 0 ? g.T : g.M;
  const int cpt = g.Cin / BK;  // stages per tap
  const int nk = g.taps * cpt;
  const int shift0 = g.phase_shift[phase] - g.pad_left;

  // ---- loader: wave w fills rows [32w, 32w + 32) of A (4 instructions of 8 rows) and rows [BN/4 * w, ..) of W ----
  const int lrow = lane >> 3, pslot = lane & 7;  // row inside the 8-row instruction, physical 16-byte slot
  int a_t[4];
  long a_off[4];
  int a_sw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + lrow, m = m0 + row;
    a_sw[i] = (pslot ^ ((row >> 1) & 7)) * 8;  // logical k offset (elements) this lane fetches
    if (m < g.M) {
      const int b = m / T;
      a_t[i] = m - b * T;
      a_off[i] = (long)b * T;
    } else {
      a_t[i] = -(1 << 28);
      a_off[i] = 0;
    }
  }
  const bf16_t* w_src[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = wave * (BN / 4) + i * 8 + lrow;
    w_src[i] = W + (size_t)min(n0 + row, g.N - 1) * K + (pslot ^ ((row >> 1) & 7)) * 8;
  }
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page) + pslot * 8;
  int nx_tap = 0, nx_c0 = 0;  // the next stage to request: running (tap, channel) counters
  auto stage_load = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sw = sa + BM * ROWB;
    const int off = shift0 + nx_tap * g.dil;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ts = a_t[i] + off;
      if (g.pad_mode == PAD_REFLECT && a_t[i] >= 0) ts = reflect_i(ts, T);
      const bool ok = ts >= 0 && ts < T;
      const bf16_t* src = ok ? A + (a_off[i] + ts) * g.lda + nx_c0 + a_sw[i] : zp;
      glds16(src, sa + (wave * 32 + i * 8) * ROWB);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) glds16(w_src[i] + (size_t)nx_tap * g.Cin + nx_c0, sw + (wave * (BN / 4) + i * 8) * ROWB);
    nx_c0 += BK;
    if (nx_c0 >= g.Cin) {
      nx_c0 = 0;
      ++nx_tap;
    }
  };

  f32x4v acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  // fragment addresses: row r, logical chunk c = 4 * kstep + fq -> byte r * 128 + ((c ^ ((r >> 1) & 7)) << 4)
  int a_rd[4], a_sx[4], w_rd[NT], w_sx[NT];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wm * 64 + i * 16 + fr;
    a_rd[i] = r * ROWB;
    a_sx[i] = (r >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = wn * (BN / 2) + j * 16 + fr;
    w_rd[j] = BM * ROWB + r * ROWB;
    w_sx[j] = (r >> 1) & 7;
  }

  stage_load(0);
  __syncthreads();  // (drains the DMA: hipcc waits vmcnt(0) in front of the barrier while a glds is in flight)
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage_load(buf ^ 1);
    const unsigned char* sb = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_rd[i] + (((ks * 4 + fq) ^ a_sx[i]) << 4));
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sb + w_rd[j] + (((ks * 4 + fq) ^ w_sx[j]) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = half_mfma16(af[i], wf[j], acc[i][j]);
    }
    __syncthreads();  // stage kt + 1 has landed (vmcnt(0) in front of the barrier) and stage kt is free to be refilled
  }

  // ---- epilogue (as gemm_mfma_kernel): lane holds rows (lane >> 4) * 4 + r, column lane & 15 of each 16 x 16 tile ----
  TC* __restrict__ C = (TC*)g.C;
  const TC* __restrict__ R = (const TC*)g.R;
  const TC* __restrict__ ADD = (const TC*)g.ADD;
  const int cr = fq * 4, cc = fr;
  const bool plain = g.act == ACT_NONE && g.act2 == ACT_NONE && !g.scale && !g.shift;
  float sc[NT], sh[NT];
  int ncol[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * (BN / 2) + j * 16 + cc;
    ncol[j] = n;
    const int nc = min(n, g.N - 1);
    sc[j] = g.scale ? g.scale[nc] : 1.f;
    sh[j] = g.shift ? g.shift[nc] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 64 + i * 16 + cr + r;
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

template <int BN, typename TC>
int launch(const GemmArgs& g, hipStream_t s) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_glds_kernel<BN, TC>), dim3(tiles_m * tiles_n * g.nphase), dim3(256), 0, s, g, tiles_m, tiles_n);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace

// Shapes this kernel takes from gemm_mfma, from the A/B of tools/bench_gemm.py (profiles/r02_gemm_ab_*.txt): with two
// stages per workgroup the DMA pipeline is shallower than gemm_mfma's register ring, so it wins where the K loop is
// long (K = taps * Cin >= 4096: BigVGAN k = 7 / 11 convs at C >= 384, conv_pre; 1.1-1.3x) and on the 64-wide tile
// (N = 192, BigVGAN stage 3: 1.1-1.5x), and loses on short K (k = 3, the 1280-deep GPT projections) and on the
// polyphase transposed convs - those stay on the register-staged kernel.
bool gemm_glds_supported(const GemmArgs& g, int ta, int tw, int tc) {
  if (ta != BF16 || tw != BF16 || (tc != BF16 && tc != F32)) return false;
  if (g.Cin % 64 != 0 || g.lda % 8 != 0 || g.in_up != 1 || g.nphase < 1 || g.nphase > 8) return false;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return false;
  if (g.N < 64 || g.M < 256) return false;
  const int bn = (g.N % 128 != 0 && g.N % 64 == 0 && g.N < 256) ? 64 : 128;
  const long tiles = (long)((g.M + BM - 1) / BM) * ((g.N + bn - 1) / bn) * g.nphase;
  if (g.nphase != 1) return false;
  // enough tiles to fill the chip twice over on the K-deep or 64-wide shapes; the narrow convolutions (N <= 512: the weight panel
  // stays in L2) already pay from one tile per CU and K >= 1024 (measured: tools/bench_gemm.py --batch 1, profiles/r04_gemm_ab_b1.txt)
  if (tiles >= 384 && (bn == 64 || (long)g.taps * g.Cin >= 4096)) return true;
  return g.N <= 512 && tiles >= 256 && (long)g.taps * g.Cin >= 1024;
}

int gemm_glds(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s) {
  ITTS_REQUIRE(g.A && g.W && g.C, "gemm_glds: null pointer");
  ITTS_REQUIRE(gemm_glds_supported(g, ta, tw, tc), "gemm_glds: unsupported shape/dtype");
  const int T = g.T > 0 ? g.T : g.M;
  ITTS_REQUIRE(g.M % T == 0 && g.lda >= g.Cin && g.ldc >= g.N * g.nphase, "gemm_glds: bad dims");
  const bool bn64 = g.N % 128 != 0 && g.N % 64 == 0 && g.N < 256;
  if (tc == BF16) return bn64 ? launch<64, bf16_t>(g, s) : launch<128, bf16_t>(g, s);
  return bn64 ? launch<64, float>(g, s) : launch<128, float>(g, s);
}

}  // namespace itts
