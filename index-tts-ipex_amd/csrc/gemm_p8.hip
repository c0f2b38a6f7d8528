// bf16 shift-GEMM, 256 x 256 tile, eight phases per two K-tiles (third generation of gemm_mfma.hip / gemm_glds.hip for the
// LARGE regular shapes: BigVGAN AMPBlock convs at C >= 384, conv_pre, the GPT prefill / latent-pass projections at batch;
// BigVGAN/models.py:20-81,149-161, gpt/model.py:521-589).  Same contract as gemm_mfma_kernel:
//     C[m, n] = epilogue(sum_{tap, c} A[row(m, tap), c] * W[n, tap * Cin + c]).
//
// Structure (cdna_hip_programming.md section 5, "the 256^2 8-phase template", rebuilt for this contract):
//   * one 512-thread workgroup per CU: 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave = 8 x 4 accumulator tiles of
//     v_mfma_f32_16x16x32_bf16 (128 accumulator VGPRs);
//   * K is consumed 64 channels per K-tile; a K-tile is FOUR phases, one per quadrant (64 rows x 32 columns of the wave's
//     output: 16 MFMAs), in the order (A0,B0) (A0,B1) (A1,B1) (A1,B0): a phase re-reads only the operand half that changed
//     (12 / 4 / 8 / 4 ds_read_b128);
//   * operands go global -> LDS by global_load_lds_dwordx4 (no VGPR staging) in 16 KB CHUNKS that follow the quadrants - chunk
//     A-q0 holds rows 0..63 of BOTH wave rows (tile rows 0..63 and 128..191), B-q0 columns 0..31 of all four wave columns -
//     so the first phase of a K-tile needs two chunks, the next three one each; rows are linear 128-byte lines, the bank
//     swizzle lives on the SOURCE address (slot p of row r holds k-chunk p ^ ((r >> 1) & 7)), as in gemm_glds.hip;
//   * two LDS buffers of four chunks (128 KiB); every phase stages ONE chunk, four to five phases ahead of its first read
//     (tile t stages B0(t+1), B1(t+1), A1(t+1), A0(t+2)), and waits with a COUNTED s_waitcnt vmcnt(6): three chunks stay in
//     flight across every barrier - the loop never drains the DMA queue;
//   * two raw s_barriers per phase: [ds_read + stage + waits] barrier [16 MFMAs] barrier, and the second wave row runs half a
//     phase behind the first (one extra barrier up front), so on every SIMD one wave multiplies while the other loads.
// Hazards, by construction (g = 4 * tile + phase; a chunk staged in phase q is first read in phase q + 4 or q + 5):
//   RAW  the wait that retires a chunk (vmcnt(6) at the end of the load segment of phase g - 1, by every wave, both wave rows)
//        is followed by a barrier every wave passes before any wave's load segment of phase g;
//   WAR  ds_reads are retired (lgkmcnt(0)) BEFORE the barrier that ends their load segment, and a chunk slot is re-staged no
//        earlier than one phase after its last read (B0: read again in phase 3 of its tile, re-staged in phase 0 of the next).
// Conv addressing, zero page for padded rows and XCD-aware tile order as gemm_glds.hip; the polyphase transposed convs (nphase > 1)
// run as extra column tiles with their own tap shift and weight slab.
// Two geometries, same per-wave work (128 x 64 outputs): 2 x 4 waves = 256 x 256 tile (N a multiple of 256) and 4 x 2 waves =
// 512 x 128 tile (N a multiple of 128 only: 384-wide convs would leave a quarter of a 256-wide column tile empty); the second one
// stages 80 KB per K-tile (4 + 4 + 1 + 1 DMA instructions per thread) and fills the 160 KB of LDS exactly.
#include <cstdlib>

#include "itts_kernels.h"

namespace itts {
namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ uint4 g_zero_page8[16];  // 256 bytes of zeros: the source of out-of-range conv rows

__device__ __forceinline__ int reflect_i8(int t, int T) {
  if (t < 0) t = -t;
  if (t >= T) t = 2 * (T - 1) - t;
  return t;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

constexpr int PBK = 64, PROW = PBK * 2;  // 128-byte LDS rows

#define P8_STR_(x) #x
#define P8_STR(x) P8_STR_(x)

// The epilogue of 8 consecutive outputs (row m, columns ncol .. ncol + 7 of phase column block pcol), shared by the kernel and by
// the split-K reduction so that both run the same operations in the same order.
template <typename TC>
__device__ __forceinline__ void p8_finish8(const GemmArgs& g, int T, int m, int ncol, int pcol, bool vec_ok, bool plain, float (&v)[8]) {
  TC* __restrict__ C = (TC*)g.C;
  const TC* __restrict__ R = (const TC*)g.R;
  const TC* __restrict__ ADD = (const TC*)g.ADD;
  const float* brow = g.bias ? g.bias + (g.bias_bstride ? (size_t)(m / T) * g.bias_bstride : 0) : nullptr;
  if (brow) {
    if (vec_ok && (((uintptr_t)(brow + ncol)) & 15) == 0) {
      const float4 b0 = *reinterpret_cast<const float4*>(brow + ncol), b1 = *reinterpret_cast<const float4*>(brow + ncol + 4);
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += brow[min(ncol + e, g.N - 1)];
    }
  }
  if (!plain) {
    // (the activation switch outside the element loop; NewGELU - c_fc of the GPT blocks, the hot case - as x * sigmoid(2 u):
    //  one v_exp and one v_rcp instead of tanhf's ~25 instructions, the same function)
    if (g.act == ACT_GELU_NEW) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = v[e], u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        v[e] = x / (1.f + __expf(-2.f * u));
      }
    } else if (g.act != ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply(g.act, v[e]);
    }
    if (g.scale || g.shift) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = min(ncol + e, g.N - 1);
        v[e] = v[e] * (g.scale ? g.scale[n] : 1.f) + (g.shift ? g.shift[n] : 0.f);
      }
    }
    if (g.act2 != ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply(g.act2, v[e]);
    }
  }
  if (vec_ok && sizeof(TC) == 2) {
    if (R) {
      const bf16x8 rv = *reinterpret_cast<const bf16x8*>(R + (size_t)m * g.ldr + pcol + ncol);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += half_bits((unsigned short)rv[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
    if (ADD) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(ADD + (size_t)m * g.ldadd + pcol + ncol);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += g.beta * half_bits((unsigned short)av[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bf16_t t = (bf16_t)v[e];
      o[e] = __builtin_bit_cast(short, t);
    }
    *reinterpret_cast<bf16x8*>(C + (size_t)m * g.ldc + pcol + ncol) = o;
  } else if (vec_ok && sizeof(TC) == 4) {  // fp32 output (the GPT residual stream): two 16-byte accesses per operand
    const float* Rf = reinterpret_cast<const float*>(R);
    const float* Af = reinterpret_cast<const float*>(ADD);
    float* Cf = reinterpret_cast<float*>(C);
    if (Rf) {
      const float4 r0 = *reinterpret_cast<const float4*>(Rf + (size_t)m * g.ldr + pcol + ncol), r1 = *reinterpret_cast<const float4*>(Rf + (size_t)m * g.ldr + pcol + ncol + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= g.alpha;
    if (Af) {
      const float4 a0 = *reinterpret_cast<const float4*>(Af + (size_t)m * g.ldadd + pcol + ncol), a1 = *reinterpret_cast<const float4*>(Af + (size_t)m * g.ldadd + pcol + ncol + 4);
      v[0] += g.beta * a0.x; v[1] += g.beta * a0.y; v[2] += g.beta * a0.z; v[3] += g.beta * a0.w;
      v[4] += g.beta * a1.x; v[5] += g.beta * a1.y; v[6] += g.beta * a1.z; v[7] += g.beta * a1.w;
    }
    *reinterpret_cast<float4*>(Cf + (size_t)m * g.ldc + pcol + ncol) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(Cf + (size_t)m * g.ldc + pcol + ncol + 4) = float4{v[4], v[5], v[6], v[7]};
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int n = ncol + e;
      if (n < g.N) {
        float x = v[e];
        if (R) x += ldf(R + (size_t)m * g.ldr + pcol + n);
        x *= g.alpha;
        if (ADD) x += g.beta * ldf(ADD + (size_t)m * g.ldadd + pcol + n);
        stf(C + (size_t)m * g.ldc + pcol + n, x);
      }
    }
  }
}

template <int WM, int WN, typename TC>
__global__ __launch_bounds__(512) void gemm_p8_kernel(GemmArgs g, int tiles_m, int tiles_n) {
  static_assert(WM * WN == 8 && (WM == 2 || WM == 4), "8 waves as 2 x 4 or 4 x 2");
  constexpr int PBM = WM * 128, PBN = WN * 64;
  constexpr int PBUF = (PBM + PBN) * PROW;  // one K-tile: A rows 0 .. PBM - 1, then W rows 0 .. PBN - 1
  constexpr int BOFF = PBM * PROW;          // W rows behind the A rows
  constexpr int NA = WM, NB_ = WN / 2;      // DMA instructions per thread for an A / a B chunk (a chunk = the rows of one quadrant half)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * PBUF];  // ONE LDS object
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int grp = wave >> 2;  // waves w and w + 4 share a SIMD: the second group runs half a phase behind the first
  // ---- XCD-aware, bijective remap: consecutive logical ids (= the column tiles of one row tile) land on one XCD ----
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int per_phase = tiles_m * tiles_n;
  // K split (g.ksplit > 1, nphase == 1): the splits of one tile are consecutive logical ids (same XCD: they read different K ranges,
  // but the tile's A and W rows interleave in the same L2 sets either way) - split = id % ksplit
  const int split = g.ksplit > 1 ? wgid % g.ksplit : 0;
  const int tid_ = g.ksplit > 1 ? wgid / g.ksplit : wgid;
  const int phase = tid_ / per_phase, rem = tid_ - phase * per_phase;
  const int tm = rem / tiles_n, tn = rem - tm * tiles_n;
  const int m0 = tm * PBM, n0 = tn * PBN;
  const bf16_t* __restrict__ A = (const bf16_t*)g.A;
  const int K = g.taps * g.Cin;
  const bf16_t* __restrict__ W = (const bf16_t*)g.W + (size_t)phase * g.N * K;
  const int T = g.T > 0 ? g.T : g.M;
  const int nk_all = g.taps * (g.Cin / PBK);
  const int kper = (nk_all + g.ksplit - 1) / g.ksplit;
  const int kt_lo = split * kper;                    // this workgroup's K-tiles [kt_lo, nk)
  const int nk = min(nk_all, kt_lo + kper);          // (host: every split has at least one)
  const int shift0 = g.phase_shift[phase] - g.pad_left;

  // ---- loader: an A chunk is WM * 64 rows (rows h * 64 .. + 63 of every wave row), a B chunk WN * 32 rows; instruction i of wave w
  //      fills chunk rows [64 i + 8 w, + 8) ----
  const int lrow = lane >> 3, pslot = lane & 7;
  // (chunk row ci = 64 i + 8 w + lrow maps to tile row (ci / 64) * 128 + h * 64 + ci % 64 (A) or (ci / 32) * 64 + h * 32 + ci % 32 (B):
  //  multiples of 8 are added to lrow, so bits 1..3 of the tile row are bits 1..3 of 8 w + lrow: the swizzle key is ((8 w + lrow) >> 1) & 7)
  const int skey = ((wave * 8 + lrow) >> 1) & 7;
  const int l_sw2 = (pslot ^ skey) * 8;  // logical k offset (elements) this lane fetches
  int a_t[2][NA], a_b[2][NA];  // [A half][instruction]: time index inside the batch item (or far negative), first row of the item
  const bf16_t* w_src[2][NB_];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int ci = i * 64 + wave * 8 + lrow;
      const int ar = (ci >> 6) * 128 + h * 64 + (ci & 63);  // A-q(h): tile row
      const int m = m0 + ar;
      if (m < g.M) {
        const int b = m / T;
        a_t[h][i] = m - b * T;
        a_b[h][i] = b * T;
      } else {
        a_t[h][i] = -(1 << 28);
        a_b[h][i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      const int ci = i * 64 + wave * 8 + lrow;
      const int wrow = (ci >> 5) * 64 + h * 32 + (ci & 31);  // B-q(h): tile column
      w_src[h][i] = W + (size_t)min(n0 + wrow, g.N - 1) * K + l_sw2;
    }
  }
  const bf16_t* zp = reinterpret_cast<const bf16_t*>(g_zero_page8) + pslot * 8;
  // Source pointer of every (half, instruction), carried from K-tile to K-tile: inside a tap the next K-tile is the next 64 channels
  // of the same rows (+ 128 bytes, or + 0 on the zero page), so the row arithmetic (tap shift, reflect, range test, 64-bit row
  // offset: ~25 VALU instructions per DMA) runs once per tap instead of once per K-tile - the load segment of a phase has to stay
  // shorter than the other wave group's 16 MFMAs (~260 cycles).
  const bf16_t* a_ptr[2][NA];
  int a_inc[2][NA];
  const int cpt = g.Cin / PBK;
  auto stage_A = [&](int buf, int h, int kt) {  // A-q(h) of K-tile kt (clamped past the end: a harmless re-read)
    const int ktc = min(kt, nk - 1);
    const int tap = ktc / cpt, cc = ktc - tap * cpt;
    unsigned char* base = smem + buf * PBUF;
    if (cc == 0 || kt >= nk || kt == kt_lo) {  // wave-uniform: first channel chunk of a tap, the split's first K-tile (or an overshoot stage): resolve the rows
      const int off = shift0 + tap * g.dil;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int ts = a_t[h][i] + off;
        if (g.pad_mode == PAD_REFLECT && a_t[h][i] >= 0) ts = reflect_i8(ts, T);
        const bool ok = ts >= 0 && ts < T;
        a_ptr[h][i] = ok ? A + (size_t)(a_b[h][i] + ts) * g.lda + cc * PBK + l_sw2 : zp;
        a_inc[h][i] = ok ? PBK : 0;
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int ci = i * 64 + wave * 8;
      glds16(a_ptr[h][i], base + ((ci >> 6) * 128 + h * 64 + (ci & 63)) * PROW);
      a_ptr[h][i] += a_inc[h][i];
    }
  };
  auto stage_B = [&](int buf, int h, int kt) {
    const int ktc = min(kt, nk - 1);
    unsigned char* base = smem + buf * PBUF + BOFF;
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      const int ci = i * 64 + wave * 8;
      glds16(w_src[h][i] + (size_t)ktc * PBK, base + ((ci >> 5) * 64 + h * 32 + (ci & 31)) * PROW);
    }
  };

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int sx = (fr >> 1) & 7;  // swizzle key of this lane's fragment rows (rows fr + multiples of 16)
  // byte offsets inside a buffer: A row wr * 128 + mi * 16 + fr, W row wc * 64 + nj * 16 + fr, k-step ks -> slot (4 ks + fq) ^ sx
  const int a_rd = (wr * 128 + fr) * PROW, b_rd = BOFF + (wc * 64 + fr) * PROW;
  const int ko0 = ((0 + fq) ^ sx) << 4, ko1 = ((4 + fq) ^ sx) << 4;
  bf16x8 af[4][2], bfr[2][2];
  auto read_A = [&](int buf, int h) {
    const unsigned char* p = smem + buf * PBUF + a_rd + h * 64 * PROW;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      af[mi][0] = *reinterpret_cast<const bf16x8*>(p + mi * 16 * PROW + ko0);
      af[mi][1] = *reinterpret_cast<const bf16x8*>(p + mi * 16 * PROW + ko1);
    }
  };
  auto read_B = [&](int buf, int h) {
    const unsigned char* p = smem + buf * PBUF + b_rd + h * 32 * PROW;
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      bfr[nj][0] = *reinterpret_cast<const bf16x8*>(p + nj * 16 * PROW + ko0);
      bfr[nj][1] = *reinterpret_cast<const bf16x8*>(p + nj * 16 * PROW + ko1);
    }
  };
  // vmcnt counts: what may stay in flight behind the chunk the NEXT phase reads (issue order per tile: B0' B1' A1' A0'')
  constexpr int VM03 = 2 * NA + NB_, VM1 = NA + 2 * NB_;  // end of phases 0 / 2 / 3, end of phase 1
#define P8_LOAD_END(N)                                                   \
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");              \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
  __builtin_amdgcn_sched_barrier(0);                   \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_sched_barrier(0);
#define P8_COMPUTE(AH, BH)                                                                                             \
  __builtin_amdgcn_s_setprio(1);                                                                                       \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int nj = 0; nj < 2; ++nj) {                   \
    acc[(AH) * 4 + mi][(BH) * 2 + nj] =                                                                                 \
        half_mfma16(af[mi][0], bfr[nj][0], acc[(AH) * 4 + mi][(BH) * 2 + nj]);     \
    acc[(AH) * 4 + mi][(BH) * 2 + nj] =                                                                                 \
        half_mfma16(af[mi][1], bfr[nj][1], acc[(AH) * 4 + mi][(BH) * 2 + nj]);     \
  }                                                                                                                    \
  __builtin_amdgcn_s_setprio(0);                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                        \
  __builtin_amdgcn_sched_barrier(0);

  // ---- prologue: what phases -5 .. -1 would have staged ----
  stage_A(0, 0, kt_lo);
  stage_B(0, 0, kt_lo);
  stage_B(0, 1, kt_lo);
  stage_A(0, 1, kt_lo);
  stage_A(1, 0, kt_lo + 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM03) : "memory");  // A0(0), B0(0) of this wave have landed
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs half a phase behind the first
  __builtin_amdgcn_sched_barrier(0);

  for (int kt = kt_lo; kt < nk; ++kt) {
    const int buf = (kt - kt_lo) & 1;
    // phase 0: quadrant (A0, B0); stage B0(t + 1)
    read_A(buf, 0);
    read_B(buf, 0);
    stage_B(buf ^ 1, 0, kt + 1);
    P8_LOAD_END(VM03)
    P8_COMPUTE(0, 0)
    // phase 1: (A0, B1); stage B1(t + 1)
    read_B(buf, 1);
    stage_B(buf ^ 1, 1, kt + 1);
    P8_LOAD_END(VM1)
    P8_COMPUTE(0, 1)
    // phase 2: (A1, B1); stage A1(t + 1)
    read_A(buf, 1);
    stage_A(buf ^ 1, 1, kt + 1);
    P8_LOAD_END(VM03)
    P8_COMPUTE(1, 1)
    // phase 3: (A1, B0); stage A0(t + 2)
    read_B(buf, 0);
    stage_A(buf, 0, kt + 2);
    P8_LOAD_END(VM03)
    P8_COMPUTE(1, 0)
  }
#undef P8_LOAD_END
#undef P8_COMPUTE
  if (grp == 0) __builtin_amdgcn_s_barrier();  // the barrier count of the second wave group
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the overshoot stages (clamped re-reads) must land before the LDS is released

  // ---- epilogue: the accumulators go through LDS (free now) so that a lane owns 8 CONSECUTIVE columns of a row: bias / act /
  //      BN affine / residual / accumulate on vectors, one 16-byte store per 8 outputs.  Every wave stages its own 128 x 64 block
  //      in four rounds of 32 rows (a wave-private LDS region: no workgroup barrier, the wave's DS operations execute in order) ----
  __builtin_amdgcn_s_barrier();  // every wave is through its last fragment reads: the staging buffers may be overwritten
  const bool plain = g.act == ACT_NONE && g.act2 == ACT_NONE && !g.scale && !g.shift;
  constexpr int EST = 68;  // floats per staged row (64 + 4: rows 272 bytes apart)
  float* est = reinterpret_cast<float*>(smem) + wave * 32 * EST;
  const int er = lane >> 3, ec = (lane & 7) * 8;  // read-back: row er + 8 * it, columns ec .. ec + 7
  const int ncol = n0 + wc * 64 + ec;  // column inside this phase's N
  const int pcol = phase * g.N;        // ... and where the phase's columns start in C / R / ADD
  const bool vec_ok = ncol + 8 <= g.N && (pcol % 8) == 0 && ((((uintptr_t)g.C | (uintptr_t)g.R | (uintptr_t)g.ADD) & 15) == 0) && (g.ldc % 8) == 0 && (!g.R || g.ldr % 8 == 0) && (!g.ADD || g.ldadd % 8 == 0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int nj = 0; nj < 4; ++nj)
#pragma unroll
        for (int r = 0; r < 4; ++r) est[(mh * 16 + fq * 4 + r) * EST + nj * 16 + fr] = acc[2 * q + mh][nj][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      const int m = m0 + wr * 128 + q * 32 + row;
      if (m >= g.M || ncol >= g.N) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = est[row * EST + ec + e];
      if (g.ksplit > 1) {  // raw sums of this split's K range (N % 64 == 0, nphase == 1: always whole 16-byte groups)
        float* wp = g.ws + ((size_t)split * g.M + m) * g.N + ncol;
        *reinterpret_cast<float4*>(wp) = float4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<float4*>(wp + 4) = float4{v[4], v[5], v[6], v[7]};
        continue;
      }
      p8_finish8<TC>(g, T, m, ncol, pcol, vec_ok, plain, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the read-back is done before the next round overwrites the region
  }
}

// Second launch of a K-split GEMM: one thread per 8 consecutive outputs adds the splits' raw sums in split order (fixed: the result
// does not depend on which workgroup finished first) and runs the epilogue of the unsplit kernel.
template <typename TC>
__global__ __launch_bounds__(256) void gemm_p8_reduce_kernel(GemmArgs g) {
  const int groups = g.N / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)g.M * groups) return;
  const int m = (int)(i / groups), ncol = (int)(i - (long)m * groups) * 8;
  float v[8];
  const size_t plane = (size_t)g.M * g.N;
  const float* wp = g.ws + (size_t)m * g.N + ncol;
  {
    const float4 a = *reinterpret_cast<const float4*>(wp), b = *reinterpret_cast<const float4*>(wp + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  for (int s = 1; s < g.ksplit; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(wp + s * plane), b = *reinterpret_cast<const float4*>(wp + s * plane + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  const int T = g.T > 0 ? g.T : g.M;
  const bool plain = g.act == ACT_NONE && g.act2 == ACT_NONE && !g.scale && !g.shift;
  const bool vec_ok = ((((uintptr_t)g.C | (uintptr_t)g.R | (uintptr_t)g.ADD) & 15) == 0) && (g.ldc % 8) == 0 && (!g.R || g.ldr % 8 == 0) && (!g.ADD || g.ldadd % 8 == 0);
  p8_finish8<TC>(g, T, m, ncol, 0, vec_ok, plain, v);
}

template <int WM, int WN, typename TC>
int launch_p8(const GemmArgs& g, hipStream_t s) {
  constexpr int BMt = WM * 128, BNt = WN * 64;
  const int tiles_m = (g.M + BMt - 1) / BMt, tiles_n = (g.N + BNt - 1) / BNt;
  hipLaunchKernelGGL((gemm_p8_kernel<WM, WN, TC>), dim3(tiles_m * tiles_n * g.nphase * g.ksplit), dim3(512), 0, s, g, tiles_m, tiles_n);
  ITTS_HIP_CHECK(hipGetLastError());
  if (g.ksplit > 1) {
    const long items = (long)g.M * (g.N / 8);
    hipLaunchKernelGGL((gemm_p8_reduce_kernel<TC>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, g);
    ITTS_HIP_CHECK(hipGetLastError());
  }
  return OK;
}

}  // namespace

// Large regular shapes only: >= 4 K-tiles, enough tiles to fill the 256 CUs at least once and a half; N a multiple of 256 takes the
// 256 x 256 tile, a multiple of 128 the 512 x 128 tile.
static bool p8_wide(const GemmArgs& g) { return g.N % 256 == 0; }
// Number of tiles the launch would have, 0 where the kernel cannot run the shape at all.  WHEN it is the right kernel is the
// selector's business (c_api.cpp gemm_which): >= 200 tiles always; 64..199 where the 128-wide LDS-DMA kernel has no better claim.
long gemm_p8_tiles(const GemmArgs& g, int ta, int tw, int tc) {
  if (ta != BF16 || tw != BF16 || (tc != BF16 && tc != F32)) return 0;
  if (g.Cin % 64 != 0 || g.lda % 8 != 0 || g.in_up != 1 || g.nphase < 1 || g.nphase > 8) return 0;
  if (((uintptr_t)g.A & 15) || ((uintptr_t)g.W & 15)) return 0;
  if (g.N < 192 || g.N % 64 != 0 || (long)g.taps * g.Cin < 256) return 0;
  const int bm = p8_wide(g) ? 256 : 512, bn = p8_wide(g) ? 256 : 128;
  return (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * g.nphase;
}
bool gemm_p8_supported(const GemmArgs& g, int ta, int tw, int tc) { return gemm_p8_tiles(g, ta, tw, tc) * (g.ksplit > 1 ? g.ksplit : 1) >= 64; }

int gemm_p8(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s) {
  ITTS_REQUIRE(g.A && g.W && g.C, "gemm_p8: null pointer");
  ITTS_REQUIRE(gemm_p8_tiles(g, ta, tw, tc) > 0, "gemm_p8: unsupported shape/dtype");  // (whether it is the RIGHT kernel: gemm_which)
  const int T = g.T > 0 ? g.T : g.M;
  ITTS_REQUIRE(g.M % T == 0 && g.lda >= g.Cin && g.ldc >= g.N * g.nphase, "gemm_p8: bad dims");
  ITTS_REQUIRE(g.ksplit >= 1 && (g.ksplit == 1 || (g.ws && g.nphase == 1 && !((uintptr_t)g.ws & 15) &&
                                                  (long)(g.ksplit - 1) * ((g.taps * (g.Cin / PBK) + g.ksplit - 1) / g.ksplit) < (long)g.taps * (g.Cin / PBK))),
               "gemm_p8: K split needs a workspace, one phase and a K-tile for every split");
  if (p8_wide(g)) return tc == BF16 ? launch_p8<2, 4, bf16_t>(g, s) : launch_p8<2, 4, float>(g, s);
  return tc == BF16 ? launch_p8<4, 2, bf16_t>(g, s) : launch_p8<4, 2, float>(g, s);
}

}  // namespace itts
