// Persistent decode engine (decode batches <= 6 rows - beam rows included -, bf16 weights): ONE token step as ONE launch - the 24
// GPT-2 blocks, then ln_f -> final_norm -> mel_head and, for greedy search, the sampler (repetition penalty, arg-max, bookkeeping,
// next input embedding).
//
// Reference hot loop: GPT2InferenceModel.forward, indextts/gpt/model.py:115-192 (HF 4.36.2 GPT2Block: LN -> c_attn ->
// attention over the cache -> c_proj + residual -> LN -> c_fc -> gelu_new -> c_proj + residual).  The launch path
// (decode2.hip, five graph-captured kernels per layer) is bound by its 120 dependent kernel boundaries per token, not by
// HBM.  Here every one of the chip's 256 CUs holds ONE 1024-thread workgroup for the whole step:
//
//   * each workgroup owns a fixed slice of every projection's output features (16 of c_attn - a head's q / k / v rows are
//     spread over the 12 workgroups of that head's group -, 5 of c_proj, 20 of c_fc, 5 of mlp.c_proj: 153.6 KB of bf16
//     weights per layer) and streams it by LDS-DMA (global_load_lds, nontemporal, no VGPR touched) into three LDS slots,
//     one to four phases ahead of its use: ONE loader wave per workgroup issues every request, in a fixed FIFO order, as
//     soon as the target slot's previous occupant has been read, and waits with counted vmcnt in front of the barriers;
//   * the five all-to-all phase edges of a layer (residual stream -> LN1, q/k/v -> attention, context -> c_proj,
//     residual stream -> LN2, gelu(fc) -> mlp.c_proj) are 8-byte {value, tag} granules: written with ONE agent-scope
//     (sc1, write-through) store by the lane that finished the value, polled with agent-scope loads by the four GATHER
//     waves of every consuming workgroup - waves that never issue a weight load, so a poll never queues behind the
//     workgroup's own weight stream (vmcnt is in-order per wave); the eleven COMPUTE waves neither poll nor prefetch;
//   * tag = a step counter kept in device memory and advanced by the kernel itself: every (layer, edge) has its own
//     granule block, so a tag can only match a value of THIS step; nothing is reset between launches (graph replay safe);
//   * one attention workgroup per (row, head) - workgroup `row` of the head's group - runs the four key splits of
//     decode_attn2_kernel<.., 256, 4> on its 16 waves and merges them in LDS: no cross-CU hop for the partials.  Its K/V
//     rows are requested right after the layer's first gather, one projection ahead of the scores.
//
// Arithmetic is the launch path's, operation for operation (same lane <-> k mapping and accumulation order of the GEMV
// dot products, same LayerNorm reduction tree, same attention window / split / merge): logits and ids are bit-identical
// to gemv_bf16_kernel + decode_attn2_kernel at 1 - 4 rows (tests/test_gpu_engine_persistent.py); 5 - 6 rows keep that
// arithmetic row for row (the launch path runs on the matrix cores there: tolerance + row-independence tests).
// LDS maps by row count: three weight slots (<= 2 rows), two slots with aliased edge buffers (3 - 4), a half slot B (5 - 6).
//
// Every spin is bounded (wall clock, s_memrealtime) and also ends on a chip-wide abort word; a workgroup that gave up
// runs on without waiting, so the grid always drains.  The host reads the abort word at status / fetch.
#include <cstdlib>
#include <type_traits>

#include "itts_decode.h"
#include "itts_engine_kernel.h"
#include "itts_sampler_dev.h"
#include "decode_pinned.h"

#ifndef ITTS_KV_PREFETCH
#define ITTS_KV_PREFETCH 1  // attention workgroups touch their cache rows (one dword per 64 bytes) at the head of the block: the rows
                            // are in this XCD's L2 when the register window is requested behind c_attn
#endif
#ifndef ITTS_EARLY_KV
#define ITTS_EARLY_KV 0  // window pairs of the cache attention requested INTO REGISTERS at the head of the block (0: all behind c_attn).
                         // Measured r04: 3 / 6 pairs early cost 12 spilled VGPRs, whose scratch reloads queue behind the HBM misses
                         // (vmcnt is in-order): 0.4365 -> 0.457 / 0.473 ms per step.  The L2 prefetch below gets the lead time without registers.
#endif

namespace itts {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// full-wave sum, uniform in every lane (decode2.hip wave_sum_rl: same tree, same bits)
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
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return half_dot2(a, b, c);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return half_lo(w); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return half_hi(w); }

struct KVec {  // 8 bf16 of one cache row (decode2.hip CacheVec<bf16_t>)
  u32x4 raw;
  __device__ __forceinline__ void load(const bf16_t* p) { raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
  __device__ __forceinline__ float get(int i) const {
    const uint32_t w = raw[i >> 1];
    return (i & 1) ? bf16_hi(w) : bf16_lo(w);
  }
};

// ---- run-time state of one workgroup's hand-offs ----
struct Rt {
  unsigned tag;
  bool dead;  // a wait of this thread gave up: later waits return at once, the kernel drains
  u64 t0;
  unsigned limit;
  unsigned* ctr;  // [0] step counter (tag), [1] abort word
  int first_delay, pass_sleep;  // s_sleep units before the first pass of a gather / between passes
};

// debugging aid (EngArgs::stamp): wall-clock (100 MHz) stamp i of block l of this workgroup
#define ENG_STAMP(i)                                                                                              \
  if (a.stamp && (tl & 63) == 0 && (i < 8 ? tl == 0 : tl == 320))                                                \
    a.stamp[((size_t)cu * a.NL + l) * 16 + (i)] = (unsigned)__builtin_amdgcn_s_memrealtime();

__device__ __forceinline__ u64 ld_gran(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(u64* p, unsigned tag, uint32_t v) {
  __hip_atomic_store(p, ((u64)tag << 32) | (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool spin_fail(Rt& rt, unsigned& spins) {
  if ((++spins & 15u) != 0) return false;
  unsigned abw;  // asm load + wait: no load the compiler counts in a gather path (see "gathering an edge" below)
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(abw) : "v"(rt.ctr + 1) : "memory");
  const bool ab = abw != 0;
  const bool to = (u64)(__builtin_amdgcn_s_memrealtime() - rt.t0) > (u64)rt.limit;
  if (ab || to) {
    if (!ab) __hip_atomic_store(rt.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    rt.dead = true;
    return true;
  }
  return false;
}

// ---- gathering an edge ----
// The gather waves never hold a load hipcc knows about: every poll is an asm load with its own wait.  (A load the
// compiler counts leaves "a write to these VGPRs may be pending" in its scoreboard at the join with the compute waves'
// path, and it then guards the next reuse of those registers with s_waitcnt vmcnt(0) - in front of the compute waves'
// LDS-DMA issue that drained the weight prefetch at every phase: 2 us per phase on the timeline.)
__device__ __forceinline__ u32x4 ld_gran2(const u64* p) {  // two adjacent granules, one 16-byte agent-scope load
  u32x4 x;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x) : "v"(p) : "memory");
  return x;
}
__device__ __forceinline__ u64 ld_gran1(const u64* p) {
  u64 x;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x) : "v"(p) : "memory");
  return x;
}
// The wait of a pass NAMES every destination register of the pass as read-write (cdna_hip_programming.md 5.7 item 1, form (ii)):
// hipcc takes an asm load's destination as written at the end of the load statement, so without this operand list it would be
// free to copy, spill or re-allocate x[j] between the load and the wait - correctness would rest on today's register
// allocation.  With the operands every later use of x[j] depends on THIS statement, and nothing between a load and it may
// touch the registers (the loads' own "=v" outputs feed only this statement).
template <int CH>
__device__ __forceinline__ void ld_wait(u32x4 (&x)[CH]) {
  static_assert(CH >= 1 && CH <= 10, "operand list of the wait");
#define ITTS_W(i) "+v"(x[i < CH ? i : 0])
  if constexpr (CH == 1) asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0) : : "memory");
  else if constexpr (CH == 2) asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1) : : "memory");
  else if constexpr (CH == 3) asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2) : : "memory");
  else if constexpr (CH == 4) asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3) : : "memory");
  else if constexpr (CH == 5) asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3), ITTS_W(4) : : "memory");
  else if constexpr (CH == 6) asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3), ITTS_W(4), ITTS_W(5) : : "memory");
  else if constexpr (CH == 7)
    asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3), ITTS_W(4), ITTS_W(5), ITTS_W(6) : : "memory");
  else if constexpr (CH == 8)
    asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3), ITTS_W(4), ITTS_W(5), ITTS_W(6), ITTS_W(7) : : "memory");
  else if constexpr (CH == 9)
    asm volatile("s_waitcnt vmcnt(0)" : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3), ITTS_W(4), ITTS_W(5), ITTS_W(6), ITTS_W(7), ITTS_W(8) : : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)"
                 : ITTS_W(0), ITTS_W(1), ITTS_W(2), ITTS_W(3), ITTS_W(4), ITTS_W(5), ITTS_W(6), ITTS_W(7), ITTS_W(8), ITTS_W(9)
                 :
                 : "memory");
#undef ITTS_W
  __builtin_amdgcn_sched_barrier(0);
}

// LDS counter of the compute waves' finished phases: a workgroup's own publish is the best local estimate of "the edge
// is being published chip-wide", so a gather starts its first pass then instead of polling a sentinel over the fabric
__device__ __forceinline__ void wait_own(unsigned ctr_lds, unsigned target, Rt& rt) {  // ctr_lds: LDS byte address
  unsigned spins = 0;
  while (!rt.dead) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(ctr_lds) : "memory");
    if (v >= target) break;
    if (spin_fail(rt, spins)) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

// One gather thread (gt = 0..255) collects the granule PAIRS gt, gt + 256, ... < npairs of an edge: every load of a pass
// in flight at once, passes repeated (with a pause) until every tag matches.  sink(granule index, value) runs once per
// granule.  SENTINEL: nothing local says when the edge will be published (the context edge on a workgroup that runs no
// attention), so ONE lane per wave polls ONE granule first - a full pass per workgroup, repeated while the producers are
// still computing, competes with their weight stream and stores for the fabric (MI355X_MICROARCH "polling-cost").
template <int PER, bool SENTINEL, typename Sink>
__device__ __forceinline__ void sweep2(const u64* __restrict__ g, int npairs, int gt, Rt& rt, Sink&& sink, int delay) {
  if (SENTINEL) {
    const int w = gt >> 6;
    const u64* __restrict__ p = g + 2 * min(npairs - 1, (npairs / 4) * w + npairs / 8);
    unsigned spins = 0;
    bool ok = rt.dead || (gt & 63) != 0;
    while (!ok) {
      ok = (unsigned)(ld_gran1(p) >> 32) == rt.tag;
      if (ok || spin_fail(rt, spins)) break;
      __builtin_amdgcn_s_sleep(2);
    }
  }
  constexpr int CH = PER <= 10 ? PER : 8;  // loads in flight per pass (registers: 4 each)
  for (int z = 0; z < delay; ++z) __builtin_amdgcn_s_sleep(1);
#pragma unroll
  for (int j0 = 0; j0 < PER; j0 += CH) {
    unsigned need = 0;
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (j0 + j < PER && gt + 256 * (j0 + j) < npairs) need |= 1u << j;
    if (rt.dead) need = 0;
    unsigned spins = 0;
    while (need) {
      u32x4 x[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j)
        if ((need >> j) & 1u) x[j] = ld_gran2(g + 2 * (gt + 256 * (j0 + j)));  // only what has not arrived yet
      // (x[j] stays undefined where the load was skipped - it is not looked at below; giving it a value there would make the
      // compiler MERGE two values behind the load statement, i.e. copy the destination register before the data has landed)
      ld_wait<CH>(x);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (((need >> j) & 1u) && x[j][1] == rt.tag && x[j][3] == rt.tag) {
          sink(2 * (gt + 256 * (j0 + j)), x[j][0]);
          sink(2 * (gt + 256 * (j0 + j)) + 1, x[j][2]);
          need &= ~(1u << j);
        }
      }
      if (!need) break;
      if (spin_fail(rt, spins)) break;
      for (int z = 0; z < rt.pass_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
  }
}

// ---- weights: global -> LDS by LDS-DMA, one phase ahead of their use ----
// One wave instruction moves a 1 KiB fragment (512 bf16 of one weight row; lane l: elements 8l .. 8l + 7 - the lane <-> k
// mapping of gemv_bf16_kernel) without touching a VGPR.  Written as asm so that hipcc neither counts it nor drains it in
// front of every barrier (cdna_hip_programming.md 5.7): the issuing wave waits with its own s_waitcnt vmcnt(0) before
// the phase barrier behind which the fragment is read.  nt: every weight byte is read once per step (nt-weights).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_addr) {  // 4 bytes per lane (bias rows)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
}

// rows [n0, n0 + ROWS) of W [N][K] -> LDS slot (64 bias floats, then compact rows of K * 2 bytes): issued by ONE wave (the
// loader), ROWS * ceil(K / 512) + 1 instructions - the counts the loader's waits are written in
template <int ROWS, int K>
struct DmaCount {
  static constexpr int NF = (K + 511) / 512, N = ROWS * NF + 1;
};
// THIN > 0: at most THIN requests in flight (the loader waits in between) - for the one projection that has to be
// requested while this CU's gather waves are polling (MI355X_MICROARCH "gather-pass": a gather pass queued behind its own
// CU's unthrottled refill burst takes 2 - 3 x as long)
template <int ROWS, int K, int THIN = 0>
__device__ __forceinline__ void dma_rows(const bf16_t* __restrict__ W, const float* __restrict__ bias, int n0, unsigned slot, int lane) {
  constexpr int NF = (K + 511) / 512, TAIL = K - (NF - 1) * 512;  // elements in the last fragment
  static_assert(TAIL == 512 || TAIL == 256, "whole or half last fragment");
  for (int r = 0; r < ROWS; ++r) {
    const bf16_t* src = W + (size_t)(n0 + r) * K + lane * 8;
    const unsigned dst = slot + 256 + (unsigned)(r * K * 2);
#pragma unroll
    for (int c = 0; c < NF; ++c) {
      if (TAIL == 512 || c != NF - 1) {
        dma16(src + c * 512, dst + c * 1024);
      } else if (lane < 32) {
        dma16(src + c * 512, dst + c * 1024);
      }
      if (THIN > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(THIN) : "memory");
    }
  }
  if (lane < ROWS) dma4(bias + n0 + lane, slot);
}
// rows [n0, n0 + ROWS) of W [N][K] -> compact rows at LDS address dst, no bias row (5 - 6 rows: c_attn arrives in two parts)
template <int ROWS, int K>
__device__ __forceinline__ void dma_rows_at(const bf16_t* __restrict__ W, int n0, unsigned dst0, int lane) {
  constexpr int NF = (K + 511) / 512, TAIL = K - (NF - 1) * 512;
  static_assert(TAIL == 512 || TAIL == 256, "whole or half last fragment");
  for (int r = 0; r < ROWS; ++r) {
    const bf16_t* src = W + (size_t)(n0 + r) * K + lane * 8;
    const unsigned dst = dst0 + (unsigned)(r * K * 2);
#pragma unroll
    for (int c = 0; c < NF; ++c)
      if (TAIL == 512 || c != NF - 1 || lane < 32) dma16(src + c * 512, dst + c * 1024);
  }
}
// columns [K0, K0 + KN) of rows [n0, n0 + ROWS) of W [N][KTOT] -> compact rows of KN * 2 bytes (5 - 6 rows: mlp.c_proj comes
// in two K halves, the second into slot A once c_fc has been read); the bias row travels with the first part
template <int ROWS, int KTOT, int K0, int KN>
__device__ __forceinline__ void dma_rows_part(const bf16_t* __restrict__ W, const float* __restrict__ bias, int n0, unsigned slot, int lane) {
  static_assert(KN % 512 == 0 && K0 % 512 == 0, "whole fragments");
  for (int r = 0; r < ROWS; ++r) {
    const bf16_t* src = W + (size_t)(n0 + r) * KTOT + K0 + lane * 8;
    const unsigned dst = slot + 256 + (unsigned)(r * KN * 2);
#pragma unroll
    for (int c = 0; c < KN / 512; ++c) dma16(src + c * 512, dst + c * 1024);
  }
  if (bias && lane < ROWS) dma4(bias + n0 + lane, slot);
}
// the loader waits until at most N of its requests are outstanding: everything older has landed in LDS (vmcnt is in-order)
template <int N>
__device__ __forceinline__ void dma_wait_keep() {
  static_assert(N >= 0 && N <= 63, "vmcnt field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// acc[b] += W[row] . x[b] in gemv_bf16_kernel's order (fragments ascending, four v_dot2c per fragment); W from its LDS slot
template <int NB, int K>
__device__ __forceinline__ void dots(const unsigned char* __restrict__ wrow, const uint32_t* __restrict__ sxb, int lane,
                                     float (&acc)[NB]) {
  constexpr int NCH = (K + 511) / 512;
  const int klast = (NCH - 1) * 512 + lane * 8;
  const bool kok = klast < K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
    // lanes past the end of a half last fragment multiply by x = 0 like the launch path; they read a valid (finite) weight
    const u32x4 w = *reinterpret_cast<const u32x4*>(wrow + c * 1024 + ((c == NCH - 1 && !kok) ? (lane & 31) : lane) * 16);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      u32x4 xq = *reinterpret_cast<const u32x4*>(sxb + (b * K + k) / 2);
      if (c == NCH - 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) xq[e] = kok ? xq[e] : 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[b] = dot2(w[e], xq[e], acc[b]);
    }
  }
}

// the same over fragments [C0, C0 + NC) of a K = KTOT row whose part lies compact at wpart (whole fragments only)
template <int NB, int KTOT, int C0, int NC>
__device__ __forceinline__ void dots_part(const unsigned char* __restrict__ wpart, const uint32_t* __restrict__ sxb, int lane,
                                          float (&acc)[NB]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = (C0 + c) * 512 + lane * 8;
    const u32x4 w = *reinterpret_cast<const u32x4*>(wpart + c * 1024 + lane * 16);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const u32x4 xq = *reinterpret_cast<const u32x4*>(sxb + (b * KTOT + k) / 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[b] = dot2(w[e], xq[e], acc[b]);
    }
  }
}

// LayerNorm without affine of xf [NB][K] fp32 -> bf16 pairs sxb [NB][K/2]: the prologue of gemv_bf16_kernel<.., PRO = 1>.
// A row is normalised by 256 threads in that kernel's thread <-> element mapping (thread t: elements [4t, 4t + 4) and, for
// t < 64, [1024 + 4t, ..)), same shifted moments, same wave / workgroup reduction order.  Rows are independent, so row 0 is
// taken by waves 0..3 and row 1 by waves 12..15 at the same time (the chain load -> moments -> 4 wave reductions -> LDS ->
// barrier -> rsqrt -> normalise -> LDS is latency-bound on one wave per SIMD: 0.9 us for two rows on one group).
template <int K>
struct LnRow {
  float xv[2][4], pivot;
  bool xok[2];
};
template <int K>
__device__ __forceinline__ void ln_row_load(const float* __restrict__ x, int t, LnRow<K>& r) {
  constexpr int NTHR = 256;
  static_assert(K > NTHR * 4 && K <= NTHR * 8, "two 4-element chunks per thread");
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = (t + j * NTHR) * 4;
    r.xok[j] = i < K;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (r.xok[j] ? i : K - 4));
#pragma unroll
    for (int e = 0; e < 4; ++e) r.xv[j][e] = v[e];
  }
  r.pivot = x[0];
}
// shifted moments (shift pv) of this thread's elements -> per-wave sums in red [4][2]; a workgroup barrier follows
template <int K>
__device__ __forceinline__ void ln_row_moments(const LnRow<K>& r, float pv, float* __restrict__ red, int t) {
  const int lane = t & 63, wave = t >> 6;
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = r.xok[j] ? r.xv[j][e] - pv : 0.f;
      s += d;
      q = fmaf(d, d, q);
    }
  s = wave_sum_rl(s);
  q = wave_sum_rl(q);
  if (lane == 0) {
    red[wave * 2] = s;
    red[wave * 2 + 1] = q;
  }
}
template <int K>
__device__ __forceinline__ void ln_row_norm(LnRow<K>& r, float pv, const float* __restrict__ red, float eps) {
  const float invK = 1.f / (float)K;
  float S = 0.f, Q = 0.f;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    S += red[ww * 2];
    Q += red[ww * 2 + 1];
  }
  const float md = __fmul_rn(S, invK);
  const float mean = __fadd_rn(pv, md);
  const float rstd = __builtin_amdgcn_rsqf(__fadd_rn(fmaxf(ln_var_rn(Q, invK, md), 0.f), eps));
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) r.xv[j][e] = (r.xv[j][e] - mean) * rstd;
}
template <int K>
__device__ __forceinline__ void ln_row_store(const LnRow<K>& r, uint32_t* __restrict__ sx, int t) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (r.xok[j]) {
      uint2 p;
      p.x = pack_bf16(r.xv[j][0], r.xv[j][1]);
      p.y = pack_bf16(r.xv[j][2], r.xv[j][3]);
      *reinterpret_cast<uint2*>(sx + (t + j * 256) * 2) = p;
    }
}
// row handled by the wave group q (= thread / 256): rows run at once on different groups; the loader wave's group (1) is used last
__device__ __forceinline__ int ln_row_of(int nb, int q) {
  return q == 0 ? 0 : q == 3 ? (nb == 2 ? 1 : nb >= 3 ? 2 : -1) : q == 2 ? (nb >= 3 ? 1 : -1) : (nb == 4 ? 3 : -1);
}
// all 1024 threads call this (two workgroup barriers inside); red: [2 passes][4 or 8 rows][4][2] floats.  5 - 6 rows: a wave
// group takes a second row (rows 4, 5) between the same two barriers.
template <int NB, int K>
__device__ __forceinline__ void ln_to_sxb(const float* __restrict__ xf, uint32_t* __restrict__ sxb, float* __restrict__ red, int t,
                                          float eps) {
  const int grp = ln_row_of(NB < 4 ? NB : 4, t >> 8);  // which row this thread works on (-1: none)
  const int grp2 = NB > 4 ? (ln_row_of(NB - 4, t >> 8) < 0 ? -1 : ln_row_of(NB - 4, t >> 8) + 4) : -1;
  const int tr = t & 255;
  LnRow<K> r, r2;
  if (grp >= 0) {
    ln_row_load<K>(xf + grp * K, tr, r);
    ln_row_moments<K>(r, r.pivot, red + grp * 8, tr);
  }
  if (NB > 4 && grp2 >= 0) {
    ln_row_load<K>(xf + grp2 * K, tr, r2);
    ln_row_moments<K>(r2, r2.pivot, red + grp2 * 8, tr);
  }
  __syncthreads();
  if (grp >= 0) {
    ln_row_norm<K>(r, r.pivot, red + grp * 8, eps);
    ln_row_store<K>(r, sxb + grp * (K / 2), tr);
  }
  if (NB > 4 && grp2 >= 0) {
    ln_row_norm<K>(r2, r2.pivot, red + grp2 * 8, eps);
    ln_row_store<K>(r2, sxb + grp2 * (K / 2), tr);
  }
  __syncthreads();
}
// ln_f (affine) then final_norm (its affine is folded into mel_head by the packer): gemv_bf16_kernel's prologue 2
template <int NB, int K>
__device__ __forceinline__ void ln2_to_sxb(const float* __restrict__ xf, uint32_t* __restrict__ sxb, float* __restrict__ red, int t,
                                           float eps, const float* __restrict__ gamma, const float* __restrict__ beta) {
  const int grp = ln_row_of(NB < 4 ? NB : 4, t >> 8);
  const int grp2 = NB > 4 ? (ln_row_of(NB - 4, t >> 8) < 0 ? -1 : ln_row_of(NB - 4, t >> 8) + 4) : -1;
  constexpr int RED2 = NB > 4 ? 64 : 32;  // second pass's sums
  const int tr = t & 255;
  LnRow<K> r, r2;
  f32x4 gm[2], bt[2];
  if (grp >= 0 || (NB > 4 && grp2 >= 0)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ic = (tr + j * 256) * 4 < K ? (tr + j * 256) * 4 : K - 4;
      gm[j] = *reinterpret_cast<const f32x4*>(gamma + ic);
      bt[j] = *reinterpret_cast<const f32x4*>(beta + ic);
    }
  }
  if (grp >= 0) {
    ln_row_load<K>(xf + grp * K, tr, r);
    ln_row_moments<K>(r, r.pivot, red + grp * 8, tr);
  }
  if (NB > 4 && grp2 >= 0) {
    ln_row_load<K>(xf + grp2 * K, tr, r2);
    ln_row_moments<K>(r2, r2.pivot, red + grp2 * 8, tr);
  }
  __syncthreads();
  if (grp >= 0) {
    ln_row_norm<K>(r, r.pivot, red + grp * 8, eps);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) r.xv[j][e] = ln_affine_rn(r.xv[j][e], gm[j][e], bt[j][e]);
    ln_row_moments<K>(r, 0.f, red + RED2 + grp * 8, tr);
  }
  if (NB > 4 && grp2 >= 0) {
    ln_row_norm<K>(r2, r2.pivot, red + grp2 * 8, eps);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) r2.xv[j][e] = ln_affine_rn(r2.xv[j][e], gm[j][e], bt[j][e]);
    ln_row_moments<K>(r2, 0.f, red + RED2 + grp2 * 8, tr);
  }
  __syncthreads();
  if (grp >= 0) {
    ln_row_norm<K>(r, 0.f, red + RED2 + grp * 8, eps);
    ln_row_store<K>(r, sxb + grp * (K / 2), tr);
  }
  if (NB > 4 && grp2 >= 0) {
    ln_row_norm<K>(r2, 0.f, red + RED2 + grp2 * 8, eps);
    ln_row_store<K>(r2, sxb + grp2 * (K / 2), tr);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// ANC: beam rows - the cache is never re-ordered when the beams are; key j of beam row b lives in the physical row its
// ancestry names (decode2.hip, decode_attn2_kernel<.., ANC>: the same gather, the same arithmetic)
template <int NB, bool ANC>
__global__ __launch_bounds__(1024) void decode_engine_kernel(EngArgs a) {
  constexpr int D = ENG_D, H = ENG_H, DH = 64, NCU = ENG_NCU;
  constexpr int GPH = NCU / H;       // workgroups per head group (12)
  constexpr int QO = 3 * DH / GPH;   // c_attn rows per workgroup (16)
  constexpr int HO = D / NCU;        // residual-projection rows per workgroup (5)
  constexpr int FO = 4 * D / NCU;    // c_fc rows per workgroup (20)
  constexpr int NCW = 11;            // compute waves 5..15; gather waves 0..3; wave 4 = the loader (weight prefetch)
  static_assert(GPH * QO == 3 * DH && HO * NCU == D && FO * NCU == 4 * D && (FO % 2) == 0, "partition");
  static_assert(QO <= 2 * NCW && HO <= NCW && FO / 2 <= NCW, "wave assignment");
  constexpr int NIT = 3, NSPLIT = 4, SLOTS = 32, LPK = 8, VEC = 8, UNC = 2;  // decode_attn2_kernel<.., 3, 256, 4>
  constexpr int SD = 2;  // rows in flight beyond the register window (the launch path takes 4: same rows, same order)
  static_assert(ATTN_NSPLIT == NSPLIT, "split count of the launch path");
  static_assert(NB <= ENG_MAX_ROWS, "LDS budget");

  // LDS map.  Three weight slots (bias row + rows): slot A holds c_attn, then c_fc; slot B mlp.c_proj; slot C c_proj.  A slot is
  // refilled only once every compute wave has finished the phase that read its previous occupant (the `own` counter).
  // Every edge lands in a buffer of its own: a gather may run while the compute waves still read the previous input.
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr unsigned SLOT = 256 + FO * D * 2;  // = 256 + HO * 4D * 2
  constexpr unsigned SLOTC = 256 + HO * D * 2;
  static_assert(SLOT >= 256 + QO * D * 2 && SLOT >= 256 + HO * 4 * D * 2, "slot size");
  constexpr bool SLOT3 = NB <= 2;
  // 5 - 6 rows: slot B holds HALF of mlp.c_proj (the first 2560 columns of its 5 rows, requested behind the second LayerNorm);
  // the second half goes into slot A once every compute wave is through c_fc, and the next block's c_attn follows into A
  // when they are through mlp.c_proj (waited for in front of the next block's first barrier, behind the E1 hop).
  constexpr bool HALFB = NB >= 5;
  constexpr unsigned SLOTB = HALFB ? 256 + HO * 2 * D * 2 : SLOT;
  constexpr int QA1 = 10;                              // HALFB: c_attn rows 0..9 at A + QOFF1, rows 10..15 at A + 256
  constexpr unsigned QOFF1 = 256 + HO * 2 * D * 2;     // behind the second half of mlp.c_proj
  static_assert(QOFF1 + QA1 * D * 2 <= SLOT && (QO - QA1) * D * 2 <= HO * 2 * D * 2, "c_attn parts fit slot A");
  static_assert(SLOTB >= SLOTC, "c_proj fits slot B");
  unsigned char* W0 = smem;                 // A
  unsigned char* W1 = smem + SLOT;          // B
  unsigned char* W2 = SLOT3 ? smem + 2 * SLOT : W1;  // C (3 - 6 rows: c_proj in slot B)
  // 3 - 4 rows: no slot C (c_proj shares slot B with mlp.c_proj, which is then requested behind the second LayerNorm), and the
  // fp32 residual-stream buffer and the context buffer live inside the gelu(fc) buffer: that one is written by the E5 gather
  // and read by mlp.c_proj only, and every gather into the aliases starts behind wait_own() = after every compute wave is
  // through the phase that read the previous content.  LDS: 147.8 KB at 3 rows, 160.6 KB of 163.8 at 4.
  // 5 - 6 rows: 149.1 / 160.9 KB.
  uint32_t* xn = reinterpret_cast<uint32_t*>(smem + SLOT + SLOTB + (SLOT3 ? SLOTC : 0));  // [NB][D / 2] LayerNorm output, bf16 pairs
  uint32_t* xa = xn + NB * D / 2;                                   // [NB][4D / 2] gelu(fc), bf16 pairs
  float* xf = SLOT3 ? reinterpret_cast<float*>(xa + NB * 2 * D) : reinterpret_cast<float*>(xa);             // [NB][D] fp32
  uint32_t* xc = SLOT3 ? reinterpret_cast<uint32_t*>(xf + NB * D) : xa + NB * D;                           // [NB][D / 2]
  float* red = reinterpret_cast<float*>(SLOT3 ? xc + NB * D / 2 : xa + NB * 2 * D);   // [2][4][4][2]
  float* hown = red + (NB > 4 ? 128 : 64);                          // [NB][8] this workgroup's slice of the residual stream
  float* qkvs = hown + (NB > 4 ? NB : 4) * 8;                                       // [3][64] q / k / v of this (row, head)
  float* so = qkvs + 3 * DH;                                        // [16][64]
  float* smx = so + 16 * DH;                                        // [16]
  float* slx = smx + 16;                                            // [16]
  float* po = slx + 16;                                             // [4][64] partial of each key split
  float* pml = po + NSPLIT * DH;                                    // [2][4]
  unsigned* own = reinterpret_cast<unsigned*>(pml + 8);             // [1] phases finished by the compute waves (x NCW)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned S0 = lds0, S1 = lds0 + SLOT, S2 = SLOT3 ? lds0 + 2 * SLOT : S1;
  static_assert(SLOT + SLOTB + (SLOT3 ? SLOTC : 0) + NB * D * 2 + NB * 4 * D * 2 + (SLOT3 ? NB * D * 4 + NB * D * 2 : 0) + (NB > 4 ? 512 : 256) +
                        (NB > 4 ? NB : 4) * 32 + (3 * DH + 16 * DH + 32 + NSPLIT * DH + 8 + 1) * 4 <= 160 * 1024, "LDS budget");
  const unsigned own_lds = lds0 + (unsigned)(reinterpret_cast<unsigned char*>(own) - smem);

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cu = blockIdx.x;
  const bool gw = wave < 4;    // gather waves
  const bool lw = wave == 4;   // the loader: every weight byte of this workgroup goes through its LDS-DMA requests
  const bool cwv = wave > 4;   // compute waves
  const int cw = wave - 5;
  const int gh = cu / GPH, gm = cu % GPH;
  const bool qcu = cu < H * GPH;        // takes part in c_attn
  const bool acu = qcu && gm < NB;      // runs the attention of (row gm, head gh)
  // c_attn rows of this workgroup: q, k or v (gm / 4) of head gh, 16 consecutive features
  const int an0 = (gm / 4) * D + gh * DH + (gm % 4) * QO;

  Rt rt;
  rt.t0 = __builtin_amdgcn_s_memrealtime();
  const u64 clk0 = __builtin_amdgcn_s_memtime();
  rt.limit = a.timeout_ticks;
  rt.ctr = a.ctr;
  rt.dead = false;
  rt.first_delay = a.first_delay;
  rt.pass_sleep = a.pass_sleep;
  // every value the kernel loads once is read to an SGPR HERE (readfirstlane waits for it): a load still "maybe pending" at
  // the head of the layer loop makes hipcc guard its first use in every iteration with s_waitcnt vmcnt(0) - which at run
  // time also drains the weight prefetch this wave has just issued
  rt.tag = __builtin_amdgcn_readfirstlane(a.ctr[0]);
  const int prefix = __builtin_amdgcn_readfirstlane(a.prefix[0]);
  const int my_pos = acu ? __builtin_amdgcn_readfirstlane(prefix + a.len[acu ? gm : 0]) : 0;  // cache row this step appends
  const int my_ks = acu ? __builtin_amdgcn_readfirstlane(a.kv_start[acu ? gm : 0]) : 0;
  const int my_par = (ANC && acu) ? __builtin_amdgcn_readfirstlane(a.len[acu ? gm : 0] & 1) : 0;  // ancestry ping-pong half

  if (t == 0) *own = 0;
  unsigned phase = 0;  // compute phases finished so far (every wave counts alike)
  auto phase_done = [&](int ln_) {  // a compute wave is through a phase: its publishes are issued
    if (cwv && ln_ == 0) __hip_atomic_fetch_add(own, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ++phase;
  };
  // The loader's requests, each right BEHIND the barrier that ends the last use of the slot's previous occupant (so the
  // issue burst - a CU takes ~1 KiB of LDS-DMA per 10 ns - runs beside the compute waves' dot products, not beside a gather:
  // a gather pass queued behind its own CU's refill burst takes 2 - 3 x as long, MI355X_MICROARCH "gather-pass"):
  //   behind B1: mlp.c_proj(l) -> B    behind B3: c_fc(l) -> A    behind B4: c_proj(l + 1) -> C    behind B5: c_attn(l + 1) -> A
  // vmcnt is in-order, so c_proj / mlp.c_proj (requested three barriers before their use) have landed whenever the request
  // after them has; the loader waits (vmcnt(0)) only in front of B1 (c_attn) and B4 (c_fc), each 2.5 - 3 us after the request.
  // (3 - 4 rows, two slots: B1 -> c_proj(l) -> B, B3 -> c_fc(l) -> A, B4 -> mlp.c_proj(l) -> B, B5 -> c_attn(l + 1) -> A, and the
  // loader waits in front of every barrier for the one request before it)
  if (lw) {
    if (SLOT3) dma_rows<HO, D>(a.L[0].wp, a.L[0].bp, cu * HO, S2, lane);
    if (qcu && HALFB) {  // (5 - 6 rows: c_attn lives in two parts - rows 0..9 behind the half of mlp.c_proj that shares slot A)
      dma_rows_at<QA1, D>(a.L[0].wa, an0, S0 + QOFF1, lane);
      dma_rows_at<QO - QA1, D>(a.L[0].wa, an0 + QA1, S0 + 256, lane);
      if (lane < QO) dma4(a.L[0].ba + an0 + lane, S0);
    } else if (qcu) {
      dma_rows<QO, D>(a.L[0].wa, a.L[0].ba, an0, S0, lane);
    }
  }
  if (t < NB * HO) hown[(t / HO) * 8 + t % HO] = a.h[(size_t)(t / HO) * D + cu * HO + t % HO];

  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing the compiler counts is pending at the head of the layer loop
  constexpr size_t LSTRIDE = (size_t)NB * D * 15 / 2;
  constexpr size_t OQKV = 0, OCTX = (size_t)NB * 3 * D, OH1 = OCTX + NB * D / 2, OACT = OH1 + NB * D, OH2 = OACT + NB * 2 * D;
  for (int l = 0; l < a.NL; ++l) {
    const EngLayerW& w = a.L[l];
    // per-thread indices re-derived from an opaque copy each layer: hipcc otherwise hoists every per-thread address of
    // the five phases out of the layer loop and spills them (53 VGPRs of scratch traffic inside the hand-off waits)
    int tl = t;
    asm volatile("" : "+v"(tl));
    const int ll = tl & 63;
    u64* __restrict__ G = a.gran + (size_t)l * LSTRIDE;

    // ---- cache rows of this block's attention (workgroup (row gm, head gh)): the register window of every key split ----
    // Requested at the HEAD of the block (a.early_kv), not behind c_attn: the rows of earlier positions do not depend on anything
    // this block computes, and an HBM miss takes 1 - 2 us under load - behind c_attn it ended 0.8 - 1.5 us after the q / k / v
    // hand-off had arrived (r03 timeline: q / k / v published at 1.8 us, polled at 3.8).  Who asks when: the compute waves at
    // once (they wait for no memory operation in this phase); a gather wave behind its E1 sweep and the loader behind its wait
    // for c_attn (vmcnt is in-order: a wait of theirs in front of these requests would hold the sweep / the first barrier for
    // an HBM miss).  Same rows, same registers, same arithmetic as before.
    constexpr int NIT_ = 3, NSPLIT_ = 4, SLOTS_ = 32, LPK_ = 8, VEC_ = 8, UNC_ = 2;
    KVec kr[2 * NIT_], vr[2 * NIT_];
    const int a_sp = wave >> 2, a_atid = (wave & 3) * 64 + ll, a_slot = a_atid / LPK_, a_sub = a_atid % LPK_;
    const size_t a_lo = ((size_t)l * a.B * H + (size_t)(acu ? gm : 0) * H + gh) * a.Smax * DH;
    const uint8_t* a_arow = nullptr;
    size_t a_lbase = 0;  // (ANC) first row of this beam's batch item in block l
    if constexpr (ANC) {
      a_arow = a.anc + ((size_t)my_par * a.B + (acu ? gm : 0)) * a.Smax;
      a_lbase = ((size_t)l * a.B + (size_t)((acu ? gm : 0) / a.nb) * a.nb) * H;
    }
    auto krow = [&](int j) -> const bf16_t* {
      if constexpr (ANC) return a.kc + ((a_lbase + (size_t)min((int)a_arow[j], a.nb - 1) * H + gh) * a.Smax + j) * DH;
      return a.kc + a_lo + (size_t)j * DH;
    };
    auto vrow = [&](int j) -> const bf16_t* {
      if constexpr (ANC) return a.vc + ((a_lbase + (size_t)min((int)a_arow[j], a.nb - 1) * H + gh) * a.Smax + j) * DH;
      return a.vc + a_lo + (size_t)j * DH;
    };
    auto kv_issue = [&](auto lo_c, auto hi_c) {  // window pairs [lo, hi)
      constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
      // split = wave / 4 is an SGPR: the "was this pair of the window requested" tests are scalar branches that skip the dead
      // half of the window (S = 380: 3 of 6 pairs per split) instead of computing it under a select
#pragma unroll
      for (int u = LO; u < HI; ++u)
        if (u < UNC_ || (u * NSPLIT_ + a_sp) * SLOTS_ < my_pos + 1) {
          const int j = min((u * NSPLIT_ + a_sp) * SLOTS_ + a_slot, a.Smax - 1);
          kr[u].load(krow(j) + a_sub * VEC_);
          vr[u].load(vrow(j) + a_sub * VEC_);
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int EKV = ITTS_EARLY_KV;  // window pairs requested at the head of the block (the rest behind c_attn)
    using Ic0 = std::integral_constant<int, 0>;
    using IcE = std::integral_constant<int, EKV>;
    using IcN = std::integral_constant<int, 2 * NIT_>;
    // L2 prefetch of the cache rows [kv_start, S) of this (row, head): one dword per 64 bytes of every K and V row, by the compute
    // waves (they wait for no memory operation in this phase, so nothing queues behind these misses), into ONE register that is
    // never read - it only has to stay out of the allocator's hands until the loads have landed (kept alive to the head of P2,
    // behind the q / k / v poll, 2 - 3 us later).  Default cache policy: the lines are to stay in L2 for the nontemporal window
    // loads behind c_attn, which then take an L2 hit (~0.1 us) instead of an HBM miss (1 - 2 us under load).
    unsigned pf_sink = 0;
    if (ITTS_KV_PREFETCH && acu && cwv) {
      const int c0 = cw * 64 + ll;  // 0 .. 703
      for (int j = my_ks + c0; j < my_pos; j += NCW * 64) {
        const char* pk = reinterpret_cast<const char*>(krow(j));
        const char* pv = reinterpret_cast<const char*>(vrow(j));
        asm volatile("global_load_dword %0, %1, off\n\tglobal_load_dword %0, %1, off offset:64\n\t"
                     "global_load_dword %0, %2, off\n\tglobal_load_dword %0, %2, off offset:64"
                     : "+v"(pf_sink)  // read-write: ONE live range over all iterations (a fresh "=v" per iteration would let the
                                      // allocator recycle the previous iteration's register while its load is still in flight)
                     : "v"(pk), "v"(pv)
                     : "memory");
      }
    }
    const bool early_kv = acu && EKV > 0;
    if (early_kv && cwv) kv_issue(Ic0{}, IcE{});

    // ================= P1: residual stream -> LN1 -> c_attn =================
    if (gw) {
      if (l == 0) {
        for (int i = tl; i < NB * D; i += 256) xf[i] = a.h[i];
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing the compiler counts stays pending past the gather branch
      } else {
        wait_own(own_lds, NCW * phase, rt);
        sweep2<(NB * D / 2 + 255) / 256, false>(G - LSTRIDE + OH2, NB * D / 2 / a.fake_div, tl, rt, [&](int i, uint32_t v) { xf[i] = __uint_as_float(v); }, rt.first_delay);
      }
      if (early_kv) kv_issue(Ic0{}, IcE{});
    } else if (lw) {
      if (HALFB && l > 0 && qcu) {  // the rest of this block's c_attn -> A: every compute wave is through the last block's mlp.c_proj
        wait_own(own_lds, NCW * phase, rt);
        dma_rows_at<QO - QA1, D>(w.wa, an0 + QA1, S0 + 256, ll);
      }
      dma_wait_keep<0>();  // c_attn (and, before it, c_proj) of this block
      if (early_kv) kv_issue(Ic0{}, IcE{});
    }
    ENG_STAMP(0)
    __syncthreads();
    ENG_STAMP(1)
    ln_to_sxb<NB, D>(xf, xn, red, tl, a.eps);
    // mlp.c_proj -> B (every wave is past the last block's); behind the LayerNorm's barriers, which the loader must not hold up.
    // Not on an attention workgroup: its loader is one of the 16 attention waves, and the cache rows it requests next would
    // queue behind these 50 KB (vmcnt is in-order) - there mlp.c_proj follows behind the second LayerNorm
    if (SLOT3) {
      if (lw && !acu) dma_rows<HO, 4 * D>(w.w2, w.b2, cu * HO, S1, ll);
    } else if (lw) {
      dma_rows<HO, D>(w.wp, w.bp, cu * HO, S1, ll);  // c_proj -> B (every wave is past the last block's mlp.c_proj)
    }
    if (qcu && cwv) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int r = cw + NCW * s;
        if (r < QO) {
          float acc[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = 0.f;
          dots<NB, D>(HALFB ? (r < QA1 ? W0 + QOFF1 + r * D * 2 : W0 + 256 + (r - QA1) * D * 2) : W0 + 256 + r * D * 2, xn, ll, acc);
          float mine = 0.f;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float x = wave_sum_rl(acc[b]);
            mine = ll == b ? x : mine;
          }
          if (ll < NB) {
            st_gran(G + OQKV + (size_t)ll * 3 * D + an0 + r, rt.tag, __float_as_uint(mine + reinterpret_cast<const float*>(W0)[r]));
            if (a.dbg && l == a.dbg_layer) a.dbg[(size_t)ll * 3 * D + an0 + r] = mine + reinterpret_cast<const float*>(W0)[r];
          }
        }
      }
    }

    phase_done(ll);
    // c_fc -> A on the workgroups that run no attention: requested early_fc x 64 clocks after this workgroup's compute waves are
    // through c_attn (slot A is free then), i.e. while the attention workgroups compute, instead of behind B3 beside the h'
    // gather (a gather pass queued behind its own CU's refill burst takes 2 - 3 x as long).  The delay matters: right behind
    // c_attn the 51 KB land beside the q / k / v hop and the cache rows (0.456 ms per step against 0.448 for "behind B3"); 1.9 us
    // later - the attention workgroups are computing, nothing else is on the fabric - 0.436.  0: behind B3 everywhere.
    if (a.early_fc > 0 && lw && !acu) {
      wait_own(own_lds, NCW * phase, rt);
      for (int z = 0; z < a.early_fc; ++z) __builtin_amdgcn_s_sleep(1);
      dma_rows<FO, D>(w.wf, w.bf, cu * FO, S0, ll);
    }
    ENG_STAMP(8)

    // ================= P2: cache attention of (row gm, head gh) =================
    if (acu) {
      // K/V rows of the register window: requested at the head of the block (early_kv) or - the r03 placement - here, before the
      // q / k / v hand-off is polled (all 16 waves; the thread <-> (split, slot, sub) mapping of decode_attn2_kernel<.., 256, 4>)
      const int sp = a_sp, atid = a_atid, slot = a_slot, sub = a_sub;
      bf16_t* kb = a.kc + a_lo;
      bf16_t* vb = a.vc + a_lo;
      kv_issue(IcE{}, IcN{});  // what was not requested at the head of the block
      const int pos = my_pos, S = pos + 1, ks = my_ks;
      __builtin_amdgcn_sched_barrier(0);
      if (gw && tl < 3 * DH) {  // q / k / v of this (row, head): 192 granules, one per thread
        const u64* __restrict__ p = G + OQKV + (size_t)gm * 3 * D + (tl / DH) * D + gh * DH + tl % DH;
        wait_own(own_lds, NCW * phase, rt);
        unsigned spins = 0;
        while (!rt.dead) {
          const u64 x = ld_gran1(p);
          if ((unsigned)(x >> 32) == rt.tag) {
            qkvs[tl] = __uint_as_float((uint32_t)x);
            break;
          }
          if (spin_fail(rt, spins)) break;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      ENG_STAMP(2)
      if (ITTS_KV_PREFETCH && cwv) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pf_sink) : : "memory");  // (the window loads are needed now anyway)
      __syncthreads();
      float qr[VEC];
      const bool own = atid < LPK && sp == 0;  // slot 0 of split 0 appends this step's row
#pragma unroll
      for (int i = 0; i < VEC; ++i) qr[i] = qkvs[sub * VEC + i] * a.scale;
      if (own) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          kb[(size_t)pos * DH + sub * VEC + i] = (bf16_t)qkvs[DH + sub * VEC + i];
          vb[(size_t)pos * DH + sub * VEC + i] = (bf16_t)qkvs[2 * DH + sub * VEC + i];
        }
      }
      float m = -INFINITY, lsum = 0.f, acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      auto score = [&](const KVec& kk) {
        float sc = 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) sc = fmaf(qr[i], kk.get(i), sc);
        sc = dpp_add<0xB1>(sc);
        sc = dpp_add<0x4E>(sc);
        sc = dpp_add<0x141>(sc);
        return sc;
      };
      {
        float sc[2 * NIT + 1];
#pragma unroll
        for (int u = 0; u < 2 * NIT; ++u) {
          const int j = (u * NSPLIT + sp) * SLOTS + slot;
          sc[u] = -INFINITY;
          if (u < UNC || (u * NSPLIT + sp) * SLOTS < S) {  // wave-uniform (scalar)
            const float tt = score(kr[u]);
            sc[u] = (j < S && j >= ks && j != pos) ? tt : -INFINITY;
          }
        }
        {  // the row appended by this step, with the cache's rounding, from LDS
          float tt = 0.f;
#pragma unroll
          for (int i = 0; i < VEC; ++i) tt = fmaf(qr[i], own ? (float)(bf16_t)qkvs[DH + sub * VEC + i] : 0.f, tt);
          tt = dpp_add<0xB1>(tt);
          tt = dpp_add<0x4E>(tt);
          tt = dpp_add<0x141>(tt);
          sc[2 * NIT] = (slot == 0 && sp == 0) ? tt : -INFINITY;
        }
        float mw = sc[0];
#pragma unroll
        for (int u = 1; u <= 2 * NIT; ++u) mw = fmaxf(mw, sc[u]);
        if (mw > -INFINITY) {
#pragma unroll
          for (int u = 0; u < 2 * NIT; ++u)
            if (u < UNC || (u * NSPLIT + sp) * SLOTS < S) {
              const float p = __expf(sc[u] - mw);
              lsum += p;
#pragma unroll
              for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, vr[u].get(i), acc[i]);
            }
          const float p = __expf(sc[2 * NIT] - mw);
          lsum += p;
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, own ? (float)(bf16_t)qkvs[2 * DH + sub * VEC + i] : 0.f, acc[i]);
          m = mw;
        }
      }
      auto consume = [&](const KVec& kk, const KVec& vv, int j) {
        const bool ok = j < S && j >= ks && j != pos;
        float sc = score(kk);
        sc = ok ? sc : -INFINITY;
        const float mn = fmaxf(m, sc);
        const float corr = mn > -INFINITY ? __expf(m - mn) : 1.f;
        const float p = ok ? __expf(sc - mn) : 0.f;
        lsum = fmaf(lsum, corr, p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(p, ok ? vv.get(i) : 0.f, acc[i] * corr);
        m = mn;
      };
      for (int cb = 2 * NIT; (cb * NSPLIT + sp) * SLOTS < S; cb += SD) {
        KVec k2[SD], v2[SD];
#pragma unroll
        for (int u = 0; u < SD; ++u) {
          const int j = min(((cb + u) * NSPLIT + sp) * SLOTS + slot, a.Smax - 1);
          k2[u].load(krow(j) + sub * VEC);
          v2[u].load(vrow(j) + sub * VEC);
        }
#pragma unroll
        for (int u = 0; u < SD; ++u) consume(k2[u], v2[u], ((cb + u) * NSPLIT + sp) * SLOTS + slot);
      }
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
      lsum *= sc0;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] *= sc0;
#pragma unroll
      for (int o = LPK; o < 64; o <<= 1) {
        lsum = bfly_sum(lsum, o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = bfly_sum(acc[i], o);
      }
      if (ll < LPK)
#pragma unroll
        for (int i = 0; i < VEC; ++i) so[wave * DH + ll * VEC + i] = acc[i];
      if (ll == 0) {
        smx[wave] = M;
        slx[wave] = lsum;
      }
      __syncthreads();
      if (atid < DH) {  // the four waves of split sp -> its partial (the tail of decode_attn2_kernel)
        float MM = smx[4 * sp];
#pragma unroll
        for (int i = 1; i < 4; ++i) MM = fmaxf(MM, smx[4 * sp + i]);
        float o = 0.f, L = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float e = smx[4 * sp + i] > -INFINITY ? __expf(smx[4 * sp + i] - MM) : 0.f;
          o = fmaf(e, so[(4 * sp + i) * DH + atid], o);
          L = fmaf(e, slx[4 * sp + i], L);
        }
        po[sp * DH + atid] = o;
        if (atid == 0) {
          pml[sp] = MM;
          pml[NSPLIT + sp] = L;
        }
      }
      __syncthreads();
      if (tl < DH / 2) {  // merge of the four partials (gemv_bf16_kernel prologue 3), two dims = one bf16 pair per thread
        const float Mx = fmaxf(fmaxf(pml[0], pml[1]), fmaxf(pml[2], pml[3]));
        float wgt[NSPLIT], L = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < NSPLIT; ++s2) {
          wgt[s2] = pml[s2] > -INFINITY ? __expf(pml[s2] - Mx) : 0.f;
          L = fmaf(wgt[s2], pml[NSPLIT + s2], L);
        }
        const float inv = 1.f / L;
        float xm[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float v = 0.f;
#pragma unroll
          for (int s2 = 0; s2 < NSPLIT; ++s2) v = fmaf(wgt[s2], po[s2 * DH + 2 * tl + e], v);
          xm[e] = v * inv;
        }
        st_gran(G + OCTX + (size_t)gm * (D / 2) + gh * (DH / 2) + tl, rt.tag, pack_bf16(xm[0], xm[1]));
      }
      ENG_STAMP(3)
    }

    // ================= P3: context -> c_proj + residual =================
    if (gw) {
      if (acu)
        sweep2<(NB * D / 4 + 255) / 256, false>(G + OCTX, NB * D / 4, tl, rt, [&](int i, uint32_t v) { xc[i] = v; }, rt.first_delay);
      else
        sweep2<(NB * D / 4 + 255) / 256, true>(G + OCTX, NB * D / 4, tl, rt, [&](int i, uint32_t v) { xc[i] = v; }, a.ctx_delay);
    } else if (!SLOT3 && lw) {
      dma_wait_keep<0>();  // c_proj (requested behind this block's first LayerNorm)
    }  // (three slots: c_proj is older in the loader's queue than c_attn, it landed before this block's first barrier)
    ENG_STAMP(4)
    __syncthreads();
    if (lw && !(a.early_fc > 0 && !acu)) {  // c_fc -> A: every wave is past c_attn
      if (a.thin_fc)
        dma_rows<FO, D, 12>(w.wf, w.bf, cu * FO, S0, ll);
      else
        dma_rows<FO, D>(w.wf, w.bf, cu * FO, S0, ll);
    }
    if (cwv && cw < HO) {
      float acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = 0.f;
      dots<NB, D>(W2 + 256 + cw * D * 2, xc, ll, acc);
      float mine = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float x = wave_sum_rl(acc[b]);
        mine = ll == b ? x : mine;
      }
      if (ll < NB) {
        const float hn = hown[ll * 8 + cw] + (mine + reinterpret_cast<const float*>(W2)[cw]);
        hown[ll * 8 + cw] = hn;
        st_gran(G + OH1 + (size_t)ll * D + cu * HO + cw, rt.tag, __float_as_uint(hn));
        if (a.dbg && l == a.dbg_layer) a.dbg[(size_t)NB * 3 * D + (size_t)ll * D + cu * HO + cw] = hn;
      }
    }
    phase_done(ll);
    ENG_STAMP(9)
    // ================= P4: residual stream -> LN2 -> c_fc -> gelu_new =================
    if (gw) {
      wait_own(own_lds, NCW * phase, rt);
      sweep2<(NB * D / 2 + 255) / 256, false>(G + OH1, NB * D / 2 / a.fake_div, tl, rt, [&](int i, uint32_t v) { xf[i] = __uint_as_float(v); }, rt.first_delay);
    } else if (lw) {
      dma_wait_keep<0>();  // c_fc (and, before it, mlp.c_proj)
    }
    ENG_STAMP(5)
    __syncthreads();
    ENG_STAMP(12)
    ln_to_sxb<NB, D>(xf, xn, red, tl, a.eps);
    if (SLOT3) {
      if (lw && acu) dma_rows<HO, 4 * D>(w.w2, w.b2, cu * HO, S1, ll);
      if (lw && l + 1 < a.NL) dma_rows<HO, D>(a.L[l + 1].wp, a.L[l + 1].bp, cu * HO, S2, ll);  // next c_proj -> C: all past this one's
    } else if (lw && HALFB) {
      dma_rows_part<HO, 4 * D, 0, 2 * D>(w.w2, w.b2, cu * HO, S1, ll);  // first half of mlp.c_proj -> B: all past c_proj
    } else if (lw) {
      dma_rows<HO, 4 * D>(w.w2, w.b2, cu * HO, S1, ll);  // mlp.c_proj -> B: all past c_proj
    }
    ENG_STAMP(13)
    if (cwv && cw < FO / 2) {  // an adjacent pair of features per wave: one bf16-pair granule per batch row
      float acc[2][NB];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
        dots<NB, D>(W0 + 256 + (2 * cw + r) * D * 2, xn, ll, acc[r]);
      }
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float x = wave_sum_rl(acc[r][b]);
          mine = ll == r * NB + b ? x : mine;
        }
      ENG_STAMP(14)
      const float v = gelu_new_rn(mine + reinterpret_cast<const float*>(W0)[2 * cw + (ll < NB ? 0 : 1)]);
      const float v1 = __shfl_down(v, NB, 64);  // (feature 1, batch row ll) for lanes < NB
      if (ll < NB) {
        st_gran(G + OACT + (size_t)ll * 2 * D + cu * (FO / 2) + cw, rt.tag, pack_bf16(v, v1));
        if (a.dbg && l == a.dbg_layer) {
          float* d = a.dbg + (size_t)NB * 4 * D + (size_t)ll * 4 * D + cu * FO + 2 * cw;
          d[0] = (float)(bf16_t)v;
          d[1] = (float)(bf16_t)v1;
        }
      }
    }

    phase_done(ll);
    ENG_STAMP(10)

    // ================= P5: gelu(fc) -> mlp.c_proj + residual =================
    if (gw) {
      wait_own(own_lds, NCW * phase, rt);
      sweep2<(NB * D + 255) / 256, false>(G + OACT, NB * D / a.fake_div, tl, rt, [&](int i, uint32_t v) { xa[i] = v; }, a.act_delay);
    } else if (lw && HALFB) {
      wait_own(own_lds, NCW * phase, rt);  // every compute wave is through c_fc: slot A is free
      dma_rows_part<HO, 4 * D, 2 * D, 2 * D>(w.w2, nullptr, cu * HO, S0, ll);  // second half of mlp.c_proj -> A
      dma_wait_keep<0>();
    } else if (lw && !SLOT3) {
      dma_wait_keep<0>();  // mlp.c_proj (requested behind LN2)
    } else if (lw && acu) {  // mlp.c_proj was requested behind LN2 here; the next block's c_proj (if any) is younger
      if (l + 1 < a.NL)
        dma_wait_keep<DmaCount<HO, D>::N>();
      else
        dma_wait_keep<0>();
    }  // (elsewhere mlp.c_proj is older in the loader's queue than c_fc: it landed before this block's c_fc barrier)
    ENG_STAMP(6)
    __syncthreads();
    ENG_STAMP(7)
    if (HALFB) {  // rows 0..9 of the next c_attn -> the part of A that the second half of mlp.c_proj leaves free (+ its bias row)
      if (lw && qcu && l + 1 < a.NL) {
        dma_rows_at<QA1, D>(a.L[l + 1].wa, an0, S0 + QOFF1, ll);
        if (ll < QO) dma4(a.L[l + 1].ba + an0 + ll, S0);
      }
    } else if (lw && qcu && l + 1 < a.NL) {
      dma_rows<QO, D>(a.L[l + 1].wa, a.L[l + 1].ba, an0, S0, ll);  // next c_attn -> A: all past c_fc
    }
    if (cwv && cw < HO) {
      float acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = 0.f;
      if (HALFB) {
        dots_part<NB, 4 * D, 0, 5>(W1 + 256 + cw * 2 * D * 2, xa, ll, acc);
        dots_part<NB, 4 * D, 5, 5>(W0 + 256 + cw * 2 * D * 2, xa, ll, acc);
      } else {
        dots<NB, 4 * D>(W1 + 256 + cw * 4 * D * 2, xa, ll, acc);
      }
      float mine = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float x = wave_sum_rl(acc[b]);
        mine = ll == b ? x : mine;
      }
      if (ll < NB) {
        const float hn = hown[ll * 8 + cw] + (mine + reinterpret_cast<const float*>(W1)[cw]);
        hown[ll * 8 + cw] = hn;
        if (a.dbg && l == a.dbg_layer) a.dbg[(size_t)NB * 8 * D + (size_t)ll * D + cu * HO + cw] = hn;
        if (l + 1 < a.NL || a.head_w)
          st_gran(G + OH2 + (size_t)ll * D + cu * HO + cw, rt.tag, __float_as_uint(hn));
        else
          a.h[(size_t)ll * D + cu * HO + cw] = hn;
      }
    }
    phase_done(ll);
    ENG_STAMP(11)
  }
  // ================= head: ln_f -> final_norm -> mel_head (lm_head, model.py:48,180) =================
  // The workgroup's 32 (33) rows of mel_head go straight to REGISTERS of the compute waves (three rows each, requested before
  // the gather: compute waves never poll, so the loads fly during the hop); logits leave as plain stores for the sampler launch.
  if (a.head_w) {
    int tl = t;
    asm volatile("" : "+v"(tl));
    const int ll = tl & 63;
    const int rows_per = a.V / NCU, extra = a.V % NCU;
    u32x4 wh[3][3];
    float bh[3];
    int hn_[3];
    if (cwv) {
      const int klast = 2 * 512 + ll * 8;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int idx = cw * 3 + r;
        hn_[r] = idx < rows_per ? cu * rows_per + idx : (idx == rows_per && cu < extra ? NCU * rows_per + cu : -1);
        const int n = hn_[r] < 0 ? 0 : hn_[r];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int k = c == 2 ? (klast < D ? klast : D - 8) : c * 512 + ll * 8;
          wh[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.head_w + (size_t)n * D + k));
        }
        bh[r] = a.head_b[n];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (gw) {
      wait_own(own_lds, NCW * phase, rt);
      sweep2<(NB * D / 2 + 255) / 256, false>(a.gran + (size_t)(a.NL - 1) * LSTRIDE + OH2, NB * D / 2, tl, rt,
                                              [&](int i, uint32_t v) { xf[i] = __uint_as_float(v); }, rt.first_delay);
    }
    __syncthreads();
    ln2_to_sxb<NB, D>(xf, xn, red, tl, a.eps, a.lnf_g, a.lnf_b);
    if (cwv) {
      const int klast = 2 * 512 + ll * 8;
      const bool kok = klast < D;
      float acc[3][NB];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int k = c == 2 ? (kok ? klast : D - 8) : c * 512 + ll * 8;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          u32x4 xq = *reinterpret_cast<const u32x4*>(xn + (b * D + k) / 2);
          if (c == 2)
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[e] = kok ? xq[e] : 0u;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][b] = dot2(wh[r][c][e], xq[e], acc[r][b]);
        }
      }
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float x = wave_sum_rl(acc[r][b]);
          mine = ll == r * NB + b ? x : mine;
        }
      if (ll < 3 * NB) {
        const int r = ll / NB, b = ll % NB;
        const int n = r == 0 ? hn_[0] : (r == 1 ? hn_[1] : hn_[2]);
        const float bias = r == 0 ? bh[0] : (r == 1 ? bh[1] : bh[2]);
        if (n >= 0) a.logits[(size_t)b * a.V + n] = mine + bias;
      }
      if (a.fold_sampler && ll < 3 * NB) {  // this lane's candidate of row b: penalty / stop suppression as sampler2_kernel
        const int r = ll / NB, b = ll % NB;
        const int n = r == 0 ? hn_[0] : (r == 1 ? hn_[1] : hn_[2]);
        const float bias = r == 0 ? bh[0] : (r == 1 ? bh[1] : bh[2]);
        float* cv = so;                                      // [NB][36] candidate scores (the attention scratch is idle)
        int* ci = reinterpret_cast<int*>(so + ENG_MAX_ROWS * 36);       // [NB][36] ids
        cv[b * 36 + cw * 3 + r] = n >= 0 ? sampler_score(a.samp, a.samp.seen + (size_t)b * a.V, mine + bias, n) : -INFINITY;
        ci[b * 36 + cw * 3 + r] = n >= 0 ? n : 0x7fffffff;
      }
    }
    if (a.fold_sampler) {
      // ---- greedy sampler: arg-max over the vocabulary = over every workgroup's best (larger score, lower id on ties:
      //      the rule of sampler2_kernel, independent of the reduction order) ----
      const float* cv = so;
      const int* ci = reinterpret_cast<const int*>(so + ENG_MAX_ROWS * 36);
      __syncthreads();
      if (tl < NB) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int q = 0; q < 3 * NCW; ++q) {
          const float v = cv[tl * 36 + q];
          const int i = ci[tl * 36 + q];
          if (v > best || (v == best && i < bi)) {
            best = v;
            bi = i;
          }
        }
        st_gran(a.cand + ((size_t)tl * NCU + cu) * 2, rt.tag, __float_as_uint(best));
        st_gran(a.cand + ((size_t)tl * NCU + cu) * 2 + 1, rt.tag, (uint32_t)bi);
      }
      if (cu < NB) {  // workgroup b finishes row b
        const int b = cu;
        const int k_pre = a.samp.step[b], unf_pre = a.samp.unfinished[b];
        float* sv = red;                                  // [4] wave bests
        int* si = reinterpret_cast<int*>(red + 8);        // [4] ... and, afterwards, [0] token [1] next position
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (gw) {  // thread tl: the candidate of workgroup tl
          const u64* p = a.cand + ((size_t)b * NCU + tl) * 2;
          unsigned spins = 0;
          bool have_v = false, have_i = false;
          while (!rt.dead && !(have_v && have_i)) {
            const u64 x = ld_gran1(p), y = ld_gran1(p + 1);
            if ((unsigned)(x >> 32) == rt.tag) {
              best = __uint_as_float((uint32_t)x);
              have_v = true;
            }
            if ((unsigned)(y >> 32) == rt.tag) {
              bi = (int)(uint32_t)y;
              have_i = true;
            }
            if (have_v && have_i) break;
            if (spin_fail(rt, spins)) break;
            __builtin_amdgcn_s_sleep(1);
          }
          if (!(have_v && have_i)) {
            best = -INFINITY;
            bi = 0x7fffffff;
          }
          auto merge = [&](float ov, int oi) {
            if (ov > best || (ov == best && oi < bi)) {
              best = ov;
              bi = oi;
            }
          };
#define ENG_ARGMAX_DPP(CTRL)                                                                          \
  merge(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xf, 0xf, true)),  \
        __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, true))
          ENG_ARGMAX_DPP(0xB1);
          ENG_ARGMAX_DPP(0x4E);
          ENG_ARGMAX_DPP(0x141);
          ENG_ARGMAX_DPP(0x140);
#undef ENG_ARGMAX_DPP
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
          if (ll == 0) {
            sv[wave] = best;
            si[wave] = bi;
          }
        }
        __syncthreads();
        if (t == 0) {
          for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
              best = sv[w];
              bi = si[w];
            }
          // A gather lane that gave up keeps the sentinel id: with the abort word set every lane of this workgroup may have
          // (all 256 candidates dead).  Nothing is committed then - no id, no seen-bit, no embedding row from an id outside
          // the vocabulary - the abort word is (re)raised and the host fails this generation (model_gpt.hip engine_check).
          if (bi < 0 || bi >= a.V) {
            __hip_atomic_store(a.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            si[0] = a.samp.stop;
            si[1] = -1;  // sampler_next_embedding: nothing to write
          } else {
            sampler_commit(a.samp, b, bi, si, k_pre, unf_pre);
          }
        }
        __syncthreads();
        sampler_next_embedding(a.samp, b, si, t);
      }
    }
  }
  // advance the step counter (never 0): every workgroup read it before its first publish, and this workgroup got here
  // only after gathering from all of them
  if (cu == 0 && t == 0) a.ctr[0] = rt.tag + 1 == 0 ? 1u : rt.tag + 1;
  if (a.stamp && t == 0) {  // shader clock over the launch: cycles per 10 ns tick, x 1000 (stamp 15 of block 0)
    const u64 c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    a.stamp[(size_t)cu * a.NL * 16 + 15] = (unsigned)((c1 - clk0) * 1000 / (r1 - rt.t0 + 1));
    if (a.NL >= 3) {  // wall clock at this workgroup's start / end (stamp 15 of blocks 1 / 2)
      a.stamp[((size_t)cu * a.NL + 1) * 16 + 15] = (unsigned)rt.t0;
      a.stamp[((size_t)cu * a.NL + 2) * 16 + 15] = (unsigned)r1;
    }
  }
}

}  // namespace

size_t eng_gran_count(int layers) { return (size_t)layers * ENG_MAX_ROWS * ENG_D * 15 / 2 + ENG_CAND_WORDS; }

int decode_engine_layers(const EngArgs& a, hipStream_t s) {
  ITTS_REQUIRE(a.B >= 1 && a.B <= ENG_MAX_ROWS && a.NL >= 1 && a.NL <= ENG_MAX_LAYERS && a.gran && a.h && a.kc && a.vc && a.ctr, "decode_engine: bad arguments");
  const size_t lds = 160 * 1024;  // three weight slots + edge buffers: the whole LDS of a CU, one workgroup per CU
#define ITTS_ENG_GO(NB, ANC_)                                                                                                   \
  {                                                                                                                             \
    static bool attr = false;                                                                                                   \
    if (!attr) {                                                                                                                \
      ITTS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_engine_kernel<NB, ANC_>),                        \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                \
      attr = true;                                                                                                              \
    }                                                                                                                           \
    hipLaunchKernelGGL((decode_engine_kernel<NB, ANC_>), dim3(ENG_NCU), dim3(1024), lds, s, a);                                 \
  }
  if (a.anc) {  // beam rows (>= 2 rows per batch item)
    ITTS_REQUIRE(a.nb >= 2 && a.nb <= a.B && a.B % a.nb == 0, "decode_engine: beam ancestry needs B to be a multiple of 2 <= nb <= B");
    if (a.B == 2) ITTS_ENG_GO(2, true)
    else if (a.B == 3) ITTS_ENG_GO(3, true)
    else if (a.B == 4) ITTS_ENG_GO(4, true)
    else if (a.B == 5) ITTS_ENG_GO(5, true)
    else ITTS_ENG_GO(6, true)
  } else if (a.B == 1) ITTS_ENG_GO(1, false)
  else if (a.B == 2) ITTS_ENG_GO(2, false)
  else if (a.B == 3) ITTS_ENG_GO(3, false)
  else if (a.B == 4) ITTS_ENG_GO(4, false)
  else if (a.B == 5) ITTS_ENG_GO(5, false)
  else ITTS_ENG_GO(6, false)
#undef ITTS_ENG_GO
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
