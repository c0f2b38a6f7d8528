// Fused anti-aliased SnakeBeta ("Activation1d"): replicate-pad -> x2 polyphase up-FIR (12 taps) ->
// x + sin^2(x*e^a)/(e^b+1e-9) -> replicate-pad -> stride-2 12-tap down-FIR, in one pass.
//
// Replaces the reference's CUDA kernel anti_alias_activation_forward
// (indextts/BigVGAN/alias_free_activation/cuda/anti_alias_activation_cuda.cu:43-181) and matches the
// torch path it mirrors (alias_free_torch/act.py:24-29, resample.py:24-33, filter.py:86-95), which is
// the parity target (SURVEY.md 2.2 "K1 caveats").
//
// MI355X design: activations are channels-last [B,T,C], so a 64-lane wave walks 64 adjacent channels
// and every load/store of a time step is one contiguous row segment.  Each lane owns one channel and
// slides along a run of RUN time steps holding the 12-sample window of activated up-sampled values
// and the 6-sample input window in registers: per output it does 2 new up-FIR phases (12 FMA), 2
// SnakeBeta evaluations and the 12-tap down-FIR (no 2x intermediate ever reaches HBM).  fp32 math,
// I/O in fp32 or bf16.
#include <cstdlib>

#include "itts_kernels.h"
#include "itts_snake_dev.h"

namespace itts {
namespace {

constexpr int RUN = 32;
constexpr int RUNL = 36;  // LDS-tiled kernel: a multiple of 6, the rotation period of its unrolled inner loop

template <typename T>
struct SnakeCtx {
  const T* __restrict__ x;  // base of this (batch, channel): element t at x[t*C]
  int Tn, C;
  float ea, inv_b;
  float fu[12], fd[12];
  __device__ __forceinline__ float xin(int t) const {
    t = t < 0 ? 0 : (t >= Tn ? Tn - 1 : t);
    return ldf(x + (size_t)t * C);
  }
  __device__ __forceinline__ float snake(float u) const {
    const float sn = __sinf(u * ea);
    return u + inv_b * sn * sn;
  }
  __device__ __forceinline__ float snake_precise(float u) const {
    const float sn = sinf(u * ea);
    return u + inv_b * sn * sn;
  }
  // activated up-sampled sample m (clamped to [0, 2T-1]) straight from global memory
  __device__ __forceinline__ float v_at(int m) const {
    m = m < 0 ? 0 : (m > 2 * Tn - 1 ? 2 * Tn - 1 : m);
    const int q = m >> 1;
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu[2 * r], xin(q + 3 - r), u);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu[2 * r + 1], xin(q + 2 - r), u);
    }
    return snake_precise(2.f * u);
  }
};

template <typename T>
__global__ __launch_bounds__(256) void snake_aa_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                       const float* __restrict__ la, const float* __restrict__ lb,
                                                       const float* __restrict__ up12, const float* __restrict__ dn12,
                                                       int B, int Tn, int C, int nchunk) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * nchunk * C;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  const int ch = (int)((gid / C) % nchunk);
  const int b = (int)(gid / ((long)C * nchunk));
  SnakeCtx<T> k;
  k.x = x + (size_t)b * Tn * C + c;
  k.Tn = Tn;
  k.C = C;
  k.ea = expf(la[c]);
  k.inv_b = 1.f / (expf(lb[c]) + 1e-9f);
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    k.fu[i] = up12[i];
    k.fd[i] = dn12[i];
  }
  T* __restrict__ yo = y + (size_t)b * Tn * C + c;
  const int t0 = ch * RUN;
  const int t1 = min(t0 + RUN, Tn);
  // window v[j] = V(2t - 5 + j), j = 0..11
  float v[12];
#pragma unroll
  for (int j = 0; j < 10; ++j) v[j] = k.v_at(2 * t0 - 5 + j);
  // input window xs[i] = x[clamp(t + i)], i = 0..5
  float xs[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) xs[i] = k.xin(t0 + i);
  const int mlast = 2 * Tn - 1;
  for (int t = t0; t < t1; ++t) {
    // new samples m = 2t+5 (odd, q = t+2) and m = 2t+6 (even, q = t+3): both read x[t .. t+5]
    float uo = 0.f, ue = 0.f;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      uo = fmaf(k.fu[2 * r], xs[5 - r], uo);
      ue = fmaf(k.fu[2 * r + 1], xs[5 - r], ue);
    }
    const float vprev = v[9];
    v[10] = (2 * t + 5 <= mlast) ? k.snake_precise(2.f * uo) : vprev;
    v[11] = (2 * t + 6 <= mlast) ? k.snake_precise(2.f * ue) : v[10];
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) o = fmaf(k.fd[j], v[j], o);
    stf(yo + (size_t)t * C, o);
#pragma unroll
    for (int j = 0; j < 10; ++j) v[j] = v[j + 2];
#pragma unroll
    for (int i = 0; i < 5; ++i) xs[i] = xs[i + 1];
    xs[5] = k.xin(t + 6);
  }
}


// ---- channels-last, LDS-tiled: the form used for every C % 8 == 0 ----------------------------------------------------
// One workgroup owns TT = (256 / CT) * RUN time steps x CT channels (CT = C for the narrow stages, a 64-channel slab
// otherwise).  The TT + 12 input rows arrive with 16-byte loads issued back to back (rows of a narrow stage are
// adjacent in memory, so the tile is one contiguous span), each lane then slides down its channel reading LDS, and the
// outputs leave through LDS as 16-byte row-contiguous stores.  The register-window kernel above pays one dependent
// global load per time step and 48-byte rows at C = 24; this one pays one memory latency per tile.
// FAST: v_sin_f32 (__sinf) as the reference's fast-math CUDA build does (bf16 I/O); fp32 I/O keeps sinf for parity.
template <typename T, bool FAST>
__global__ __launch_bounds__(256) void snake_aa_lds_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ la, const float* __restrict__ lb,
                                                           const float* __restrict__ up12, const float* __restrict__ dn12,
                                                           int Tn, int C, int CT, int tiles_t, int slabs) {
  constexpr int VEC = 16 / (int)sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_snake[];
  const int runs = 256 / CT, TT = runs * RUNL;
  T* sx = reinterpret_cast<T*>(smem_snake);                       // [(TT + 12)][CT]
  T* so = sx + (size_t)(TT + 12) * CT;                            // [TT][CT]
  int bid = blockIdx.x;
  const int slab = bid % slabs;
  bid /= slabs;
  const int tile = bid % tiles_t, b = bid / tiles_t;
  const int t0 = tile * TT, c0 = slab * CT;
  const T* __restrict__ xb = x + (size_t)b * Tn * C + c0;
  T* __restrict__ yb = y + (size_t)b * Tn * C + c0;
  const int tid = threadIdx.x;
  {
    const int vpr = CT / VEC, nvec = (TT + 12) * vpr;
    const int di = 256 / vpr, dq = 256 - di * vpr;  // (row, vector) stepped without a division per vector
    int i = tid / vpr, q = tid - i * vpr;
    for (int v = tid; v < nvec; v += 256, i += di, q += dq) {
      if (q >= vpr) {
        q -= vpr;
        ++i;
      }
      int t = t0 - 6 + i;
      t = t < 0 ? 0 : (t >= Tn ? Tn - 1 : t);  // replicate padding resolved at load time
      *reinterpret_cast<uint4*>(sx + (size_t)i * CT + q * VEC) = *reinterpret_cast<const uint4*>(xb + (size_t)t * C + q * VEC);
    }
  }
  __syncthreads();
  const int c = tid % CT, run = tid / CT;
  if (run < runs) {
    const float ea = expf(la[c0 + c]);
    const float inv_b = 1.f / (expf(lb[c0 + c]) + 1e-9f);
    float fu[12], fd[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      fu[i] = up12[i];
      fd[i] = dn12[i];
    }
    const int ts = t0 + run * RUNL;
    const int te = min(ts + RUNL, Tn);
    // row i of the tile holds x[clamp(t0 - 6 + i)] (the replicate padding was resolved when the tile was loaded)
    if (ts < Tn) snake_run<T, FAST>(sx + c, CT, t0 - 6, ts, te, Tn, ea, inv_b, fu, fd, [&](int t, float o) { stf(so + (t - t0) * CT + c, o); });
  }
  __syncthreads();
  {
    const int vpr = CT / VEC, rows = min(TT, Tn - t0), nvec = rows * vpr;
    const int di = 256 / vpr, dq = 256 - di * vpr;
    int i = tid / vpr, q = tid - i * vpr;
    for (int v = tid; v < nvec; v += 256, i += di, q += dq) {
      if (q >= vpr) {
        q -= vpr;
        ++i;
      }
      *reinterpret_cast<uint4*>(yb + (size_t)(t0 + i) * C + q * VEC) = *reinterpret_cast<const uint4*>(so + (size_t)i * CT + q * VEC);
    }
  }
}

template <typename T, bool FAST>
static int launch_snake_lds(void* y, const void* x, const float* la, const float* lb, const float* up12, const float* dn12,
                            int B, int Tn, int C, hipStream_t s) {
  int CT = C;
  if (C > 64) {
    CT = 64;
    while (C % CT) CT -= 8;
  }
  const int runs = 256 / CT, TT = runs * RUNL;
  const int tiles_t = (Tn + TT - 1) / TT, slabs = C / CT;
  const size_t lds = ((size_t)(TT + 12) * CT + (size_t)TT * CT) * sizeof(T);
  hipLaunchKernelGGL((snake_aa_lds_kernel<T, FAST>), dim3((unsigned)((long)B * tiles_t * slabs)), dim3(256), lds, s, (T*)y,
                     (const T*)x, la, lb, up12, dn12, Tn, C, CT, tiles_t, slabs);
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}


// ---- [B,C,T] layout (the reference's native-op contract): one (b,c) row per blockIdx.y/z, LDS-staged tile ----
constexpr int BCT_TILE = 1024;

template <typename T>
__global__ __launch_bounds__(256) void snake_aa_bct_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                           const float* __restrict__ la, const float* __restrict__ lb,
                                                           const float* __restrict__ up12, const float* __restrict__ dn12,
                                                           int C, int Tn) {
  __shared__ float xs[BCT_TILE + 12];
  __shared__ float vs[2 * BCT_TILE + 12];
  const int c = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * BCT_TILE;
  const int nt = min(BCT_TILE, Tn - t0);
  const T* __restrict__ xr = x + ((size_t)b * C + c) * Tn;
  T* __restrict__ yr = y + ((size_t)b * C + c) * Tn;
  const float ea = expf(la[c]);
  const float inv_b = 1.f / (expf(lb[c]) + 1e-9f);
  float fu[12], fd[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    fu[i] = up12[i];
    fd[i] = dn12[i];
  }
  for (int i = threadIdx.x; i < nt + 12; i += 256) {
    int t = t0 - 6 + i;
    t = t < 0 ? 0 : (t >= Tn ? Tn - 1 : t);
    xs[i] = ldf(xr + t);
  }
  __syncthreads();
  const int mlast = 2 * Tn - 1;
  for (int i = threadIdx.x; i < 2 * nt + 10; i += 256) {
    int m = 2 * t0 - 5 + i;
    m = m < 0 ? 0 : (m > mlast ? mlast : m);
    const int q = m >> 1;
    const int base = q - (t0 - 6);  // xs index of x[q]
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu[2 * r], xs[base + 3 - r], u);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu[2 * r + 1], xs[base + 2 - r], u);
    }
    u *= 2.f;
    const float sn = sinf(u * ea);
    vs[i] = u + inv_b * sn * sn;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt; i += 256) {
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) o = fmaf(fd[j], vs[2 * i + j], o);
    stf(yr + t0 + i, o);
  }
}

}  // namespace

int snake_aa_bct(void* y, const void* x, const float* log_alpha, const float* log_beta, const float* up12,
                 const float* down12, int B, int C, int T, int dt, hipStream_t s) {
  ITTS_REQUIRE(y && x && log_alpha && log_beta && up12 && down12, "snake_aa_bct: null pointer");
  ITTS_REQUIRE(B > 0 && T > 0 && C > 0 && C <= 65535 && B <= 65535, "snake_aa_bct: bad dims");
  dim3 grid((T + BCT_TILE - 1) / BCT_TILE, C, B);
  if (dt == F32)
    hipLaunchKernelGGL(snake_aa_bct_kernel<float>, grid, dim3(256), 0, s, (float*)y, (const float*)x, log_alpha, log_beta,
                       up12, down12, C, T);
  else if (dt == BF16)
    hipLaunchKernelGGL(snake_aa_bct_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)y, (const bf16_t*)x, log_alpha,
                       log_beta, up12, down12, C, T);
  else if (dt == F16)  // the reference dispatches float / half / bf16 (type_shim.h:20-43); fp32 arithmetic inside, as its kernel
    hipLaunchKernelGGL(snake_aa_bct_kernel<f16_t>, grid, dim3(256), 0, s, (f16_t*)y, (const f16_t*)x, log_alpha,
                       log_beta, up12, down12, C, T);
  else {
    set_error("snake_aa_bct: unsupported dtype");
    return E_INVALID;
  }
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

int snake_aa(void* y, const void* x, const float* log_alpha, const float* log_beta, const float* up12,
             const float* down12, int B, int T, int C, int dt, hipStream_t s) {
  ITTS_REQUIRE(y && x && log_alpha && log_beta && up12 && down12, "snake_aa: null pointer");
  ITTS_REQUIRE(B > 0 && T > 0 && C > 0, "snake_aa: bad dims");
  static const bool no_lds = getenv("ITTS_SNAKE_REG") != nullptr;
  if (!no_lds && C % 8 == 0 && C >= 8 && (dt == F32 || dt == BF16) && !(((uintptr_t)x | (uintptr_t)y) & 15)) {
    if (dt == F32) return launch_snake_lds<float, false>(y, x, log_alpha, log_beta, up12, down12, B, T, C, s);
    return launch_snake_lds<bf16_t, true>(y, x, log_alpha, log_beta, up12, down12, B, T, C, s);
  }
  const int nchunk = (T + RUN - 1) / RUN;
  const long total = (long)B * nchunk * C;
  const int blocks = (int)((total + 255) / 256);
  if (dt == F32)
    hipLaunchKernelGGL(snake_aa_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)y, (const float*)x, log_alpha,
                       log_beta, up12, down12, B, T, C, nchunk);
  else if (dt == BF16)
    hipLaunchKernelGGL(snake_aa_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (bf16_t*)y, (const bf16_t*)x, log_alpha,
                       log_beta, up12, down12, B, T, C, nchunk);
  else if (dt == F16)
    hipLaunchKernelGGL(snake_aa_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (f16_t*)y, (const f16_t*)x, log_alpha,
                       log_beta, up12, down12, B, T, C, nchunk);
  else {
    set_error("snake_aa: unsupported dtype");
    return E_INVALID;
  }
  ITTS_HIP_CHECK(hipGetLastError());
  return OK;
}

}  // namespace itts
