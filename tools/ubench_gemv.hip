// Standalone dissection of the decode GEMV fixed costs on MI355X (build + run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ub tools/ubench_gemv.hip && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float bf(unsigned w, int hi) { return __uint_as_float(hi ? (w & 0xFFFF0000u) : (w << 16)); }

__device__ int g_xstride = 0;
__global__ void k_set(int v) { g_xstride = v; }
__global__ void k_null(float* y) { if (threadIdx.x == 0 && blockIdx.x == 0) y[0] = 1.f; }

// MODE 0: weight loads + reduce only; 1: + X via LDS (plain); 2: + LayerNorm; NT: nontemporal loads
template <int RPW, int NCH, int NB, int MODE, bool NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gemv(float* __restrict__ Y, const float* __restrict__ X,
                                                     const unsigned short* __restrict__ W, const float* __restrict__ gam,
                                                     const float* __restrict__ bet, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float sx[];
  __shared__ float red[WAVES][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT_ = WAVES * 64;
  const int n0 = (blockIdx.x * WAVES + wave) * RPW;
  u32x4 w[RPW][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < K)
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const u32x4* p = reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + r, N - 1) * K + k);
        w[r][c] = NT ? __builtin_nontemporal_load(p) : *p;
      }
  }
  if (MODE >= 1) {
    for (int i = tid * 4; i < NB * K; i += NT_ * 4) {
      float4 v = *reinterpret_cast<const float4*>(X + i);
      *reinterpret_cast<float4*>(sx + i) = v;
    }
    __syncthreads();
    if (MODE >= 2) {
      float s[NB], q[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        s[b] = q[b] = 0.f;
        for (int i = tid; i < K; i += NT_) { const float d = sx[b * K + i]; s[b] += d; q[b] = fmaf(d, d, q[b]); }
        s[b] = wsum(s[b]); q[b] = wsum(q[b]);
        if (lane == 0) { red[wave][2 * b] = s[b]; red[wave][2 * b + 1] = q[b]; }
      }
      __syncthreads();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float S = 0, Q = 0;
        for (int ww = 0; ww < WAVES; ++ww) { S += red[ww][2 * b]; Q += red[ww][2 * b + 1]; }
        const float m = S / K, r = rsqrtf(fmaxf(Q / K - m * m, 0.f) + 1e-5f);
        for (int i = tid; i < K; i += NT_) sx[b * K + i] = (sx[b * K + i] - m) * r * gam[i] + bet[i];
      }
      __syncthreads();
    }
  }
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
        float x[8];
        if (MODE >= 1) {
          const float4 a = *reinterpret_cast<const float4*>(sx + b * K + k), bb = *reinterpret_cast<const float4*>(sx + b * K + k + 4);
          x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = bb.x; x[5] = bb.y; x[6] = bb.z; x[7] = bb.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = 1.f + b;
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[r][b] = fmaf(x[i], bf(w[r][c][i >> 1], i & 1), acc[r][b]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wsum(acc[r][b]);
  if (lane == 0)
#pragma unroll
    for (int r = 0; r < RPW; ++r)
      if (n0 + r < N)
#pragma unroll
        for (int b = 0; b < NB; ++b) Y[(size_t)b * N + n0 + r] = acc[r][b];
}


// MODE 3: activations + LN params requested FIRST, weights second (vmcnt is in-order: LN overlaps the weight stream)
template <int RPW, int NCH, int NB, bool LN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gemv3(float* __restrict__ Y, const float* __restrict__ X,
                                                      const unsigned short* __restrict__ W, const float* __restrict__ gam,
                                                      const float* __restrict__ bet, int N, int K) {
  constexpr int NT_ = WAVES * 64;
  constexpr int XCH = (NB * NCH * 512 + NT_ * 4 - 1) / (NT_ * 4);
  extern __shared__ __attribute__((aligned(16))) float sx[];
  __shared__ float red[WAVES][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (blockIdx.x * WAVES + wave) * RPW, BK = NB * K;
  X += (size_t)blockIdx.x * g_xstride;
  float4 x[XCH], gm[XCH], bt[XCH];
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int i = tid * 4 + j * NT_ * 4;
    if (i < BK) {
      x[j] = *reinterpret_cast<const float4*>(X + i);
      if (LN) { const int col = i % K; gm[j] = *reinterpret_cast<const float4*>(gam + col); bt[j] = *reinterpret_cast<const float4*>(bet + col); }
    }
  }
  float pivot[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) pivot[b] = LN ? X[(size_t)b * K] : 0.f;
  u32x4 w[RPW][NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 512 + lane * 8;
    if (k < K)
#pragma unroll
      for (int r = 0; r < RPW; ++r) w[r][c] = *reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + r, N - 1) * K + k);
  }
  if (LN) {
    float s[NB], q[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = q[b] = 0.f;
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const int i = tid * 4 + j * NT_ * 4;
      if (i < BK) {
        const int b = i / K;
        const float v[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) if (bb == b)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float d = v[e] - pivot[bb]; s[bb] += d; q[bb] = fmaf(d, d, q[bb]); }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) { s[b] = wsum(s[b]); q[b] = wsum(q[b]); if (lane == 0) { red[wave][2 * b] = s[b]; red[wave][2 * b + 1] = q[b]; } }
    __syncthreads();
    float mean[NB], rstd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float S = 0, Q = 0;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) { S += red[ww][2 * b]; Q += red[ww][2 * b + 1]; }
      const float md = S / K; mean[b] = pivot[b] + md; rstd[b] = rsqrtf(fmaxf(Q / K - md * md, 0.f) + 1e-5f);
    }
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const int i = tid * 4 + j * NT_ * 4;
      if (i < BK) {
        const int b = i / K; float m = 0, r = 1;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) if (bb == b) { m = mean[bb]; r = rstd[bb]; }
        x[j].x = (x[j].x - m) * r * gm[j].x + bt[j].x; x[j].y = (x[j].y - m) * r * gm[j].y + bt[j].y;
        x[j].z = (x[j].z - m) * r * gm[j].z + bt[j].z; x[j].w = (x[j].w - m) * r * gm[j].w + bt[j].w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < XCH; ++j) { const int i = tid * 4 + j * NT_ * 4; if (i < BK) *reinterpret_cast<float4*>(sx + i) = x[j]; }
  __syncthreads();
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
        const float4 a = *reinterpret_cast<const float4*>(sx + b * K + k), bb = *reinterpret_cast<const float4*>(sx + b * K + k + 4);
        const float xv[8] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[r][b] = fmaf(xv[i], bf(w[r][c][i >> 1], i & 1), acc[r][b]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wsum(acc[r][b]);
  if (lane == 0)
#pragma unroll
    for (int r = 0; r < RPW; ++r)
      if (n0 + r < N)
#pragma unroll
        for (int b = 0; b < NB; ++b) Y[(size_t)b * N + n0 + r] = acc[r][b];
}

// ---- MODE 4: branch-free (clamped addresses, masks), DPP reductions ----
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wsum_dpp(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror  -> every lane holds its 16-lane row sum
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <int RPW, int NCH, int NB, bool LN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gemv4(float* __restrict__ Y, const float* __restrict__ X,
                                                      const unsigned short* __restrict__ W, const float* __restrict__ gam,
                                                      const float* __restrict__ bet, int N, int K) {
  constexpr int NT_ = WAVES * 64;
  constexpr int XCH = (NB * NCH * 512 + NT_ * 4 - 1) / (NT_ * 4);
  extern __shared__ __attribute__((aligned(16))) float sx[];
  __shared__ float red[WAVES][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (blockIdx.x * WAVES + wave) * RPW, BK = NB * K;
  float4 x[XCH], gm[XCH], bt[XCH];
  int xb[XCH];
  bool xok[XCH];
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int i = tid * 4 + j * NT_ * 4;
    xok[j] = i < BK;
    const int ic = xok[j] ? i : BK - 4;
    xb[j] = ic / K;
    x[j] = *reinterpret_cast<const float4*>(X + ic);
    if (LN) { const int col = ic - xb[j] * K; gm[j] = *reinterpret_cast<const float4*>(gam + col); bt[j] = *reinterpret_cast<const float4*>(bet + col); }
  }
  float pivot[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) pivot[b] = LN ? X[(size_t)b * K] : 0.f;
  u32x4 w[RPW][NCH];
  const int klast = (NCH - 1) * 512 + lane * 8;
  const bool kok = klast < K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
#pragma unroll
    for (int r = 0; r < RPW; ++r) w[r][c] = *reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + r, N - 1) * K + k);
  }
  if (LN) {
    float s[NB], q[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = q[b] = 0.f;
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const float v[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const bool m = xok[j] && xb[j] == bb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = m ? v[e] - pivot[bb] : 0.f; s[bb] += d; q[bb] = fmaf(d, d, q[bb]); }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) { s[b] = wsum_dpp(s[b]); q[b] = wsum_dpp(q[b]); }
    if (lane == 0)
#pragma unroll
      for (int b = 0; b < NB; ++b) { red[wave][2 * b] = s[b]; red[wave][2 * b + 1] = q[b]; }
    __syncthreads();
    float mean[NB], rstd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float S = 0, Q = 0;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) { S += red[ww][2 * b]; Q += red[ww][2 * b + 1]; }
      const float md = S / K; mean[b] = pivot[b] + md; rstd[b] = rsqrtf(fmaxf(Q / K - md * md, 0.f) + 1e-5f);
    }
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      float m = mean[0], r = rstd[0];
#pragma unroll
      for (int bb = 1; bb < NB; ++bb) { m = xb[j] == bb ? mean[bb] : m; r = xb[j] == bb ? rstd[bb] : r; }
      x[j].x = (x[j].x - m) * r * gm[j].x + bt[j].x; x[j].y = (x[j].y - m) * r * gm[j].y + bt[j].y;
      x[j].z = (x[j].z - m) * r * gm[j].z + bt[j].z; x[j].w = (x[j].w - m) * r * gm[j].w + bt[j].w;
    }
  }
#pragma unroll
  for (int j = 0; j < XCH; ++j) { const int i = tid * 4 + j * NT_ * 4; if (xok[j]) *reinterpret_cast<float4*>(sx + i) = x[j]; }
  __syncthreads();
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
      const float4 a = *reinterpret_cast<const float4*>(sx + b * K + k), bb = *reinterpret_cast<const float4*>(sx + b * K + k + 4);
      float xv[8] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w};
      if (c == NCH - 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = kok ? xv[i] : 0.f;
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r][b] = fmaf(xv[i], bf(w[r][c][i >> 1], i & 1), acc[r][b]);
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wsum_dpp(acc[r][b]);
  if (lane == 0)
#pragma unroll
    for (int r = 0; r < RPW; ++r)
      if (n0 + r < N)
#pragma unroll
        for (int b = 0; b < NB; ++b) Y[(size_t)b * N + n0 + r] = acc[r][b];
}

template <int RPW, int NCH, int NB, bool LN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_gemv4s(unsigned long long* __restrict__ STAMP, float* __restrict__ Y, const float* __restrict__ X,
                                                      const unsigned short* __restrict__ W, const float* __restrict__ gam,
                                                      const float* __restrict__ bet, int N, int K) {
  constexpr int NT_ = WAVES * 64;
  constexpr int XCH = (NB * NCH * 512 + NT_ * 4 - 1) / (NT_ * 4);
  extern __shared__ __attribute__((aligned(16))) float sx[];
  __shared__ float red[WAVES][2 * NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long st[8];
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[0] = t; }
  const int n0 = (blockIdx.x * WAVES + wave) * RPW, BK = NB * K;
  float4 x[XCH], gm[XCH], bt[XCH];
  int xb[XCH];
  bool xok[XCH];
#pragma unroll
  for (int j = 0; j < XCH; ++j) {
    const int i = tid * 4 + j * NT_ * 4;
    xok[j] = i < BK;
    const int ic = xok[j] ? i : BK - 4;
    xb[j] = ic / K;
    x[j] = *reinterpret_cast<const float4*>(X + ic);
    if (LN) { const int col = ic - xb[j] * K; gm[j] = *reinterpret_cast<const float4*>(gam + col); bt[j] = *reinterpret_cast<const float4*>(bet + col); }
  }
  float pivot[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) pivot[b] = LN ? X[(size_t)b * K] : 0.f;
  u32x4 w[RPW][NCH];
  const int klast = (NCH - 1) * 512 + lane * 8;
  const bool kok = klast < K;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c == NCH - 1 ? (kok ? klast : K - 8) : c * 512 + lane * 8;
#pragma unroll
    for (int r = 0; r < RPW; ++r) w[r][c] = *reinterpret_cast<const u32x4*>(W + (size_t)min(n0 + r, N - 1) * K + k);
  }
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[1] = t; }
  { float probe = x[0].x; asm volatile("" :: "v"(probe)); }
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[2] = t; }
  if (LN) {
    float s[NB], q[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = q[b] = 0.f;
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      const float v[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
      for (int bb = 0; bb < NB; ++bb) {
        const bool m = xok[j] && xb[j] == bb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = m ? v[e] - pivot[bb] : 0.f; s[bb] += d; q[bb] = fmaf(d, d, q[bb]); }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) { s[b] = wsum_dpp(s[b]); q[b] = wsum_dpp(q[b]); }
    if (lane == 0)
#pragma unroll
      for (int b = 0; b < NB; ++b) { red[wave][2 * b] = s[b]; red[wave][2 * b + 1] = q[b]; }
    __syncthreads();
    float mean[NB], rstd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float S = 0, Q = 0;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) { S += red[ww][2 * b]; Q += red[ww][2 * b + 1]; }
      const float md = S / K; mean[b] = pivot[b] + md; rstd[b] = rsqrtf(fmaxf(Q / K - md * md, 0.f) + 1e-5f);
    }
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
      float m = mean[0], r = rstd[0];
#pragma unroll
      for (int bb = 1; bb < NB; ++bb) { m = xb[j] == bb ? mean[bb] : m; r = xb[j] == bb ? rstd[bb] : r; }
      x[j].x = (x[j].x - m) * r * gm[j].x + bt[j].x; x[j].y = (x[j].y - m) * r * gm[j].y + bt[j].y;
      x[j].z = (x[j].z - m) * r * gm[j].z + bt[j].z; x[j].w = (x[j].w - m) * r * gm[j].w + bt[j].w;
    }
  }
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[3] = t; }
#pragma unroll
  for (int j = 0; j < XCH; ++j) { const int i = tid * 4 + j * NT_ * 4; if (xok[j]) *reinterpret_cast<float4*>(sx + i) = x[j]; }
  __syncthreads();
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[4] = t; }

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
      const float4 a = *reinterpret_cast<const float4*>(sx + b * K + k), bb = *reinterpret_cast<const float4*>(sx + b * K + k + 4);
      float xv[8] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w};
      if (c == NCH - 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = kok ? xv[i] : 0.f;
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r][b] = fmaf(xv[i], bf(w[r][c][i >> 1], i & 1), acc[r][b]);
    }
  }
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[5] = t; }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = wsum_dpp(acc[r][b]);
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[6] = t; }

  if (lane == 0)
#pragma unroll
    for (int r = 0; r < RPW; ++r)
      if (n0 + r < N)
#pragma unroll
        for (int b = 0; b < NB; ++b) Y[(size_t)b * N + n0 + r] = acc[r][b];
  { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); if (lane == 0) st[7] = t; }
  if (lane == 0 && wave == 0) for (int i = 0; i < 8; ++i) STAMP[(size_t)blockIdx.x * 8 + i] = st[i];
}

template <typename F>
float timeit(F launch, int reps, hipStream_t s) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / (5 * reps);
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 3840, K = 1280, REP = 60, blocks = N / 8;
  const size_t slab = (size_t)N * K;
  unsigned short* W; CK(hipMalloc(&W, REP * slab * 2)); CK(hipMemset(W, 0x3c, REP * slab * 2));
  float *X, *Y, *g, *b; unsigned long long* ST;
  CK(hipMalloc(&X, 4 * 5120 * 4)); CK(hipMalloc(&Y, (size_t)4 * 8194 * 4 * REP)); CK(hipMalloc(&g, 5120 * 4)); CK(hipMalloc(&b, 5120 * 4));
  CK(hipMalloc(&ST, (size_t)REP * blocks * 8 * 8));
  CK(hipMemset(X, 0, 4 * 5120 * 4)); CK(hipMemset(g, 0, 5120 * 4)); CK(hipMemset(b, 0, 5120 * 4));
  hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s, 0); CK(hipStreamSynchronize(s));
  float us = timeit([&](int i) { hipLaunchKernelGGL((k_gemv4s<2, 3, 2, true, 4>), dim3(blocks), dim3(256), 2 * K * 4, s, ST + (size_t)i * blocks * 8,
                                 Y + (size_t)i * 4 * 8194, X, W + (size_t)i * slab, g, b, N, K); }, REP, s);
  printf("stamped kernel %.2f us/launch\n", us);
  std::vector<unsigned long long> h((size_t)REP * blocks * 8);
  CK(hipMemcpy(h.data(), ST, h.size() * 8, hipMemcpyDeviceToHost));
  const char* names[] = {"start", "loads issued", "X arrived", "LN done", "LDS+barrier", "dot done (W arrived)", "reduced", "end"};
  // s_memtime bases differ per XCD: only differences inside one block / one XCD group (blockIdx % 8) are meaningful
  for (int ph = 1; ph < 8; ++ph) {
    std::vector<double> v;
    for (int r = 10; r < REP; ++r)
      for (int bl = 0; bl < blocks; ++bl) v.push_back((double)(h[((size_t)r * blocks + bl) * 8 + ph] - h[((size_t)r * blocks + bl) * 8]));
    std::sort(v.begin(), v.end());
    printf("%-22s since own start: median %6.0f  p10 %6.0f  p90 %6.0f ticks\n", names[ph], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
  }
  for (int x = 0; x < 8; x += 7) {
    std::vector<double> skew, span, gap;
    for (int r = 10; r < REP; ++r) {
      unsigned long long t0 = ~0ull, t1 = 0, e1 = 0, ep = 0;
      for (int bl = x; bl < blocks; bl += 8) {
        t0 = std::min(t0, h[((size_t)r * blocks + bl) * 8]); t1 = std::max(t1, h[((size_t)r * blocks + bl) * 8]);
        e1 = std::max(e1, h[((size_t)r * blocks + bl) * 8 + 7]); ep = std::max(ep, h[((size_t)(r - 1) * blocks + bl) * 8 + 7]);
      }
      skew.push_back((double)(t1 - t0)); span.push_back((double)(e1 - t0)); gap.push_back((double)t0 - (double)ep);
    }
    std::sort(skew.begin(), skew.end()); std::sort(span.begin(), span.end()); std::sort(gap.begin(), gap.end());
    printf("XCD group %d: block start skew %6.0f, first start -> last end %6.0f, previous kernel last end -> first start %6.0f ticks\n", x,
           skew[skew.size() / 2], span[span.size() / 2], gap[gap.size() / 2]);
  }
  return 0;
}
