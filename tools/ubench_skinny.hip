// Dissection of the batched-decode skinny GEMM (decode_mfma.hip) on MI355X: which stream costs what.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/us tools/ubench_skinny.hip && /tmp/us
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int SW = 8;

// XMODE 0: no X loads (constant), 1: row-major [B][K], 2: fragment-tiled [K/32][BT][64 lanes][8]
// WMODE 0: no W loads, 1: nontemporal row-major [N][K], 2: plain loads, 3: nontemporal fragment-tiled [N/16][K/32][64 lanes][8]
template <int BT, int NT, int SU, int XMODE, int WMODE, int WAVES, bool XFIRST = false>
__global__ __launch_bounds__(WAVES * 64) void k_skinny(float* __restrict__ Y, const unsigned short* __restrict__ X,
                                                       const unsigned short* __restrict__ W, int B, int N, int K) {
  __shared__ float red[WAVES][NT * BT][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * (16 * NT);
  const int nks_all = K >> 5, S = gridDim.y, split = blockIdx.y, per = (nks_all + S - 1) / S;
  const int ks0 = split * per, nks = min(per, nks_all - ks0);
  const unsigned short* wp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    wp[t] = WMODE == 3 ? W + (((size_t)min(n0 / 16 + t, N / 16 - 1) * nks_all + ks0) * 64 + lane) * 8
                       : W + (size_t)min(n0 + t * 16 + fr, N - 1) * K + fg * 8 + (size_t)ks0 * 32;
  const unsigned short* xp[BT];
#pragma unroll
  for (int bt = 0; bt < BT; ++bt)
    xp[bt] = XMODE == 2 ? X + ((size_t)ks0 * BT + bt) * 512 + lane * 8 : X + (size_t)min(bt * 16 + fr, B - 1) * K + fg * 8 + (size_t)ks0 * 32;
  f32x4v acc[NT][BT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) acc[t][bt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (int k0 = wave; k0 < nks; k0 += WAVES * SU) {
    bf16x8 wf[SU][NT], xf[SU][BT];
    if (XFIRST) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = min(k0 + u * WAVES, nks - 1);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        if (XMODE == 1) xf[u][bt] = *reinterpret_cast<const bf16x8*>(xp[bt] + (size_t)ks * 32);
        else if (XMODE == 2) xf[u][bt] = *reinterpret_cast<const bf16x8*>(xp[bt] + (size_t)ks * BT * 512);
        else xf[u][bt] = bf16x8{(short)ks, 1, 2, 3, 4, 5, 6, (short)lane};
      }
    }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = min(k0 + u * WAVES, nks - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (WMODE == 3) wf[u][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * 512));
        else if (WMODE == 1) wf[u][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * 32));
        else if (WMODE == 2) wf[u][t] = *reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * 32);
        else wf[u][t] = bf16x8{(short)ks, 1, 2, 3, 4, 5, 6, (short)lane};
      }
    }
    } else {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = min(k0 + u * WAVES, nks - 1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (WMODE == 3) wf[u][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * 512));
        else if (WMODE == 1) wf[u][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * 32));
        else if (WMODE == 2) wf[u][t] = *reinterpret_cast<const bf16x8*>(wp[t] + (size_t)ks * 32);
        else wf[u][t] = bf16x8{(short)ks, 1, 2, 3, 4, 5, 6, (short)lane};
      }
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int ks = min(k0 + u * WAVES, nks - 1);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        if (XMODE == 1) xf[u][bt] = *reinterpret_cast<const bf16x8*>(xp[bt] + (size_t)ks * 32);
        else if (XMODE == 2) xf[u][bt] = *reinterpret_cast<const bf16x8*>(xp[bt] + (size_t)ks * BT * 512);
        else xf[u][bt] = bf16x8{(short)ks, 1, 2, 3, 4, 5, 6, (short)lane};
      }
    }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < SU; ++u)
      if (k0 + u * WAVES < nks)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int bt = 0; bt < BT; ++bt) acc[t][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[u][bt], wf[u][t], acc[t][bt], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][t * BT + bt][r * 64 + lane] = acc[t][bt][r];
  __syncthreads();
  for (int idx = tid; idx < NT * BT * 256; idx += WAVES * 64) {
    const int tb = idx >> 8, t = tb / BT, bt = tb - t * BT, e = idx & 255, r = e >> 6, l = e & 63;
    const int b = bt * 16 + (l >> 4) * 4 + r, n = n0 + t * 16 + (l & 15);
    if (b >= B || n >= N) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) v += red[w][tb][e];
    Y[((size_t)split * B + b) * N + n] = v;
  }
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
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3f / (5 * reps);
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int REP = 48, B = 64;
  const size_t slab = (size_t)5120 * 1280;
  unsigned short *W, *X; float* Y;
  CK(hipMalloc(&W, REP * slab * 2)); CK(hipMemset(W, 0x3c, REP * slab * 2));
  CK(hipMalloc(&X, (size_t)128 * 5120 * 2)); CK(hipMemset(X, 0x3c, (size_t)128 * 5120 * 2));
  CK(hipMalloc(&Y, (size_t)8 * 128 * 8194 * 4));
#define RUN(name, BT, NT, SU, XM, WM, WV, N, K, S)                                                                        \
  {                                                                                                                      \
    const int tiles = (N + 16 * NT - 1) / (16 * NT);                                                                     \
    float us = timeit([&](int i) { hipLaunchKernelGGL((k_skinny<BT, NT, SU, XM, WM, WV>), dim3(tiles, S), dim3(WV * 64), 0, s, Y, X, \
                                                      W + (size_t)i * slab, B, N, K); }, REP, s);                        \
    printf("%-52s N=%4d K=%4d S=%d blocks=%4d  %.2f us\n", name, N, K, S, tiles * S, us);                                 \
  }
  RUN("qkv  cur (X rowmajor, W nt)", 4, 1, 5, 1, 1, 8, 3840, 1280, 1)
  RUN("qkv  no X", 4, 1, 5, 0, 1, 8, 3840, 1280, 1)
  RUN("qkv  no W", 4, 1, 5, 1, 0, 8, 3840, 1280, 1)
  RUN("qkv  no X no W", 4, 1, 5, 0, 0, 8, 3840, 1280, 1)
  RUN("qkv  X tiled", 4, 1, 5, 2, 1, 8, 3840, 1280, 1)
  RUN("qkv  X tiled, W plain", 4, 1, 5, 2, 2, 8, 3840, 1280, 1)
  RUN("qkv  X tiled NT=2 (120 blocks)", 4, 2, 5, 2, 1, 8, 3840, 1280, 1)
  RUN("qkv  4 waves SU=10 X tiled", 4, 1, 10, 2, 1, 4, 3840, 1280, 1)
  RUN("qkv  4 waves SU=5 X tiled", 4, 1, 5, 2, 1, 4, 3840, 1280, 1)
  RUN("qkv  16 waves SU=3 X tiled", 4, 1, 3, 2, 1, 16, 3840, 1280, 1)
#define RUNX(name, BT, NT, SU, XM, WM, WV, N, K, S)                                                                       \
  {                                                                                                                      \
    const int tiles = (N + 16 * NT - 1) / (16 * NT);                                                                     \
    float us = timeit([&](int i) { hipLaunchKernelGGL((k_skinny<BT, NT, SU, XM, WM, WV, true>), dim3(tiles, S), dim3(WV * 64), 0, s, Y, X, \
                                                      W + (size_t)i * slab, B, N, K); }, REP, s);                        \
    printf("%-52s N=%4d K=%4d S=%d blocks=%4d  %.2f us\n", name, N, K, S, tiles * S, us);                                 \
  }
  RUNX("qkv  X tiled, X loads issued first", 4, 1, 5, 2, 1, 8, 3840, 1280, 1)
  RUNX("fc   NT=2 X tiled, X first", 4, 2, 5, 2, 1, 8, 5120, 1280, 1)
  RUNX("proj2 NT=2 S=4 X tiled, X first", 4, 2, 5, 2, 1, 8, 1280, 5120, 4)
  RUNX("proj  NT=1 S=2 X tiled, X first", 4, 1, 5, 2, 1, 8, 1280, 1280, 2)
  // all 256 CUs, small X per workgroup: wide feature tiles x deep K split (partials)
  RUN("fc   NT=4 S=3 (240 WGs) 8 waves", 4, 4, 2, 2, 1, 8, 5120, 1280, 3)
  RUN("fc   NT=5 S=4 (256 WGs) 8 waves", 4, 5, 2, 2, 1, 8, 5120, 1280, 4)
  RUN("fc   NT=5 S=4 (256 WGs) 4 waves", 4, 5, 3, 2, 1, 4, 5120, 1280, 4)
  RUN("fc   NT=2 S=2 (320 WGs) 8 waves", 4, 2, 3, 2, 1, 8, 5120, 1280, 2)
  RUN("fc   NT=4 S=4 (320 WGs) 4 waves", 4, 4, 3, 2, 1, 4, 5120, 1280, 4)
  RUN("qkv  NT=3 S=3 (240 WGs) 8 waves", 4, 3, 2, 2, 1, 8, 3840, 1280, 3)
  RUN("qkv  NT=5 S=5 (240 WGs) 8 waves", 4, 5, 1, 2, 1, 8, 3840, 1280, 5)
  RUN("qkv  NT=5 S=5 (240 WGs) 4 waves", 4, 5, 2, 2, 1, 4, 3840, 1280, 5)
  RUN("qkv  NT=2 S=2 (240 WGs) 8 waves", 4, 2, 3, 2, 1, 8, 3840, 1280, 2)
  RUN("proj2 NT=5 S=16 (256 WGs) 8 waves", 4, 5, 2, 2, 1, 8, 1280, 5120, 16)
  RUN("proj2 NT=5 S=16 (256 WGs) 4 waves", 4, 5, 3, 2, 1, 4, 1280, 5120, 16)
  RUN("proj2 NT=4 S=12 (240 WGs) 8 waves", 4, 4, 2, 2, 1, 8, 1280, 5120, 12)
  RUN("proj2 NT=2 S=6 (240 WGs) 8 waves", 4, 2, 4, 2, 1, 8, 1280, 5120, 6)
  RUN("proj  NT=2 S=6 (240 WGs) 8 waves", 4, 2, 1, 2, 1, 8, 1280, 1280, 6)
  RUN("proj  NT=5 S=8 (128 WGs) 4 waves", 4, 5, 2, 2, 1, 4, 1280, 1280, 8)
  RUN("proj  NT=2 S=4 (160 WGs) 8 waves", 4, 2, 2, 2, 1, 8, 1280, 1280, 4)
  // W in fragment tiles too (every wave-load 1 KiB contiguous)
  RUN("qkv  X tiled, W tiled", 4, 1, 5, 2, 3, 8, 3840, 1280, 1)
  RUN("qkv  no X, W tiled", 4, 1, 5, 0, 3, 8, 3840, 1280, 1)
  RUN("fc   NT=2 X tiled, W tiled", 4, 2, 5, 2, 3, 8, 5120, 1280, 1)
  RUN("fc   NT=2 no X, W tiled", 4, 2, 5, 0, 3, 8, 5120, 1280, 1)
  RUN("fc   NT=1 X tiled, W tiled", 4, 1, 5, 2, 3, 8, 5120, 1280, 1)
  RUN("fc   NT=4 S=3 W tiled", 4, 4, 2, 2, 3, 8, 5120, 1280, 3)
  RUN("proj2 NT=2 S=4 X tiled, W tiled", 4, 2, 5, 2, 3, 8, 1280, 5120, 4)
  RUN("proj2 NT=2 S=6 W tiled", 4, 2, 4, 2, 3, 8, 1280, 5120, 6)
  RUN("proj  NT=1 S=2 X tiled, W tiled", 4, 1, 5, 2, 3, 8, 1280, 1280, 2)
  RUN("proj  NT=2 S=6 W tiled", 4, 2, 1, 2, 3, 8, 1280, 1280, 6)
  RUN("fc   cur NT=2", 4, 2, 5, 1, 1, 8, 5120, 1280, 1)
  RUN("fc   NT=2 X tiled", 4, 2, 5, 2, 1, 8, 5120, 1280, 1)
  RUN("fc   NT=1 X tiled", 4, 1, 5, 2, 1, 8, 5120, 1280, 1)
  RUN("fc   NT=2 no X", 4, 2, 5, 0, 1, 8, 5120, 1280, 1)
  RUN("proj2 cur NT=2 S=4", 4, 2, 5, 1, 1, 8, 1280, 5120, 4)
  RUN("proj2 NT=2 S=4 X tiled", 4, 2, 5, 2, 1, 8, 1280, 5120, 4)
  RUN("proj2 NT=1 S=4 X tiled", 4, 1, 5, 2, 1, 8, 1280, 5120, 4)
  RUN("proj2 NT=1 S=2 X tiled", 4, 1, 5, 2, 1, 8, 1280, 5120, 2)
  RUN("proj2 NT=2 S=8 X tiled", 4, 2, 5, 2, 1, 8, 1280, 5120, 8)
  RUN("proj  cur NT=1 S=2", 4, 1, 5, 1, 1, 8, 1280, 1280, 2)
  RUN("proj  NT=1 S=2 X tiled", 4, 1, 5, 2, 1, 8, 1280, 1280, 2)
  RUN("proj  NT=1 S=4 X tiled", 4, 1, 5, 2, 1, 8, 1280, 1280, 4)
  RUN("proj  NT=1 S=1 X tiled", 4, 1, 5, 2, 1, 8, 1280, 1280, 1)
  return 0;
}
