// What does a graph kernel node cost beyond an empty launch?  Argument block size, dynamic LDS, register footprint.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ul2 tools/ubench_launch2.hip && /tmp/ul2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
struct Big { float* y; int pad[60]; };
__global__ void k_empty() {}
__global__ void k_small(float* y) { if (y == nullptr) y[0] = 1.f; }
__global__ void k_big(Big a) { if (a.y == nullptr) a.y[0] = (float)a.pad[59]; }
__global__ void k_lds(Big a) { extern __shared__ float sm[]; if (a.y == nullptr) a.y[0] = sm[threadIdx.x]; }
__global__ __launch_bounds__(256) void k_regs(Big a) {
  float v[160];
#pragma unroll
  for (int i = 0; i < 160; ++i) v[i] = (float)(a.pad[i % 60] + i + threadIdx.x);
  float s = 0;
#pragma unroll
  for (int i = 0; i < 160; ++i) s += v[i] * v[(i * 7) % 160];
  if (s == 12345.678f) a.y[0] = s;
}
template <typename F> float timeit(F launch, int reps, hipStream_t s) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s)); for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3f / (5 * reps);
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* y; CK(hipMalloc(&y, 1 << 20));
  Big a{}; a.y = y;
  for (int g : {160, 640}) {
    printf("grid %3d x 256: empty %.2f | 8-byte arg %.2f | 248-byte arg %.2f | + 10 KB dynamic LDS %.2f | + 40 KB LDS %.2f | ~160 VGPRs %.2f us\n", g,
           timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(g), dim3(256), 0, s); }, 100, s),
           timeit([&](int) { hipLaunchKernelGGL(k_small, dim3(g), dim3(256), 0, s, y); }, 100, s),
           timeit([&](int) { hipLaunchKernelGGL(k_big, dim3(g), dim3(256), 0, s, a); }, 100, s),
           timeit([&](int) { hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 10240, s, a); }, 100, s),
           timeit([&](int) { hipLaunchKernelGGL(k_lds, dim3(g), dim3(256), 40960, s, a); }, 100, s),
           timeit([&](int) { hipLaunchKernelGGL(k_regs, dim3(g), dim3(256), 0, s, a); }, 100, s));
  }
  printf("grid 160 x 1024: empty %.2f us;  grid 40 x 1024: %.2f us; grid 2 x 1024: %.2f us\n",
         timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(160), dim3(1024), 0, s); }, 100, s),
         timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(40), dim3(1024), 0, s); }, 100, s),
         timeit([&](int) { hipLaunchKernelGGL(k_empty, dim3(2), dim3(1024), 0, s); }, 100, s));
  return 0;
}
