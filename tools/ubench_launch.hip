// Kernel-boundary cost inside a hipGraph vs grid shape (MI355X).  hipcc --offload-arch=gfx950 -O3 -o /tmp/ul tools/ubench_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void k_null(float* y) { if (threadIdx.x == 0 && blockIdx.x == 0) y[0] += 1.f; }
__global__ void k_touch(float* y, const float* x, int n) {  // every block reads 4 KB then writes one float
  float s = 0; for (int i = threadIdx.x; i < 1024; i += blockDim.x) s += x[(blockIdx.x * 1024 + i) % n];
  if (s == 123.f) y[blockIdx.x] = s;
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
  float *y, *x; CK(hipMalloc(&y, 1 << 20)); CK(hipMalloc(&x, 64 << 20)); CK(hipMemset(x, 0, 64 << 20)); CK(hipMemset(y, 0, 1 << 20));
  int grids[] = {1, 20, 40, 120, 240, 480, 960, 2048};
  int wgs[] = {64, 256, 1024};
  for (int w : wgs) for (int g : grids)
    printf("null   grid=%4d wg=%4d : %.2f us\n", g, w, timeit([&](int) { hipLaunchKernelGGL(k_null, dim3(g), dim3(w), 0, s, y); }, 100, s));
  for (int w : wgs) for (int g : grids)
    printf("touch  grid=%4d wg=%4d : %.2f us\n", g, w, timeit([&](int) { hipLaunchKernelGGL(k_touch, dim3(g), dim3(w), 0, s, y, x, 16 << 20); }, 100, s));
  return 0;
}
