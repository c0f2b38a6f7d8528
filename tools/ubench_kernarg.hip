// How long does a wave wait for its kernel arguments?  t0 = s_memtime at entry, t1 = after the kernarg s_load returned.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/uk tools/ubench_kernarg.hip && /tmp/uk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

struct Big { unsigned long long* out; int pad[30]; };  // 128-byte argument block like GemvArgs

__global__ void probe(Big a, int slot) {
  unsigned long long t0, t1, t2;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  unsigned long long* p = a.out;                       // first use of a kernel argument
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "s"(p));
  unsigned long long v = p[0];                         // first global load (L2 / HBM latency)
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) : "s"(v));
  if (threadIdx.x == 0) {
    p[(size_t)(slot * gridDim.x + blockIdx.x) * 4 + 8] = t0;
    p[(size_t)(slot * gridDim.x + blockIdx.x) * 4 + 9] = t1;
    p[(size_t)(slot * gridDim.x + blockIdx.x) * 4 + 10] = t2;
  }
}

int main() {
  const int REP = 64, blocks = 160;
  unsigned long long* d; CK(hipMalloc(&d, (size_t)(REP * blocks * 4 + 16) * 8)); CK(hipMemset(d, 0, (size_t)(REP * blocks * 4 + 16) * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  Big a{}; a.out = d;
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) {
      for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, a, i);
      CK(hipStreamSynchronize(s));
    } else {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, a, i);
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    }
    std::vector<unsigned long long> h((size_t)REP * blocks * 4 + 16);
    CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ka, gl;
    for (int i = 8; i < REP; ++i) for (int b = 0; b < blocks; ++b) {
      const unsigned long long* e = &h[(size_t)(i * blocks + b) * 4 + 8];
      ka.push_back((double)(e[1] - e[0])); gl.push_back((double)(e[2] - e[1]));
    }
    std::sort(ka.begin(), ka.end()); std::sort(gl.begin(), gl.end());
    printf("%s: kernarg wait median %.0f p10 %.0f p90 %.0f ticks; first global load median %.0f p90 %.0f ticks (s_memtime ticks)\n",
           mode ? "hipGraph" : "stream  ", ka[ka.size() / 2], ka[ka.size() / 10], ka[ka.size() * 9 / 10], gl[gl.size() / 2], gl[gl.size() * 9 / 10]);
  }
  // tick rate
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  return 0;
}
