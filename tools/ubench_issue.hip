// Timeline inside a weight-streaming block: how long do 6 / 12 independent 16-byte loads per lane take to ISSUE, and to land?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ui tools/ubench_issue.hip && /tmp/ui
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")

template <int NL, bool NT>
__global__ __launch_bounds__(256) void probe(unsigned long long* __restrict__ ST, const u32x4* __restrict__ W, float* __restrict__ Y, size_t slab) {
  unsigned long long t0, t1, t2, t3;
  STAMP(t0);
  // contiguous per wave-instruction (lane stride 16 B), instruction stride = one 4-KiB row of the block
  const u32x4* p = W + slab + (size_t)blockIdx.x * 256 * NL + threadIdx.x;
  u32x4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = NT ? __builtin_nontemporal_load(p + i * 256) : p[i * 256];
  STAMP(t1);
  { unsigned a = v[0].x; asm volatile("" :: "v"(a)); }
  STAMP(t2);
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) acc += v[i].x ^ v[i].w;
  { asm volatile("" :: "v"(acc)); }
  STAMP(t3);
  if (acc == 0x12345678u) Y[threadIdx.x] = 1.f;
  if (threadIdx.x == 0) {
    unsigned long long* o = ST + (size_t)blockIdx.x * 4;
    o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
  }
}

template <int NL, bool NT>
int run(const char* name, int blocks, hipStream_t s, unsigned long long* ST, const u32x4* W, float* Y, size_t slab_elems) {
  const int REP = 40;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < REP; ++i) hipLaunchKernelGGL((probe<NL, NT>), dim3(blocks), dim3(256), 0, s, ST + (size_t)i * blocks * 4, W, Y, (size_t)i * slab_elems);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h((size_t)REP * blocks * 4);
  CK(hipMemcpy(h.data(), ST, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> d1, d2, d3;
  for (int i = 5; i < REP; ++i) for (int bl = 0; bl < blocks; ++bl) {
    const unsigned long long* e = &h[((size_t)i * blocks + bl) * 4];
    d1.push_back((double)(e[1] - e[0])); d2.push_back((double)(e[2] - e[0])); d3.push_back((double)(e[3] - e[0]));
  }
  std::sort(d1.begin(), d1.end()); std::sort(d2.begin(), d2.end()); std::sort(d3.begin(), d3.end());
  auto q = [](std::vector<double>& v, double f) { return v[(size_t)(v.size() * f)]; };
  printf("%-28s blocks %4d  %.2f us/launch | issued %5.0f (p90 %5.0f)  first landed %5.0f (p90 %5.0f)  all landed %5.0f (p90 %5.0f) ticks\n", name, blocks,
         ms * 1e3 / REP, q(d1, .5), q(d1, .9), q(d2, .5), q(d2, .9), q(d3, .5), q(d3, .9));
  return 0;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t slab = (size_t)16 << 20;  // bytes per launch (distinct region per launch: cold)
  u32x4* W; CK(hipMalloc(&W, slab * 40)); CK(hipMemset(W, 1, slab * 40));
  unsigned long long* ST; CK(hipMalloc(&ST, (size_t)40 * 2048 * 4 * 8));
  float* Y; CK(hipMalloc(&Y, 4096));
  run<6, true>("6 loads/lane nt, 9.8 MB", 640, s, ST, W, Y, slab / 16);
  run<6, false>("6 loads/lane, 9.8 MB", 640, s, ST, W, Y, slab / 16);
  run<12, true>("12 loads/lane nt, 9.8 MB", 320, s, ST, W, Y, slab / 16);
  run<3, true>("3 loads/lane nt, 9.8 MB", 1280, s, ST, W, Y, slab / 16);
  run<6, true>("6 loads/lane nt, 2.5 MB", 160, s, ST, W, Y, slab / 16);
  run<1, true>("1 load/lane nt, 1.6 MB", 640, s, ST, W, Y, slab / 16);
  return 0;
}
