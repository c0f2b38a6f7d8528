// Semantics probe: v_permlane16_swap / v_permlane32_swap on gfx950 with both operands = lane id.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* y) {
  const unsigned v = threadIdx.x;
  u2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  u2 q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  y[threadIdx.x] = r.x; y[64 + threadIdx.x] = r.y; y[128 + threadIdx.x] = q.x; y[192 + threadIdx.x] = q.y;
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* n[] = {"p32.x", "p32.y", "p16.x", "p16.y"};
  for (int a = 0; a < 4; ++a) { printf("%s:", n[a]); for (int i = 0; i < 64; ++i) printf(" %u", h[a * 64 + i]); printf("\n"); }
  return 0;
}
