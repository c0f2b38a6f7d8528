// Where does the LDS-tiled snake kernel spend its time?  Build with -DSNAKE_EXPERIMENT=0/1/2/3:
//   0 full kernel, 2 no warm-up (10 v_at)   (1 = copy through LDS only and 3 = no sin were measured on the first version
//   of the kernel: copy floor 90 us, warm-up 38 us, sin 16 us of 227 us at C = 24)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iindex-tts-ipex_amd/csrc -Iinclude -DSNAKE_EXPERIMENT=0 -o /tmp/sn0 tools/ubench_snake.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
namespace itts { void set_error(const std::string&) {} const char* last_error() { return ""; } }
#include "../index-tts-ipex_amd/csrc/snake.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
int main() {
  const int B = 8, T = 491520, C = 24;
  const size_t n = (size_t)B * T * C;
  void *x, *y; float *a, *b, *f;
  CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMemset(x, 0x3c, n * 2));
  CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&f, 64)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(b, 0, 4096)); CK(hipMemset(f, 0, 64));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 10; ++i) itts::snake_aa(y, x, a, b, f, f, B, T, C, itts::BF16, s);
    CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("experiment %d: %.1f us per launch, %.2f TB/s r+w\n", SNAKE_EXPERIMENT, ms * 100, n * 4 / (ms * 1e-4) / 1e12 * 1e-6 * 1e6 / 1e6);
  }
  return 0;
}
