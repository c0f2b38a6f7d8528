// Phase timeline of the PRODUCTION decode GEMV (decode2.hip compiled with -DITTS_GEMV_STAMPS) for the four projections
// of one GPT layer at 2 rows.  Ticks are s_memtime counts; only differences inside one block are meaningful.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iindex-tts-ipex_amd/csrc -Iinclude -DITTS_GEMV_STAMPS -o /tmp/ug2 tools/ubench_gemv2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
#include <algorithm>
namespace itts { static thread_local std::string g_err; void set_error(const std::string& m) { g_err = m; } const char* last_error() { return g_err.c_str(); } }
#include "../index-tts-ipex_amd/csrc/decode2.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)
using namespace itts;

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int REP = 48, B = 2, D = 1280;
  const size_t slab = (size_t)5120 * 1280 * 2;
  char* W; CK(hipMalloc(&W, REP * slab)); CK(hipMemset(W, 0x3c, REP * slab));
  float *Xf, *Y, *bias, *scale; void* Xb; unsigned long long* ST;
  CK(hipMalloc(&Xf, 4 * 5120 * 4)); CK(hipMemset(Xf, 0, 4 * 5120 * 4));
  CK(hipMalloc(&Xb, 4 * 5120 * 2)); CK(hipMemset(Xb, 0, 4 * 5120 * 2));
  CK(hipMalloc(&Y, 4 * 8194 * 4)); CK(hipMalloc(&bias, 8194 * 4)); CK(hipMemset(bias, 0, 8194 * 4));
  CK(hipMalloc(&scale, 8194 * 4)); CK(hipMemset(scale, 0, 8194 * 4));
  CK(hipMalloc(&ST, (size_t)REP * 2048 * 8 * 8));
  struct Case { const char* name; int N, K, pro, xbf, ybf, act, acc, fp8; } cases[] = {
      {"qkv   (LN, fp32 out)", 3 * D, D, 1, 0, 0, ACT_NONE, 0, 0},   {"proj  (bf16 x, +=)", D, D, 0, 1, 0, ACT_NONE, 1, 0},
      {"fc    (LN, gelu, bf16 out)", 4 * D, D, 1, 0, 1, ACT_GELU_NEW, 0, 0}, {"proj2 (bf16 x, +=)", D, 4 * D, 0, 1, 0, ACT_NONE, 1, 0},
      {"qkv   fp8 weights", 3 * D, D, 1, 0, 0, ACT_NONE, 0, 1},      {"fc    fp8 weights", 4 * D, D, 1, 0, 1, ACT_GELU_NEW, 0, 1}};
  for (const Case& c : cases) {
    const int blocks = (c.N + 7) / 8;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < REP; ++i) {
      GemvArgs a;
      a.X = c.xbf ? (const float*)Xb : Xf; a.x_bf16 = c.xbf; a.Y = Y; a.y_bf16 = c.ybf; a.bias = bias; a.B = B; a.N = c.N; a.K = c.K; a.ldy = c.N;
      a.act = c.act; a.accumulate = c.acc; a.prologue = c.pro; a.stamp = ST + (size_t)i * 2048 * 8;
      if (c.fp8) { a.W8 = W + (size_t)i * slab; a.wscale = scale; } else a.W = W + (size_t)i * slab;
      if (gemv_bf16(a, s) != OK) { printf("launch failed: %s\n", last_error()); return 1; }
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)REP * 2048 * 8);
    CK(hipMemcpy(h.data(), ST, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%-28s %4d blocks %.2f us/launch |", c.name, blocks, ms * 1e3 / (5 * REP));
    const char* names[] = {"", "requested", "X landed", "LDS ready", "dots done", "end"};
    for (int ph = 1; ph < 6; ++ph) {
      std::vector<double> v;
      for (int i = 8; i < REP; ++i) for (int b = 0; b < blocks; ++b) v.push_back((double)(h[((size_t)i * 2048 + b) * 8 + ph] - h[((size_t)i * 2048 + b) * 8]));
      std::sort(v.begin(), v.end());
      printf(" %s %5.0f (p90 %5.0f)", names[ph], v[v.size() / 2], v[v.size() * 9 / 10]);
    }
    printf("\n");
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // ---- cache attention (2 rows x 20 heads, S = 380 of Smax = 768), same stamping ----
  {
    const int H = 20, Smax = 768, S = 380;
    bf16_t *kc, *vc, *ctx; float* qkv; int *len, *kvs, *pre; unsigned long long* AS;
    const size_t per = (size_t)B * H * Smax * 64;
    CK(hipMalloc(&kc, per * 2 * REP)); CK(hipMalloc(&vc, per * 2 * REP)); CK(hipMemset(kc, 0, per * 2 * REP)); CK(hipMemset(vc, 0, per * 2 * REP));
    CK(hipMalloc(&ctx, B * 1280 * 2)); CK(hipMalloc(&qkv, B * 3840 * 4)); CK(hipMemset(qkv, 0, B * 3840 * 4));
    CK(hipMalloc(&len, 16)); CK(hipMalloc(&kvs, 16)); CK(hipMalloc(&pre, 64)); CK(hipMemset(kvs, 0, 16));
    int hl[2] = {S - 140, S - 140}, hp = 139;
    CK(hipMemcpy(len, hl, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(pre, &hp, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&AS, (size_t)64 * 8 * 8)); CK(hipMemset(AS, 0, (size_t)64 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamp), &AS, sizeof(AS)));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < REP; ++i)
      if (decode_attn2(ctx, BF16, qkv, kc + per * i, vc + per * i, len, kvs, pre, B, H, 64, Smax, BF16, s, 0) != OK) return 1;
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(64 * 8);
    CK(hipMemcpy(h.data(), AS, h.size() * 8, hipMemcpyDeviceToHost));
    printf("cache attention S=%d          40 blocks %.2f us/launch |", S, ms * 1e3 / (5 * REP));
    const char* an[] = {"", "requests out", "q/k/v used", "softmax done", "wave merged", "barrier", "end"};
    for (int ph = 1; ph < 7; ++ph) {
      std::vector<double> v;
      for (int b = 0; b < 40; ++b) v.push_back((double)(h[(size_t)b * 8 + ph] - h[(size_t)b * 8]));
      std::sort(v.begin(), v.end());
      printf(" %s %5.0f", an[ph], v[v.size() / 2]);
    }
    printf("\n");
  }
  return 0;
}
