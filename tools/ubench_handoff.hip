// What does an in-launch all-to-all edge cost at THIS model's sizes?  (the B <= 4 decode step is 122
// grid-wide synchronisations; a persistent engine replaces kernel boundaries by such edges.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/uh tools/ubench_handoff.hip && /tmp/uh
// One workgroup per CU (256 x 256 threads).  Per edge every workgroup publishes its slice of an n-value vector as 8-byte
// {data, tag} granules (one sc1 store each: the tag travels with the data, no fence), then every workgroup polls the
// whole vector with sc1 loads until all tags carry the edge's sequence number, and uses the data.  Optionally each
// workgroup also streams `wbytes` of "weights" from HBM per edge (issued before the poll, consumed after it), which is
// what the engine would overlap with the edge.  Reported: microseconds per edge, against the 1.6-1.9 us of a kernel
// boundary + ~0.5 us for the consumer's first L2 read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_granule(u32x2* p, unsigned data, unsigned tag) {
  u32x2 v = {data, tag};
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ u32x2 load_granule(const u32x2* p) {
  u32x2 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// NV = values per edge (granules); wkb = KB of "weights" streamed per workgroup and edge by a dedicated loader wave
// (wave 3: its own vmcnt queue, as the engine's LDS-DMA loader would be), waves 0-2 poll the granules
template <int NV, bool SWEEP>
__global__ __launch_bounds__(256) void edge_kernel(u32x2* gran, const u32x4* weights, int wkb, int nedges, int tag0,
                                                   float* out, int* err) {
  constexpr int PER = (NV + 191) / 192;  // granules each polling thread reads per edge (192 polling threads)
  const int wg = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x, wave = tid >> 6, lane = tid & 63;
  float acc = 0.f;
  for (int e = 0; e < nedges; ++e) {
    const unsigned tag = (unsigned)(tag0 + e + 1);
    u32x2* buf = gran + (size_t)(e % 3) * NV;
    // produce: this workgroup's slice of the vector (NV / nwg values, rounded up), one lane per granule
    const int per_wg = (NV + nwg - 1) / nwg;
    if (tid < per_wg) {
      const int i = wg * per_wg + tid;
      if (i < NV) store_granule(buf + i, (unsigned)(i * 7 + e), tag);
    }
    if (wave == 3) {
      // loader: wkb KB for this workgroup, 8 x 1 KB in flight
      const u32x4* wp = weights + ((size_t)(e % 32) * nwg + wg) * 64 * 64;  // 64 KB slab per (edge slot, workgroup)
      for (int i = 0; i < wkb; i += 8) {
        u32x4 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = __builtin_nontemporal_load(wp + (size_t)(i + j) * 64 + lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += (float)(w[j][0] & 1u);
      }
    } else if (SWEEP) {
      // consume, batched: one sweep = all of this thread's granule loads in flight together (sc1 = relaxed agent-scope
      // 8-byte atomic loads, L2-served), then the tags are checked; repeat only for the granules that were not there yet
      unsigned long long v[PER];
      bool ok[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) ok[k] = k * 192 + tid >= NV;
      int spins = 0;
      for (;;) {
#pragma unroll
        for (int k = 0; k < PER; ++k)
          if (!ok[k]) v[k] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(buf + k * 192 + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          if (!ok[k]) ok[k] = (unsigned)(v[k] >> 32) == tag;
          all = all && ok[k];
        }
        if (all) break;
        if (++spins > (1 << 20)) {
          *err = 1;
          return;
        }
      }
#pragma unroll
      for (int k = 0; k < PER; ++k)
        if (k * 192 + tid < NV) acc += (float)((unsigned)v[k] & 0xff);
    } else {
      // consume: poll every granule until its tag is this edge's (bounded), one at a time
      int spins = 0;
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = k * 192 + tid;
        if (i < NV) {
          const u32x2* p = buf + i;
          u32x2 v = load_granule(p);
          while (v[1] != tag) {
            if (++spins > (1 << 22)) {
              *err = 1;
              return;
            }
            __builtin_amdgcn_s_sleep(1);
            v = load_granule(p);
          }
          acc += (float)(v[0] & 0xff);
        }
      }
    }
    __syncthreads();  // the whole workgroup has the vector (an engine would now run its dot products)
  }
  if (tid == 0) out[wg] = acc;
}

// the launch-per-phase structure for comparison: nedges dependent trivial kernels that read the vector
template <int NV>
__global__ __launch_bounds__(256) void phase_kernel(u32x2* gran, const u32x4* weights, int wkb, int e, float* out) {
  constexpr int PER = (NV + 255) / 256;
  const int wg = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x;
  float acc = 0.f;
  const u32x4* wp = weights + ((size_t)(e % 32) * nwg + wg) * 64 * 64;
  u32x4 w[8];  // up to 32 KB per workgroup requested up front by all 256 threads (4 KB per instruction)
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j * 4 < wkb) w[j] = __builtin_nontemporal_load(wp + (size_t)j * 256 + tid);
  u32x2* in = gran + (size_t)(e % 3) * NV;
  u32x2* nxt = gran + (size_t)((e + 1) % 3) * NV;
#pragma unroll
  for (int k = 0; k < PER; ++k)
    if (k * 256 + tid < NV) acc += (float)(in[k * 256 + tid][0] & 0xff);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j * 4 < wkb) acc += (float)(w[j][0] & 1u);
  const int per_wg = (NV + nwg - 1) / nwg;
  if (tid < per_wg && wg * per_wg + tid < NV) nxt[wg * per_wg + tid] = u32x2{(unsigned)acc, (unsigned)e};
  if (tid == 0) out[wg] = acc;
}

template <int NV, bool SWEEP>
void run(const char* name, int wkb, hipStream_t s, u32x2* gran, u32x4* weights, float* out, int* err) {
  const int NE = 120 * 4, NWG = 256;
  static int tag0 = 0;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  // persistent
  hipLaunchKernelGGL((edge_kernel<NV, SWEEP>), dim3(NWG), dim3(256), 0, s, gran, weights, wkb, NE, tag0, out, err);
  tag0 += NE;
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int r = 0; r < 5; ++r) {
    hipLaunchKernelGGL((edge_kernel<NV, SWEEP>), dim3(NWG), dim3(256), 0, s, gran, weights, wkb, NE, tag0, out, err);
    tag0 += NE;
  }
  CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  const float us_edge = ms * 1e3f / (5 * NE);
  // launches (graph of NE dependent kernels)
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int e = 0; e < NE; ++e) hipLaunchKernelGGL((phase_kernel<NV>), dim3(NWG), dim3(256), 0, s, gran, weights, wkb, e, out);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
  CK(hipEventElapsedTime(&ms, a, b));
  const float us_launch = ms * 1e3f / (5 * NE);
  printf("%-38s granules %5d (%3d KB)  weights/WG/edge %3d KB   in-launch edge %.2f us   kernel per phase %.2f us%s\n", name, NV,
         NV * 8 / 1024, wkb, us_edge, us_launch, herr ? "   [SPIN LIMIT HIT]" : "");
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  u32x2* gran; u32x4* weights; float* out; int* err;
  CK(hipMalloc(&gran, 3 * 8192 * 8)); CK(hipMemset(gran, 0, 3 * 8192 * 8));
  const size_t wbytes = (size_t)32 * 256 * 64 * 1024;  // 32 edge slots x 256 workgroups x 64 KB = 512 MB (past the 256 MB MALL)
  CK(hipMalloc(&weights, wbytes)); CK(hipMemset(weights, 1, wbytes));
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  for (int wv : {0, 16, 32}) {  // KB of weights per workgroup and edge (x 256 workgroups = 0 / 4.2 / 8.4 MB per edge; a layer is 39 MB / 5)
    run<1280, false>("ctx bf16 pairs (2x1280), serial poll", wv, s, gran, weights, out, err);
    run<1280, true>("ctx bf16 pairs (2x1280), sweep", wv, s, gran, weights, out, err);
    run<2560, true>("h fp32 (2x1280), sweep", wv, s, gran, weights, out, err);
    run<5120, true>("act bf16 pairs (2x5120), sweep", wv, s, gran, weights, out, err);
  }
  return 0;
}
