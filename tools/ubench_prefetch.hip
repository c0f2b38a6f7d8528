// Does a paced L2 prefetcher on a parallel graph branch shorten a chain of weight-streaming kernels?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/up tools/ubench_prefetch.hip && /tmp/up
// Main chain (one graph): 24 "layers" x 5 dependent kernels shaped like the B <= 4 decode step (c_attn 480 blocks x 20 KB,
// attention 640 x 6 KB of cache, c_proj 160 x 20 KB, c_fc 640 x 20 KB, proj2 160 x 80 KB; every block = 256 threads that
// request their whole chunk up front, reduce and store), each reading bytes nobody has touched for > 256 MB.
// Prefetch branch: ONE kernel, 256 workgroups, forked at the graph root.  It follows a progress word that block 0 of
// every main kernel stores, and `ahead` phases before a kernel runs it loads that kernel's chunks with default-policy
// loads from a workgroup ON THE SAME XCD as the block that will read them (block b of a dispatch runs on XCD b % 8; the
// prefetcher reads its own XCC_ID), so the main kernel's loads hit that XCD's L2.  Variants: off / matched / mismatched
// XCD (Infinity Cache only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Phase { unsigned long long off; unsigned chunk16, nblk; };  // chunk16: 16-byte units per block
#define NPH_MAX 128
__constant__ Phase c_ph[NPH_MAX];

__device__ unsigned long long g_ts[4][NPH_MAX];  // main start / end (block 0), prefetch start / landed (workgroup 0), 100 MHz ticks
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

template <int NL>
__global__ __launch_bounds__(256) void phase_kernel(const u32x4* __restrict__ W, const float* __restrict__ x, float* __restrict__ y,
                                                    unsigned* progress, int q, unsigned char* xcc_of) {
  __shared__ float sx[1280];
  const int tid = threadIdx.x;
  // progress word: phase index, and the XCD block 0 landed on (the dispatcher's round robin carries over from launch to launch)
  if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(progress, (unsigned)q | (xcc_id() << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x == 0 && tid == 0) g_ts[0][q] = wall_clock64();
  float xv[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xv[i] = x[tid + i * 256];
  const u32x4* p = W + c_ph[q].off + (size_t)blockIdx.x * 256 * NL + tid;
  u32x4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = __builtin_nontemporal_load(p + i * 256);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 5; ++i) sx[tid + i * 256] = xv[i];
  __syncthreads();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NL; ++i) acc += __uint_as_float(v[i].x & 0x3f800000u) * sx[(tid * 4 + i) % 1280] + (float)(v[i].w & 1u);
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((tid & 63) == 0) y[blockIdx.x * 4 + (tid >> 6)] = acc;
  if (blockIdx.x == gridDim.x - 1 && tid == 0) g_ts[1][q] = wall_clock64();
  if (xcc_of && tid == 0) xcc_of[blockIdx.x] = (unsigned char)xcc_id();
  if (xcc_of && tid == 0 && blockIdx.x < 16) reinterpret_cast<unsigned*>(xcc_of + 512)[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
}

__global__ __launch_bounds__(256) void prefetch_kernel(const u32x4* __restrict__ W, const unsigned* progress, int nph, int ahead,
                                                       int xcd_shift, int* stats) {
  const int tid = threadIdx.x;
  unsigned xcd = 0;
  unsigned acc = 0;
  __shared__ u32x4 dummy[4 * 256];
  const int slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;  // assumes blockIdx % 8 == XCC_ID; verified via stats[1]
  if (tid == 0 && (blockIdx.x & 7) != xcc_id()) atomicAdd(stats + 1, 1);
  if (xcd_shift == 98) return;  // fork / join only
  for (int q = 0; q < nph; ++q) {
    {
      // wait for phase max(q - ahead, 0) to have started; its block 0 ran on XCD start, so (every grid being a multiple of 8)
      // block b of phase q will run on XCD (start + b) % 8: this workgroup takes the blocks with (start + b) % 8 == its own XCD
      int spins = 0;
      unsigned pw;
      while ((int)((pw = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffffff) < max(q - ahead, 0) || pw == 0xffffffffu) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 16)) {
          if (tid == 0) atomicAdd(stats, 1);
          return;
        }
      }
      xcd = (xcc_id() - (pw >> 24) + xcd_shift) & 7;
    }
    const Phase ph = c_ph[q];
    // fire and forget: LDS-DMA loads (no destination registers, nothing waits on them) into a 4 KB dummy LDS area; up to 63
    // per wave in flight
    if (xcd_shift == 99) continue;  // pace only, no loads
    for (unsigned b = xcd + 8 * slot; b < ph.nblk; b += 8 * nslot) {
      const u32x4* p = W + ph.off + (size_t)b * ph.chunk16;
      for (unsigned i = tid; i < ph.chunk16; i += 256)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i),
                                         (__attribute__((address_space(3))) void*)(dummy + (tid >> 6) * 256), 16, 0, 0);
    }
    // one phase in flight at a time: demand misses of the main chain must not queue behind requests for later phases
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && tid == 0) g_ts[3][q] = wall_clock64();
  }
  if (acc == 0x9e3779b9u) stats[0] = -1;  // keeps the loads alive
}

__global__ void reset_kernel(unsigned* progress) { *progress = 0xffffffffu; }

// serial test: block b warms the chunk that block (b + shift) % nblk of the next launch will read
template <int NL, bool NT>
__global__ __launch_bounds__(256) void warm_kernel(const u32x4* __restrict__ W, int q, int shift, unsigned* sink) {
  const unsigned b = (blockIdx.x + shift) % gridDim.x;
  const u32x4* p = W + c_ph[q].off + (size_t)b * 256 * NL + threadIdx.x;
  unsigned acc = 0;
  u32x4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = NT ? __builtin_nontemporal_load(p + i * 256) : p[i * 256];
#pragma unroll
  for (int i = 0; i < NL; ++i) acc ^= v[i].x;
  if (acc == 0x9e3779b9u) *sink = 1;
}

int main(int argc, char** argv) {
  const int NL_LAYERS = 24;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  // phase table: per layer c_attn, attention (cache), c_proj, c_fc, proj2
  const unsigned nblk[5] = {480, 640, 160, 640, 160};
  const unsigned nl[5] = {5, 2, 5, 5, 20};
  std::vector<Phase> ph;
  unsigned long long off = 0;
  for (int l = 0; l < NL_LAYERS; ++l)
    for (int p = 0; p < 5; ++p) {
      ph.push_back({off, nl[p] * 256, nblk[p]});
      off += (unsigned long long)nl[p] * 256 * nblk[p];
    }
  const int nph = (int)ph.size();
  CK(hipMemcpyToSymbol(HIP_SYMBOL(c_ph), ph.data(), nph * sizeof(Phase)));
  const size_t wbytes = off * 16;
  printf("phases %d, bytes per pass %.1f MB\n", nph, wbytes / 1e6);
  u32x4* W; float *x, *y; unsigned* progress; int* stats; unsigned char* xcc_of;
  CK(hipMalloc(&W, wbytes)); CK(hipMemset(W, 1, wbytes));
  CK(hipMalloc(&x, 1280 * 4)); CK(hipMemset(x, 0, 1280 * 4));
  CK(hipMalloc(&y, 4096 * 4)); CK(hipMalloc(&progress, 256)); CK(hipMalloc(&stats, 8)); CK(hipMalloc(&xcc_of, 1024));
  hipEvent_t ef, ej, a, b; CK(hipEventCreate(&ef)); CK(hipEventCreate(&ej)); CK(hipEventCreate(&a)); CK(hipEventCreate(&b));

  auto build = [&](int mode, int ahead, int pf_blocks, bool skip_weight_free) {
    // mode 0: no prefetch branch, 1: matched XCD, 2: mismatched XCD (+4)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(1), 0, s, progress);
    if (mode) {
      CK(hipEventRecord(ef, s)); CK(hipStreamWaitEvent(s2, ef, 0));
      hipLaunchKernelGGL(prefetch_kernel, dim3(pf_blocks), dim3(256), 0, s2, W, progress, nph, ahead, mode == 2 ? 4 : mode == 3 ? 99 : mode == 4 ? 98 : 0, stats);
      CK(hipEventRecord(ej, s2));
    }
    for (int q = 0; q < nph; ++q) {
      const int p = q % 5;
      if (skip_weight_free && p == 1) continue;
      unsigned char* xo = q == 7 ? xcc_of : nullptr;  // c_proj of layer 1: 160 blocks
      if (nl[p] == 5) hipLaunchKernelGGL((phase_kernel<5>), dim3(nblk[p]), dim3(256), 0, s, W, x, y, progress, q, xo);
      else if (nl[p] == 2) hipLaunchKernelGGL((phase_kernel<2>), dim3(nblk[p]), dim3(256), 0, s, W, x, y, progress, q, xo);
      else hipLaunchKernelGGL((phase_kernel<20>), dim3(nblk[p]), dim3(256), 0, s, W, x, y, progress, q, xo);
    }
    if (mode) CK(hipStreamWaitEvent(s, ej, 0));
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return ge;
  };
  auto time_graph = [&](hipGraphExec_t ge, const char* name) {
    CK(hipMemset(stats, 0, 8));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    const int R = 10;
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    int hs[2]; CK(hipMemcpy(hs, stats, 8, hipMemcpyDeviceToHost));
    unsigned char hx[160]; CK(hipMemcpy(hx, xcc_of, 160, hipMemcpyDeviceToHost));
    int match = 0;
    for (int i = 0; i < 160; ++i) match += hx[i] == ((i + hx[0]) & 7);
    unsigned raw[16]; CK(hipMemcpy(raw, xcc_of + 512, 64, hipMemcpyDeviceToHost));
    printf("   XCC_ID raw of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %x", raw[i]);
    printf("\n");
    printf("%-44s %.1f us per pass = %.2f us per layer  (prefetcher gave up %d, prefetch WGs off their XCD %d, main blocks on b%%8: %d/160)\n", name,
           ms * 1e3 / R, ms * 1e3 / R / NL_LAYERS, hs[0], hs[1], match);
  };
  // serial: warm(q) -> phase(q) for the 24 c_attn (480 x 20 KB) and the 24 proj2 (160 x 80 KB) phases
  for (int shift : {0, 4, -1}) {
    for (int which : {0, 4}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int l = 0; l < NL_LAYERS; ++l) {
        const int q = l * 5 + which;
        if (which == 0) {
          if (shift >= 0) hipLaunchKernelGGL((warm_kernel<5, false>), dim3(480), dim3(256), 0, s, W, q, shift, (unsigned*)stats);
          hipLaunchKernelGGL((phase_kernel<5>), dim3(480), dim3(256), 0, s, W, x, y, progress, q, (unsigned char*)nullptr);
        } else {
          if (shift >= 0) hipLaunchKernelGGL((warm_kernel<20, false>), dim3(160), dim3(256), 0, s, W, q, shift, (unsigned*)stats);
          hipLaunchKernelGGL((phase_kernel<20>), dim3(160), dim3(256), 0, s, W, x, y, progress, q, (unsigned char*)nullptr);
        }
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
      // flush caches between passes: read another 600 MB (the other phases' weights) by running a full pass first
      float tot = 0;
      hipGraphExec_t flush = build(0, 0, 0, false);
      for (int r = 0; r < 5; ++r) {
        CK(hipGraphLaunch(flush, s));
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r) tot += ms;
      }
      printf("serial %s, %s: %.2f us per (warm +) phase pair\n", which == 0 ? "c_attn 480 x 20 KB" : "proj2 160 x 80 KB",
             shift < 0 ? "no warm kernel" : shift == 0 ? "warm on the same XCD" : "warm on another XCD", tot * 1e3 / 4 / NL_LAYERS);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphExecDestroy(flush));
    }
  }
  // prefetcher OUTSIDE the graph: launched eagerly on a second stream just before the (single-branch) graph
  {
    hipGraphExec_t g0 = build(0, 0, 0, false);
    for (int variant = 0; variant < 6; ++variant) {
      const int wgs[6] = {0, 256, 256, 256, 128, 256}, ahead[6] = {0, 1, 2, 2, 2, 3}, shift[6] = {0, 0, 0, 99, 0, 4};
      float tot = 0;
      for (int r = 0; r < 8; ++r) {
        CK(hipMemsetAsync(progress, 0xff, 4, s)); CK(hipStreamSynchronize(s));
        if (wgs[variant]) hipLaunchKernelGGL(prefetch_kernel, dim3(wgs[variant]), dim3(256), 0, s2, W, progress, nph, ahead[variant], shift[variant], stats);
        CK(hipEventRecord(a, s)); CK(hipGraphLaunch(g0, s)); CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r >= 2) tot += ms;
      }
      if (variant <= 2) {
        unsigned long long ts[4][NPH_MAX];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts));
        printf("  timeline (us after main phase 50 started): phase: main start / last block end | prefetch start / landed\n");
        for (int q = 50; q < 60; ++q)
          printf("   q %3d (%d): main %6.2f / %6.2f | prefetch %6.2f / %6.2f\n", q, q % 5, (double)(long long)(ts[0][q] - ts[0][50]) * 0.01, (double)(long long)(ts[1][q] - ts[0][50]) * 0.01,
                 (double)(long long)(ts[2][q] - ts[0][50]) * 0.01, (double)(long long)(ts[3][q] - ts[0][50]) * 0.01);
      }
      printf("outside the graph: %3d WGs, ahead %d, %-14s %.2f us per layer\n", wgs[variant], ahead[variant],
             shift[variant] == 99 ? "pace only" : shift[variant] ? "other XCD" : wgs[variant] ? "matched XCD" : "no prefetcher", tot * 1e3 / 6 / NL_LAYERS);
    }
    CK(hipGraphExecDestroy(g0));
  }
  for (int rep = 0; rep < 1; ++rep) {
    hipGraphExec_t g0 = build(0, 0, 0, false);
    time_graph(g0, "no prefetch");
    for (int ahead : {1, 2, 3}) {
      char nm[96];
      hipGraphExec_t g1 = build(1, ahead, 256, false);
      snprintf(nm, sizeof nm, "prefetch matched XCD, ahead %d, 256 WGs", ahead);
      time_graph(g1, nm);
      CK(hipGraphExecDestroy(g1));
    }
    hipGraphExec_t g3 = build(1, 2, 512, false);
    time_graph(g3, "prefetch matched XCD, ahead 2, 512 WGs");
    hipGraphExec_t g4 = build(3, 2, 256, false);
    time_graph(g4, "prefetch branch paces only (no loads)");
    hipGraphExec_t g5 = build(4, 2, 256, false);
    time_graph(g5, "prefetch branch exits at once (fork / join only)");
    hipGraphExec_t g6 = build(3, 2, 32, false);
    time_graph(g6, "paces only, 32 WGs");
    hipGraphExec_t g7 = build(1, 2, 64, false);
    time_graph(g7, "prefetch matched XCD, ahead 2, 64 WGs");
    hipGraphExec_t g2 = build(2, 2, 256, false);
    time_graph(g2, "prefetch other XCD (Infinity Cache only), ahead 2");
    CK(hipGraphExecDestroy(g0)); CK(hipGraphExecDestroy(g2)); CK(hipGraphExecDestroy(g3));
  }
  return 0;
}
