// Device-side core of the fused anti-aliased SnakeBeta ("Activation1d": alias_free_torch/act.py:24-29, resample.py:24-33,
// filter.py:86-95; the reference's CUDA twin anti_alias_activation_cuda.cu:43-181), shared by the stand-alone kernel
// (snake.hip, snake_aa_lds_kernel) and by the narrow-stage convolution that applies it while it stages its input tile
// (conv_lds.hip, ACT): one lane slides along ONE channel over a run of time steps, holding the 12-sample window of activated
// up-sampled values and the 6-sample input window in registers - per output 2 new up-FIR phases (12 FMA), 2 SnakeBeta
// evaluations and the 12-tap down-FIR.  Same operations in the same order in both users, so the fused convolution sees exactly
// the bf16 values the stand-alone kernel would have written.
#pragma once
#include "itts_common.h"

namespace itts {

// col: this channel's column of an LDS tile whose row j holds x[clamp(tbase + j, 0, Tn - 1)] (replicate padding resolved when the
// tile was loaded), rows `stride` elements apart.  Emits out(t, y[t]) for t in [ts, te) (0 <= ts < te <= Tn).
// FAST: v_sin_f32 (__sinf) as the reference's fast-math CUDA build; otherwise sinf.
// Instruction diet (r04): the down-FIR runs as TWO chains (even / odd taps) on packed fp32 (v_pk_fma_f32: 6 + 1 instructions instead
// of 12), the factor 2 of the zero-stuffed up-sampling is folded into the up-FIR taps (exact: a power of two), and FAST evaluates
// v_sin_f32 (argument in revolutions) on u * (e^alpha / 2 pi) directly - one multiply per sine instead of two.
typedef float snk_f2 __attribute__((ext_vector_type(2)));

template <typename T, bool FAST, typename Out>
__device__ __forceinline__ void snake_run(const T* __restrict__ col, int stride, int tbase, int ts, int te, int Tn, float ea, float inv_b,
                                          const float (&fu)[12], const float (&fd)[12], Out&& out) {
  const float ear = ea * 0.15915494309189535f;  // FAST: e^alpha in revolutions per unit
  auto act = [&](float u) {
    const float sn = FAST ? __builtin_amdgcn_sinf(u * ear) : sinf(u * ea);
    return u + inv_b * sn * sn;
  };
  auto act2 = [&](snk_f2 u) {
    snk_f2 sn;
    sn.x = FAST ? __builtin_amdgcn_sinf(u.x * ear) : sinf(u.x * ea);
    sn.y = FAST ? __builtin_amdgcn_sinf(u.y * ear) : sinf(u.y * ea);
    return __builtin_elementwise_fma(sn * inv_b, sn, u);
  };
  auto xin = [&](int t) { return ldf(col + (t - tbase) * stride); };
  float fu2[12];  // 2 x the up-FIR taps
  snk_f2 fup[6], fdp[6];  // (odd-phase tap, even-phase tap) of input r; down-FIR taps (2 i, 2 i + 1)
#pragma unroll
  for (int j = 0; j < 12; ++j) fu2[j] = 2.f * fu[j];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    fup[r] = snk_f2{fu2[2 * r], fu2[2 * r + 1]};
    fdp[r] = snk_f2{fd[2 * r], fd[2 * r + 1]};
  }
  const int mlast = 2 * Tn - 1;
  auto v_at = [&](int m) {  // activated up-sampled sample m (clamped to [0, 2 Tn - 1])
    m = m < 0 ? 0 : (m > mlast ? mlast : m);
    const int q = m >> 1;
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu2[2 * r], xin(q + 3 - r), u);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu2[2 * r + 1], xin(q + 2 - r), u);
    }
    return act(u);
  };
  snk_f2 vp[6];  // vp[i] = (V(2t - 5 + 2i), V(2t - 4 + 2i))
  float xs[6];   // xs[i] = x[clamp(t + i)]
#pragma unroll
  for (int i = 0; i < 5; ++i) vp[i] = snk_f2{v_at(2 * ts - 5 + 2 * i), v_at(2 * ts - 4 + 2 * i)};
#pragma unroll
  for (int i = 0; i < 6; ++i) xs[i] = xin(ts + i);
  const T* px = col + (ts + 6 - tbase) * stride;  // next input row to enter the window
  auto down = [&]() {
    snk_f2 oo = fdp[0] * vp[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) oo = __builtin_elementwise_fma(fdp[i], vp[i], oo);
    return oo.x + oo.y;
  };
  auto up = [&]() {
    snk_f2 u = fup[0] * snk_f2{xs[5], xs[5]};
#pragma unroll
    for (int r = 1; r < 6; ++r) u = __builtin_elementwise_fma(fup[r], snk_f2{xs[5 - r], xs[5 - r]}, u);
    return u;
  };
  if (2 * (te - 1) + 6 <= mlast) {
    // interior run: no end-of-stream selects; unrolled by the rotation period of the two register windows so the shifts become renames
#pragma unroll 6
    for (int t = ts; t < te; ++t) {
      vp[5] = act2(up());
      out(t, down());
#pragma unroll
      for (int i = 0; i < 5; ++i) vp[i] = vp[i + 1];
#pragma unroll
      for (int i = 0; i < 5; ++i) xs[i] = xs[i + 1];
      xs[5] = ldf(px);
      px += stride;
    }
  } else {
    for (int t = ts; t < te; ++t) {
      const snk_f2 u = up();
      const float vprev = vp[4].y;
      const float a = (2 * t + 5 <= mlast) ? act(u.x) : vprev;
      const float b = (2 * t + 6 <= mlast) ? act(u.y) : a;
      vp[5] = snk_f2{a, b};
      out(t, down());
#pragma unroll
      for (int i = 0; i < 5; ++i) vp[i] = vp[i + 1];
#pragma unroll
      for (int i = 0; i < 5; ++i) xs[i] = xs[i + 1];
      xs[5] = ldf(px);
      px += stride;
    }
  }
}

}  // namespace itts
