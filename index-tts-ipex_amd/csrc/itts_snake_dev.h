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
template <typename T, bool FAST, typename Out>
__device__ __forceinline__ void snake_run(const T* __restrict__ col, int stride, int tbase, int ts, int te, int Tn, float ea, float inv_b,
                                          const float (&fu)[12], const float (&fd)[12], Out&& out) {
  auto act = [&](float u) {
    const float sn = FAST ? __sinf(u * ea) : sinf(u * ea);
    return u + inv_b * sn * sn;
  };
  auto xin = [&](int t) { return ldf(col + (t - tbase) * stride); };
  const int mlast = 2 * Tn - 1;
  auto v_at = [&](int m) {  // activated up-sampled sample m (clamped to [0, 2 Tn - 1])
    m = m < 0 ? 0 : (m > mlast ? mlast : m);
    const int q = m >> 1;
    float u = 0.f;
    if (m & 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu[2 * r], xin(q + 3 - r), u);
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) u = fmaf(fu[2 * r + 1], xin(q + 2 - r), u);
    }
    return act(2.f * u);
  };
  float v[12], xs[6];  // v[j] = V(2t - 5 + j), xs[i] = x[clamp(t + i)]
#pragma unroll
  for (int j = 0; j < 10; ++j) v[j] = v_at(2 * ts - 5 + j);
#pragma unroll
  for (int i = 0; i < 6; ++i) xs[i] = xin(ts + i);
  const T* px = col + (ts + 6 - tbase) * stride;  // next input row to enter the window
  if (2 * (te - 1) + 6 <= mlast) {
    // interior run: no end-of-stream selects; unrolled by the rotation period of the two register windows so the shifts become renames
#pragma unroll 6
    for (int t = ts; t < te; ++t) {
      float uo = 0.f, ue = 0.f;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        uo = fmaf(fu[2 * r], xs[5 - r], uo);
        ue = fmaf(fu[2 * r + 1], xs[5 - r], ue);
      }
      v[10] = act(2.f * uo);
      v[11] = act(2.f * ue);
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 12; ++j) o = fmaf(fd[j], v[j], o);
      out(t, o);
#pragma unroll
      for (int j = 0; j < 10; ++j) v[j] = v[j + 2];
#pragma unroll
      for (int i = 0; i < 5; ++i) xs[i] = xs[i + 1];
      xs[5] = ldf(px);
      px += stride;
    }
  } else {
    for (int t = ts; t < te; ++t) {
      float uo = 0.f, ue = 0.f;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        uo = fmaf(fu[2 * r], xs[5 - r], uo);
        ue = fmaf(fu[2 * r + 1], xs[5 - r], ue);
      }
      const float vprev = v[9];
      v[10] = (2 * t + 5 <= mlast) ? act(2.f * uo) : vprev;
      v[11] = (2 * t + 6 <= mlast) ? act(2.f * ue) : v[10];
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < 12; ++j) o = fmaf(fd[j], v[j], o);
      out(t, o);
#pragma unroll
      for (int j = 0; j < 10; ++j) v[j] = v[j + 2];
#pragma unroll
      for (int i = 0; i < 5; ++i) xs[i] = xs[i + 1];
      xs[5] = ldf(px);
      px += stride;
    }
  }
}

}  // namespace itts
