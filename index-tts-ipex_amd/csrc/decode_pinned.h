// Floating-point expressions of the <= 4-row decode step whose contraction must not be left to the compiler: the
// persistent engine (decode_engine.hip) and the launch path (decode2.hip) have to produce the same bits, and hipcc fuses
// `a * b + c` differently depending on what the SLP vectoriser found around it (measured: LayerNorm's variance came out
// as fma(-md, md, Q / K) in one kernel and fma(Q, 1 / K, -(md * md)) in the other - a one-ulp difference in rstd that
// flipped a bf16 rounding in block 20).  Every operation here is a single correctly rounded instruction.
#pragma once
#include "itts_common.h"

namespace itts {

// E[d^2] - E[d]^2 with Q = sum d^2, md = mean d
__device__ __forceinline__ float ln_var_rn(float Q, float invK, float md) { return fmaf(-md, md, __fmul_rn(Q, invK)); }

// LayerNorm affine (ln_f in front of final_norm, head GEMV prologue 2)
__device__ __forceinline__ float ln_affine_rn(float v, float g, float b) { return fmaf(v, g, b); }

// HF NewGELU: 0.5 x (1 + tanh(sqrt(2 / pi) (x + 0.044715 x^3)))
__device__ __forceinline__ float gelu_new_rn(float x) {
  const float x3 = __fmul_rn(__fmul_rn(x, x), x);
  const float inner = __fmul_rn(0.7978845608028654f, fmaf(0.044715f, x3, x));
  return __fmul_rn(__fmul_rn(0.5f, x), __fadd_rn(1.f, tanhf(inner)));
}

}  // namespace itts
