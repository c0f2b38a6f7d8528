// Common definitions for libitts_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace itts {

// The 16-bit storage type of the engine ("half"): bfloat16 in libitts_hip.so, IEEE binary16 in libitts_hip_f16.so, which is
// the SAME sources compiled with -DITTS_HALF_F16 - the reference's GPU precision (`is_fp16=True` is fp16 autocast / .half(),
// indextts/infer.py:39,44,52).  Every kernel is written against `bf16_t` and the helpers below; in the f16 build the name is a
// misnomer for _Float16 (weights, activations, K/V cache in binary16; fp32 accumulation everywhere as before).
#ifdef ITTS_HALF_F16
typedef _Float16 bf16_t;
#else
typedef __bf16 bf16_t;
#endif
typedef _Float16 f16_t;

enum DType : int { F32 = 0, BF16 = 1, I32 = 2, I64 = 3, FP8 = 4, F16 = 5 };  // F16: operator-level boundary only (itts_snake_aa_fwd)  // FP8 = OCP e4m3fn bytes (weights only)
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_GELU_NEW = 3, ACT_GELU_ERF = 4, ACT_TANH = 5, ACT_SIGMOID = 6 };
enum PadMode : int { PAD_ZERO = 0, PAD_REFLECT = 1 };

inline size_t dtype_size(int dt) { return dt == F32 ? 4 : (dt == BF16 || dt == F16) ? 2 : dt == I32 ? 4 : dt == FP8 ? 1 : 8; }

// status codes of the C ABI (0 ok, negative = error; message via itts_last_error())
// E_HANDOFF: an in-launch hand-off of the persistent decode engine (or of the fused q/k/v launch) timed out - the codes of this
// generation are not valid, the engine object has switched to the launch path, the caller may simply generate again
enum Status : int { OK = 0, E_INVALID = -1, E_HIP = -2, E_NOMEM = -3, E_STATE = -4, E_MISSING = -5, E_HANDOFF = -6 };

void set_error(const std::string& msg);
const char* last_error();

#define ITTS_HIP_CHECK(expr)                                                                         \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      ::itts::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                        std::to_string(__LINE__));                                                   \
      return ::itts::E_HIP;                                                                          \
    }                                                                                                \
  } while (0)

#define ITTS_REQUIRE(cond, msg)                                                         \
  do {                                                                                  \
    if (!(cond)) {                                                                      \
      ::itts::set_error(std::string("invalid argument: ") + msg + " (" #cond ") at " + __FILE__ + ":" + \
                        std::to_string(__LINE__));                                      \
      return ::itts::E_INVALID;                                                         \
    }                                                                                   \
  } while (0)

#define ITTS_TRY(expr)              \
  do {                              \
    int _s = (expr);                \
    if (_s != ::itts::OK) return _s; \
  } while (0)

// ---- device helpers ----
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16_t* p) { return (float)(*p); }
#ifndef ITTS_HALF_F16  // (the same type there)
__device__ __forceinline__ float ldf(const f16_t* p) { return (float)(*p); }
#endif
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16_t* p, float v) { *p = (bf16_t)v; }
#ifndef ITTS_HALF_F16
__device__ __forceinline__ void stf(f16_t* p, float v) { *p = (f16_t)v; }
#endif

// ---- the half type at the bit level: two halves of a 32-bit word, the 16x16x32 matrix instruction, the 2-way dot product ----
typedef bf16_t half2_t __attribute__((ext_vector_type(2)));
typedef short half8_bits __attribute__((ext_vector_type(8)));  // MFMA fragments travel as 8 x 16 bits
typedef float f32x4_acc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float half_lo(uint32_t w) {  // element 0 (low 16 bits) of a pair
#ifdef ITTS_HALF_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu));
#else
  return __uint_as_float(w << 16);
#endif
}
__device__ __forceinline__ float half_hi(uint32_t w) {
#ifdef ITTS_HALF_F16
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16));
#else
  return __uint_as_float(w & 0xFFFF0000u);
#endif
}
__device__ __forceinline__ float half_bits(unsigned short h) {
#ifdef ITTS_HALF_F16
  return (float)__builtin_bit_cast(_Float16, h);
#else
  return __uint_as_float((uint32_t)h << 16);
#endif
}
__device__ __forceinline__ float half_dot2(uint32_t a, uint32_t b, float c) {  // c + a.lo * b.lo + a.hi * b.hi, fp32 accumulate
#ifdef ITTS_HALF_F16
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b), c, false);  // v_dot2_f32_f16
#else
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, a), __builtin_bit_cast(b2, b), c, false);  // v_dot2c_f32_bf16
#endif
}
__device__ __forceinline__ f32x4_acc half_mfma16(half8_bits a, half8_bits b, f32x4_acc c) {  // v_mfma_f32_16x16x32_{bf16,f16}
#ifdef ITTS_HALF_F16
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_SILU: return x / (1.f + __expf(-x));
    case ACT_GELU_NEW: {
      const float k = 0.7978845608028654f;  // sqrt(2/pi)
      return 0.5f * x * (1.f + tanhf(k * (x + 0.044715f * x * x * x)));
    }
    case ACT_GELU_ERF: return 0.5f * x * (1.f + erff(x * 0.7071067811865476f));
    case ACT_TANH: return tanhf(x);
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    default: return x;
  }
}

// The matrix-core GEMM epilogues (half-precision outputs): NewGELU as x * sigmoid(2 u) - one v_exp and one v_rcp instead of
// tanhf's ~25 instructions, the same function (the decode step keeps its pinned tanhf form, decode_pinned.h; the fp32 parity
// engine runs gemm_simple with act_apply)
__device__ __forceinline__ float act_apply_fast(int act, float x) {
  if (act == ACT_GELU_NEW) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.f + __expf(-2.f * u));
  }
  return act_apply(act, x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Arguments of the generic "shift-GEMM" (linear / conv1d / transposed conv1d / upsampled conv) on
// channels-last activations:   C[m, n] = epilogue( sum_{tap, c} A[row(m, tap), c] * W[n, tap*Cin + c] )
struct GemmArgs {
  const void* A = nullptr;  // [M rows][lda], element type TA
  const void* W = nullptr;  // [nphase][N][taps*Cin], element type TW
  void* C = nullptr;        // [M rows][ldc] (row m, col phase*N + n), element type TC
  int M = 0, N = 0, Cin = 0, taps = 1;
  int lda = 0, ldc = 0;
  int T = 0;         // rows per batch item in OUTPUT-row space (M = B*T); shifts never cross an item
  int dil = 1;       // row offset of tap j = phase_shift + j*dil - pad_left
  int pad_left = 0;
  int pad_mode = PAD_ZERO;
  int in_up = 1;     // input is nearest-upsampled by in_up: source row = (t + off) / in_up
  int nphase = 1;    // transposed conv: output columns [phase*N, (phase+1)*N), W slab per phase
  int phase_shift[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // epilogue: v = acc + bias[b*bias_bstride + col]; v = act(v); v = v*scale[col] + shift[col]; v = act2(v);
  //           v += R[m, col]; v *= alpha; v += beta * ADD[m, col]
  const float* bias = nullptr;
  int bias_bstride = 0;
  int act = ACT_NONE;
  const float* scale = nullptr;
  const float* shift = nullptr;
  int act2 = ACT_NONE;
  const void* R = nullptr;  // residual, element type TC
  int ldr = 0;
  float alpha = 1.f;
  const void* ADD = nullptr;  // element type TC
  int ldadd = 0;
  float beta = 0.f;
  // Activation1d in front of the convolution (AMPBlock1: x -> act -> conv, BigVGAN/models.py:65-74), applied while the input tile
  // is staged - only the kernels that say so in conv_lds_act_supported() take it; A is then the PRE-activation tensor
  const float* pre_alpha = nullptr;  // log-scale alpha / beta [Cin] of the SnakeBeta, the 12-tap kaiser-sinc filter (up = down)
  const float* pre_beta = nullptr;
  const float* pre_filt = nullptr;
  // K split over workgroups (gemm_p8 only; set by the selector, never by callers): split s of ksplit accumulates its share of the
  // K-tiles into ws[s][M][N] (fp32, raw sums) and a second launch adds the shares in split order and runs the epilogue
  float* ws = nullptr;
  int ksplit = 1;
};

}  // namespace itts
