// Row-wise / elementwise kernels of the conditioning encoder, ECAPA-TDNN and the GPT glue.
// All HBM-bound; lanes always walk the contiguous (channel) axis.
#include "itts_kernels.h"

namespace itts {
namespace {

#define DISPATCH_T(dt, T, ...)            \
  if ((dt) == F32) {                      \
    typedef float T;                      \
    __VA_ARGS__;                          \
  } else if ((dt) == BF16) {              \
    typedef bf16_t T;                     \
    __VA_ARGS__;                          \
  } else {                                \
    set_error("unsupported dtype");       \
    return E_INVALID;                     \
  }

// ---------------- LayerNorm: one wave per row ----------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(TO* __restrict__ y, const TI* __restrict__ x,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        int rows, int D, int ldx, int ldy, float eps, int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const TI* xr = x + (size_t)row * ldx;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += ldf(xr + i);
  const float mean = wave_sum(s) / D;
  float v = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = ldf(xr + i) - mean;
    v = fmaf(d, d, v);
  }
  const float rstd = rsqrtf(wave_sum(v) / D + eps);
  TO* yr = y + (size_t)row * ldy;
  for (int i = lane; i < D; i += 64) {
    float o = (ldf(xr + i) - mean) * rstd;
    if (g) o = o * g[i] + bta[i];
    stf(yr + i, act_apply(act, o));
  }
}

// Same LayerNorm, one wave per row, for D % 4 == 0 and D <= 2048: the row is read ONCE with 16-byte (fp32) / 8-byte (bf16)
// loads into registers (lane owns elements i*256 + lane*4 .. +4), both statistics passes run on registers, the
// output leaves as vectors.  The scalar kernel above re-reads the row three times with 2/4-byte loads (14 us for
// 1242 x 1280 in the latent pass).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(TO* __restrict__ y, const TI* __restrict__ x,
                                                            const float* __restrict__ g, const float* __restrict__ bta,
                                                            int rows, int D, int ldx, int ldy, float eps, int act) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const TI* xr = x + (size_t)row * ldx;
  float v[8][4];
  bool ok[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = i * 256 + lane * 4;
    ok[i] = k < D;
    const int kc = ok[i] ? k : 0;
    if constexpr (sizeof(TI) == 4) {
      const float4 t = *reinterpret_cast<const float4*>(xr + kc);
      v[i][0] = t.x; v[i][1] = t.y; v[i][2] = t.z; v[i][3] = t.w;
    } else {
      const uint2 t = *reinterpret_cast<const uint2*>(xr + kc);
      v[i][0] = half_lo(t.x); v[i][1] = half_hi(t.x);
      v[i][2] = half_lo(t.y); v[i][3] = half_hi(t.y);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) s += ok[i] ? v[i][e] : 0.f;
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = ok[i] ? v[i][e] - mean : 0.f;
      q = fmaf(d, d, q);
    }
  const float rstd = rsqrtf(wave_sum(q) / D + eps);
  TO* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (!ok[i]) continue;
    const int k = i * 256 + lane * 4;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = (v[i][e] - mean) * rstd;
      if (g) o[e] = o[e] * g[k + e] + bta[k + e];
      o[e] = act_apply(act, o[e]);
    }
    if constexpr (sizeof(TO) == 4) {
      *reinterpret_cast<float4*>(yr + k) = float4{o[0], o[1], o[2], o[3]};
    } else {
      bf16_t t[4] = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
      *reinterpret_cast<uint2*>(yr + k) = *reinterpret_cast<const uint2*>(t);
    }
  }
}

// F.normalize(x, dim=-1) * sqrt(D) * gamma   (perceiver.py:167-186)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void rmsnorm_unit_kernel(TO* __restrict__ y, const TI* __restrict__ x,
                                                           const float* __restrict__ g, int rows, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const TI* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = ldf(xr + i);
    s = fmaf(d, d, s);
  }
  const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
  const float sc = sqrtf((float)D) / nrm;
  for (int i = lane; i < D; i += 64) stf(y + (size_t)row * D + i, ldf(xr + i) * sc * g[i]);
}

template <typename T>
__global__ void glu_kernel(T* __restrict__ y, const T* __restrict__ x, long n, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / C;
  const int c = (int)(i - r * C);
  const float a = ldf(x + r * 2 * C + c), b = ldf(x + r * 2 * C + C + c);
  stf(y + i, a / (1.f + __expf(-b)));
}

// GEGLU (perceiver.py:204-207): x, gate = chunk(2); gelu(gate) * x.  Output row stride ldy >= inner, the
// tail [inner, ldy) is zero-filled so a K-padded GEMM can consume it.
template <typename T>
__global__ void geglu_kernel(T* __restrict__ y, const T* __restrict__ x, int rows, int inner, int ldy) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * ldy) return;
  const long r = i / ldy;
  const int c = (int)(i - r * ldy);
  float o = 0.f;
  if (c < inner) {
    const float a = ldf(x + r * 2 * inner + c), gt = ldf(x + r * 2 * inner + inner + c);
    o = act_apply(ACT_GELU_ERF, gt) * a;
  }
  stf(y + i, o);
}

// depthwise conv along time, zero padding (k-1)/2, weights [C][k] fp32 (conformer_encoder.py:126-137)
template <typename T>
__global__ void dwconv_kernel(T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ w,
                              const float* __restrict__ bias, int B, int Tn, int C, int k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Tn * C) return;
  const int c = (int)(i % C);
  const int t = (int)((i / C) % Tn);
  const long b = i / ((long)C * Tn);
  const int pad = (k - 1) / 2;
  float acc = bias ? bias[c] : 0.f;
  for (int j = 0; j < k; ++j) {
    const int ts = t + j - pad;
    if (ts >= 0 && ts < Tn) acc = fmaf(w[c * k + j], ldf(x + (b * Tn + ts) * C + c), acc);
  }
  stf(y + i, acc);
}

// Conv2d(1, odim, 3, stride 2) + ReLU over (t, f) of mel [B, F, idim]; output row t' holds (c, f') c-major
// (subsampling.py:151-153,182-184).  Weights [odim][3][3] fp32.
template <typename T>
__global__ void conv2d_sub2_kernel(T* __restrict__ y, const T* __restrict__ mel, const float* __restrict__ w,
                                   const float* __restrict__ bias, int B, int F, int idim, int odim, int Fo, int fo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Fo * odim * fo;
  if (i >= total) return;
  const int f = (int)(i % fo);
  const int c = (int)((i / fo) % odim);
  const int t = (int)((i / ((long)fo * odim)) % Fo);
  const long b = i / ((long)fo * odim * Fo);
  float acc = bias[c];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int d = 0; d < 3; ++d)
      acc = fmaf(w[c * 9 + a * 3 + d], ldf(mel + (b * F + 2 * t + a) * idim + 2 * f + d), acc);
  stf(y + i, acc > 0.f ? acc : 0.f);
}

// y[r] = (ia[r] >= 0 ? A[ia[r]] : 0) + (ib[r] >= 0 ? B[ib[r]] : 0)
template <typename TT, typename TO>
__global__ void gather_add_kernel(TO* __restrict__ y, int ldy, const TT* __restrict__ A, const int* __restrict__ ia,
                                  const TT* __restrict__ Bt, const int* __restrict__ ib, int rows, int D) {
  const int r = blockIdx.x;
  const int a = ia ? ia[r] : -1, b = (Bt && ib) ? ib[r] : -1;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    float v = 0.f;
    if (a >= 0) v += ldf(A + (size_t)a * D + i);
    if (b >= 0) v += ldf(Bt + (size_t)b * D + i);
    stf(y + (size_t)r * ldy + i, v);
  }
}

template <typename T>
__global__ void transpose_kernel(T* __restrict__ y, const T* __restrict__ x, int R, int C) {
  __shared__ float tile[32][33];
  const long b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = ldf(x + (b * R + r) * C + c);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) stf(y + (b * C + c) * R + r, tile[tx][i]);
  }
}

template <typename TI, typename TO>
__global__ void cast_kernel(TO* __restrict__ y, const TI* __restrict__ x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stf(y + i, ldf(x + i));
}

template <typename T>
__global__ void copy_rows_kernel(T* __restrict__ y, int ldy, const T* __restrict__ x, int ldx, int rows, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * D) return;
  const long r = i / D;
  const int c = (int)(i - r * D);
  y[r * ldy + c] = x[r * ldx + c];
}

template <typename T>
__global__ void add_strided_kernel(T* __restrict__ y, int ldy, const T* __restrict__ a, int lda,
                                   const T* __restrict__ b, int ldb, int rows, int D) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * D) return;
  const long r = i / D;
  const int c = (int)(i - r * D);
  stf(y + r * ldy + c, ldf(a + r * lda + c) + ldf(b + r * ldb + c));
}

// per (b, c): mean over t [and population std clamped at 1e-12] (ECAPA_TDNN.py:223-242, 283-320)
template <typename T, bool STD>
__global__ void col_stats_kernel(float* __restrict__ out, const T* __restrict__ x, int B, int Tn, int C, int ldx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int c = i % C, b = i / C;
  const T* xb = x + (size_t)b * Tn * ldx + c;
  float s = 0.f;
  for (int t = 0; t < Tn; ++t) s += ldf(xb + (size_t)t * ldx);
  const float mean = s / Tn;
  if (!STD) {
    out[i] = mean;
    return;
  }
  float v = 0.f;
  for (int t = 0; t < Tn; ++t) {
    const float d = ldf(xb + (size_t)t * ldx) - mean;
    v = fmaf(d, d, v);
  }
  out[(size_t)b * 2 * C + c] = mean;
  out[(size_t)b * 2 * C + C + c] = sqrtf(fmaxf(v / Tn, 1e-12f));
}

// y = x * sc[b, c] + res   (SE gate + block residual, ECAPA_TDNN.py:241,425)
template <typename T>
__global__ void scale_cols_add_kernel(T* __restrict__ y, int ldy, const T* __restrict__ x, int ldx,
                                      const float* __restrict__ sc, const T* __restrict__ res, int ldr, int B, int Tn,
                                      int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Tn * C) return;
  const int c = (int)(i % C);
  const long r = i / C;
  const long b = r / Tn;
  float v = ldf(x + r * ldx + c) * sc[b * C + c];
  if (res) v += ldf(res + r * ldr + c);
  stf(y + r * ldy + c, v);
}

// attentive statistics pooling tail (ECAPA_TDNN.py:328-338): per (b,c) softmax over time of logits, weighted
// mean / std (clamp 1e-12), then the eval BatchNorm affine of asp_bn: out[b, c] and out[b, C + c].
template <typename T>
__global__ void asp_pool_kernel(float* __restrict__ out, const T* __restrict__ lg, const T* __restrict__ x,
                                const float* __restrict__ bs, const float* __restrict__ bsh, int B, int Tn, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int c = i % C, b = i / C;
  const T* lb = lg + (size_t)b * Tn * C + c;
  const T* xb = x + (size_t)b * Tn * C + c;
  float mx = -INFINITY;
  for (int t = 0; t < Tn; ++t) mx = fmaxf(mx, ldf(lb + (size_t)t * C));
  float den = 0.f, m1 = 0.f;
  for (int t = 0; t < Tn; ++t) {
    const float e = expf(ldf(lb + (size_t)t * C) - mx);
    den += e;
    m1 = fmaf(e, ldf(xb + (size_t)t * C), m1);
  }
  const float mean = m1 / den;
  float v = 0.f;
  for (int t = 0; t < Tn; ++t) {
    const float e = expf(ldf(lb + (size_t)t * C) - mx) / den;
    const float d = ldf(xb + (size_t)t * C) - mean;
    v = fmaf(e, d * d, v);
  }
  const float sd = sqrtf(fmaxf(v, 1e-12f));
  out[(size_t)b * 2 * C + c] = mean * bs[c] + bsh[c];
  out[(size_t)b * 2 * C + C + c] = sd * bs[C + c] + bsh[C + c];
}

// Conformer rel-pos attention without rel-shift (attention.py:295-309):
//   scores = ((q+u) k^T + (q+v) p^T)/sqrt(dk)  ==  [q+u | q+v] . [k | p]
// qkv [T, 3*H*dk] (q | k | v), p [T, H*dk] -> Qc, Kc [T, H, 2dk]
template <typename T>
__global__ void relpos_pack_kernel(T* __restrict__ qc, T* __restrict__ kc, const T* __restrict__ qkv,
                                   const T* __restrict__ p, const float* __restrict__ bu, const float* __restrict__ bv,
                                   int Tn, int H, int dk) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int D = H * dk;
  if (i >= (long)Tn * D) return;
  const int d = (int)(i % dk);
  const int h = (int)((i / dk) % H);
  const long t = i / D;
  const float q = ldf(qkv + t * 3 * D + h * dk + d);
  const float k = ldf(qkv + t * 3 * D + D + h * dk + d);
  const long o = (t * H + h) * 2 * dk;
  stf(qc + o + d, q + bu[h * dk + d]);
  stf(qc + o + dk + d, q + bv[h * dk + d]);
  stf(kc + o + d, k);
  stf(kc + o + dk + d, ldf(p + t * D + h * dk + d));
}

template <typename T>
__global__ void tanh_kernel(T* __restrict__ y, const T* __restrict__ x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stf(y + i, tanhf(ldf(x + i)));
}

inline int nblk(long n, int bs = 256) { return (int)((n + bs - 1) / bs); }

}  // namespace

#define CHECK_LAUNCH()                    \
  ITTS_HIP_CHECK(hipGetLastError());      \
  return OK

int layernorm(void* y, int ty, const void* x, int tx, const float* gamma, const float* beta, int rows, int D, int ldx,
              int ldy, float eps, int act, hipStream_t s) {
  ITTS_REQUIRE(y && x && rows > 0 && D > 0 && (gamma == nullptr) == (beta == nullptr), "layernorm");
  dim3 grid((rows + 3) / 4), blk(256);
  const bool vec = D % 4 == 0 && D <= 2048 && ldx % 4 == 0 && ldy % 4 == 0 && !(((uintptr_t)x | (uintptr_t)y) & 15);
#define LN(TI, TO)                                                                                                       \
  if (vec)                                                                                                               \
    hipLaunchKernelGGL((layernorm_vec_kernel<TI, TO>), grid, blk, 0, s, (TO*)y, (const TI*)x, gamma, beta, rows, D, ldx, \
                       ldy, eps, act);                                                                                   \
  else                                                                                                                   \
    hipLaunchKernelGGL((layernorm_kernel<TI, TO>), grid, blk, 0, s, (TO*)y, (const TI*)x, gamma, beta, rows, D, ldx, ldy, eps, act)
  if (tx == F32 && ty == F32) LN(float, float);
  else if (tx == F32 && ty == BF16) LN(float, bf16_t);
  else if (tx == BF16 && ty == BF16) LN(bf16_t, bf16_t);
  else if (tx == BF16 && ty == F32) LN(bf16_t, float);
  else { set_error("layernorm: dtype"); return E_INVALID; }
#undef LN
  CHECK_LAUNCH();
}

int rmsnorm_unit(void* y, int ty, const void* x, int tx, const float* gamma, int rows, int D, hipStream_t s) {
  ITTS_REQUIRE(y && x && gamma && rows > 0 && D > 0, "rmsnorm");
  dim3 grid((rows + 3) / 4), blk(256);
  if (tx == F32 && ty == F32)
    hipLaunchKernelGGL((rmsnorm_unit_kernel<float, float>), grid, blk, 0, s, (float*)y, (const float*)x, gamma, rows, D);
  else if (tx == BF16 && ty == F32)
    hipLaunchKernelGGL((rmsnorm_unit_kernel<bf16_t, float>), grid, blk, 0, s, (float*)y, (const bf16_t*)x, gamma, rows, D);
  else { set_error("rmsnorm: dtype"); return E_INVALID; }
  CHECK_LAUNCH();
}

int glu(void* y, const void* x, int rows, int C, int dt, hipStream_t s) {
  const long n = (long)rows * C;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(glu_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, (const T*)x, n, C));
  CHECK_LAUNCH();
}

int geglu(void* y, const void* x, int rows, int inner, int ldy, int dt, hipStream_t s) {
  const long n = (long)rows * ldy;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(geglu_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, (const T*)x, rows, inner, ldy));
  CHECK_LAUNCH();
}

int dwconv(void* y, const void* x, const float* w, const float* bias, int B, int T_, int C, int k, int dt, hipStream_t s) {
  const long n = (long)B * T_ * C;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(dwconv_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, (const T*)x, w, bias, B, T_, C, k));
  CHECK_LAUNCH();
}

int conv2d_sub2(void* y, const void* mel, const float* w, const float* bias, int B, int F, int idim, int odim, int dt,
                hipStream_t s) {
  ITTS_REQUIRE(F >= 3 && idim >= 3, "conv2d_sub2: input too small");
  const int Fo = (F - 3) / 2 + 1, fo = (idim - 3) / 2 + 1;
  const long n = (long)B * Fo * odim * fo;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(conv2d_sub2_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, (const T*)mel, w, bias, B, F, idim, odim, Fo, fo));
  CHECK_LAUNCH();
}

int gather_add(void* y, int ty, int ldy, const void* ta, const int* ia, const void* tb, const int* ib, int ttab,
               int rows, int D, hipStream_t s) {
  ITTS_REQUIRE(y && rows > 0 && D > 0, "gather_add");
  dim3 grid(rows), blk(256);
#define GA(TT, TO) \
  hipLaunchKernelGGL((gather_add_kernel<TT, TO>), grid, blk, 0, s, (TO*)y, ldy, (const TT*)ta, ia, (const TT*)tb, ib, rows, D)
  if (ttab == F32 && ty == F32) GA(float, float);
  else if (ttab == BF16 && ty == F32) GA(bf16_t, float);
  else if (ttab == BF16 && ty == BF16) GA(bf16_t, bf16_t);
  else if (ttab == F32 && ty == BF16) GA(float, bf16_t);
  else { set_error("gather_add: dtype"); return E_INVALID; }
#undef GA
  CHECK_LAUNCH();
}

int transpose_brc(void* y, const void* x, int B, int R, int C, int dt, hipStream_t s) {
  dim3 grid((C + 31) / 32, (R + 31) / 32, B);
  DISPATCH_T(dt, T, hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, s, (T*)y, (const T*)x, R, C));
  CHECK_LAUNCH();
}

int cast_copy(void* y, int ty, const void* x, int tx, long n, hipStream_t s) {
  dim3 grid(nblk(n)), blk(256);
  if (tx == F32 && ty == BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, blk, 0, s, (bf16_t*)y, (const float*)x, n);
  else if (tx == BF16 && ty == F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, blk, 0, s, (float*)y, (const bf16_t*)x, n);
  else if (tx == F32 && ty == F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, blk, 0, s, (float*)y, (const float*)x, n);
  else if (tx == BF16 && ty == BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, blk, 0, s, (bf16_t*)y, (const bf16_t*)x, n);
  else { set_error("cast_copy: dtype"); return E_INVALID; }
  CHECK_LAUNCH();
}

int copy_rows(void* y, int ldy, const void* x, int ldx, int rows, int D, int dt, hipStream_t s) {
  const long n = (long)rows * D;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, ldy, (const T*)x, ldx, rows, D));
  CHECK_LAUNCH();
}

int add_strided(void* y, int ldy, const void* a, int lda, const void* b, int ldb, int rows, int D, int dt, hipStream_t s) {
  const long n = (long)rows * D;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(add_strided_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, ldy, (const T*)a, lda, (const T*)b, ldb, rows, D));
  CHECK_LAUNCH();
}

int col_mean(float* mean, const void* x, int B, int T_, int C, int ldx, int dt, hipStream_t s) {
  DISPATCH_T(dt, T, hipLaunchKernelGGL((col_stats_kernel<T, false>), dim3(nblk((long)B * C)), dim3(256), 0, s, mean, (const T*)x, B, T_, C, ldx));
  CHECK_LAUNCH();
}

int col_mean_std(float* out, const void* x, int B, int T_, int C, int ldx, int dt, hipStream_t s) {
  DISPATCH_T(dt, T, hipLaunchKernelGGL((col_stats_kernel<T, true>), dim3(nblk((long)B * C)), dim3(256), 0, s, out, (const T*)x, B, T_, C, ldx));
  CHECK_LAUNCH();
}

int scale_cols_add(void* y, int ldy, const void* x, int ldx, const float* sc, const void* res, int ldr, int B, int T_,
                   int C, int dt, hipStream_t s) {
  const long n = (long)B * T_ * C;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(scale_cols_add_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, ldy, (const T*)x, ldx, sc, (const T*)res, ldr, B, T_, C));
  CHECK_LAUNCH();
}

int asp_pool(float* out, const void* logits, const void* x, const float* bn_scale, const float* bn_shift, int B, int T_,
             int C, int dt, hipStream_t s) {
  DISPATCH_T(dt, T, hipLaunchKernelGGL(asp_pool_kernel<T>, dim3(nblk((long)B * C)), dim3(256), 0, s, out, (const T*)logits, (const T*)x, bn_scale, bn_shift, B, T_, C));
  CHECK_LAUNCH();
}

int relpos_pack(void* qc, void* kc, const void* qkv, const void* p, const float* bu, const float* bv, int T_, int H,
                int dk, int dt, hipStream_t s) {
  const long n = (long)T_ * H * dk;
  DISPATCH_T(dt, T, hipLaunchKernelGGL(relpos_pack_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)qc, (T*)kc, (const T*)qkv, (const T*)p, bu, bv, T_, H, dk));
  CHECK_LAUNCH();
}

int tanh_rows(void* y, const void* x, long n, int dt, hipStream_t s) {
  DISPATCH_T(dt, T, hipLaunchKernelGGL(tanh_kernel<T>, dim3(nblk(n)), dim3(256), 0, s, (T*)y, (const T*)x, n));
  CHECK_LAUNCH();
}

}  // namespace itts
