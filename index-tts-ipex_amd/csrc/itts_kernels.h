// Host-side launchers of every HIP kernel in libitts_hip.  All tensors are channels-last / row-major,
// all launches are asynchronous on the given stream, no launcher allocates or synchronises
// (graph-capture safe).  dtype arguments are itts::DType values.
#pragma once
#include "itts_common.h"

namespace itts {

// ---- GEMM / conv family (gemm_simple.hip, gemm_mfma.hip) ----
int gemm_simple(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s);
bool gemm_mfma_supported(const GemmArgs& g, int ta, int tw, int tc);
int gemm_mfma(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s);
bool gemm_glds_supported(const GemmArgs& g, int ta, int tw, int tc);  // LDS-DMA staged 128 x 128 / 128 x 64 tiles (gemm_glds.hip)
int gemm_glds(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s);
long gemm_p8_tiles(const GemmArgs& g, int ta, int tw, int tc);      // 0 = cannot run the shape
bool gemm_p8_supported(const GemmArgs& g, int ta, int tw, int tc);
int gemm_ksplit_plan(const GemmArgs& g, int ta, int tw, int tc, size_t ws_bytes);  // K split of the few-tile deep-K shapes (c_api.cpp); 1 = none  // 256 x 256 tile, 8-phase LDS-DMA pipeline (gemm_p8.hip)
int gemm_p8(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s);
bool conv_lds_supported(const GemmArgs& g, int ta, int tw, int tc);
bool conv_lds_act_supported(const GemmArgs& g, int ta, int tw, int tc);  // ... with Activation1d fused into the tile load (g.pre_*)
int conv_lds(const GemmArgs& g, hipStream_t s);
int gemm(const GemmArgs& g, int ta, int tw, int tc, hipStream_t s);  // dispatcher
int gemm_which(const GemmArgs& g, int ta, int tw, int tc);           // the kernel family the dispatcher takes: 0 VALU, 1 mfma, 2 glds, 3 p8, 4 conv_lds

// ---- anti-aliased SnakeBeta (snake.hip); x,y [B,T,C] ----
int snake_aa(void* y, const void* x, const float* log_alpha, const float* log_beta, const float* up12,
             const float* down12, int B, int T, int C, int dt, hipStream_t s);
// same op on the reference's [B,C,T] layout (drop-in for anti_alias_activation_cuda.forward)
int snake_aa_bct(void* y, const void* x, const float* log_alpha, const float* log_beta, const float* up12,
                 const float* down12, int B, int C, int T, int dt, hipStream_t s);

// ---- row-wise ops (elementwise.hip) ----
int layernorm(void* y, int ty, const void* x, int tx, const float* gamma, const float* beta, int rows, int D,
              int ldx, int ldy, float eps, int act, hipStream_t s);
int rmsnorm_unit(void* y, int ty, const void* x, int tx, const float* gamma, int rows, int D, hipStream_t s);
int glu(void* y, const void* x, int rows, int C, int dt, hipStream_t s);
int geglu(void* y, const void* x, int rows, int inner, int ldy, int dt, hipStream_t s);
int dwconv(void* y, const void* x, const float* w, const float* bias, int B, int T, int C, int k, int dt, hipStream_t s);
int conv2d_sub2(void* y, const void* mel, const float* w, const float* bias, int B, int F, int idim, int odim, int dt,
                hipStream_t s);
int gather_add(void* y, int ty, int ldy, const void* ta, const int* ia, const void* tb, const int* ib, int ttab,
               int rows, int D, hipStream_t s);
int transpose_brc(void* y, const void* x, int B, int R, int C, int dt, hipStream_t s);
int cast_copy(void* y, int ty, const void* x, int tx, long n, hipStream_t s);
int copy_rows(void* y, int ldy, const void* x, int ldx, int rows, int D, int dt, hipStream_t s);
int add_strided(void* y, int ldy, const void* a, int lda, const void* b, int ldb, int rows, int D, int dt, hipStream_t s);
int col_mean(float* mean, const void* x, int B, int T, int C, int ldx, int dt, hipStream_t s);
int col_mean_std(float* out, const void* x, int B, int T, int C, int ldx, int dt, hipStream_t s);
int scale_cols_add(void* y, int ldy, const void* x, int ldx, const float* sc, const void* res, int ldr, int B, int T,
                   int C, int dt, hipStream_t s);
int asp_pool(float* out, const void* logits, const void* x, const float* bn_scale, const float* bn_shift, int B, int T,
             int C, int dt, hipStream_t s);
int relpos_pack(void* qc, void* kc, const void* qkv, const void* p, const float* bu, const float* bv, int T, int H,
                int dk, int dt, hipStream_t s);
int tanh_rows(void* y, const void* x, long n, int dt, hipStream_t s);

// ---- attention (attention.hip) ----
struct AttnArgs {
  const void* q = nullptr;  // [B, Sq] rows, element (b,i,h,d) at q[(b*Sq+i)*ldq + h*dqk + d]
  const void* k = nullptr;  // (b,j,h,d) at k[(b*Sk+j)*ldk + h*dqk + d]
  const void* v = nullptr;  // (b,j,h,d) at v[(b*Sk+j)*ldv + h*dv + d]
  void* o = nullptr;        // (b,i,h,d) at o[(b*Sq+i)*ldo + h*dv + d]
  int B = 1, H = 1, Sq = 0, Sk = 0, dqk = 64, dv = 64;
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  float scale = 1.f;
  int causal = 0;                 // key j visible to query i iff j <= i + (Sk - Sq)
  const int* kv_start = nullptr;  // per batch row: keys < kv_start[b] are masked (left padding)
};
int attention_simple(const AttnArgs& a, int dt, hipStream_t s);
bool attention_mfma_supported(const AttnArgs& a, int dt);
int attention_mfma(const AttnArgs& a, int dt, hipStream_t s);
int attention(const AttnArgs& a, int dt, hipStream_t s);  // dispatcher

}  // namespace itts
