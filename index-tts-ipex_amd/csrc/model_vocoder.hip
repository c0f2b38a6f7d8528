// BigVGAN2 generator, ECAPA-TDNN speaker encoder, DVAE decoder - all on channels-last activations.
// Reference: BigVGAN.forward / AMPBlock1.forward (indextts/BigVGAN/models.py:65-74,201-250), Activation1d
// (alias_free_torch/act.py, resample.py, filter.py), ECAPA_TDNN.forward (BigVGAN/ECAPA_TDNN.py:545-581),
// DiscreteVAE.decode (vqvae/xtts_dvae.py:332-351).
#include <cmath>

#include "engine.h"

namespace itts {

namespace {

// out[b, n] = a[b, n] + v[n]
__global__ void add_vec_rows_kernel(float* __restrict__ out, const float* __restrict__ a, const float* __restrict__ v,
                                    int B, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * N) out[i] = a[i] + v[i % N];
}

template <typename T>
__global__ void embed_codes_kernel(T* __restrict__ y, const T* __restrict__ table, const int* __restrict__ codes, int D) {
  const int r = blockIdx.x;
  const int c = codes[r];
  for (int i = threadIdx.x; i < D; i += blockDim.x) y[(size_t)r * D + i] = table[(size_t)c * D + i];
}

// rows paired for a stride-2 convolution: y[b, t, :] = [x[b, 2t, :], x[b, 2t + 1, :]] (zero beyond the end)
template <typename T>
__global__ void pair_rows_kernel(T* __restrict__ y, const T* __restrict__ x, int Tin, int Tout, int C) {
  const int r = blockIdx.x, b = r / Tout, t = r % Tout;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    const int tt = 2 * t + i / C;
    y[(size_t)r * 2 * C + i] = tt < Tin ? x[((size_t)b * Tin + tt) * C + (i % C)] : (T)0.f;
  }
}

// Quantize.forward's code choice (xtts_dvae.py:86-89): argmax_n -(|x|^2 - 2 x.e_n + |e_n|^2) = argmin_n (|e_n|^2 - 2 x.e_n),
// the first index on ties
__global__ __launch_bounds__(256) void dvae_argmin_kernel(int* __restrict__ codes, const float* __restrict__ dots,
                                                          const float* __restrict__ esq, int N) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const int r = blockIdx.x, tid = threadIdx.x;
  float best = INFINITY;
  int bi = 0x7fffffff;
  for (int n = tid; n < N; n += 256) {
    const float d = esq[n] - 2.f * dots[(size_t)r * N + n];
    if (d < best || (d == best && n < bi)) {
      best = d;
      bi = n;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov < best || (ov == best && oi < bi)) {
      best = ov;
      bi = oi;
    }
  }
  if ((tid & 63) == 0) {
    sv[tid >> 6] = best;
    si[tid >> 6] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] < best || (sv[w] == best && si[w] < bi)) {
        best = sv[w];
        bi = si[w];
      }
    codes[r] = bi;
  }
}

}  // namespace

#define K(call)               \
  do {                        \
    if (!dry) ITTS_TRY(call); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// ECAPA-TDNN
// ------------------------------------------------------------------------------------------------
int Engine::ecapa(const void* mel, int B, int F, float* spk_out, hipStream_t s) {
  if (!finalized || !ec.ok) {
    set_error("ecapa: speaker-encoder weights not bound");
    return E_STATE;
  }
  const itts_config& c = cfg;
  ITTS_REQUIRE(mel && spk_out && B > 0, "ecapa: bad arguments");
  const int* chs = c.ec_channels;
  const int maxdil = std::max(std::max(c.ec_dils[1], c.ec_dils[2]), c.ec_dils[3]);
  ITTS_REQUIRE(F > std::max(c.ec_kernels[0] / 2, maxdil), "ecapa: prompt shorter than the reflect padding");
  const int M = B * F, C = chs[0], hc = C / c.ec_scale, CC = chs[3] * 3, C4 = chs[4], att = c.ec_att;
  auto tdnn = [&](void* out, int ldc, const void* in, int lda, const Lin& w, int dil) -> int {
    GemmArgs g;
    g.A = in;
    g.W = w.w;
    g.C = out;
    g.M = M;
    g.N = w.N;
    g.Cin = w.Cin;
    g.taps = w.taps;
    g.lda = lda;
    g.ldc = ldc;
    g.T = F;
    g.dil = dil;
    g.pad_left = dil * (w.taps - 1) / 2;
    g.pad_mode = PAD_REFLECT;
    g.bias = w.b;
    g.act = ACT_RELU;
    g.scale = w.bn_scale;
    g.shift = w.bn_shift;
    return conv(g, adt, w.dt, adt, s);
  };
  auto body = [&]() -> int {
    void* x0 = alloc((size_t)M * C * es);
    const int melp = (c.bv_num_mels + 7) / 8 * 8;  // 100 mel bins -> 104 zero-padded channels: 16-byte rows for the MFMA conv
    const void* mel_in = mel;
    if (melp != c.bv_num_mels) {
      void* mp = alloc((size_t)M * melp * es);
      if (!dry) ITTS_HIP_CHECK(hipMemsetAsync(mp, 0, (size_t)M * melp * es, s));
      K(copy_rows(mp, melp, mel, c.bv_num_mels, M, c.bv_num_mels, adt, s));
      mel_in = mp;
    }
    ITTS_TRY(tdnn(x0, C, mel_in, melp, ec.b0, c.ec_dils[0]));
    void* cat = alloc((size_t)M * CC * es);  // outputs of the three SE-Res2Net blocks, concatenated
    void* y1 = alloc((size_t)M * C * es);
    void* y2 = alloc((size_t)M * C * es);
    void* y3 = alloc((size_t)M * C * es);
    void* tmp = alloc((size_t)M * hc * es);
    float* sm = (float*)alloc((size_t)B * C * 4);
    float* s1 = (float*)alloc((size_t)B * c.ec_se * 4);
    float* s2 = (float*)alloc((size_t)B * C * 4);
    const void* xin = x0;
    int ldin = C;
    for (int i = 0; i < 3; ++i) {
      const EcapaW::Blk& Bk = ec.blks[i];
      const int dil = c.ec_dils[i + 1];
      ITTS_TRY(tdnn(y1, C, xin, ldin, Bk.tdnn1, 1));
      // Res2Net: chunk 0 passes through, chunk j = tdnn(x_j + y_{j-1})   (ECAPA_TDNN.py:172-191)
      K(copy_rows(y2, C, y1, C, M, hc, adt, s));
      for (int j = 1; j < c.ec_scale; ++j) {
        const char* xj = (const char*)y1 + (size_t)j * hc * es;
        char* yj = (char*)y2 + (size_t)j * hc * es;
        if (j == 1) {
          ITTS_TRY(tdnn(yj, C, xj, C, Bk.res[j - 1], dil));
        } else {
          K(add_strided(tmp, hc, xj, C, yj - (size_t)hc * es, C, M, hc, adt, s));
          ITTS_TRY(tdnn(yj, C, tmp, hc, Bk.res[j - 1], dil));
        }
      }
      ITTS_TRY(tdnn(y3, C, y2, C, Bk.tdnn2, 1));
      // SE block (:223-242): mean over time -> conv1 relu -> conv2 sigmoid -> gate, + block residual
      K(col_mean(sm, y3, B, F, C, C, adt, s));
      ITTS_TRY(lin(s1, F32, sm, F32, C, Bk.se1, B, c.ec_se, s, ACT_RELU));
      ITTS_TRY(lin(s2, F32, s1, F32, c.ec_se, Bk.se2, B, C, s, ACT_SIGMOID));
      char* xo = (char*)cat + (size_t)i * C * es;
      K(scale_cols_add(xo, CC, y3, C, s2, xin, ldin, B, F, C, adt, s));
      xin = xo;
      ldin = CC;
    }
    void* mf = alloc((size_t)M * C4 * es);
    ITTS_TRY(tdnn(mf, C4, cat, CC, ec.mfa, 1));
    // attentive statistics pooling with global context (:283-338)
    float* ms = (float*)alloc((size_t)B * 2 * C4 * 4);
    K(col_mean_std(ms, mf, B, F, C4, C4, adt, s));
    float* gb = (float*)alloc((size_t)B * att * 4);  // per-batch bias = W_ms [mean;std] + b
    ITTS_TRY(lin(gb, F32, ms, F32, 2 * C4, ec.asp_ms, B, att, s));
    void* a1 = alloc((size_t)M * att * es);
    {
      GemmArgs g;
      g.A = mf;
      g.W = ec.asp_x.w;
      g.C = a1;
      g.M = M;
      g.N = att;
      g.Cin = C4;
      g.lda = C4;
      g.ldc = att;
      g.T = F;
      g.bias = gb;
      g.bias_bstride = att;
      g.act = ACT_RELU;
      g.scale = ec.asp_x.bn_scale;
      g.shift = ec.asp_x.bn_shift;
      g.act2 = ACT_TANH;
      ITTS_TRY(conv(g, adt, ec.asp_x.dt, adt, s));
    }
    void* a2 = alloc((size_t)M * C4 * es);
    ITTS_TRY(lin(a2, adt, a1, adt, att, ec.asp_conv, M, C4, s));
    float* pooled = (float*)alloc((size_t)B * 2 * C4 * 4);
    K(asp_pool(pooled, a2, mf, ec.aspbn_scale, ec.aspbn_shift, B, F, C4, adt, s));
    ITTS_TRY(lin(spk_out, F32, pooled, F32, 2 * C4, ec.fc, B, c.bv_spk_dim, s));
    return OK;
  };
  ArenaSwap arena(*this);  // its own scratch: may run beside this engine's work on another stream
  return two_pass(body, s);
}

// ------------------------------------------------------------------------------------------------
// BigVGAN generator
// ------------------------------------------------------------------------------------------------
int Engine::bigvgan(const void* latent, const float* spk, int B, int T, float* wav, hipStream_t s) {
  if (!finalized || !bv.ok) {
    set_error("bigvgan: generator weights not bound");
    return E_STATE;
  }
  const itts_config& c = cfg;
  ITTS_REQUIRE(latent && spk && wav && B > 0 && T > 0, "bigvgan: bad arguments");
  const int C0 = c.bv_init_ch, E = c.bv_spk_dim, nk = c.bv_num_res, nd = c.bv_num_dil;
  auto body = [&]() -> int {
    // speaker conditioning -> per-batch biases (cond_layer / conds[i] are 1x1 convs of a length-1 signal)
    float* cb = (float*)alloc((size_t)B * C0 * 4);
    float* cbias = (float*)alloc((size_t)B * C0 * 4);
    ITTS_TRY(lin(cb, F32, spk, F32, E, bv.cond_layer, B, C0, s));
    if (!dry) {
      hipLaunchKernelGGL(add_vec_rows_kernel, dim3((B * C0 + 255) / 256), dim3(256), 0, s, cbias, cb, bv.conv_pre.b, B, C0);
      ITTS_HIP_CHECK(hipGetLastError());
    }
    // conv_pre (k7, pad 3) + cond
    void* x = alloc((size_t)B * T * C0 * es);
    {
      GemmArgs g;
      g.A = latent;
      g.W = bv.conv_pre.w;
      g.C = x;
      g.M = B * T;
      g.N = C0;
      g.Cin = c.bv_gpt_dim;
      g.taps = 7;
      g.lda = c.bv_gpt_dim;
      g.ldc = C0;
      g.T = T;
      g.pad_left = 3;
      g.bias = cbias;
      g.bias_bstride = C0;
      ITTS_TRY(conv(g, adt, bv.conv_pre.dt, adt, s));
    }
    ITTS_TRY(tap("bv_pre", x, adt, (int64_t)B * T * C0, s));
    int Tc = T, ch = C0;
    for (int i = 0; i < c.bv_num_up; ++i) {
      const int u = c.bv_up_rates[i], k = c.bv_up_kernels[i], p = (k - u) / 2;
      const int cout = ch / 2, To = Tc * u;
      const Lin& up = bv.ups[i];
      float* ub = (float*)alloc((size_t)B * cout * 4);
      float* ubias = (float*)alloc((size_t)B * cout * 4);
      ITTS_TRY(lin(ub, F32, spk, F32, E, bv.conds[i], B, cout, s));
      if (!dry) {
        hipLaunchKernelGGL(add_vec_rows_kernel, dim3((B * cout + 255) / 256), dim3(256), 0, s, ubias, ub, up.b, B, cout);
        ITTS_HIP_CHECK(hipGetLastError());
      }
      // ConvTranspose1d as u polyphase GEMMs: out[q*u + ph] = sum_m x[q + floor((ph+p)/u) - m] . W[ph][m]
      void* xu = alloc((size_t)B * To * cout * es);
      {
        GemmArgs g;
        g.A = x;
        g.W = up.w;
        g.C = xu;
        g.M = B * Tc;
        g.N = cout;
        g.Cin = ch;
        g.taps = k / u;
        g.lda = ch;
        g.ldc = u * cout;
        g.T = Tc;
        g.dil = -1;
        g.pad_left = 0;
        g.nphase = u;
        for (int ph = 0; ph < u; ++ph) g.phase_shift[ph] = (ph + p) / u;
        g.bias = ubias;
        g.bias_bstride = cout;
        ITTS_TRY(conv(g, adt, up.dt, adt, s));
      }
      if (i == 0) ITTS_TRY(tap("bv_up0", xu, adt, (int64_t)B * To * cout, s));
      Tc = To;
      ch = cout;
      const size_t nel = (size_t)B * Tc * ch;
      void* xs = alloc(nel * es);
      void* xa = alloc(nel * es);
      void* xb = alloc(nel * es);
      void* t1 = alloc(nel * es);
      void* t2 = alloc(nel * es);
      // a -> conv (AMPBlock1, models.py:65-74): on the narrow stages the convolution applies Activation1d itself while it stages its
      // input tile (conv_lds.hip ACT), elsewhere the activation is its own launch into `tact`
      auto act_conv = [&](GemmArgs& q, const void* xraw, const float* la, const float* lb, void* tact, int wdt) -> int {
        q.A = xraw;
        q.pre_alpha = la;
        q.pre_beta = lb;
        q.pre_filt = bv.filter;
        if (!force_simple && gemm_which(q, adt, wdt, adt) == 4 && conv_lds_act_supported(q, adt, wdt, adt)) return conv(q, adt, wdt, adt, s);
        q.pre_alpha = q.pre_beta = q.pre_filt = nullptr;
        K(snake_aa(tact, xraw, la, lb, bv.filter, bv.filter, B, Tc, ch, adt, s));
        q.A = tact;
        return conv(q, adt, wdt, adt, s);
      };
      for (int j = 0; j < nk; ++j) {
        const AmpW& A = bv.res[i * nk + j];
        const int ks = c.bv_res_kernels[j];
        const void* cur = xu;
        for (int l = 0; l < nd; ++l) {
          const int d = c.bv_res_dils[j][l];
          GemmArgs g;
          g.W = A.c1[l].w;
          g.C = t2;
          g.M = B * Tc;
          g.N = ch;
          g.Cin = ch;
          g.taps = ks;
          g.lda = g.ldc = ch;
          g.T = Tc;
          g.dil = d;
          g.pad_left = (ks * d - d) / 2;
          g.bias = A.c1[l].b;
          ITTS_TRY(act_conv(g, cur, A.a1[l], A.b1[l], t1, A.c1[l].dt));
          GemmArgs h;
          h.W = A.c2[l].w;
          h.M = B * Tc;
          h.N = ch;
          h.Cin = ch;
          h.taps = ks;
          h.lda = h.ldc = ch;
          h.T = Tc;
          h.dil = 1;
          h.pad_left = (ks - 1) / 2;
          h.bias = A.c2[l].b;
          h.R = cur;
          h.ldr = ch;
          const bool last = l == nd - 1;
          if (last) {
            // x = (sum_j AMP_j(x)) / nk  (models.py:236-243), accumulated in the epilogue
            h.C = xs;
            h.alpha = 1.f / nk;
            if (j > 0) {
              h.ADD = xs;
              h.ldadd = ch;
              h.beta = 1.f;
            }
          } else {
            h.C = (cur == xa) ? xb : xa;
          }
          ITTS_TRY(act_conv(h, t2, A.a2[l], A.b2[l], t1, A.c2[l].dt));
          cur = h.C;
          if (i == 0 && j == 0 && l == nd - 1 && nk == 1) ITTS_TRY(tap("bv_amp0", xs, adt, (int64_t)nel, s));
        }
      }
      ITTS_TRY(tap(("bv_stage" + std::to_string(i)).c_str(), xs, adt, (int64_t)nel, s));
      x = xs;
    }
    void* t1 = alloc((size_t)B * Tc * ch * es);
    K(snake_aa(t1, x, bv.post_alpha, bv.post_beta, bv.filter, bv.filter, B, Tc, ch, adt, s));
    {
      GemmArgs g;
      g.A = t1;
      g.W = bv.conv_post.w;
      g.C = wav;
      g.M = B * Tc;
      g.N = 1;
      g.Cin = ch;
      g.taps = 7;
      g.lda = ch;
      g.ldc = 1;
      g.T = Tc;
      g.pad_left = 3;
      g.bias = bv.conv_post.b;
      g.act = ACT_TANH;
      ITTS_TRY(conv(g, adt, bv.conv_post.dt, F32, s));
    }
    return OK;
  };
  return two_pass(body, s);
}

// ------------------------------------------------------------------------------------------------
// DVAE decoder
// ------------------------------------------------------------------------------------------------
int Engine::dvae_decode(const int32_t* codes_host, int B, int T, void* mel_out, hipStream_t s) {
  if (!finalized || !dv.ok) {
    set_error("dvae_decode: DVAE weights not bound");
    return E_STATE;
  }
  const itts_config& c = cfg;
  ITTS_REQUIRE(codes_host && mel_out && B > 0 && T > 0, "dvae_decode: bad arguments");
  for (long i = 0; i < (long)B * T; ++i)
    ITTS_REQUIRE(codes_host[i] >= 0 && codes_host[i] < c.dv_tokens, "dvae_decode: code out of range");
  const int inner = c.dv_hidden << (c.dv_layers - 1), cb = c.dv_codebook;
  auto conv1d = [&](void* out, int tc, const void* in, const Lin& w, int Tn, int act, const void* R, int up) -> int {
    GemmArgs g;
    g.A = in;
    g.W = w.w;
    g.C = out;
    g.M = B * Tn;
    g.N = w.N;
    g.Cin = w.Cin;
    g.taps = w.taps;
    g.lda = w.Cin;
    g.ldc = w.N;
    g.T = Tn;
    g.pad_left = (w.taps - 1) / 2;
    g.in_up = up;
    g.bias = w.b;
    g.act = act;
    g.R = R;
    g.ldr = w.N;
    return conv(g, adt, w.dt, tc, s);
  };
  auto body = [&]() -> int {
    int* codes_dev = (int*)alloc((size_t)B * T * 4);
    void* e0 = alloc((size_t)B * T * cb * es);
    if (!dry) {
      ITTS_HIP_CHECK(hipMemcpyAsync(codes_dev, codes_host, (size_t)B * T * 4, hipMemcpyHostToDevice, s));
      if (adt == F32)
        hipLaunchKernelGGL(embed_codes_kernel<float>, dim3(B * T), dim3(128), 0, s, (float*)e0, (const float*)dv.codebook, codes_dev, cb);
      else
        hipLaunchKernelGGL(embed_codes_kernel<bf16_t>, dim3(B * T), dim3(128), 0, s, (bf16_t*)e0, (const bf16_t*)dv.codebook, codes_dev, cb);
      ITTS_HIP_CHECK(hipGetLastError());
    }
    void* x = alloc((size_t)B * T * inner * es);
    void* y1 = alloc((size_t)B * T * inner * es);
    void* y2 = alloc((size_t)B * T * inner * es);
    ITTS_TRY(conv1d(x, adt, e0, dv.in_conv, T, ACT_NONE, nullptr, 1));
    for (int i = 0; i < c.dv_resblocks; ++i) {
      const DvaeW::RB& rb = dv.rbs[i];
      ITTS_TRY(conv1d(y1, adt, x, rb.c0, T, ACT_RELU, nullptr, 1));
      ITTS_TRY(conv1d(y2, adt, y1, rb.c2, T, ACT_RELU, nullptr, 1));
      ITTS_TRY(conv1d(x, adt, y2, rb.c4, T, ACT_NONE, x, 1));
    }
    int Tn = T;
    const void* cur = x;
    for (int i = 0; i < c.dv_layers; ++i) {
      Tn *= 2;
      void* o = alloc((size_t)B * Tn * dv.ups[i].N * es);
      ITTS_TRY(conv1d(o, adt, cur, dv.ups[i], Tn, ACT_RELU, nullptr, 2));
      cur = o;
    }
    ITTS_TRY(conv1d(mel_out, adt, cur, dv.out_conv, Tn, ACT_NONE, nullptr, 1));
    return OK;
  };
  ITTS_TRY(two_pass(body, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  return OK;
}

// ------------------------------------------------------------------------------------------------
// DVAE encoder: DiscreteVAE.get_codebook_indices (vqvae/xtts_dvae.py:325-330) = encoder (strided convs + ReLU,
// ResBlocks, 1x1 conv, :251-291) then Quantize.forward's nearest-code search (:86-92).  mel [B, T, channels] in the
// engine dtype -> codes int32 [B, T'] on the host, T' = ceil(ceil(T / 2) / 2) for the two stride-2 layers.
// ------------------------------------------------------------------------------------------------
int Engine::dvae_encode(const void* mel, int B, int T, int32_t* codes_host, hipStream_t s) {
  if (!finalized || !dv.ok || !dv.enc_ok) {
    set_error("dvae_encode: DVAE encoder weights not bound");
    return E_STATE;
  }
  const itts_config& c = cfg;
  ITTS_REQUIRE(mel && codes_host && B > 0 && T > 0, "dvae_encode: bad arguments");
  const int inner = c.dv_hidden << (c.dv_layers - 1);
  auto conv1d = [&](void* out, int tc, const void* in, const Lin& w, int Tn, int act, const void* R, int pad_left) -> int {
    GemmArgs g;
    g.A = in;
    g.W = w.w;
    g.C = out;
    g.M = B * Tn;
    g.N = w.N;
    g.Cin = w.Cin;
    g.taps = w.taps;
    g.lda = w.Cin;
    g.ldc = w.N;
    g.T = Tn;
    g.pad_left = pad_left;
    g.bias = w.b;
    g.act = act;
    g.R = R;
    g.ldr = w.N;
    return conv(g, adt, w.dt, tc, s);
  };
  int Tout = T;
  for (int i = 0; i < c.dv_layers; ++i) Tout = (Tout + 1) / 2;
  auto body = [&]() -> int {
    const void* cur = mel;
    int Tn = T, C = c.dv_channels;
    for (int i = 0; i < c.dv_layers; ++i) {
      const int Tp = (Tn + 1) / 2;
      void* pr = alloc((size_t)B * Tp * 2 * C * es);
      if (!dry) {
        if (adt == F32)
          hipLaunchKernelGGL(pair_rows_kernel<float>, dim3(B * Tp), dim3(256), 0, s, (float*)pr, (const float*)cur, Tn, Tp, C);
        else
          hipLaunchKernelGGL(pair_rows_kernel<bf16_t>, dim3(B * Tp), dim3(256), 0, s, (bf16_t*)pr, (const bf16_t*)cur, Tn, Tp, C);
        ITTS_HIP_CHECK(hipGetLastError());
      }
      void* o = alloc((size_t)B * Tp * dv.enc[i].N * es);
      ITTS_TRY(conv1d(o, adt, pr, dv.enc[i], Tp, ACT_RELU, nullptr, 1));  // taps (pair t-1, pair t)
      cur = o;
      Tn = Tp;
      C = dv.enc[i].N;
    }
    void* x = const_cast<void*>(cur);
    void* y1 = alloc((size_t)B * Tn * inner * es);
    void* y2 = alloc((size_t)B * Tn * inner * es);
    for (int i = 0; i < c.dv_resblocks; ++i) {
      const DvaeW::RB& rb = dv.erbs[i];
      ITTS_TRY(conv1d(y1, adt, x, rb.c0, Tn, ACT_RELU, nullptr, 1));
      ITTS_TRY(conv1d(y2, adt, y1, rb.c2, Tn, ACT_RELU, nullptr, 1));
      ITTS_TRY(conv1d(x, adt, y2, rb.c4, Tn, ACT_NONE, x, 0));
    }
    void* z = alloc((size_t)B * Tn * c.dv_codebook * es);
    ITTS_TRY(conv1d(z, adt, x, dv.eout, Tn, ACT_NONE, nullptr, 0));
    float* dots = (float*)alloc((size_t)B * Tn * c.dv_tokens * 4);
    ITTS_TRY(conv1d(dots, F32, z, dv.quant, Tn, ACT_NONE, nullptr, 0));
    int* codes_dev = (int*)alloc((size_t)B * Tn * 4);
    if (!dry) {
      hipLaunchKernelGGL(dvae_argmin_kernel, dim3(B * Tn), dim3(256), 0, s, codes_dev, dots, dv.codebook_sq, c.dv_tokens);
      ITTS_HIP_CHECK(hipGetLastError());
      ITTS_HIP_CHECK(hipMemcpyAsync(codes_host, codes_dev, (size_t)B * Tn * 4, hipMemcpyDeviceToHost, s));
    }
    return OK;
  };
  ITTS_TRY(two_pass(body, s));
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  (void)Tout;
  return OK;
}

}  // namespace itts
