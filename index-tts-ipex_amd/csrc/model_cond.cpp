// Speaker conditioning: ConformerEncoder (6 blocks) + PerceiverResampler -> 32 latents.
// Reference: UnifiedVoice.get_conditioning (indextts/gpt/model.py:490-502), ConformerEncoder
// (gpt/conformer_encoder.py:232-313,400-436; conformer/subsampling.py:135-186; conformer/attention.py:235-312),
// PerceiverResampler (gpt/perceiver.py:263-317).  Runs once per prompt; the result is cached by the host
// across sentences (the reference recomputes it twice per sentence, model.py:540,670).
#include <cmath>

#include "engine.h"

namespace itts {

#define K(call)                  \
  do {                           \
    if (!dry) ITTS_TRY(call);    \
  } while (0)

// F_total > F: the prompt is the first F frames of a tensor padded to F_total frames (get_conditioning with cond_mel_lengths,
// model.py:490-502).  The subsampled mask keeps row i iff 2 i + 2 < F (subsampling.py:186), i.e. exactly the rows a valid
// 3 x 3 stride-2 convolution of the first F frames has, each computed from valid frames only; masked keys (attention.py:104-117)
// are keys the shorter sequence does not have.  One thing differs from a plain cut: the convolution module zero-fills the
// masked ROWS BEFORE its first pointwise convolution (conformer_encoder.py:143-146), so behind the end of the prompt its depthwise
// convolution (k = 15, pad 7) sees GLU(pw1 bias), not zeros - reproduced by running pw1 / GLU / the depthwise convolution over
// min(7, masked rows) extra zero rows.
int Engine::conditioning(const void* mel, int F, float* cond_out, hipStream_t s, int F_total) {
  if (!finalized || !cond.ok) {
    set_error("conditioning: conformer/perceiver weights not bound");
    return E_STATE;
  }
  ITTS_REQUIRE(mel && cond_out && F >= 3, "conditioning: bad arguments");
  const itts_config& c = cfg;
  const int od = c.cond_dim, idim = c.cond_idim, H = c.cond_heads, dk = od / H, D = c.model_dim;
  const int Fo = (F - 3) / 2 + 1, fo = (idim - 3) / 2 + 1, nl = c.cond_latents;
  ITTS_REQUIRE(Fo <= cond.pe_len, "conditioning: prompt longer than the positional table");
  ITTS_REQUIRE(F_total == 0 || F_total >= F, "conditioning: padded length shorter than the prompt");
  const int tail = F_total > F ? std::min(7, ((F_total - 3) / 2 + 1) - Fo) : 0;  // masked rows the depthwise convolution reaches
  ITTS_REQUIRE(2 * dk <= 128 && dk <= 64, "conditioning: head dim too large");
  auto body = [&]() -> int {
    void* sub = alloc((size_t)Fo * od * fo * es);
    K(conv2d_sub2(sub, mel, cond.conv_w, cond.conv_b, 1, F, idim, od, adt, s));
    void* x = alloc((size_t)Fo * od * es);
    ITTS_TRY(lin(x, adt, sub, adt, od * fo, cond.embed_out, Fo, od, s, ACT_NONE, nullptr, 0, std::sqrt((float)od)));
    void* xn = alloc((size_t)(Fo + tail) * od * es);
    if (tail > 0 && !dry) ITTS_HIP_CHECK(hipMemsetAsync((char*)xn + (size_t)Fo * od * es, 0, (size_t)tail * od * es, s));  // zero-filled masked rows
    void* qkv = alloc((size_t)Fo * 3 * od * es);
    void* pp = alloc((size_t)Fo * od * es);
    void* qc = alloc((size_t)Fo * 2 * od * es);
    void* kc = alloc((size_t)Fo * 2 * od * es);
    void* ctx = alloc((size_t)Fo * od * es);
    void* g1 = alloc((size_t)(Fo + tail) * 2 * od * es);
    void* g2 = alloc((size_t)(Fo + tail) * od * es);
    void* g3 = alloc((size_t)(Fo + tail) * od * es);
    void* f1 = alloc((size_t)Fo * c.cond_ff * es);
    for (int i = 0; i < c.cond_blocks; ++i) {
      const ConformerLayerW& L = cond.layers[i];
      // --- rel-pos MHA ---
      ITTS_TRY(ln(xn, adt, x, adt, L.norm_mha, Fo, od, s));
      ITTS_TRY(lin(qkv, adt, xn, adt, od, L.qkv, Fo, 3 * od, s));
      ITTS_TRY(lin(pp, adt, cond.pe, adt, od, L.pos, Fo, od, s));
      K(relpos_pack(qc, kc, qkv, pp, L.bu, L.bv, Fo, H, dk, adt, s));
      AttnArgs a;
      a.q = qc;
      a.k = kc;
      a.v = (const char*)qkv + (size_t)2 * od * es;
      a.o = ctx;
      a.B = 1;
      a.H = H;
      a.Sq = a.Sk = Fo;
      a.dqk = 2 * dk;
      a.dv = dk;
      a.ldq = a.ldk = 2 * od;
      a.ldv = 3 * od;
      a.ldo = od;
      a.scale = 1.f / std::sqrt((float)dk);
      K(force_simple ? attention_simple(a, adt, s) : attention(a, adt, s));
      ITTS_TRY(lin(x, adt, ctx, adt, od, L.out, Fo, od, s, ACT_NONE, x, od));
      // --- convolution module ---
      ITTS_TRY(ln(xn, adt, x, adt, L.norm_conv, Fo, od, s));
      ITTS_TRY(lin(g1, adt, xn, adt, od, L.pw1, Fo + tail, 2 * od, s));
      K(glu(g2, g1, Fo + tail, od, adt, s));
      K(dwconv(g3, g2, L.dw_w, L.dw_b, 1, Fo + tail, od, 15, adt, s));
      ITTS_TRY(ln(g2, adt, g3, adt, L.conv_norm, Fo, od, s, ACT_SILU));
      ITTS_TRY(lin(x, adt, g2, adt, od, L.pw2, Fo, od, s, ACT_NONE, x, od));
      // --- feed forward ---
      ITTS_TRY(ln(xn, adt, x, adt, L.norm_ff, Fo, od, s));
      ITTS_TRY(lin(f1, adt, xn, adt, od, L.w1, Fo, c.cond_ff, s, ACT_SILU));
      ITTS_TRY(lin(x, adt, f1, adt, c.cond_ff, L.w2, Fo, od, s, ACT_NONE, x, od));
      ITTS_TRY(ln(x, adt, x, adt, L.norm_final, Fo, od, s));
    }
    ITTS_TRY(ln(xn, adt, x, adt, cond.after_norm, Fo, od, s));
    ITTS_TRY(tap("conformer_out", xn, adt, (int64_t)Fo * od, s));
    // --- perceiver resampler ---
    const int inner = cond.inner, ffi = cond.ffi, ffp = cond.ffi_pad, Sk = nl + Fo;
    void* kvsrc = alloc((size_t)Sk * D * es);
    ITTS_TRY(lin((char*)kvsrc + (size_t)nl * D * es, adt, xn, adt, od, cond.proj, Fo, D, s));
    void* lat = alloc((size_t)nl * D * es);
    K(cast_copy(lat, adt, cond.latents, F32, (long)nl * D, s));
    void* q = alloc((size_t)nl * inner * es);
    void* kv = alloc((size_t)Sk * 2 * inner * es);
    void* ao = alloc((size_t)nl * inner * es);
    void* hcat = alloc((size_t)nl * 2 * ffi * es);
    void* gg = alloc((size_t)nl * ffp * es);
    for (int j = 0; j < c.perc_layers; ++j) {
      const CondW::PL& L = cond.pl[j];
      K(copy_rows(kvsrc, D, lat, D, nl, D, adt, s));
      ITTS_TRY(lin(q, adt, lat, adt, D, L.to_q, nl, inner, s));
      ITTS_TRY(lin(kv, adt, kvsrc, adt, D, L.to_kv, Sk, 2 * inner, s));
      AttnArgs a;
      a.q = q;
      a.k = kv;
      a.v = (const char*)kv + (size_t)inner * es;
      a.o = ao;
      a.B = 1;
      a.H = inner / 64;
      a.Sq = nl;
      a.Sk = Sk;
      a.dqk = a.dv = 64;
      a.ldq = inner;
      a.ldk = a.ldv = 2 * inner;
      a.ldo = inner;
      a.scale = 0.125f;
      K(force_simple ? attention_simple(a, adt, s) : attention(a, adt, s));
      ITTS_TRY(lin(lat, adt, ao, adt, inner, L.to_out, nl, D, s, ACT_NONE, lat, D));
      ITTS_TRY(lin(hcat, adt, lat, adt, D, L.ff1, nl, 2 * ffi, s));
      K(geglu(gg, hcat, nl, ffi, ffp, adt, s));
      ITTS_TRY(lin(lat, adt, gg, adt, ffp, L.ff2, nl, D, s, ACT_NONE, lat, D));
    }
    K(rmsnorm_unit(cond_out, F32, lat, adt, cond.gamma, nl, D, s));
    return OK;
  };
  return two_pass(body, s);
}

}  // namespace itts
