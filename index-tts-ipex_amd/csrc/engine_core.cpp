// Engine infrastructure: tensor registry, name resolution, workspace, op wrappers.
#include "engine.h"

#include <cstring>
#include <sstream>

namespace itts {

Engine::~Engine() {
  if (ws) (void)hipFree(ws);
  if (ws_b) (void)hipFree(ws_b);
  if (gpt_tiles) (void)hipFree(gpt_tiles);
  DecodeState& d = ds;
  void* ptrs[] = {d.kc, d.vc, d.h, d.qkv, d.ctx, d.act, d.hn, d.logits, d.len, d.prefix_dev,
                  d.kv_start, d.cur_tok, d.ids, d.unfinished, d.seen, d.partial, d.attn_o, d.attn_ml, d.uniforms, d.forced, d.scores2, d.gran, d.fuse_err, d.eng_gran, d.eng_ctr, d.beam_ids, d.anc, d.beam_scores, d.cand_sc, d.cand_tok, d.cand_n, d.hyp_tok,
                  d.hyp_score, d.hyp_len, d.hyp_order, d.hyp_n, d.hyp_worst, d.hyp_counter, d.beam_done};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ksplit_ws) (void)hipFree(ksplit_ws);
  if (d.graph) (void)hipGraphExecDestroy(d.graph);
  if (d.graphK) (void)hipGraphExecDestroy(d.graphK);
}

int Engine::ws_reserve(size_t bytes, hipStream_t s) {
  if (bytes <= ws_cap) return OK;
  ITTS_HIP_CHECK(hipStreamSynchronize(s));
  if (ws) ITTS_HIP_CHECK(hipFree(ws));
  ws = nullptr;
  ws_cap = 0;
  size_t want = bytes + bytes / 4;
  if (hipMalloc((void**)&ws, want) != hipSuccess) {
    (void)hipGetLastError();
    set_error("workspace allocation of " + std::to_string(want) + " bytes failed");
    return E_NOMEM;
  }
  ws_cap = want;
  return OK;
}

int Engine::tap(const char* name, const void* p, int dt, int64_t n, hipStream_t s) {
  if (!debug || dry) return OK;
  std::vector<float>& v = taps[name];
  v.resize(n);
  if (dt == F32) {
    ITTS_HIP_CHECK(hipMemcpyAsync(v.data(), p, n * 4, hipMemcpyDeviceToHost, s));
    ITTS_HIP_CHECK(hipStreamSynchronize(s));
  } else {
    std::vector<uint16_t> tmp(n);
    ITTS_HIP_CHECK(hipMemcpyAsync(tmp.data(), p, n * 2, hipMemcpyDeviceToHost, s));
    ITTS_HIP_CHECK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) {
#ifdef ITTS_HALF_F16
      _Float16 h;
      memcpy(&h, &tmp[i], 2);
      v[i] = (float)h;
#else
      uint32_t u = (uint32_t)tmp[i] << 16;
      float f;
      memcpy(&f, &u, 4);
      v[i] = f;
#endif
    }
  }
  return OK;
}

int Engine::lin(void* C, int tc, const void* A, int ta, int lda, const Lin& w, int M, int ldc, hipStream_t s, int act,
                const void* R, int ldr, float alpha) {
  GemmArgs g;
  g.A = A;
  g.W = w.w;
  g.C = C;
  g.M = M;
  g.N = w.N;
  g.Cin = w.Cin;
  g.taps = 1;
  g.lda = lda;
  g.ldc = ldc;
  g.bias = w.b;
  g.act = act;
  g.scale = w.bn_scale;
  g.shift = w.bn_shift;
  g.R = R;
  g.ldr = ldr;
  g.alpha = alpha;
  return conv(g, ta, w.dt, tc, s);
}

int Engine::conv(GemmArgs& g, int ta, int tw, int tc, hipStream_t s) {
  if (dry) return OK;
  if (force_simple) return gemm_simple(g, ta, tw, tc, s);
  // few-tile deep-K shapes split K over workgroups (this engine's own fp32 workspace: engines on other streams have theirs)
  const int S = ksplit_off ? 1 : gemm_ksplit_plan(g, ta, tw, tc, KSPLIT_WS_BYTES);
  if (S > 1) {
    if (!ksplit_ws) {  // first use (prefill / latent / vocoder: never inside a graph capture)
      if (hipMalloc((void**)&ksplit_ws, KSPLIT_WS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        ksplit_ws = nullptr;
        set_error("K-split workspace allocation failed");
        return E_NOMEM;
      }
    }
    g.ws = ksplit_ws;
    g.ksplit = S;
  }
  return gemm(g, ta, tw, tc, s);
}

int Engine::ln(void* y, int ty, const void* x, int tx, const Norm& n, int rows, int D, hipStream_t s, int act, float eps) {
  if (dry) return OK;
  return layernorm(y, ty, x, tx, n.g, n.b, rows, D, D, D, eps, act, s);  // null gamma/beta = no affine
}

// ------------------------------------------------------------------------------------------------
// name resolution
// ------------------------------------------------------------------------------------------------
namespace {

struct Resolver {
  Engine& e;
  std::ostringstream err;
  int nerr = 0;
  explicit Resolver(Engine& e_) : e(e_) {}
  bool has(const std::string& n) const { return e.tensors.count(n) > 0; }
  const Tensor* get(const std::string& n, int dt, std::initializer_list<int64_t> dims) {
    auto it = e.tensors.find(n);
    if (it == e.tensors.end()) {
      if (nerr++ < 12) err << "missing tensor '" << n << "'; ";
      return nullptr;
    }
    const Tensor& t = it->second;
    bool ok = t.dt == dt && t.nd == (int)dims.size();
    int i = 0;
    for (int64_t d : dims) {
      if (ok && d >= 0 && t.d[i] != d) ok = false;
      ++i;
    }
    if (!ok) {
      if (nerr++ < 12) {
        err << "tensor '" << n << "' has dtype " << t.dt << " dims [";
        for (int k = 0; k < t.nd; ++k) err << t.d[k] << (k + 1 < t.nd ? "," : "");
        err << "], expected dtype " << dt << " dims [";
        int k = 0;
        for (int64_t d : dims) err << d << (++k < (int)dims.size() ? "," : "");
        err << "]; ";
      }
      return nullptr;
    }
    return &t;
  }
  const float* f32(const std::string& n, std::initializer_list<int64_t> dims) {
    const Tensor* t = get(n, F32, dims);
    return t ? (const float*)t->p : nullptr;
  }
  // W [nphase*N? , taps*Cin] stored as dims [nphase, N, taps*Cin] when nphase > 1 else [N, taps*Cin]
  Lin lin(const std::string& n, int dt, int N, int Cin, int taps = 1, bool bias = true, int nphase = 1, bool bn = false) {
    Lin l;
    l.N = N;
    l.Cin = Cin;
    l.taps = taps;
    l.nphase = nphase;
    l.dt = dt;
    const Tensor* t = nphase > 1 ? get(n + ".weight", dt, {nphase, N, (int64_t)taps * Cin})
                                 : get(n + ".weight", dt, {N, (int64_t)taps * Cin});
    if (t) l.w = t->p;
    if (bias) l.b = f32(n + ".bias", {N});
    if (bn) {
      l.bn_scale = f32(n + ".bn_scale", {N});
      l.bn_shift = f32(n + ".bn_shift", {N});
    }
    return l;
  }
  Norm norm(const std::string& n, int D) {
    Norm r;
    r.g = f32(n + ".weight", {D});
    r.b = f32(n + ".bias", {D});
    return r;
  }
};

bool any_prefix(const Engine& e, const std::string& p) {
  for (auto& kv : e.tensors)
    if (kv.first.compare(0, p.size(), p) == 0) return true;
  return false;
}

}  // namespace

int Engine::finalize() {
  Resolver r(*this);
  const itts_config& c = cfg;
  const int wdt = adt;
  // ---- conformer + perceiver ----
  if (any_prefix(*this, "cond.")) {
    const int od = c.cond_dim, fo = (c.cond_idim - 3) / 2 + 1, D = c.model_dim;
    cond = CondW();
    cond.conv_w = r.f32("cond.embed.conv.weight", {od, 9});
    cond.conv_b = r.f32("cond.embed.conv.bias", {od});
    cond.embed_out = r.lin("cond.embed.out", wdt, od, od * fo);
    const Tensor* pe = r.get("cond.pe", wdt, {-1, od});
    if (pe) {
      cond.pe = pe->p;
      cond.pe_len = (int)pe->d[0];
    }
    for (int i = 0; i < c.cond_blocks; ++i) {
      const std::string p = "cond." + std::to_string(i) + ".";
      ConformerLayerW L;
      L.norm_mha = r.norm(p + "norm_mha", od);
      L.norm_conv = r.norm(p + "norm_conv", od);
      L.norm_ff = r.norm(p + "norm_ff", od);
      L.norm_final = r.norm(p + "norm_final", od);
      L.conv_norm = r.norm(p + "conv.norm", od);
      L.qkv = r.lin(p + "qkv", wdt, 3 * od, od);
      L.pos = r.lin(p + "pos", wdt, od, od, 1, false);
      L.out = r.lin(p + "out", wdt, od, od);
      L.pw1 = r.lin(p + "conv.pw1", wdt, 2 * od, od);
      L.pw2 = r.lin(p + "conv.pw2", wdt, od, od);
      L.w1 = r.lin(p + "ff.w1", wdt, c.cond_ff, od);
      L.w2 = r.lin(p + "ff.w2", wdt, od, c.cond_ff);
      L.bu = r.f32(p + "pos_bias_u", {od});
      L.bv = r.f32(p + "pos_bias_v", {od});
      L.dw_w = r.f32(p + "conv.dw.weight", {od, 15});
      L.dw_b = r.f32(p + "conv.dw.bias", {od});
      cond.layers.push_back(L);
    }
    cond.after_norm = r.norm("cond.after_norm", od);
    cond.inner = c.cond_heads * 64;
    cond.ffi = c.perc_inner;
    cond.ffi_pad = (c.perc_inner + 31) / 32 * 32;
    cond.latents = r.f32("perc.latents", {c.cond_latents, D});
    cond.proj = r.lin("perc.proj", wdt, D, od);
    for (int j = 0; j < c.perc_layers; ++j) {
      const std::string p = "perc." + std::to_string(j) + ".";
      CondW::PL L;
      L.to_q = r.lin(p + "to_q", wdt, cond.inner, D, 1, false);
      L.to_kv = r.lin(p + "to_kv", wdt, 2 * cond.inner, D, 1, false);
      L.to_out = r.lin(p + "to_out", wdt, D, cond.inner, 1, false);
      L.ff1 = r.lin(p + "ff1", wdt, 2 * cond.ffi, D);
      L.ff2 = r.lin(p + "ff2", wdt, D, cond.ffi_pad);
      cond.pl.push_back(L);
    }
    cond.gamma = r.f32("perc.norm.gamma", {D});
    cond.ok = true;
  }
  // ---- GPT ----
  if (any_prefix(*this, "gpt.")) {
    const int D = c.model_dim, V = c.number_mel_codes;
    gpt = GptW();
    if (gpt_tiles) (void)hipFree(gpt_tiles);  // fragment-tiled copies belong to the weights they were made from
    gpt_tiles = nullptr;
    for (int i = 0; i < c.layers; ++i) {
      const std::string p = "gpt.h." + std::to_string(i) + ".";
      GptLayerW L;
      // ln_1 / ln_2 gamma,beta are folded into c_attn / c_fc by the packer (L.ln1 / L.ln2 stay null = plain normalise)
      L.attn = r.lin(p + "attn.c_attn", wdt, 3 * D, D);
      L.proj = r.lin(p + "attn.c_proj", wdt, D, D);
      L.fc = r.lin(p + "mlp.c_fc", wdt, 4 * D, D);
      L.proj2 = r.lin(p + "mlp.c_proj", wdt, D, 4 * D);
      // optional fp8 (e4m3) copies for the decode GEMV: "<name>.weight_fp8" [N, K] bytes + "<name>.weight_scale" [N]
      auto fp8 = [&](Lin& l, const std::string& n) {
        if (!r.has(n + ".weight_fp8")) return;
        const Tensor* q = r.get(n + ".weight_fp8", FP8, {l.N, l.Cin});
        const float* sc = r.f32(n + ".weight_scale", {l.N});
        if (q && sc) {
          l.w8 = q->p;
          l.wscale = sc;
        }
      };
      fp8(L.attn, p + "attn.c_attn");
      fp8(L.proj, p + "attn.c_proj");
      fp8(L.fc, p + "mlp.c_fc");
      fp8(L.proj2, p + "mlp.c_proj");
      gpt.layers.push_back(L);
    }
    gpt.ln_f = r.norm("gpt.ln_f", D);
    gpt.final_norm = r.norm("gpt.final_norm", D);
    gpt.head = r.lin("gpt.mel_head", wdt, V, D);
    if (r.has("gpt.mel_head.weight_fp8")) {
      const Tensor* q = r.get("gpt.mel_head.weight_fp8", FP8, {V, D});
      const float* sc = r.f32("gpt.mel_head.weight_scale", {V});
      if (q && sc) {
        gpt.head.w8 = q->p;
        gpt.head.wscale = sc;
      }
    }
    const Tensor* t;
    if ((t = r.get("gpt.text_embedding", wdt, {c.number_text_tokens + 1, D}))) gpt.text_emb = t->p;
    if ((t = r.get("gpt.mel_embedding", wdt, {V, D}))) gpt.mel_emb = t->p;
    if ((t = r.get("gpt.mel_pos", wdt, {c.max_mel_tokens + 3, D}))) gpt.mel_pos = t->p;
    if ((t = r.get("gpt.text_pos", wdt, {c.max_text_tokens + 2, D}))) gpt.text_pos = t->p;
    gpt.ok = true;
  }
  // ---- BigVGAN generator ----
  if (any_prefix(*this, "bv.")) {
    bv = BigvganW();
    const int C0 = c.bv_init_ch, E = c.bv_spk_dim;
    bv.conv_pre = r.lin("bv.conv_pre", wdt, C0, c.bv_gpt_dim, 7);
    bv.cond_layer = r.lin("bv.cond_layer", F32, C0, E);
    int ch = C0;
    for (int i = 0; i < c.bv_num_up; ++i) {
      const int u = c.bv_up_rates[i], k = c.bv_up_kernels[i];
      const int cin = C0 >> i, cout = C0 >> (i + 1);
      bv.ups.push_back(r.lin("bv.ups." + std::to_string(i), wdt, cout, cin, k / u, true, u));
      bv.conds.push_back(r.lin("bv.conds." + std::to_string(i), F32, cout, E));
      ch = cout;
      for (int j = 0; j < c.bv_num_res; ++j) {
        const std::string p = "bv.res." + std::to_string(i * c.bv_num_res + j) + ".";
        AmpW a;
        for (int l = 0; l < c.bv_num_dil; ++l) {
          a.c1[l] = r.lin(p + "c1." + std::to_string(l), wdt, ch, ch, c.bv_res_kernels[j]);
          a.c2[l] = r.lin(p + "c2." + std::to_string(l), wdt, ch, ch, c.bv_res_kernels[j]);
          a.a1[l] = r.f32(p + "act." + std::to_string(2 * l) + ".alpha", {ch});
          a.b1[l] = r.f32(p + "act." + std::to_string(2 * l) + ".beta", {ch});
          a.a2[l] = r.f32(p + "act." + std::to_string(2 * l + 1) + ".alpha", {ch});
          a.b2[l] = r.f32(p + "act." + std::to_string(2 * l + 1) + ".beta", {ch});
        }
        bv.res.push_back(a);
      }
    }
    bv.post_alpha = r.f32("bv.act_post.alpha", {ch});
    bv.post_beta = r.f32("bv.act_post.beta", {ch});
    bv.conv_post = r.lin("bv.conv_post", wdt, 1, ch, 7);
    bv.filter = r.f32("bv.filter", {12});
    bv.ok = true;
  }
  // ---- ECAPA ----
  if (any_prefix(*this, "spk.")) {
    ec = EcapaW();
    const int* chs = c.ec_channels;
    ec.b0 = r.lin("spk.b0", wdt, chs[0], (c.bv_num_mels + 7) / 8 * 8, c.ec_kernels[0], true, 1, true);  // mel bins padded to x8
    for (int i = 1; i <= 3; ++i) {
      const std::string p = "spk.b" + std::to_string(i) + ".";
      EcapaW::Blk b;
      b.tdnn1 = r.lin(p + "tdnn1", wdt, chs[i], chs[i - 1], 1, true, 1, true);
      const int hc = chs[i] / c.ec_scale;
      for (int q = 0; q < c.ec_scale - 1; ++q)
        b.res.push_back(r.lin(p + "res." + std::to_string(q), wdt, hc, hc, c.ec_kernels[i], true, 1, true));
      b.tdnn2 = r.lin(p + "tdnn2", wdt, chs[i], chs[i], 1, true, 1, true);
      b.se1 = r.lin(p + "se1", F32, c.ec_se, chs[i]);
      b.se2 = r.lin(p + "se2", F32, chs[i], c.ec_se);
      ec.blks.push_back(b);
    }
    ec.mfa = r.lin("spk.mfa", wdt, chs[4], chs[3] * 3, 1, true, 1, true);
    ec.asp_x = r.lin("spk.asp.tdnn_x", wdt, c.ec_att, chs[4], 1, false);
    ec.asp_ms = r.lin("spk.asp.tdnn_ms", F32, c.ec_att, chs[4] * 2, 1, true);
    ec.asp_x.bn_scale = r.f32("spk.asp.tdnn.bn_scale", {c.ec_att});
    ec.asp_x.bn_shift = r.f32("spk.asp.tdnn.bn_shift", {c.ec_att});
    ec.asp_conv = r.lin("spk.asp.conv", wdt, chs[4], c.ec_att);
    ec.aspbn_scale = r.f32("spk.asp_bn.scale", {chs[4] * 2});
    ec.aspbn_shift = r.f32("spk.asp_bn.shift", {chs[4] * 2});
    ec.fc = r.lin("spk.fc", F32, c.bv_spk_dim, chs[4] * 2);
    ec.ok = true;
  }
  // ---- DVAE decoder ----
  if (any_prefix(*this, "dvae.")) {
    dv = DvaeW();
    const int inner = c.dv_hidden << (c.dv_layers - 1);
    const Tensor* t = r.get("dvae.codebook", wdt, {c.dv_tokens, c.dv_codebook});
    if (t) dv.codebook = t->p;
    dv.in_conv = r.lin("dvae.in", wdt, inner, c.dv_codebook);
    for (int i = 0; i < c.dv_resblocks; ++i) {
      const std::string p = "dvae.rb" + std::to_string(i) + ".";
      DvaeW::RB b;
      b.c0 = r.lin(p + "c0", wdt, inner, inner, 3);
      b.c2 = r.lin(p + "c2", wdt, inner, inner, 3);
      b.c4 = r.lin(p + "c4", wdt, inner, inner, 1);
      dv.rbs.push_back(b);
    }
    int ci = inner;
    for (int i = 0; i < c.dv_layers; ++i) {
      const int co = inner >> i;  // dec_chans = [inner, inner, inner/2, ...]
      dv.ups.push_back(r.lin("dvae.up" + std::to_string(i), wdt, co, ci, c.dv_kernel));
      ci = co;
    }
    dv.out_conv = r.lin("dvae.out", wdt, c.dv_channels, ci);
    dv.ok = true;
    if (tensors.count("dvae.enc0.weight")) {
      int cin = c.dv_channels;
      for (int i = 0; i < c.dv_layers; ++i) {
        const int co = c.dv_hidden << i;
        dv.enc.push_back(r.lin("dvae.enc" + std::to_string(i), wdt, co, 2 * cin, 2));
        cin = co;
      }
      for (int i = 0; i < c.dv_resblocks; ++i) {
        const std::string p = "dvae.erb" + std::to_string(i) + ".";
        DvaeW::RB b;
        b.c0 = r.lin(p + "c0", wdt, inner, inner, 3);
        b.c2 = r.lin(p + "c2", wdt, inner, inner, 3);
        b.c4 = r.lin(p + "c4", wdt, inner, inner, 1);
        dv.erbs.push_back(b);
      }
      dv.eout = r.lin("dvae.eout", wdt, c.dv_codebook, inner);
      dv.quant = Lin();
      dv.quant.w = dv.codebook;
      dv.quant.N = c.dv_tokens;
      dv.quant.Cin = c.dv_codebook;
      dv.quant.dt = wdt;
      dv.codebook_sq = r.f32("dvae.codebook_sq", {c.dv_tokens});
      dv.enc_ok = true;
    }
  }
  if (r.nerr) {
    set_error("finalize: " + r.err.str() + (r.nerr > 12 ? "(+" + std::to_string(r.nerr - 12) + " more)" : ""));
    return E_MISSING;
  }
  finalized = true;
  return OK;
}

}  // namespace itts
