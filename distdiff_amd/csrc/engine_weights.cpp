// Weight construction of the expansion engine: packed bf16 / fp32 GEMM operands, BatchNorm and LayerNorm folding (engine_internal.h).
#include "engine_internal.h"

namespace ddi {

// ---------------------------------------------------------------------------------------------------
// weight construction
// ---------------------------------------------------------------------------------------------------
// Every device buffer that holds weight-derived data goes through E->wupload(): it is registered, in creation order, in the engine's
// packed-weight list (dd_packed_bytes / dd_export_packed / dd_import_packed), and an engine that only knows the tensor SHAPES
// (dd_declare_tensor: a rank that will receive the packed buffers over RCCL) allocates it without packing anything on the host.

// fp32 packing (guide program): w is [Cout][Cin/groups][KH][KW]
ConvW* make_conv_f32(dd_engine* E, const float* w, const float* bias, int Cout, int Cin, int KH, int KW, int pad, int groups,
                     bool need_bwd) {
  auto cw = std::make_unique<ConvW>();
  cw->Cout = Cout; cw->Cin = Cin; cw->KH = KH; cw->KW = KW; cw->pad = pad; cw->groups = groups; cw->f32 = true;
  for (int mode = 0; mode < (need_bwd ? 2 : 1); ++mode) {
    PackedConv& sh = mode ? cw->sb : cw->sf;
    sh = pack_conv_shape_f32(Cout, Cin, KH, KW, mode, groups);
    std::vector<float> wp;
    std::vector<int> tt;
    if (!E->shape_only) {
      wp.resize((size_t)sh.N * sh.K); tt.resize(sh.ntaps);
      pack_conv_weight_f32(w, Cout, Cin, KH, KW, pad, mode, groups, wp.data(), tt.data());
    }
    float* d = (float*)E->wupload(wp.data(), (size_t)sh.N * sh.K * 4);
    int* t = (int*)E->wupload(tt.data(), (size_t)sh.ntaps * 4);
    if (mode) { cw->wf_bwd = d; cw->tap_bwd = t; } else { cw->wf_fwd = d; cw->tap_fwd = t; }
  }
  if (bias || E->shape_only) cw->bias = (float*)E->wupload(bias, (size_t)Cout * 4);
  E->convs.push_back(std::move(cw));
  return E->convs.back().get();
}

// has_bias is passed explicitly: a shape-only engine has no host data to look at.
// ln_gamma / ln_beta (host, [Cin]; `fold` tells a shape-only engine): the LayerNorm in front of this linear is folded into it --
//   LN(x) W^T + b = rstd * (x (gamma o W)^T - mean * c1) + (b + W beta),  c1[n] = sum_k (gamma o W)[n, k]
// (kernels.h CF_LNFOLD): the FORWARD packing holds gamma o W, the bias W beta + b, ln_c1 the column sums of the bf16-rounded folded
// weights (the rank-1 correction must cancel what the MFMAs actually accumulate); the input-gradient packing keeps the plain W, the
// LayerNorm's own backward multiplies by gamma as before.
ConvW* make_conv_raw(dd_engine* E, const float* w_in, const float* bias_in, bool has_bias, int Cout, int Cin, int KH, int KW, int pad,
                     bool geglu, bool need_bwd, bool fold, const float* ln_gamma, const float* ln_beta, int qrows, float qscale) {
  // the softmax scale of an attention query projection, folded into its rows (both packings: the attention backward returns the
  // gradient with respect to the SCALED query)
  std::vector<float> wq, bq;
  const float* w = w_in;
  const float* bias = bias_in;
  if (qrows > 0 && qscale != 1.f && !E->shape_only) {
    const size_t per = (size_t)Cin * KH * KW;
    wq.assign(w_in, w_in + (size_t)Cout * per);
    for (size_t i = 0; i < (size_t)qrows * per; ++i) wq[i] *= qscale;
    w = wq.data();
    if (bias_in) {
      bq.assign(bias_in, bias_in + Cout);
      for (int n = 0; n < qrows; ++n) bq[n] *= qscale;
      bias = bq.data();
    }
  }
  auto cw = std::make_unique<ConvW>();
  cw->Cout = Cout; cw->Cin = Cin; cw->KH = KH; cw->KW = KW; cw->pad = pad; cw->geglu = geglu;
  if (fold && (KH != 1 || KW != 1)) throw std::runtime_error("LayerNorm folding needs a linear layer");
  std::vector<float> wfold, bfold, c1;
  if (fold && !E->shape_only) {
    wfold.resize((size_t)Cout * Cin); bfold.assign(Cout, 0.f); c1.assign(Cout, 0.f);
    for (int n = 0; n < Cout; ++n) {
      double sb = bias ? bias[n] : 0.0, sc = 0.0;
      for (int k = 0; k < Cin; ++k) {
        const float wf = w[(size_t)n * Cin + k] * ln_gamma[k];
        wfold[(size_t)n * Cin + k] = wf;
        sb += (double)w[(size_t)n * Cin + k] * ln_beta[k];
        sc += host_bf2f(host_f2bf(wf));
      }
      bfold[n] = (float)sb; c1[n] = (float)sc;
    }
  }
  for (int mode = 0; mode < (need_bwd ? 2 : 1); ++mode) {
    PackedConv& sh = mode ? cw->sb : cw->sf;
    sh = pack_conv_shape(Cout, Cin, KH, KW, mode);
    std::vector<bf16_t> wp;
    std::vector<int> tt;
    if (!E->shape_only) {
      wp.resize((size_t)sh.N * sh.K); tt.resize(sh.ntaps);
      pack_conv_weight((fold && mode == 0) ? wfold.data() : w, Cout, Cin, KH, KW, pad, mode, geglu, wp.data(), tt.data());
    }
    bf16_t* d = (bf16_t*)E->wupload(wp.data(), (size_t)sh.N * sh.K * 2);
    int* t = (int*)E->wupload(tt.data(), (size_t)sh.ntaps * 4);
    if (mode) { cw->w_bwd = d; cw->tap_bwd = t; } else { cw->w_fwd = d; cw->tap_fwd = t; }
  }
  if (has_bias || fold) {
    std::vector<float> b, c;
    if (!E->shape_only) {
      b.resize(Cout);
      const float* src = fold ? bfold.data() : bias;
      for (int n = 0; n < Cout; ++n) b[n] = src[geglu ? geglu_perm(n, Cout / 2) : n];
      if (fold) { c.resize(Cout); for (int n = 0; n < Cout; ++n) c[n] = c1[geglu ? geglu_perm(n, Cout / 2) : n]; }
    }
    cw->bias = (float*)E->wupload(b.data(), (size_t)Cout * 4);
    if (fold) cw->ln_c1 = (float*)E->wupload(c.data(), (size_t)Cout * 4);
  }
  E->convs.push_back(std::move(cw));
  return E->convs.back().get();
}

bool ln_fold_enabled() { static const bool on = !getenv("DD_NO_LN_FOLD"); return on; }
bool attn_prescale() { static const bool on = !(getenv("DD_ATTN_PRESCALE") && atoi(getenv("DD_ATTN_PRESCALE")) == 0); return on; }

// ln: prefix of the LayerNorm to fold into this linear ("" = none)
ConvW* make_conv(dd_engine* E, const std::string& model, const std::string& prefix, int pad, bool geglu, bool has_bias, const std::string& ln,
                 int qrows, float qscale) {
  const HostTensor& w = E->get(model, prefix + ".weight");
  const int Cout = (int)w.shape[0], Cin = (int)w.shape[1];
  const int KH = w.shape.size() == 4 ? (int)w.shape[2] : 1, KW = w.shape.size() == 4 ? (int)w.shape[3] : 1;
  const bool hb = has_bias && E->has(model, prefix + ".bias");
  const float* b = hb ? E->get(model, prefix + ".bias").data.data() : nullptr;
  const bool fold = !ln.empty();
  return make_conv_raw(E, w.data.data(), b, hb, Cout, Cin, KH, KW, pad, geglu, E->cfg.enable_grad != 0, fold,
                       fold ? E->get(model, ln + ".weight").data.data() : nullptr, fold ? E->get(model, ln + ".bias").data.data() : nullptr,
                       qrows, qscale);
}

// several linears sharing the input, concatenated along Cout (fused QKV)
ConvW* make_conv_cat(dd_engine* E, const std::string& model, const std::vector<std::string>& prefixes, bool with_bias,
                     const std::string& ln, int qrows, float qscale) {
  std::vector<float> w, b;
  int Cin = 0, Cout = 0;
  for (auto& p : prefixes) {
    const HostTensor& t = E->get(model, p + ".weight");
    Cin = (int)t.shape[1];
    Cout += (int)t.shape[0];
    w.insert(w.end(), t.data.begin(), t.data.end());
    if (with_bias) { const HostTensor& bb = E->get(model, p + ".bias"); b.insert(b.end(), bb.data.begin(), bb.data.end()); }
  }
  const bool fold = !ln.empty();
  return make_conv_raw(E, w.data(), with_bias ? b.data() : nullptr, with_bias, Cout, Cin, 1, 1, 0, false, E->cfg.enable_grad != 0, fold,
                       fold ? E->get(model, ln + ".weight").data.data() : nullptr, fold ? E->get(model, ln + ".bias").data.data() : nullptr,
                       qrows, qscale);
}

// conv (no bias) followed by eval-mode BatchNorm, folded: w' = w * g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps)
// `cin_total` = channels of the input tensor: groups = cin_total / weight.shape[1] (ResNeXt, model_utils.py:56-63).  The guide
// program is fp32 (guide_f32.hip).
ConvW* make_conv_bn(dd_engine* E, const std::string& model, const std::string& conv, const std::string& bn, int pad, float eps,
                    int cin_total) {
  const HostTensor& w = E->get(model, conv + ".weight");
  const HostTensor& g = E->get(model, bn + ".weight");
  const HostTensor& be = E->get(model, bn + ".bias");
  const HostTensor& mu = E->get(model, bn + ".running_mean");
  const HostTensor& var = E->get(model, bn + ".running_var");
  const int Cout = (int)w.shape[0], Cg = (int)w.shape[1], KH = (int)w.shape[2], KW = (int)w.shape[3];
  if (Cg < 1 || cin_total % Cg || Cout % (cin_total / Cg))
    throw std::runtime_error("guide conv " + conv + ": weight shape does not divide the input channels (groups)");
  const int groups = cin_total / Cg;
  std::vector<float> wf, bf(Cout);
  if (!E->shape_only) {
    wf.resize(w.numel());
    const size_t per = (size_t)Cg * KH * KW;
    for (int n = 0; n < Cout; ++n) {
      const float sc = g.data[n] / sqrtf(var.data[n] + eps);
      for (size_t i = 0; i < per; ++i) wf[n * per + i] = w.data[n * per + i] * sc;
      bf[n] = be.data[n] - mu.data[n] * sc;
    }
  }
  return make_conv_f32(E, wf.data(), bf.data(), Cout, cin_total, KH, KW, pad, groups, E->cfg.enable_grad != 0);
}

NormW* make_norm(dd_engine* E, const std::string& model, const std::string& prefix) {
  auto nw = std::make_unique<NormW>();
  const HostTensor& g = E->get(model, prefix + ".weight");
  const HostTensor& b = E->get(model, prefix + ".bias");
  nw->C = (int)g.shape[0];
  nw->gamma = (float*)E->wupload(g.data.data(), (size_t)nw->C * 4);
  nw->beta = (float*)E->wupload(b.data.data(), (size_t)nw->C * 4);
  E->norms.push_back(std::move(nw));
  return E->norms.back().get();
}


}  // namespace ddi
