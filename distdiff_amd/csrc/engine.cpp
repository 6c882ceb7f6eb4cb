// The expansion engine: a static op-graph executor for the SD-1.x UNet, the AutoencoderKL decoder and the
// ResNet-50 guide, with a hand-derived reverse program (input-gradients only, no autograd) composed from
// per-op VJPs, plus the guided DDIM sampler drivers behind the C ABI of include/distdiff_hip.h.
//
// Reference functions replaced: generate_data.py:109-121 (denoise_one_step), :687-732 (transform_guidance),
// :735-767 (direct_guidance), :1161-1228 (loop + decode); diffusers / timm module forwards as listed in
// SURVEY.md section 8a. Activations are NHWC bf16 matrices; every forward op stashes what its VJP needs
// (288 GB HBM makes a full stash viable), the reverse program is built once at finalize time.
#include "engine_internal.h"

namespace {
// ---------------------------------------------------------------------------------------------------
// sampler drivers
// ---------------------------------------------------------------------------------------------------
struct Run {
  dd_engine* E; hipStream_t s; int B;
  Ctx ctx(const Program& P, char* act) {
    Ctx c; c.act = act; c.grad = E->grad_slab; c.tr = E->tr_slab; c.tr_stride = P.tr_max; c.scratch_partial = E->scratch_partial; c.partial_cap = E->partial_cap;
    c.scratch_tmp = E->scratch_tmp; c.tmp_cap = E->tmp_cap; c.tap1x1 = E->tap1x1; c.gn_scratch = E->gn_scratch; c.rowpart = E->rowpart; c.gn_coef = E->gn_coef; c.s = s; c.B = B; c.cross_kv = &E->cross_kv; c.flops = &E->flops; c.prof = &E->prof;
    return c;
  }
};

void check_batch(dd_engine* E, int B) {
  if (B != E->cfg.max_batch) throw std::runtime_error("this engine was built for batch " + std::to_string(E->cfg.max_batch) +
                                                      " (static shapes); got B=" + std::to_string(B));
  if (!E->finalized) throw std::runtime_error("dd_finalize_weights has not been called");
}

// UNet forward on instance k: z fp32 NCHW -> eps2 fp32 NHWC [2B*HW, ld] inside the slab
void unet_fwd(dd_engine* E, int k, const float* z, int step_index, hipStream_t s, bool stash = true) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx ctx = r.ctx(E->unet, E->inst[k].unet);
  ctx.step_index = step_index;
  ctx.stash = stash;
  if (c.unet_add_time_dim > 0) {
    if (!E->added_cond_set) throw std::runtime_error("this UNet has text_time additional conditioning: call dd_set_added_cond first");
    ctx.img_bias = 2 * c.max_batch;
  }
  const Tn& in = E->unet.t[E->unet_in];
  HIPCHK(launch_nchw_f32_to_nhwc_bf16(z, act_ptr(ctx, in), c.max_batch, c.unet_in_channels, c.latent_size, c.latent_size, in.ld, in.ld,
                                      in.B == 2 * c.max_batch ? 1 : 0, 1.f, s));
  run_fwd(E->unet, ctx);
}

void vae_fwd(dd_engine* E, int k, const float* x0, hipStream_t s) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx ctx = r.ctx(E->vae, E->inst[k].vae);
  const Tn& in = E->vae.t[E->vae_in];
  HIPCHK(launch_nchw_f32_to_nhwc_bf16(x0, act_ptr(ctx, in), c.max_batch, c.vae_latent_channels, c.latent_size, c.latent_size, in.ld,
                                      in.ld, 0, 1.f / c.vae_scaling_factor, s));
  run_fwd(E->vae, ctx);
}

// features of a finished guide forward -> feats [B, D] fp32: global average pool of the last feature map (ResNets, model_utils.py:31-33)
// or the projected class token (ViT)
void guide_features_out(dd_engine* E, const Ctx& gc, float* feats, hipStream_t s, int use_max = 0) {
  const dd_config& c = E->cfg;
  const Tn& f = E->guide.t[E->guide_feat];
  if (c.guide_kind == 1) HIPCHK(hipMemcpy2DAsync(feats, (size_t)f.C * 4, act_f32(gc, f), (size_t)f.ld * 4, (size_t)f.C * 4, f.rows, hipMemcpyDeviceToDevice, s));
  else HIPCHK(launch_gap_f32(act_f32(gc, f), f.ld, feats, nullptr, c.max_batch, f.H * f.W, f.C, use_max, s));
}
// cotangent of the features [B, D] fp32 -> gradient of the guide's output tensor (reverse of guide_features_out)
void guide_features_grad_in(dd_engine* E, const Ctx& gc, const float* gfeat, hipStream_t s) {
  const dd_config& c = E->cfg;
  const Tn& f = E->guide.t[E->guide_feat];
  if (c.guide_kind == 1) {
    if (f.ld != f.C) throw std::runtime_error("ViT guide: feature dim must be a multiple of 8");
    HIPCHK(launch_f32_to_bf16(gfeat, grad_ptr(gc, f), (size_t)f.rows * f.C, s));
  } else {
    HIPCHK(launch_gap_bwd_f32(gfeat, grad_f32(gc, f), f.ld, c.max_batch, f.H * f.W, f.C, nullptr, s));
  }
}

// guide forward from the decoded image of instance k (bicubic -> guide network -> features) -> feat [B, D]
void guide_fwd_from_image(dd_engine* E, int k, hipStream_t s) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx gc = r.ctx(E->guide, E->inst[k].guide);
  Ctx vc = r.ctx(E->vae, E->inst[k].vae);
  const Tn& img = E->vae.t[E->vae_out];
  const Tn& gin = E->guide.t[E->guide_in];
  // the decoder's conv_out stores the image in fp32: image, bicubic resize and the whole guide stay fp32
  if (E->image_override) {  // parity tests: the guide (its ReLU / max-pool masks) is evaluated AT the given image, gradients flow as usual
    const size_t per = (size_t)c.max_batch * c.vae_out_channels * img.H * img.W;
    const float* src = E->image_override + per * std::min(k, E->image_override_count - 1);
    HIPCHK(launch_nchw_to_nhwc_f32(src, act_f32(vc, img), c.max_batch, c.vae_out_channels, img.H, img.W, img.ld, img.ld, s));
  }
  HIPCHK(launch_bicubic_f32(act_f32(vc, img), img.ld, act_f32(gc, gin), gin.ld, c.max_batch, img.H, img.W, gin.H, gin.W, 3, gin.ld, s));
  run_fwd(E->guide, gc);
  guide_features_out(E, gc, E->inst[k].feat, s);
}

// reverse of guide_fwd_from_image + vae_fwd: gfeat -> g_x0 (fp32 NCHW)
void guide_vae_bwd(dd_engine* E, int k, float* g_x0, hipStream_t s) {
  const dd_config& c = E->cfg;
  Run r{E, s, c.max_batch};
  Ctx gc = r.ctx(E->guide, E->inst[k].guide);
  Ctx vc = r.ctx(E->vae, E->inst[k].vae);
  const Tn& f = E->guide.t[E->guide_feat];
  // GAP^T; the ReLU mask of the last bottleneck is applied by that conv op's backward
  (void)f;
  guide_features_grad_in(E, gc, E->inst[k].gfeat, s);
  run_bwd(E->guide, gc);
  const Tn& gin = E->guide.t[E->guide_in];
  const Tn& img = E->vae.t[E->vae_out];
  // the guide (fp32) and VAE (bf16) gradient regions are disjoint parts of the shared gradient slab (see finalize)
  HIPCHK(launch_bicubic_bwd_f32(grad_f32(gc, gin), gin.ld, grad_ptr(vc, img), 1, img.ld, c.max_batch, img.H, img.W, gin.H, gin.W, 3, s));
  run_bwd(E->vae, vc);
  const Tn& vin = E->vae.t[E->vae_in];
  HIPCHK(launch_nhwc_to_nchw_f32(grad_ptr(vc, vin), 0, g_x0, c.max_batch, c.vae_latent_channels, c.latent_size, c.latent_size, vin.ld,
                                 1.f / c.vae_scaling_factor, 0.f, 0, 0.f, 0.f, s));
}

// one guided forward step on instance k: z_in -> (z_next, x0, feat) ; energy accumulates into score, writes gfeat
void guided_forward(dd_engine* E, int k, const float* z_in, int step_index, const int* targets, int normalize, float weight,
                    float* score, hipStream_t s) {
  const dd_config& c = E->cfg;
  auto& I = E->inst[k];
  const int HW = c.latent_size * c.latent_size;
  unet_fwd(E, k, z_in, step_index, s);
  const Tn& out = E->unet.t[E->unet_out];
  HIPCHK(launch_cfg_ddim((const float*)(I.unet + out.off), out.ld, z_in, I.z_next, I.x0, c.max_batch, c.unet_out_channels, HW,
                         E->coef_table + (size_t)step_index * 8, s));
  vae_fwd(E, k, I.x0, s);
  guide_fwd_from_image(E, k, s);
  HIPCHK(launch_energy(I.feat, E->Pc, E->Pg, targets, c.max_batch, E->pD, E->pK, E->sp.gs, E->sp.ls, E->sp.use_global, E->sp.use_local,
                       normalize, weight, E->sample_w_set ? E->sample_w : nullptr, score, E->image_scores, I.gfeat, s));
}

// reverse of guided_forward: given g_znext (may be null) returns g_z (fp32 NCHW) in g_z_out
void guided_backward(dd_engine* E, int k, int step_index, const float* g_znext, float* g_z_out, float* g_x0_tmp, hipStream_t s) {
  const dd_config& c = E->cfg;
  const int HW = c.latent_size * c.latent_size;
  Run r{E, s, c.max_batch};
  guide_vae_bwd(E, k, g_x0_tmp, s);
  Ctx uc = r.ctx(E->unet, E->inst[k].unet);
  uc.step_index = step_index;
  const Tn& out = E->unet.t[E->unet_out];
  HIPCHK(launch_cfg_ddim_bwd(g_x0_tmp, g_znext, grad_ptr(uc, out), out.ld, g_z_out, c.max_batch, c.unet_out_channels, HW,
                             E->coef_table + (size_t)step_index * 8, s));
  run_bwd(E->unet, uc);
  const Tn& in = E->unet.t[E->unet_in];
  HIPCHK(launch_dup_bwd(grad_ptr(uc, in), in.ld, g_z_out, c.max_batch, c.unet_in_channels, HW, 1, in.B == 2 * c.max_batch ? 2 : 1, s));
}

void set_config_defaults(dd_config& c) {
  if (c.max_guidance_period < 1) c.max_guidance_period = 1;
  // same-binary A/B from the environment (tools/, bench.py --config sdxl); an engine built with the field set does not need it
  if (!c.unet_attn_fp8 && getenv("DD_ATTN_FP8")) c.unet_attn_fp8 = atoi(getenv("DD_ATTN_FP8")) != 0;
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
#define DD_TRY(E, ...)                                    \
  try { __VA_ARGS__; return DD_OK; }                      \
  catch (const std::exception& ex) { (E)->err = ex.what(); return DD_ERR_HIP; }

extern "C" {

int dd_abi_version(void) { return DD_ABI_VERSION; }

int dd_create(const dd_config* cfg, dd_engine** out) {
  if (!cfg || !out) return DD_ERR_ARG;
  if (cfg->abi_version != DD_ABI_VERSION) {
    fprintf(stderr, "dd_create: dd_config.abi_version %d, library built as %d (caller compiled against another include/distdiff_hip.h?)\n",
            cfg->abi_version, DD_ABI_VERSION);
    return DD_ERR_ARG;
  }
  if (cfg->unet_levels > DD_MAX_LEVELS || cfg->vae_levels > DD_MAX_LEVELS || cfg->guide_stages > DD_MAX_LEVELS || cfg->max_batch < 1)
    return DD_ERR_ARG;
  dd_engine* e = new dd_engine();
  e->cfg = *cfg;
  set_config_defaults(e->cfg);
  *out = e;
  return DD_OK;
}

void dd_destroy(dd_engine* e) {
  if (!e) return;
  for (auto& kv : e->step_graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
  if (e->gstream) hipStreamDestroy(e->gstream);
  if (e->ev_in) hipEventDestroy(e->ev_in);
  if (e->ev_out) hipEventDestroy(e->ev_out);
  for (void* p : e->dev_allocs) hipFree(p);
  delete e;
}

const char* dd_last_error(dd_engine* e) { return e ? e->err.c_str() : "null engine"; }

int dd_load_tensor(dd_engine* e, const char* model, const char* key, const float* data, int ndim, const int64_t* shape) {
  if (!e || !model || !key || !data || ndim < 1 || ndim > 4) return DD_ERR_ARG;
  if (e->finalized) { e->err = "dd_load_tensor after dd_finalize_weights"; return DD_ERR_STATE; }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  t.data.assign(data, data + t.numel());
  e->raw[std::string(model) + "/" + key] = std::move(t);
  return DD_OK;
}

int dd_declare_tensor(dd_engine* e, const char* model, const char* key, int ndim, const int64_t* shape) {
  if (!e || !model || !key || ndim < 1 || ndim > 4 || !shape) return DD_ERR_ARG;
  if (e->finalized) { e->err = "dd_declare_tensor after dd_finalize_weights"; return DD_ERR_STATE; }
  HostTensor t;
  t.shape.assign(shape, shape + ndim);
  e->raw[std::string(model) + "/" + key] = std::move(t);
  e->declared++;
  return DD_OK;
}

size_t dd_packed_bytes(dd_engine* e) {
  if (!e) return 0;
  size_t n = 0;
  for (auto& p : e->packed) n += rup_sz(p.second, 256);
  return n;
}

// the packed weights as one virtual byte array (each buffer padded to 256 bytes): copy [offset, offset + bytes) to / from `buf`
static int packed_copy(dd_engine* E, char* buf, size_t offset, size_t bytes, bool to_engine, hipStream_t s) {
  DD_TRY(E, {
    if (!E->finalized) throw std::runtime_error("dd_finalize_weights has not been called");
    size_t pos = 0;
    const size_t end = offset + bytes;
    for (auto& p : E->packed) {
      const size_t lo = std::max(pos, offset), hi = std::min(pos + p.second, end);
      if (lo < hi) {
        char* dev = p.first + (lo - pos);
        char* b = buf + (lo - offset);
        HIPCHK(hipMemcpyAsync(to_engine ? dev : b, to_engine ? b : dev, hi - lo, hipMemcpyDeviceToDevice, s));
      }
      pos += rup_sz(p.second, 256);
      if (pos >= end) break;
    }
  });
}
int dd_export_packed(dd_engine* e, void* dst, size_t offset, size_t bytes, void* stream) {
  if (!e || !dst) return DD_ERR_ARG;
  return packed_copy(e, (char*)dst, offset, bytes, false, (hipStream_t)stream);
}
int dd_import_packed(dd_engine* e, const void* src, size_t offset, size_t bytes, void* stream) {
  if (!e || !src) return DD_ERR_ARG;
  return packed_copy(e, (char*)src, offset, bytes, true, (hipStream_t)stream);
}

int dd_finalize_weights(dd_engine* E) {
  if (!E) return DD_ERR_ARG;
  if (E->finalized) { E->err = "already finalized"; return DD_ERR_STATE; }
  if (E->declared && E->declared != (int)E->raw.size()) { E->err = "dd_declare_tensor and dd_load_tensor cannot be mixed"; return DD_ERR_STATE; }
  E->shape_only = E->declared > 0;
  DD_TRY(E, {
    const dd_config& c = E->cfg;
    build_unet(E);
    build_vae(E);
    if (c.guide_kind == 1) build_guide_vit(E);
    else if (c.guide_kind == 2) build_guide_mbv2(E);
    else build_guide(E);
    const bool have_venc = E->has("vae", "encoder.conv_in.weight");
    const bool have_text = E->has("text", "text_model.embeddings.token_embedding.weight");
    if (have_venc) build_vae_encoder(E);
    const bool have_text2 = E->has("text2", "text_model.embeddings.token_embedding.weight");
    if (have_text2 && !have_text) throw std::runtime_error("a second text tower (text2) needs the first one (text)");
    if (have_text) build_text_encoder(E, 0);
    if (have_text2) build_text_encoder(E, 1);
    // time embedding MLP weights (fp32, setup-time only)
    auto up = [&](const char* key) {
      const HostTensor& t = E->get("unet", key);
      return (float*)E->wupload(t.data.data(), t.numel() * 4);
    };
    E->temb_w1 = up("time_embedding.linear_1.weight"); E->temb_b1 = up("time_embedding.linear_1.bias");
    E->temb_w2 = up("time_embedding.linear_2.weight"); E->temb_b2 = up("time_embedding.linear_2.bias");
    if (c.unet_add_time_dim > 0) {
      E->add_w1 = up("add_embedding.linear_1.weight"); E->add_b1 = up("add_embedding.linear_1.bias");
      E->add_w2 = up("add_embedding.linear_2.weight"); E->add_b2 = up("add_embedding.linear_2.bias");
    }
    E->raw.clear();
    // activation slabs
    const int P = c.enable_grad ? c.max_guidance_period : 1;
    const int B = c.max_batch, L = c.latent_size;
    const size_t zbytes = (size_t)B * std::max(c.unet_in_channels, c.vae_latent_channels) * L * L * 4;
    E->inst.resize(P);
    for (int k = 0; k < P; ++k) {
      auto& I = E->inst[k];
      I.unet = (char*)E->dmalloc(E->unet.act_bytes);
      if (k == 0 || c.enable_grad) {
        I.vae = (char*)E->dmalloc(E->vae.act_bytes);
        I.guide = (char*)E->dmalloc(E->guide.act_bytes);
      }
      I.z_in = (float*)E->dmalloc(zbytes); I.z_next = (float*)E->dmalloc(zbytes); I.x0 = (float*)E->dmalloc(zbytes);
      I.feat = (float*)E->dmalloc((size_t)B * guide_feat_dim(c) * 4);
      I.gfeat = (float*)E->dmalloc((size_t)B * guide_feat_dim(c) * 4);
    }
    E->tr_slab = (char*)E->dmalloc(2 * std::max({E->unet.tr_max, E->vae.tr_max, E->guide.tr_max, E->venc.tr_max, E->text.tr_max, E->text2.tr_max, (size_t)256}));
    if (c.enable_grad) {
      // UNet gradients alone; VAE and guide gradients live side by side (bicubic^T bridges them)
      const size_t g = std::max(E->unet.grad_bytes, E->vae.grad_bytes + E->guide.grad_bytes);
      E->grad_slab = (char*)E->dmalloc(g);
      for (auto& t : E->guide.t) t.goff += E->vae.grad_bytes;
    }
    // encoder programs run before the loop: the image encoder borrows the decoder slab of instance 0 when it fits
    if (have_venc) E->venc_slab = E->venc.act_bytes <= E->vae.act_bytes ? E->inst[0].vae : (char*)E->dmalloc(E->venc.act_bytes);
    if (have_text) E->text_slab = (char*)E->dmalloc(E->text.act_bytes);
    if (have_text2) E->text2_slab = (char*)E->dmalloc(E->text2.act_bytes);
    E->partial_cap = std::max({E->unet.scratch_partial, E->vae.scratch_partial, E->guide.scratch_partial, E->venc.scratch_partial,
                               E->text.scratch_partial, E->text2.scratch_partial, (size_t)1 << 20});
    E->scratch_partial = (char*)E->dmalloc(E->partial_cap, false);
    E->tmp_cap = std::max({E->unet.scratch_tmp, E->vae.scratch_tmp, E->guide.scratch_tmp, E->venc.scratch_tmp, E->text.scratch_tmp, E->text2.scratch_tmp, (size_t)256});
    if (!getenv("DD_ATTN_FLASH_ONLY")) {   // A/B switch: keep the flash kernels for wide heads too
      const int tap = (32 << 6) | 32;
      E->tap1x1 = (int*)E->dmalloc(sizeof(int), false);
      HIPCHK(hipMemcpy(E->tap1x1, &tap, sizeof(int), hipMemcpyHostToDevice));
    }
    E->scratch_tmp = (char*)E->dmalloc(E->tmp_cap);
    const int maxG = std::max(c.unet_groups, c.vae_groups);
    E->gn_scratch = (float*)E->dmalloc(groupnorm_scratch_bytes(2 * B, maxG), false);
    E->rowpart = (float*)E->dmalloc(std::max(E->unet.scratch_rowpart, (size_t)256), false);
    {   // GroupNorm affine of the op that ran last (CF_GNFOLD): [images][channels][2] of the widest GroupNorm input
      size_t n = 256;
      for (const Program* P : {&E->unet, &E->vae, &E->venc})
        for (const Op& o : P->ops)
          if (o.kind == OP_GN) n = std::max(n, (size_t)P->t[o.x].B * P->t[o.x].C * 8);
      E->gn_coef = (float*)E->dmalloc(n, false);
    }
    for (auto& f : E->f32_tmp) f = (float*)E->dmalloc(zbytes);
    E->img_tmp = (float*)E->dmalloc((size_t)B * 3 * 64 * L * L * 4);
    E->score_tmp = (float*)E->dmalloc(256);
    E->sample_w = (float*)E->dmalloc((size_t)B * 4);
    E->image_scores = (float*)E->dmalloc((size_t)B * 4);
    // cross-attention K/V buffers
    E->ctx_bf16 = (bf16_t*)E->dmalloc((size_t)2 * B * c.text_len * rup(c.unet_cross_dim, 8) * 2);
    for (auto& sl : E->cross_slots) {
      bf16_t* k = (bf16_t*)E->dmalloc((size_t)2 * B * c.text_len * sl.C * 2);
      bf16_t* v = (bf16_t*)E->dmalloc((size_t)2 * B * c.text_len * sl.C * 2);
      E->cross_kv.push_back({k, v});
    }
    HIPCHK(hipDeviceSynchronize());
    E->finalized = true;
  });
}

int dd_set_schedule(dd_engine* E, const int* timesteps, int n, const float* alphas_cumprod, int num_train, float final_alpha,
                    const dd_sampler_params* sp) {
  if (!E || !timesteps || n < 1 || !alphas_cumprod || !sp) return DD_ERR_ARG;
  if (!E->finalized) { E->err = "finalize first"; return DD_ERR_STATE; }
  DD_TRY(E, {
    const dd_config& c = E->cfg;
    E->timesteps.assign(timesteps, timesteps + n);
    E->sp = *sp;
    HIPCHK(hipDeviceSynchronize());
    for (void* q : E->sched_allocs) E->dfree(q);
    E->sched_allocs.clear();
    std::vector<float> coef((size_t)n * 8, 0.f);
    const int ratio = num_train / n;
    for (int i = 0; i < n; ++i) {
      const int t = timesteps[i], prev = t - ratio;
      if (t < 0 || t >= num_train) throw std::runtime_error("timestep out of range");
      const double a = alphas_cumprod[t], ap = prev >= 0 ? alphas_cumprod[prev] : final_alpha;
      float* q = &coef[(size_t)i * 8];
      q[0] = sp->guidance_scale; q[1] = (float)sqrt(a); q[2] = (float)sqrt(1 - a); q[3] = (float)sqrt(ap); q[4] = (float)sqrt(1 - ap);
    }
    E->coef_table = (float*)E->dmalloc(coef.size() * 4, false);
    E->sched_allocs.push_back(E->coef_table);
    HIPCHK(hipMemcpy(E->coef_table, coef.data(), coef.size() * 4, hipMemcpyHostToDevice));
    // sinusoidal timestep embedding (diffusers Timesteps: flip_sin_to_cos, freq_shift) on host, MLP + projections on GPU (fp32)
    const int C0 = c.unet_block_out_channels[0], half = C0 / 2, TE = C0 * 4;
    std::vector<float> sinus((size_t)n * C0);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < half; ++j) {
        const float freq = expf(-logf(10000.f) * (float)j / ((float)half - c.unet_freq_shift));
        const float a = (float)timesteps[i] * freq;
        const float sn = sinf(a), cs = cosf(a);
        if (c.unet_flip_sin_to_cos) { sinus[(size_t)i * C0 + j] = cs; sinus[(size_t)i * C0 + half + j] = sn; }
        else { sinus[(size_t)i * C0 + j] = sn; sinus[(size_t)i * C0 + half + j] = cs; }
      }
    float* d_sin = (float*)E->dmalloc(sinus.size() * 4, false);
    HIPCHK(hipMemcpy(d_sin, sinus.data(), sinus.size() * 4, hipMemcpyHostToDevice));
    float* d_h1 = (float*)E->dmalloc((size_t)n * TE * 4, false);
    float* d_emb = (float*)E->dmalloc((size_t)n * TE * 4, false);
    E->sched_allocs.push_back(d_sin); E->sched_allocs.push_back(d_h1); E->sched_allocs.push_back(d_emb);
    E->d_emb = d_emb;
    E->added_cond_set = false;
    HIPCHK(launch_linear_f32(d_sin, E->temb_w1, E->temb_b1, d_h1, n, TE, C0, 0, nullptr));
    HIPCHK(launch_linear_f32(d_h1, E->temb_w2, E->temb_b2, d_emb, n, TE, TE, 1, nullptr));
    for (ConvW* cw : E->temb_convs) {
      cw->bias_table = (float*)E->dmalloc((size_t)n * cw->Cout * 4, false);
      E->sched_allocs.push_back(cw->bias_table);
      HIPCHK(launch_linear_f32(d_emb, cw->temb_w, cw->temb_b, cw->bias_table, n, cw->Cout, TE, 1, nullptr));
      if (cw->bias) HIPCHK(launch_add_rowvec_f32(cw->bias_table, cw->bias, n, cw->Cout, nullptr));   // + conv1.bias
      if (c.unet_add_time_dim > 0) {
        cw->bias_table_img = (float*)E->dmalloc((size_t)n * 2 * c.max_batch * cw->Cout * 4, false);
        E->sched_allocs.push_back(cw->bias_table_img);
      }
    }
    HIPCHK(hipDeviceSynchronize());
  });
}

int dd_set_prototypes(dd_engine* E, const float* Pc, const float* Pg, int C, int K, int D) {
  if (!E || C < 1 || D < 1) return DD_ERR_ARG;
  DD_TRY(E, {
    E->pC = C; E->pK = K; E->pD = D;
    HIPCHK(hipDeviceSynchronize());
    E->dfree(E->Pc); E->dfree(E->Pg);
    E->Pc = nullptr; E->Pg = nullptr;
    if (Pc) { E->Pc = (float*)E->dmalloc((size_t)C * D * 4, false); HIPCHK(hipMemcpy(E->Pc, Pc, (size_t)C * D * 4, hipMemcpyHostToDevice)); }
    if (Pg) { E->Pg = (float*)E->dmalloc((size_t)C * K * D * 4, false); HIPCHK(hipMemcpy(E->Pg, Pg, (size_t)C * K * D * 4, hipMemcpyHostToDevice)); }
  });
}

int dd_set_prompt(dd_engine* E, const float* embeds, int B, void* stream) {
  if (!E || !embeds) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int rows = 2 * B * c.text_len, ld = rup(c.unet_cross_dim, 8);
    // [rows, cross_dim] fp32 -> bf16 rows (treated as NCHW with H*W = 1 per row)
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(embeds, E->ctx_bf16, rows, c.unet_cross_dim, 1, 1, ld, ld, 0, 1.f, s));
    for (size_t i = 0; i < E->cross_slots.size(); ++i) {
      auto& sl = E->cross_slots[i];
      for (int which = 0; which < 2; ++which) {
        ConvW* w = which ? sl.wv : sl.wk;
        ConvGemmParams p; memset(&p, 0, sizeof p);
        p.alpha = 1.f; p.partial = (float*)E->scratch_partial;
        p.x = E->ctx_bf16; p.x_ld = ld; p.w = w->w_fwd; p.taptab = w->tap_fwd;
        p.y = which ? E->cross_kv[i].second : E->cross_kv[i].first; p.y_ld = sl.C;
        p.B = 1; p.H = rows; p.W = 1; p.Ho = rows; p.Wo = 1; p.stride = 1;
        p.cin = w->sf.cin; p.ntaps = w->sf.ntaps; p.M = rows; p.N = w->sf.N; p.K = w->sf.K;
        HIPCHK(launch_conv_gemm(p, E->partial_cap, s));
      }
    }
  });
}

int dd_set_added_cond(dd_engine* E, const float* text_embeds, const float* time_ids, int B, void* stream) {
  if (!E || !text_embeds || !time_ids) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (c.unet_add_time_dim <= 0) throw std::runtime_error("this UNet has no additional conditioning (unet_add_time_dim == 0)");
    if (!E->d_emb) throw std::runtime_error("dd_set_schedule first");
    hipStream_t s = (hipStream_t)stream;
    const int nb = 2 * B, n = (int)E->timesteps.size(), TE = c.unet_block_out_channels[0] * 4;
    const int Dt = c.unet_add_text_dim, Ds = 6 * c.unet_add_time_dim, Din = Dt + Ds;
    // scratch (setup path: plain allocations, freed at the end)
    float* cat = nullptr; float* h1 = nullptr; float* aug = nullptr; float* X = nullptr;
    HIPCHK(hipMalloc((void**)&cat, (size_t)nb * Din * 4)); HIPCHK(hipMalloc((void**)&h1, (size_t)nb * TE * 4));
    HIPCHK(hipMalloc((void**)&aug, (size_t)nb * TE * 4)); HIPCHK(hipMalloc((void**)&X, (size_t)n * nb * TE * 4));
    // cat[text_embeds, add_time_proj(time_ids).reshape(B, -1)]
    HIPCHK(hipMemcpy2DAsync(cat, (size_t)Din * 4, text_embeds, (size_t)Dt * 4, (size_t)Dt * 4, nb, hipMemcpyDeviceToDevice, s));
    for (int k = 0; k < 6; ++k) {
      // time id k of every image -> columns [Dt + k*dim, Dt + (k+1)*dim): gather the k-th id with a strided copy first
      HIPCHK(hipMemcpy2DAsync(h1, 4, time_ids + k, 6 * 4, 4, nb, hipMemcpyDeviceToDevice, s));
      HIPCHK(launch_sinusoid_f32(h1, cat, nb, c.unet_add_time_dim, Din, Dt + k * c.unet_add_time_dim, 1, s));
    }
    HIPCHK(launch_linear_f32(cat, E->add_w1, E->add_b1, h1, nb, TE, Din, 0, s));
    HIPCHK(launch_linear_f32(h1, E->add_w2, E->add_b2, aug, nb, TE, TE, 1, s));
    HIPCHK(launch_add_outer_f32(E->d_emb, aug, X, n, nb, TE, s));                 // emb[step, image] = time_embedding(t) + aug_emb
    for (ConvW* cw : E->temb_convs) {
      HIPCHK(launch_linear_f32(X, cw->temb_w, cw->temb_b, cw->bias_table_img, n * nb, cw->Cout, TE, 1, s));
      if (cw->bias) HIPCHK(launch_add_rowvec_f32(cw->bias_table_img, cw->bias, n * nb, cw->Cout, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    hipFree(cat); hipFree(h1); hipFree(aug); hipFree(X);
    E->added_cond_set = true;
  });
}

int dd_add_noise(dd_engine* E, const float* x, const float* noise, float* out, int B, int step_index, void* stream) {
  if (!E || !x || !noise || !out) return DD_ERR_ARG;
  DD_TRY(E, {
    if (step_index < 0 || step_index >= (int)E->timesteps.size()) throw std::runtime_error("step_index out of range");
    const dd_config& c = E->cfg;
    // coef_table[step][1..2] = sqrt(a_t), sqrt(1-a_t)
    HIPCHK(launch_axpby(x, noise, out, (size_t)B * c.unet_in_channels * c.latent_size * c.latent_size,
                        E->coef_table + (size_t)step_index * 8 + 1, (hipStream_t)stream));
  });
}

int dd_unet_forward(dd_engine* E, const float* z, int step_index, float* eps2_out, int B, void* stream) {
  if (!E || !z || !eps2_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    unet_fwd(E, 0, z, step_index, s);
    const Tn& out = E->unet.t[E->unet_out];
    HIPCHK(launch_nhwc_to_nchw_f32(E->inst[0].unet + out.off, 1, eps2_out, 2 * B, c.unet_out_channels, c.latent_size, c.latent_size, out.ld,
                                   1.f, 0.f, 0, 0.f, 0.f, s));
  });
}

// the launch sequence of one plain step: UNet forward (no stash) + CFG + DDIM
static void denoise_step_enqueue(dd_engine* E, const float* z, int step_index, float* z_prev_out, float* x0_out, hipStream_t s) {
  const dd_config& c = E->cfg;
  unet_fwd(E, 0, z, step_index, s, /*stash=*/false);
  const Tn& out = E->unet.t[E->unet_out];
  HIPCHK(launch_cfg_ddim((const float*)(E->inst[0].unet + out.off), out.ld, z, z_prev_out, x0_out, c.max_batch, c.unet_out_channels,
                         c.latent_size * c.latent_size, E->coef_table + (size_t)step_index * 8, s));
}

int dd_denoise_step(dd_engine* E, const float* z, int step_index, float* z_prev_out, float* x0_out, int B, void* stream) {
  if (!E || !z || !z_prev_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    if (step_index < 0 || step_index >= (int)E->timesteps.size()) throw std::runtime_error("step_index out of range");
    hipStream_t s = (hipStream_t)stream;
    // Opt-in (DD_GRAPH=1): measured on MI355X the replay does not pay -- B = 16: 1485 vs 1481 ms per batch, B = 1: 313 vs 302 ms
    // (DESIGN.md section 6): even at B = 1 the ~6300 launches of an image are GPU-bound small kernels, not launch-bound.
    static const bool use_graph = getenv("DD_GRAPH") != nullptr && atoi(getenv("DD_GRAPH")) != 0;
    if (!use_graph || !E->graphs_ok || E->prof.on) { denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s); return DD_OK; }
    // first sighting of a (step, buffers) key: plain launches (also runs every one-time hipFuncSetAttribute); second: capture on the
    // engine's own stream (the caller's may be the legacy default stream, which cannot be captured) and instantiate; then replay
    char key[96];
    snprintf(key, sizeof key, "%d/%p/%p/%p", step_index, (const void*)z, (void*)z_prev_out, (void*)x0_out);
    auto it = E->step_graphs.find(key);
    if (it == E->step_graphs.end()) {
      if (E->step_graphs.size() >= 256) { denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s); return DD_OK; }
      dd_engine::StepGraph g;
      const double f0 = E->flops;
      denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s);
      g.flops = E->flops - f0; g.seen = 1;
      E->step_graphs[key] = g;
      return DD_OK;
    }
    dd_engine::StepGraph& g = it->second;
    if (!E->gstream) {
      HIPCHK(hipStreamCreateWithFlags(&E->gstream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&E->ev_in, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&E->ev_out, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(E->ev_in, s));
    HIPCHK(hipStreamWaitEvent(E->gstream, E->ev_in, 0));
    if (!g.exec) {
      hipGraph_t graph = nullptr;
      const double f0 = E->flops;
      bool ok = hipStreamBeginCapture(E->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        try { denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, E->gstream); } catch (...) { ok = false; }
        ok = (hipStreamEndCapture(E->gstream, &graph) == hipSuccess) && ok && graph;
      }
      E->flops = f0;
      if (ok) ok = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) hipGraphDestroy(graph);
      if (!ok) {   // capture is an optimisation only: fall back to plain launches for good
        (void)hipGetLastError();
        g.exec = nullptr; E->graphs_ok = false;
        denoise_step_enqueue(E, z, step_index, z_prev_out, x0_out, s);
        return DD_OK;
      }
    }
    HIPCHK(hipGraphLaunch(g.exec, E->gstream));
    E->flops += g.flops;
    HIPCHK(hipEventRecord(E->ev_out, E->gstream));
    HIPCHK(hipStreamWaitEvent(s, E->ev_out, 0));
  });
}

int dd_decode(dd_engine* E, const float* z, float* image_out, int denormalize, int B, void* stream) {
  if (!E || !z || !image_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    vae_fwd(E, 0, z, s);
    const Tn& img = E->vae.t[E->vae_out];
    HIPCHK(launch_nhwc_to_nchw_f32(E->inst[0].vae + img.off, 1, image_out, B, c.vae_out_channels, img.H, img.W, img.ld,
                                   denormalize ? 0.5f : 1.f, denormalize ? 0.5f : 0.f, denormalize, 0.f, 1.f, s));
  });
}

int dd_vae_encode(dd_engine* E, const float* images, const float* noise, float* latents_out, float* moments_out, int B, void* stream) {
  if (!E || !images || !latents_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    if (!E->venc_slab) throw std::runtime_error("no VAE encoder weights were loaded (vae/encoder.* keys)");
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    Run r{E, s, B};
    Ctx ctx = r.ctx(E->venc, E->venc_slab);
    ctx.stash = false;
    const Tn& in = E->venc.t[E->venc_in];
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(images, act_ptr(ctx, in), B, c.vae_out_channels, in.H, in.W, in.ld, in.ld, 0, 1.f, s));
    run_fwd(E->venc, ctx);
    const Tn& mo = E->venc.t[E->venc_out];
    HIPCHK(launch_vae_sample((const float*)(ctx.act + mo.off), mo.ld, noise, latents_out, moments_out, B, c.vae_latent_channels,
                             mo.H * mo.W, c.vae_scaling_factor, s));
  });
}

int dd_text_encode_tower(dd_engine* E, int which, const int* input_ids, float* hidden_out, float* pooled_out, int n, void* stream) {
  if (!E || !input_ids || (!hidden_out && !pooled_out) || which < 0 || which > 1) return DD_ERR_ARG;
  DD_TRY(E, {
    if (!E->finalized) throw std::runtime_error("dd_finalize_weights has not been called");
    char* slab = which ? E->text2_slab : E->text_slab;
    if (!slab) throw std::runtime_error(which ? "no second text tower was loaded (text2/text_model.* keys)" : "no text encoder weights were loaded (text/text_model.* keys)");
    if (n < 1 || n > E->text_batch) throw std::runtime_error("dd_text_encode: n must be in [1, 2*max_batch]");
    if (pooled_out && (!which || !E->text2_proj_w)) throw std::runtime_error("pooled text embeddings come from the second tower's text_projection (text2/text_projection.weight)");
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    Program& P = which ? E->text2 : E->text;
    const int T = c.text_len, Cc = which ? E->text2_hidden : E->text_hidden;
    HIPCHK(hipMemsetAsync(E->text_ids, 0, (size_t)E->text_batch * T * 4, s));
    HIPCHK(hipMemcpyAsync(E->text_ids, input_ids, (size_t)n * T * 4, hipMemcpyDeviceToDevice, s));
    Run r{E, s, E->text_batch};
    Ctx ctx = r.ctx(P, slab);
    ctx.stash = false;
    const Tn& in = P.t[which ? E->text2_in : E->text_in];
    HIPCHK(launch_clip_embed(E->text_ids, which ? E->tok_emb2 : E->tok_emb, which ? E->pos_emb2 : E->pos_emb, act_ptr(ctx, in), in.ld, in.rows, T, Cc,
                             which ? E->text2_vocab : E->text_vocab, s));
    run_fwd(P, ctx);
    if (hidden_out) {
      const Tn& o = P.t[which ? E->text2_out : E->text_out];
      HIPCHK(launch_rows_bf16_to_f32(act_ptr(ctx, o), o.ld, hidden_out, n * T, Cc, s));
    }
    if (pooled_out) {
      const Tn& f = P.t[E->text2_final];
      HIPCHK(launch_clip_pool_project(E->text_ids, act_ptr(ctx, f), f.ld, E->text2_proj_w, pooled_out, n, T, Cc, E->text2_proj, s));
    }
  });
}

int dd_text_encode(dd_engine* E, const int* input_ids, float* embeds_out, int n, void* stream) {
  if (!E || !input_ids || !embeds_out) return DD_ERR_ARG;
  if (E->text2_slab) { E->err = "this model has two text towers: use dd_text_encode_tower"; return DD_ERR_STATE; }
  return dd_text_encode_tower(E, 0, input_ids, embeds_out, nullptr, n, stream);
}

int dd_guide_encode_pooled(dd_engine* E, const float* images, float* feats, int B, int use_max, void* stream);
int dd_guide_encode(dd_engine* E, const float* images, float* feats, int B, void* stream) {
  return dd_guide_encode_pooled(E, images, feats, B, 0, stream);
}
int dd_guide_encode_pooled(dd_engine* E, const float* images, float* feats, int B, int use_max, void* stream) {
  if (!E || !images || !feats) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    if (use_max && E->cfg.guide_kind == 1) throw std::runtime_error("the ViT guide has no spatial pooling (encode_image = the projected class token)");
    const dd_config& c = E->cfg;
    hipStream_t s = (hipStream_t)stream;
    Run r{E, s, B};
    Ctx gc = r.ctx(E->guide, E->inst[0].guide);
    const Tn& gin = E->guide.t[E->guide_in];
    HIPCHK(launch_nchw_to_nhwc_f32(images, act_f32(gc, gin), B, 3, gin.H, gin.W, gin.ld, gin.ld, s));
    run_fwd(E->guide, gc);
    guide_features_out(E, gc, feats, s, use_max);
  });
}

int dd_transform_guidance(dd_engine* E, const float* z, const int* targets, const float* ch_e, const float* ch_b, int first_step_index,
                          int P, float* z_out, float* score_out, float* grad_eb_out, int B, void* stream) {
  if (!E || !z || !targets || !ch_e || !ch_b || !z_out || P < 1) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    if (P > (int)E->inst.size()) throw std::runtime_error("guidance period exceeds max_guidance_period");
    if (first_step_index < 0 || first_step_index + P > (int)E->timesteps.size()) throw std::runtime_error("guide steps out of range");
    if ((E->sp.use_global && !E->Pc) || (E->sp.use_local && !E->Pg)) throw std::runtime_error("prototypes not set");
    hipStream_t s = (hipStream_t)stream;
    const int BC = B * c.unet_in_channels, HW = c.latent_size * c.latent_size;
    float* score = score_out ? score_out : E->score_tmp;
    HIPCHK(hipMemsetAsync(score, 0, sizeof(float), s));
    HIPCHK(hipMemsetAsync(E->image_scores, 0, (size_t)B * sizeof(float), s));
    // z0 = z*(1+e)+b  (generate_data.py:696)
    HIPCHK(launch_affine(z, ch_e, ch_b, E->inst[0].z_in, BC, HW, s));
    const float weight = 1.f / (float)E->sp.guidance_period;   // score / args.guidance_period (:719)
    for (int k = 0; k < P; ++k) {
      const float* zin = E->inst[k].z_in;
      guided_forward(E, k, zin, first_step_index + k, targets, 0, weight, score, s);
      if (k + 1 < P) HIPCHK(hipMemcpyAsync(E->inst[k + 1].z_in, E->inst[k].z_next, (size_t)BC * HW * 4, hipMemcpyDeviceToDevice, s));
    }
    float* g_next = nullptr;
    float* gbuf[2] = {E->f32_tmp[0], E->f32_tmp[1]};
    for (int k = P - 1; k >= 0; --k) {
      float* g_z = gbuf[k & 1];
      guided_backward(E, k, first_step_index + k, g_next, g_z, E->f32_tmp[2], s);
      g_next = g_z;
    }
    if (grad_eb_out) HIPCHK(hipMemcpyAsync(grad_eb_out, g_next, (size_t)BC * HW * 4, hipMemcpyDeviceToDevice, s));
    // e -= rho*ge ; b -= rho*gb ; z' = clamp(z*(1+e)+b, z-c, z+c)  (:721-728)
    HIPCHK(launch_transform_update(z, g_next, ch_e, ch_b, z_out, BC, HW, E->sp.rho, E->sp.constraint_value, s));
  });
}

int dd_direct_guidance(dd_engine* E, const float* z, const int* targets, int step_index, float* z_next_out, float* x0_out,
                       float* score_out, float* grad_z_out, int B, void* stream) {
  if (!E || !z || !targets || !z_next_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    if (step_index < 0 || step_index >= (int)E->timesteps.size()) throw std::runtime_error("step_index out of range");
    if ((E->sp.use_global && !E->Pc) || (E->sp.use_local && !E->Pg)) throw std::runtime_error("prototypes not set");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * c.unet_in_channels * c.latent_size * c.latent_size;
    float* score = score_out ? score_out : E->score_tmp;
    HIPCHK(hipMemsetAsync(score, 0, sizeof(float), s));
    HIPCHK(hipMemsetAsync(E->image_scores, 0, (size_t)B * sizeof(float), s));
    HIPCHK(hipMemcpyAsync(E->inst[0].z_in, z, n * 4, hipMemcpyDeviceToDevice, s));
    guided_forward(E, 0, E->inst[0].z_in, step_index, targets, 1, 1.f, score, s);
    float* g_z = E->f32_tmp[0];
    guided_backward(E, 0, step_index, nullptr, g_z, E->f32_tmp[2], s);
    if (grad_z_out) HIPCHK(hipMemcpyAsync(grad_z_out, g_z, n * 4, hipMemcpyDeviceToDevice, s));
    if (x0_out) HIPCHK(hipMemcpyAsync(x0_out, E->inst[0].x0, n * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(launch_sub_scaled(E->inst[0].z_next, g_z, z_next_out, n, E->sp.rho, s));   // :762
  });
}

int dd_expand(dd_engine* E, const dd_expand_args* a, void* stream) {
  if (!E || !a || !a->image_latents || !a->noise || !a->z_out) return DD_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)E->timesteps.size();
  if (a->start_index < 0 || a->start_index >= n) { E->err = "start_index out of range"; return DD_ERR_ARG; }
  int rc = dd_add_noise(E, a->image_latents, a->noise, E->f32_tmp[3], a->B, a->start_index, stream);
  if (rc) return rc;
  float* cur = E->f32_tmp[3];
  float* nxt = E->f32_tmp[4];
  for (int i = a->start_index; i < n; ++i) {
    if (a->guidance_type == 1 && a->guide_count > 0 && i == a->guide_first) {
      // transform guidance at t == guide_timesteps[0], then the step is executed again from the corrected latent (:1203-1207)
      rc = dd_transform_guidance(E, cur, a->targets, a->e, a->b, a->guide_first, a->guide_count, nxt, a->score_out, nullptr, a->B, stream);
      if (rc) return rc;
      std::swap(cur, nxt);
      rc = dd_denoise_step(E, cur, i, nxt, nullptr, a->B, stream);
    } else if (a->guidance_type == 2 && i >= a->guide_first && i < a->guide_first + a->guide_count) {
      rc = dd_direct_guidance(E, cur, a->targets, i, nxt, nullptr, a->score_out, nullptr, a->B, stream);
    } else {
      rc = dd_denoise_step(E, cur, i, nxt, nullptr, a->B, stream);
    }
    if (rc) return rc;
    std::swap(cur, nxt);
  }
  try {
    const dd_config& c = E->cfg;
    HIPCHK(hipMemcpyAsync(a->z_out, cur, (size_t)a->B * c.unet_in_channels * c.latent_size * c.latent_size * 4, hipMemcpyDeviceToDevice, s));
  } catch (const std::exception& ex) { E->err = ex.what(); return DD_ERR_HIP; }
  if (a->image_out) return dd_decode(E, cur, a->image_out, 1, a->B, stream);
  return DD_OK;
}

int dd_set_sample_weights(dd_engine* E, const float* w_host, int B) {
  if (!E) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    E->sample_w_set = w_host != nullptr;
    if (w_host) {
      HIPCHK(hipDeviceSynchronize());   // a previous guidance call may still be reading the weights
      HIPCHK(hipMemcpy(E->sample_w, w_host, (size_t)B * sizeof(float), hipMemcpyHostToDevice));
    }
  });
}

int dd_get_image_scores(dd_engine* E, float* scores_out, int B, void* stream) {
  if (!E || !scores_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    HIPCHK(hipMemcpyAsync(scores_out, E->image_scores, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  });
}

// ---- per-module VJP diagnostics (parity tests of the hand-derived reverse programs against torch.autograd) ----
int dd_unet_vjp(dd_engine* E, const float* z, int step_index, const float* g_eps2, float* g_z_out, int B, void* stream) {
  if (!E || !z || !g_eps2 || !g_z_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    hipStream_t s = (hipStream_t)stream;
    const int HW = c.latent_size * c.latent_size;
    unet_fwd(E, 0, z, step_index, s);
    Run r{E, s, B};
    Ctx uc = r.ctx(E->unet, E->inst[0].unet);
    uc.step_index = step_index;
    const Tn& out = E->unet.t[E->unet_out];
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(g_eps2, grad_ptr(uc, out), 2 * B, c.unet_out_channels, c.latent_size, c.latent_size, out.ld, out.ld, 0,
                                        1.f, s));
    run_bwd(E->unet, uc);
    const Tn& in = E->unet.t[E->unet_in];
    HIPCHK(launch_dup_bwd(grad_ptr(uc, in), in.ld, g_z_out, B, c.unet_in_channels, HW, 0, in.B == 2 * B ? 2 : 1, s));
  });
}

int dd_decode_vjp(dd_engine* E, const float* z, const float* g_image, float* g_z_out, int B, void* stream) {
  if (!E || !z || !g_image || !g_z_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    hipStream_t s = (hipStream_t)stream;
    vae_fwd(E, 0, z, s);
    Run r{E, s, B};
    Ctx vc = r.ctx(E->vae, E->inst[0].vae);
    const Tn& img = E->vae.t[E->vae_out];
    HIPCHK(launch_nchw_f32_to_nhwc_bf16(g_image, grad_ptr(vc, img), B, c.vae_out_channels, img.H, img.W, img.ld, img.ld, 0, 1.f, s));
    run_bwd(E->vae, vc);
    const Tn& vin = E->vae.t[E->vae_in];
    HIPCHK(launch_nhwc_to_nchw_f32(grad_ptr(vc, vin), 0, g_z_out, B, c.vae_latent_channels, c.latent_size, c.latent_size, vin.ld,
                                   1.f / c.vae_scaling_factor, 0.f, 0, 0.f, 0.f, s));
  });
}

int dd_guide_vjp(dd_engine* E, const float* images, const float* g_feats, float* g_images_out, int B, void* stream) {
  if (!E || !images || !g_feats || !g_images_out) return DD_ERR_ARG;
  DD_TRY(E, {
    check_batch(E, B);
    const dd_config& c = E->cfg;
    if (!c.enable_grad) throw std::runtime_error("engine created with enable_grad=0");
    hipStream_t s = (hipStream_t)stream;
    Run r{E, s, B};
    Ctx gc = r.ctx(E->guide, E->inst[0].guide);
    const Tn& gin = E->guide.t[E->guide_in];
    HIPCHK(launch_nchw_to_nhwc_f32(images, act_f32(gc, gin), B, 3, gin.H, gin.W, gin.ld, gin.ld, s));
    run_fwd(E->guide, gc);
    guide_features_grad_in(E, gc, g_feats, s);
    run_bwd(E->guide, gc);
    HIPCHK(launch_nhwc_to_nchw_f32(grad_f32(gc, gin), 1, g_images_out, B, 3, gin.H, gin.W, gin.ld, 1.f, 0.f, 0, 0.f, 0.f, s));
  });
}

int dd_profile_enable(dd_engine* E, int on) {
  if (!E) return DD_ERR_ARG;
  E->prof.on = on != 0;
  E->prof.used = 0;
  E->prof.chain = false;
  E->prof.recs.clear();
  return DD_OK;
}
// out[fam*3 + {0,1,2}] = {total ms, algorithmic flops, launches(op count)} for fam in {conv_gemm, attention, norm, other}
int dd_profile_read(dd_engine* E, double* out12) {
  if (!E || !out12) return DD_ERR_ARG;
  DD_TRY(E, {
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < 12; ++i) out12[i] = 0;
    FILE* dump = getenv("DD_PROFILE_DUMP") ? fopen(getenv("DD_PROFILE_DUMP"), "w") : nullptr;
    if (dump) fprintf(dump, "fam,bwd,M,N,K,flops,ms\n");
    for (auto& r : E->prof.recs) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, E->prof.pool[r.e0], E->prof.pool[r.e1]));
      if (dump) fprintf(dump, "%d,%d,%d,%d,%d,%.0f,%.5f\n", r.fam, r.bwd, r.M, r.N, r.K, r.flops, ms);
      out12[r.fam * 3 + 0] += ms; out12[r.fam * 3 + 1] += r.flops; out12[r.fam * 3 + 2] += 1;
    }
    if (dump) fclose(dump);
    E->prof.used = 0; E->prof.chain = false; E->prof.recs.clear();
  });
}

// ---- debug introspection: copy activation / gradient of tensor `idx` of program `prog` (0 unet, 1 vae, 2 guide; + 16 * instance)
// to a HOST fp32 buffer [rows*ld]; info4 = {rows, C, ld, is_f32}. Synchronises.
int dd_debug_tensor(dd_engine* E, int prog_inst, int idx, int want_grad, float* host_out, int* info4) {
  const int prog = prog_inst & 15, k = prog_inst >> 4;   // bits 4.. select the instance (chained guided step) of an activation
  if (!E || prog < 0 || prog > 2 || k < 0 || k >= (int)E->inst.size()) return DD_ERR_ARG;
  DD_TRY(E, {
    Program& P = prog == 0 ? E->unet : prog == 1 ? E->vae : E->guide;
    if (idx < 0) idx += (int)P.t.size();   // negative: from the end (-1 = the program's output tensor)
    if (idx < 0 || idx >= (int)P.t.size()) throw std::runtime_error("tensor index out of range");
    const Tn& t = P.t[idx];
    if (info4) { info4[0] = t.rows; info4[1] = t.C; info4[2] = t.ld; info4[3] = t.f32 ? 1 : 0; }
    if (!host_out) return DD_OK;
    // the gradient slab is packed by liveness: after a run only the pinned ranges (program inputs / outputs, padded tensors) still hold
    // the tensor's own gradient -- an intermediate's range may have been reused by another tensor since
    if (want_grad && t.grad && !(t.glo < 0 && t.ghi >= (int)P.ops.size()) && !getenv("DD_NO_GRAD_REUSE"))
      throw std::runtime_error("gradient of an intermediate tensor: its range of the liveness-packed slab is reused during the reverse run "
                               "(build the engine with DD_NO_GRAD_REUSE=1 to read it)");
    HIPCHK(hipDeviceSynchronize());
    const dd_engine::Inst& I = E->inst[k];
    char* slab = prog == 0 ? I.unet : prog == 1 ? I.vae : I.guide;
    if (!want_grad && !slab) throw std::runtime_error("this instance has no slab for that program");
    char* base = want_grad ? E->grad_slab + t.goff : t.transient ? E->tr_slab + (size_t)t.tr_slot * P.tr_max : slab + t.off;
    const size_t n = (size_t)t.rows * t.ld;
    if ((t.f32 && !want_grad) || (want_grad && t.gf32)) { HIPCHK(hipMemcpy(host_out, base, n * 4, hipMemcpyDeviceToHost)); }
    else {
      std::vector<bf16_t> tmp(n);
      HIPCHK(hipMemcpy(tmp.data(), base, n * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) host_out[i] = host_bf2f(tmp[i]);
    }
  });
}
// parity-test hook: image DEVICE fp32 [B,3,8L,8L] (not denormalised) replaces the decoder's output in front of the bicubic resize of
// every later guided forward (the gradient still flows through the decoder); NULL switches it off.  The caller keeps the buffer alive.
int dd_debug_set_images(dd_engine* E, const float* images, int count) {
  if (!E || count < 0) return DD_ERR_ARG;
  E->image_override = count > 0 ? images : nullptr;
  E->image_override_count = E->image_override ? count : 0;
  return DD_OK;
}
int dd_debug_set_image(dd_engine* E, const float* image) { return dd_debug_set_images(E, image, 1); }
int dd_debug_num_tensors(dd_engine* E, int prog) {
  if (!E || prog < 0 || prog > 2) return DD_ERR_ARG;
  return (int)(prog == 0 ? E->unet : prog == 1 ? E->vae : E->guide).t.size();
}

// output stage (generate_data.py:1227-1234): [B,3,H,W] fp32 in [0,1] -> uint8 HWC with torchvision.save_image's quantisation
int dd_image_to_u8(dd_engine* E, const float* image, uint8_t* out_hwc, int B, void* stream) {
  if (!E || !image || !out_hwc || B < 1) return DD_ERR_ARG;
  DD_TRY(E, {
    const dd_config& c = E->cfg;
    HIPCHK(launch_to_uint8(image, out_hwc, B, c.vae_out_channels, 8 * c.latent_size, 8 * c.latent_size, (hipStream_t)stream));
  });
}

size_t dd_workspace_bytes(dd_engine* e) { return e ? e->total_bytes : 0; }
double dd_flops_last(dd_engine* e) { if (!e) return 0; const double f = e->flops; e->flops = 0; return f; }

}  // extern "C"
