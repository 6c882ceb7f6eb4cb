/* distdiff_hip.h — production C ABI of libdistdiff_hip.so: the MI355X-native replacement of the guided
 * DDIM expansion hot path of haoweiz23/DistDiff (generate_data.py). No torch types cross this boundary:
 * plain pointers, sizes, a HIP stream handle. Every call returns 0 or a negative dd_status and leaves a
 * message in dd_last_error(); no C++ exception crosses the ABI.
 *
 * The reference has no FFI (pure Python over diffusers/timm); each entry point replaces the Python
 * function / third-party call named beside it (file:line in /root/reference):
 *
 *   dd_create / dd_load_tensor / dd_finalize_weights
 *        UNet2DConditionModel.from_pretrained, AutoencoderKL.from_pretrained  generate_data.py:912-922
 *        create_model(...) + checkpoint load                                   model_utils.py:43-104
 *   dd_set_schedule        DDIMScheduler.from_pretrained + retrieve_timesteps   generate_data.py:863, 1043-1044
 *   dd_set_prototypes      total_global_proto / total_local_proto               generate_data.py:1113-1125
 *   dd_set_prompt          prompt_embeds = cat[negative, prompt]                generate_data.py:1147-1148, 1184
 *   dd_add_noise           noise_scheduler.add_noise                            generate_data.py:1176
 *   dd_denoise_step        denoise_one_step                                     generate_data.py:109-121
 *   dd_transform_guidance  transform_guidance (+ linfball_proj)                 generate_data.py:687-732, 124-137
 *   dd_direct_guidance     direct_guidance                                      generate_data.py:735-767
 *   dd_decode              vae.decode + image_processor.postprocess             generate_data.py:1221-1228
 *   dd_expand              the per-(batch, expand index) denoise loop           generate_data.py:1161-1228
 *   dd_guide_encode        image_encoder.encode_image                           model_utils.py:29-41
 *   dd_vae_encode          vae.encode(x).latent_dist.sample() * scaling_factor  dataloader.py:808-809   (stage before the loop, 8f-2)
 *   dd_text_encode         text_encoder(input_ids)[0]                           dataloader.py:633-646   (stage before the loop, 8f-2)
 *
 * Ownership: every tensor argument is caller-owned DEVICE memory unless marked host; the engine borrows it for
 * the duration of the enqueued work. The engine owns packed weights and its activation workspace.
 * Threading: one engine per device, not re-entrant. All work is enqueued on the caller's stream; nothing
 * synchronises unless documented (dd_finalize_weights and dd_set_schedule do, they are one-time setup).
 * Layouts: latents / images are NCHW fp32 exactly as the reference's torch tensors ([B,4,L,L], [B,3,8L,8L]);
 * text embeddings [2B, text_len, cross_dim] fp32 with the negative (unconditional) half first.
 */
#ifndef DISTDIFF_HIP_H
#define DISTDIFF_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct dd_engine dd_engine;

enum dd_status { DD_OK = 0, DD_ERR_ARG = -1, DD_ERR_HIP = -2, DD_ERR_STATE = -3, DD_ERR_WEIGHT = -4, DD_ERR_OOM = -5 };

#define DD_MAX_LEVELS 8

/* Layout version of the structs and argument lists of this header and distdiff_hip_ops.h.  A caller sets dd_config.abi_version =
 * DD_ABI_VERSION (after zero-initialising the struct: every struct of this ABI must be zero-initialised, new fields are appended and
 * mean "off" at 0); dd_create refuses another value with DD_ERR_ARG, and dd_abi_version() tells what the loaded library was built
 * as.  6: dd_config gained unet_attn_fp8 + abi_version, AttnParams gained no_shortk; 5 (unversioned): workspace_bytes in the dd_op_attention_gemm_* lists,
 * ConvGemmParams.wgroup_rows / wgroup_elems, AttnParams.pv_fp8. */
#define DD_ABI_VERSION 6

typedef struct dd_config {
  /* UNet2DConditionModel (unet/config.json) */
  int unet_in_channels, unet_out_channels, unet_levels;
  int unet_block_out_channels[DD_MAX_LEVELS];
  int unet_layers_per_block;
  int unet_down_attn[DD_MAX_LEVELS], unet_up_attn[DD_MAX_LEVELS];
  int unet_num_heads, unet_cross_dim, unet_groups;
  float unet_eps, unet_freq_shift;
  int unet_flip_sin_to_cos;
  /* AutoencoderKL decoder (vae/config.json) */
  int vae_latent_channels, vae_out_channels, vae_levels;
  int vae_block_out_channels[DD_MAX_LEVELS];
  int vae_layers_per_block, vae_groups;
  float vae_eps, vae_scaling_factor;
  /* guide model: timm resnet50 family (model_utils.py:47-55) */
  int guide_stem, guide_stages;
  int guide_planes[DD_MAX_LEVELS], guide_blocks[DD_MAX_LEVELS];
  int guide_expansion, guide_input_size;
  float guide_bn_eps;
  /* problem size */
  int latent_size, text_len, max_batch;
  int enable_grad;      /* 1: build the VJP programs and stash activations (energy guidance) */
  int max_guidance_period; /* P: chained guided steps kept alive for transform_guidance */
  /* CLIPTextModel (text_encoder/config.json); vocabulary, width, depth and MLP size are taken from the state dict */
  int text_heads;       /* num_attention_heads (0: 12) */
  int text_act;         /* hidden_act: 0 quick_gelu (SD-1.x), 1 gelu */
  float text_eps;       /* layer_norm_eps (0: 1e-5) */
  /* guide model family: 0 = timm Bottleneck ResNets (resnet50 / resnext50_32x4d / wide_resnet50_2, exact fp32), 1 = the image tower of
   * an open_clip ViT (open_clip_vit_b32, model_utils.py:80-87; bf16 MFMA): width, depth and MLP size come from the state dict,
   * 2 = timm mobilenetv2_100 (model_utils.py:64-71, exact fp32): guide_stages stages of guide_blocks[s] blocks, the first of stride
   * guide_strides[s]; channel counts, expansion and depthwise groups come from the weight shapes */
  int guide_kind;
  int guide_strides[DD_MAX_LEVELS];
  int guide_vit_heads, guide_vit_patch;
  int guide_vit_act;    /* 0 quick_gelu, 1 gelu (open_clip 'ViT-B-32': gelu) */
  int guide_feature_dim; /* D of encode_image: 2048 (ResNets) / 512 (ViT-B/32 projection) */
  /* SDXL-style UNets (BASELINE.json configs[4]; unet/config.json transformer_layers_per_block, per-level attention_head_dim,
   * addition_embed_type "text_time").  0 = the SD-1.x structure. */
  int unet_transformer_depth[DD_MAX_LEVELS]; /* BasicTransformerBlocks per attention of a level (0: 1) */
  int unet_level_heads[DD_MAX_LEVELS];       /* attention heads of a level (0: unet_num_heads) */
  int unet_add_time_dim, unet_add_text_dim;  /* text_time conditioning: 6 time ids x add_time_dim sinusoids + pooled text embedding */
  /* SDXL's text side (diffusers StableDiffusionXLPipeline.encode_prompt): two CLIP text towers, "text" (CLIPTextModel) and "text2"
   * (CLIPTextModelWithProjection, text_encoder_2/config.json), each read at hidden_states[-2]; prompt embedding = cat[text, text2]
   * along the width (= cross_attention_dim), pooled embedding = text2's text_projection(final_layer_norm(last)[eos]). */
  int text_hidden_layer;  /* 0: last_hidden_state after final_layer_norm (SD-1.x, dataloader.py:633-646); -2: hidden_states[-2] */
  int text2_heads;        /* second tower (loaded under model "text2"): num_attention_heads (0: 20) */
  int text2_act;          /* 0 quick_gelu, 1 gelu (SDXL: gelu) */
  float text2_eps;        /* layer_norm_eps (0: 1e-5) */
  /* BASELINE.json configs[4] "fp8 MFMA attention": 1 = the UNet's d = 64 heads (SDXL) run P.V on the block-scaled fp8 MFMA (e4m3
   * probabilities and values, fp32 accumulation; QK^T, softmax and LSE as in the bf16 form).  Per engine, so both forms can live in one
   * process.  0 = bf16 (the default: faster on MI355X). */
  int unet_attn_fp8;
  int abi_version;        /* must be DD_ABI_VERSION */
} dd_config;

typedef struct dd_sampler_params {
  float guidance_scale;   /* classifier-free guidance scale (--guidance_scale) */
  float gs, ls;           /* global / local prototype energy weights (--gs, --ls) */
  float rho;              /* guidance step size (--rho) */
  float constraint_value; /* L-inf radius (--constraint_value) */
  int use_global, use_local; /* --optimize_targets "global_prototype-local_prototype" */
  int guidance_period;    /* score divisor (args.guidance_period, generate_data.py:719) */
} dd_sampler_params;

typedef struct dd_expand_args {
  const float* image_latents; /* [B,4,L,L] */
  const float* noise;         /* [B,4,L,L] */
  const float* e;             /* [B,4] channel_noise  ~ U[0,1)   (transform guidance) */
  const float* b;             /* [B,4] channel bias   ~ N(0,1) */
  const int* targets;         /* [B] int32 class ids */
  int B;
  int start_index;            /* int((1-strength)*n_steps), generate_data.py:1174 */
  int guidance_type;          /* 0 none, 1 transform_guidance, 2 direct_guidance */
  int guide_first, guide_count; /* guide_timesteps = timesteps[guide_first : guide_first+guide_count] (:1178) */
  float* z_out;               /* [B,4,L,L] final latents */
  float* image_out;           /* [B,3,8L,8L] in [0,1] (may be NULL) */
  float* score_out;           /* [1] device scalar: last guidance score (may be NULL) */
} dd_expand_args;

int dd_abi_version(void);                     /* DD_ABI_VERSION the library was built with */
int dd_create(const dd_config* cfg, dd_engine** out);   /* DD_ERR_ARG when cfg->abi_version != DD_ABI_VERSION */
void dd_destroy(dd_engine* e);
const char* dd_last_error(dd_engine* e);

/* model: "unet" | "vae" | "guide" | "text"; key: Hugging Face / timm state-dict key; data: HOST fp32 */
int dd_load_tensor(dd_engine* e, const char* model, const char* key, const float* data, int ndim, const int64_t* shape);
int dd_finalize_weights(dd_engine* e);
/* Multi-GPU start-up (replaces every process of scripts/exps/expand_diff.sh:19-24 loading its own copy from disk): rank 0 loads and
 * finalizes as above; every other rank DECLARES the same tensors (shapes only, no data), finalizes -- the op graph and every packed
 * weight buffer are built from the shapes alone, in the same order -- and then receives the packed buffers (bf16 MFMA layouts, fp32
 * guide / norm / embedding tables) through its own device staging buffer, e.g. an RCCL broadcast over xGMI.  The packed weights are
 * addressed as ONE virtual byte array of dd_packed_bytes() bytes; export / import copy the range [offset, offset + bytes) between it
 * and a caller-owned DEVICE buffer on the given stream. */
int dd_declare_tensor(dd_engine* e, const char* model, const char* key, int ndim, const int64_t* shape);
size_t dd_packed_bytes(dd_engine* e);
int dd_export_packed(dd_engine* e, void* dst, size_t offset, size_t bytes, void* stream);
int dd_import_packed(dd_engine* e, const void* src, size_t offset, size_t bytes, void* stream);

/* timesteps: host int32[n] (descending, e.g. 981..1); alphas_cumprod: host float[num_train]; */
int dd_set_schedule(dd_engine* e, const int* timesteps, int n, const float* alphas_cumprod, int num_train_timesteps,
                    float final_alpha_cumprod, const dd_sampler_params* sp);
/* Pc [C,D], Pg [C,K,D] HOST fp32, already L2-normalised by the caller as the reference does */
int dd_set_prototypes(dd_engine* e, const float* Pc, const float* Pg, int C, int K, int D);
/* embeds: DEVICE fp32 [2B, text_len, cross_dim], negative half first */
int dd_set_prompt(dd_engine* e, const float* embeds, int B, void* stream);

/* SDXL text_time conditioning (diffusers' added_cond_kwargs): text_embeds DEVICE fp32 [2B, add_text_dim], time_ids DEVICE fp32 [2B, 6]
 * (original size, crop top-left, target size), negative half first like the prompt.  Builds the per-image, per-timestep bias tables
 * emb = time_embedding(t) + add_embedding(cat[text_embeds, sinusoid(time_ids)]) -> time_emb_proj of every ResnetBlock2D.  Needed
 * once per batch of images, after dd_set_schedule, when unet_add_time_dim > 0. */
int dd_set_added_cond(dd_engine* e, const float* text_embeds, const float* time_ids, int B, void* stream);
int dd_add_noise(dd_engine* e, const float* x, const float* noise, float* out, int B, int step_index, void* stream);
int dd_denoise_step(dd_engine* e, const float* z, int step_index, float* z_prev_out, float* x0_out, int B, void* stream);
int dd_transform_guidance(dd_engine* e, const float* z, const int* targets, const float* ch_e, const float* ch_b,
                          int first_step_index, int P, float* z_out, float* score_out, float* grad_eb_out, int B,
                          void* stream);
int dd_direct_guidance(dd_engine* e, const float* z, const int* targets, int step_index, float* z_next_out, float* x0_out,
                       float* score_out, float* grad_z_out, int B, void* stream);
int dd_decode(dd_engine* e, const float* z, float* image_out, int denormalize, int B, void* stream);
int dd_expand(dd_engine* e, const dd_expand_args* a, void* stream);
/* The reference's energy is a `.mean()` over ITS batch (train_batch_size images: generate_data.py:709, :716, :751, :758), so each
 * image's gradient carries 1/train_batch_size.  A caller that packs several reference batches into one engine batch passes
 * w[i] = 1 / |reference batch of image i| (0 for padding rows): HOST float[B], or NULL to return to the default 1/B.  Applies to
 * every later guidance call; score_out then holds sum_i w[i] E_i. */
int dd_set_sample_weights(dd_engine* e, const float* w_host, int B);
/* per-image energies E_i of the most recent guidance call (for transform guidance already divided by guidance_period, :719):
 * DEVICE float[B]; the caller averages them over its reference batches for the score log line (:1208-1216) */
int dd_get_image_scores(dd_engine* e, float* scores_out, int B, void* stream);
/* output stage: image DEVICE fp32 [B,3,8L,8L] in [0,1] -> DEVICE uint8 [B,8L,8L,3] (mul 255, add 0.5, clamp, truncate: save_image, :1227-1234) */
int dd_image_to_u8(dd_engine* e, const float* image, uint8_t* out_hwc, int B, void* stream);
/* images: DEVICE fp32 [B,3,S,S] (S = guide_input_size) -> feats DEVICE fp32 [B, D] */
int dd_guide_encode(dd_engine* e, const float* images, float* feats, int B, void* stream);
/* encode_image(x, pooling='max') of model_utils.py:34-35 (AdaptiveMaxPool2d instead of the default average pool; forward only: the
 * expansion path always uses 'avg', generate_data.py:705, :746) */
int dd_guide_encode_pooled(dd_engine* e, const float* images, float* feats, int B, int use_max, void* stream);
/* The stage before the loop; available when the state dicts carried vae/encoder.* + quant_conv.* and
 * text/text_model.* keys.  images: DEVICE fp32 [B,3,8L,8L] in [-1,1]; noise: DEVICE fp32 [B,4,L,L] ~ N(0,1) or NULL (the
 * distribution's mode); latents_out [B,4,L,L] already multiplied by scaling_factor; moments_out optional [B,8,L,L]
 * (mean | logvar clamped to [-30,20]). */
int dd_vae_encode(dd_engine* e, const float* images, const float* noise, float* latents_out, float* moments_out, int B, void* stream);
/* input_ids: DEVICE int32 [n, text_len] (tokenizer output, padded to text_len); embeds_out DEVICE fp32 [n, text_len, cross_dim]
 * = last_hidden_state after final_layer_norm; 1 <= n <= 2*max_batch */
int dd_text_encode(dd_engine* e, const int* input_ids, float* embeds_out, int n, void* stream);
/* One tower of a two-tower model (which = 0: "text", 1: "text2"): hidden_out DEVICE fp32 [n, text_len, width of that tower] at
 * dd_config.text_hidden_layer; pooled_out (which = 1 only, may be NULL) DEVICE fp32 [n, projection_dim] = text_embeds of transformers'
 * CLIPTextModelWithProjection: text_projection(final_layer_norm(last layer)[position of the largest token id = eos]). */
int dd_text_encode_tower(dd_engine* e, int which, const int* input_ids, float* hidden_out, float* pooled_out, int n, void* stream);
/* diagnostic: raw UNet forward, eps2_out DEVICE fp32 [2B,4,L,L] (uncond half first) */
int dd_unet_forward(dd_engine* e, const float* z, int step_index, float* eps2_out, int B, void* stream);

/* per-module VJP diagnostics (parity of the hand-derived reverse programs vs torch.autograd.grad, generate_data.py:721/761):
 *   g_z = J_unet(z)^T g_eps2 ; g_z = J_decode(z)^T g_image ; g_images = J_guide(images)^T g_feats */
int dd_unet_vjp(dd_engine* e, const float* z, int step_index, const float* g_eps2, float* g_z_out, int B, void* stream);
int dd_decode_vjp(dd_engine* e, const float* z, const float* g_image, float* g_z_out, int B, void* stream);
int dd_guide_vjp(dd_engine* e, const float* images, const float* g_feats, float* g_images_out, int B, void* stream);

/* live HIP-event timing of every enqueued op on the caller's stream, by kernel family
 * (0 conv_gemm, 1 attention, 2 norm, 3 other): out12[fam*3 + {0,1,2}] = {ms, algorithmic FLOPs, ops}. dd_profile_read synchronises. */
int dd_profile_enable(dd_engine* e, int on);
int dd_profile_read(dd_engine* e, double* out12);

/* test-only introspection hooks (dd_debug_*) live in include/distdiff_hip_ops.h */

size_t dd_workspace_bytes(dd_engine* e);
/* algorithmic MFMA-eligible FLOPs (conv/linear/attention, 2 per MAC) enqueued since the last call */
double dd_flops_last(dd_engine* e);

#ifdef __cplusplus
}
#endif
#endif
