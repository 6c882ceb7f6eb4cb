/* distdiff_hip_ops.h — op-level C ABI of libdistdiff_hip.so (diagnostic / unit-test surface).
 *
 * Each entry point enqueues ONE hand-written gfx950 kernel family on the caller's HIP stream with
 * plain device pointers; the parity tests in tests/ call these against the CPU oracle. The
 * production boundary (the reference's sampler functions, generate_data.py:109-121, 687-767) is
 * include/distdiff_hip.h. All functions return 0 on success or a hipError_t value.
 *
 * Reference operation each one replaces (file:line in haoweiz23/DistDiff):
 *   dd_op_conv_gemm      diffusers Conv2d / Linear / timm conv+BN inside unet(), vae.decode(), encode_image()
 *                        (generate_data.py:112, :701, :705) and their input-gradients (:721, :761)
 *   dd_op_groupnorm_*    GroupNorm(+SiLU) in ResnetBlock2D / Transformer2D / VAE decoder (:112, :701)
 *   dd_op_layernorm_*    LayerNorm in BasicTransformerBlock (:112)
 *   dd_op_attention_*    scaled-dot-product attention (self, cross, VAE mid block) (:112, :701)
 *   dd_op_cfg_ddim*      classifier-free guidance + DDIMScheduler.step (:116-119)
 *   dd_op_bicubic*       F.interpolate(..., (224,224), 'bicubic') (:704, :745)
 *   dd_op_conv_f32       timm conv+BN(+ReLU) of image_encoder.encode_image and its input-gradient in exact fp32
 *                        (model_utils.py:29-41; generate_data.py:705, :721, :746, :761): v_mfma_f32_32x32x2_f32
 *   dd_op_energy         prototype energy terms (:707-717, :747-759)
 *   dd_op_transform_update  SGD step on (e, b), re-affine, linfball_proj (:721-728, :124-137)
 */
#ifndef DISTDIFF_HIP_OPS_H
#define DISTDIFF_HIP_OPS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* parameter blocks are the POD structs of distdiff_amd/csrc/kernels.h, passed by pointer */
struct ConvGemmParams;
struct GroupNormParams;
struct LayerNormParams;
struct AttnParams;
struct ConvF32Params;

/* taptab contract: entry t = ((dy + 32) << 6) | (dx + 32).  A launch with ONE tap, stride 1 and an output of the input's size is a
 * pointwise (1x1 / linear) layer and must carry the centre tap (dy = dx = 0) -- the persistent kernels skip the table for it.  The
 * entry point is stream-asynchronous and does not read device memory: the owner of the table checks it on the host when the weights
 * are packed (distdiff_amd/ops.py raises on any other single tap). */
int dd_op_conv_gemm(const struct ConvGemmParams* p, size_t partial_cap_bytes, void* stream);
/* Synchronous check of that contract against the DEVICE table (waits for the stream, copies the table back): 0 when every entry is in
 * range and a pointwise launch carries the centre tap.  For ABI users without a host copy of their tables; never called by the engine. */
int dd_op_conv_gemm_check(const struct ConvGemmParams* p, void* stream);
int dd_op_groupnorm_fwd(const struct GroupNormParams* p, void* stream);
int dd_op_groupnorm_bwd(const struct GroupNormParams* p, void* stream);
size_t dd_op_groupnorm_scratch_bytes(int B, int G);
int dd_op_layernorm_fwd(const struct LayerNormParams* p, void* stream);
int dd_op_layernorm_bwd(const struct LayerNormParams* p, void* stream);
int dd_op_attention_fwd(const struct AttnParams* p, void* stream);
int dd_op_attention_bwd(const struct AttnParams* p, void* stream);
/* wide heads (d >= 256, the AutoencoderKL mid-block attention): the same attention through GEMMs on a materialised N x N score
 * matrix per image.  workspace / workspace_bytes: device scratch of at least dd_op_attention_gemm_workspace() bytes (ONE image); given
 * k times that, single-head layers run up to min(k, 8) images per launch (grouped GEMMs); tap1x1: device int = (32 << 6) | 32;
 * partial / partial_cap: split-K scratch as for dd_op_conv_gemm. */
size_t dd_op_attention_gemm_workspace(int Nq, int Nk, int D, int bwd);
int dd_op_attention_gemm_fwd(const struct AttnParams* p, void* workspace, size_t workspace_bytes, const int* tap1x1, float* partial, size_t partial_cap, void* stream);
int dd_op_attention_gemm_bwd(const struct AttnParams* p, void* workspace, size_t workspace_bytes, const int* tap1x1, float* partial, size_t partial_cap, void* stream);

/* fp32 implicit-GEMM convolution / dgrad of the guide network (guide_f32.hip) and its weight packing: w is
 * [Cout][Cin/groups][KH][KW] fp32 (torch grouped layout); out4 = {N, K, cin, ntaps}; wp may be NULL to query sizes */
int dd_op_conv_f32(const struct ConvF32Params* p, void* stream);
int dd_pack_conv_weight_f32(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int groups, float* wp, int* taptab,
                            int* out4);

/* host-side weight packing: returns N, K, cin, ntaps through out[4]; wp may be NULL to query sizes */
int dd_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int KH, int KW, int pad, int mode, int geglu,
                        uint16_t* wp, int* taptab, int* out4);

int dd_op_nchw_f32_to_nhwc_bf16(const float* src, uint16_t* dst, int B, int C, int H, int W, int Cpad, int ld, int dup,
                                float scale, void* stream);
int dd_op_nhwc_to_nchw_f32(const void* src, int src_f32, float* dst, int B, int C, int H, int W, int ld, float scale,
                           float shift, int clamp, float lo, float hi, void* stream);
int dd_op_cfg_ddim(const float* eps2, int ld, const float* z, float* z_prev, float* x0, int B, int C, int HW,
                   const float* coef_dev, void* stream);
int dd_op_cfg_ddim_bwd(const float* g_x0, const float* g_zprev, uint16_t* g_eps2, int ld, float* g_z, int B, int C, int HW,
                       const float* coef_dev, void* stream);
int dd_op_sumpool2x2(const uint16_t* src, int src_ld, uint16_t* dst, int dst_ld, int B, int H, int W, int C, int accumulate,
                     void* stream);
int dd_op_geglu_bwd(const uint16_t* raw, int ld_raw, const uint16_t* dout, int ld_dout, uint16_t* draw, int ld_draw, int M,
                    int F, void* stream);
int dd_op_maxpool3x3s2(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, void* stream);
int dd_op_maxpool3x3s2_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int B, int H, int W, int C, void* stream);
int dd_op_bicubic(const uint16_t* src, int ld_s, uint16_t* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C,
                  int Cpad, void* stream);
int dd_op_bicubic_bwd(const uint16_t* ddst, int ld_d, uint16_t* dsrc, int ld_s, int B, int Hs, int Ws, int Hd, int Wd, int C,
                      void* stream);
int dd_op_gap(const uint16_t* x, int ld, float* f, int B, int HW, int C, void* stream);
int dd_op_energy(const float* f, const float* Pc, const float* Pg, const int* targets, int B, int D, int K, float gs, float ls,
                 int use_c, int use_g, int normalize, float weight, float* score_out, float* gf, void* stream);
int dd_op_transform_update(const float* z, const float* g, const float* e, const float* b, float* z_out, int BC, int HW,
                           float rho, float c, void* stream);
int dd_op_affine(const float* z, const float* e, const float* b, float* out, int BC, int HW, void* stream);

/* ---- test-only hooks into a dd_engine (include/distdiff_hip.h); no production caller ----
 * debug introspection of the op graph: host copy of an activation or gradient of tensor idx (negative: from the end, -1 = the
 * program's output) of program (prog & 15) = 0 unet / 1 vae / 2 guide, chained-step instance (prog >> 4) */
struct dd_engine;
int dd_debug_tensor(struct dd_engine* e, int prog, int idx, int want_grad, float* host_out, int* info4);
int dd_debug_num_tensors(struct dd_engine* e, int prog);
/* parity-test hook: evaluate the guide network of every later guided forward AT these images (DEVICE fp32 [count][B,3,8L,8L], the
 * decoder's output range, caller-owned; chained guided step k reads image min(k, count-1)) instead of the decoder's own output;
 * gradients still flow through the decoder.  The input-gradient of the ReLU / max-pool guide is piecewise constant in the image, so
 * two implementations of torch.autograd.grad(E, ...) (generate_data.py:721, :761) can only be compared at the same image.
 * NULL / count 0 switches it off.  dd_debug_set_image = count 1. */
int dd_debug_set_images(struct dd_engine* e, const float* images, int count);
int dd_debug_set_image(struct dd_engine* e, const float* image);

#ifdef __cplusplus
}
#endif
#endif
