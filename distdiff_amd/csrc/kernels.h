// Host-side launch interface of the hand-written gfx950 kernels.
// Everything here enqueues on the given stream and never synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;

// ----------------------------------------------------------------------------------------------
// K1/K2/K8: implicit-GEMM convolution == GEMM == linear (bf16 MFMA, fp32 accumulate).
//   out[m, n] = epilogue( alpha * sum_k A[m, k] * Wp[n, k] ),  m = (b, oy, ox) NHWC pixel,
//   k = (tap, cin) with cin fastest, A gathered on the fly from the NHWC input (zero padding,
//   optional fused nearest-2x upsample, optional stride-2, optional input-dilated "transposed"
//   gather used for the dgrad of stride-2 convolutions).
// ----------------------------------------------------------------------------------------------
enum ConvFlags {
  CF_BIAS = 1,       // + bias[n]          (fp32; optionally selected per timestep through bias_sel)
  CF_RES = 2,        // + res[m, n]        (bf16)
  CF_RELU = 4,       // max(., 0)
  CF_GEGLU = 8,      // packed (hidden|gate) 16-column groups -> hidden * gelu(gate); N counts packed columns
  CF_OUT_F32 = 16,   // store fp32 instead of bf16
  CF_MASK = 32,      // multiply by (mask[m, n] > 0)   (ReLU backward)
  CF_RES_F32 = 64,   // residual is fp32
  CF_GEGLU_RAW = 128,// with CF_GEGLU: also store the raw packed pre-activation (bf16) to raw[m, N]
  CF_RELU6 = 512,    // min(max(., 0), 6)   (fp32 guide kernels: MobileNetV2)
  CF_STATS = 256,    // also emit per-(64-row block, output channel) partial statistics (mean, M2) of the stored values: the
                     // GroupNorm that consumes this tensor then needs no statistics pass of its own (conv_gemm_can_emit_stats)
  CF_LNFOLD = 1024,  // the A operand is the RAW input x of a LayerNorm whose gamma is folded into the packed weights (W' = gamma o W) and
                     // whose beta into the bias (b' = b + W beta): out = rstd[m] * (acc - mean[m] * c1[n]) + b'[n] with
                     // c1[n] = sum_k W'[n, k] (of the bf16-rounded W'), (mean, rstd) = ln_stats[m] -- LayerNorm(x) W^T + b without ever
                     // materialising LayerNorm(x).  Pointwise (1x1 / linear) layers only.
  CF_GNFOLD = 4096,  // the input is the RAW input x of a GroupNorm(+SiLU): the halo-resident 3x3 kernel applies silu(x * a[b, c] + b[b, c]) to its
                     // staged input tile (once per tile and 64-channel chunk, zero padding kept) instead of a separate apply pass; gn_coef =
                     // per-(image, channel) (a, b) = (rstd * gamma, beta - mean * rstd * gamma).  conv_gemm_can_fold_gn() says when.
  CF_ROWSTATS = 2048 // also emit, per output row and per 80-/64-column wave span, (sum v, sum v^2) of the fp32 values in front of their bf16 rounding
                     // (one-pass fp32: rstd of the stored rows within 4 % up to |mean| = 50 std, tests/test_kernels_gpu.py) into rowpart: the
                     // LayerNorm that consumes this tensor takes its row statistics from there (conv_gemm_can_emit_rowstats)
};

struct ConvGemmParams {
  const bf16_t* x;      // input activations, NHWC, row stride x_ld elements
  const bf16_t* w;      // packed weights [N][K], K contiguous, K % 64 == 0 (zero padded)
  const int* taptab;    // ntaps entries: ((dy + 32) << 6) | (dx + 32), logical-input offset of each filter tap
  void* y;              // output (bf16 or fp32), row stride y_ld
  const float* bias;    // [N] fp32 or null
  const int* bias_sel;  // optional device scalar: bias += (*bias_sel) * bias_stride
  const void* res;      // residual or null
  const bf16_t* mask;   // relu mask source or null
  bf16_t* raw;          // GEGLU raw output or null
  float* partial;       // split-K workspace [ksplit][M][N] fp32 (ksplit > 1)
  float* stats;         // CF_STATS: [M / 64][stats_ld][2] fp32 (mean, M2 of 64 rows), already offset to this op's first channel
  const float* ln_stats;// CF_LNFOLD: [M][2] fp32 (mean, rstd) of the LayerNorm's input rows
  const float* ln_c1;   // CF_LNFOLD: [N] fp32 column sums of the folded weights (packed order for GEGLU)
  const float* gn_coef; // CF_GNFOLD: [B][cin][2] fp32 (a, b) per image and input channel
  float* rowpart;       // CF_ROWSTATS: [M][rowpart_ld][2] fp32 (sum, sum of squares) per row and column span of conv_gemm_rowstat_span(N)
  int x_ld, y_ld, res_ld, mask_ld, raw_ld, bias_stride, stats_ld, rowpart_ld, gn_silu;
  int B, H, W;          // stored input geometry
  int Ho, Wo;           // output geometry
  int stride;           // output->logical-input stride (1 or 2)
  int shift;            // 1: logical input is 2x the stored input (nearest upsample or dilation)
  int parity;           // 1 with shift: only even logical coordinates are populated (transposed conv)
  int cin, ntaps;       // channels per tap (multiple of 8), number of taps; K = roundup(ntaps*cin, 64)
  int M, N, K;
  int ksplit;
  int flags;
  float alpha;
  int force_small;      // diagnostics: 1 forces the 128x128-tile kernel
  int wgroup_rows;      // > 0: GROUPED weights -- output rows [g * wgroup_rows, (g + 1) * wgroup_rows) use the weight matrix at
  long long wgroup_elems; //   w + g * wgroup_elems (same [N][K] shape each): a batch of independent GEMMs stacked along M in one launch
                        //   (attention_gemm.hip: S_b = Q_b K_b^T for all images b).  Persistent big-tile kernel only; 0 otherwise.
};

// Chooses tile configuration / split-K from the shape. `partial_cap_bytes` bounds split-K workspace.
hipError_t launch_conv_gemm(ConvGemmParams p, size_t partial_cap_bytes, hipStream_t stream);
// Preferred split for a shape (used by the engine to size the workspace).
int conv_gemm_pick_split(int M, int N, int K);
// true iff launch_conv_gemm(p, partial_cap_bytes) will honour CF_STATS for this problem (persistent big-tile kernel, no split-K,
// M % 64 == 0, plain bf16 output): the caller asks before setting the flag and falls back to the GroupNorm statistics pass otherwise
bool conv_gemm_can_emit_stats(ConvGemmParams p, size_t partial_cap_bytes);
// CF_ROWSTATS: true iff launch_conv_gemm will honour it for this problem; *spans = number of (sum, sum^2) pairs per row it writes
// (N / columns per wave of the tile the launcher picks); rowpart_ld must be >= *spans
bool conv_gemm_can_emit_rowstats(ConvGemmParams p, size_t partial_cap_bytes, int* spans);
// CF_GNFOLD: true iff launch_conv_gemm will send this problem to the halo-resident 3x3 kernel (the only form that can apply a GroupNorm
// to its input tile)
bool conv_gemm_can_fold_gn(ConvGemmParams p);

// ----------------------------------------------------------------------------------------------
// K3: GroupNorm (+ optional SiLU) forward / backward on NHWC bf16.
// ----------------------------------------------------------------------------------------------
struct GroupNormParams {
  const bf16_t* x; int x_ld;
  bf16_t* y; int y_ld;
  const float* gamma; const float* beta;   // [C]
  float* stats;        // [B][G][2] (mean, rstd), written by fwd, read by bwd
  float* scratch;      // [B][S][G][3] partial (count, mean, M2) / bwd partial sums
  float* coef;             // fwd, optional: instead of applying, write the per-(image, channel) affine (a, b) [B][C][2] for a consumer
                           //   that applies it itself (CF_GNFOLD); y is then unused
  const float* chan_part;  // fwd, optional: per-(64-row block, channel) partials (mean, M2) emitted by the producing convolutions
  int part_ld;             //   (CF_STATS), [B*HW/64][part_ld][2], already offset to channel 0 of x: replaces the statistics pass
  int B, HW, C, G;
  float eps;
  int silu;
  // backward only
  const bf16_t* dy; int dy_ld;
  bf16_t* dx; int dx_ld;
  int accumulate;      // dx += result
};
size_t groupnorm_scratch_bytes(int B, int G);
hipError_t launch_groupnorm_fwd(const GroupNormParams& p, hipStream_t stream);
hipError_t launch_groupnorm_bwd(const GroupNormParams& p, hipStream_t stream);

// ----------------------------------------------------------------------------------------------
// K4: LayerNorm over the channel dim of [M, C] bf16 rows.
// ----------------------------------------------------------------------------------------------
struct LayerNormParams {
  const bf16_t* x; int x_ld;
  bf16_t* y; int y_ld;
  const float* gamma; const float* beta;
  float* stats;        // [M][2] mean, rstd
  int M, C; float eps;
  const float* rowpart; int rowpart_ld, spans;   // fwd, y == null: statistics only -- from the producing GEMM's row partials when given
                                                 // ([M][rowpart_ld][2] (sum, sum^2) over `spans` column spans), else from x
  const bf16_t* dy; int dy_ld;
  bf16_t* dx; int dx_ld;
  int accumulate;
};
hipError_t launch_layernorm_fwd(const LayerNormParams& p, hipStream_t stream);
hipError_t launch_layernorm_bwd(const LayerNormParams& p, hipStream_t stream);

// ----------------------------------------------------------------------------------------------
// K5: flash attention forward / backward. Q rows [B*Nq, ldq], head h at column h*D.
// ----------------------------------------------------------------------------------------------
struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  bf16_t* o; float* lse;                    // lse [B][H][Nq] (natural log domain, of scaled scores)
  int ldq, ldk, ldv, ldo;
  int B, H, Nq, Nk, D;
  float scale;
  // backward
  const bf16_t* d_o; int lddo;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;       // dk/dv may be null (cross attention: dQ only)
  int lddq, lddk, lddv;
  float* delta;                             // [B][H][Nq] scratch: rowsum(dO*O)
  int q_prescaled;                          // Q already carries 1/sqrt(D) * log2(e) (folded into the to_q weights): scores are log2-domain; pass scale = ln 2
  int causal;                               // forward only: key j visible to query i iff j <= i (CLIP text encoder)
  int pv_fp8;                               // forward only, D = 64: P.V on the block-scaled fp8 MFMA (e4m3 probabilities and values)
  int no_shortk;                            // diagnostics: keep <= 80-key launches on the streaming forward (accuracy A/B inside one process)
};
hipError_t launch_attention_fwd(const AttnParams& p, hipStream_t stream);
hipError_t launch_attention_bwd(const AttnParams& p, hipStream_t stream);
// short key sequences (<= 80 keys: the 77-token cross-attention), attention_shortk.hip; launch_attention_fwd takes it when it applies
bool attention_shortk_supported(const AttnParams& p);
hipError_t launch_attention_fwd_shortk(const AttnParams& p, hipStream_t stream);
hipError_t launch_attention_delta(const AttnParams& p, hipStream_t stream);   // delta[b,h,q] = sum_d dO*O (first stage of the backward)
// Wide heads (d >= 256: the AutoencoderKL mid-block attention) through the GEMM kernel with a materialised N x N score matrix per image
// (attention_gemm.hip).  workspace: at least attention_gemm_workspace() bytes (one image); with more, single-head layers run up to 8
// images per launch (grouped GEMMs); tap1x1: device int holding the 1x1 tap ((32 << 6) | 32);
// partial: split-K scratch of launch_conv_gemm.  Same arguments / results as the flash launchers.
bool attention_gemm_supported(const AttnParams& p);
size_t attention_gemm_workspace(int Nq, int Nk, int D, int bwd);
hipError_t launch_attention_gemm_fwd(const AttnParams& p, void* workspace, size_t workspace_bytes, const int* tap1x1, float* partial, size_t partial_cap, hipStream_t stream);
hipError_t launch_attention_gemm_bwd(const AttnParams& p, void* workspace, size_t workspace_bytes, const int* tap1x1, float* partial, size_t partial_cap, hipStream_t stream);

// ----------------------------------------------------------------------------------------------
// K6/K7/K9/K10/K12: small HBM-bound kernels.
// ----------------------------------------------------------------------------------------------
// NCHW fp32 [B,C,H,W] -> NHWC bf16 [B*H*W, ld] (channels >= C zero-filled up to Cpad); dup copies
// the batch twice (CFG: cat[z, z]).  scale multiplies.
hipError_t launch_nchw_f32_to_nhwc_bf16(const float* src, bf16_t* dst, int B, int C, int H, int W, int Cpad, int ld,
                                        int dup, float scale, hipStream_t s);
// NHWC (bf16 or fp32) [B*H*W, ld] -> NCHW fp32 [B,C,H,W];  out = in * scale + shift, optional clamp to [lo, hi]
hipError_t launch_nhwc_to_nchw_f32(const void* src, int src_f32, float* dst, int B, int C, int H, int W, int ld,
                                   float scale, float shift, int clamp, float lo, float hi, hipStream_t s);
// CFG + DDIM step (generate_data.py:116-119): eps2 NHWC fp32 [2B*HW, ld] (uncond first), z NCHW fp32
//   coef = {guidance_scale, sqrt_a_t, sqrt_1m_a_t, sqrt_a_prev, sqrt_1m_a_prev}
hipError_t launch_cfg_ddim(const float* eps2, int ld, const float* z, float* z_prev, float* x0, int B, int C, int HW,
                           const float* coef_dev, hipStream_t s);
// backward of cfg_ddim wrt (z via direct path) and eps2: see DESIGN.md / SURVEY appendix A
//   g_x0, g_zprev NCHW fp32 (either may be null) -> g_eps2 NHWC bf16 [2B*HW, ld] and g_z_direct NCHW fp32
hipError_t launch_cfg_ddim_bwd(const float* g_x0, const float* g_zprev, bf16_t* g_eps2, int ld, float* g_z, int B, int C,
                               int HW, const float* coef_dev, hipStream_t s);
// backward of cat[z, z] + NCHW->NHWC: g_z[b,c,pix] (+)= gin[b*HW+pix, c] + gin[(B+b)*HW+pix, c]   (halves = 2), or of the plain
// layout change when the UNet input itself is not duplicated (halves = 1: the two CFG halves share their prefix, engine.cpp)
hipError_t launch_dup_bwd(const bf16_t* gin, int ld, float* g_z, int B, int C, int HW, int accumulate, int halves, hipStream_t s);
// y = dy * (mask > 0)  (ReLU backward)
hipError_t launch_mask_bf16(const bf16_t* dy, int ldd, const bf16_t* mask, int ldm, bf16_t* y, int ldy, int M, int C,
                            hipStream_t s);
// add_noise: out = sa * x + sb * n   (fp32, n elements), coefficients from device memory {sa, sb}
hipError_t launch_axpby(const float* x, const float* n, float* out, size_t count, const float* coef_dev, hipStream_t s);
// 2x2 sum pooling of NHWC bf16 (backward of the fused nearest-2x upsample): [B,2H,2W,C] -> [B,H,W,C]
hipError_t launch_sumpool2x2(const bf16_t* src, int src_ld, bf16_t* dst, int dst_ld, int B, int H, int W, int C,
                             int accumulate, hipStream_t s);
// y = a + b (bf16 rows), used where a fan-in cannot be fused into an epilogue
hipError_t launch_add_bf16(const bf16_t* a, int lda, const bf16_t* b, int ldb, bf16_t* y, int ldy, int M, int C,
                           hipStream_t s);
// copy rows (bf16) with strides
hipError_t launch_copy_bf16(const bf16_t* a, int lda, bf16_t* y, int ldy, int M, int C, hipStream_t s);
// GEGLU backward: raw packed [M, 2F] (16-col groups hidden|gate), dout [M, F] -> draw [M, 2F] packed
hipError_t launch_geglu_bwd(const bf16_t* raw, int ld_raw, const bf16_t* dout, int ld_dout, bf16_t* draw, int ld_draw,
                            int M, int F, hipStream_t s);
// max pool 3x3 stride 2 pad 1 NHWC bf16, and its backward (recomputes the argmax: first max in scan order)
hipError_t launch_maxpool3x3s2(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_maxpool3x3s2_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C,
                                   hipStream_t s);
// bicubic (A=-0.75, align_corners=False, no antialias) NHWC: src [B,Hs,Ws,ld_s] (C real channels) -> dst [B,Hd,Wd,ld_d]
// channels >= C in dst are zero-filled up to Cpad.
hipError_t launch_bicubic(const bf16_t* src, int ld_s, bf16_t* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C,
                          int Cpad, hipStream_t s);
// transpose of the above: dsrc [B,Hs,Ws,C] (+)= sum of taps of ddst
hipError_t launch_bicubic_bwd(const bf16_t* ddst, int ld_d, bf16_t* dsrc, int ld_s, int B, int Hs, int Ws, int Hd, int Wd,
                              int C, hipStream_t s);
// global average pool [B, HW, C] bf16 -> f [B, C] fp32 ; backward broadcasts g/HW as bf16
hipError_t launch_gap(const bf16_t* x, int ld, float* f, int B, int HW, int C, hipStream_t s);
hipError_t launch_gap_bwd(const float* gf, bf16_t* dx, int ld, int B, int HW, int C, const bf16_t* mask, int mask_ld,
                          hipStream_t s);
// energy (generate_data.py:707-717 / 747-759) and its gradient wrt f.
//   f [B,D] fp32, Pc [Ccls,D], Pg [Ccls,K,D], targets [B] int32. normalize: direct-guidance L2-normalise first.
//   E_i = gs*||f_i-pc|| + ls*||f_i-pg*||;  score_out[0] += weight * sum_i w_i E_i;  gf [i,:] = weight * w_i * dE_i/df_i
//   w_i = sample_w[i] (device [B]) or 1/B when null: the reference's `.mean()` runs over its train_batch_size group
//   (:709-719, :750-760), which the caller expresses as w_i = 1/|group of i| when it packs several groups into one engine batch.
//   image_scores (device [B], may be null): image_scores[i] += weight * E_i.
hipError_t launch_energy(const float* f, const float* Pc, const float* Pg, const int* targets, int B, int D, int K,
                         float gs, float ls, int use_c, int use_g, int normalize, float weight, const float* sample_w,
                         float* score_out, float* image_scores, float* gf, hipStream_t s);
// transform guidance update (generate_data.py:696, 721-728): per (b,c): ge = sum_hw g*z, gb = sum_hw g;
//   e -= rho*ge, b -= rho*gb ; z_out = clamp(z*(1+e)+b, z-c, z+c) (lower bound first)
hipError_t launch_affine(const float* z, const float* e, const float* b, float* out, int BC, int HW, hipStream_t s);
hipError_t launch_transform_update(const float* z, const float* g, const float* e, const float* b, float* z_out, int BC,
                                   int HW, float rho, float c, hipStream_t s);
// out = a - rho * g  (direct guidance :762)
hipError_t launch_sub_scaled(const float* a, const float* g, float* out, size_t n, float rho, hipStream_t s);
// NHWC bf16 [M, ld] (first C channels) -> NCHW fp32, used for gradients wrt latents
// fp32 -> bf16 pack rows (weights upload helpers)
hipError_t launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s);
// setup-time fp32 linear: y = act(x) W^T + b (time-embedding tables)
hipError_t launch_linear_f32(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int silu_in, hipStream_t s);
hipError_t launch_fill_f32(float* dst, float v, size_t n, hipStream_t s);
// SDXL added conditioning (setup-time, fp32): sinusoidal embedding of n scalars (diffusers Timesteps: flip_sin_to_cos, shift 0) into
// out[i*ld + off .. + dim); out[(s*nb + b), :] = a[s, :] + v[b, :]
hipError_t launch_sinusoid_f32(const float* x, float* out, int n, int dim, int ld, int off, int flip, hipStream_t s);
hipError_t launch_add_outer_f32(const float* a, const float* v, float* out, int ns, int nb, int cols, hipStream_t s);
// y[r, c] += v[c]  (fp32 rows; setup-time: the per-timestep bias tables)
hipError_t launch_add_rowvec_f32(float* y, const float* v, int rows, int cols, hipStream_t s);
// (x/2+0.5).clamp(0,1)*255+0.5 -> uint8 HWC (output stage, generate_data.py:1227 + save_image quantisation)
hipError_t launch_to_uint8(const float* nchw, uint8_t* hwc, int B, int C, int H, int W, hipStream_t s);

// f-2, the stage before the loop (dataloader.py:633-661, 750-811): CLIP token + position embedding gather, CLIP MLP activation
// (kind 0 quick_gelu, 1 erf-GELU), DiagonalGaussian sample * scaling_factor from fp32 NHWC moments, bf16 rows -> fp32 rows
hipError_t launch_clip_embed(const int* ids, const float* tok, const float* pos, bf16_t* out, int ld, int rows, int T, int C, int vocab,
                            hipStream_t s);
hipError_t launch_act_bf16(const bf16_t* x, int ldx, bf16_t* y, int ldy, int M, int C, int kind, hipStream_t s);
// the open_clip ViT guide (model_utils.py:80-87): GELU / quick_gelu backward, non-overlapping patch embedding as a GEMM (image fp32
// NHWC -> rows [B*(S/p)^2, C*p*p], k = (c, iy, ix)) and its transpose, class token + positional embedding, class-token selection
hipError_t launch_act_bwd_bf16(const bf16_t* x, int ldx, const bf16_t* dy, int ldd, bf16_t* dx, int ldo, int M, int C, int kind, int accumulate,
                               hipStream_t s);
hipError_t launch_patchify(const float* img, int ld, bf16_t* out, int B, int S, int p, int C, hipStream_t s);
hipError_t launch_patchify_bwd(const bf16_t* gout, float* gimg, int ld, int B, int S, int p, int C, hipStream_t s);
hipError_t launch_vit_embed(const bf16_t* patches, int ldp, const float* cls, const float* pos, bf16_t* out, int ldo, int B, int np, int W,
                            hipStream_t s);
hipError_t launch_vit_embed_bwd(const bf16_t* gout, int ldo, bf16_t* gp, int ldp, int B, int np, int W, hipStream_t s);
hipError_t launch_select_rows(const bf16_t* x, int ldx, bf16_t* y, int ldy, int B, int stride, int C, hipStream_t s);
hipError_t launch_select_rows_bwd(const bf16_t* dy, int ldy, bf16_t* dx, int ldx, int B, int stride, int C, int accumulate, hipStream_t s);
hipError_t launch_vae_sample(const float* moments, int ld, const float* noise, float* latents, float* moments_out, int B, int C, int HW,
                             float scale, hipStream_t s);
hipError_t launch_rows_bf16_to_f32(const bf16_t* x, int ld, float* y, int M, int C, hipStream_t s);
hipError_t launch_clip_pool_project(const int* ids, const bf16_t* x, int ld, const float* W, float* out, int n, int T, int C, int Pd, hipStream_t s);

// ----------------------------------------------------------------------------------------------
// The guide network in exact fp32 (guide_f32.hip): implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 and the fp32
// forms of the guide's side kernels.  Same tap-table / stride / dilated-gather conventions as ConvGemmParams.
// ----------------------------------------------------------------------------------------------
struct ConvF32Params {
  const float* x;       // input activations, NHWC fp32, row stride x_ld (multiple of 4)
  const float* w;       // packed weights [N][K] fp32, k = (tap, cin), K % 16 == 0 (zero padded)
  const int* taptab;
  float* y;             // output fp32, row stride y_ld
  const float* bias;    // [N] or null
  const float* res;     // residual (CF_RES); may alias y (accumulating dgrad)
  const float* mask;    // CF_MASK: multiply by (mask[m, n] > 0)
  int x_ld, y_ld, res_ld, mask_ld;
  int B, H, W, Ho, Wo, stride, shift, parity;
  int cin, ntaps;       // channels per tap (multiple of 4; of 16 when groups > 1), taps
  int M, N, K;
  int groups, cpg_in, cpg_out;   // groups > 1: output column n reads K-channels [g*cpg_in, (g+1)*cpg_in), g = n / cpg_out
  int flags;            // CF_BIAS | CF_RES | CF_RELU | CF_MASK
};
hipError_t launch_conv_f32(const ConvF32Params& p, hipStream_t s);
hipError_t launch_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int M, int C, hipStream_t s);
hipError_t launch_copy_f32(const float* a, int lda, float* y, int ldy, int M, int C, hipStream_t s);
// y = dy * (mask > 0 [&& mask < hi when hi > 0])   (ReLU / ReLU6 backward from the stored forward output)
hipError_t launch_mask_f32(const float* dy, int ldd, const float* mask, int ldm, float* y, int ldy, int M, int C, float hi, hipStream_t s);
hipError_t launch_maxpool3x3s2_f32(const float* x, float* y, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_maxpool3x3s2_bwd_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s);
// src fp32 NHWC [B,Hs,Ws,ld_s] -> dst fp32 [B,Hd,Wd,ld_d]; the transpose writes bf16 rows (dsrc_bf16) or fp32
hipError_t launch_bicubic_f32(const float* src, int ld_s, float* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C, int Cpad,
                              hipStream_t s);
hipError_t launch_bicubic_bwd_f32(const float* ddst, int ld_d, void* dsrc, int dsrc_bf16, int ld_s, int B, int Hs, int Ws, int Hd, int Wd,
                                  int C, hipStream_t s);
// global average (or max, model_utils.py:34-35) pool of fp32 rows -> f [B, C]; argmax [B, C] only for the max form
hipError_t launch_gap_f32(const float* x, int ld, float* f, int* argmax, int B, int HW, int C, int use_max, hipStream_t s);
hipError_t launch_gap_bwd_f32(const float* gf, float* dx, int ld, int B, int HW, int C, const int* argmax, hipStream_t s);
hipError_t launch_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cpad, int ld, hipStream_t s);

// ----------------------------------------------------------------------------------------------
// host-side weight packing (weights.cpp)
// ----------------------------------------------------------------------------------------------
struct PackedConv {
  int N, K, cin, ntaps;   // GEMM N, padded K, padded channels per tap, taps
};
// mode 0: forward conv/linear weights [Cout][Cin][KH][KW] -> [N=Cout][K], taps (ky-pad, kx-pad)
// mode 1: input-gradient (dgrad) weights               -> [N=Cin][K=(tap, cout)], taps (pad-ky, pad-kx)
// geglu: the Cout dimension is permuted into 32-wide (16 hidden | 16 gate) groups (see CF_GEGLU)
PackedConv pack_conv_shape(int Cout, int Cin, int KH, int KW, int mode);
void pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int geglu, bf16_t* wp, int* taptab);
// fp32 packing for the guide network (guide_f32.hip): cin padded to 4 (16 when groups > 1), K padded to 16, k = (tap, cin);
// w is [Cout][Cin/groups][KH][KW] (torch grouped-conv layout), packed as the dense block-diagonal matrix
PackedConv pack_conv_shape_f32(int Cout, int Cin, int KH, int KW, int mode, int groups);
void pack_conv_weight_f32(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int groups, float* wp, int* taptab);
int geglu_perm(int packed_index, int F);  // packed column -> original row of the [2F] projection
bf16_t host_f2bf(float f);
float host_bf2f(bf16_t v);
