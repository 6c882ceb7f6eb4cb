// extern "C" op-level entry points (include/distdiff_hip_ops.h): thin forwards to the launchers.
#include "kernels.h"
#include "../../include/distdiff_hip_ops.h"

#define S(x) ((hipStream_t)(x))
extern "C" {
int dd_op_conv_gemm(const ConvGemmParams* p, size_t cap, void* st) {
  // Stream-asynchronous like every other launcher (no host round trip: graph-capturable, and back-to-back launches stay back to back).
  // Contract: a one-tap, stride-1, same-size launch is a 1x1 / linear layer and takes the persistent kernels' pointwise path, which
  // does not read the tap table -- it IS the centre tap, what every packer emits for such layers.  The host side that owns the table
  // validates it when the weights are packed (distdiff_amd/ops.py: PackedConv.taptab_host), not here per launch.
  return (int)launch_conv_gemm(*p, cap, S(st));
}
int dd_op_conv_gemm_check(const ConvGemmParams* p, void* st) {
  // The synchronous companion for ABI users who want the contract checked against the DEVICE table: waits for the stream, reads the
  // tap table back and validates it (entries in range; a one-tap stride-1 same-size launch carries the centre tap).  0 = fine.
  if (!p || !p->taptab || p->ntaps < 1 || p->ntaps > 64) return (int)hipErrorInvalidValue;
  int taps[64];
  hipError_t e = hipStreamSynchronize(S(st));
  if (e == hipSuccess) e = hipMemcpy(taps, p->taptab, sizeof(int) * p->ntaps, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  for (int t = 0; t < p->ntaps; ++t) {
    const int dy = ((taps[t] >> 6) & 63) - 32, dx = (taps[t] & 63) - 32;
    if ((taps[t] >> 12) != 0 || dy < -31 || dy > 31 || dx < -31 || dx > 31) return (int)hipErrorInvalidValue;
  }
  if (p->ntaps == 1 && p->stride == 1 && !p->shift && p->H == p->Ho && p->W == p->Wo && taps[0] != ((32 << 6) | 32)) return (int)hipErrorInvalidValue;
  return 0;
}
int dd_op_groupnorm_fwd(const GroupNormParams* p, void* st) { return (int)launch_groupnorm_fwd(*p, S(st)); }
int dd_op_groupnorm_bwd(const GroupNormParams* p, void* st) { return (int)launch_groupnorm_bwd(*p, S(st)); }
size_t dd_op_groupnorm_scratch_bytes(int B, int G) { return groupnorm_scratch_bytes(B, G); }
int dd_op_layernorm_fwd(const LayerNormParams* p, void* st) { return (int)launch_layernorm_fwd(*p, S(st)); }
int dd_op_layernorm_bwd(const LayerNormParams* p, void* st) { return (int)launch_layernorm_bwd(*p, S(st)); }
int dd_op_attention_fwd(const AttnParams* p, void* st) { return (int)launch_attention_fwd(*p, S(st)); }
int dd_op_attention_bwd(const AttnParams* p, void* st) { return (int)launch_attention_bwd(*p, S(st)); }
size_t dd_op_attention_gemm_workspace(int Nq, int Nk, int D, int bwd) { return attention_gemm_workspace(Nq, Nk, D, bwd); }
int dd_op_attention_gemm_fwd(const AttnParams* p, void* ws, size_t ws_bytes, const int* tap1x1, float* partial, size_t cap, void* st) {
  return (int)launch_attention_gemm_fwd(*p, ws, ws_bytes, tap1x1, partial, cap, S(st));
}
int dd_op_attention_gemm_bwd(const AttnParams* p, void* ws, size_t ws_bytes, const int* tap1x1, float* partial, size_t cap, void* st) {
  return (int)launch_attention_gemm_bwd(*p, ws, ws_bytes, tap1x1, partial, cap, S(st));
}

int dd_op_conv_f32(const ConvF32Params* p, void* st) { return (int)launch_conv_f32(*p, S(st)); }
int dd_pack_conv_weight_f32(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int groups, float* wp, int* taptab,
                            int* out4) {
  const PackedConv s = pack_conv_shape_f32(Cout, Cin, KH, KW, mode, groups);
  if (out4) { out4[0] = s.N; out4[1] = s.K; out4[2] = s.cin; out4[3] = s.ntaps; }
  if (wp) pack_conv_weight_f32(w, Cout, Cin, KH, KW, pad, mode, groups, wp, taptab);
  return 0;
}
int dd_pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int geglu, uint16_t* wp,
                        int* taptab, int* out4) {
  const PackedConv s = pack_conv_shape(Cout, Cin, KH, KW, mode);
  if (out4) { out4[0] = s.N; out4[1] = s.K; out4[2] = s.cin; out4[3] = s.ntaps; }
  if (wp) pack_conv_weight(w, Cout, Cin, KH, KW, pad, mode, geglu, wp, taptab);
  return 0;
}
int dd_op_nchw_f32_to_nhwc_bf16(const float* src, uint16_t* dst, int B, int C, int H, int W, int Cpad, int ld, int dup,
                                float scale, void* st) {
  return (int)launch_nchw_f32_to_nhwc_bf16(src, dst, B, C, H, W, Cpad, ld, dup, scale, S(st));
}
int dd_op_nhwc_to_nchw_f32(const void* src, int src_f32, float* dst, int B, int C, int H, int W, int ld, float scale, float shift,
                           int clamp, float lo, float hi, void* st) {
  return (int)launch_nhwc_to_nchw_f32(src, src_f32, dst, B, C, H, W, ld, scale, shift, clamp, lo, hi, S(st));
}
int dd_op_cfg_ddim(const float* eps2, int ld, const float* z, float* z_prev, float* x0, int B, int C, int HW, const float* coef,
                   void* st) {
  return (int)launch_cfg_ddim(eps2, ld, z, z_prev, x0, B, C, HW, coef, S(st));
}
int dd_op_cfg_ddim_bwd(const float* g_x0, const float* g_zprev, uint16_t* g_eps2, int ld, float* g_z, int B, int C, int HW,
                       const float* coef, void* st) {
  return (int)launch_cfg_ddim_bwd(g_x0, g_zprev, g_eps2, ld, g_z, B, C, HW, coef, S(st));
}
int dd_op_sumpool2x2(const uint16_t* src, int src_ld, uint16_t* dst, int dst_ld, int B, int H, int W, int C, int acc, void* st) {
  return (int)launch_sumpool2x2(src, src_ld, dst, dst_ld, B, H, W, C, acc, S(st));
}
int dd_op_geglu_bwd(const uint16_t* raw, int ld_raw, const uint16_t* dout, int ld_dout, uint16_t* draw, int ld_draw, int M, int F,
                    void* st) {
  return (int)launch_geglu_bwd(raw, ld_raw, dout, ld_dout, draw, ld_draw, M, F, S(st));
}
int dd_op_maxpool3x3s2(const uint16_t* x, uint16_t* y, int B, int H, int W, int C, void* st) {
  return (int)launch_maxpool3x3s2(x, y, B, H, W, C, S(st));
}
int dd_op_maxpool3x3s2_bwd(const uint16_t* x, const uint16_t* dy, uint16_t* dx, int B, int H, int W, int C, void* st) {
  return (int)launch_maxpool3x3s2_bwd(x, dy, dx, B, H, W, C, S(st));
}
int dd_op_bicubic(const uint16_t* src, int ld_s, uint16_t* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C, int Cpad,
                  void* st) {
  return (int)launch_bicubic(src, ld_s, dst, ld_d, B, Hs, Ws, Hd, Wd, C, Cpad, S(st));
}
int dd_op_bicubic_bwd(const uint16_t* ddst, int ld_d, uint16_t* dsrc, int ld_s, int B, int Hs, int Ws, int Hd, int Wd, int C,
                      void* st) {
  return (int)launch_bicubic_bwd(ddst, ld_d, dsrc, ld_s, B, Hs, Ws, Hd, Wd, C, S(st));
}
int dd_op_gap(const uint16_t* x, int ld, float* f, int B, int HW, int C, void* st) { return (int)launch_gap(x, ld, f, B, HW, C, S(st)); }
int dd_op_energy(const float* f, const float* Pc, const float* Pg, const int* targets, int B, int D, int K, float gs, float ls,
                 int use_c, int use_g, int normalize, float weight, float* score_out, float* gf, void* st) {
  return (int)launch_energy(f, Pc, Pg, targets, B, D, K, gs, ls, use_c, use_g, normalize, weight, nullptr, score_out, nullptr, gf, S(st));
}
int dd_op_transform_update(const float* z, const float* g, const float* e, const float* b, float* z_out, int BC, int HW, float rho,
                           float c, void* st) {
  return (int)launch_transform_update(z, g, e, b, z_out, BC, HW, rho, c, S(st));
}
int dd_op_affine(const float* z, const float* e, const float* b, float* out, int BC, int HW, void* st) {
  return (int)launch_affine(z, e, b, out, BC, HW, S(st));
}
}
