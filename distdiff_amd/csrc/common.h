// Common device/host helpers for the distdiff_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bf16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DD_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN preserved
  return __builtin_bit_cast(unsigned short, h);
}
// two floats -> one dword of two bf16 (lo in bits 0-15): a single v_cvt_pk_bf16_f32 (RNE), not two conversions + shift + or
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v; v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU per element: GroupNorm+SiLU touches every activation)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) { float s = __builtin_amdgcn_rcpf(1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
// erf-GELU (torch default, approximate='none') and its derivative
// erf for the GEGLU epilogue: Abramowitz-Stegun 7.1.26, branch-free (rcp + exp2 + 6 fma), |error| <= 5e-7 in fp32 -- four orders of
// magnitude below the bf16 rounding of the result (the reference evaluates GELU in fp16).  The library erff costs ~3x the VALU and the
// 5-step GEGLU items are epilogue-bound (conv family -14 ms per bench step, tools/ab_ops.sh).  The exact erff stays in dgelu_f.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  return copysignf(__builtin_fmaf(-p * t, e, 1.f), x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erf_as(x * 0.70710678118654752f)); }
// GELU for the GEGLU epilogues WITHOUT transcendentals: gelu(x) = x * Phi(x), Phi(x) ~ 0.5 + t * P(t^2) with t = clamp(x, -4.5, 4.5) and P
// the degree-9 minimax polynomial (10 coefficients, fitted by linear programming under the constraint Phi~(4.5) = 1, so that the
// clamped ends are exactly 0 / 1).  13 full-rate VALU (v_med3, v_mul, 9 v_fma, v_fma, v_mul) instead of ~20 + v_rcp + v_exp: the K = 320
// GEGLU projection is bound by the vector issue port (an MFMA holds it for 8 of its 16 cycles, MI355X guide), not by the matrix pipe.
// Accuracy (tests/test_oracle.py::test_gelu_polynomial, evaluated in fp32 from THESE literals): |gelu~ - gelu| <= 5e-5 everywhere,
// relative error <= 1.2e-5 for x > 0 and <= 1e-4 for x > -2; beyond -3 the value itself is < 4e-3 and only the absolute bound holds.
// One bf16 rounding of the product is 2e-3 relative: 0.1 / 0.4 / 6.8 % of the bf16 outputs differ from the rounded exact product for
// gate pre-activations of sigma 0.5 / 1 / 2 (A&S form above: 0.003 / 0.009 / 1.4 %), by one ulp except in the tail.
#define DD_GELU_CLAMP 4.5f
#define DD_GELU_POLY { 3.989228904e-01f, -6.641460210e-02f, 9.885823354e-03f, -1.140101929e-03f, 1.011194836e-04f, \
                       -6.729636425e-06f, 3.209943316e-07f, -1.023312812e-08f, 1.934523375e-10f, -1.629367648e-12f }
__device__ __forceinline__ float gelu_poly_f(float x) {
  constexpr float c[10] = DD_GELU_POLY;
  const float t = __builtin_amdgcn_fmed3f(x, -DD_GELU_CLAMP, DD_GELU_CLAMP);
  const float u = t * t;
  float p = c[9];
#pragma unroll
  for (int k = 8; k >= 0; --k) p = __builtin_fmaf(p, u, c[k]);
  return x * __builtin_fmaf(t, p, 0.5f);
}
#ifdef DD_GELU_ERF_AS      // A/B build: the rcp + exp2 form in the epilogues
__device__ __forceinline__ float gelu_f(float x) { return gelu_erf_f(x); }
#else
__device__ __forceinline__ float gelu_f(float x) { return gelu_poly_f(x); }
#endif
__device__ __forceinline__ float dgelu_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
