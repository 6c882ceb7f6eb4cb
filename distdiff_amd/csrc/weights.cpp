// Host-side weight pre-packing for the implicit-GEMM kernel (conv_gemm.hip): OIHW fp32 -> [N][K] bf16 with
// k = (tap, cin) and the per-tap offset table; transposed + flipped packing for the input-gradient GEMMs.
#include <string.h>
#include "kernels.h"

bf16_t host_f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
float host_bf2f(bf16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }

int geglu_perm(int pr, int F) {
  const int group = pr >> 5, within = pr & 31;
  return within < 16 ? group * 16 + within : F + group * 16 + (within - 16);
}

PackedConv pack_conv_shape(int Cout, int Cin, int KH, int KW, int mode) {
  PackedConv s;
  const int cin = mode == 0 ? Cin : Cout;
  s.N = mode == 0 ? Cout : Cin;
  s.cin = (cin + 7) / 8 * 8;
  s.ntaps = KH * KW;
  s.K = (s.ntaps * s.cin + 63) / 64 * 64;
  return s;
}

void pack_conv_weight(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int geglu, bf16_t* wp, int* taptab) {
  const PackedConv s = pack_conv_shape(Cout, Cin, KH, KW, mode);
  memset(wp, 0, (size_t)s.N * s.K * sizeof(bf16_t));
  const int F = Cout / 2;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int tap = ky * KW + kx;
      const int dy = mode == 0 ? ky - pad : pad - ky, dx = mode == 0 ? kx - pad : pad - kx;
      taptab[tap] = ((dy + 32) << 6) | (dx + 32);
    }
  // K order: (tap, channel) in general; when the channel count is a multiple of 64 the kernels walk K as
  // (64-channel chunk, tap, channel-in-chunk), so the 9 taps of one channel chunk are consecutive K-steps and re-read the
  // same (shifted) input rows while they are still in L2 (conv_gemm.hip / conv_gemm2.hip, uniform-tap path)
  const int nt = KH * KW;
  auto kidx = [&](int t, int c) -> size_t {
    if ((s.cin & 63) == 0) return (size_t)((c >> 6) * nt + t) * 64 + (c & 63);
    return (size_t)t * s.cin + c;
  };
  if (mode == 0) {
    for (int n = 0; n < Cout; ++n) {
      const int on = geglu ? geglu_perm(n, F) : n;
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < nt; ++t) wp[(size_t)n * s.K + kidx(t, c)] = host_f2bf(w[((size_t)on * Cin + c) * nt + t]);
    }
  } else {
    for (int c = 0; c < Cin; ++c)
      for (int n = 0; n < Cout; ++n) {
        const int on = geglu ? geglu_perm(n, F) : n;
        for (int t = 0; t < nt; ++t) wp[(size_t)c * s.K + kidx(t, n)] = host_f2bf(w[((size_t)on * Cin + c) * nt + t]);
      }
  }
}

PackedConv pack_conv_shape_f32(int Cout, int Cin, int KH, int KW, int mode, int groups) {
  PackedConv s;
  const int cin = mode == 0 ? Cin : Cout;
  const int al = groups > 1 ? 16 : 4;
  s.N = mode == 0 ? Cout : Cin;
  s.cin = (cin + al - 1) / al * al;
  s.ntaps = KH * KW;
  s.K = (s.ntaps * s.cin + 15) / 16 * 16;
  return s;
}

void pack_conv_weight_f32(const float* w, int Cout, int Cin, int KH, int KW, int pad, int mode, int groups, float* wp, int* taptab) {
  const PackedConv s = pack_conv_shape_f32(Cout, Cin, KH, KW, mode, groups);
  memset(wp, 0, (size_t)s.N * s.K * sizeof(float));
  const int nt = KH * KW;
  for (int ky = 0; ky < KH; ++ky)
    for (int kx = 0; kx < KW; ++kx) {
      const int dy = mode == 0 ? ky - pad : pad - ky, dx = mode == 0 ? kx - pad : pad - kx;
      taptab[ky * KW + kx] = ((dy + 32) << 6) | (dx + 32);
    }
  const int cpg_in = Cin / groups, cpg_out = Cout / groups;
  for (int n = 0; n < Cout; ++n) {
    const int g = n / cpg_out;
    for (int ci = 0; ci < cpg_in; ++ci) {
      const int c = g * cpg_in + ci;
      for (int t = 0; t < nt; ++t) {
        const float v = w[((size_t)n * cpg_in + ci) * nt + t];
        if (mode == 0) wp[(size_t)n * s.K + (size_t)t * s.cin + c] = v;
        else wp[(size_t)c * s.K + (size_t)t * s.cin + n] = v;
      }
    }
  }
}
