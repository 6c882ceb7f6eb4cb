// Shared epilogue of the implicit-GEMM kernels: bias / per-timestep bias / GEGLU / residual / ReLU / ReLU-mask, bf16 or fp32 out.
#pragma once
#include "common.h"
#include "kernels.h"

namespace {

struct Epi {
  // applies bias / GEGLU / residual / relu / mask and stores 4 consecutive output columns of row m.
  static __device__ __forceinline__ void apply(const ConvGemmParams& p, const float* bias, int m, int nb, float* h,
                                               float* g, int nb_gate) {
    const int flags = p.flags;
    int ncols;  // logical output columns
    int ob;     // output column base
    if (flags & CF_LNFOLD) {
      // LayerNorm folded into this GEMM (kernels.h CF_LNFOLD): acc -> rstd * (acc - mean * c1[n]); the folded bias follows as usual
      const float mean = p.ln_stats[(size_t)m * 2], rstd = p.ln_stats[(size_t)m * 2 + 1];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (nb + r < p.N) h[r] = rstd * (h[r] - mean * p.ln_c1[nb + r]);
        if ((flags & CF_GEGLU) && nb_gate + r < p.N) g[r] = rstd * (g[r] - mean * p.ln_c1[nb_gate + r]);
      }
    }
    if (flags & CF_GEGLU) {
      ncols = p.N >> 1;
      ob = (nb >> 5) * 16 + (nb & 15);
      float hvv[4], gvv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float hv = h[r] * p.alpha, gv = g[r] * p.alpha;
        if (flags & CF_BIAS) {
          if (nb + r < p.N) hv += bias[nb + r];
          if (nb_gate + r < p.N) gv += bias[nb_gate + r];
        }
        hvv[r] = hv; gvv[r] = gv;
        h[r] = hv * gelu_f(gv);
      }
      if (flags & CF_GEGLU_RAW) {   // stash of the pre-activations for the VJP: two 8-byte stores
        bf16_t* rp = p.raw + (size_t)m * p.raw_ld;
        if (nb_gate + 4 <= p.N && !(p.raw_ld & 3)) {
          uint2 a, b;
          a.x = pack2bf(hvv[0], hvv[1]); a.y = pack2bf(hvv[2], hvv[3]);
          b.x = pack2bf(gvv[0], gvv[1]); b.y = pack2bf(gvv[2], gvv[3]);
          *(uint2*)(rp + nb) = a;
          *(uint2*)(rp + nb_gate) = b;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb_gate + r < p.N) { rp[nb + r] = f2bf(hvv[r]); rp[nb_gate + r] = f2bf(gvv[r]); }
        }
      }
    } else {
      ncols = p.N;
      ob = nb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = h[r] * p.alpha;
        if ((flags & CF_BIAS) && nb + r < p.N) v += bias[nb + r];
        h[r] = v;
      }
    }
    const bool full = (ob + 4 <= ncols);
    if (flags & CF_RES) {
      if (flags & CF_RES_F32) {
        const float* rp = (const float*)p.res + (size_t)m * p.res_ld + ob;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ob + r < ncols) h[r] += rp[r];
      } else {
        const bf16_t* rp = (const bf16_t*)p.res + (size_t)m * p.res_ld + ob;
        if (full && !(p.res_ld & 3)) {
          uint2 rv = *(const uint2*)rp;
          h[0] += __uint_as_float(rv.x << 16); h[1] += __uint_as_float(rv.x & 0xffff0000u);
          h[2] += __uint_as_float(rv.y << 16); h[3] += __uint_as_float(rv.y & 0xffff0000u);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ob + r < ncols) h[r] += bf2f(rp[r]);
        }
      }
    }
    if (flags & CF_RELU) {
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = fmaxf(h[r], 0.f);
    }
    if (flags & CF_MASK) {
      const bf16_t* mp = p.mask + (size_t)m * p.mask_ld + ob;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ob + r < ncols && !(bf2f(mp[r]) > 0.f)) h[r] = 0.f;
    }
    if (flags & CF_OUT_F32) {
      float* yp = (float*)p.y + (size_t)m * p.y_ld + ob;
      if (full && !(p.y_ld & 3)) {
        *(float4*)yp = make_float4(h[0], h[1], h[2], h[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ob + r < ncols) yp[r] = h[r];
      }
    } else {
      bf16_t* yp = (bf16_t*)p.y + (size_t)m * p.y_ld + ob;
      if (full && !(p.y_ld & 3)) {
        uint2 o; o.x = pack2bf(h[0], h[1]); o.y = pack2bf(h[2], h[3]);
        *(uint2*)yp = o;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ob + r < ncols) yp[r] = f2bf(h[r]);
      }
    }
  }
};

}  // namespace
