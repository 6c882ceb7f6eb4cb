// K6/K7/K9/K10/K12 — the HBM-bound side kernels of the guided DDIM loop (gfx950):
// layout conversion, CFG + DDIM update and its VJP (generate_data.py:116-119), add_noise (:1176),
// affine perturbation / SGD / L-inf projection of transform_guidance (:696, :721-728, :124-137),
// direct-guidance update (:762), nearest-2x transpose (2x2 sum pool), GEGLU backward,
// max-pool, bicubic 224 resize (A=-0.75, align_corners=False) and its transpose, GAP,
// prototype energy + argmax + gradient (:707-717 / :747-759).
#include "common.h"
#include "kernels.h"

namespace {

#define GRID_STRIDE(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); i += (size_t)gridDim.x * blockDim.x)

inline int nblocks(size_t n, int threads = 256, int cap = 8192) {
  size_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}

__global__ void nchw_to_nhwc_kernel(const float* src, bf16_t* dst, int B, int C, int HW, int Cpad, int ld, int dup, float scale) {
  const size_t total = (size_t)(dup ? 2 * B : B) * HW * Cpad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const size_t row = i / Cpad;
    const int pix = (int)(row % HW);
    const int b = (int)((row / HW) % B);
    const float v = c < C ? src[((size_t)b * C + c) * HW + pix] * scale : 0.f;
    dst[row * ld + c] = f2bf(v);
  }
}

__global__ void nhwc_to_nchw_kernel(const void* src, int src_f32, float* dst, int B, int C, int HW, int ld, float scale,
                                    float shift, int clamp, float lo, float hi) {
  const size_t total = (size_t)B * C * HW;
  GRID_STRIDE(i, total) {
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((size_t)HW * C));
    const size_t si = ((size_t)b * HW + pix) * ld + c;
    float v = src_f32 ? ((const float*)src)[si] : bf2f(((const bf16_t*)src)[si]);
    v = v * scale + shift;
    if (clamp) v = fminf(fmaxf(v, lo), hi);
    dst[i] = v;
  }
}

__global__ void cfg_ddim_kernel(const float* eps2, int ld, const float* z, float* z_prev, float* x0, int B, int C, int HW,
                                const float* coef) {
  const float s = coef[0], sa = coef[1], s1m = coef[2], sap = coef[3], s1mp = coef[4];
  const size_t total = (size_t)B * C * HW;
  GRID_STRIDE(i, total) {
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((size_t)HW * C));
    const float eu = eps2[((size_t)b * HW + pix) * ld + c];
    const float ec = eps2[((size_t)(B + b) * HW + pix) * ld + c];
    const float eps = eu + s * (ec - eu);
    const float x = (z[i] - s1m * eps) / sa;
    if (x0) x0[i] = x;
    z_prev[i] = sap * x + s1mp * eps;
  }
}

__global__ void cfg_ddim_bwd_kernel(const float* g_x0, const float* g_zprev, bf16_t* g_eps2, int ld, float* g_z, int B, int C,
                                    int HW, int Cpad, const float* coef) {
  const float s = coef[0], sa = coef[1], s1m = coef[2], sap = coef[3], s1mp = coef[4];
  const size_t total = (size_t)B * HW * Cpad;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cpad);
    const size_t row = i / Cpad;
    const int pix = (int)(row % HW);
    const int b = (int)(row / HW);
    float gu = 0.f, gc = 0.f;
    if (c < C) {
      const size_t zi = ((size_t)b * C + c) * HW + pix;
      const float gx = g_x0 ? g_x0[zi] : 0.f;
      const float gp = g_zprev ? g_zprev[zi] : 0.f;
      const float h = gx + sap * gp;
      const float ge = -s1m / sa * h + s1mp * gp;
      g_z[zi] = h / sa;
      gu = (1.f - s) * ge; gc = s * ge;
    }
    g_eps2[((size_t)b * HW + pix) * ld + c] = f2bf(gu);
    g_eps2[((size_t)(B + b) * HW + pix) * ld + c] = f2bf(gc);
  }
}

// g_z[b,c,pix] += gin[(b*HW+pix), c] + gin[((B+b)*HW+pix), c]   (backward of cat[z, z] + layout change)
__global__ void dup_bwd_kernel(const bf16_t* gin, int ld, float* g_z, int B, int C, int HW, int accumulate, int halves) {
  const size_t total = (size_t)B * C * HW;
  GRID_STRIDE(i, total) {
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((size_t)HW * C));
    float v = bf2f(gin[((size_t)b * HW + pix) * ld + c]);
    if (halves == 2) v += bf2f(gin[((size_t)(B + b) * HW + pix) * ld + c]);
    g_z[i] = accumulate ? g_z[i] + v : v;
  }
}

__global__ void axpby_kernel(const float* x, const float* n, float* out, size_t count, const float* coef) {
  const float a = coef[0], b = coef[1];
  GRID_STRIDE(i, count) out[i] = a * x[i] + b * n[i];
}

__global__ void sumpool_kernel(const bf16_t* src, int src_ld, bf16_t* dst, int dst_ld, int B, int H, int W, int C, int acc) {
  const int VC = C >> 3;
  const size_t total = (size_t)B * H * W * VC;
  GRID_STRIDE(i, total) {
    const int vc = (int)(i % VC);
    const size_t pix = i / VC;
    const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float v[8];
        unpack8(*(const uint4*)(src + (((size_t)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * src_ld + vc * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += v[e];
      }
    bf16_t* d = dst + pix * dst_ld + vc * 8;
    if (acc) {
      float o[8];
      unpack8(*(const uint4*)d, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += o[e];
    }
    *(uint4*)d = pack8(a);
  }
}

__global__ void add_kernel(const bf16_t* a, int lda, const bf16_t* b, int ldb, bf16_t* y, int ldy, int M, int C) {
  const int VC = C >> 3;
  GRID_STRIDE(i, (size_t)M * VC) {
    const int vc = (int)(i % VC);
    const size_t m = i / VC;
    float x[8], z[8];
    unpack8(*(const uint4*)(a + m * lda + vc * 8), x);
    if (b) {
      unpack8(*(const uint4*)(b + m * ldb + vc * 8), z);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] += z[e];
    }
    *(uint4*)(y + m * ldy + vc * 8) = pack8(x);
  }
}

// y = dy * (mask > 0)
__global__ void mask_kernel(const bf16_t* dy, int ldd, const bf16_t* mask, int ldm, bf16_t* y, int ldy, int M, int C) {
  const int VC = C >> 3;
  GRID_STRIDE(i, (size_t)M * VC) {
    const int vc = (int)(i % VC);
    const size_t m = i / VC;
    float x[8], z[8];
    unpack8(*(const uint4*)(dy + m * ldd + vc * 8), x);
    unpack8(*(const uint4*)(mask + m * ldm + vc * 8), z);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = z[e] > 0.f ? x[e] : 0.f;
    *(uint4*)(y + m * ldy + vc * 8) = pack8(x);
  }
}

// raw packed [M, 2F]: 32-column groups = 16 hidden | 16 gate. out col o <-> hidden (o/16)*32 + o%16, gate +16
__global__ void geglu_bwd_kernel(const bf16_t* raw, int ld_raw, const bf16_t* dout, int ld_dout, bf16_t* draw, int ld_draw,
                                 int M, int F) {
  const int VC = F >> 3;
  GRID_STRIDE(i, (size_t)M * VC) {
    const int vc = (int)(i % VC);
    const size_t m = i / VC;
    const int o = vc * 8;
    const int hc = (o >> 4) * 32 + (o & 15);
    float h[8], g[8], d[8], dh[8], dg[8];
    unpack8(*(const uint4*)(raw + m * ld_raw + hc), h);
    unpack8(*(const uint4*)(raw + m * ld_raw + hc + 16), g);
    unpack8(*(const uint4*)(dout + m * ld_dout + o), d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dh[e] = d[e] * gelu_f(g[e]);
      dg[e] = d[e] * h[e] * dgelu_f(g[e]);
    }
    *(uint4*)(draw + m * ld_draw + hc) = pack8(dh);
    *(uint4*)(draw + m * ld_draw + hc + 16) = pack8(dg);
  }
}

__global__ void maxpool_kernel(const bf16_t* x, bf16_t* y, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, VC = C >> 3;
  GRID_STRIDE(i, (size_t)B * Ho * Wo * VC) {
    const int vc = (int)(i % VC);
    const size_t pix = i / VC;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
        if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
        float v[8];
        unpack8(*(const uint4*)(x + (((size_t)b * H + iy) * W + ix) * C + vc * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    *(uint4*)(y + pix * C + vc * 8) = pack8(m);
  }
}

// gradient goes to the first maximum of each window in (ky, kx) scan order (PyTorch CPU/CUDA max_pool2d).
__global__ void maxpool_bwd_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, VC = C >> 3;
  GRID_STRIDE(i, (size_t)B * H * W * VC) {
    const int vc = (int)(i % VC);
    const size_t pix = i / VC;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows (oy, ox) that contain (iy, ix): oy*2-1 <= iy <= oy*2+1
    for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        float m[8]; int am[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; am[e] = -1; }
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int yy = oy * 2 + ky - 1, xx = ox * 2 + kx - 1;
            if (yy < 0 || xx < 0 || yy >= H || xx >= W) continue;
            float v[8];
            unpack8(*(const uint4*)(x + (((size_t)b * H + yy) * W + xx) * C + vc * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (v[e] > m[e]) { m[e] = v[e]; am[e] = yy * W + xx; }
          }
        float d[8];
        unpack8(*(const uint4*)(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + vc * 8), d);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (am[e] == iy * W + ix) g[e] += d[e];
      }
    }
    *(uint4*)(dx + pix * C + vc * 8) = pack8(g);
  }
}

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
  const float A = -0.75f;
  w[0] = cc2(t + 1.f, A); w[1] = cc1(t, A); w[2] = cc1(1.f - t, A); w[3] = cc2(2.f - t, A);
}

__global__ void bicubic_kernel(const bf16_t* src, int ld_s, bf16_t* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C,
                               int Cpad) {
  const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
  GRID_STRIDE(i, (size_t)B * Hd * Wd) {
    const int ox = (int)(i % Wd), oy = (int)((i / Wd) % Hd), b = (int)(i / ((size_t)Wd * Hd));
    const float ry = sh * (oy + 0.5f) - 0.5f, rx = sw * (ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    float wy[4], wx[4];
    cubic_coeffs(ry - fy, wy); cubic_coeffs(rx - fx, wx);
    const int iy = (int)fy, ix = (int)fx;
    for (int c = 0; c < Cpad; ++c) {
      float acc = 0.f;
      if (c < C) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int yy = min(max(iy - 1 + a, 0), Hs - 1);
          float r = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int xx = min(max(ix - 1 + q, 0), Ws - 1);
            r += wx[q] * bf2f(src[(((size_t)b * Hs + yy) * Ws + xx) * ld_s + c]);
          }
          acc += wy[a] * r;
        }
      }
      dst[i * ld_d + c] = f2bf(acc);
    }
  }
}

// Transposed bicubic as a gather over destination pixels whose 4x4 footprints (after border clamping) hit this
// source pixel: deterministic, no atomics. For a down-scale factor s the candidate dst range is small.
__global__ void bicubic_bwd_kernel(const bf16_t* ddst, int ld_d, bf16_t* dsrc, int ld_s, int B, int Hs, int Ws, int Hd, int Wd,
                                   int C) {
  const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
  GRID_STRIDE(i, (size_t)B * Hs * Ws) {
    const int sx = (int)(i % Ws), sy = (int)((i / Ws) % Hs), b = (int)(i / ((size_t)Ws * Hs));
    // dst rows whose taps (fy-1..fy+2, clamped) can touch sy: fy in [sy-2, sy+1] -> invert ry = sh*(d+.5)-.5
    int dy_lo = (int)floorf((sy - 2 + 0.5f) / sh - 0.5f) - 1, dy_hi = (int)ceilf((sy + 2 + 0.5f) / sh - 0.5f) + 1;
    int dx_lo = (int)floorf((sx - 2 + 0.5f) / sw - 0.5f) - 1, dx_hi = (int)ceilf((sx + 2 + 0.5f) / sw - 0.5f) + 1;
    if (sy == 0) dy_lo = 0;
    if (sy == Hs - 1) dy_hi = Hd - 1;
    if (sx == 0) dx_lo = 0;
    if (sx == Ws - 1) dx_hi = Wd - 1;
    dy_lo = max(dy_lo, 0); dy_hi = min(dy_hi, Hd - 1); dx_lo = max(dx_lo, 0); dx_hi = min(dx_hi, Wd - 1);
    float acc[4] = {0, 0, 0, 0};
    for (int oy = dy_lo; oy <= dy_hi; ++oy) {
      const float ry = sh * (oy + 0.5f) - 0.5f, fy = floorf(ry);
      float wy[4]; cubic_coeffs(ry - fy, wy);
      float wys = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) wys += (min(max((int)fy - 1 + a, 0), Hs - 1) == sy) ? wy[a] : 0.f;
      if (wys == 0.f) continue;
      for (int ox = dx_lo; ox <= dx_hi; ++ox) {
        const float rx = sw * (ox + 0.5f) - 0.5f, fx = floorf(rx);
        float wx[4]; cubic_coeffs(rx - fx, wx);
        float wxs = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) wxs += (min(max((int)fx - 1 + q, 0), Ws - 1) == sx) ? wx[q] : 0.f;
        if (wxs == 0.f) continue;
        const bf16_t* d = ddst + (((size_t)b * Hd + oy) * Wd + ox) * ld_d;
        for (int c = 0; c < C && c < 4; ++c) acc[c] += wys * wxs * bf2f(d[c]);
      }
    }
    for (int c = 0; c < C && c < 4; ++c) dsrc[i * ld_s + c] = f2bf(acc[c]);
  }
}

__global__ void gap_kernel(const bf16_t* x, int ld, float* f, int B, int HW, int C) {
  GRID_STRIDE(i, (size_t)B * C) {
    const int c = (int)(i % C), b = (int)(i / C);
    float s = 0.f;
    for (int pxl = 0; pxl < HW; ++pxl) s += bf2f(x[((size_t)b * HW + pxl) * ld + c]);
    f[i] = s / HW;
  }
}

__global__ void gap_bwd_kernel(const float* gf, bf16_t* dx, int ld, int B, int HW, int C, const bf16_t* mask, int mask_ld) {
  GRID_STRIDE(i, (size_t)B * HW * C) {
    const int c = (int)(i % C);
    const size_t row = i / C;
    const int b = (int)(row / HW);
    float v = gf[(size_t)b * C + c] / HW;
    if (mask && !(bf2f(mask[row * mask_ld + c]) > 0.f)) v = 0.f;
    dx[row * ld + c] = f2bf(v);
  }
}

__device__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// single block: loops over the batch so that the score accumulation order is fixed (deterministic)
__global__ __launch_bounds__(256) void energy_kernel(const float* f, const float* Pc, const float* Pg, const int* targets, int B,
                                                     int D, int K, float gs, float ls, int use_c, int use_g, int normalize,
                                                     float weight, const float* sample_w, float* score_out, float* image_scores,
                                                     float* gf) {
  __shared__ float red[8];
  __shared__ float dots[64];
  float score = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* fb = f + (size_t)b * D;
    const int y = targets[b];
    const float wb = sample_w ? sample_w[b] : 1.f / B;   // the reference's mean over its batch group (:709, :716, :751, :758)
    float nrm = 1.f;
    if (normalize) {
      float s = 0.f;
      for (int d = threadIdx.x; d < D; d += blockDim.x) s += fb[d] * fb[d];
      nrm = sqrtf(block_sum(s, red));
    }
    const float inv = 1.f / nrm;
    float dc = 0.f, dg = 0.f;
    int idx = 0;
    const float* pc = Pc ? Pc + (size_t)y * D : nullptr;
    if (use_c) {
      float s = 0.f;
      for (int d = threadIdx.x; d < D; d += blockDim.x) { const float t = fb[d] * inv - pc[d]; s += t * t; }
      dc = sqrtf(block_sum(s, red));
    }
    const float* pg = nullptr;
    if (use_g) {
      for (int k = 0; k < K; ++k) {
        const float* pk = Pg + ((size_t)y * K + k) * D;
        float s = 0.f;
        for (int d = threadIdx.x; d < D; d += blockDim.x) s += fb[d] * inv * pk[d];
        s = block_sum(s, red);
        if (threadIdx.x == 0) dots[k] = s;
      }
      __syncthreads();
      float best = dots[0];
      for (int k = 1; k < K; ++k)
        if (dots[k] > best) { best = dots[k]; idx = k; }   // first maximum, like torch.argmax
      pg = Pg + ((size_t)y * K + idx) * D;
      float s = 0.f;
      for (int d = threadIdx.x; d < D; d += blockDim.x) { const float t = fb[d] * inv - pg[d]; s += t * t; }
      dg = sqrtf(block_sum(s, red));
    }
    const float Eb = (use_c ? gs * dc : 0.f) + (use_g ? ls * dg : 0.f);
    score += wb * Eb;
    if (image_scores && threadIdx.x == 0) image_scores[b] += weight * Eb;
    if (gf) {
      // gradient wrt fhat, then through the optional normalisation
      const float wc = use_c ? weight * gs * wb / dc : 0.f, wg = use_g ? weight * ls * wb / dg : 0.f;
      float dotp = 0.f;
      if (normalize) {
        float s = 0.f;
        for (int d = threadIdx.x; d < D; d += blockDim.x) {
          const float fh = fb[d] * inv;
          float g = 0.f;
          if (use_c) g += wc * (fh - pc[d]);
          if (use_g) g += wg * (fh - pg[d]);
          s += g * fh;
        }
        dotp = block_sum(s, red);
      }
      for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float fh = fb[d] * inv;
        float g = 0.f;
        if (use_c) g += wc * (fh - pc[d]);
        if (use_g) g += wg * (fh - pg[d]);
        if (normalize) g = (g - fh * dotp) * inv;
        gf[(size_t)b * D + d] = g;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) score_out[0] += weight * score;
}

__global__ void affine_kernel(const float* z, const float* e, const float* b, float* out, int BC, int HW) {
  GRID_STRIDE(i, (size_t)BC * HW) {
    const int bc = (int)(i / HW);
    out[i] = z[i] * (1.f + e[bc]) + b[bc];
  }
}

// one block per (b, c)
__global__ __launch_bounds__(256) void transform_update_kernel(const float* z, const float* g, const float* e, const float* b,
                                                               float* z_out, int HW, float rho, float c) {
  __shared__ float red[8];
  const int bc = blockIdx.x;
  const float* zp = z + (size_t)bc * HW;
  const float* gp = g + (size_t)bc * HW;
  float s0 = 0.f, s1 = 0.f;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) { s0 += gp[i] * zp[i]; s1 += gp[i]; }
  const float ge = block_sum(s0, red);
  const float gb = block_sum(s1, red);
  const float en = e[bc] - rho * ge, bn = b[bc] - rho * gb;
  for (int i = threadIdx.x; i < HW; i += blockDim.x) {
    const float zz = zp[i];
    float v = zz * (1.f + en) + bn;
    const float lo = zz - c, hi = zz + c;
    if (v < lo) v = lo;   // tensor_clamp: lower bound first (generate_data.py:129-132)
    if (v > hi) v = hi;
    z_out[(size_t)bc * HW + i] = v;
  }
}

__global__ void sub_scaled_kernel(const float* a, const float* g, float* out, size_t n, float rho) {
  GRID_STRIDE(i, n) out[i] = a[i] - rho * g[i];
}
__global__ void f32_to_bf16_kernel(const float* s, bf16_t* d, size_t n) { GRID_STRIDE(i, n) d[i] = f2bf(s[i]); }
__global__ void fill_kernel(float* d, float v, size_t n) { GRID_STRIDE(i, n) d[i] = v; }
__global__ void to_uint8_kernel(const float* nchw, uint8_t* hwc, int B, int C, int HW) {
  GRID_STRIDE(i, (size_t)B * HW * C) {
    const int c = (int)(i % C);
    const size_t row = i / C;
    const int pix = (int)(row % HW), b = (int)(row / HW);
    float v = nchw[((size_t)b * C + c) * HW + pix];      // already in [0,1]
    v = fminf(fmaxf(v * 255.f + 0.5f, 0.f), 255.f);      // torchvision save_image: mul(255).add_(0.5).clamp_(0,255).to(uint8)
    hwc[i] = (uint8_t)v;
  }
}

// one wave per output element block: y[m, n] = sum_k act(x[m,k]) * W[n,k] + b[n]   (fp32, setup-time only)
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* x, const float* W, const float* b, float* y, int M, int N, int K,
                                                         int silu_in) {
  const int lane = threadIdx.x & 63;
  const size_t wid = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  if (wid >= (size_t)M * N) return;
  const int m = (int)(wid / N), n = (int)(wid % N);
  float s = 0.f;
  for (int k = lane; k < K; k += 64) {
    float xv = x[(size_t)m * K + k];
    if (silu_in) xv = silu_f(xv);
    s += xv * W[(size_t)n * K + k];
  }
  s = wave_sum(s);
  if (lane == 0) y[(size_t)m * N + n] = s + (b ? b[n] : 0.f);
}
__global__ void sinusoid_kernel(const float* x, float* out, int n, int dim, int ld, int off, int flip) {
  const int half = dim / 2;
  GRID_STRIDE(i, (size_t)n * half) {
    const int j = (int)(i % half), r = (int)(i / half);
    const float a = x[r] * __expf(-logf(10000.f) * (float)j / (float)half);
    const float sn = sinf(a), cs = cosf(a);
    float* o = out + (size_t)r * ld + off;
    if (flip) { o[j] = cs; o[half + j] = sn; } else { o[j] = sn; o[half + j] = cs; }
  }
}
__global__ void add_outer_kernel(const float* a, const float* v, float* out, int ns, int nb, int cols) {
  GRID_STRIDE(i, (size_t)ns * nb * cols) {
    const int c = (int)(i % cols);
    const size_t row = i / cols;
    out[i] = a[(row / nb) * cols + c] + v[(row % nb) * cols + c];
  }
}
__global__ void add_rowvec_kernel(float* y, const float* v, int rows, int cols) { GRID_STRIDE(i, (size_t)rows * cols) y[i] += v[i % cols]; }
__global__ void scale_rows_kernel(float* x, const float* s, int rows, int cols) { GRID_STRIDE(i, (size_t)rows * cols) x[i] *= s[0]; }

// ---- f-2: stage before the loop (dataloader.py:633-661, 750-811) -----------------------------------------------
// CLIPTextEmbeddings: out[b*T + t, :] = token_embedding[ids[b, t]] + position_embedding[t]
__global__ void clip_embed_kernel(const int* ids, const float* tok, const float* pos, bf16_t* out, int ld, int rows, int T, int C,
                                  int vocab) {
  const size_t total = (size_t)rows * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    const int r = (int)(i / C);
    int id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    out[(size_t)r * ld + c] = f2bf(tok[(size_t)id * C + c] + pos[(size_t)(r % T) * C + c]);
  }
}
// CLIP MLP activation: kind 0 quick_gelu x*sigmoid(1.702x) (SD-1.x text encoder), kind 1 erf-GELU
__global__ void act_bf16_kernel(const bf16_t* x, int ldx, bf16_t* y, int ldy, int M, int C, int kind) {
  const size_t total = (size_t)M * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    const size_t r = i / C;
    const float v = bf2f(x[r * ldx + c]);
    const float o = kind == 0 ? v / (1.f + __expf(-1.702f * v)) : 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    y[r * ldy + c] = f2bf(o);
  }
}
// y = act'(x) * dy  (kind 0 quick_gelu, 1 erf-GELU): backward of act_bf16_kernel (CLIP ViT guide MLP)
__global__ void act_bwd_bf16_kernel(const bf16_t* x, int ldx, const bf16_t* dy, int ldd, bf16_t* dx, int ldo, int M, int C, int kind, int acc) {
  const size_t total = (size_t)M * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    const size_t r = i / C;
    const float v = bf2f(x[r * ldx + c]);
    float d;
    if (kind == 0) { const float sg = 1.f / (1.f + __expf(-1.702f * v)); d = sg * (1.f + 1.702f * v * (1.f - sg)); }
    else d = dgelu_f(v);
    float o = d * bf2f(dy[r * ldd + c]);
    if (acc) o += bf2f(dx[r * ldo + c]);
    dx[r * ldo + c] = f2bf(o);
  }
}
// CLIP ViT patch embedding as a GEMM: image fp32 NHWC [B,S,S,ld] -> rows [B*(S/p)^2, C*p*p] bf16 with k = (c, iy, ix), the flattening
// of conv1.weight [width, C, p, p] (open_clip VisionTransformer.conv1, stride = kernel = p: the patches do not overlap); and its
// transpose (every pixel belongs to exactly one patch: a pure re-layout, no accumulation)
__global__ void patchify_kernel(const float* img, int ld, bf16_t* out, int B, int S, int p, int C) {
  const int g = S / p, K = C * p * p;
  GRID_STRIDE(i, (size_t)B * g * g * K) {
    const int k = (int)(i % K);
    const size_t row = i / K;
    const int px = (int)(row % g), py = (int)((row / g) % g), b = (int)(row / ((size_t)g * g));
    const int ix = k % p, iy = (k / p) % p, c = k / (p * p);
    out[i] = f2bf(img[(((size_t)b * S + py * p + iy) * S + px * p + ix) * ld + c]);
  }
}
__global__ void patchify_bwd_kernel(const bf16_t* gout, float* gimg, int ld, int B, int S, int p, int C) {
  const int g = S / p, K = C * p * p;
  GRID_STRIDE(i, (size_t)B * S * S * C) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int x = (int)(pix % S), y = (int)((pix / S) % S), b = (int)(pix / ((size_t)S * S));
    const size_t row = ((size_t)b * g + y / p) * g + x / p;
    gimg[pix * ld + c] = bf2f(gout[row * K + (c * p + y % p) * p + x % p]);
  }
}
// x = cat([class_embedding, patches], dim=1) + positional_embedding: rows [B*(np+1), W]; backward drops the class row
__global__ void vit_embed_kernel(const bf16_t* patches, int ldp, const float* cls, const float* pos, bf16_t* out, int ldo, int B, int np, int W) {
  GRID_STRIDE(i, (size_t)B * (np + 1) * W) {
    const int c = (int)(i % W);
    const size_t row = i / W;
    const int t = (int)(row % (np + 1)), b = (int)(row / (np + 1));
    const float v = t == 0 ? cls[c] : bf2f(patches[((size_t)b * np + t - 1) * ldp + c]);
    out[row * ldo + c] = f2bf(v + pos[(size_t)t * W + c]);
  }
}
__global__ void vit_embed_bwd_kernel(const bf16_t* gout, int ldo, bf16_t* gp, int ldp, int B, int np, int W) {
  GRID_STRIDE(i, (size_t)B * np * W) {
    const int c = (int)(i % W);
    const size_t row = i / W;
    const int t = (int)(row % np), b = (int)(row / np);
    gp[row * ldp + c] = gout[((size_t)b * (np + 1) + t + 1) * ldo + c];
  }
}
// y[b, :] = x[b * stride, :] (the class token of every image); backward: dx = 0 except those rows (first write of dx)
__global__ void select_rows_kernel(const bf16_t* x, int ldx, bf16_t* y, int ldy, int B, int stride, int C) {
  GRID_STRIDE(i, (size_t)B * C) { const int c = (int)(i % C); const size_t b = i / C; y[b * ldy + c] = x[b * stride * ldx + c]; }
}
__global__ void select_rows_bwd_kernel(const bf16_t* dy, int ldy, bf16_t* dx, int ldx, int B, int stride, int C, int acc) {
  GRID_STRIDE(i, (size_t)B * stride * C) {
    const int c = (int)(i % C);
    const size_t row = i / C;
    const bool sel = (row % stride) == 0;
    float v = sel ? bf2f(dy[(row / stride) * ldy + c]) : 0.f;
    if (acc) v += bf2f(dx[row * ldx + c]);
    dx[row * ldx + c] = f2bf(v);
  }
}
// DiagonalGaussianDistribution.sample() * scaling_factor (dataloader.py:808-809): moments NHWC fp32 [B*HW, ld] = (mean | logvar),
// logvar clamped to [-30, 20]; noise NCHW fp32 or null (-> the mode).  latents NCHW fp32; optional moments_out NCHW [B, 2C, HW].
__global__ void vae_sample_kernel(const float* mom, int ld, const float* noise, float* lat, float* mom_out, int B, int C, int HW,
                                  float scale) {
  const size_t total = (size_t)B * C * HW;
  GRID_STRIDE(i, total) {
    const int pix = (int)(i % HW);
    const int c = (int)((i / HW) % C);
    const int b = (int)(i / ((size_t)HW * C));
    const size_t row = ((size_t)b * HW + pix) * ld;
    const float mean = mom[row + c];
    const float logvar = fminf(fmaxf(mom[row + C + c], -30.f), 20.f);
    const float n = noise ? noise[i] : 0.f;
    lat[i] = (mean + __expf(0.5f * logvar) * n) * scale;
    if (mom_out) {
      mom_out[((size_t)b * 2 * C + c) * HW + pix] = mean;
      mom_out[((size_t)b * 2 * C + C + c) * HW + pix] = logvar;
    }
  }
}
// bf16 rows [M, ld] -> fp32 rows [M, C]
// pooled head of transformers' CLIPTextModelWithProjection (SDXL's second text tower): out[n][j] = sum_c x[n T + eos(n)][c] W[j][c] with
// eos(n) = the FIRST position of the largest token id of prompt n (input_ids.argmax(-1): the eos token is the largest id of the CLIP
// vocabulary); x = final_layer_norm(last layer) in bf16, W = text_projection.weight [Pd][C] fp32 (no bias).  One wave per output.
__global__ void clip_pool_project_kernel(const int* ids, const bf16_t* x, int ld, const float* W, float* out, int T, int C, int Pd) {
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int best = ids[(size_t)n * T], pos = 0;
  for (int t = 1; t < T; ++t) {
    const int v = ids[(size_t)n * T + t];
    if (v > best) { best = v; pos = t; }
  }
  const bf16_t* row = x + ((size_t)n * T + pos) * ld;
  const int j = blockIdx.x * 4 + wave;
  if (j >= Pd) return;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) acc = __builtin_fmaf(bf2f(row[c]), W[(size_t)j * C + c], acc);
  acc = wave_sum(acc);
  if (lane == 0) out[(size_t)n * Pd + j] = acc;
}
__global__ void rows_bf16_to_f32_kernel(const bf16_t* x, int ld, float* y, int M, int C) {
  const size_t total = (size_t)M * C;
  GRID_STRIDE(i, total) { y[i] = bf2f(x[(i / C) * ld + (i % C)]); }
}

}  // namespace

#define LAUNCH(kern, n, ...) hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(256), 0, s, __VA_ARGS__); return hipGetLastError()

hipError_t launch_nchw_f32_to_nhwc_bf16(const float* src, bf16_t* dst, int B, int C, int H, int W, int Cpad, int ld, int dup,
                                        float scale, hipStream_t s) {
  LAUNCH(nchw_to_nhwc_kernel, (size_t)(dup ? 2 : 1) * B * H * W * Cpad, src, dst, B, C, H * W, Cpad, ld, dup, scale);
}
hipError_t launch_nhwc_to_nchw_f32(const void* src, int src_f32, float* dst, int B, int C, int H, int W, int ld, float scale,
                                   float shift, int clamp, float lo, float hi, hipStream_t s) {
  LAUNCH(nhwc_to_nchw_kernel, (size_t)B * C * H * W, src, src_f32, dst, B, C, H * W, ld, scale, shift, clamp, lo, hi);
}
hipError_t launch_cfg_ddim(const float* eps2, int ld, const float* z, float* z_prev, float* x0, int B, int C, int HW,
                           const float* coef_dev, hipStream_t s) {
  LAUNCH(cfg_ddim_kernel, (size_t)B * C * HW, eps2, ld, z, z_prev, x0, B, C, HW, coef_dev);
}
hipError_t launch_cfg_ddim_bwd(const float* g_x0, const float* g_zprev, bf16_t* g_eps2, int ld, float* g_z, int B, int C, int HW,
                               const float* coef_dev, hipStream_t s) {
  LAUNCH(cfg_ddim_bwd_kernel, (size_t)B * HW * ld, g_x0, g_zprev, g_eps2, ld, g_z, B, C, HW, ld, coef_dev);
}
hipError_t launch_dup_bwd(const bf16_t* gin, int ld, float* g_z, int B, int C, int HW, int accumulate, int halves, hipStream_t s) {
  LAUNCH(dup_bwd_kernel, (size_t)B * C * HW, gin, ld, g_z, B, C, HW, accumulate, halves);
}
hipError_t launch_axpby(const float* x, const float* n, float* out, size_t count, const float* coef_dev, hipStream_t s) {
  LAUNCH(axpby_kernel, count, x, n, out, count, coef_dev);
}
hipError_t launch_sumpool2x2(const bf16_t* src, int src_ld, bf16_t* dst, int dst_ld, int B, int H, int W, int C, int accumulate,
                             hipStream_t s) {
  LAUNCH(sumpool_kernel, (size_t)B * H * W * (C / 8), src, src_ld, dst, dst_ld, B, H, W, C, accumulate);
}
hipError_t launch_add_bf16(const bf16_t* a, int lda, const bf16_t* b, int ldb, bf16_t* y, int ldy, int M, int C, hipStream_t s) {
  LAUNCH(add_kernel, (size_t)M * (C / 8), a, lda, b, ldb, y, ldy, M, C);
}
hipError_t launch_copy_bf16(const bf16_t* a, int lda, bf16_t* y, int ldy, int M, int C, hipStream_t s) {
  LAUNCH(add_kernel, (size_t)M * (C / 8), a, lda, (const bf16_t*)nullptr, 0, y, ldy, M, C);
}
hipError_t launch_mask_bf16(const bf16_t* dy, int ldd, const bf16_t* mask, int ldm, bf16_t* y, int ldy, int M, int C,
                            hipStream_t s) {
  LAUNCH(mask_kernel, (size_t)M * (C / 8), dy, ldd, mask, ldm, y, ldy, M, C);
}
hipError_t launch_geglu_bwd(const bf16_t* raw, int ld_raw, const bf16_t* dout, int ld_dout, bf16_t* draw, int ld_draw, int M,
                            int F, hipStream_t s) {
  LAUNCH(geglu_bwd_kernel, (size_t)M * (F / 8), raw, ld_raw, dout, ld_dout, draw, ld_draw, M, F);
}
hipError_t launch_maxpool3x3s2(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, hipStream_t s) {
  LAUNCH(maxpool_kernel, (size_t)B * (H / 2) * (W / 2) * (C / 8), x, y, B, H, W, C);
}
hipError_t launch_maxpool3x3s2_bwd(const bf16_t* x, const bf16_t* dy, bf16_t* dx, int B, int H, int W, int C, hipStream_t s) {
  LAUNCH(maxpool_bwd_kernel, (size_t)B * H * W * (C / 8), x, dy, dx, B, H, W, C);
}
hipError_t launch_bicubic(const bf16_t* src, int ld_s, bf16_t* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C,
                          int Cpad, hipStream_t s) {
  LAUNCH(bicubic_kernel, (size_t)B * Hd * Wd, src, ld_s, dst, ld_d, B, Hs, Ws, Hd, Wd, C, Cpad);
}
hipError_t launch_bicubic_bwd(const bf16_t* ddst, int ld_d, bf16_t* dsrc, int ld_s, int B, int Hs, int Ws, int Hd, int Wd, int C,
                              hipStream_t s) {
  if (C > 4) return hipErrorInvalidValue;
  LAUNCH(bicubic_bwd_kernel, (size_t)B * Hs * Ws, ddst, ld_d, dsrc, ld_s, B, Hs, Ws, Hd, Wd, C);
}
hipError_t launch_gap(const bf16_t* x, int ld, float* f, int B, int HW, int C, hipStream_t s) {
  LAUNCH(gap_kernel, (size_t)B * C, x, ld, f, B, HW, C);
}
hipError_t launch_gap_bwd(const float* gf, bf16_t* dx, int ld, int B, int HW, int C, const bf16_t* mask, int mask_ld,
                          hipStream_t s) {
  LAUNCH(gap_bwd_kernel, (size_t)B * HW * C, gf, dx, ld, B, HW, C, mask, mask_ld);
}
hipError_t launch_energy(const float* f, const float* Pc, const float* Pg, const int* targets, int B, int D, int K, float gs,
                         float ls, int use_c, int use_g, int normalize, float weight, const float* sample_w, float* score_out,
                         float* image_scores, float* gf, hipStream_t s) {
  if (K > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(energy_kernel, dim3(1), dim3(256), 0, s, f, Pc, Pg, targets, B, D, K, gs, ls, use_c, use_g, normalize, weight,
                     sample_w, score_out, image_scores, gf);
  return hipGetLastError();
}
hipError_t launch_affine(const float* z, const float* e, const float* b, float* out, int BC, int HW, hipStream_t s) {
  LAUNCH(affine_kernel, (size_t)BC * HW, z, e, b, out, BC, HW);
}
hipError_t launch_transform_update(const float* z, const float* g, const float* e, const float* b, float* z_out, int BC, int HW,
                                   float rho, float c, hipStream_t s) {
  hipLaunchKernelGGL(transform_update_kernel, dim3(BC), dim3(256), 0, s, z, g, e, b, z_out, HW, rho, c);
  return hipGetLastError();
}
hipError_t launch_sub_scaled(const float* a, const float* g, float* out, size_t n, float rho, hipStream_t s) {
  LAUNCH(sub_scaled_kernel, n, a, g, out, n, rho);
}
hipError_t launch_f32_to_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t s) { LAUNCH(f32_to_bf16_kernel, n, src, dst, n); }
hipError_t launch_fill_f32(float* dst, float v, size_t n, hipStream_t s) { LAUNCH(fill_kernel, n, dst, v, n); }
hipError_t launch_sinusoid_f32(const float* x, float* out, int n, int dim, int ld, int off, int flip, hipStream_t s) {
  LAUNCH(sinusoid_kernel, (size_t)n * (dim / 2), x, out, n, dim, ld, off, flip);
}
hipError_t launch_add_outer_f32(const float* a, const float* v, float* out, int ns, int nb, int cols, hipStream_t s) {
  LAUNCH(add_outer_kernel, (size_t)ns * nb * cols, a, v, out, ns, nb, cols);
}
hipError_t launch_add_rowvec_f32(float* y, const float* v, int rows, int cols, hipStream_t s) {
  LAUNCH(add_rowvec_kernel, (size_t)rows * cols, y, v, rows, cols);
}
hipError_t launch_to_uint8(const float* nchw, uint8_t* hwc, int B, int C, int H, int W, hipStream_t s) {
  LAUNCH(to_uint8_kernel, (size_t)B * H * W * C, nchw, hwc, B, C, H * W);
}
hipError_t launch_linear_f32(const float* x, const float* W, const float* b, float* y, int M, int N, int K, int silu_in, hipStream_t s) {
  const size_t waves = (size_t)M * N;
  hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, W, b, y, M, N, K, silu_in);
  return hipGetLastError();
}
hipError_t launch_clip_embed(const int* ids, const float* tok, const float* pos, bf16_t* out, int ld, int rows, int T, int C, int vocab,
                            hipStream_t s) {
  LAUNCH(clip_embed_kernel, (size_t)rows * C, ids, tok, pos, out, ld, rows, T, C, vocab);
}
hipError_t launch_act_bf16(const bf16_t* x, int ldx, bf16_t* y, int ldy, int M, int C, int kind, hipStream_t s) {
  LAUNCH(act_bf16_kernel, (size_t)M * C, x, ldx, y, ldy, M, C, kind);
}
hipError_t launch_act_bwd_bf16(const bf16_t* x, int ldx, const bf16_t* dy, int ldd, bf16_t* dx, int ldo, int M, int C, int kind, int accumulate,
                               hipStream_t s) {
  LAUNCH(act_bwd_bf16_kernel, (size_t)M * C, x, ldx, dy, ldd, dx, ldo, M, C, kind, accumulate);
}
hipError_t launch_patchify(const float* img, int ld, bf16_t* out, int B, int S, int p, int C, hipStream_t s) {
  if (S % p) return hipErrorInvalidValue;
  LAUNCH(patchify_kernel, (size_t)B * S * S * C, img, ld, out, B, S, p, C);
}
hipError_t launch_patchify_bwd(const bf16_t* gout, float* gimg, int ld, int B, int S, int p, int C, hipStream_t s) {
  if (S % p) return hipErrorInvalidValue;
  LAUNCH(patchify_bwd_kernel, (size_t)B * S * S * C, gout, gimg, ld, B, S, p, C);
}
hipError_t launch_vit_embed(const bf16_t* patches, int ldp, const float* cls, const float* pos, bf16_t* out, int ldo, int B, int np, int W,
                            hipStream_t s) {
  LAUNCH(vit_embed_kernel, (size_t)B * (np + 1) * W, patches, ldp, cls, pos, out, ldo, B, np, W);
}
hipError_t launch_vit_embed_bwd(const bf16_t* gout, int ldo, bf16_t* gp, int ldp, int B, int np, int W, hipStream_t s) {
  LAUNCH(vit_embed_bwd_kernel, (size_t)B * np * W, gout, ldo, gp, ldp, B, np, W);
}
hipError_t launch_select_rows(const bf16_t* x, int ldx, bf16_t* y, int ldy, int B, int stride, int C, hipStream_t s) {
  LAUNCH(select_rows_kernel, (size_t)B * C, x, ldx, y, ldy, B, stride, C);
}
hipError_t launch_select_rows_bwd(const bf16_t* dy, int ldy, bf16_t* dx, int ldx, int B, int stride, int C, int accumulate, hipStream_t s) {
  LAUNCH(select_rows_bwd_kernel, (size_t)B * stride * C, dy, ldy, dx, ldx, B, stride, C, accumulate);
}
hipError_t launch_vae_sample(const float* moments, int ld, const float* noise, float* latents, float* moments_out, int B, int C, int HW,
                             float scale, hipStream_t s) {
  LAUNCH(vae_sample_kernel, (size_t)B * C * HW, moments, ld, noise, latents, moments_out, B, C, HW, scale);
}
hipError_t launch_clip_pool_project(const int* ids, const bf16_t* x, int ld, const float* W, float* out, int n, int T, int C, int Pd, hipStream_t s) {
  hipLaunchKernelGGL(clip_pool_project_kernel, dim3((Pd + 3) / 4, n), dim3(256), 0, s, ids, x, ld, W, out, T, C, Pd);
  return hipGetLastError();
}
hipError_t launch_rows_bf16_to_f32(const bf16_t* x, int ld, float* y, int M, int C, hipStream_t s) {
  LAUNCH(rows_bf16_to_f32_kernel, (size_t)M * C, x, ld, y, M, C);
}
