// The guide network in exact fp32 (gfx950): implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 plus the fp32 forms of the
// guide's side kernels (max-pool, GAP, bicubic 224 resize and their transposes, ReLU masks, fan-in adds).
//
// Why fp32: image_encoder.encode_image (model_utils.py:29-41) is a ReLU / max-pool network, and the energy gradient
// torch.autograd.grad(E, [e, b]) / (E, z) (generate_data.py:721, :761) goes through its ReLU masks.  The gradient of such a
// network is piecewise constant in the input: a forward rounding of 2^-9 (bf16) flips the masks of every activation within that
// distance of zero and moves the input-gradient by 20-30 % (measured against the fp32 oracle, tests/test_guide_f32_gpu.py), while
// the whole guide is < 0.1 % of the FLOPs of an expansion.  So the guide forward, its masks and its VJP run in fp32:
// v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain (1/16 of the bf16 MFMA rate, 157 TFLOP/s peak).
//
// Layout: activations NHWC fp32 rows [pixels, ld] (ld % 4 == 0), weights packed [N][K] fp32 with k = (tap, cin), cin padded to
// a multiple of 4, K padded to 16; the same tap table / stride / dilated-gather conventions as conv_gemm.hip, so the dgrad of a
// stride-2 convolution is the same kernel on the transposed + flipped packing.  Grouped convolutions (ResNeXt, model_utils.py:56-63)
// use the dense block-diagonal packing and skip the K-steps whose channels lie outside the groups of the workgroup's output columns.
// Tile 64 x 64 x 16 per 256-thread workgroup, one 32x32 accumulator per wave, LDS K-major ([k][row], row stride 68 floats:
// staging writes and fragment reads are both bank-conflict free), register-staged prefetch of the next K-step under the MFMAs.
// The MFMA is issued swapped (A = weights, B = pixels) so a lane owns 4 consecutive output channels of one pixel (float4 epilogue).
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define GRID_STRIDE(i, n) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); i += (size_t)gridDim.x * blockDim.x)

inline int nblocks(size_t n, int threads = 256, int cap = 8192) {
  size_t b = (n + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}

__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF32Params p) {
  constexpr int BM = 64, BN = 64, BK = 16, LDT = 68;
  __shared__ float As[2][BK][LDT];
  __shared__ float Ws[2][BK][LDT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;

  // K-step list. cin % 16 == 0: a K-step lies inside one tap and the channel range may be restricted to the groups of this
  // workgroup's output columns [c_lo, c_hi); otherwise K is walked linearly and the tap is found per 4-channel vector.
  const int cin = p.cin;
  const bool tapwise = (cin & 15) == 0;
  int c_lo = 0, c_hi = cin;
  if (p.groups > 1) {
    const int g_lo = n0 / p.cpg_out, g_hi = min(p.N - 1, n0 + BN - 1) / p.cpg_out;
    c_lo = (g_lo * p.cpg_in) & ~15;
    c_hi = min(cin, ((g_hi + 1) * p.cpg_in + 15) & ~15);
  }
  const int spt = tapwise ? (c_hi - c_lo) >> 4 : 1;
  const int nsteps = tapwise ? p.ntaps * spt : p.K >> 4;

  // staging assignment: one float4 of A and one of W per thread per K-step
  const int r = tid >> 2, kv = tid & 3;
  int pixb, iy0, ix0;
  {
    const int m = m0 + r;
    if (m < p.M) {
      const int HoWo = p.Ho * p.Wo;
      const int b = m / HoWo, rem = m - b * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      pixb = b * p.H * p.W; iy0 = oy * p.stride; ix0 = ox * p.stride;
    } else { pixb = 0; iy0 = -1000000; ix0 = 0; }
  }
  const int wrow = n0 + r;
  const bool wok = wrow < p.N;
  const float* wbase = p.w + (size_t)(wok ? wrow : 0) * p.K;
  const int shift = p.shift, parity = p.parity;

  float4 ra, rw;
  auto load_tile = [&](int kt) {
    int tap, c, wk;
    bool ev = true;
    if (tapwise) {
      tap = kt / spt;
      c = c_lo + ((kt - tap * spt) << 4) + kv * 4;
      wk = tap * cin + c;
    } else {
      wk = (kt << 4) + kv * 4;
      tap = wk / cin;
      c = wk - tap * cin;
      ev = tap < p.ntaps;
    }
    const int e = p.taptab[ev ? tap : 0];
    const int dx = (e & 63) - 32, dy = ((e >> 6) & 63) - 32;
    const int ly = iy0 + dy, lx = ix0 + dx;
    const int sy = ly >> shift, sx = lx >> shift;
    bool ok = ev && ly >= 0 && lx >= 0 && sy < p.H && sx < p.W;
    if (parity) ok = ok && (((ly | lx) & 1) == 0);
    ra = ok ? *(const float4*)(p.x + (size_t)(pixb + sy * p.W + sx) * p.x_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    rw = wok ? *(const float4*)(wbase + wk) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store_tile = [&](int buf) {
    As[buf][kv * 4 + 0][r] = ra.x; As[buf][kv * 4 + 1][r] = ra.y; As[buf][kv * 4 + 2][r] = ra.z; As[buf][kv * 4 + 3][r] = ra.w;
    Ws[buf][kv * 4 + 0][r] = rw.x; Ws[buf][kv * 4 + 1][r] = rw.y; Ws[buf][kv * 4 + 2][r] = rw.z; Ws[buf][kv * 4 + 3][r] = rw.w;
  };

  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  const int fr = lane & 31, fh = lane >> 5;

  if (nsteps > 0) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nsteps; ++kt) {
      const int cur = kt & 1;
      const bool more = kt + 1 < nsteps;
      if (more) load_tile(kt + 1);
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const float a = Ws[cur][kk * 2 + fh][wn * 32 + fr];
        const float b = As[cur][kk * 2 + fh][wm * 32 + fr];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // epilogue: acc[v] = out[m = m0 + wm*32 + fr][n = n0 + wn*32 + (v>>2)*8 + fh*4 + (v&3)]
  const int m = m0 + wm * 32 + fr;
  if (m >= p.M) return;
  const int flags = p.flags;
#pragma unroll
  for (int vg = 0; vg < 4; ++vg) {
    const int nb = n0 + wn * 32 + vg * 8 + fh * 4;
    if (nb >= p.N) continue;
    float h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = acc[vg * 4 + q];
    const bool full = nb + 4 <= p.N;
    if (flags & CF_BIAS) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (nb + q < p.N) h[q] += p.bias[nb + q];
    }
    if (flags & CF_RES) {
      const float* rp = p.res + (size_t)m * p.res_ld + nb;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (nb + q < p.N) h[q] += rp[q];
    }
    if (flags & CF_RELU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) h[q] = fmaxf(h[q], 0.f);
    }
    if (flags & CF_RELU6) {
#pragma unroll
      for (int q = 0; q < 4; ++q) h[q] = fminf(fmaxf(h[q], 0.f), 6.f);
    }
    if (flags & CF_MASK) {
      const float* mp = p.mask + (size_t)m * p.mask_ld + nb;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (nb + q < p.N && !(mp[q] > 0.f)) h[q] = 0.f;
    }
    float* yp = p.y + (size_t)m * p.y_ld + nb;
    if (full && !(p.y_ld & 3)) {
      *(float4*)yp = make_float4(h[0], h[1], h[2], h[3]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (nb + q < p.N) yp[q] = h[q];
    }
  }
}

// Narrow outputs (N <= 4: the input gradient of the guide's first convolution -- 3 image channels from 49 taps x 64 channels -- would fill
// 3 of the 64 columns of an MFMA tile): one thread per output pixel, the weights of all N columns through the scalar cache, the SAME
// k-ordered fmaf chain as the MFMA kernel.  A tap that the gather rejects (outside the image, wrong parity of a dilated gather)
// contributes exact zeros there and is skipped here, so the results are bit-identical (tests/test_guide_f32_gpu.py).
__global__ __launch_bounds__(256) void conv_f32_narrow_kernel(ConvF32Params p) {
  const int tid = threadIdx.x;
  const float* __restrict__ wsm = p.w;                // [N][K]: wave-uniform addresses -> scalar loads, the weights are SGPR operands of the fmas
  const int m = blockIdx.x * 256 + tid;
  if (m >= p.M) return;
  const int HoWo = p.Ho * p.Wo;
  const int b = m / HoWo, rem = m - b * HoWo;
  const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
  const int pixb = b * p.H * p.W, iy0 = oy * p.stride, ix0 = ox * p.stride;
  const int shift = p.shift, cin = p.cin, K = p.K, N = p.N;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int tap = 0; tap < p.ntaps; ++tap) {
    const int e = p.taptab[tap];
    const int dx = (e & 63) - 32, dy = ((e >> 6) & 63) - 32;
    const int ly = iy0 + dy, lx = ix0 + dx;
    const int sy = ly >> shift, sx = lx >> shift;
    bool ok = ly >= 0 && lx >= 0 && sy < p.H && sx < p.W;
    if (p.parity) ok = ok && (((ly | lx) & 1) == 0);
    if (!ok) continue;
    const float* xp = p.x + (size_t)(pixb + sy * p.W + sx) * p.x_ld;
    const float* wp = wsm + tap * cin;
    for (int c = 0; c < cin; c += 4) {
      const float4 xv = *(const float4*)(xp + c);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (n < N) {
          const float4 wv = *(const float4*)(wp + n * K + c);
          acc[n] = __builtin_fmaf(wv.x, xv.x, acc[n]);
          acc[n] = __builtin_fmaf(wv.y, xv.y, acc[n]);
          acc[n] = __builtin_fmaf(wv.z, xv.z, acc[n]);
          acc[n] = __builtin_fmaf(wv.w, xv.w, acc[n]);
        }
      }
    }
  }
  const int flags = p.flags;
  for (int n = 0; n < N; ++n) {
    float h = acc[n];
    if (flags & CF_BIAS) h += p.bias[n];
    if (flags & CF_RES) h += p.res[(size_t)m * p.res_ld + n];
    if (flags & CF_RELU) h = fmaxf(h, 0.f);
    if (flags & CF_RELU6) h = fminf(fmaxf(h, 0.f), 6.f);
    if ((flags & CF_MASK) && !(p.mask[(size_t)m * p.mask_ld + n] > 0.f)) h = 0.f;
    p.y[(size_t)m * p.y_ld + n] = h;
  }
}

// ---- side kernels, fp32 rows with C % 4 == 0 -----------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }

// y = a (+ b)
__global__ void add_f32_kernel(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int M, int C) {
  const int VC = C >> 2;
  GRID_STRIDE(i, (size_t)M * VC) {
    const int vc = (int)(i % VC);
    const size_t m = i / VC;
    float4 x = ld4(a + m * lda + vc * 4);
    if (b) { const float4 z = ld4(b + m * ldb + vc * 4); x.x += z.x; x.y += z.y; x.z += z.z; x.w += z.w; }
    st4(y + m * ldy + vc * 4, x);
  }
}
// y = dy * (mask > 0 && mask < hi)   (hi = +inf: ReLU; 6: ReLU6 = hardtanh(0, 6), whose gradient is 1 strictly inside the interval)
__global__ void mask_f32_kernel(const float* dy, int ldd, const float* mask, int ldm, float* y, int ldy, int M, int C, float hi) {
  const int VC = C >> 2;
  GRID_STRIDE(i, (size_t)M * VC) {
    const int vc = (int)(i % VC);
    const size_t m = i / VC;
    float4 x = ld4(dy + m * ldd + vc * 4);
    const float4 z = ld4(mask + m * ldm + vc * 4);
    x.x = (z.x > 0.f && z.x < hi) ? x.x : 0.f; x.y = (z.y > 0.f && z.y < hi) ? x.y : 0.f;
    x.z = (z.z > 0.f && z.z < hi) ? x.z : 0.f; x.w = (z.w > 0.f && z.w < hi) ? x.w : 0.f;
    st4(y + m * ldy + vc * 4, x);
  }
}

__global__ void maxpool_f32_kernel(const float* x, float* y, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, VC = C >> 2;
  GRID_STRIDE(i, (size_t)B * Ho * Wo * VC) {
    const int vc = (int)(i % VC);
    const size_t pix = i / VC;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
        if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
        const float4 v = ld4(x + (((size_t)b * H + iy) * W + ix) * C + vc * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    st4(y + pix * C + vc * 4, m);
  }
}

// gradient goes to the first maximum of each window in (ky, kx) scan order (torch max_pool2d); gather form, no atomics
__global__ void maxpool_bwd_f32_kernel(const float* x, const float* dy, float* dx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, VC = C >> 2;
  GRID_STRIDE(i, (size_t)B * H * W * VC) {
    const int vc = (int)(i % VC);
    const size_t pix = i / VC;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
    float g[4] = {0, 0, 0, 0};
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        float m[4]; int am[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { m[e] = -INFINITY; am[e] = -1; }
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) {
            const int yy = oy * 2 + ky - 1, xx = ox * 2 + kx - 1;
            if (yy < 0 || xx < 0 || yy >= H || xx >= W) continue;
            const float4 t = ld4(x + (((size_t)b * H + yy) * W + xx) * C + vc * 4);
            const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (v[e] > m[e]) { m[e] = v[e]; am[e] = yy * W + xx; }
          }
        const float4 t = ld4(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + vc * 4);
        const float d[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (am[e] == iy * W + ix) g[e] += d[e];
      }
    }
    st4(dx + pix * C + vc * 4, make_float4(g[0], g[1], g[2], g[3]));
  }
}

__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
  const float A = -0.75f;
  w[0] = cc2(t + 1.f, A); w[1] = cc1(t, A); w[2] = cc1(1.f - t, A); w[3] = cc2(2.f - t, A);
}

// F.interpolate(img, (Hd, Wd), mode='bicubic') (generate_data.py:704, :745): A = -0.75, align_corners=False, no antialias,
// border-clamped taps.  src fp32 NHWC [B,Hs,Ws,ld_s] -> dst fp32 [B,Hd,Wd,ld_d], channels >= C zero-filled up to Cpad.
__global__ void bicubic_f32_kernel(const float* src, int ld_s, float* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C,
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
          float rr = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int xx = min(max(ix - 1 + q, 0), Ws - 1);
            rr += wx[q] * src[(((size_t)b * Hs + yy) * Ws + xx) * ld_s + c];
          }
          acc += wy[a] * rr;
        }
      }
      dst[i * ld_d + c] = acc;
    }
  }
}

// transpose of the above as a gather over the destination pixels whose (border-clamped) 4x4 footprints hit this source pixel:
// deterministic, no atomics.  ddst fp32 -> dsrc bf16 rows (the VAE decoder's gradient slab) or fp32.
template <bool OUT_BF16>
__global__ void bicubic_bwd_f32_kernel(const float* ddst, int ld_d, void* dsrc, int ld_s, int B, int Hs, int Ws, int Hd, int Wd, int C) {
  const float sh = (float)Hs / (float)Hd, sw = (float)Ws / (float)Wd;
  GRID_STRIDE(i, (size_t)B * Hs * Ws) {
    const int sx = (int)(i % Ws), sy = (int)((i / Ws) % Hs), b = (int)(i / ((size_t)Ws * Hs));
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
        const float* d = ddst + (((size_t)b * Hd + oy) * Wd + ox) * ld_d;
        for (int c = 0; c < C && c < 4; ++c) acc[c] += wys * wxs * d[c];
      }
    }
    for (int c = 0; c < C && c < 4; ++c) {
      if (OUT_BF16) ((bf16_t*)dsrc)[i * ld_s + c] = f2bf(acc[c]);
      else ((float*)dsrc)[i * ld_s + c] = acc[c];
    }
  }
}

// encode_image's pooling (model_utils.py:31-37): 'avg' = AdaptiveAvgPool2d(1), 'max' = AdaptiveMaxPool2d(1); also records the
// argmax pixel of the max form for its VJP (first maximum in scan order, like torch)
__global__ void gap_f32_kernel(const float* x, int ld, float* f, int* arg, int B, int HW, int C, int use_max) {
  GRID_STRIDE(i, (size_t)B * C) {
    const int c = (int)(i % C), b = (int)(i / C);
    if (use_max) {
      float m = -INFINITY; int am = 0;
      for (int pxl = 0; pxl < HW; ++pxl) { const float v = x[((size_t)b * HW + pxl) * ld + c]; if (v > m) { m = v; am = pxl; } }
      f[i] = m;
      if (arg) arg[i] = am;
    } else {
      float s = 0.f;
      for (int pxl = 0; pxl < HW; ++pxl) s += x[((size_t)b * HW + pxl) * ld + c];
      f[i] = s / HW;
    }
  }
}
__global__ void gap_bwd_f32_kernel(const float* gf, float* dx, int ld, int B, int HW, int C, const int* arg) {
  GRID_STRIDE(i, (size_t)B * HW * C) {
    const int c = (int)(i % C);
    const size_t row = i / C;
    const int b = (int)(row / HW), pxl = (int)(row % HW);
    const float g = gf[(size_t)b * C + c];
    dx[row * ld + c] = arg ? (arg[(size_t)b * C + c] == pxl ? g : 0.f) : g / HW;
  }
}

// NCHW fp32 [B,C,H,W] -> NHWC fp32 rows [B*H*W, ld] (channels >= C zero-filled up to Cpad) and back
__global__ void nchw_to_nhwc_f32_kernel(const float* src, float* dst, int B, int C, int HW, int Cpad, int ld) {
  GRID_STRIDE(i, (size_t)B * HW * Cpad) {
    const int c = (int)(i % Cpad);
    const size_t row = i / Cpad;
    const int pix = (int)(row % HW), b = (int)(row / HW);
    dst[row * ld + c] = c < C ? src[((size_t)b * C + c) * HW + pix] : 0.f;
  }
}

}  // namespace

#define LAUNCH(kern, n, ...) hipLaunchKernelGGL(kern, dim3(nblocks(n)), dim3(256), 0, s, __VA_ARGS__); return hipGetLastError()

hipError_t launch_conv_f32(const ConvF32Params& p, hipStream_t s) {
  if ((p.K & 15) || (p.cin & 3) || (p.x_ld & 3)) return hipErrorInvalidValue;
  if (p.groups > 1 && ((p.cin & 15) || p.cpg_in < 1 || p.cpg_out < 1)) return hipErrorInvalidValue;
  if ((size_t)p.B * p.H * p.W * (size_t)p.x_ld >= 0x7FFF0000ull) return hipErrorInvalidValue;
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  static const int narrow = getenv("DD_F32_NARROW") ? atoi(getenv("DD_F32_NARROW")) : 1;
  if (narrow && p.N <= 4 && p.groups <= 1 && (size_t)p.N * p.K * 4 <= 65536 && p.M >= 4096) {
    hipLaunchKernelGGL(conv_f32_narrow_kernel, dim3((p.M + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
  }
  const int ntm = (p.M + 63) / 64, ntn = (p.N + 63) / 64;
  hipLaunchKernelGGL(conv_f32_kernel, dim3(ntm * ntn), dim3(256), 0, s, p);
  return hipGetLastError();
}
hipError_t launch_add_f32(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int M, int C, hipStream_t s) {
  LAUNCH(add_f32_kernel, (size_t)M * (C / 4), a, lda, b, ldb, y, ldy, M, C);
}
hipError_t launch_copy_f32(const float* a, int lda, float* y, int ldy, int M, int C, hipStream_t s) {
  LAUNCH(add_f32_kernel, (size_t)M * (C / 4), a, lda, (const float*)nullptr, 0, y, ldy, M, C);
}
hipError_t launch_mask_f32(const float* dy, int ldd, const float* mask, int ldm, float* y, int ldy, int M, int C, float hi, hipStream_t s) {
  LAUNCH(mask_f32_kernel, (size_t)M * (C / 4), dy, ldd, mask, ldm, y, ldy, M, C, hi > 0.f ? hi : INFINITY);
}
hipError_t launch_maxpool3x3s2_f32(const float* x, float* y, int B, int H, int W, int C, hipStream_t s) {
  LAUNCH(maxpool_f32_kernel, (size_t)B * (H / 2) * (W / 2) * (C / 4), x, y, B, H, W, C);
}
hipError_t launch_maxpool3x3s2_bwd_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, hipStream_t s) {
  LAUNCH(maxpool_bwd_f32_kernel, (size_t)B * H * W * (C / 4), x, dy, dx, B, H, W, C);
}
hipError_t launch_bicubic_f32(const float* src, int ld_s, float* dst, int ld_d, int B, int Hs, int Ws, int Hd, int Wd, int C, int Cpad,
                              hipStream_t s) {
  LAUNCH(bicubic_f32_kernel, (size_t)B * Hd * Wd, src, ld_s, dst, ld_d, B, Hs, Ws, Hd, Wd, C, Cpad);
}
hipError_t launch_bicubic_bwd_f32(const float* ddst, int ld_d, void* dsrc, int dsrc_bf16, int ld_s, int B, int Hs, int Ws, int Hd, int Wd,
                                  int C, hipStream_t s) {
  if (C > 4) return hipErrorInvalidValue;
  if (dsrc_bf16) { LAUNCH(bicubic_bwd_f32_kernel<true>, (size_t)B * Hs * Ws, ddst, ld_d, dsrc, ld_s, B, Hs, Ws, Hd, Wd, C); }
  LAUNCH(bicubic_bwd_f32_kernel<false>, (size_t)B * Hs * Ws, ddst, ld_d, dsrc, ld_s, B, Hs, Ws, Hd, Wd, C);
}
hipError_t launch_gap_f32(const float* x, int ld, float* f, int* argmax, int B, int HW, int C, int use_max, hipStream_t s) {
  LAUNCH(gap_f32_kernel, (size_t)B * C, x, ld, f, argmax, B, HW, C, use_max);
}
hipError_t launch_gap_bwd_f32(const float* gf, float* dx, int ld, int B, int HW, int C, const int* argmax, hipStream_t s) {
  LAUNCH(gap_bwd_f32_kernel, (size_t)B * HW * C, gf, dx, ld, B, HW, C, argmax);
}
hipError_t launch_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cpad, int ld, hipStream_t s) {
  LAUNCH(nchw_to_nhwc_f32_kernel, (size_t)B * H * W * Cpad, src, dst, B, C, H * W, Cpad, ld);
}
