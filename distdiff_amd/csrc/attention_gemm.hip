// K5 (wide heads, d >= 256) — attention of the AutoencoderKL mid block (1 head, d = 512, 4096 tokens; reference call sites
// generate_data.py:701/743 vae.decode, dataloader.py:808 vae.encode, and their autograd backward :721/:761) through the
// implicit-GEMM kernel instead of the flash kernels.
//
// Why: with d = 512 a flash workgroup can only keep 16-64 queries (or 16 keys) resident, so every workgroup streams the whole K/V
// (or Q/dO) panel again: the d = 512 flash kernels run at 70-150 TFLOP/s, L2-stream-bound.  Here the N x N score matrix of one image
// is materialised (fp32 scores, bf16 probabilities: 64 + 32 MB at N = 4096, reused image after image) and every product is a plain
// [N x N x d] GEMM on the conv kernel (900+ TFLOP/s):
//   forward : S = Q K^T (fp32, pre-scaled) -> row softmax -> P (bf16), lse -> O = P V
//   backward: S, dP = dO V^T (fp32) -> P = exp(S - lse), dS = P (dP - delta) scale, written row-major and transposed (bf16)
//             -> dV = P^T dO, dK = dS^T Q, dQ = dS K
// The GEMM wants its second operand as [N][K] rows with K contiguous, so K, V are copied out of the fused qkv rows and the operands
// that enter "transposed" (V for O, Q / dO / K for the gradients) go through a tiled transpose first (4 MB each).
// Same numerics as the flash path: bf16 P and dS feed the MFMAs, fp32 accumulation, lse in the natural-log domain of the scaled scores.
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "kernels.h"

namespace {

// dst[c][r] = src[r][c]  (bf16), 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int R, int C, size_t src_img, size_t dst_img) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  src += blockIdx.z * src_img; dst += blockIdx.z * dst_img;      // grid.z = image of the group
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? src[(size_t)(r0 + r) * lds_ + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && r0 + r < R) dst[(size_t)(c0 + c) * ldd + r0 + r] = tile[r][c];
  }
}

// one workgroup per score row (the rows of every image of the group): P[row, :] = softmax(S[row, :]) (S already scaled), lse[row] = log sum exp.
// Rows of up to 4096 scores stay in registers (16 per thread): ONE read of the fp32 scores instead of three.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* S, int Nk, bf16_t* P, float* lse) {
  __shared__ float red[8];
  const size_t row = blockIdx.x;
  const float* s = S + row * Nk;
  const bool in_regs = Nk <= 4096;
  float4 v[4];
  float m = -INFINITY;
  if (in_regs) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = c * 1024 + threadIdx.x * 4;
      v[c] = i < Nk ? *(const float4*)(s + i) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      m = fmaxf(fmaxf(m, fmaxf(v[c].x, v[c].y)), fmaxf(v[c].z, v[c].w));
    }
  } else {
    for (int i = threadIdx.x * 4; i < Nk; i += 1024) {
      const float4 w = *(const float4*)(s + i);
      m = fmaxf(fmaxf(m, fmaxf(w.x, w.y)), fmaxf(w.z, w.w));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float l = 0.f;
  if (in_regs) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      v[c].x = __expf(v[c].x - m); v[c].y = __expf(v[c].y - m); v[c].z = __expf(v[c].z - m); v[c].w = __expf(v[c].w - m);
      l += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    }
  } else {
    for (int i = threadIdx.x * 4; i < Nk; i += 1024) {
      const float4 w = *(const float4*)(s + i);
      l += (__expf(w.x - m) + __expf(w.y - m)) + (__expf(w.z - m) + __expf(w.w - m));
    }
  }
  l = wave_sum(l);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = l;
  __syncthreads();
  l = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.f / l;
  if (in_regs) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = c * 1024 + threadIdx.x * 4;
      if (i < Nk) *(uint2*)(P + row * Nk + i) = make_uint2(pack2bf(v[c].x * inv, v[c].y * inv), pack2bf(v[c].z * inv, v[c].w * inv));
    }
  } else {
    for (int i = threadIdx.x * 4; i < Nk; i += 1024) {
      const float4 w = *(const float4*)(s + i);
      *(uint2*)(P + row * Nk + i) = make_uint2(pack2bf(__expf(w.x - m) * inv, __expf(w.y - m) * inv), pack2bf(__expf(w.z - m) * inv, __expf(w.w - m) * inv));
    }
  }
  if (threadIdx.x == 0) lse[row] = m + __logf(l);
}

// 64 x 64 tile of the score matrix: P = exp(S - lse[q]), dS = P (dP - delta[q]) scale; row-major and transposed copies (bf16)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* S, const float* dP, const float* lse, const float* delta, float scale,
                                                          int Nq, int Nk, bf16_t* P, bf16_t* dS, bf16_t* PT, bf16_t* dST) {
  __shared__ bf16_t tp[64][66], td[64][66];
  const int q0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  {  // grid.z = image of the group: every matrix is [image][Nq][Nk] (or [image][Nk][Nq]), lse / delta [image][Nq]
    const size_t im = (size_t)blockIdx.z * Nq * Nk;
    S += im; dP += im; P += im; dS += im; PT += im; dST += im; lse += (size_t)blockIdx.z * Nq; delta += (size_t)blockIdx.z * Nq;
  }
  for (int i = threadIdx.x; i < 64 * 16; i += 256) {
    const int r = i >> 4, c = (i & 15) * 4;
    const size_t off = (size_t)(q0 + r) * Nk + k0 + c;
    const float4 s = *(const float4*)(S + off), d = *(const float4*)(dP + off);
    const float l = lse[q0 + r], dl = delta[q0 + r];
    const float p0 = __expf(s.x - l), p1 = __expf(s.y - l), p2 = __expf(s.z - l), p3 = __expf(s.w - l);
    const float g0 = p0 * (d.x - dl) * scale, g1 = p1 * (d.y - dl) * scale, g2 = p2 * (d.z - dl) * scale, g3 = p3 * (d.w - dl) * scale;
    uint2 up, ug;
    up.x = pack2bf(p0, p1); up.y = pack2bf(p2, p3);
    ug.x = pack2bf(g0, g1); ug.y = pack2bf(g2, g3);
    *(uint2*)(P + off) = up;
    *(uint2*)(dS + off) = ug;
    tp[r][c] = (bf16_t)(up.x & 0xffff); tp[r][c + 1] = (bf16_t)(up.x >> 16); tp[r][c + 2] = (bf16_t)(up.y & 0xffff); tp[r][c + 3] = (bf16_t)(up.y >> 16);
    td[r][c] = (bf16_t)(ug.x & 0xffff); td[r][c + 1] = (bf16_t)(ug.x >> 16); td[r][c + 2] = (bf16_t)(ug.y & 0xffff); td[r][c + 3] = (bf16_t)(ug.y >> 16);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    const size_t off = (size_t)(k0 + c) * Nq + q0 + r;
    PT[off] = tp[r][c];
    dST[off] = td[r][c];
  }
}

// out[g * rows + m, :] = A[g * rows + m, :] W_g^T for the `groups` matrices W_g = W + g * N * K stacked along M (groups == 1: a plain GEMM).
// One grouped launch when the persistent kernel takes the shape (whole tiles per group, enough tiles to fill the chip), else one per group.
hipError_t gemm(const bf16_t* A, int lda, const bf16_t* W, void* out, int ldo, int M, int N, int K, bool out_f32, float alpha,
                const int* tap1x1, float* partial, size_t partial_cap, hipStream_t s, int groups = 1) {
  ConvGemmParams p;
  memset(&p, 0, sizeof p);
  p.x = A; p.x_ld = lda; p.w = W; p.taptab = tap1x1; p.y = out; p.y_ld = ldo;
  p.B = 1; p.W = 1; p.Wo = 1; p.stride = 1;
  p.cin = K; p.ntaps = 1; p.N = N; p.K = K;
  p.flags = out_f32 ? CF_OUT_F32 : 0;
  p.alpha = alpha;
  p.partial = partial;
  if (groups > 1) {
    p.wgroup_rows = M; p.wgroup_elems = (long long)N * K;
    p.H = p.Ho = p.M = M * groups;
    const hipError_t e = launch_conv_gemm(p, partial_cap, s);
    if (e != hipErrorInvalidValue) return e;
    p.wgroup_rows = 0; p.wgroup_elems = 0;
  }
  p.H = p.Ho = p.M = M;
  for (int g = 0; g < groups; ++g) {
    p.x = A + (size_t)g * M * lda;
    p.w = W + (size_t)g * N * K;
    p.y = (char*)out + (size_t)g * M * ldo * (out_f32 ? 4 : 2);
    const hipError_t e = launch_conv_gemm(p, partial_cap, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// dst_i[c][r] = src_i[r][c] for the `images` matrices src + i * src_img -> dst + i * dst_img
hipError_t transpose(const bf16_t* src, int lds_, bf16_t* dst, int ldd, int R, int C, hipStream_t s, int images = 1, size_t src_img = 0, size_t dst_img = 0) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64, images), dim3(256), 0, s, src, lds_, dst, ldd, R, C, src_img, dst_img);
  return hipGetLastError();
}

struct Ws {   // carve-out of the caller's scratch (256-byte aligned pieces)
  char* p;
  template <class T> T* take(size_t count) { T* r = (T*)p; p += (count * sizeof(T) + 255) / 256 * 256; return r; }
};

}  // namespace

#define AG_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

bool attention_gemm_supported(const AttnParams& p) {
  return p.D >= 256 && (p.D % 64) == 0 && (p.Nq % 64) == 0 && (p.Nk % 64) == 0 && !p.causal && !(p.ldq & 7) && !(p.ldk & 7) && !(p.ldv & 7) && !(p.ldo & 7);
}

// scratch of ONE image; the launchers take `workspace_bytes` and process as many images per launch as fit (at most 8)
size_t attention_gemm_workspace(int Nq, int Nk, int D, int bwd) {
  const size_t nn = (size_t)Nq * Nk;
  size_t b = nn * 4 + nn * 2 + 2 * (size_t)Nk * D * 2 + 4096;                       // S, P, Kc, V^T
  if (bwd) b = 2 * nn * 4 + 4 * nn * 2 + 3 * (size_t)Nk * D * 2 + 2 * (size_t)Nq * D * 2 + 8192;   // S, dP | P, dS, P^T, dS^T | Kc, Vc, K^T | Q^T, dO^T
  return b;
}
// images per launch: single-head layers only (with H > 1 the heads of one image interleave along the columns of the same rows, so the
// images of a group do not stack along M), whole 256-row tiles per image, as many as the scratch holds
static int group_size(const AttnParams& p, size_t workspace_bytes, int bwd) {
  static const int gmax = getenv("DD_ATTN_GEMM_GROUP") ? atoi(getenv("DD_ATTN_GEMM_GROUP")) : 8;
  if (p.H != 1 || (p.Nq & 255) || (p.Nk & 255) || gmax <= 1) return 1;
  const size_t per = attention_gemm_workspace(p.Nq, p.Nk, p.D, bwd);
  size_t g = workspace_bytes / per;
  if (g > (size_t)gmax) g = gmax;
  if (g > (size_t)p.B) g = p.B;
  return g < 1 ? 1 : (int)g;
}

hipError_t launch_attention_gemm_fwd(const AttnParams& p, void* workspace, size_t workspace_bytes, const int* tap1x1, float* partial, size_t partial_cap, hipStream_t s) {
  if (!attention_gemm_supported(p) || !p.lse || workspace_bytes < attention_gemm_workspace(p.Nq, p.Nk, p.D, 0)) return hipErrorInvalidValue;
  const int Nq = p.Nq, Nk = p.Nk, D = p.D;
  const int G = group_size(p, workspace_bytes, 0);
  Ws ws{(char*)workspace};
  float* S = ws.take<float>((size_t)G * Nq * Nk);
  bf16_t* P = ws.take<bf16_t>((size_t)G * Nq * Nk);
  bf16_t* Kc = ws.take<bf16_t>((size_t)G * Nk * D);
  bf16_t* VT = ws.take<bf16_t>((size_t)G * Nk * D);
  const int images = p.B * p.H;
  for (int i0 = 0; i0 < images; i0 += G) {
    const int g = images - i0 < G ? images - i0 : G;        // (G > 1 only with H == 1: image i = batch index i)
    const int b = i0 / p.H, h = i0 % p.H;
    const bf16_t* q = p.q + (size_t)b * Nq * p.ldq + h * D;
    const bf16_t* k = p.k + (size_t)b * Nk * p.ldk + h * D;
    const bf16_t* v = p.v + (size_t)b * Nk * p.ldv + h * D;
    AG_CHK(launch_copy_bf16(k, p.ldk, Kc, D, g * Nk, D, s));
    AG_CHK(transpose(v, p.ldv, VT, Nk, Nk, D, s, g, (size_t)Nk * p.ldv, (size_t)Nk * D));
    AG_CHK(gemm(q, p.ldq, Kc, S, Nk, Nq, Nk, D, true, p.scale, tap1x1, partial, partial_cap, s, g));
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(g * Nq), dim3(256), 0, s, (const float*)S, Nk, P, p.lse + ((size_t)b * p.H + h) * Nq);
    AG_CHK(hipGetLastError());
    AG_CHK(gemm(P, Nk, VT, p.o + (size_t)b * Nq * p.ldo + h * D, p.ldo, Nq, D, Nk, false, 1.f, tap1x1, partial, partial_cap, s, g));
  }
  return hipSuccess;
}

hipError_t launch_attention_gemm_bwd(const AttnParams& p, void* workspace, size_t workspace_bytes, const int* tap1x1, float* partial, size_t partial_cap, hipStream_t s) {
  if (!attention_gemm_supported(p) || !p.lse || !p.delta || !p.dq || !p.dk || !p.dv || (p.lddo & 7) || (p.lddq & 7) || (p.lddk & 7) || (p.lddv & 7) ||
      workspace_bytes < attention_gemm_workspace(p.Nq, p.Nk, p.D, 1))
    return hipErrorInvalidValue;
  const int Nq = p.Nq, Nk = p.Nk, D = p.D;
  AG_CHK(launch_attention_delta(p, s));
  const int G = group_size(p, workspace_bytes, 1);
  Ws ws{(char*)workspace};
  float* S = ws.take<float>((size_t)G * Nq * Nk);
  float* dP = ws.take<float>((size_t)G * Nq * Nk);
  bf16_t* P = ws.take<bf16_t>((size_t)G * Nq * Nk);
  bf16_t* dS = ws.take<bf16_t>((size_t)G * Nq * Nk);
  bf16_t* PT = ws.take<bf16_t>((size_t)G * Nq * Nk);
  bf16_t* dST = ws.take<bf16_t>((size_t)G * Nq * Nk);
  bf16_t* Kc = ws.take<bf16_t>((size_t)G * Nk * D);
  bf16_t* Vc = ws.take<bf16_t>((size_t)G * Nk * D);
  bf16_t* KT = ws.take<bf16_t>((size_t)G * Nk * D);
  bf16_t* QT = ws.take<bf16_t>((size_t)G * Nq * D);
  bf16_t* dOT = ws.take<bf16_t>((size_t)G * Nq * D);
  const int images = p.B * p.H;
  for (int i0 = 0; i0 < images; i0 += G) {
    const int g = images - i0 < G ? images - i0 : G;
    const int b = i0 / p.H, h = i0 % p.H;
    const bf16_t* q = p.q + (size_t)b * Nq * p.ldq + h * D;
    const bf16_t* k = p.k + (size_t)b * Nk * p.ldk + h * D;
    const bf16_t* v = p.v + (size_t)b * Nk * p.ldv + h * D;
    const bf16_t* d_o = p.d_o + (size_t)b * Nq * p.lddo + h * D;
    const float* lse = p.lse + ((size_t)b * p.H + h) * Nq;
    const float* delta = p.delta + ((size_t)b * p.H + h) * Nq;
    AG_CHK(launch_copy_bf16(k, p.ldk, Kc, D, g * Nk, D, s));
    AG_CHK(launch_copy_bf16(v, p.ldv, Vc, D, g * Nk, D, s));
    AG_CHK(transpose(k, p.ldk, KT, Nk, Nk, D, s, g, (size_t)Nk * p.ldk, (size_t)Nk * D));
    AG_CHK(transpose(q, p.ldq, QT, Nq, Nq, D, s, g, (size_t)Nq * p.ldq, (size_t)Nq * D));
    AG_CHK(transpose(d_o, p.lddo, dOT, Nq, Nq, D, s, g, (size_t)Nq * p.lddo, (size_t)Nq * D));
    AG_CHK(gemm(q, p.ldq, Kc, S, Nk, Nq, Nk, D, true, p.scale, tap1x1, partial, partial_cap, s, g));
    AG_CHK(gemm(d_o, p.lddo, Vc, dP, Nk, Nq, Nk, D, true, 1.f, tap1x1, partial, partial_cap, s, g));
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(Nk / 64, Nq / 64, g), dim3(256), 0, s, (const float*)S, (const float*)dP, lse, delta, p.scale, Nq,
                       Nk, P, dS, PT, dST);
    AG_CHK(hipGetLastError());
    AG_CHK(gemm(PT, Nq, dOT, p.dv + (size_t)b * Nk * p.lddv + h * D, p.lddv, Nk, D, Nq, false, 1.f, tap1x1, partial, partial_cap, s, g));
    AG_CHK(gemm(dST, Nq, QT, p.dk + (size_t)b * Nk * p.lddk + h * D, p.lddk, Nk, D, Nq, false, 1.f, tap1x1, partial, partial_cap, s, g));
    AG_CHK(gemm(dS, Nk, KT, p.dq + (size_t)b * Nq * p.lddq + h * D, p.lddq, Nq, D, Nk, false, 1.f, tap1x1, partial, partial_cap, s, g));
  }
  return hipSuccess;
}
