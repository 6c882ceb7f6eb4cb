// K3/K4 — GroupNorm(+SiLU) and LayerNorm forward/backward on NHWC bf16 rows (gfx950).
// Replaces torch.nn.GroupNorm / F.silu / LayerNorm inside diffusers' ResnetBlock2D, Transformer2DModel,
// AutoencoderKL decoder (reference call sites generate_data.py:112, :701) and their autograd backward (:721).
// HBM-bound: 16-byte vector loads, fp32 statistics, deterministic two-stage reductions (no float atomics
// anywhere, so results are bitwise reproducible run to run).
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SPLIT = 256;

// rows per partial block: ~32K elements per block so that mid-size tensors still spread over hundreds of blocks
__host__ __device__ inline int gn_split(int HW, int C) {
  int rows = 32768 / C;
  if (rows < 4) rows = 4;
  int s = (HW + rows - 1) / rows;
  if (s < 1) s = 1;
  if (s > GN_MAX_SPLIT) s = GN_MAX_SPLIT;
  return s;
}

// Pass 1 (fwd): per (b, split, group) partial (count, mean, M2).
// Pass 1 (bwd): per (b, split, group) partial (s1 = sum dxhat, s2 = sum dxhat*xhat).
template <bool BWD>
__global__ __launch_bounds__(GN_THREADS) void gn_partial_kernel(GroupNormParams p) {
  // sh[r][2][C]: per staging-row partial sums, combined in a fixed order (no atomics: bitwise reproducible)
  extern __shared__ float sh[];
  const int C = p.C, G = p.G, cpg = C / G, VC = C >> 3;
  const int b = blockIdx.y, s = blockIdx.x, S = gridDim.x;
  const int rows_per = (p.HW + S - 1) / S;
  const int row_begin = s * rows_per, row_end = min(p.HW, row_begin + rows_per);
  const int VCt = min(VC, GN_THREADS);
  const int R = GN_THREADS / VCt;
  const int my_r = threadIdx.x / VCt, my_vc0 = threadIdx.x % VCt;
  if (my_r < R) {
    for (int vc = my_vc0; vc < VC; vc += VCt) {
      float a0[8], a1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
      float ga[8], be[8], mean[8], rstd[8];
      if (BWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = vc * 8 + e, g = c / cpg;
          ga[e] = p.gamma[c]; be[e] = p.beta[c];
          mean[e] = p.stats[((size_t)b * G + g) * 2]; rstd[e] = p.stats[((size_t)b * G + g) * 2 + 1];
        }
      }
      // 4 rows in flight per thread (memory-level parallelism); accumulation order stays fixed
      for (int row0 = row_begin + my_r; row0 < row_end; row0 += 4 * R) {
        uint4 xr[4], dr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = row0 + u * R;
          const size_t pix = (size_t)b * p.HW + (row < row_end ? row : row_begin);
          xr[u] = *(const uint4*)(p.x + pix * p.x_ld + vc * 8);
          if (BWD) dr[u] = *(const uint4*)(p.dy + pix * p.dy_ld + vc * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (row0 + u * R >= row_end) continue;
          float xv[8];
          unpack8(xr[u], xv);
          if (!BWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a0[e] += xv[e]; a1[e] += xv[e] * xv[e]; }
          } else {
            float dv[8];
            unpack8(dr[u], dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float xh = (xv[e] - mean[e]) * rstd[e];
              float d = dv[e];
              if (p.silu) d *= dsilu_f(xh * ga[e] + be[e]);
              d *= ga[e];
              a0[e] += d; a1[e] += d * xh;
            }
          }
        }
      }
      float* dst = sh + (size_t)my_r * 2 * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) { dst[vc * 8 + e] = a0[e]; dst[C + vc * 8 + e] = a1[e]; }
    }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += GN_THREADS) {
    float t0 = 0.f, t1 = 0.f;
    for (int r = 0; r < R; ++r) {
      const float* src = sh + (size_t)r * 2 * C;
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) { t0 += src[c]; t1 += src[C + c]; }
    }
    float* out = p.scratch + (((size_t)b * S + s) * G + g) * 3;
    if (!BWD) {
      const float n = (float)(row_end - row_begin) * cpg;
      const float m = n > 0 ? t0 / n : 0.f;
      out[0] = n; out[1] = m; out[2] = fmaxf(t1 - t0 * m, 0.f);  // M2 = sumsq - n*mean^2
    } else {
      out[0] = t0; out[1] = t1; out[2] = 0.f;
    }
  }
}

// Pass 2: one wave per (b, g) merges the S partials in a fixed tree order.
//   fwd: Chan et al. parallel merge of (n, mean, M2) -> stats[b][g] = (mean, rstd)
//   bwd: sums -> fin[b][g] = (s1/n, s2/n)
template <bool BWD>
__global__ __launch_bounds__(GN_THREADS) void gn_finalize_kernel(GroupNormParams p, int S, float* fin) {
  const int lane = threadIdx.x & 63;
  const int bg = blockIdx.x * (GN_THREADS / 64) + (threadIdx.x >> 6);
  if (bg >= p.B * p.G) return;
  const int b = bg / p.G, g = bg % p.G;
  const int cpg = p.C / p.G;
  if (!BWD) {
    float n = 0.f, mean = 0.f, M2 = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float* in = p.scratch + (((size_t)b * S + s) * p.G + g) * 3;
      const float nb = in[0], mb = in[1], M2b = in[2];
      if (nb > 0.f) {
        const float nn = n + nb, d = mb - mean;
        mean += d * (nb / nn);
        M2 += M2b + d * d * (n * nb / nn);
        n = nn;
      }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), M2b = __shfl_xor(M2, o, 64);
      // symmetric merge (both partners compute the same result, in the same operand order)
      const float nn = n + nb;
      if (nn > 0.f) {
        const float lo_n = (lane & o) ? nb : n, hi_n = (lane & o) ? n : nb;
        const float lo_m = (lane & o) ? mb : mean, hi_m = (lane & o) ? mean : mb;
        const float lo_M = (lane & o) ? M2b : M2, hi_M = (lane & o) ? M2 : M2b;
        const float d = hi_m - lo_m;
        mean = lo_m + d * (hi_n / nn);
        M2 = lo_M + hi_M + d * d * (lo_n * hi_n / nn);
        n = nn;
      }
    }
    if (lane == 0) {
      p.stats[((size_t)b * p.G + g) * 2] = mean;
      p.stats[((size_t)b * p.G + g) * 2 + 1] = rsqrtf(M2 / n + p.eps);
    }
  } else {
    float s1 = 0.f, s2 = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float* in = p.scratch + (((size_t)b * S + s) * p.G + g) * 3;
      s1 += in[0]; s2 += in[1];
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
      const float n = (float)p.HW * cpg;
      fin[((size_t)b * p.G + g) * 2] = s1 / n;
      fin[((size_t)b * p.G + g) * 2 + 1] = s2 / n;
    }
  }
}

// Pass 1+2 when the producing convolutions emitted per-(64-row block, channel) partials (mean, M2 over 64 rows; CF_STATS of the
// implicit-GEMM epilogue): one 256-thread workgroup per (b, g) merges the (HW / 64) x (C / G) partials of its group, no pass over the
// tensor.  All partials have the same count, so the merge is mean = avg(mean_p), M2 = sum(M2_p) + 64 * sum((mean_p - mean)^2), with
// every thread's partials held in registers between the two reductions; fixed-order wave + LDS sums (bitwise reproducible).
__device__ __forceinline__ float gn_block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(GN_THREADS) void gn_finalize_chan_kernel(GroupNormParams p) {
  __shared__ float red[4];
  const int bg = blockIdx.x;
  const int b = bg / p.G, g = bg % p.G;
  const int cpg = p.C / p.G, nb = p.HW >> 6, P = nb * cpg;
  constexpr int MAXP = 12;                       // partials per thread kept in registers (P <= 3072); beyond that they are re-read
  float pm[MAXP], pM[MAXP];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int e = threadIdx.x + i * GN_THREADS;
    pm[i] = 0.f; pM[i] = 0.f;
    if (e < P) {
      const int rb = e / cpg, c = g * cpg + (e - rb * cpg);
      const float2 in = *(const float2*)(p.chan_part + (((size_t)b * nb + rb) * p.part_ld + c) * 2);
      pm[i] = in.x; pM[i] = in.y; s += in.x;
    }
  }
  for (int e = threadIdx.x + MAXP * GN_THREADS; e < P; e += GN_THREADS) {
    const int rb = e / cpg, c = g * cpg + (e - rb * cpg);
    s += p.chan_part[(((size_t)b * nb + rb) * p.part_ld + c) * 2];
  }
  const float mean = gn_block_sum(s, red) / (float)P;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    const int e = threadIdx.x + i * GN_THREADS;
    if (e < P) { const float d = pm[i] - mean; q += pM[i] + 64.f * d * d; }
  }
  for (int e = threadIdx.x + MAXP * GN_THREADS; e < P; e += GN_THREADS) {
    const int rb = e / cpg, c = g * cpg + (e - rb * cpg);
    const float2 in = *(const float2*)(p.chan_part + (((size_t)b * nb + rb) * p.part_ld + c) * 2);
    const float d = in.x - mean;
    q += in.y + 64.f * d * d;
  }
  const float M2 = gn_block_sum(q, red);
  if (threadIdx.x == 0) {
    p.stats[((size_t)b * p.G + g) * 2] = mean;
    p.stats[((size_t)b * p.G + g) * 2 + 1] = rsqrtf(fmaxf(M2, 0.f) / (64.f * (float)P) + p.eps);
  }
}

// Instead of pass 3 when the consumer applies the normalisation itself (CF_GNFOLD): the per-(image, channel) affine y = x * a + b
__global__ __launch_bounds__(GN_THREADS) void gn_coef_kernel(GroupNormParams p) {
  const int b = blockIdx.x, cpg = p.C / p.G;
  for (int c = threadIdx.x; c < p.C; c += GN_THREADS) {
    const int g = c / cpg;
    const float mean = p.stats[((size_t)b * p.G + g) * 2], rstd = p.stats[((size_t)b * p.G + g) * 2 + 1];
    const float a = rstd * p.gamma[c];
    *(float2*)(p.coef + ((size_t)b * p.C + c) * 2) = make_float2(a, p.beta[c] - mean * a);
  }
}

// Pass 3: build the per-channel affine in LDS, stream the tensor (4 vectors in flight per thread).
template <bool BWD>
__global__ __launch_bounds__(GN_THREADS) void gn_apply_kernel(GroupNormParams p, const float* fin) {
  extern __shared__ float sh[];
  const int C = p.C, G = p.G, cpg = C / G, VC = C >> 3;
  const int b = blockIdx.y;
  float* A = sh;          // fwd: scale ; bwd: gamma
  float* Bv = sh + C;     // fwd: shift ; bwd: mean
  float* Cv = sh + 2 * C; // bwd: rstd
  float* Dv = sh + 3 * C; // bwd: s1/n
  float* Ev = sh + 4 * C; // bwd: s2/n
  float* Fv = sh + 5 * C; // bwd: beta
  for (int c = threadIdx.x; c < C; c += GN_THREADS) {
    const int g = c / cpg;
    const float mean = p.stats[((size_t)b * G + g) * 2], rstd = p.stats[((size_t)b * G + g) * 2 + 1];
    if (!BWD) {
      const float a = rstd * p.gamma[c];
      A[c] = a; Bv[c] = p.beta[c] - mean * a;
    } else {
      A[c] = p.gamma[c]; Bv[c] = mean; Cv[c] = rstd; Fv[c] = p.beta[c];
      Dv[c] = fin[((size_t)b * G + g) * 2]; Ev[c] = fin[((size_t)b * G + g) * 2 + 1];
    }
  }
  __syncthreads();
  // a thread owns one 8-channel vector column and walks rows: its 8 (or 48) coefficients live in registers for the whole sweep, no
  // per-vector index division, no LDS traffic in the streaming loop (the kernel was VALU/LDS-bound at ~2.8 TB/s before)
  const int VCt = min(VC, GN_THREADS);
  const int R = GN_THREADS / VCt;
  const int my_r = threadIdx.x / VCt, my_vc0 = threadIdx.x % VCt;
  if (my_r >= R) return;
  const int rstep = gridDim.x * R;
  for (int vc = my_vc0; vc < VC; vc += VCt) {
    float ca[8], cb[8], cc[8], cd[8], ce[8], cf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = vc * 8 + e;
      ca[e] = A[c]; cb[e] = Bv[c];
      if (BWD) { cc[e] = Cv[c]; cd[e] = Dv[c]; ce[e] = Ev[c]; cf[e] = Fv[c]; }
    }
    for (int row0 = blockIdx.x * R + my_r; row0 < p.HW; row0 += 4 * rstep) {
      uint4 xr[4], dr[4], orr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = row0 + u * rstep;
        if (row < p.HW) {
          const size_t pix = (size_t)b * p.HW + row;
          xr[u] = *(const uint4*)(p.x + pix * p.x_ld + vc * 8);
          if (BWD) {
            dr[u] = *(const uint4*)(p.dy + pix * p.dy_ld + vc * 8);
            if (p.accumulate) orr[u] = *(const uint4*)(p.dx + pix * p.dx_ld + vc * 8);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = row0 + u * rstep;
        if (row >= p.HW) continue;
        const size_t pix = (size_t)b * p.HW + row;
        float xv[8], ov[8];
        unpack8(xr[u], xv);
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float y = xv[e] * ca[e] + cb[e];
            ov[e] = p.silu ? silu_f(y) : y;
          }
          *(uint4*)(p.y + pix * p.y_ld + vc * 8) = pack8(ov);
        } else {
          float dv[8];
          unpack8(dr[u], dv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (xv[e] - cb[e]) * cc[e];
            float d = dv[e];
            if (p.silu) d *= dsilu_f(xh * ca[e] + cf[e]);
            d *= ca[e];
            ov[e] = cc[e] * (d - cd[e] - xh * ce[e]);
          }
          if (p.accumulate) {
            float old[8];
            unpack8(orr[u], old);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] += old[e];
          }
          *(uint4*)(p.dx + pix * p.dx_ld + vc * 8) = pack8(ov);
        }
      }
    }
  }
}

// ---------------- LayerNorm forward, narrow rows (C <= 1536): a wave takes LN_RPW consecutive rows, all their loads in flight at
// once (a 320-channel row is only 640 bytes: one row per wave left the kernel latency-bound), gamma / beta in registers ------------
constexpr int LN_RPW = 4;
template <int VI>
__global__ __launch_bounds__(256) void ln_rows_kernel(LayerNormParams p) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_RPW;
  if (row0 >= p.M) return;
  const int VC = p.C >> 3;
  const bool stats_only = p.y == nullptr;     // the LayerNorm is folded into the GEMM that follows (CF_LNFOLD): statistics only
  float ga[VI][8], be[VI][8];
#pragma unroll
  for (int i = 0; i < VI; ++i) {
    const int vc = lane + 64 * i;
    if (vc < VC && !stats_only) {
      const float4 g0 = *(const float4*)(p.gamma + vc * 8), g1 = *(const float4*)(p.gamma + vc * 8 + 4);
      const float4 b0 = *(const float4*)(p.beta + vc * 8), b1 = *(const float4*)(p.beta + vc * 8 + 4);
      ga[i][0] = g0.x; ga[i][1] = g0.y; ga[i][2] = g0.z; ga[i][3] = g0.w; ga[i][4] = g1.x; ga[i][5] = g1.y; ga[i][6] = g1.z; ga[i][7] = g1.w;
      be[i][0] = b0.x; be[i][1] = b0.y; be[i][2] = b0.z; be[i][3] = b0.w; be[i][4] = b1.x; be[i][5] = b1.y; be[i][6] = b1.z; be[i][7] = b1.w;
    }
  }
  uint4 raw[LN_RPW][VI];
#pragma unroll
  for (int r = 0; r < LN_RPW; ++r)
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int vc = lane + 64 * i;
      raw[r][i] = make_uint4(0, 0, 0, 0);
      if (vc < VC && row0 + r < p.M) raw[r][i] = *(const uint4*)(p.x + (size_t)(row0 + r) * p.x_ld + vc * 8);
    }
#pragma unroll
  for (int r = 0; r < LN_RPW; ++r) {
    const int row = row0 + r;
    if (row >= p.M) break;                    // wave-uniform
    float xv[VI][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      unpack8(raw[r][i], xv[i]);
      if (lane + 64 * i < VC) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xv[i][e];
      }
    }
    const float mean = wave_sum(s) / p.C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < VI; ++i)
      if (lane + 64 * i < VC) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[i][e] - mean; v += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(v) / p.C + p.eps);
    if (lane == 0 && p.stats) { p.stats[(size_t)row * 2] = mean; p.stats[(size_t)row * 2 + 1] = rstd; }
    if (stats_only) continue;
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int vc = lane + 64 * i;
      if (vc < VC) {
        float ov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (xv[i][e] - mean) * rstd * ga[i][e] + be[i][e];
        *(uint4*)(p.y + (size_t)row * p.y_ld + vc * 8) = pack8(ov);
      }
    }
  }
}

// ---------------- LayerNorm: one wave per row, up to 4 x 64 x 8 = 2048 channels ----------------
template <bool BWD>
__global__ __launch_bounds__(256) void ln_kernel(LayerNormParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.M) return;
  const int VC = p.C >> 3;
  float xv[4][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vc = lane + 64 * i;
    if (vc < VC) {
      unpack8(*(const uint4*)(p.x + (size_t)row * p.x_ld + vc * 8), xv[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[i][e];
    }
  }
  float mean, rstd;
  if (!BWD) {
    mean = wave_sum(s) / p.C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vc = lane + 64 * i;
      if (vc < VC) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = xv[i][e] - mean; v += d * d; }
      }
    }
    rstd = rsqrtf(wave_sum(v) / p.C + p.eps);
    if (lane == 0 && p.stats) { p.stats[(size_t)row * 2] = mean; p.stats[(size_t)row * 2 + 1] = rstd; }
    if (p.y == nullptr) return;               // statistics only (the LayerNorm is folded into the GEMM that follows)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vc = lane + 64 * i;
      if (vc < VC) {
        float ov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = vc * 8 + e;
          ov[e] = (xv[i][e] - mean) * rstd * p.gamma[c] + p.beta[c];
        }
        *(uint4*)(p.y + (size_t)row * p.y_ld + vc * 8) = pack8(ov);
      }
    }
  } else {
    mean = p.stats[(size_t)row * 2]; rstd = p.stats[(size_t)row * 2 + 1];
    float dv[4][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vc = lane + 64 * i;
      if (vc < VC) {
        unpack8(*(const uint4*)(p.dy + (size_t)row * p.dy_ld + vc * 8), dv[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = vc * 8 + e;
          const float xh = (xv[i][e] - mean) * rstd;
          const float d = dv[i][e] * p.gamma[c];
          dv[i][e] = d; xv[i][e] = xh;
          s1 += d; s2 += d * xh;
        }
      }
    }
    s1 = wave_sum(s1) / p.C; s2 = wave_sum(s2) / p.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vc = lane + 64 * i;
      if (vc < VC) {
        float ov[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = rstd * (dv[i][e] - s1 - xv[i][e] * s2);
        bf16_t* dst = p.dx + (size_t)row * p.dx_ld + vc * 8;
        if (p.accumulate) {
          float old[8];
          unpack8(*(const uint4*)dst, old);
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] += old[e];
        }
        *(uint4*)dst = pack8(ov);
      }
    }
  }
}

}  // namespace

size_t groupnorm_scratch_bytes(int B, int G) { return ((size_t)B * GN_MAX_SPLIT * G * 3 + (size_t)B * G * 2) * sizeof(float); }

static size_t gn_partial_lds(int C) {
  const int VC = C / 8, VCt = VC < GN_THREADS ? VC : GN_THREADS, R = GN_THREADS / VCt;
  return (size_t)R * 2 * C * sizeof(float);
}
static hipError_t gn_check(const GroupNormParams& p) {
  if (p.C % 8 || p.C % p.G || (p.x_ld & 7)) return hipErrorInvalidValue;
  return hipSuccess;
}
static int gn_apply_blocks(const GroupNormParams& p) {
  int blocks = (p.HW * (p.C / 8) + GN_THREADS * 8 - 1) / (GN_THREADS * 8);
  if (blocks < 1) blocks = 1;
  static const int total = getenv("DD_GN_BLOCKS") ? atoi(getenv("DD_GN_BLOCKS")) : 4096;   // blocks per launch over all images (2048 -> 4096: norm family 265 -> 260 ms per 32-image batch)
  const int cap = total / (p.B > 0 ? p.B : 1) + 1;
  if (blocks > cap) blocks = cap;
  return blocks;
}

template <bool BWD>
static hipError_t gn_launch(const GroupNormParams& p, hipStream_t stream) {
  const int S = gn_split(p.HW, p.C);
  float* fin = p.scratch + (size_t)p.B * GN_MAX_SPLIT * p.G * 3;   // [B][G][2] finalize output of the backward sums
  if (!BWD && p.chan_part) {
    if (p.HW & 63) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_finalize_chan_kernel, dim3(p.B * p.G), dim3(GN_THREADS), 0, stream, p);
  } else {
    hipLaunchKernelGGL((gn_partial_kernel<BWD>), dim3(S, p.B), dim3(GN_THREADS), gn_partial_lds(p.C), stream, p);
    hipLaunchKernelGGL((gn_finalize_kernel<BWD>), dim3((p.B * p.G + 3) / 4), dim3(GN_THREADS), 0, stream, p, S, fin);
  }
  if (!BWD && p.coef) {
    hipLaunchKernelGGL(gn_coef_kernel, dim3(p.B), dim3(GN_THREADS), 0, stream, p);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((gn_apply_kernel<BWD>), dim3(gn_apply_blocks(p), p.B), dim3(GN_THREADS), (BWD ? 6 : 2) * p.C * sizeof(float), stream, p,
                     (const float*)fin);
  return hipGetLastError();
}

hipError_t launch_groupnorm_fwd(const GroupNormParams& p, hipStream_t stream) {
  if (gn_check(p) != hipSuccess || (!p.coef && (p.y_ld & 7))) return hipErrorInvalidValue;
  return gn_launch<false>(p, stream);
}

hipError_t launch_groupnorm_bwd(const GroupNormParams& p, hipStream_t stream) {
  if (gn_check(p) != hipSuccess || (p.dy_ld & 7) || (p.dx_ld & 7)) return hipErrorInvalidValue;
  return gn_launch<true>(p, stream);
}

// (mean, rstd) of every row from the (sum, sum^2) partials the producing GEMM emitted per column span (CF_ROWSTATS): M x spans x 8 bytes
// read instead of the M x C x 2 of a pass over the tensor.  fp32 sums of the fp32 values in front of their bf16 rounding.
__global__ __launch_bounds__(256) void ln_rowpart_finalize_kernel(LayerNormParams p) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= p.M) return;
  const float2* pp = (const float2*)p.rowpart + (size_t)row * p.rowpart_ld;
  float a = 0.f, q = 0.f;
  for (int s = 0; s < p.spans; ++s) { const float2 v = pp[s]; a += v.x; q += v.y; }
  const float mean = a / p.C;
  const float var = fmaxf(q / p.C - mean * mean, 0.f);
  *(float2*)(p.stats + (size_t)row * 2) = make_float2(mean, rsqrtf(var + p.eps));
}

hipError_t launch_layernorm_fwd(const LayerNormParams& p, hipStream_t stream) {
  if (p.C % 8 || p.C > 2048 || (p.x_ld & 7) || (p.y && (p.y_ld & 7))) return hipErrorInvalidValue;
  if (!p.y) {
    if (!p.stats) return hipErrorInvalidValue;
    if (p.rowpart) {
      if (p.spans < 1 || p.rowpart_ld < p.spans) return hipErrorInvalidValue;
      hipLaunchKernelGGL(ln_rowpart_finalize_kernel, dim3((p.M + 255) / 256), dim3(256), 0, stream, p);
      return hipGetLastError();
    }
  }
  static const int rows_kernel = getenv("DD_LN_ROWS") ? atoi(getenv("DD_LN_ROWS")) : 1;   // A/B switch
  if (rows_kernel && p.C <= 1536) {
    const int vi = (p.C / 8 + 63) / 64;
    const int blocks = (p.M + 4 * LN_RPW - 1) / (4 * LN_RPW);
    switch (vi) {
      case 1: hipLaunchKernelGGL((ln_rows_kernel<1>), dim3(blocks), dim3(256), 0, stream, p); break;
      case 2: hipLaunchKernelGGL((ln_rows_kernel<2>), dim3(blocks), dim3(256), 0, stream, p); break;
      default: hipLaunchKernelGGL((ln_rows_kernel<3>), dim3(blocks), dim3(256), 0, stream, p); break;
    }
    return hipGetLastError();
  }
  hipLaunchKernelGGL((ln_kernel<false>), dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_layernorm_bwd(const LayerNormParams& p, hipStream_t stream) {
  if (p.C % 8 || p.C > 2048 || (p.x_ld & 7) || (p.dy_ld & 7) || (p.dx_ld & 7) || !p.stats) return hipErrorInvalidValue;
  hipLaunchKernelGGL((ln_kernel<true>), dim3((p.M + 3) / 4), dim3(256), 0, stream, p);
  return hipGetLastError();
}
