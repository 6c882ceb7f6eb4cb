// K1/K2/K8 — implicit-GEMM convolution / GEMM / linear on bf16 MFMA (v_mfma_f32_16x16x32_bf16), gfx950.
//
// Replaces the cuDNN/cuBLAS kernels PyTorch dispatches for diffusers' Conv2d / Linear inside
// UNet2DConditionModel, AutoencoderKL.decode and timm resnet50 (reference call sites:
// generate_data.py:112, :701, :705) and their input-gradients (torch.autograd.grad, :721/:761).
//
// Layout: activations NHWC bf16 (a [pixels, C] row-major matrix, arbitrary row stride so channel
// concatenation is a view), weights pre-packed [N][K] with k = (tap, cin), cin fastest (cin % 8 == 0).
// Tile: (WM*64) x (WN*64) x 64 per workgroup, one 64x64 sub-tile per wave (4x4 MFMA 16x16x32
// tiles, fp32 accumulators), LDS double-buffered with 128-byte rows XOR-swizzled by (row & 7) so that
// both the staging ds_write_b128 and the fragment ds_read_b128 are bank-conflict free
// (tools/lds_conflicts.py), register-staged global->LDS prefetch of the next K-step under the MFMAs.
// The MFMA is issued "swapped" (A = weight rows, B = pixel rows) so each lane owns 4 consecutive
// output channels of one pixel: epilogue loads/stores are 8-byte (bf16x4) / 16-byte (fp32x4) vectors.
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"

namespace {

template <int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(ConvGemmParams p) {
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int NT = WM * WN * 64;
  constexpr int RPT = NT / 8;   // tile rows covered by one staging pass (8 x 16-byte slots per 128-byte row)
  constexpr int AV = BM / RPT;  // staging vectors per thread per K-step
  constexpr int BV = BN / RPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BUF_BYTES = (BM + BN) * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile mapping: XCD-aware (blocks b, b+8, ... share an L2) with n-tiles fastest, so that the
  // blocks of one XCD walk neighbouring pixel rows against the same weights.
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int nblk = ntm * ntn;
  int logical;
  {
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (logical / ntn) * BM;
  const int n0 = (logical % ntn) * BN;

  const int ksteps = p.K >> 6;
  const int per = (ksteps + p.ksplit - 1) / p.ksplit;
  const int kz = blockIdx.y;
  const int k_begin = kz * per;
  const int k_end = min(ksteps, k_begin + per);

  // ---- staging assignment
  const int j = tid & 7;
  const int r0 = tid >> 3;
  int pixb[AV], iy0[AV], ix0[AV];
  {
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int m = m0 + r0 + RPT * i;
      if (m < p.M) {
        const int b = m / HoWo;
        const int rem = m - b * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        pixb[i] = b * p.H * p.W;
        iy0[i] = oy * p.stride;
        ix0[i] = ox * p.stride;
      } else {
        pixb[i] = 0; iy0[i] = -1000000; ix0[i] = 0;
      }
    }
  }
  uint4 ra[AV], rb[BV];
  const int shift = p.shift, parity = p.parity;

  unsigned okmask = 0;
  const int cin = p.cin, cin8 = p.cin >> 3;
  const bool uniform_tap = (cin & 63) == 0;  // every 64-deep K-step lies inside one filter tap
  auto load_tile = [&](int kt) {
    okmask = 0;
    int e, coff;
    bool ev;
    if (uniform_tap) {
      const int chunk = kt / p.ntaps;             // wave-uniform: scalar ALU + s_load; K order = (64-channel chunk, tap)
      const int tap = kt - chunk * p.ntaps;
      coff = chunk * 64 + j * 8;
      e = p.taptab[tap];
      ev = true;
    } else {
      const int k8 = kt * 8 + j;
      const int tap = k8 / cin8;
      ev = tap < p.ntaps;
      coff = (k8 - tap * cin8) * 8;
      e = p.taptab[ev ? tap : 0];
    }
    const int dx = (e & 63) - 32, dy = ((e >> 6) & 63) - 32;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int ly = iy0[i] + dy, lx = ix0[i] + dx;
      const int sy = ly >> shift, sx = lx >> shift;
      bool ok = ev && ly >= 0 && lx >= 0 && sy < p.H && sx < p.W;
      if (parity) ok = ok && (((ly | lx) & 1) == 0);
      const unsigned off = ok ? (unsigned)(pixb[i] + sy * p.W + sx) * (unsigned)p.x_ld + (unsigned)coff : 0u;
      ra[i] = *(const uint4*)(p.x + off);  // unconditional load from a safe address; zero-select at LDS-store time
      okmask |= ok ? (1u << i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int n = n0 + r0 + RPT * i;
      const bool okn = n < p.N;
      rb[i] = *(const uint4*)(p.w + (size_t)(okn ? n : 0) * p.K + (size_t)kt * 64 + j * 8);
      okmask |= okn ? (1u << (16 + i)) : 0u;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* A = smem + buf * BUF_BYTES;
    unsigned char* Bt = A + BM * 128;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int row = r0 + RPT * i;
      const bool ok = (okmask >> i) & 1u;
      uint4 v = ra[i];
      v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
      *(uint4*)(A + row * 128 + ((j ^ (row & 7)) << 4)) = v;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int row = r0 + RPT * i;
      const bool ok = (okmask >> (16 + i)) & 1u;
      uint4 v = rb[i];
      v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
      *(uint4*)(Bt + row * 128 + ((j ^ (row & 7)) << 4)) = v;
    }
  };

  f32x4 acc[4][4];  // [jn][i]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const unsigned char* A = smem + buf * BUF_BYTES;
    const unsigned char* Bt = A + BM * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 xf[4], wf[4];
      const int slot = fq + 4 * ks;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        xf[i] = *(const bf16x8*)(A + row * 128 + ((slot ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int row = wn * 64 + jn * 16 + fr;
        wf[jn] = *(const bf16x8*)(Bt + row * 128 + ((slot ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int jn = 0; jn < 4; ++jn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jn], xf[i], acc[jn][i], 0, 0, 0);
    }
  };

  if (k_begin < k_end) {
    load_tile(k_begin);
    store_tile(0);
    __syncthreads();
    for (int kt = k_begin; kt < k_end; ++kt) {
      const int cur = (kt - k_begin) & 1;
      const bool more = (kt + 1 < k_end);
      if (more) load_tile(kt + 1);
      compute(cur);
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue. acc[jn][i][r]: n = n0 + wn*64 + jn*16 + fq*4 + r ; m = m0 + wm*64 + i*16 + fr
  if (p.ksplit > 1) {
    float* part = p.partial + (size_t)kz * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int nb = n0 + wn * 64 + jn * 16 + fq * 4;
        float* pp = part + (size_t)m * p.N + nb;
        if (nb + 4 <= p.N && !(p.N & 3)) {
          *(float4*)pp = make_float4(acc[jn][i][0], acc[jn][i][1], acc[jn][i][2], acc[jn][i][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (nb + r < p.N) pp[r] = acc[jn][i][r];
        }
      }
    }
    return;
  }
  const float* bias = p.bias;
  if ((p.flags & CF_BIAS) && p.bias_sel) bias += (size_t)(*p.bias_sel) * p.bias_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
    if (m >= p.M) continue;
    if (p.flags & CF_GEGLU) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int nb = n0 + wn * 64 + (2 * t) * 16 + fq * 4;
        if (nb >= p.N) continue;
        float h[4], g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { h[r] = acc[2 * t][i][r]; g[r] = acc[2 * t + 1][i][r]; }
        Epi::apply(p, bias, m, nb, h, g, nb + 16);
      }
    } else {
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int nb = n0 + wn * 64 + jn * 16 + fq * 4;
        if (nb >= p.N) continue;
        float h[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = acc[jn][i][r];
        Epi::apply(p, bias, m, nb, h, h, 0);
      }
    }
  }
}

// split-K reduction + epilogue: one thread per (row, 4 logical output columns)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvGemmParams p) {
  const int geglu = (p.flags & CF_GEGLU) ? 1 : 0;
  const int ncols = geglu ? (p.N >> 1) : p.N;
  const int ngrp = (ncols + 3) >> 2;
  const size_t total = (size_t)p.M * ngrp;
  const float* bias = p.bias;
  if ((p.flags & CF_BIAS) && p.bias_sel) bias += (size_t)(*p.bias_sel) * p.bias_stride;
  const bool vec = !(p.N & 3);                             // whole float4 groups: 16-byte loads of the partial slabs
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / ngrp);
    const int ob = (int)(idx % ngrp) * 4;
    int nb, nbg = 0;
    if (geglu) { nb = (ob >> 4) * 32 + (ob & 15); nbg = nb + 16; } else nb = ob;
    float h[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0};
    if (vec) {
      for (int z = 0; z < p.ksplit; ++z) {
        const float* pp = p.partial + ((size_t)z * p.M + m) * p.N;
        const float4 v = *(const float4*)(pp + nb);
        h[0] += v.x; h[1] += v.y; h[2] += v.z; h[3] += v.w;
        if (geglu) { const float4 u = *(const float4*)(pp + nbg); g[0] += u.x; g[1] += u.y; g[2] += u.z; g[3] += u.w; }
      }
    } else {
      for (int z = 0; z < p.ksplit; ++z) {
        const float* pp = p.partial + ((size_t)z * p.M + m) * p.N;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nb + r < p.N) h[r] += pp[nb + r];
          if (geglu && nbg + r < p.N) g[r] += pp[nbg + r];
        }
      }
    }
    Epi::apply(p, bias, m, nb, h, g, nbg);
  }
}

}  // namespace

int conv_gemm_big_config(int M, int N, int K, int flags);
int conv_halo_config(const ConvGemmParams& p);           // conv_halo.hip: 0, or the tile form of the halo-resident 3x3 kernel
int conv_halo_split(const ConvGemmParams& p);            // conv_halo.hip: chunk split of the 8 x 8 level (1: none)
hipError_t launch_conv_halo(const ConvGemmParams& p, int tn, hipStream_t stream);
int gemm_pp_config(const ConvGemmParams& p);             // conv_halo.hip: pointwise ping-pong GEMM (narrow N)
hipError_t launch_gemm_pp(const ConvGemmParams& p, int tn, hipStream_t stream);
int gemm_ws_config(const ConvGemmParams& p);             // gemm_ws.hip: weight-stationary GEMM (K = 320 pointwise layers)
int gemm_ws_rowstat_spans(const ConvGemmParams& p);
hipError_t launch_gemm_ws(const ConvGemmParams& p, int tn, hipStream_t stream);
void conv_gemm_big_tile(int cfg, int* bm, int* bn);
hipError_t launch_conv_gemm_big(const ConvGemmParams& p, int cfg, hipStream_t stream);

static int small_split(int M, int N, int K) {
  // 128x128 (or 256x64) tiles: split K when the grid is small and K is deep
  const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
  const int ksteps = K / 64;
  if (tiles >= 192 || ksteps < 8) return 1;
  int s = (384 + tiles - 1) / tiles;
  if (s > ksteps / 4) s = ksteps / 4;
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  return s;
}
static int big_split(int cfg, int M, int N, int K) {
  int bm, bn;
  conv_gemm_big_tile(cfg, &bm, &bn);
  const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
  const int ksteps = K / 64;
  if (tiles >= 200) return 1;
  int s = (256 + tiles - 1) / tiles;
  if (s > ksteps / 8) s = ksteps / 8;
  if (s > 32) s = 32;
  if (s < 1) s = 1;
  return s;
}
static int nonempty_split(int split, int ksteps) {
  if (split < 1) split = 1;
  if (split > ksteps) split = ksteps;
  const int per = (ksteps + split - 1) / split;
  return (ksteps + per - 1) / per;
}

int conv_gemm_pick_split(int M, int N, int K) {
  int s = small_split(M, N, K);
  // the chunk split of the halo-resident kernel at the 8 x 8 level (conv_halo_split: it needs geometry this query does not have,
  // so every small-M 3x3-shaped problem with N % 320 == 0 reserves its scratch)
  if (M <= 16384 && (M & 255) == 0 && N % 320 == 0 && K % 576 == 0 && K >= 1152) {
    int hs = 1;
    while ((M / 256) * (N / 320) * hs < 192 && hs < 16) hs *= 2;
    if (hs > s) s = hs;
  }
  for (int flags = 0; flags <= CF_GEGLU; flags += CF_GEGLU) {
    const int cfg = conv_gemm_big_config(M, N, K, flags);
    if (cfg) { const int b = big_split(cfg, M, N, K); if (b > s) s = b; }
  }
  return s;
}

// kernel / split-K selection shared by the launcher and conv_gemm_can_emit_stats
static void select_config(const ConvGemmParams& p, size_t partial_cap_bytes, int* cfg_out, int* split_out) {
  const int ksteps = p.K / 64;
  int cfg = ((p.force_small & 1) ? 0 : conv_gemm_big_config(p.M, p.N, p.K, p.flags));
  if (cfg && p.ksplit <= 0) {
    // the persistent kernel runs one 8-wave workgroup per CU: it needs >= ~3/4 of the 256 CUs busy, else the
    // 128x128 kernel (2-3 workgroups per CU, more of them) wins
    int bm, bn;
    conv_gemm_big_tile(cfg, &bm, &bn);
    const int items = ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * big_split(cfg, p.M, p.N, p.K);
    if (items < 192) cfg = 0;
  }
  int split = p.ksplit > 0 ? p.ksplit : (cfg ? big_split(cfg, p.M, p.N, p.K) : small_split(p.M, p.N, p.K));
  while (split > 1 && (size_t)split * p.M * p.N * sizeof(float) > partial_cap_bytes) --split;
  if (!p.partial) split = 1;
  *cfg_out = cfg;
  *split_out = nonempty_split(split, ksteps);
}

bool conv_gemm_can_emit_stats(ConvGemmParams p, size_t partial_cap_bytes) {
  if ((p.K & 63) || p.M <= 0 || p.N <= 0 || (p.M & 63) || (p.N & 7) || (p.y_ld & 7)) return false;
  if (p.flags & (CF_GEGLU | CF_OUT_F32 | CF_MASK | CF_RES_F32)) return false;
  if ((p.flags & CF_RES) && (p.res_ld & 7)) return false;
  { ConvGemmParams q = p; q.flags |= CF_STATS; if (conv_halo_config(q) || gemm_pp_config(q)) return true; }
  int cfg, split;
  select_config(p, partial_cap_bytes, &cfg, &split);
  if (!cfg || split != 1) return false;
  return true;
}

int conv_gemm_big_rowstat_span(int cfg);

bool conv_gemm_can_emit_rowstats(ConvGemmParams p, size_t partial_cap_bytes, int* spans) {
  if ((p.K & 63) || p.M <= 0 || p.N <= 0 || (p.N & 7) || (p.y_ld & 7)) return false;
  if (p.flags & (CF_GEGLU | CF_OUT_F32 | CF_MASK | CF_RES_F32 | CF_STATS | CF_LNFOLD)) return false;
  if ((p.flags & CF_RES) && (p.res_ld & 7)) return false;
  // the batched epilogue of the two-workgroup pointwise forms only (fast staging: Cin % 64 == 0, one tap, no upsample)
  if ((p.cin & 63) || p.shift || p.ntaps != 1 || p.stride != 1 || p.H != p.Ho || p.W != p.Wo) return false;
  if ((size_t)p.B * p.H * p.W * (size_t)p.x_ld * 2 >= 0xF0000000ull) return false;
  { ConvGemmParams q = p; q.flags |= CF_ROWSTATS;
    if (gemm_ws_config(q)) { if (spans) *spans = gemm_ws_rowstat_spans(q); return true; }
    if (gemm_pp_config(q)) { if (spans) *spans = p.N / 80; return true; } }
  int cfg, split;
  select_config(p, partial_cap_bytes, &cfg, &split);
  const int span = cfg ? conv_gemm_big_rowstat_span(cfg) : 0;
  if (!span || split != 1 || p.N % (2 * span)) return false;
  if (spans) *spans = p.N / span;
  return true;
}

bool conv_gemm_can_fold_gn(ConvGemmParams p) {
  p.flags |= CF_GNFOLD;
  static float dummy;
  if (!p.gn_coef) p.gn_coef = &dummy;
  return conv_halo_config(p) != 0;
}

hipError_t launch_conv_gemm(ConvGemmParams p, size_t partial_cap_bytes, hipStream_t stream) {
  if (p.K & 63) return hipErrorInvalidValue;
  if ((p.flags & CF_GNFOLD) && !conv_halo_config(p)) return hipErrorInvalidValue;
  if ((p.flags & CF_LNFOLD) && (!p.ln_stats || !p.ln_c1 || p.ntaps != 1)) return hipErrorInvalidValue;
  if (p.flags & CF_ROWSTATS) {
    int spans = 0;
    if (!p.rowpart || !conv_gemm_can_emit_rowstats(p, partial_cap_bytes, &spans) || p.rowpart_ld < spans) return hipErrorInvalidValue;
  }
  // the kernels index the input with 32-bit element offsets (outputs and residuals use 64-bit offsets)
  if ((size_t)p.B * p.H * p.W * (size_t)p.x_ld >= 0xFFFF0000ull) return hipErrorInvalidValue;
  if (p.M <= 0 || p.N <= 0) return hipSuccess;
  if ((p.flags & CF_STATS) && (!p.stats || !conv_gemm_can_emit_stats(p, partial_cap_bytes))) return hipErrorInvalidValue;
  if (p.wgroup_rows > 0) {
    // grouped weights (a batch of GEMMs stacked along M): the persistent big-tile kernel only, whole tiles per group, no split-K
    int cfg, split;
    select_config(p, 0, &cfg, &split);
    int bm = 0, bn = 0;
    if (cfg) conv_gemm_big_tile(cfg, &bm, &bn);
    if (!cfg || split != 1 || p.wgroup_rows % bm || p.ntaps != 1 || (p.flags & ~(CF_OUT_F32 | CF_BIAS))) return hipErrorInvalidValue;
    p.ksplit = 1;
    return launch_conv_gemm_big(p, cfg, stream);
  }
  {
    // deep 3x3 / stride 1: halo-resident input tile; at the 8 x 8 level with a chunk split into the split-K scratch + the reduce kernel
    ConvGemmParams q = p;
    const int hs = conv_halo_split(q);
    if (hs > 1 && (size_t)hs * p.M * p.N * sizeof(float) > partial_cap_bytes) q.partial = nullptr;     // scratch too small: no split form
    if (const int tn = conv_halo_config(q)) {
      q.ksplit = conv_halo_split(q) > 1 ? hs : q.ksplit;
      hipError_t e = launch_conv_halo(q, tn, stream);
      if (e == hipSuccess && q.ksplit > 1) {
        const size_t total = (size_t)q.M * ((q.N + 3) / 4);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, q);
        e = hipGetLastError();
      }
      return e;
    }
  }
  if (const int tn = gemm_ws_config(p)) return launch_gemm_ws(p, tn, stream);           // K = 320 pointwise layers: weight-stationary GEMM
  if (const int tn = gemm_pp_config(p)) return launch_gemm_pp(p, tn, stream);           // narrow pointwise layers: ping-pong GEMM
  int cfg, split;
  select_config(p, partial_cap_bytes, &cfg, &split);
  p.ksplit = split;
  if (cfg) {
    hipError_t e = launch_conv_gemm_big(p, cfg, stream);
    if (e != hipSuccess) return e;
    if (split > 1) {
      const int ncols = (p.flags & CF_GEGLU) ? p.N / 2 : p.N;
      const size_t total = (size_t)p.M * ((ncols + 3) / 4);
      int blocks = (int)((total + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
      e = hipGetLastError();
    }
    return e;
  }
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute((const void*)conv_gemm_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (256 + 64) * 128);
    hipFuncSetAttribute((const void*)conv_gemm_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 128) * 128);
    attr_done = true;
  }
  const bool narrow = ((p.N % 128) != 0 && (p.N % 128) <= 64) && p.M >= 256;
  if (narrow) {
    const int ntm = (p.M + 255) / 256, ntn = (p.N + 63) / 64;
    dim3 grid(ntm * ntn, split);
    hipLaunchKernelGGL((conv_gemm_kernel<4, 1>), grid, dim3(256), 2 * (256 + 64) * 128, stream, p);
  } else {
    const int ntm = (p.M + 127) / 128, ntn = (p.N + 127) / 128;
    dim3 grid(ntm * ntn, split);
    hipLaunchKernelGGL((conv_gemm_kernel<2, 2>), grid, dim3(256), 2 * (128 + 128) * 128, stream, p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (split > 1) {
    const int ncols = (p.flags & CF_GEGLU) ? p.N / 2 : p.N;
    const size_t total = (size_t)p.M * ((ncols + 3) / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, p);
    e = hipGetLastError();
  }
  return e;
}
