// K1 (deep 3x3 shapes) -- implicit-GEMM 3x3 / stride-1 / pad-1 convolution (forward and input-gradient) with a HALO-resident input
// tile and a phase-alternating ("ping-pong") K loop, bf16 MFMA, gfx950.  Same packed weights, tap tables, NHWC row layout and epilogue
// semantics (bias / residual / ReLU / CF_STATS GroupNorm partials) as conv_gemm2.hip; launch_conv_gemm sends eligible shapes here.
//
// Why (profiles/r03_gemm_schedule_micro.txt): on a CU the LDS-DMA operand stream and the MFMA stream do not overlap
// to their individual rates -- a K-step costs about delivery time + MFMA time in every schedule tried -- so the lever is bytes per FLOP:
//   * the nine taps of a 64-channel chunk read the SAME input pixels shifted by (dy, dx): the tile's 256 output pixels (256 / W image
//     rows) plus a one-pixel border -- (R + 2) x (W + 2) pixels x 128 B -- are staged ONCE per chunk, zero-filled outside the image by
//     the buffer load's range check, and the A fragments of tap (dy, dx) are read at pixel + dy * (W + 2) + dx.  Input traffic per
//     K-step falls from 32 KB to ~5.6 KB; only the weights stream every K-step;
//   * 512 x 160 / 512 x 128 / 256 x 320 / 256 x 256 tiles, 8 waves as 4 x 2 or 2 x 4, wave tile 128 x 80 (64): 20-40 KB of weights per
//     10.5 MFLOP K-step;
//   * the two waves of a SIMD (different row groups) run the same program one s_barrier apart: between two barriers one issues the
//     8 * TN MFMAs of a K half while the other reads fragments and issues its share of the next K-step's weight DMA
//     (cdna_hip_programming.md section 5, "The 256^2 8-phase template"; MI355X_MICROARCH.md "Two waves per SIMD").
// The 3x3 kernel is not persistent (one workgroup per tile, XCD-aware tile order, n-tiles fastest inside an XCD: a persistent form
// measured slower); the pointwise GEMM of the same loop (gemm_pps_kernel, below) is.
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"

namespace {

__device__ __forceinline__ void hdma16(const void* base, void* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffff00u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}
template <int CTRL>
__device__ __forceinline__ float hdpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float hrow16_sum(float v) {
  v = hdpp_add<0xB1>(v); v = hdpp_add<0x4E>(v); v = hdpp_add<0x141>(v); v = hdpp_add<0x140>(v);
  return v;
}

// Shared epilogue of the ping-pong kernels: a lane owns, per 16-row tile a, 8 consecutive channels of a tile pair (16-byte loads /
// stores) and 4 of an odd last tile.  bias / residual / ReLU / CF_STATS (GroupNorm partials per 64-row block) as in conv_gemm2.hip, plus
// CF_LNFOLD (out = rstd[m] * (acc - mean[m] * c1[n]) + b'[n]: the LayerNorm in front of this linear is folded into its weights) and
// CF_ROWSTATS ((sum, sum^2) of every output row over this wave's TN * 16 columns, for the LayerNorm that consumes the tensor).
// stats_s: the (mean, rstd) rows of the tile in LDS (persistent GEMM: they arrive with the bias through the LDS-DMA ring), or null (global loads).
// LA: 16-row tiles whose residual rows / statistics are requested together (2: one exposed memory latency per 32 rows; 4: per 64 rows --
// 16 more registers, which only the TN = 4 forms have: with TN = 5 it spilled)
template <int TN, int LA = 2, class MOf>
__device__ __forceinline__ void pp_epilogue(const ConvGemmParams& p, f32x4 (&acc)[8][TN], MOf m_of, int wr, int wc, int n0,
                                            const float* bias_s, const float* c1_s, int span, int fr, int fq, const float* stats_s = nullptr) {
  constexpr int TNP = TN & ~1;
  const int fl = p.flags;
  const int wb = n0 + wc * (TN * 16);
  const float* bw = bias_s + wc * (TN * 16);
  const float* cw = c1_s + wc * (TN * 16);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {                   // 64-row blocks (the granule of the GroupNorm partials)
    float s1[TN][4], s2[TN][4];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[jn][r] = 0.f; s2[jn][r] = 0.f; }
#pragma unroll
    for (int hb = 0; hb < 4 / LA; ++hb) {
    // the residual rows / LayerNorm statistics of LA 16-row tiles are requested before their first use: one exposed memory latency per
    // 32 (64) rows instead of one per tile pair (the accumulators and the GroupNorm sums leave ~50 registers free here)
    uint4 rvp[LA][TN / 2];
    uint2 rvo[LA];
    float2 lst[LA];
    int mrow[LA];
#pragma unroll
    for (int a4 = 0; a4 < LA; ++a4) {
      mrow[a4] = m_of(wr * 128 + (blk * 4 + hb * LA + a4) * 16 + fr);
      const bf16_t* rp = (const bf16_t*)p.res + (size_t)mrow[a4] * p.res_ld + wb;
      if (fl & CF_RES) {
#pragma unroll
        for (int t = 0; t < TN / 2; ++t) rvp[a4][t] = *(const uint4*)(rp + t * 32 + fq * 8);
        if constexpr (TN & 1) rvo[a4] = *(const uint2*)(rp + (TN - 1) * 16 + fq * 4);
      } else {
#pragma unroll
        for (int t = 0; t < TN / 2; ++t) rvp[a4][t] = make_uint4(0, 0, 0, 0);
        rvo[a4] = make_uint2(0, 0);
      }
      lst[a4] = !(fl & CF_LNFOLD) ? make_float2(0.f, 1.f)
                : stats_s ? *(const float2*)(stats_s + (wr * 128 + (blk * 4 + hb * LA + a4) * 16 + fr) * 2) : *(const float2*)(p.ln_stats + (size_t)mrow[a4] * 2);
    }
#pragma unroll
    for (int a4 = 0; a4 < LA; ++a4) {
      const int a = blk * 4 + hb * LA + a4;
      const int m = mrow[a4];
      bf16_t* yp = (bf16_t*)p.y + (size_t)m * p.y_ld;
      const float rs = lst[a4].y * p.alpha, nm = -lst[a4].y * lst[a4].x;      // CF_LNFOLD: rstd and -rstd * mean of this lane's row (1, 0 otherwise)
      float r1 = 0.f, r2 = 0.f;                          // CF_ROWSTATS
      auto four = [&](const f32x4& v, int col, unsigned q0, unsigned q1, float* t1, float* t2) {
        float4 b = *(const float4*)(bw + col);
        if (fl & CF_LNFOLD) {
          const float4 c = *(const float4*)(cw + col);
          b.x = __builtin_fmaf(nm, c.x, b.x); b.y = __builtin_fmaf(nm, c.y, b.y); b.z = __builtin_fmaf(nm, c.z, b.z); b.w = __builtin_fmaf(nm, c.w, b.w);
        }
        float v0 = __builtin_fmaf(v[0], rs, b.x), v1 = __builtin_fmaf(v[1], rs, b.y), v2 = __builtin_fmaf(v[2], rs, b.z), v3 = __builtin_fmaf(v[3], rs, b.w);
        if (fl & CF_RES) {
          v0 += __uint_as_float(q0 << 16); v1 += __uint_as_float(q0 & 0xffff0000u);
          v2 += __uint_as_float(q1 << 16); v3 += __uint_as_float(q1 & 0xffff0000u);
        }
        if (fl & CF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        if (fl & CF_STATS) {
          t1[0] += v0; t1[1] += v1; t1[2] += v2; t1[3] += v3;
          t2[0] = __builtin_fmaf(v0, v0, t2[0]); t2[1] = __builtin_fmaf(v1, v1, t2[1]);
          t2[2] = __builtin_fmaf(v2, v2, t2[2]); t2[3] = __builtin_fmaf(v3, v3, t2[3]);
        }
        if (fl & CF_ROWSTATS) {
          r1 += (v0 + v1) + (v2 + v3);
          r2 = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, r2))));
        }
        return make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
      };
#pragma unroll
      for (int t = 0; t < TN / 2; ++t) {
        const int col = t * 32 + fq * 8;                // column of the pair's first value inside the wave's span
        const uint4 rv = rvp[a4][t];
        const uint2 lo = four(acc[a][2 * t], col, rv.x, rv.y, s1[2 * t], s2[2 * t]);
        const uint2 hi = four(acc[a][2 * t + 1], col + 4, rv.z, rv.w, s1[2 * t + 1], s2[2 * t + 1]);
        *(uint4*)(yp + wb + col) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      if constexpr (TN & 1) {
        const int col = (TN - 1) * 16 + fq * 4;
        *(uint2*)(yp + wb + col) = four(acc[a][TN - 1], col, rvo[a4].x, rvo[a4].y, s1[TN - 1], s2[TN - 1]);
      }
      if (fl & CF_ROWSTATS) {
        // the row's TN * 16 columns of this wave sit in lanes fr, fr + 16, fr + 32, fr + 48
        r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
        r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
        if (fq == 0) *(float2*)(p.rowpart + ((size_t)m * p.rowpart_ld + span) * 2) = make_float2(r1, r2);
      }
    }
    }
    if (fl & CF_STATS) {
      // per-(64-row block, channel) (mean, M2) of the stored values for the GroupNorm that consumes this tensor (conv_gemm2.hip emit_stats)
      const int m0w = m_of(wr * 128 + blk * 64);           // 64 consecutive output rows (tw >= 64, or whole image rows)
      float* dst0 = p.stats + ((size_t)(m0w >> 6) * p.stats_ld) * 2;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        float o[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sa = hrow16_sum(s1[jn][r]), sq = hrow16_sum(s2[jn][r]);
          const float mean = sa * (1.f / 64.f);
          o[2 * r] = mean; o[2 * r + 1] = fmaxf(sq - sa * mean, 0.f);
        }
        const int col = jn < TNP ? (jn >> 1) * 32 + fq * 8 + (jn & 1) * 4 : jn * 16 + fq * 4;
        if (fr == 0) {
          float* dst = dst0 + (size_t)(wb + col) * 2;                  // p.stats is already offset to this op's first channel
          *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
          *(float4*)(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    }
  }
}

// Tile geometry (host: halo_geometry): a tile is th x tw OUTPUT pixels of one image (th * tw = 256, tw = min(Wo, 128) a power of two),
// i.e. 256 / Wo whole image rows for Wo <= 128 and a 2 x 128 block for wider images; its halo is (th + 2) x (tw + 2) LOGICAL input
// pixels (the fused nearest-2x upsample of the decoder's / UNet's upsamplers reads stored pixel (iy >> shift, ix >> shift)).
struct HaloGeo { int ltw, th, halo_px, tiles_x, tiles_y, ipt, stagger, tab; };   // tab: the one-tile kernel uses the LDS halo address table   // stagger (persistent form): start delay per phase, in 10 ns ticks   // ipt: images per tile (4 at 8 x 8: a tile is 4 whole images, each with its own 10 x 10 halo block)

// WN = 4: 256 x (64 TN) tiles, waves 2 (M) x 4 (N); WN = 2: 512 x (32 TN) tiles, waves 4 (M) x 2 (N) -- the narrow outputs of the decoder's
// last level (N = 128).  Either way a wave owns 128 rows x TN * 16 columns and waves w, w + 4 (one SIMD) sit in different row groups.
// MI: the 8 x 8 level's form -- multi-image tiles + chunk split with fp32 partial sums; its own instantiation so that the hot forms
// keep their register allocation (as runtime branches the additions took the TN = 5 kernels from 12 to 80 bytes of scratch).
template <int TN, int WN, bool MI = false>
__global__ __launch_bounds__(512, 1) void conv_halo_kernel(ConvGemmParams p, HaloGeo geo) {
  const int lw = geo.ltw, halo_px = geo.halo_px;
  constexpr int BM = (8 / WN) * 128, BN = WN * TN * 16;
  constexpr int NPC = BN / 8;                           // weight pieces (8 rows x 128 B) per K-step
  constexpr int NWP = (NPC + 7) / 8;                    // ... per wave (the last one only on the first NPC % 8 waves when BN % 64 != 0)
  constexpr int WB = BN * 128;                          // bytes of one weight stage
  constexpr unsigned OOB = 0xfffffff0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem + 2 * WB;
  float* const bias_s = (float*)(halo + ((halo_px + 7) & ~7) * 128);
  float* const coef_s = bias_s + BN;                     // CF_GNFOLD: (a, b) of the chunk's 64 channels for this tile's image

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int grp = wave >> 2;                            // SIMD partners w, w + 4 are in different groups: group 1 runs one barrier behind
  const int fr = lane & 15, fq = lane >> 4;
  const int Wd = 1 << lw, W2 = Wd + 2;

  // ---- tile of this workgroup (n-tiles fastest; blocks b, b + 8, ... share an XCD)
  const int ntn = (p.N + BN - 1) / BN, tiles = (p.M / BM) * ntn;      // (N < BN: the narrow form TN = 1, one n-tile)
  int tile;
  {
    const int q = tiles >> 3, r = tiles & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int n0 = (tile % ntn) * BN;
  const int mt = tile / ntn, tpi = geo.tiles_x * geo.tiles_y;
  const int ipt = MI ? geo.ipt : 1;                       // > 1: the tile is ipt whole (8 x 8) images, output rows contiguous
  const int img = MI ? mt * ipt : mt / tpi, tin = MI ? 0 : mt - img * tpi;
  const int y0 = (tin / geo.tiles_x) * geo.th, x0 = (tin % geo.tiles_x) << lw;    // first output pixel of the tile inside its image
  const int himg = MI ? (geo.th + 2) * W2 : 0;            // halo pixels of one image of a multi-image tile
  const int kz = MI ? blockIdx.y : 0;                     // chunk split (8 x 8 level: too few tiles to fill the chip): fp32 partial sums
  const int mimg = img * p.Ho * p.Wo;
  auto m_of = [&](int r) { return mimg + (y0 + (r >> lw)) * p.Wo + x0 + (r & (Wd - 1)); };   // output row of tile pixel r
  const int chunks_all = p.cin >> 6;
  const int cper = MI ? (chunks_all + p.ksplit - 1) / (p.ksplit > 0 ? p.ksplit : 1) : chunks_all;
  const int c_begin = MI ? kz * cper : 0, c_end = MI ? min(chunks_all, c_begin + cper) : chunks_all;
  const int KT = (c_end - c_begin) * 9;

  // ---- per-tap offsets: lane t holds tap t ((dy + 32) << 6 | (dx + 32)); read with readlane where needed
  const int v_taps = lane < 9 ? p.taptab[lane] : 0;
  if (tid < BN) bias_s[tid] = ((p.flags & CF_BIAS) && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;

  // ---- weight staging: wave w moves pieces w, w + 8, ... (8 rows x 128 B) of the BN weight rows of a K-step.  LDS row R of a wave's
  // TN * 16 span holds output channel chan_of_row(R): tiles are paired so that a lane's 4 + 4 accumulator rows of a pair are 8
  // consecutive channels (16-byte epilogue stores), exactly as in conv_gemm2.hip.
  const int prow = lane >> 3, jw = (lane & 7) ^ prow;
  constexpr int TNP = TN & ~1;
  unsigned woff[NWP];
#pragma unroll
  for (int i = 0; i < NWP; ++i) {
    const int R = (wave + 8 * i) * 8 + prow;
    const int wv = R / (TN * 16), q = R - wv * (TN * 16), jn = q >> 4, f = q & 15;
    const int ch = jn < TNP ? wv * (TN * 16) + (jn >> 1) * 32 + (f >> 2) * 8 + (jn & 1) * 4 + (f & 3) : R;
    // (rows behind the last output channel -- the narrow form pads N = 3 / 4 to a 32-row stage -- read as zeros: out of range)
    woff[i] = (((NPC & 7) == 0 || R < BN) && n0 + ch < p.N) ? ((unsigned)(n0 + ch) * (unsigned)p.K + (unsigned)(jw * 8)) * 2u : OOB;
  }
  auto issue_w = [&](int kt, int i) {
    if ((NPC & 7) == 0 || wave + 8 * i < NPC) hdma16(p.w, smem + (kt & 1) * WB + (wave + 8 * i) * 1024, woff[i], (unsigned)(c_begin * 9 + kt) * 128u);
  };
  // ---- halo staging (row half 0 only): pieces wave, wave + 4, ... of ceil(halo_px / 8); a lane's pixel hp = 8 * piece + (lane >> 3)
  const float inv_w2 = 1.f / (float)W2;
  const bf16_t* ximg = p.x + (size_t)img * p.H * p.W * p.x_ld;     // per-image base: 32-bit byte offsets only span one image
  // Halo address table (geo.tab; see 3.8.8 of DESIGN.md and conv_halo_persist_kernel): a workgroup of this kernel has ONE tile, so the byte
  // offset of every (piece, lane) of its halo -- out-of-image pixels as the out-of-range offset that reads zeros -- is the same for every
  // chunk.  All 8 waves compute their pieces once, in the prologue, store them behind the bias / coef block and request chunk c_begin with
  // them; the refills at the chunk boundaries (row group 0) are a table read + a request per piece.
  unsigned* const htab = (unsigned*)(coef_s + 128);
  const bool use_tab = geo.tab != 0;
  auto halo_voff = [&](int pc) {
    const int hp = pc * 8 + prow;
    const int hi = MI ? (int)(((float)hp + 0.5f) * (1.f / (float)(himg > 0 ? himg : 1))) : 0;    // image of the tile
    const int hq = hp - hi * himg;
    const int hy = (int)(((float)hq + 0.5f) * inv_w2), hx = hq - hy * W2;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;       // logical input pixel (= output pixel coordinates: stride 1, pad 1)
    const bool ok = hp < halo_px && iy >= 0 && iy < p.Ho && ix >= 0 && ix < p.Wo;
    const int j = (lane & 7) ^ (hx & 7);
    return ok ? ((unsigned)(hi * p.H * p.W + (iy >> p.shift) * p.W + (ix >> p.shift)) * (unsigned)p.x_ld + (unsigned)(j * 8)) * 2u : OOB;
  };
  auto first_halo = [&](int chunk) {                     // prologue, all 8 waves
    const int npc = (halo_px + 7) >> 3;
    for (int pc = wave; pc < npc; pc += 8) {
      const unsigned voff = halo_voff(pc);
      htab[pc * 64 + lane] = voff;
      hdma16(ximg, halo + pc * 1024, voff, (unsigned)chunk * 128u);
    }
  };
  auto issue_halo = [&](int chunk) {                     // row group 0 (waves 0 .. 3)
    const int npc = (halo_px + 7) >> 3;
    const unsigned soff = (unsigned)chunk * 128u;
    if (use_tab) {
      for (int pc = wave; pc < npc; pc += 16) {         // four pieces per round: table reads first, then the requests back to back
        unsigned e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = pc + 4 * u < npc ? htab[(pc + 4 * u) * 64 + lane] : OOB;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (pc + 4 * u >= npc) break;
          hdma16(ximg, halo + (pc + 4 * u) * 1024, e[u], soff);
        }
      }
      return;
    }
    for (int pc = wave; pc < npc; pc += 4) hdma16(ximg, halo + pc * 1024, halo_voff(pc), soff);
  };

  // CF_GNFOLD: GroupNorm(+SiLU) applied to the staged halo chunk in place -- y = silu(x * a[c] + b[c]) on every pixel that lies inside
  // the image (the zero padding stays zero) -- once per tile and chunk instead of a pass over the tensor in front of this convolution.
  // The two row groups take alternate 256-vector slices; 16 bytes = 8 channels of one pixel per thread and step.
  const bool gnf = (p.flags & CF_GNFOLD) != 0;
  auto load_coef = [&](int chunk) {        // wave 0: 64 channels x (a, b)
    if (wave == 0) *(float2*)(coef_s + lane * 2) = *(const float2*)(p.gn_coef + ((size_t)img * p.cin + chunk * 64 + lane) * 2);
  };
  auto apply_gn = [&](int g) {
    const int nv = halo_px * 8;
    for (int v = g * 256 + (tid & 255); v < nv; v += 512) {
      const int hp = v >> 3, ps = v & 7;
      const int hy = (int)(((float)hp + 0.5f) * inv_w2), hx = hp - hy * W2;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      if (iy < 0 || iy >= p.Ho || ix < 0 || ix >= p.Wo) continue;
      const int j = ps ^ (hx & 7);
      uint4* px = (uint4*)(halo + v * 16);
      float xv[8];
      unpack8(*px, xv);
      const float4 c0 = *(const float4*)(coef_s + j * 16), c1 = *(const float4*)(coef_s + j * 16 + 4);
      const float4 c2 = *(const float4*)(coef_s + j * 16 + 8), c3 = *(const float4*)(coef_s + j * 16 + 12);
      const float ca[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z}, cb[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = xv[e] * ca[e] + cb[e];
        xv[e] = p.gn_silu ? silu_f(y) : y;
      }
      *px = pack8(xv);
    }
  };

  f32x4 acc[8][TN];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue
  if (gnf) load_coef(c_begin);
  if (use_tab) first_halo(c_begin);
  else if (grp == 0) issue_halo(c_begin);
#pragma unroll
  for (int i = 0; i < NWP; ++i) issue_w(0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                      // also publishes bias_s / coef_s
  if (gnf) { apply_gn(grp); __syncthreads(); }
  if (grp == 1) __builtin_amdgcn_s_barrier();           // the second group runs one barrier behind

  // Two sections per K-step, one per 32-deep K half: 8 x TN MFMAs between two barriers (the barrier hand-off between the SIMD partners is
  // not hidden by anything: tools/micro/pingpong_gemm.hip, four 32-row strips per K-step cost 0.5 us of barrier skeleton per K-step,
  // two K halves 0.35), TN + 8 fragments live instead of 2 TN + 4.
  bf16x8 wf[TN], xf[8];
  // (narrow form, TN = 1: the second column wave of a row group only multiplies zero padding, columns 16 .. 31.  Letting it skip its
  // fragment reads and MFMAs measured SLOWER -- 1206 -> 1504 us on the decoder's conv_out: the form is bound by the exposed halo refill
  // of its two short chunks, not by LDS reads, and the branch cost the schedule.)
  int kt = 0;
  for (int c = c_begin; c < c_end; ++c) {
    if (c > c_begin) {
      // every read of the previous chunk's halo has retired (both groups waited lgkmcnt(0) in front of their last X barrier)
      if (grp == 0) {
        if (gnf) load_coef(c);
#ifndef DD_HALO_ABL       // timing ablation (-DDD_HALO_ABL: results wrong): the halo of chunk 0 serves every chunk -- what the refill at a chunk boundary costs
        issue_halo(c);
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if (gnf) {
        // group 1 arrives here one barrier late, i.e. behind the barrier in front of which group 0 waited for the new halo: it applies
        // its slice now, group 0 applies its own behind that same barrier, and the extra barrier below closes both
        apply_gn(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (gnf) {
        if (grp == 0) { apply_gn(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
      }
    }
    for (int t = 0; t < 9; ++t, ++kt) {
      const int e = __builtin_amdgcn_readlane(v_taps, t);
      const int dy = ((e >> 6) & 63) - 32, dx = (e & 63) - 32;
      const int tapoff = dy * W2 + dx;
      const unsigned char* Bb = smem + (kt & 1) * WB;
      const bool more = kt + 1 < KT;
      const int xs = (fr + 1 + dx) & 7;                   // 16-pixel row tiles start at multiples of 16 inside an image row (tw >= 16)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // ---- load section (the SIMD partner is in its MFMA section)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int row = wc * (TN * 16) + jn * 16 + fr;
          wf[jn] = *(const bf16x8*)(Bb + row * 128 + (((fq + 4 * ks) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int r = wr * 128 + a * 16 + fr;
          // halo pixel of output pixel r for this tap (multi-image tiles: + the two border rows of every image in front of r's)
          const int hp = r + 2 * (r >> lw) + (MI ? 2 * W2 * (r >> 6) : 0) + Wd + 3 + tapoff;
          xf[a] = *(const bf16x8*)(halo + hp * 128 + (((fq + 4 * ks) ^ xs) << 4));
        }
        if (ks == 0 && more) {
#pragma unroll
          for (int q = 0; q < NWP; ++q) issue_w(kt + 1, q);
        }
        // second half: this wave's weight pieces of K-step kt + 1 have landed and its LDS reads of this stage (and, on tap 8, of the
        // halo) have retired BEFORE the barrier behind which the other half reads the new stage / the halo is refilled
        if (ks == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- MFMA section
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[a][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jn], xf[a], acc[a][jn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();           // balance the barrier count of the two groups

  if constexpr (MI) {
    // chunk split: raw fp32 partial sums [split][M][N]; splitk_reduce_kernel adds them and applies the epilogue
    constexpr int TNP2 = TN & ~1;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
      float* pr = p.partial + ((size_t)kz * p.M + m_of(wr * 128 + a * 16 + fr)) * p.N + n0 + wc * (TN * 16);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const int col = jn < TNP2 ? (jn >> 1) * 32 + fq * 8 + (jn & 1) * 4 : jn * 16 + fq * 4;
        *(float4*)(pr + col) = make_float4(acc[a][jn][0], acc[a][jn][1], acc[a][jn][2], acc[a][jn][3]);
      }
    }
    return;
  }
  if constexpr (TN == 1) {
    // narrow form (conv_out: N <= 4): channels 0 .. 3 of a pixel are the four accumulator rows of lane group 0 of the first column wave;
    // bias only; fp32 (CF_OUT_F32: the image / eps rows are 8 floats wide and only the N valid ones may be written) or bf16 output
    if (wc == 0 && fq == 0) {
      const float4 b4 = *(const float4*)bias_s;
#pragma unroll
      for (int a = 0; a < 8; ++a) {
        const int m = m_of(wr * 128 + a * 16 + fr);
        const float v[4] = {acc[a][0][0] + b4.x, acc[a][0][1] + b4.y, acc[a][0][2] + b4.z, acc[a][0][3] + b4.w};
        if (p.flags & CF_OUT_F32) {
          float* yp = (float*)p.y + (size_t)m * p.y_ld;
          if (p.N == 4) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
          else
#pragma unroll
            for (int r = 0; r < 4; ++r) if (r < p.N) yp[r] = v[r];
        } else {
          bf16_t* yp = (bf16_t*)p.y + (size_t)m * p.y_ld;
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < p.N) yp[r] = f2bf(v[r]);
        }
      }
    }
  } else {
    pp_epilogue<TN>(p, acc, m_of, wr, wc, n0, bias_s, bias_s, 0, fr, fq);
  }
}

#ifdef DD_TRACE
// debug build only (tools/pp_trace.py, tools/halo_trace.py): per tile stamps in s_memrealtime ticks (10 ns)
__device__ unsigned long long g_pp_trace[8192 * 6];
#define HP_STAMP(t, i) do { if (threadIdx.x == 0 && (t) < 8192) g_pp_trace[(t) * 6 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HP_STAMP(t, i) do { } while (0)
#endif
// PERSISTENT form of the halo kernel for the tile shapes whose K loop is short (the decoder's levels: Cin = 128 / 256 / 512, i.e. 18 / 36 /
// 72 K-steps per 512 x 128 tile).  There a tile's prologue (100 KB of halo + the first weight stage: DMA issue, flight and drain with
// nothing beside them) and its epilogue (128 KB of stores + the GroupNorm partials) are a quarter of the tile, and with one workgroup per
// CU (LDS) nothing overlaps them.  One workgroup per CU walks its tiles: behind the last K-step of a tile the halo buffer and the idle
// weight stage are free, so the NEXT tile's first halo chunk, first weight stage and bias row are requested BEFORE this tile's epilogue
// and land while it stores.  Same tile -> XCD map as the one-tile kernel (an XCD's workgroups share a contiguous tile range, n-tiles
// fastest), same K loop, same epilogue; no CF_GNFOLD, N % BN == 0.  Results are bitwise those of conv_halo_kernel (same order of
// operations per tile).
template <int TN, int WN>
__global__ __launch_bounds__(512, 1) void conv_halo_persist_kernel(ConvGemmParams p, HaloGeo geo) {
  const int lw = geo.ltw, halo_px = geo.halo_px;
  constexpr int BM = (8 / WN) * 128, BN = WN * TN * 16;
  constexpr int NPC = BN / 8, NWP = (NPC + 7) / 8, WB = BN * 128;
  static_assert((NPC & 7) == 0, "whole weight pieces per wave");
  constexpr unsigned OOB = 0xfffffff0u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem + 2 * WB;
  float* const bias_base = (float*)(halo + ((halo_px + 7) & ~7) * 128);      // two slots of BN floats (tile parity)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int grp = wave >> 2;
  const int fr = lane & 15, fq = lane >> 4;
  const int Wd = 1 << lw, W2 = Wd + 2;
  const int ntn = p.N / BN, tiles = (p.M / BM) * ntn, tpi = geo.tiles_x * geo.tiles_y;
  // tiles of this workgroup: XCD x owns [xs, xs + xc); its gridDim.x / 8 workgroups take xs + idx, xs + idx + per, ...
  const int per = gridDim.x >> 3;
  int xs, xc;
  {
    const int q = tiles >> 3, r = tiles & 7, xcd = blockIdx.x & 7;
    xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    xc = q + (xcd < r ? 1 : 0);
  }
  int t = blockIdx.x >> 3;
  if (t >= xc) return;
  if (geo.stagger > 0) {
    // de-phase the workgroups: launched together and walking tiles of identical cost they would refill their halos in the same
    // microsecond, chip-wide, at the HBM rate (~11 B/clk per CU) instead of a CU's own (MI355X guide, "prologue HBM burst").  16 phases
    // over the workgroups of an XCD; every wave of the workgroup waits the same time
    const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)((blockIdx.x >> 3) & 15) * (unsigned)geo.stagger;
    while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(4);
  }
  const int chunks = p.cin >> 6, KT = chunks * 9;
  const int v_taps = lane < 9 ? p.taptab[lane] : 0;
  const int prow = lane >> 3, jw = (lane & 7) ^ prow;
  constexpr int TNP = TN & ~1;
  unsigned woff[NWP];                                   // lane part of the weight offsets (n0 goes into the scalar offset)
#pragma unroll
  for (int i = 0; i < NWP; ++i) {
    const int R = (wave + 8 * i) * 8 + prow;
    const int wv = R / (TN * 16), q = R - wv * (TN * 16), jn = q >> 4, f = q & 15;
    const int ch = jn < TNP ? wv * (TN * 16) + (jn >> 1) * 32 + (f >> 2) * 8 + (jn & 1) * 4 + (f & 3) : R;
    woff[i] = ((unsigned)ch * (unsigned)p.K + (unsigned)(jw * 8)) * 2u;
  }
  const float inv_w2 = 1.f / (float)W2;
  // ---- halo address table (LDS, built once per workgroup): the LDS-DMA requests of a halo refill were bound by their ADDRESS ARITHMETIC
  // (~35 instructions and two divergent branches per 1 KB piece: ~210 cycles a piece, 2.5 us per refill with cache-hot data as with cold,
  // profiles/r06_decoder_persist.txt), and everything in it but the tile's origin is the same for every tile and chunk.  Entry (piece,
  // lane) = halo row (5 bits) | halo column (8 bits) | byte offset from the halo's first pixel / 16 (19 bits, swizzle included); per
  // tile only the origin pointer and the range of rows / columns that lie inside the image change (scalars).
  unsigned* const htab = (unsigned*)(bias_base + 2 * BN);
  const int npc = (halo_px + 7) >> 3;
  for (int pc = wave; pc < npc; pc += 8) {
    const int hp = pc * 8 + prow;
    const int hy = (int)(((float)hp + 0.5f) * inv_w2), hx = hp - hy * W2;
    const int sy = p.shift ? (hy + 1) >> 1 : hy, sx = p.shift ? (hx + 1) >> 1 : hx;     // stored pixel relative to the halo's first one
    const int j = (lane & 7) ^ (hx & 7);
    const unsigned rel = ((unsigned)(sy * p.W + sx) * (unsigned)p.x_ld + (unsigned)(j * 8)) * 2u;
    htab[pc * 64 + lane] = hp < halo_px ? ((unsigned)hy << 27) | ((unsigned)hx << 19) | (rel >> 4) : 0xf8000000u;    // row 31: never inside
  }
  struct Tile { int n0, img, y0, x0; };
  auto tile_of = [&](int tt) {
    const int tile = xs + tt;
    Tile g;
    g.n0 = (tile % ntn) * BN;
    const int mt = tile / ntn;
    g.img = mt / tpi;
    const int tin = mt - g.img * tpi;
    g.y0 = (tin / geo.tiles_x) * geo.th; g.x0 = (tin % geo.tiles_x) << lw;
    return g;
  };
  // weight K-step kt of tile g into stage st
  auto issue_w = [&](const Tile& g, int kt, int st, int i) {
    hdma16(p.w, smem + st * WB + (wave + 8 * i) * 1024, woff[i], (unsigned)g.n0 * (unsigned)p.K * 2u + (unsigned)kt * 128u);
  };
  auto issue_halo = [&](const Tile& g, int chunk, int nw) {     // nw = 4: waves 0 .. 3 (refill inside the K loop); 8: all waves (between tiles)
    // first stored pixel of the halo (logical (y0 - 1, x0 - 1); may lie in front of the image: those lanes are masked below)
    const int oy = p.shift ? (g.y0 >> 1) - 1 : g.y0 - 1, ox = p.shift ? (g.x0 >> 1) - 1 : g.x0 - 1;
    const bf16_t* xo = p.x + ((long long)g.img * p.H * p.W + (long long)oy * p.W + ox) * p.x_ld;
    // halo rows / columns inside the image: logical pixel y0 - 1 + hy in [0, Ho)
    const unsigned ylo = (unsigned)max(0, 1 - g.y0), yn = (unsigned)min(geo.th + 2, p.Ho - g.y0 + 1) - ylo;
    const unsigned xlo = (unsigned)max(0, 1 - g.x0), xn = (unsigned)min(Wd + 2, p.Wo - g.x0 + 1) - xlo;
    const unsigned soff = (unsigned)chunk * 128u;
    for (int pc = wave; pc < npc; pc += 4 * nw) {       // four pieces per round: table reads first, then the requests back to back
      unsigned e[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = pc + nw * u < npc ? htab[(pc + nw * u) * 64 + lane] : 0xf8000000u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (pc + nw * u >= npc) break;
        const bool ok = ((e[u] >> 27) - ylo) < yn && (((e[u] >> 19) & 255u) - xlo) < xn;
        hdma16(xo, halo + (pc + nw * u) * 1024, ok ? (e[u] & 0x7ffffu) << 4 : OOB, soff);
      }
    }
  };
  auto stage_first = [&](const Tile& g, int st, int slot) {   // everything a tile needs before its first K-step (all waves are between tiles here)
    issue_halo(g, 0, 8);
#pragma unroll
    for (int i = 0; i < NWP; ++i) issue_w(g, 0, st, i);
    if (tid < BN) bias_base[slot * BN + tid] = (p.flags & CF_BIAS) ? p.bias[g.n0 + tid] : 0.f;
  };

  Tile g = tile_of(t);
  int kg = 0;                                             // K-steps issued so far: K-step kt of the current tile lives in stage (kg + kt) & 1
  __syncthreads();                                      // the address table is complete
  stage_first(g, 0, 0);
  bf16x8 wf[TN], xf[8];
  for (int it = 0;; ++it) {
    HP_STAMP(xs + t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // halo chunk 0, weight stage, bias slot of this tile are in LDS
    HP_STAMP(xs + t, 1);
    if (grp == 1) __builtin_amdgcn_s_barrier();         // the second group runs one barrier behind
    f32x4 acc[8][TN];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    int kt = 0;
    for (int c = 0; c < chunks; ++c) {
      if (c > 0) {
        if (grp == 0) {
#ifndef DD_PERSIST_ABL
#define DD_PERSIST_ABL 0      // timing ablations (results wrong): 1 no refill at the chunk boundary; 2 the refill re-reads chunk 0 (cache-hot lines, same issue work)
#endif
          if (DD_PERSIST_ABL != 1) issue_halo(g, DD_PERSIST_ABL == 2 ? 0 : c, 4);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
      }
      for (int tp = 0; tp < 9; ++tp, ++kt) {
        const int e = __builtin_amdgcn_readlane(v_taps, tp);
        const int dy = ((e >> 6) & 63) - 32, dx = (e & 63) - 32;
        const int tapoff = dy * W2 + dx;
        const int st = (kg + kt) & 1;
        const unsigned char* Bb = smem + st * WB;
        const bool more = kt + 1 < KT;
        const int xsw = (fr + 1 + dx) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) {
            const int row = wc * (TN * 16) + jn * 16 + fr;
            wf[jn] = *(const bf16x8*)(Bb + row * 128 + (((fq + 4 * ks) ^ (row & 7)) << 4));
          }
#pragma unroll
          for (int a = 0; a < 8; ++a) {
            const int r = wr * 128 + a * 16 + fr;
            const int hp = r + 2 * (r >> lw) + Wd + 3 + tapoff;
            xf[a] = *(const bf16x8*)(halo + hp * 128 + (((fq + 4 * ks) ^ xsw) << 4));
          }
          if (ks == 0 && more) {
#pragma unroll
            for (int q = 0; q < NWP; ++q) issue_w(g, kt + 1, st ^ 1, q);
          }
          if (ks == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
              acc[a][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jn], xf[a], acc[a][jn], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();         // balance the barrier count of the two groups: every LDS read of this tile has retired
    HP_STAMP(xs + t, 2);
    kg += KT;
    const int tn_ = t + per;
    const bool have_next = tn_ < xc;
    Tile gn = g;
    if (have_next) {
      gn = tile_of(tn_);
      stage_first(gn, kg & 1, (it + 1) & 1);            // lands under the epilogue below
    }
    HP_STAMP(xs + t, 3);
    const int mimg = g.img * p.Ho * p.Wo, y0 = g.y0, x0 = g.x0;
    auto m_of = [&](int r) { return mimg + (y0 + (r >> lw)) * p.Wo + x0 + (r & (Wd - 1)); };
    const float* bias_s = bias_base + (it & 1) * BN;
#ifndef DD_PERSIST_LA
#define DD_PERSIST_LA 2      // 4 measured 2-3 % SLOWER on every decoder shape (256 VGPRs + 10 spills; profiles/r06_decoder_persist.txt)
#endif
    pp_epilogue<TN, DD_PERSIST_LA>(p, acc, m_of, wr, wc, g.n0, bias_s, bias_s, 0, fr, fq);
    HP_STAMP(xs + t, 4);
    if (!have_next) break;
    g = gn; t = tn_;
  }
}

// (DD_TRACE: g_pp_trace above -- persistent GEMM: wait for the first K-step, K loop start, K loop end, end of the epilogue)
// GEGLU epilogue of the persistent ping-pong GEMM (TN = 4: a wave owns two packed (16 hidden | 16 gate) groups).  The weight rows are
// assigned to MFMA rows so that a lane holds, per 16-row tile, the hidden AND gate pre-activations of 8 consecutive output columns
// (fq * 8 .. + 7 of the wave's 32): 16-byte stores of the product and of both halves of the CF_GEGLU_RAW stash.  bias / c1 in packed order.
template <class MOf>
__device__ __forceinline__ void pp_epilogue_geglu(const ConvGemmParams& p, f32x4 (&acc)[8][4], MOf m_of, int wr, int wc, int n0,
                                                  const float* bias_s, const float* c1_s, int fr, int fq, const float* stats_s) {
  const int fl = p.flags;
  const int pk = wc * 64 + (fq >> 1) * 32 + (fq & 1) * 8;      // packed column (inside the tile) of this lane's first hidden value
  float4 bh[2], bg[2], ch[2], cg[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    bh[t] = *(const float4*)(bias_s + pk + t * 4); bg[t] = *(const float4*)(bias_s + pk + 16 + t * 4);
    ch[t] = *(const float4*)(c1_s + pk + t * 4); cg[t] = *(const float4*)(c1_s + pk + 16 + t * 4);
  }
  const int oc = (n0 >> 1) + wc * 32 + fq * 8;                  // output column of the lane's 8 products
#pragma unroll
  for (int a2 = 0; a2 < 8; a2 += 2) {
    float2 lst[2];
    int mrow[2];
#pragma unroll
    for (int a4 = 0; a4 < 2; ++a4) {
      mrow[a4] = m_of(wr * 128 + (a2 + a4) * 16 + fr);
      lst[a4] = (fl & CF_LNFOLD) ? *(const float2*)(stats_s + (wr * 128 + (a2 + a4) * 16 + fr) * 2) : make_float2(0.f, 1.f);
    }
#pragma unroll
    for (int a4 = 0; a4 < 2; ++a4) {
      const int a = a2 + a4, m = mrow[a4];
      const float rs = lst[a4].y * p.alpha, nm = -lst[a4].y * lst[a4].x;
      float h[8], g[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float bhv[4] = {bh[t].x, bh[t].y, bh[t].z, bh[t].w}, bgv[4] = {bg[t].x, bg[t].y, bg[t].z, bg[t].w};
        const float chv[4] = {ch[t].x, ch[t].y, ch[t].z, ch[t].w}, cgv[4] = {cg[t].x, cg[t].y, cg[t].z, cg[t].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[t * 4 + r] = __builtin_fmaf(rs, acc[a][2 * t][r], __builtin_fmaf(nm, chv[r], bhv[r]));
          g[t * 4 + r] = __builtin_fmaf(rs, acc[a][2 * t + 1][r], __builtin_fmaf(nm, cgv[r], bgv[r]));
        }
      }
      if (fl & CF_GEGLU_RAW) {
        bf16_t* rp = p.raw + (size_t)m * p.raw_ld + n0 + pk;
        *(uint4*)rp = pack8(h);
        *(uint4*)(rp + 16) = pack8(g);
      }
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = h[e] * gelu_f(g[e]);
      *(uint4*)((bf16_t*)p.y + (size_t)m * p.y_ld + oc) = pack8(o);
    }
  }
}

// The same ping-pong K loop for pointwise (1x1 / linear) layers, PERSISTENT: 256 x (64 TN) tiles, both operands streamed through two
// 64-deep stages (TN = 5: A 32 KB + W 40 KB per K-step, 6.9 B per kFLOP against 13.8 for the 128 x 160 two-workgroup form, every A
// row read once).  One workgroup per CU walks its share of the tiles (an XCD owns a contiguous range, n-tiles fastest, dealt round-robin
// to its workgroups); the first K-step of the NEXT tile (and its bias / c1 rows, 16-byte LDS-DMA pieces into a two-slot ring) is
// requested during the last K-step of the current one, so launch, prologue latency and the drain of the epilogue's stores no longer sit
// between two K loops (tools/pp_trace.py: 2.9 + 1.1 us of a 21 us tile at K = 320).  GEGLU: TN = 4 with the packed (hidden | gate) epilogue.
// ------------------------------------------------------------------------------------------------------------------------------------
#ifdef DD_TRACE
#define PPS_STAMP(t, i) do { if (threadIdx.x == 0 && (t) < 8192) g_pp_trace[(t) * 6 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PPS_STAMP(t, i) do { } while (0)
#endif
template <int TN, bool GEGLU>
__global__ __launch_bounds__(512, 1) void gemm_pps_kernel(ConvGemmParams p) {
  constexpr int BM = 256, BN = 4 * TN * 16;
  constexpr int BUF = (BM + BN) * 128;
  constexpr int NP = 4 + TN;                           // pieces per wave and K-step: 4 of A, TN of W
  constexpr int AUX = 2 * BUF;                         // [slot][bias 2 KB | c1 2 KB | (mean, rstd) of the tile's 256 rows 2 KB]
  constexpr int SLOT = 6144;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int ntn = p.N / BN, tiles = (p.M / BM) * ntn;
  // this workgroup's tiles: first, first + per, ... (count of them) inside its XCD's contiguous range
  int first, count;
  const int per = gridDim.x >> 3;
  {
    const int q = tiles >> 3, r = tiles & 7, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int xbase = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, xcount = q + (xcd < r ? 1 : 0);
    first = xbase + slot;
    count = slot < xcount ? (xcount - slot + per - 1) / per : 0;
  }
  if (count == 0) return;
  const int KT = p.K >> 6;
  const int prow = lane >> 3, j = (lane & 7) ^ prow;
  constexpr int TNP = TN & ~1;
  unsigned aoff[4], woff[TN];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = ((unsigned)((wave + 8 * i) * 8 + prow) * (unsigned)p.x_ld + (unsigned)(j * 8)) * 2u;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int R = (wave + 8 * i) * 8 + prow;             // MFMA-ordered row of the W stage -> weight row of the tile
    const int wv = R / (TN * 16), q = R - wv * (TN * 16), jn = q >> 4, f = q & 15;
    int ch;
    if (GEGLU) {
      const int c = (f >> 2) * 8 + (jn >> 1) * 4 + (f & 3);          // output column inside the wave's 32
      ch = wv * 64 + (c >> 4) * 32 + (jn & 1) * 16 + (c & 15);
    } else {
      ch = jn < TNP ? wv * (TN * 16) + (jn >> 1) * 32 + (f >> 2) * 8 + (jn & 1) * 4 + (f & 3) : R;
    }
    woff[i] = ((unsigned)ch * (unsigned)p.K + (unsigned)(j * 8)) * 2u;
  }
  // bias / c1 pieces of a tile: waves 0, 1 bring bias[n0 .. n0 + BN), waves 2, 3 c1 (256 floats per piece; absent rows read as zeros)
  const unsigned auxoff = (wave < 4 && (wave & 1) * 256 + lane * 4 < BN && ((wave < 2) ? (p.flags & CF_BIAS) : (p.flags & CF_LNFOLD)))
                              ? (unsigned)((wave & 1) * 256 + lane * 4) * 4u : 0xffffff00u;
  const float* auxbase = wave < 2 ? p.bias : p.ln_c1;
  // piece `which` of K-step kt of the tile at (xt, n0) into stage `buf`
  auto issue = [&](const bf16_t* xt, unsigned wsoff, int kt, int buf, int which) {
    unsigned char* b = smem + buf * BUF;
    if (which < 4) hdma16(xt, b + (wave + 8 * which) * 1024, aoff[which], (unsigned)kt * 128u);
    else hdma16(p.w, b + BM * 128 + (wave + 8 * (which - 4)) * 1024, woff[which - 4], wsoff + (unsigned)kt * 128u);
  };
  const unsigned statoff = (p.flags & CF_LNFOLD) ? (unsigned)((wave & 1) * 1024 + lane * 16) : 0xffffff00u;    // waves 4, 5: 128 rows x 8 B each
  auto issue_aux = [&](int m0, int n0, int slot) {
    if (wave < 4) hdma16(auxbase ? (const void*)auxbase : (const void*)p.w, smem + AUX + slot * SLOT + (wave >> 1) * 2048 + (wave & 1) * 1024, auxoff, (unsigned)n0 * 4u);
    else if (wave < 6) hdma16(p.ln_stats ? (const void*)(p.ln_stats + (size_t)m0 * 2) : (const void*)p.w, smem + AUX + slot * SLOT + 4096 + (wave & 1) * 1024, statoff, 0u);
  };
  int tile = first;
  int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
  const bf16_t* xt = p.x + (size_t)m0 * p.x_ld;          // per-tile base: 32-bit byte offsets only span 256 rows
  unsigned wsoff = (unsigned)n0 * (unsigned)p.K * 2u;
  int cur = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) issue(xt, wsoff, 0, 0, q);
  issue_aux(m0, n0, 0);
  f32x4 acc[8][TN];
  bf16x8 wf[TN], xf[8];
  for (int it = 0; it < count; ++it) {
    PPS_STAMP(tile, 0);
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // next tile (wave-uniform)
    const bool have_next = it + 1 < count;
    const int tile_n = tile + per;
    const int m0n = (tile_n / ntn) * BM, n0n = (tile_n % ntn) * BN;
    const bf16_t* xtn = p.x + (size_t)m0n * p.x_ld;
    const unsigned wsoffn = (unsigned)n0n * (unsigned)p.K * 2u;
    // the first K-step of this tile was requested before the previous tile's epilogue (a counted wait that leaves the epilogue's last
    // stores in flight measured the same: what is left here is the skew between the eight waves' epilogues, tools/pp_trace.py)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PPS_STAMP(tile, 1);
    if (wr == 1) __builtin_amdgcn_s_barrier();            // the lower row half runs one barrier behind
    for (int kt = 0; kt < KT; ++kt) {
      const unsigned char* Ab = smem + cur * BUF;
      const unsigned char* Bb = Ab + BM * 128;
      const bool more = kt + 1 < KT;
      const bool pre = more || have_next;
      const bf16_t* nx = more ? xt : xtn;
      const unsigned nw = more ? wsoff : wsoffn;
      const int nk = more ? kt + 1 : 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                  // two sections per K-step, one per K half (see conv_halo_kernel)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int row = wc * (TN * 16) + jn * 16 + fr;
          wf[jn] = *(const bf16x8*)(Bb + row * 128 + (((fq + 4 * ks) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int a = 0; a < 8; ++a) {
          const int row = wr * 128 + a * 16 + fr;
          xf[a] = *(const bf16x8*)(Ab + row * 128 + (((fq + 4 * ks) ^ (row & 7)) << 4));
        }
        if (ks == 0 && pre) {
#pragma unroll
          for (int q = 0; q < NP; ++q) issue(nx, nw, nk, cur ^ 1, q);
          if (!more) issue_aux(m0n, n0n, (it + 1) & 1);
        }
        if (ks == 1) {
          if (more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
            acc[a][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jn], xf[a], acc[a][jn], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
      }
      cur ^= 1;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    PPS_STAMP(tile, 2);
    const float* bias_s = (const float*)(smem + AUX + (it & 1) * SLOT);
    const int m0c = m0;
    auto m_of = [&](int r) { return m0c + r; };
    if constexpr (GEGLU) pp_epilogue_geglu(p, acc, m_of, wr, wc, n0, bias_s, bias_s + 512, fr, fq, bias_s + 1024);
    else pp_epilogue<TN>(p, acc, m_of, wr, wc, n0, bias_s, bias_s + 512, (n0 / BN) * 4 + wc, fr, fq, bias_s + 1024);
    PPS_STAMP(tile, 3);
    tile = tile_n; m0 = m0n; n0 = n0n; xt = xtn; wsoff = wsoffn;
  }
}

bool halo_geometry(const ConvGemmParams& p, int bm, HaloGeo* g) {
  const int Wo = p.Wo, Ho = p.Ho;
  g->ipt = 1;
  if (Wo == 8 && Ho == 8 && bm == 256 && p.B % 4 == 0 && !p.shift) {
    // 8 x 8 level: a 256-pixel tile is four whole images, each with its own 10 x 10 halo block
    g->ltw = 3; g->th = 8; g->halo_px = 4 * 100; g->tiles_x = 1; g->tiles_y = 1; g->ipt = 4; g->stagger = 0; g->tab = 0;
    return true;
  }
  if (Wo < 16 || (Wo & (Wo - 1))) return false;
  // wide images: 512-pixel tiles are 8 rows x 64 pixels (halo 10 x 66 = 660 pixels) rather than 4 x 128 (6 x 130 = 780): the halo refill
  // at every 64-channel chunk boundary is the largest exposed cost of the decoder's short K loops (tools/halo_trace.py: ~5 us of a
  // 14.6 us chunk), and it scales with the halo's bytes.  tw >= 64 keeps a CF_STATS block (64 consecutive output pixels) inside one row.
  static const int tw512 = getenv("DD_HALO_TW512") ? atoi(getenv("DD_HALO_TW512")) : 64;
  const int tw = Wo < 128 ? Wo : (bm == 512 ? tw512 : 128), th = bm / tw;
  if (Ho % th) return false;
  int l = 0;
  while ((1 << l) < tw) ++l;
  g->ltw = l; g->th = th; g->halo_px = (th + 2) * (tw + 2); g->tiles_x = Wo / tw; g->tiles_y = Ho / th;
  static const int stagger = getenv("DD_HALO_STAGGER") ? atoi(getenv("DD_HALO_STAGGER")) : 0;
  g->stagger = stagger; g->tab = 0;
  return true;
}

template <int TN, int WN, bool MI = false>
hipError_t run_halo(const ConvGemmParams& p, const HaloGeo& g, hipStream_t stream) {
  constexpr int BM = (8 / WN) * 128, BN = WN * TN * 16;
  int lds = 2 * BN * 128 + ((g.halo_px + 7) & ~7) * 128 + BN * 4 + 512 + 64;
  // halo address table behind the bias / coef block when it fits
  static const int tab_on = getenv("DD_HALO_TAB") ? atoi(getenv("DD_HALO_TAB")) : 1;
  HaloGeo gg = g;
  const int tab_bytes = ((g.halo_px + 7) >> 3) * 256;
  gg.tab = (tab_on && lds + tab_bytes <= 163840) ? 1 : 0;
  if (gg.tab) lds += tab_bytes;
  static int attr = 0;
  if (attr < lds) { hipFuncSetAttribute((const void*)conv_halo_kernel<TN, WN, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = lds; }
  const int tiles = (p.M / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((conv_halo_kernel<TN, WN, MI>), dim3(tiles, MI ? p.ksplit : 1), dim3(512), lds, stream, p, gg);
  return hipGetLastError();
}

template <int TN, int WN>
hipError_t run_halo_persist(const ConvGemmParams& p, const HaloGeo& g, hipStream_t stream) {
  constexpr int BM = (8 / WN) * 128, BN = WN * TN * 16;
  const int lds = 2 * BN * 128 + ((g.halo_px + 7) & ~7) * 128 + 2 * BN * 4 + ((g.halo_px + 7) >> 3) * 256 + 64;
  static int attr = 0;
  if (attr < lds) { hipFuncSetAttribute((const void*)conv_halo_persist_kernel<TN, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = lds; }
  static const int cus = [] { int d = 0, n = 256; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 8 ? n & ~7 : 8; }();
  const int tiles = (p.M / BM) * (p.N / BN);
  const int grid = tiles >= cus ? cus : (tiles + 7) & ~7;
  hipLaunchKernelGGL((conv_halo_persist_kernel<TN, WN>), dim3(grid), dim3(512), lds, stream, p, g);
  return hipGetLastError();
}

}  // namespace

// chunk split of the 8 x 8 level (M = 64 pixels x images: 64 tiles of 256 x 320 at 64 images -- a quarter of the chip): the smallest
// power of two that gives >= 192 workgroups -- halved until it DIVIDES the chunk count (the kernel hands ceil(chunks / split) chunks to
// every blockIdx.y: with a remainder the last workgroups would start behind the last chunk; Cin = 2560 at 16 images: 16 -> 8) -- with
// every workgroup >= 2 chunks; 1 = this is not that case.  Needs the split-K scratch.
int conv_halo_split(const ConvGemmParams& p) {
  static const int on = getenv("DD_HALO_8X8") ? atoi(getenv("DD_HALO_8X8")) : 1;
  if (!on || p.Wo != 8 || p.Ho != 8 || p.H != 8 || p.W != 8 || p.shift || p.stride != 1 || (p.B & 3) || p.N % 320 || !p.partial) return 1;
  if (p.flags & ~(CF_BIAS | CF_RES | CF_RELU)) return 1;
  const int tiles = (p.M / 256) * (p.N / 320), chunks = p.cin >> 6;
  int s = 1;
  while (tiles * s < 192 && s < 16) s *= 2;
  while (s > 1 && (chunks % s || chunks / s < 2)) s >>= 1;
  return s;
}

// 0 = not eligible, else the tile form: 5 = 256 x 320, 4 = 256 x 256, 6 = 512 x 160, 2 = 512 x 128 (N = 128), 1 = 512 x 32 (narrow: N <= 4).  The host reads the tap table once per weight tensor elsewhere: here the
// caller guarantees a 3x3 / pad 1 table (ntaps == 9 with offsets in {-1, 0, 1}^2), which every packer emits for KH = KW = 3, pad = 1.
int conv_halo_config(const ConvGemmParams& p) {
  static const int on = getenv("DD_CONV_HALO") ? atoi(getenv("DD_CONV_HALO")) : 1;
  if (!on || p.force_small) return 0;
  if (p.ntaps != 9 || p.stride != 1 || p.shift > 1 || p.parity || (p.H << p.shift) != p.Ho || (p.W << p.shift) != p.Wo || (p.cin & 63) ||
      p.K != 9 * p.cin) return 0;
  // narrow outputs (conv_out of the decoder / the UNet, N <= 4): a 512 x 32 form whose weight stage is 4 KB -- the input tile is read once
  // from HBM (halo) instead of being gathered tap by tap through the 128-wide tiles of the general kernels (60 of 64 columns wasted)
  static const int narrow_on = getenv("DD_HALO_NARROW") ? atoi(getenv("DD_HALO_NARROW")) : 1;
  const bool narrow = narrow_on && p.N <= 4 && !(p.flags & ~(CF_BIAS | CF_OUT_F32 | CF_GNFOLD)) && !p.bias_sel && !p.shift && p.ksplit <= 1;
  if (!narrow && ((p.flags & ~(CF_BIAS | CF_RES | CF_RELU | CF_STATS | CF_GNFOLD)) || p.bias_sel)) return 0;
  if ((p.flags & CF_GNFOLD) && (!p.gn_coef || p.shift)) return 0;
  // tile forms by preference: 512 x 160 / 512 x 128 (the halo-resident input is cheap, the streamed weights are not: 20 / 16 KB of
  // weights + ~10 KB of halo per K-step instead of 40 / 32 + 5.6) where the image geometry allows 512-pixel tiles, else 256 x 320 / 256 x 256
  static const int tall = getenv("DD_HALO_TALL") ? atoi(getenv("DD_HALO_TALL")) : 1;
  if ((!narrow && (p.y_ld & 7)) || ((p.flags & CF_RES) && (p.res_ld & 7)) || (p.x_ld & 7) || p.alpha != 1.f) return 0;
  if ((size_t)p.H * p.W * (size_t)p.x_ld * 2 >= 0xF0000000ull) return 0;         // byte offsets are per image
  if (p.M != p.B * p.Ho * p.Wo) return 0;
  if (narrow) {
    HaloGeo g;
    if (p.N == 4 && (p.flags & CF_OUT_F32) && (p.y_ld & 3)) return 0;             // float4 stores
    if (!halo_geometry(p, 512, &g) || p.M / 512 < 192) return 0;
    if (2 * 32 * 128 + ((g.halo_px + 7) & ~7) * 128 + 32 * 4 + 512 + 64 > 163840) return 0;
    return 1;
  }
  if (conv_halo_split(p) > 1) return 5;                  // 8 x 8 level: 256 x 320 multi-image tiles + chunk split (fp32 partials)
  if (p.ksplit > 1) return 0;
  const int forms[4][3] = {{6, 512, 160}, {2, 512, 128}, {5, 256, 320}, {4, 256, 256}};
  for (int f = 0; f < 4; ++f) {
    const int tn = forms[f][0], bm = forms[f][1], bn = forms[f][2];
    if (p.N % bn) continue;
    if (bm == 512 && !tall && p.N != 128) continue;
    if (tn == 2 && p.N % 320 == 0) continue;            // 320-multiples: 160-wide tiles
    HaloGeo g;
    if (!halo_geometry(p, bm, &g) || g.ipt > 1) continue;   // multi-image tiles (8 x 8) exist in the chunk-split form only (above)
    if ((p.M / bm) * (p.N / bn) < 192) continue;        // needs (most of) the chip: small grids keep the split-K forms
    if (2 * bn * 128 + ((g.halo_px + 7) & ~7) * 128 + bn * 4 + 512 + 64 > 163840) continue;
    return tn;
  }
  return 0;
}

// pointwise ping-pong GEMM: 0 = not eligible, else TN (5: 256 x 320 tiles; 4: 256 x 256 tiles of a GEGLU projection)
int gemm_pp_config(const ConvGemmParams& p) {
  static const int on = getenv("DD_GEMM_PP") ? atoi(getenv("DD_GEMM_PP")) : 1;
  static const int geglu_on = getenv("DD_GEMM_PP_GEGLU") ? atoi(getenv("DD_GEMM_PP_GEGLU")) : 1;
  static const int nmax = getenv("DD_GEMM_PP_NMAX") ? atoi(getenv("DD_GEMM_PP_NMAX")) : 3840;
  if (!on || p.force_small) return 0;
  if (p.ntaps != 1 || p.stride != 1 || p.shift || p.parity || p.H != p.Ho || p.W != p.Wo || (p.cin & 63) || p.K != p.cin) return 0;
  if ((p.M & 255) || p.K < 256 || p.ksplit > 1 || p.bias_sel || (p.x_ld & 7) || (p.y_ld & 7)) return 0;
  if ((size_t)256 * p.x_ld * 2 >= 0xF0000000ull || (size_t)p.N * p.K * 2 >= 0xF0000000ull) return 0;
  if (p.flags & CF_GEGLU) {
    if (!geglu_on) return 0;
    if (p.flags & ~(CF_BIAS | CF_GEGLU | CF_GEGLU_RAW | CF_LNFOLD)) return 0;
    if ((p.N & 255) || ((p.flags & CF_GEGLU_RAW) && (p.raw_ld & 7))) return 0;
    if ((p.M / 256) * (p.N / 256) < 192) return 0;
    return 4;
  }
  if (p.flags & ~(CF_BIAS | CF_RES | CF_RELU | CF_STATS | CF_ROWSTATS | CF_LNFOLD)) return 0;
  if (p.N % 320 || p.N > nmax) return 0;
  if ((p.flags & CF_RES) && (p.res_ld & 7)) return 0;
  if ((p.M / 256) * (p.N / 320) < 192) return 0;
  return 5;
}
template <int TN, bool GEGLU>
static hipError_t run_pps(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * TN;
  const int lds = 2 * (256 + BN) * 128 + 2 * 6144;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)gemm_pps_kernel<TN, GEGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  static const int cus = [] { int d = 0, n = 256; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 8 ? n & ~7 : 8; }();
  const int tiles = (p.M / 256) * (p.N / BN);
  const int grid = tiles >= cus ? cus : (tiles + 7) & ~7;
  hipLaunchKernelGGL((gemm_pps_kernel<TN, GEGLU>), dim3(grid), dim3(512), lds, stream, p);
  return hipGetLastError();
}
hipError_t launch_gemm_pp(const ConvGemmParams& p, int tn, hipStream_t stream) {
  return tn == 4 ? run_pps<4, true>(p, stream) : run_pps<5, false>(p, stream);
}

#ifdef DD_TRACE
extern "C" int dd_debug_read_pp_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pp_trace), sizeof(unsigned long long) * n);
}
#endif

hipError_t launch_conv_halo(const ConvGemmParams& p, int tn, hipStream_t stream) {
  HaloGeo g;
  if (!halo_geometry(p, tn == 2 || tn == 6 || tn == 1 ? 512 : 256, &g)) return hipErrorInvalidValue;
  if (tn == 1) return run_halo<1, 2>(p, g, stream);
  // 512 x 128 tiles with a short K loop (the decoder's levels): the persistent form, next tile's first stage requested under the epilogue
  static const int persist = getenv("DD_HALO_PERSIST") ? atoi(getenv("DD_HALO_PERSIST")) : 1;
  // (address table: halo row < 31, column < 256, byte offsets inside the halo's stored rows below 8 MB)
  if (tn == 2 && persist && g.ipt == 1 && !(p.flags & CF_GNFOLD) && p.N % 128 == 0 && (p.cin >> 6) <= persist * 8 &&
      g.th + 2 < 31 && (1 << g.ltw) + 2 < 256 && (size_t)(g.th + 3) * p.W * p.x_ld * 2 < (8u << 20) &&
      2 * 128 * 128 + ((g.halo_px + 7) & ~7) * 128 + 2 * 128 * 4 + ((g.halo_px + 7) >> 3) * 256 + 64 <= 163840)
    return run_halo_persist<4, 2>(p, g, stream);
  if (g.ipt > 1) return (tn == 5 && p.ksplit > 1 && (p.cin >> 6) % p.ksplit == 0) ? run_halo<5, 4, true>(p, g, stream) : hipErrorInvalidValue;
  return tn == 5 ? run_halo<5, 4>(p, g, stream) : tn == 4 ? run_halo<4, 4>(p, g, stream) : tn == 6 ? run_halo<5, 2>(p, g, stream) : run_halo<4, 2>(p, g, stream);
}
