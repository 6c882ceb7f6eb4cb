// K2 (short-K pointwise layers, K = 320: the transformer blocks of the 64x64 level -- fused QKV, to_q, to_out, proj_in / proj_out and the
// GEGLU projection; SURVEY.md 8a row A2, diffusers BasicTransformerBlock / Transformer2DModel) -- WEIGHT-STATIONARY GEMM, bf16 MFMA, gfx950.
//
// Why (DESIGN.md "short-K GEMMs"): with K = 320 a tile of the streaming ping-pong GEMM (conv_halo.hip, gemm_pps_kernel) is 5 K-steps
// long; every K-step waits on an LDS-DMA stage that was requested one K-step earlier (in-order vmcnt: nothing older may stay in flight),
// so HBM-fresh activation rows cost their full latency per K-step, and the epilogue (HBM stores, residual loads, the erf of GEGLU) runs
// with nothing beside it.  Here the roles are split so that NO wave ever waits for a fast load behind a slow one:
//   * the weights of a workgroup's column block do not move at all: wave (wc, kh) keeps W[80 (64) columns][its 160-deep K half] in 100
//     (80) VGPRs for the whole kernel; there is no weight staging, no weight LDS traffic, no per-K-step barrier;
//   * only activations stream: 64-row tiles (40 KB) through a two-slot LDS ring, requested a whole tile ahead by LDS-DMA
//     (buffer_load ... lds), by the kh = 0 waves only -- these waves never store, so their vmcnt queue holds nothing else;
//   * the two waves of a SIMD (w, w + 4) split K: per 16-row step the kh = 0 wave ("P") hands its partial 16 x 80 fp32 tile to the
//     kh = 1 wave ("E") through LDS; E adds its own half and runs the epilogue (bias / LayerNorm fold / residual / row statistics /
//     GEGLU / stores) while P's MFMAs of the next step run on the same SIMD: epilogue and MFMA overlap by construction, and E's
//     queue holds only residual loads and stores.
// One barrier per 16-row step.  Same packed weights ([N][K], GEGLU (16 hidden | 16 gate) packing), NHWC row layout, flags and epilogue
// semantics as gemm_pps_kernel; accumulation is fp32 in two K halves (kh = 0, then kh = 1 added to it).
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ void wdma16(const void* base, void* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffff00u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int WS_RT = 64;        // rows per tile
constexpr int WS_KH = 5;         // 32-deep K chunks per K half (K = 320)
constexpr int WS_TILE_B = WS_RT * 640;   // bytes of one activation tile in LDS: 5 blocks of [64 rows][128 B]

// TN: 16-column MFMA tiles per wave (5: 80 columns, BN = 320; 4: 64 columns, BN = 256, GEGLU).  CB column blocks; RL row lanes per XCD.
// RES: CF_RES launches (their residual loads are inline asm with counted waits: E's queue then holds, in order, the residual loads of
// three steps and the stores of two, and a step never waits for anything younger than the loads it is about to use).
template <int TN, bool GEGLU, bool RES>
__global__ __launch_bounds__(512, 1) void gemm_ws_kernel(ConvGemmParams p, int CB, int RL) {
  constexpr int BN = 4 * TN * 16;
  constexpr int PART = 2 * WS_TILE_B;                    // partial exchange: [slot 2][pair 4][TN][64 lanes] x 16 B
  constexpr int STAT = PART + 2 * 4 * TN * 1024;         // CF_LNFOLD: (mean, rstd) of the tile's 64 rows, [slot 2][1 KB] (512 B used)
  constexpr int AUXO = STAT + 2048;                      // bias [BN] | c1 [BN] fp32
  constexpr int K = 320;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, kh = wave >> 2;
  const int fr = lane & 15, fq = lane >> 4;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int cb = j % CB, rl = j / CB;
  if (rl >= RL) return;
  const int n0 = cb * BN;
  const int ntiles = p.M / WS_RT;
  // row tiles of this workgroup: (i * RL + rl) * 8 + xcd -- the CB workgroups of one row lane (same XCD: one L2) walk the same tiles
  const int tstep = RL * 8, tfirst = rl * 8 + xcd;
  if (tfirst >= ntiles) return;
  const int count = (ntiles - tfirst + tstep - 1) / tstep;
  const int fl = p.flags;

  // ---- weights of (wc, kh) into registers: MFMA A operand, row fr of 16-column tile jn = packed weight row chan(jn, fr), k = kc * 32 + fq * 8 ..
  bf16x8 wreg[TN][WS_KH];
  {
    constexpr int TNP = TN & ~1;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      int ch;
      if (GEGLU) {
        const int c = (fr >> 2) * 8 + (jn >> 1) * 4 + (fr & 3);          // output column inside the wave's 32
        ch = wc * 64 + (c >> 4) * 32 + (jn & 1) * 16 + (c & 15);
      } else {
        ch = jn < TNP ? wc * (TN * 16) + (jn >> 1) * 32 + (fr >> 2) * 8 + (jn & 1) * 4 + (fr & 3) : wc * (TN * 16) + jn * 16 + fr;
      }
      const bf16_t* wp = p.w + (size_t)(n0 + ch) * K + kh * 160 + fq * 8;
#pragma unroll
      for (int kc = 0; kc < WS_KH; ++kc) wreg[jn][kc] = *(const bf16x8*)(wp + kc * 32);
    }
  }
  float* const bias_s = (float*)(smem + AUXO);
  float* const c1_s = bias_s + BN;
  if (tid < BN) {
    bias_s[tid] = (fl & CF_BIAS) ? p.bias[n0 + tid] : 0.f;
    c1_s[tid] = (fl & CF_LNFOLD) ? p.ln_c1[n0 + tid] : 0.f;
  }

  // ---- activation staging (kh = 0 waves): tile = 40 pieces of 8 rows x 128 B; wave wc moves pieces wc, wc + 4, ...
  const int prow = lane >> 3, jx = (lane & 7) ^ prow;
  auto issue_a = [&](const bf16_t* xt, int slot, int q) {      // piece q = (64-channel block q >> 3, rows (q & 7) * 8 ..)
    const unsigned voff = ((unsigned)((q & 7) * 8 + prow) * (unsigned)p.x_ld + (unsigned)(jx * 8)) * 2u;
    wdma16(xt, smem + slot * WS_TILE_B + q * 1024, voff, (unsigned)(q >> 3) * 128u);
  };
  // CF_LNFOLD: the tile's 64 (mean, rstd) pairs = 512 B arrive with it (lanes 32 .. 63 read out of range: zeros)
  const unsigned statoff = ((fl & CF_LNFOLD) && lane < 32) ? (unsigned)lane * 16u : 0xfffffff0u;
  auto issue_stats = [&](int t, int slot) {
    wdma16((fl & CF_LNFOLD) ? (const void*)(p.ln_stats + (size_t)t * WS_RT * 2) : (const void*)p.w, smem + STAT + slot * 1024, statoff, 0u);
  };
  int tile = tfirst;
  if (kh == 0) {
    const bf16_t* xt = p.x + (size_t)tile * WS_RT * p.x_ld;
#pragma unroll
    for (int i = 0; i < 10; ++i) issue_a(xt, 0, wc + 4 * i);
    if (wc == 0) issue_stats(tile, 0);
  }
  __syncthreads();      // (vmcnt(0) + barrier: tile 0 landed, bias_s / c1_s published)

  // A fragment (MFMA B operand) of 16-row step a, K chunk sc (0 .. 9) of the tile in `slot`
  auto xfrag = [&](int slot, int a, int sc) {
    const int row = a * 16 + fr;
    return *(const bf16x8*)(smem + slot * WS_TILE_B + (sc >> 1) * 8192 + row * 128 + ((((sc & 1) * 4 + fq) ^ (row & 7)) << 4));
  };

  if (kh == 0) {
    // =========================================== P: K half 0, hands its partial sums to E ===========================================
    for (int it = 0; it < count; ++it, tile += tstep) {
      const int slot = it & 1;
      const bool have_next = it + 1 < count;
      const bf16_t* xtn = p.x + (size_t)(tile + tstep) * WS_RT * p.x_ld;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        bf16x8 xf[WS_KH];
#pragma unroll
        for (int kc = 0; kc < WS_KH; ++kc) xf[kc] = xfrag(slot, a, kc);
        if (have_next) {
          // the other slot was read for the last time before the previous tile's last barrier
          if (a == 0) {
            issue_a(xtn, slot ^ 1, wc); issue_a(xtn, slot ^ 1, wc + 4); issue_a(xtn, slot ^ 1, wc + 8); issue_a(xtn, slot ^ 1, wc + 12);
            if (wc == 0) issue_stats(tile + tstep, slot ^ 1);
          }
          if (a == 1) { issue_a(xtn, slot ^ 1, wc + 16); issue_a(xtn, slot ^ 1, wc + 20); issue_a(xtn, slot ^ 1, wc + 24); }
          if (a == 2) { issue_a(xtn, slot ^ 1, wc + 28); issue_a(xtn, slot ^ 1, wc + 32); issue_a(xtn, slot ^ 1, wc + 36); }
        }
        f32x4 acc[TN];
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < WS_KH; ++kc)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[jn][kc], xf[kc], acc[jn], 0, 0, 0);
        unsigned char* pb = smem + PART + (((a & 1) * 4 + wc) * TN) * 1024 + lane * 16;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) *(f32x4*)(pb + jn * 1024) = acc[jn];
        // the next tile's pieces were requested at least one step ago: landed before the barrier that opens the next tile
        if (a == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // ============================================= E: K half 1 + partial of P + epilogue ==============================================
  const int wb = n0 + wc * (TN * 16);                    // first output column (packed column for GEGLU) of this wave
  const float* bw = bias_s + wc * (TN * 16);
  const float* cw = c1_s + wc * (TN * 16);
  // Residual rows (RES): requested TWO steps ahead with inline-asm loads, so the compiler's waitcnt pass does not drain the queue at
  // their use; the wait in front of a step's epilogue is counted by hand.  Queue of an E wave, in issue order:
  //   ... loads(s) | stores(s-2) | loads(s+1) | stores(s-1) | loads(s+2) | [wait for loads(s)] stores(s) ...
  // NLD loads and at least NST stores per step (CF_ROWSTATS adds one more store: the wait is then stricter by two old stores), so
  // everything younger than loads(s) is 2 * (NLD + NST) operations.
  constexpr int NP = TN / 2;                             // 16-byte column pairs per row
  constexpr int NLD = NP + (TN & 1), NST = NP + (TN & 1);
  u32x4 rq[4][NP > 0 ? NP : 1];
  u32x2 ro[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int t = 0; t < (NP > 0 ? NP : 1); ++t) rq[b][t] = u32x4{0, 0, 0, 0};
    ro[b] = u32x2{0, 0};
  }
  auto res_issue = [&](int b, int m) {
    if constexpr (RES) {
      const bf16_t* rp = (const bf16_t*)p.res + (size_t)m * p.res_ld + wb;
#pragma unroll
      for (int t = 0; t < NP; ++t) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rq[b][t]) : "v"(rp + t * 32 + fq * 8) : "memory");
      if constexpr ((TN & 1) != 0) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ro[b]) : "v"(rp + (TN - 1) * 16 + fq * 4) : "memory");
    }
  };
  // steps 0 and 1 of the first tile
  res_issue(0, tile * WS_RT + fr);
  res_issue(1, tile * WS_RT + 16 + fr);
  for (int it = 0; it < count; ++it, tile += tstep) {
    const int slot = it & 1;
    const int m0 = tile * WS_RT;
    const int m0n = it + 1 < count ? (tile + tstep) * WS_RT : m0;     // (the last tile requests rows of its own again: never used)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bf16x8 xf[WS_KH];
#pragma unroll
      for (int kc = 0; kc < WS_KH; ++kc) xf[kc] = xfrag(slot, a, WS_KH + kc);
      res_issue((a + 2) & 3, a < 2 ? m0 + (a + 2) * 16 + fr : m0n + (a - 2) * 16 + fr);
      f32x4 acc[TN];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < WS_KH; ++kc)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[jn][kc], xf[kc], acc[jn], 0, 0, 0);
      __builtin_amdgcn_s_barrier();
      const unsigned char* pb = smem + PART + (((a & 1) * 4 + wc) * TN) * 1024 + lane * 16;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const f32x4 pv = *(const f32x4*)(pb + jn * 1024);
        acc[jn] = pv + acc[jn];
      }
      // ---- epilogue of rows m0 + a * 16 + fr
      const int m = m0 + a * 16 + fr;
      float2 lc = make_float2(0.f, 1.f);
      if (fl & CF_LNFOLD) lc = *(const float2*)(smem + STAT + slot * 1024 + (a * 16 + fr) * 8);
      const float rs = lc.y * p.alpha, nm = -lc.y * lc.x;      // CF_LNFOLD: rstd and -rstd * mean of this lane's row (1, 0 otherwise)
      if constexpr (GEGLU) {
        // a lane holds hidden (tiles 0, 2) and gate (tiles 1, 3) pre-activations of 8 consecutive output columns
        const int pk = (fq >> 1) * 32 + (fq & 1) * 8;            // packed column (inside the wave's 64) of the lane's first hidden value
        float h[8], g[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4 bh = *(const float4*)(bw + pk + t * 4), bg = *(const float4*)(bw + pk + 16 + t * 4);
          const float4 ch = *(const float4*)(cw + pk + t * 4), cg = *(const float4*)(cw + pk + 16 + t * 4);
          const float bhv[4] = {bh.x, bh.y, bh.z, bh.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
          const float chv[4] = {ch.x, ch.y, ch.z, ch.w}, cgv[4] = {cg.x, cg.y, cg.z, cg.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            h[t * 4 + r] = __builtin_fmaf(rs, acc[2 * t][r], __builtin_fmaf(nm, chv[r], bhv[r]));
            g[t * 4 + r] = __builtin_fmaf(rs, acc[2 * t + 1][r], __builtin_fmaf(nm, cgv[r], bgv[r]));
          }
        }
        if (fl & CF_GEGLU_RAW) {
          bf16_t* rp = p.raw + (size_t)m * p.raw_ld + wb + pk;
          *(uint4*)rp = pack8(h);
          *(uint4*)(rp + 16) = pack8(g);
        }
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = h[e] * gelu_f(g[e]);
        *(uint4*)((bf16_t*)p.y + (size_t)m * p.y_ld + (n0 >> 1) + wc * 32 + fq * 8) = pack8(o);
      } else {
        bf16_t* yp = (bf16_t*)p.y + (size_t)m * p.y_ld + wb;
        float r1 = 0.f, r2 = 0.f;                            // CF_ROWSTATS
        if constexpr (RES) {
          // the residual rows of this step: everything up to them has returned once at most 2 * (NLD + NST) younger operations are pending
          if constexpr ((TN & 1) != 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(rq[a][0]), "+v"(ro[a]) : "n"(2 * (NLD + NST)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rq[a][0]) : "n"(2 * (NLD + NST)) : "memory");
#pragma unroll
          for (int t = 1; t < NP; ++t) asm volatile("" : "+v"(rq[a][t]));
        }
        auto four = [&](const f32x4& v, int col, unsigned q0, unsigned q1) {
          float4 b = *(const float4*)(bw + col);
          if (fl & CF_LNFOLD) {
            const float4 c = *(const float4*)(cw + col);
            b.x = __builtin_fmaf(nm, c.x, b.x); b.y = __builtin_fmaf(nm, c.y, b.y); b.z = __builtin_fmaf(nm, c.z, b.z); b.w = __builtin_fmaf(nm, c.w, b.w);
          }
          float v0 = __builtin_fmaf(v[0], rs, b.x), v1 = __builtin_fmaf(v[1], rs, b.y), v2 = __builtin_fmaf(v[2], rs, b.z), v3 = __builtin_fmaf(v[3], rs, b.w);
          if constexpr (RES) {
            v0 += __uint_as_float(q0 << 16); v1 += __uint_as_float(q0 & 0xffff0000u);
            v2 += __uint_as_float(q1 << 16); v3 += __uint_as_float(q1 & 0xffff0000u);
          }
          if (fl & CF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          if (fl & CF_ROWSTATS) {
            r1 += (v0 + v1) + (v2 + v3);
            r2 = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, r2))));
          }
          return make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        };
#pragma unroll
        for (int t = 0; t < NP; ++t) {
          const int col = t * 32 + fq * 8;
          const uint2 lo = four(acc[2 * t], col, rq[a][t].x, rq[a][t].y);
          const uint2 hi = four(acc[2 * t + 1], col + 4, rq[a][t].z, rq[a][t].w);
          *(uint4*)(yp + col) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        if constexpr ((TN & 1) != 0) {
          const int col = (TN - 1) * 16 + fq * 4;
          *(uint2*)(yp + col) = four(acc[TN - 1], col, ro[a].x, ro[a].y);
        }
        if (fl & CF_ROWSTATS) {
          // the row's TN * 16 columns of this wave sit in lanes fr, fr + 16, fr + 32, fr + 48
          r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
          r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
          if (fq == 0) *(float2*)(p.rowpart + ((size_t)m * p.rowpart_ld + cb * 4 + wc) * 2) = make_float2(r1, r2);
        }
      }
    }
  }
  if constexpr (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the two look-ahead requests behind the last step
}

template <int TN, bool GEGLU, bool RES>
hipError_t run_ws(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * TN;
  const int lds = 2 * WS_TILE_B + 2 * 4 * TN * 1024 + 2048 + 2 * BN * 4;
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)gemm_ws_kernel<TN, GEGLU, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  static const int cus = [] { int d = 0, n = 256; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 8 ? n & ~7 : 8; }();
  const int per = cus / 8;                               // workgroups per XCD label
  const int CB = p.N / BN, RL = per / CB;
  hipLaunchKernelGGL((gemm_ws_kernel<TN, GEGLU, RES>), dim3(cus), dim3(512), lds, stream, p, CB, RL);
  return hipGetLastError();
}

}  // namespace

// weight-stationary GEMM for K = 320 pointwise layers: 0 = not eligible, 5: 320-column blocks, 4: 256-column blocks (GEGLU)
int gemm_ws_config(const ConvGemmParams& p) {
  static const int on = getenv("DD_GEMM_WS") ? atoi(getenv("DD_GEMM_WS")) : 1;
  static const int mmin = getenv("DD_GEMM_WS_MMIN") ? atoi(getenv("DD_GEMM_WS_MMIN")) : 32768;
  if (!on || p.force_small) return 0;
  if (p.ntaps != 1 || p.stride != 1 || p.shift || p.parity || p.H != p.Ho || p.W != p.Wo || p.cin != 320 || p.K != 320) return 0;
  if ((p.M & 63) || p.M < mmin || p.ksplit > 1 || p.bias_sel || (p.x_ld & 7) || (p.y_ld & 7)) return 0;
  if ((size_t)64 * p.x_ld * 2 >= 0xF0000000ull) return 0;
  static const int cus = [] { int d = 0, n = 256; hipGetDevice(&d); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 8 ? n & ~7 : 8; }();
  const int per = cus / 8;
  if (p.flags & CF_GEGLU) {
    if (p.flags & ~(CF_BIAS | CF_GEGLU | CF_GEGLU_RAW | CF_LNFOLD)) return 0;
    if ((p.N & 255) || p.N / 256 > per || ((p.flags & CF_GEGLU_RAW) && (p.raw_ld & 7))) return 0;
    return 4;
  }
  if (p.flags & ~(CF_BIAS | CF_RES | CF_RELU | CF_ROWSTATS | CF_LNFOLD)) return 0;
  if (p.N % 320 || p.N / 320 > per) return 0;
  if ((p.flags & CF_RES) && (p.res_ld & 7)) return 0;
  return 5;
}
hipError_t launch_gemm_ws(const ConvGemmParams& p, int tn, hipStream_t stream) {
  return tn == 4 ? run_ws<4, true, false>(p, stream) : (p.flags & CF_RES) ? run_ws<5, false, true>(p, stream) : run_ws<5, false, false>(p, stream);
}
