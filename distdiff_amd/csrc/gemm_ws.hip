// K2 (short-K pointwise layers, K = 320: the transformer blocks of the 64x64 level -- fused QKV, to_q, proj_in and the GEGLU projection;
// SURVEY.md 8a row A2, diffusers BasicTransformerBlock / Transformer2DModel) -- WEIGHT-STATIONARY GEMM, bf16 MFMA, gfx950.
//
// Why (DESIGN.md "short-K GEMMs"): with K = 320 a tile of the streaming ping-pong GEMM (conv_halo.hip, gemm_pps_kernel) is 5 K-steps
// long; every K-step waits on an LDS-DMA stage that was requested one K-step earlier (vmcnt retires in order: nothing older may stay
// in flight), so HBM-fresh activation rows cost their full latency per K-step, and the epilogue (HBM stores, the erf of GEGLU) runs with
// nothing beside it.  Here nothing fast ever waits behind something slow:
//   * the weights do not move at all: each of the 8 waves keeps W[its 48 / 32 output columns][all 320 K] in 120 / 80 VGPRs for the
//     whole kernel (a SIMD's two waves share an 80-column span as 3 + 2 MFMA tiles; GEGLU: 2 + 2 tiles of a 64-column span).  No weight
//     staging, no weight LDS traffic, no K split, no partial sums;
//   * only activations stream: 64-row tiles (40 KB) through a THREE-slot LDS ring, each requested two tiles ahead by LDS-DMA
//     (buffer_load ... lds, 5 pieces per wave) -- 80 KB per CU in flight, enough to cover the HBM latency at full rate;
//   * ONE barrier per 64-row tile (ring hand-over), and the two waves of a SIMD arrive at it one phase apart (waves 0 .. 3 behind the
//     MFMAs of the tile's last 16-row step, waves 4 .. 7 in front of theirs), so that one runs an epilogue (bias / LayerNorm fold / row
//     statistics / GEGLU / stores) while the other runs MFMAs; fragments are read one step ahead;
//   * the LDS-DMA requests are inline asm and every wait on them is counted by hand: hipcc's waitcnt pass treats an in-flight LDS-DMA
//     builtin as a pending LDS write that later ds_reads may alias and drains the whole queue (vmcnt(0)) in front of them.
// Same packed weights ([N][K], GEGLU (16 hidden | 16 gate) packing), NHWC row layout, flags and epilogue semantics as gemm_pps_kernel.
// NOT taken: launches with a residual (to_out, proj_out).  Their residual rows have to be requested two steps ahead to cover HBM latency;
// as inline-asm register loads the compiler is free to copy the destination registers before the hand-counted wait (it did: v_mov of the
// stale registers in front of s_waitcnt, non-finite rows in the engine while the op-level tests passed), as ordinary loads its own waits
// drain the LDS-DMA queue.  The way in is an LDS destination (LDS-DMA of the residual tile, 30 KB), not built.
// History: commit 7d952d8 is the first form (K split over the two waves of a SIMD, fp32 partial hand-off through LDS, a barrier per
// 16-row step): parity green, slower than the ping-pong GEMM -- the finishing wave's epilogue serialised behind its own MFMAs.
// tools/ws_trace.py (-DWS_TRACE stamps) and the WS_ABL ablations are what the current form was derived from (profiles/r05_ws_*.txt).
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA (buffer_load_dwordx4 ... lds: 16 bytes per lane straight into LDS at M0 + lane * 16) as inline asm: hipcc's waitcnt pass
// treats every in-flight LDS-DMA builtin as a pending LDS write that any later ds_read may alias and puts s_waitcnt vmcnt(0) in front
// of the first such read (found in the ISA of the builtin form: one full drain of the queue per tile, in every wave); an asm request
// is invisible to it, and every wait on these requests is counted by hand below.
__device__ __forceinline__ i32x4 wrsrc(const void* base) {
  const unsigned long long a = (unsigned long long)base;
  // (readfirstlane: the descriptor must sit in SGPRs whatever the compiler thinks of the pointer's uniformity)
  return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)), (int)0xffffff00u, 0x00020000};
}
__device__ __forceinline__ void wdma16(const i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef WS_TRACE
// debug build only (tools/ws_trace.py): s_memtime stamps of workgroup 0, [wave 8][tile 16][stamp 10]: tile start, after the DMA issue,
// then per 16-row step: after its MFMAs (+ next fragment reads / ring hand-over), after its epilogue
__device__ unsigned long long g_ws_trace[8 * 16 * 10];
#define WS_STAMP(it, i) do { if (blockIdx.x == 0 && (it) < 16 && (threadIdx.x & 63) == 0) g_ws_trace[((threadIdx.x >> 6) * 16 + (it)) * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(it, i) do { } while (0)
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifndef WS_VAR
#define WS_VAR 1      // A/B switches of this file: 1 hand-built fragment addresses (374 -> 341 vector instructions per GEGLU tile and wave, 235 -> 182 VGPRs;
                      // isolated 581 -> 577 us: kept for the registers), 2 / 4 s_setprio 1 for waves 4 .. 7 / 0 .. 3 (no gain: profiles/r06_ws_variants.txt)
#endif
#ifndef WS_ABL
#define WS_ABL 0      // timing ablations (results wrong): 1 no DMA behind the first two tiles, 2 no epilogue arithmetic, 4 no MFMAs, 8 no fragment reads, 16 no stores, 32 no barrier
#endif
constexpr int WS_RT = 64;                 // rows per tile
constexpr int WS_KC = 10;                 // 32-deep K chunks (K = 320)
constexpr int WS_TILE_B = WS_RT * 640;    // bytes of one activation tile in LDS: 5 blocks of [64 rows][128 B]
constexpr int WS_SLOTS = 3;
constexpr int WS_STAT = WS_SLOTS * WS_TILE_B;     // CF_LNFOLD: (mean, rstd) of a tile's 64 rows, [slot][4 quarters of 1 KB, 128 B used]
constexpr int WS_AUX = WS_STAT + WS_SLOTS * 4096; // bias [BN] | c1 [BN] fp32
constexpr int WS_NDMA = 5;                // LDS-DMA activation pieces per wave and tile (waves 4 .. 7: + a quarter of the statistics)

// One wave: NTW 16-column MFMA tiles starting at column colw of the workgroup's column block (packed column for GEGLU).  wave = 0 .. 7;
// HS: this wave also brings a quarter of the tile's LayerNorm statistics (waves 4 .. 7).  RS (CF_ROWSTATS) / RAW (CF_GEGLU_RAW) are
// compile-time so that a step is one straight-line block: the scheduler then interleaves the epilogue of step s with the MFMAs of step
// s + 1 (the accumulators are renamed), which runtime flag tests between them prevent; bias / LayerNorm fold / ReLU are branch-free.
template <int NTW, bool GEGLU, bool RS, bool RAW, bool HS>
__device__ __forceinline__ void ws_wave(const ConvGemmParams& p, unsigned char* smem, int BN, int n0, int colw, int span, int wave,
                                        int tfirst, int tstep, int count) {
  constexpr int K = 320;
  constexpr int NP = GEGLU ? 0 : NTW / 2;              // 16-byte column pairs per row
  constexpr int ODD = GEGLU ? 0 : (NTW & 1);
  constexpr int NST = (GEGLU ? (RAW ? 3 : 1) : NP + ODD) + (RS ? 1 : 0);   // stores per step
  constexpr int ND = WS_NDMA + (HS ? 1 : 0);           // LDS-DMA requests per tile
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int fl = p.flags;
  const bool lnf = (fl & CF_LNFOLD) != 0;
  const float relu_floor = (fl & CF_RELU) ? 0.f : -INFINITY;
  const float* bw = (const float*)(smem + WS_AUX) + colw;
  const float* cw = bw + BN;

  // ---- weights into registers: MFMA A operand, row fr of tile jn = packed weight row n0 + chan(jn, fr), k = kc * 32 + fq * 8 ..
  bf16x8 wreg[NTW][WS_KC];
#pragma unroll
  for (int jn = 0; jn < NTW; ++jn) {
    // non-GEGLU tiles are paired so that a lane's 4 + 4 accumulator rows of a pair are 8 consecutive channels (16-byte stores)
    const int ch = (!GEGLU && jn < 2 * NP) ? colw + (jn >> 1) * 32 + (fr >> 2) * 8 + (jn & 1) * 4 + (fr & 3) : colw + jn * 16 + fr;
    const bf16_t* wp = p.w + (size_t)(n0 + ch) * K + fq * 8;
#pragma unroll
    for (int kc = 0; kc < WS_KC; ++kc) wreg[jn][kc] = *(const bf16x8*)(wp + kc * 32);
  }

  // ---- activation staging: tile = 40 pieces of 8 rows x 128 B, wave w moves pieces w, w + 8, ...; waves 4 .. 7 + 16 rows of statistics
  const int prow = lane >> 3, jx = (lane & 7) ^ prow;
  const int lw = wave & 3;
  const unsigned statoff = (lnf && lane < 8) ? (unsigned)(lw * 128 + lane * 16) : 0xfffffff0u;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_off(smem));
  auto issue_tile = [&](int t, int slot) {
    const i32x4 rx = wrsrc(p.x + (size_t)t * WS_RT * p.x_ld);
#pragma unroll
    for (int i = 0; i < WS_NDMA; ++i) {
      const int q = wave + 8 * i;                        // piece q = (64-channel block q >> 3 = i, rows (q & 7) * 8 ..)
      const unsigned voff = ((unsigned)((q & 7) * 8 + prow) * (unsigned)p.x_ld + (unsigned)(jx * 8)) * 2u;
      wdma16(rx, lds0 + slot * WS_TILE_B + q * 1024, voff, (unsigned)(q >> 3) * 128u);
    }
    // 16 rows x (mean, rstd) = 128 B in lanes 0 .. 7 (the other lanes read out of range: zeros into the rest of this wave's own 1 KB)
    if constexpr (HS) wdma16(wrsrc(lnf ? (const void*)(p.ln_stats + (size_t)t * WS_RT * 2) : (const void*)p.w), lds0 + WS_STAT + slot * 4096 + lw * 1024, statoff, 0u);
  };
  int tile = tfirst;
  issue_tile(tile, 0);
  if (count > 1) issue_tile(tile + tstep, 1);
  wait_vm<0>();
  __syncthreads();      // tiles 0 and 1 landed, bias / c1 published

  // bias / c1 of a GEGLU wave's 4 + 4 columns live in registers for the whole kernel
  float bhv[4], bgv[4], chv[4], cgv[4];
  if constexpr (GEGLU) {
    const float4 bh = *(const float4*)(bw + fq * 4), bg = *(const float4*)(bw + 16 + fq * 4);
    const float4 ch = *(const float4*)(cw + fq * 4), cg = *(const float4*)(cw + 16 + fq * 4);
    bhv[0] = bh.x; bhv[1] = bh.y; bhv[2] = bh.z; bhv[3] = bh.w; bgv[0] = bg.x; bgv[1] = bg.y; bgv[2] = bg.z; bgv[3] = bg.w;
    chv[0] = ch.x; chv[1] = ch.y; chv[2] = ch.z; chv[3] = ch.w; cgv[0] = cg.x; cgv[1] = cg.y; cgv[2] = cg.z; cgv[3] = cg.w;
  }

  // A fragments (MFMA B operand) of 16-row step a of the tile in `slot`: requested one step ahead, behind the previous step's MFMAs.
  // The sched_barrier lets VALU / SALU / MFMA cross (so the previous epilogue still interleaves with the next MFMAs) but no LDS or
  // memory instruction: left to itself the scheduler sinks each ds_read to just in front of its MFMA and every K chunk pays the LDS latency.
  bf16x8 xf[WS_KC];
#if WS_VAR & 1
  // the swizzle term depends on the lane only (row & 7 == fr & 7 for every 16-row step): two lane bases (even / odd K chunk), everything
  // else -- step, 64-channel block -- is an immediate offset of the ds_read (the compiler otherwise keeps ~12 address registers and
  // spends two integer instructions per fragment read on them)
  const unsigned xfe = (unsigned)(fr * 128 + ((fq ^ (fr & 7)) << 4)), xfo = (unsigned)(fr * 128 + (((4 + fq) ^ (fr & 7)) << 4));
#endif
  auto load_xf = [&](int slot, int a) {
    const int row = a * 16 + fr;
#if WS_VAR & 1
    typedef const __attribute__((address_space(3))) unsigned char* lds_cp;
    unsigned be = lds0 + (unsigned)slot * WS_TILE_B + xfe, bo = lds0 + (unsigned)slot * WS_TILE_B + xfo;
    asm volatile("" : "+v"(be), "+v"(bo));              // opaque: no strength reduction into a register per (step, parity)
    const lds_cp Ae = (lds_cp)(unsigned long long)be, Ao = (lds_cp)(unsigned long long)bo;
#pragma unroll
    for (int kc = 0; kc < WS_KC; ++kc) xf[kc] = *(const __attribute__((address_space(3))) bf16x8*)(((kc & 1) ? Ao : Ae) + a * 2048 + (kc >> 1) * 8192);
#else
    const unsigned char* At = smem + slot * WS_TILE_B + row * 128;
#pragma unroll
    for (int kc = 0; kc < WS_KC; ++kc) {
      if (WS_ABL & 8) { if (slot == 0 && a == 0) xf[kc] = *(const bf16x8*)(At + kc * 16); continue; }
      xf[kc] = *(const bf16x8*)(At + (kc >> 1) * 8192 + ((((kc & 1) * 4 + fq) ^ (row & 7)) << 4));
    }
#endif
    __builtin_amdgcn_sched_barrier(0x000F);
  };
  load_xf(0, 0);
  if ((WS_VAR & 2) && HS) __builtin_amdgcn_s_setprio(1);      // the younger half loses every arbitration otherwise (MI355X guide, "Two waves per SIMD" item 4)
  if ((WS_VAR & 4) && !HS) __builtin_amdgcn_s_setprio(1);
  for (int it = 0; it < count; ++it, tile += tstep) {
    const int slot = it % WS_SLOTS;
    const int m0 = tile * WS_RT;
    const bool dma = it + 2 < count && !(WS_ABL & 1);
    WS_STAMP(it, 0);
    // the slot of tile it + 2 held tile it - 1: every wave passed the barrier behind that tile's last MFMAs
    if (dma) issue_tile(tile + 2 * tstep, (it + 2) % WS_SLOTS);
    WS_STAMP(it, 1);
    // the rows' LayerNorm statistics are read in front of the ring hand-over (the slot is rewritten behind it)
    float2 lcs[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      lcs[a] = *(const float2*)(smem + WS_STAT + slot * 4096 + a * 1024 + fr * 8);
      lcs[a].x = lnf ? lcs[a].x : 0.f; lcs[a].y = lnf ? lcs[a].y : 1.f;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int row = a * 16 + fr;
      // ring hand-over (see below), waves 4 .. 7: in front of the MFMAs of the tile's last step
      if constexpr (HS) if (a == 3 && it + 1 < count) {
        if (it > 0) {
          constexpr int NY = 7 * NST;
          if (dma) wait_vm<NY + ND>(); else wait_vm<NY>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(WS_ABL & 32)) __builtin_amdgcn_s_barrier();
      }
      const float2 lc = lcs[a];
      f32x4 acc[NTW];
#pragma unroll
      for (int jn = 0; jn < NTW; ++jn) acc[jn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < WS_KC; ++kc)
#pragma unroll
        for (int jn = 0; jn < NTW; ++jn) {
          if (WS_ABL & 4) { if (kc == 0) { acc[jn][0] = (float)xf[jn][0]; acc[jn][1] = (float)wreg[jn][a][0]; } continue; }
          acc[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[jn][kc], xf[kc], acc[jn], 0, 0, 0);
        }
      if (a < 3) {
        load_xf(slot, a + 1);
      } else if (it + 1 < count) {
        // ring hand-over, ONE barrier per tile: the requests of tile it + 1 (issued at the start of tile it - 1) have landed once at most
        // the operations issued behind them are pending -- the 4 steps of tile it - 1, this tile's own requests, 4 load groups and 3 steps
        // of stores -- and every fragment of tile it is in registers (they are read one step ahead).  Waves 0 .. 3 arrive behind the MFMAs
        // of the tile's last step, waves 4 .. 7 (the SIMD partners) in front of theirs: behind the barrier one group runs an epilogue
        // while the other runs MFMAs, and they stay one phase apart -- started together they would do both phases in lock step and
        // the matrix pipe would idle through every epilogue (tools/ws_trace.py).
        if constexpr (!HS) {
          if (it > 0) {
            constexpr int NY = 7 * NST;
            if (dma) wait_vm<NY + ND>(); else wait_vm<NY>();
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!(WS_ABL & 32)) __builtin_amdgcn_s_barrier();
        }
        load_xf((it + 1) % WS_SLOTS, 0);
      }
      WS_STAMP(it, 2 + 2 * a);
      // ---- epilogue of rows m0 + a * 16 + fr
      const int m = m0 + row;
      const float rs = lc.y * p.alpha, nm = -lc.y * lc.x;      // CF_LNFOLD: rstd and -rstd * mean of this lane's row (1, 0 otherwise)
      if constexpr (GEGLU) {
        // tile 0: hidden, tile 1: gate pre-activations of output columns fq * 4 .. + 3 of this wave's 16
        float h[4], g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[r] = __builtin_fmaf(rs, acc[0][r], __builtin_fmaf(nm, chv[r], bhv[r]));
          g[r] = __builtin_fmaf(rs, acc[1][r], __builtin_fmaf(nm, cgv[r], bgv[r]));
        }
        if constexpr (RAW) {
          bf16_t* rp = p.raw + (size_t)m * p.raw_ld + n0 + colw + fq * 4;
          *(uint2*)rp = make_uint2(pack2bf(h[0], h[1]), pack2bf(h[2], h[3]));
          *(uint2*)(rp + 16) = make_uint2(pack2bf(g[0], g[1]), pack2bf(g[2], g[3]));
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (WS_ABL & 2) ? h[e] + g[e] : h[e] * gelu_f(g[e]);
        *(uint2*)((bf16_t*)p.y + (size_t)m * p.y_ld + ((n0 + colw) >> 1) + fq * 4) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
      } else {
        bf16_t* yp = (bf16_t*)p.y + (size_t)m * p.y_ld + n0 + colw;
        float r1 = 0.f, r2 = 0.f;                            // CF_ROWSTATS
        auto four = [&](const f32x4& v, int col) {
          float4 b = *(const float4*)(bw + col);
          const float4 c = *(const float4*)(cw + col);       // zeros without CF_LNFOLD
          b.x = __builtin_fmaf(nm, c.x, b.x); b.y = __builtin_fmaf(nm, c.y, b.y); b.z = __builtin_fmaf(nm, c.z, b.z); b.w = __builtin_fmaf(nm, c.w, b.w);
          float v0 = __builtin_fmaf(v[0], rs, b.x), v1 = __builtin_fmaf(v[1], rs, b.y), v2 = __builtin_fmaf(v[2], rs, b.z), v3 = __builtin_fmaf(v[3], rs, b.w);
          v0 = fmaxf(v0, relu_floor); v1 = fmaxf(v1, relu_floor); v2 = fmaxf(v2, relu_floor); v3 = fmaxf(v3, relu_floor);
          if constexpr (RS) {
            r1 += (v0 + v1) + (v2 + v3);
            r2 = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, r2))));
          }
          return make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        };
#pragma unroll
        for (int t = 0; t < NP; ++t) {
          const int col = t * 32 + fq * 8;
          const uint2 lo = four(acc[2 * t], col);
          const uint2 hi = four(acc[2 * t + 1], col + 4);
          *(uint4*)(yp + col) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        if constexpr (ODD != 0) {
          const int col = NP * 32 + fq * 4;
          *(uint2*)(yp + col) = four(acc[NTW - 1], col);
        }
        if constexpr (RS) {
          // the row's NTW * 16 columns of this wave sit in lanes fr, fr + 16, fr + 32, fr + 48; every lane stores (same address per
          // row: the instruction count of the step stays fixed for the counted waits)
          r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
          r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
          if (fq == 0) *(float2*)(p.rowpart + ((size_t)m * p.rowpart_ld + span) * 2) = make_float2(r1, r2);
        }
      }
      WS_STAMP(it, 3 + 2 * a);
    }
  }
}

// TN = 5: 320-column blocks (spans of 80 columns as 48 + 32); TN = 4: 256-column blocks (GEGLU, spans of 64 packed columns as 32 + 32).
// CB column blocks; RL row lanes per XCD label: the CB workgroups of one row lane (same XCD: one L2) walk the same row tiles.
template <int TN, bool GEGLU, bool RS, bool RAW>
__global__ __launch_bounds__(512, 1) void gemm_ws_kernel(ConvGemmParams p, int CB, int RL) {
  constexpr int BN = 4 * TN * 16;
  constexpr int NTA = (TN + 1) / 2, NTB = TN / 2;      // tiles of the first (w < 4) and second (w >= 4) wave of a SIMD
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 3, second = wave >> 2;
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int cb = j % CB, rl = j / CB;
  if (rl >= RL) return;
  const int n0 = cb * BN;
  const int ntiles = p.M / WS_RT;
  const int tstep = RL * 8, tfirst = rl * 8 + xcd;     // row tiles (i * RL + rl) * 8 + xcd
  if (tfirst >= ntiles) return;
  const int count = (ntiles - tfirst + tstep - 1) / tstep;
  float* const bias_s = (float*)(smem + WS_AUX);
  if (tid < BN) {
    bias_s[tid] = (p.flags & CF_BIAS) ? p.bias[n0 + tid] : 0.f;
    bias_s[BN + tid] = (p.flags & CF_LNFOLD) ? p.ln_c1[n0 + tid] : 0.f;
  }
  const int span = (cb * 4 + wc) * 2 + second;          // CF_ROWSTATS: column span index of this wave (N / 40 spans per row)
  if (second) ws_wave<NTB, GEGLU, RS, RAW, true>(p, smem, BN, n0, wc * (TN * 16) + NTA * 16, span, wave, tfirst, tstep, count);
  else ws_wave<NTA, GEGLU, RS, RAW, false>(p, smem, BN, n0, wc * (TN * 16), span, wave, tfirst, tstep, count);
}

template <int TN, bool GEGLU, bool RS, bool RAW>
hipError_t run_ws(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int BN = 64 * TN;
  const int lds = WS_AUX + 2 * BN * 4;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)gemm_ws_kernel<TN, GEGLU, RS, RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  static const int cus = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 8 ? n & ~7 : 8; }();
  const int per = cus / 8;                               // workgroups per XCD label
  const int CB = p.N / BN, RL = per / CB;
  hipLaunchKernelGGL((gemm_ws_kernel<TN, GEGLU, RS, RAW>), dim3(cus), dim3(512), lds, stream, p, CB, RL);
  return hipGetLastError();
}

}  // namespace
// weight-stationary GEMM for K = 320 pointwise layers: 0 = not eligible, 5: 320-column blocks, 4: 256-column blocks (GEGLU)
int gemm_ws_config(const ConvGemmParams& p) {
  static const int on = getenv("DD_GEMM_WS") ? atoi(getenv("DD_GEMM_WS")) : 1;
  static const int mmin = getenv("DD_GEMM_WS_MMIN") ? atoi(getenv("DD_GEMM_WS_MMIN")) : 32768;
  static const int mask = getenv("DD_GEMM_WS_MASK") ? atoi(getenv("DD_GEMM_WS_MASK")) : 15;   // diagnostics: 1 GEGLU, 2 row statistics, 4 LayerNorm-folded, 8 the rest
  if (!on || p.force_small) return 0;
  if (!(mask & ((p.flags & CF_GEGLU) ? 1 : (p.flags & CF_ROWSTATS) ? 2 : (p.flags & CF_LNFOLD) ? 4 : 8))) return 0;
  if (p.ntaps != 1 || p.stride != 1 || p.shift || p.parity || p.H != p.Ho || p.W != p.Wo || p.cin != 320 || p.K != 320) return 0;
  if ((p.M & 63) || p.M < mmin || p.ksplit > 1 || p.bias_sel || (p.x_ld & 7) || (p.y_ld & 7)) return 0;
  if ((size_t)64 * p.x_ld * 2 >= 0xF0000000ull) return 0;
  static const int cus = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n > 8 ? n & ~7 : 8; }();
  const int per = cus / 8;
  if (p.flags & CF_GEGLU) {
    if (p.flags & ~(CF_BIAS | CF_GEGLU | CF_GEGLU_RAW | CF_LNFOLD)) return 0;
    if ((p.N & 255) || p.N / 256 > per || ((p.flags & CF_GEGLU_RAW) && (p.raw_ld & 3))) return 0;
    return 4;
  }
  if (p.flags & ~(CF_BIAS | CF_RELU | CF_ROWSTATS | CF_LNFOLD)) return 0;     // (CF_RES: see the header)
  if (p.N % 320 || p.N / 320 > per) return 0;
  return 5;
}
// CF_ROWSTATS spans of the weight-stationary form: 48- and 32-column spans, N / 40 per row
int gemm_ws_rowstat_spans(const ConvGemmParams& p) { return p.N / 40; }
#ifdef WS_TRACE
extern "C" int dd_debug_read_ws_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws_trace), sizeof(unsigned long long) * n);
}
#endif
hipError_t launch_gemm_ws(const ConvGemmParams& p, int tn, hipStream_t stream) {
  if (tn == 4) return (p.flags & CF_GEGLU_RAW) ? run_ws<4, true, false, true>(p, stream) : run_ws<4, true, false, false>(p, stream);
  return (p.flags & CF_ROWSTATS) ? run_ws<5, false, true, false>(p, stream) : run_ws<5, false, false, false>(p, stream);
}
