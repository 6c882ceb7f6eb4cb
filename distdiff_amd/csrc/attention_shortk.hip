// K5 (short key sequences) -- the cross-attention of the UNet's transformer blocks: 77 prompt tokens against 4096 / 1024 queries per image
// and head (diffusers Attention processor inside unet(), generate_data.py:112).  bf16 MFMA 16x16x32, gfx950.
//
// Why its own kernel: with <= 80 keys there is no key loop to pipeline.  The streaming forward (attention.hip, attn_fwd_dma_kernel) gives a
// workgroup 256 queries of one head; it stages the same 25 KB of K / V again for every such block (8192 workgroups per launch at 64x64:
// more L2 -> LDS bytes than the queries themselves), loads its Q fragments from global memory at the top of a dependent chain (load -> QK^T
// -> softmax -> P.V -> store, two barriers) and leaves: 145 us per launch against 61 us for reading Q and writing O once.  Here
//   * K and V of one (image, head) are staged ONCE per workgroup (LDS-DMA, zero rows behind the last key) and stay;
//   * every wave then walks its own 32-query tiles on its own: the next tile's Q rows are requested by LDS-DMA into the wave's private
//     two-slot ring while the current tile is computed, so no wave ever waits on a barrier or on another wave after the prologue;
//   * all scores of a query row (<= 80 keys = 5 MFMA tiles) are in registers at once: exact softmax in one pass -- row maximum, exp2, row
//     sum -- no online rescaling, no lazy reference;
//   * the LDS-DMA requests are inline asm with hand-counted waits (gemm_ws.hip explains why): vmcnt retires in order, the only younger
//     operations behind a tile's Q request are the previous tile's stores and the next request, both of fixed count (Nq % 32 == 0).
// Same contract as launch_attention_fwd: O bf16 [B * Nq, ldo] (head h at columns h * D), LSE fp32 [B][H][Nq] in the natural-log domain.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 sk_row_frag(const unsigned char* base, int row, int S, int slot) {
  return *(const bf16x8*)(base + row * S + slot * 16);
}
// 8 k-values (the permuted order of attention.hip's header) of column col16 * 16 + (lane & 15): rows r0 + 4 * (lane >> 4) + {0 .. 3} and + 16
__device__ __forceinline__ bf16x8 sk_col_frag(const unsigned char* base, int r0, int S, int col16, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const unsigned char* a = base + (r0 + 4 * g + (i >> 2)) * S + (col16 * 16 + 4 * (i & 3)) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 16 * S));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float sk_max2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ bf16x8 sk_pack(const f32x4& a, const f32x4& b) {
  uint4 u;
  u.x = pack2bf(a[0], a[1]); u.y = pack2bf(a[2], a[3]); u.z = pack2bf(b[0], b[1]); u.w = pack2bf(b[2], b[3]);
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ i32x4 sk_rsrc(const void* base) {
  const unsigned long long a = (unsigned long long)base;
  return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)), (int)0xffffff00u, 0x00020000};
}
__device__ __forceinline__ void sk_dma16(const i32x4 rsrc, unsigned lds_addr, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(0u) : "memory", "m0");
}
__device__ __forceinline__ unsigned sk_lds_off(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
template <int N>
__device__ __forceinline__ void sk_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D>
struct SkGeo {
  static constexpr int DPK = (D + 31) / 32 * 32;
  static constexpr int RB = D <= 40 ? 96 : D <= 80 ? 160 : DPK * 2 + 32;   // LDS row bytes: conflict-free for row and column fragments (attention.hip)
  static constexpr int RG = RB / 16, DG = D / 8;
  static constexpr int KS = DPK / 32, DVT = (D + 15) / 16;
};
constexpr float SK_LOG2E = 1.4426950408889634f;
constexpr int SK_NKT = 5;                   // 16-key score tiles: Nk <= 80
constexpr int SK_KROWS = 96;                // staged key rows (three 32-key chunks of P.V; rows behind the last key are zeros)
constexpr unsigned SK_OOR = 0xffffff00u;

// One 32-query tile of one head: Q fragments from the wave's LDS slot Qt, scores against the resident K block, exact softmax, P.V against
// the resident V block, O and LSE stores (2 DVT (+ 2) store instructions, always issued: the callers' counted waits rely on that).
template <int D>
__device__ __forceinline__ void sk_tile_frags(const AttnParams& p, const unsigned char* Ks, const unsigned char* Vs, bf16x8 (&qf)[2][SkGeo<D>::KS], float c,
                                              int kfull, int b, int h, int t, int lane) {
  using G = SkGeo<D>;
  constexpr int S = G::RB, KS = G::KS, DVT = G::DVT, QROWS = 32;
  constexpr bool ONES = (D % 16) == 8;
  const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // columns behind D: a fragment slot past the row's data granules holds the padding granule (zeros) or the next row
      if (ks * 32 + g * 8 >= D) qf[qt][ks] = __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
    }
  // ---- S^T = K Q^T: lane (i16, g) holds, for query i16 of tile qt, keys 16 kt + 4 g + {0 .. 3}
  f32x4 st[2][SK_NKT];
#pragma unroll
  for (int kt = 0; kt < SK_NKT; ++kt) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) st[qt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 kf = sk_row_frag(Ks, kt * 16 + i16, S, g + 4 * ks);
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) st[qt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ks], st[qt][kt], 0, 0, 0);
    }
  }
  f32x4 o[2][DVT];
  float lse_v[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    // keys behind Nk (zero rows of K: score 0) leave the softmax; whole tiles in front of Nk carry no mask code (wave-uniform test)
#pragma unroll
    for (int kt = 0; kt < SK_NKT; ++kt)
      if (kt >= kfull) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kt * 16 + 4 * g + r >= p.Nk) st[qt][kt][r] = -INFINITY;
      }
    float mx = st[qt][0][0];
#pragma unroll
    for (int kt = 0; kt < SK_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = sk_max2(mx, st[qt][kt][r]);
    mx = sk_max2(mx, __shfl_xor(mx, 16, 64));
    mx = sk_max2(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < SK_NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qt][kt][r], c, -mc));
        st[qt][kt][r] = e;
        if (!ONES) ps += e;
      }
    // ---- O^T = V^T P^T over three 32-key chunks (the sixth score tile is all padding: zeros)
    const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 pf[3] = {sk_pack(st[qt][0], st[qt][1]), sk_pack(st[qt][2], st[qt][3]), sk_pack(st[qt][4], zero)};
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      o[qt][dt] = zero;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc)
        o[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sk_col_frag(Vs, 32 * cc, S, dt, lane), pf[cc], o[qt][dt], 0, 0, 0);
    }
    // row sum: d % 16 == 8 (d = 40) -- column D of the staged V block is 1.0, so P.V delivered the sum of the bf16-ROUNDED probabilities in
    // O column D (lane group (D % 16) / 4, register 0): O is then an exact convex combination of the value rows, as in the streaming
    // kernel; else the fp32 sum of the unrounded probabilities
    if (ONES) ps = __shfl(o[qt][DVT - 1][(D % 16) % 4], ((D % 16) / 4) * 16 + i16, 64);
    else { ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64); }
    const float inv = 1.f / ps;
    lse_v[qt] = (mc + log2f(ps)) * 0.6931471805599453f;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[qt][dt][r] *= inv;
  }
  // every fragment of this Q slot is in registers: the request of the tile after next may overwrite it (issued at the top of the next
  // iteration, behind these ds_reads in program order and behind the lgkmcnt wait the compiler puts in front of their first use)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qrow = t * QROWS + qt * 16 + i16;
    bf16_t* op = p.o + ((size_t)b * p.Nq + qrow) * p.ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < DVT; ++dt) {
      const int dv = dt * 16 + 4 * g;
      // (D % 16 != 0: the lanes behind D store nothing; the instruction is issued all the same -- the counted waits rely on that)
      if (dv < D) *(uint2*)(op + dv) = make_uint2(pack2bf(o[qt][dt][0], o[qt][dt][1]), pack2bf(o[qt][dt][2], o[qt][dt][3]));
    }
    if (p.lse && g == 0) p.lse[((size_t)b * p.H + h) * p.Nq + qrow] = lse_v[qt];
  }
}

template <int D>
__device__ __forceinline__ void sk_tile(const AttnParams& p, const unsigned char* Ks, const unsigned char* Vs, const unsigned char* Qt, float c,
                                        int kfull, int b, int h, int t, int lane) {
  using G = SkGeo<D>;
  bf16x8 qf[2][G::KS];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) qf[qt][ks] = sk_row_frag(Qt, qt * 16 + (lane & 15), G::RB, (lane >> 4) + 4 * ks);
  sk_tile_frags<D>(p, Ks, Vs, qf, c, kfull, b, h, t, lane);
}

// 1-D grid of (query range, head) workgroups, XCD-aware: the H heads of one (image, query range) read and write 2 D-byte slices of the
// SAME rows, so they are given consecutive slots of ONE XCD (block n: XCD n % 8; slot n / 8 = (range group, head)) -- they run at the same
// time on CUs that share an L2 and every 128-byte line of Q / O moves between HBM and that L2 once.  (As a (ranges, H, B) grid with few
// ranges per head the heads of a row landed on four different XCDs: 169 us for the 64x64 level, no better than the streaming kernel.)
// NW waves; wave w of range x (of `wgs` per image and head) walks the 32-query tiles (x + j wgs) * NW + w.
template <int D, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_fwd_shortk_kernel(AttnParams p, int ntiles, int wgs) {
  using G = SkGeo<D>;
  constexpr int S = G::RB, RG = G::RG, DG = G::DG, KS = G::KS, DVT = G::DVT;
  constexpr int QROWS = 32;                                   // queries per wave and tile (two 16-query MFMA tiles)
  constexpr int QSLOT = ((QROWS * S + 1023) / 1024) * 1024;   // bytes of one Q slot (whole 1 KB DMA pieces)
  constexpr int NPQ = QSLOT / 1024;                           // LDS-DMA pieces per Q tile
  constexpr int KVB = SK_KROWS * S;                           // bytes of the K (or V) block
  constexpr int NPK = (KVB + 1023) / 1024;                    // pieces of K (or V)
  constexpr int NST = 2 * DVT;                                // O stores per tile (+ 2 LSE stores when asked for)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Ks = smem;
  unsigned char* const Vs = smem + NPK * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const Qs = smem + 2 * NPK * 1024 + wave * 2 * QSLOT;
  const int i16 = lane & 15, g = lane >> 4;
  const int slot_ = blockIdx.x >> 3, h = slot_ % p.H, grp = (slot_ / p.H) * 8 + (blockIdx.x & 7);      // range group = (image, range)
  if (grp >= p.B * wgs) return;
  const int b = grp / wgs, bx = grp - b * wgs;

  // ---- zero the whole block once: padding granules, rows behind the last key, the tails the 64-wide fragments read past a row's end
  for (int v = tid; v < (2 * NPK * 1024 + NW * 2 * QSLOT) / 16; v += NW * 64) *(uint4*)(smem + v * 16) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if constexpr ((D % 16) == 8) {                               // column D of every staged V row = 1.0: P.V also delivers the row sums
    for (int r = tid; r < SK_KROWS; r += NW * 64) *(unsigned*)(Vs + r * S + DG * 16) = 0x3f80u;
    __syncthreads();
  }

  // ---- K and V of this (image, head): piece pc of the workgroup's 2 NPK, lane -> (row, granule); padding granules stay off
  {
    const bf16_t* kg = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const bf16_t* vg = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    const i32x4 rk = sk_rsrc(kg), rv = sk_rsrc(vg);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(sk_lds_off(smem));
    for (int pc = wave; pc < 2 * NPK; pc += NW) {
      const bool isv = pc >= NPK;
      const int gi = (isv ? pc - NPK : pc) * 64 + lane, row = gi / RG, v = gi - row * RG;
      const unsigned voff = (v < DG && row < p.Nk) ? ((unsigned)row * (unsigned)(isv ? p.ldv : p.ldk) + (unsigned)v * 8u) * 2u : SK_OOR;
      if (v < DG) sk_dma16(isv ? rv : rk, lds0 + pc * 1024, voff);
    }
  }

  // ---- this wave's query tiles
  const int wt0 = bx * NW + wave, wstep = wgs * NW;                   // tile index = 32-query block of this (image, head)
  const bf16_t* qg = p.q + (size_t)b * p.Nq * p.ldq + h * D;
  const unsigned qlds = __builtin_amdgcn_readfirstlane(sk_lds_off(Qs));
  unsigned qvoff[NPQ];
  bool qon[NPQ];
#pragma unroll
  for (int i = 0; i < NPQ; ++i) {
    const int gi = i * 64 + lane, row = gi / RG, v = gi - row * RG;
    qon[i] = v < DG && row < QROWS;
    qvoff[i] = ((unsigned)row * (unsigned)p.ldq + (unsigned)v * 8u) * 2u;
  }
  auto issue_q = [&](int t, int slot) {
    const i32x4 rq = sk_rsrc(qg + (size_t)t * QROWS * p.ldq);
#pragma unroll
    for (int i = 0; i < NPQ; ++i)
      if (qon[i]) sk_dma16(rq, qlds + slot * QSLOT + i * 1024, qvoff[i]);
  };
  // (an inactive lane of an LDS-DMA instruction leaves its 16-byte slot alone, tools/micro/buflds.hip: the zeroed padding survives)
  int t = wt0;
  if (t < ntiles) issue_q(t, 0);
  sk_wait_vm<0>();
  __syncthreads();                                             // K / V complete for everybody (and this wave's first Q tile)

  const float c = p.q_prescaled ? 1.f : p.scale * SK_LOG2E;    // scores -> log2 domain
  const int kfull = p.Nk >> 4;                                 // score tiles that are whole
  for (int it = 0; t < ntiles; ++it, t += wstep) {
    const int slot = it & 1;
    const bool more = t + wstep < ntiles;
    if (more) issue_q(t + wstep, slot ^ 1);
    // this tile's rows have landed once only the operations issued behind their request are pending: the previous tile's stores and the
    // request just issued (the LSE stores are not counted: with them the wait is a little early, never late)
    if (it > 0) { if (more) sk_wait_vm<NST + NPQ>(); else sk_wait_vm<NST>(); }
    else if (more) sk_wait_vm<NPQ>();
    sk_tile<D>(p, Ks, Vs, Qs + slot * QSLOT, c, kfull, b, h, t, lane);
  }
}

template <int D, int NW>
hipError_t run_shortk(const AttnParams& p, hipStream_t s) {
  using G = SkGeo<D>;
  constexpr int QSLOT = ((32 * G::RB + 1023) / 1024) * 1024;
  constexpr int NPK = (SK_KROWS * G::RB + 1023) / 1024;
  const size_t lds = 2 * NPK * 1024 + (size_t)NW * 2 * QSLOT;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_shortk_kernel<D, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  const int ntiles = p.Nq / 32;
  // tiles per wave: enough workgroups to fill the chip twice over, at most 8 tiles per wave (the K / V staging is amortised over them)
  int wgs = (ntiles + NW - 1) / NW;                            // one tile per wave
  const int heads = p.H * p.B;
  int per = 8;
  while (per > 1 && (long long)((wgs + per - 1) / per) * heads < 1024) per >>= 1;
  wgs = (wgs + per - 1) / per;
  const int groups = (p.B * wgs + 7) & ~7;
  hipLaunchKernelGGL((attn_fwd_shortk_kernel<D, NW>), dim3(groups * p.H), dim3(NW * 64), lds, s, p, ntiles, wgs);
  return hipGetLastError();
}

}  // namespace

// Eligible: non-causal, <= 80 keys, whole 32-query tiles, d in {40, 64, 80}, rows within the 32-bit byte offsets of one image
bool attention_shortk_supported(const AttnParams& p) {
  static const int on = getenv("DD_ATTN_SHORTK") ? atoi(getenv("DD_ATTN_SHORTK")) : 1;
  if (!on || p.no_shortk || p.causal || p.pv_fp8 || p.Nk < 1 || p.Nk > 80 || (p.Nq & 31) || p.Nq < 32) return false;
  if (p.D != 40 && p.D != 64 && p.D != 80) return false;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 3)) return false;
  if ((size_t)p.Nq * (size_t)p.ldq * 2 >= 0xF0000000ull || (size_t)p.Nk * (size_t)(p.ldk > p.ldv ? p.ldk : p.ldv) * 2 >= 0xF0000000ull) return false;
  return true;
}

hipError_t launch_attention_fwd_shortk(const AttnParams& p, hipStream_t s) {
  if (!attention_shortk_supported(p)) return hipErrorInvalidValue;
  switch (p.D) {
    case 40: return run_shortk<40, 8>(p, s);
    case 64: return run_shortk<64, 4>(p, s);
    case 80: return run_shortk<80, 4>(p, s);
    default: return hipErrorInvalidValue;
  }
}
