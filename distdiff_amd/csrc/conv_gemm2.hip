// K1/K2 (large shapes) — persistent implicit-GEMM convolution / linear on bf16 MFMA, gfx950.
//
// Same math, operand layout, LDS swizzle and epilogue semantics as conv_gemm.hip, restructured for the shapes that dominate the
// UNet / VAE time.  DESIGN.md section 3 describes the design and section 6 what was measured.
//   * Two forms of one kernel template:
//       8 waves, tiles 128x256 / 256x160 / 256x128, three LDS stages, one workgroup per CU  (items deeper than ~100 K-steps, big VAE shapes);
//       4 waves, tiles 128x160 / 128x128, two LDS stages, TWO workgroups per CU            (shallow items: one group's epilogue and
//       pipeline refill overlap the other's MFMAs).  BN = 160 removes the N-padding waste of the SD-1.x channel counts.
//   * Staging is buffer_load_dwordx4 ... lds (LDS-DMA): per-lane byte offsets fixed per work item, the K-step moves through the scalar
//     offset, out-of-range lanes are zero-filled by the hardware (padding taps, ragged rows).  The swizzle is applied on the source side.
//   * The K loop is software-pipelined through registers (fragment sets F0/F1), one barrier per 64-wide K-step, every LDS read
//     unconditional and every counter scalar (see the notes at the loop: both matter to what the compiler emits).
//   * Persistent: each workgroup walks a list of (tile, K-split) work items with the loader running ahead into the next item; items are
//     dealt to XCDs in contiguous chunks (blocks b, b+8, ... share an L2) with n-tiles fastest.
//   * Epilogue: paired output columns -> 16-byte residual loads / stores; a batched form (FE) for short items, the generic
//     Epi::apply form for deep / split-K items (its instantiation has the faster K loop).
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "conv_epilogue.h"

namespace {

__device__ uint4 g_zero16[4];
#ifdef DD_TRACE
// debug build only (tools/conv_trace.py): s_memrealtime stamps (10 ns ticks) of the first and the last workgroup, thread 0
__device__ unsigned long long g_trace[512];
__device__ __forceinline__ void trace_stamp(int& n, int tag) {
  const bool lastb = blockIdx.x == gridDim.x - 1;
  if ((blockIdx.x == 0 || lastb) && threadIdx.x == 0 && n < 127) {
    const int o = lastb ? 256 : 0;
    g_trace[o + 2 * n] = __builtin_amdgcn_s_memrealtime(); g_trace[o + 2 * n + 1] = tag; ++n;
  }
}
#endif   // zero page for padded taps / ragged rows (device globals are zero-initialised)

// One 1 KB LDS-DMA piece: buffer_load_dwordx4 ... lds from base + voff (per lane) + soff (scalar); a lane whose voff is beyond the
// 4 GB - 256 B range gets zeros written to its LDS slot (hardware range check) -- that is how padding taps and ragged rows are
// staged.  The resource builtins only exist in the device pass (a kernel template that names them loses its host stub otherwise).
__device__ __forceinline__ void dma16(const void* base, void* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffff00u, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
#endif
}

// sum over the 16 lanes of a DPP row (the 16 pixel rows of an MFMA tile); every lane of the row receives the total
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return v;
}

// FM: 1 = fast staging path only (Cin % 64 == 0, no fused upsample / dilation, <= 32 taps: buffer loads with scalar tap offsets),
//     0 = general path only (per-lane address arithmetic, global_load_lds).  FE: batched epilogue compiled in.
// ST: epilogue extras, each its own instantiation (the code of the plain kernels -- above all the register allocation of their K
// loop -- stays exactly what it was without them): 1 = CF_STATS (GroupNorm partials), 2 = CF_ROWSTATS (LayerNorm row partials),
// 3 = CF_LNFOLD (this GEMM consumes the raw input of a LayerNorm folded into its weights; c1 arrives through a second LDS ring).
template <int WM, int WN, int TM, int TN, int NS, bool FE, int FM, int ST>
__global__ __launch_bounds__(WM* WN * 64, (WM * WN == 4) ? 2 : 1) void conv_gemm_big_kernel(ConvGemmParams p) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int NW = WM * WN;   // 8 waves (one workgroup per CU) or 4 waves (two workgroups per CU, <= 256 VGPRs each)
  constexpr int NT = NW * 64;
  constexpr int RPT = NW * 8;   // tile rows per full staging pass: NT threads x 16 B = RPT rows of 128 B
  constexpr int AV = BM / RPT;  // A passes (all full)
  constexpr int BV = (BN + RPT - 1) / RPT;          // W passes; the last one may be a half pass (32 rows, lanes 0-31 of every wave)
  constexpr bool B_HALF = (BN % RPT) == RPT / 2;
  static_assert((NW == 8 || NW == 4) && BM % RPT == 0 && (BN % RPT == 0 || B_HALF) && (NS == 2 || NS == 3), "bad tile");
  constexpr int BUF_BYTES = (BM + BN) * 128;
  constexpr int DMA_PER_STAGE = AV + BV;
  // bias of the n-tile of an item: one 1 KB LDS-DMA piece issued when the loader reaches the item (a ring of three: the loader is
  // at most two items ahead of the epilogue), read from LDS by the batched epilogue -- a global load there costs an exposed L2 / HBM
  // latency per item (tools/conv_trace.py: 3 us of epilogue on a 5-step item, half of it the bias wait)
  constexpr int BIAS_OFF = NS * BUF_BYTES + 256;
  // ST == 3: column sums of the LayerNorm-folded weights and the (mean, rstd) of the item's BM = 128 rows (1 KB each) arrive the same
  // way; two slots each: with K >= 192 (three K-steps per item, checked by the host) the two-stage loader reaches item w + 2 only
  // after the epilogue of item w has run
  constexpr int C1_OFF = BIAS_OFF + 3 * 1024;
  constexpr int LNS_OFF = C1_OFF + 2 * 1024;
  static_assert(ST != 3 || (NS == 2 && BM == 128), "LayerNorm-fold form: two-stage 128-row tiles only");
  int l_slot2 = 0, c_slot2 = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
#ifdef DD_TRACE
  int tn = 0;
  trace_stamp(tn, 3);
#endif

  // ---- work list of this workgroup
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int ksteps = p.K >> 6;
  const int per = (ksteps + p.ksplit - 1) / p.ksplit;      // host guarantees every split is non-empty
  const int W = ntm * ntn * p.ksplit;
  const int Gx = gridDim.x >> 3;
  int w_first, w_end;
  {
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int q = W >> 3, r = W & 7;
    const int start = xcd * q + min(xcd, r);
    w_first = start + jb;
    w_end = start + q + (xcd < r ? 1 : 0);
  }
  if (w_first >= w_end) return;

  const int cin = p.cin, cin8 = p.cin >> 3;
  const bool uniform_tap = (cin & 63) == 0;
  const int shift = p.shift, parity = p.parity;
  const int HoWo = p.Ho * p.Wo;

  // ---- loader iterator (runs one K-step ahead of the compute iterator). Staging is LDS-DMA
  // (global_load_lds_dwordx4: no staging VGPRs, no ds_write, no zero-select): each wave-instruction fills 8 rows x 128 B
  // linearly, so the XOR swizzle is applied on the SOURCE side: lane (row = lane>>3, physical slot = lane&7) fetches
  // logical K-slot (lane&7) ^ (row&7). Out-of-image taps / ragged rows read a 16-byte zero page instead.
  const int ps = lane & 7;
  const int j = ps ^ ((lane >> 3) & 7);       // logical 16-byte K-slot this lane fetches (full passes: row & 7 == (lane >> 3) & 7)
  const int r0 = tid >> 3;                    // tile row of staging pass 0
  const int rh = wave * 4 + (lane >> 3);      // row inside a half pass (lanes 0-31 only)
  const int jh = ps ^ (rh & 7);
  constexpr bool fast = FM == 1;   // the host picks FM == 1 exactly when Cin % 64 == 0, shift == 0 and ntaps <= 32 (run_big_fe)
  // the per-tap offset table lives in LDS: a global load inside the pipeline would force s_waitcnt vmcnt(0)
  // (vmcnt retires in order) and drain the in-flight DMA stages
  int* taps = (int*)(smem + NS * BUF_BYTES);
  // 1x1 / linear layers (one tap, same-size output: the packers give them the centre tap): no tap table at all -- its global load
  // is an HBM-latency wait plus three barriers in front of the first DMA of every workgroup, and these are the short kernels
  const bool pointwise = fast && p.ntaps == 1 && p.stride == 1 && p.H == p.Ho && p.W == p.Wo;
  if (!pointwise) {
    for (int t = tid; t < p.ntaps; t += NT) taps[t] = p.taptab[t];
    __syncthreads();
  }
  // Fast path staging uses buffer_load ... lds: the per-lane byte offset of a row is fixed for a whole work item, the K-step
  // (tap, 64-channel chunk) moves through the scalar offset, and an out-of-range offset makes the hardware write zeros
  // (padding taps, ragged rows) -- 3 VALU per 1 KB piece instead of a 64-bit address select.  The base is moved back by the most
  // negative tap offset so that the scalar offset stays unsigned.
  constexpr unsigned OOB = 0xfffffff0u;
  int tap_bias = 0;
  if (fast && !pointwise) {
    for (int t = 0; t < p.ntaps; ++t) {
      const int e = taps[t];
      tap_bias = max(tap_bias, -((((e >> 6) & 63) - 32) * p.W + ((e & 63) - 32)) * p.x_ld * 2);
    }
    tap_bias = __builtin_amdgcn_readfirstlane(tap_bias);
    __syncthreads();
    for (int t = tid; t < p.ntaps; t += NT) {
      const int e = taps[t];
      taps[32 + t] = tap_bias + ((((e >> 6) & 63) - 32) * p.W + ((e & 63) - 32)) * p.x_ld * 2;
    }
    __syncthreads();
  }
  const char* xbase = (const char*)p.x - tap_bias;
  const float* bias = p.bias;
  if ((p.flags & CF_BIAS) && p.bias_sel) bias += (size_t)(*p.bias_sel) * p.bias_stride;
  int l_slot = 0, c_slot = 0;                 // bias ring slots of the loader's / the epilogue's item
  int lw = w_first, l_kt = 0, l_kend = 0;
  const bf16_t* l_w = p.w;                               // weight matrix of the loader's item (grouped weights: one per row group)
  int l_tap = 0, l_chunk = 0, l_tb = 0;       // fast path: tap / chunk of K-step l_kt and its (prefetched) byte offset
  int pixb[AV], iy0[AV], ix0[AV];             // general path
  unsigned tapmask[AV];                       // fast path: bit t = tap t is inside the image for this row
  unsigned wrow[BV];                          // element offset of this lane's weight row (+ K-slot), or ~0u
  // Output-column layout of a wave's TN 16-wide MFMA tiles.  Tiles are paired (2t, 2t+1): LDS weight row (jn, f) of a pair holds
  // channel t*32 + (f>>2)*8 + (jn&1)*4 + (f&3), so a lane's 4+4 accumulator rows of the pair are 8 CONSECUTIVE output channels and
  // the epilogue moves 16 bytes per lane (4 lanes = 64 contiguous bytes per pixel) instead of 8.  An odd last tile keeps the plain
  // layout; GEGLU has its own (hidden | gate) pairing.  The permutation costs nothing: it only changes which weight row a DMA
  // lane fetches.
  const bool pair_cols = !(p.flags & CF_GEGLU);
  constexpr int TNP = TN & ~1;
  auto chan_of_row = [&](int R) {
    const int wv = R / (TN * 16), q = R - wv * (TN * 16);
    const int jn = q >> 4, f = q & 15;
    return (pair_cols && jn < TNP) ? wv * (TN * 16) + (jn >> 1) * 32 + (f >> 2) * 8 + (jn & 1) * 4 + (f & 3) : R;
  };
  auto col_of = [&](int jn) {   // first of the 4 consecutive columns this lane holds of tile jn, relative to the wave's span
    return (pair_cols && jn < TNP) ? (jn >> 1) * 32 + fq * 8 + (jn & 1) * 4 : jn * 16 + fq * 4;
  };
  auto setup_loader = [&](int w) {
    const int kz = w % p.ksplit, tile = w / p.ksplit;
    const int m0 = (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;
    l_kt = kz * per;
    l_kend = min(ksteps, l_kt + per);
    if (p.wgroup_rows > 0) l_w = p.w + (size_t)(m0 / p.wgroup_rows) * (size_t)p.wgroup_elems;
    {
      const int nb = n0 + lane * 4;
      const unsigned voff = ((p.flags & CF_BIAS) && lane * 4 < BN && nb + 4 <= p.N) ? (unsigned)nb * 4u : OOB;
      dma16(bias, smem + BIAS_OFF + l_slot * 1024, voff, 0);
      if constexpr (ST == 3) {
        const unsigned v1 = (lane * 4 < BN && nb + 4 <= p.N) ? (unsigned)nb * 4u : OOB;
        dma16(p.ln_c1, smem + C1_OFF + l_slot2 * 1024, v1, 0);
        const int mr = m0 + lane * 2;                                   // 16 bytes = (mean, rstd) of two rows
        dma16(p.ln_stats, smem + LNS_OFF + l_slot2 * 1024, mr + 2 <= p.M ? (unsigned)mr * 8u : OOB, 0);
        l_slot2 ^= 1;
      }
      l_slot = l_slot == 2 ? 0 : l_slot + 1;
    }
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int m = m0 + r0 + RPT * i;
      unsigned tm = 0;
      int pb = 0, y0 = -1000000, x0 = 0;
      if (pointwise) {
        // 1x1 / linear layers (the shallow-K items, where per-item setup is exposed): output row == input row, one tap, no padding
        if (m < p.M) { pb = m * p.x_ld * 2 + j * 16; tm = 1u; }
      } else if (m < p.M) {
        const int b = m / HoWo;
        const int rem = m - b * HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        y0 = oy * p.stride;
        x0 = ox * p.stride;
        if (fast) {
          pb = ((b * p.H + y0) * p.W + x0) * p.x_ld * 2 + j * 16;     // byte offset of the centre pixel + this lane's K-slot
          for (int t = 0; t < p.ntaps; ++t) {
            const int e = taps[t];
            const int yy = y0 + ((e >> 6) & 63) - 32, xx = x0 + (e & 63) - 32;
            tm |= (yy >= 0 && xx >= 0 && yy < p.H && xx < p.W) ? (1u << t) : 0u;
          }
        } else {
          pb = b * p.H * p.W;
        }
      }
      pixb[i] = pb; iy0[i] = y0; ix0[i] = x0; tapmask[i] = tm;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const bool half = B_HALF && i == BV - 1;
      const int n = n0 + chan_of_row(RPT * i + (half ? rh : r0));
      wrow[i] = n < p.N ? ((unsigned)n * (unsigned)p.K + (unsigned)((half ? jh : j) * 8)) * 2u : OOB;   // bytes
    }
    if (fast) {
      // (explicitly scalar: the division runs on the VALU, and a VGPR-resident counter turns every buffer_load's scalar offset
      //  into a readfirstlane waterfall loop)
      l_chunk = __builtin_amdgcn_readfirstlane(pointwise ? l_kt : l_kt / p.ntaps);
      l_tap = l_kt - l_chunk * p.ntaps;
    }
  };
  // byte offset of the loader's tap, read one K-step before issue_step consumes it (unconditional: see the note on the peeled step)
  auto prefetch_tap = [&]() { if (fast && !pointwise) l_tb = taps[32 + l_tap]; };
  auto issue_step = [&](int buf) {   // enqueue the LDS-DMA of K-step l_kt of the loader's item into buffer `buf`
    const int kt = l_kt;

    unsigned char* Abase = smem + buf * BUF_BYTES + wave * 1024;
    unsigned char* Bbase = smem + buf * BUF_BYTES + BM * 128 + wave * 1024;
    if (fast) {
      // K order = (64-channel chunk, tap): wave-uniform scalar offset, per-lane offsets fixed since setup_loader
      const unsigned soff = (unsigned)(__builtin_amdgcn_readfirstlane(l_tb) + l_chunk * 128);
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const unsigned voff = ((tapmask[i] >> l_tap) & 1u) ? (unsigned)pixb[i] : OOB;
        dma16(xbase, Abase + i * (RPT * 128), voff, soff);
      }
    } else {
      int e, coff;
      bool ev;
      if (uniform_tap) {
        const int chunk = kt / p.ntaps;
        const int tap = kt - chunk * p.ntaps;
        coff = chunk * 64 + j * 8;
        e = taps[tap];
        ev = true;
      } else {
        const int k8 = kt * 8 + j;
        const int tap = k8 / cin8;
        ev = tap < p.ntaps;
        coff = (k8 - tap * cin8) * 8;
        e = taps[ev ? tap : 0];
      }
      const int dx = (e & 63) - 32, dy = ((e >> 6) & 63) - 32;
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const int ly = iy0[i] + dy, lx = ix0[i] + dx;
        const int sy = ly >> shift, sx = lx >> shift;
        bool ok = ev && ly >= 0 && lx >= 0 && sy < p.H && sx < p.W;
        if (parity) ok = ok && (((ly | lx) & 1) == 0);
        const bf16_t* src = ok ? p.x + ((unsigned)(pixb[i] + sy * p.W + sx) * (unsigned)p.x_ld + (unsigned)coff) : (const bf16_t*)g_zero16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Abase + i * (RPT * 128)), 16, 0, 0);
      }
    }
    const unsigned wsoff = (unsigned)kt * 128u;
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      if (B_HALF && i == BV - 1) {
        // half pass: every wave moves 4 rows with its lanes 0-31 (same DMA count in every wave, so counted vmcnt waits are uniform)
        if (lane < 32)
          dma16(l_w, smem + buf * BUF_BYTES + BM * 128 + i * (RPT * 128) + wave * 512, wrow[i], wsoff);
      } else {
        dma16(l_w, Bbase + i * (RPT * 128), wrow[i], wsoff);
      }
    }
  };
  // advance the loader to the next K-step; returns false when the work list is exhausted
  auto advance_loader = [&]() -> bool {
    if (++l_kt < l_kend) {
      if (fast && ++l_tap == p.ntaps) { l_tap = 0; ++l_chunk; }
      return true;
    }
    lw += Gx;
    if (lw >= w_end) return false;
    setup_loader(lw);
    return true;
  };

  f32x4 acc[TN][TM];
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // fragments of one 32-wide K half of a stage: TN weight rows + TM pixel rows per lane (ds_read_b128 each)
  struct Frags { bf16x8 wf[TN]; bf16x8 xf[TM]; };
  auto load_frags = [&](Frags& F, int buf, int ks) {
    const unsigned char* A = smem + buf * BUF_BYTES;
    const unsigned char* Bt = A + BM * 128;
    const int slot = fq + 4 * ks;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int row = wn * (TN * 16) + jn * 16 + fr;
      F.wf[jn] = *(const bf16x8*)(Bt + row * 128 + ((slot ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (TM * 16) + i * 16 + fr;
      F.xf[i] = *(const bf16x8*)(A + row * 128 + ((slot ^ (row & 7)) << 4));
    }
  };
  auto mma = [&](const Frags& F) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.wf[jn], F.xf[i], acc[jn][i], 0, 0, 0);
  };
  // CF_STATS: per-(64-row block, channel) partial statistics of the values this wave stores, for the GroupNorm that consumes the
  // tensor: s1 / s2 hold this lane's sums over its 4 pixel rows, the 16 lanes of an MFMA row are summed with 4 DPP adds per value,
  // lane fr == 0 stores (mean, M2) of its 4 channels.  ~8 * TN * 4 VALU per wave and work item.
  auto emit_stats = [&](float (&s1)[TN][4], float (&s2)[TN][4], int m0w, int nW) {
    if (m0w >= p.M) return;
    float* dst0 = p.stats + ((size_t)(m0w >> 6) * p.stats_ld) * 2;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[jn][r]), q = row16_sum(s2[jn][r]);
        const float mean = a * (1.f / 64.f);
        o[2 * r] = mean; o[2 * r + 1] = fmaxf(q - a * mean, 0.f);
      }
      const int nb = nW + col_of(jn);
      if (fr == 0 && nb + 4 <= p.N) {
        float* dst = dst0 + (size_t)nb * 2;
        *(float4*)dst = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  };
  auto epilogue = [&](int w) {
    const int kz = w % p.ksplit, tile = w / p.ksplit;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    if (p.ksplit > 1) {
      float* part = p.partial + (size_t)kz * p.M * p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (TM * 16) + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int nb = n0 + wn * (TN * 16) + col_of(jn);
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
    // fast paths: every dependent global load (bias, residual) of the tile row is issued before its first use, so the
    // epilogue pays one memory latency per row instead of one per 4-column group
    if constexpr (FE) {
    const int ncol0 = n0 + wn * (TN * 16) + fq * 4;
    const bool tile_full = (n0 + wn * (TN * 16) + TN * 16 <= p.N) && !(p.y_ld & 7) && !(p.N & 7);
    const int fl = p.flags;
    // (a CF_LNFOLD launch that did not get its ST == 3 instantiation -- an eight-wave or general-staging form -- takes the generic
    //  epilogue below, whose Epi::apply folds from global memory)
    const bool fold_ok = ST == 3 || !(fl & CF_LNFOLD);
    if constexpr (ST == 3) {
      // LayerNorm folded into this linear (QKV / to_q of a transformer block): out = rstd[m] * (acc - mean[m] * c1[n]) + b'[n].  No
      // residual, no activation; tile pairs in the outer loop so that only one pair's bias / c1 vectors are live at a time.
      if (tile_full && !(fl & (CF_GEGLU | CF_MASK | CF_RES | CF_RES_F32 | CF_OUT_F32 | CF_RELU))) {
        const int wb = n0 + wn * (TN * 16);
        const int cp = wb + fq * 8, co = wb + (TN - 1) * 16 + fq * 4;
        float rs[TM], nm[TM];                     // rstd and -rstd * mean of this lane's rows
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float2 st = *(const float2*)(smem + LNS_OFF + c_slot2 * 1024 + (wm * (TM * 16) + i * 16 + fr) * 8);
          rs[i] = st.y * p.alpha; nm[i] = -st.y * st.x;
        }
        auto four = [&](const f32x4& a, const float4& b, const float4& c, int i) {
          const float v0 = __builtin_fmaf(rs[i], a[0], __builtin_fmaf(nm[i], c.x, b.x)), v1 = __builtin_fmaf(rs[i], a[1], __builtin_fmaf(nm[i], c.y, b.y));
          const float v2 = __builtin_fmaf(rs[i], a[2], __builtin_fmaf(nm[i], c.z, b.z)), v3 = __builtin_fmaf(rs[i], a[3], __builtin_fmaf(nm[i], c.w, b.w));
          return make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        };
        const unsigned char* bl = smem + BIAS_OFF + c_slot * 1024 + (wn * (TN * 16)) * 4;
        const unsigned char* cl = smem + C1_OFF + c_slot2 * 1024 + (wn * (TN * 16)) * 4;
#pragma unroll
        for (int t = 0; t < TN / 2; ++t) {
          const float4 b0 = *(const float4*)(bl + col_of(2 * t) * 4), b1 = *(const float4*)(bl + col_of(2 * t + 1) * 4);
          const float4 c0 = *(const float4*)(cl + col_of(2 * t) * 4), c1 = *(const float4*)(cl + col_of(2 * t + 1) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 16) + i * 16 + fr;
            if (m >= p.M) continue;
            const uint2 lo = four(acc[2 * t][i], b0, c0, i), hi = four(acc[2 * t + 1][i], b1, c1, i);
            *(uint4*)((bf16_t*)p.y + (size_t)m * p.y_ld + cp + 32 * t) = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
        }
        if constexpr (TN & 1) {
          const float4 b0 = *(const float4*)(bl + col_of(TN - 1) * 4), c0 = *(const float4*)(cl + col_of(TN - 1) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 16) + i * 16 + fr;
            if (m >= p.M) continue;
            *(uint2*)((bf16_t*)p.y + (size_t)m * p.y_ld + co) = four(acc[TN - 1][i], b0, c0, i);
          }
        }
        return;
      }
    }
    if (ST != 3 && fold_ok && tile_full && !(fl & (CF_GEGLU | CF_MASK | CF_RES_F32 | CF_OUT_F32)) && (!(fl & CF_RES) || !(p.res_ld & 7))) {
      // 16-byte loads / stores per tile pair; bias and every residual row are requested before the first use.  (Keeping the
      // bias in registers from the start of the item was measured slower: 20 VGPRs live across the K loop, tools/ab_ops.sh.)
      const int wb = n0 + wn * (TN * 16);
      float4 bv[TN];
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) bv[jn] = *(const float4*)(smem + BIAS_OFF + c_slot * 1024 + (wn * (TN * 16) + col_of(jn)) * 4);
      const int cp = wb + fq * 8, co = wb + (TN - 1) * 16 + fq * 4;       // pair t: cp + 32 t ; odd last tile: co
      uint4 rvp[TM][TN / 2];
      uint2 rvo[TM];
      if (fl & CF_RES) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = min(m0 + wm * (TM * 16) + i * 16 + fr, p.M - 1);
          const bf16_t* rp = (const bf16_t*)p.res + (size_t)m * p.res_ld;
#pragma unroll
          for (int t = 0; t < TN / 2; ++t) rvp[i][t] = *(const uint4*)(rp + cp + 32 * t);
          if constexpr (TN & 1) rvo[i] = *(const uint2*)(rp + co);
        }
      }
      float s1[TN][4], s2[TN][4];
      if (ST == 1 && (fl & CF_STATS)) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[jn][r] = 0.f; s2[jn][r] = 0.f; }
      }
      float rs1[ST == 2 ? TM : 1], rs2[ST == 2 ? TM : 1];     // ST == 2: per-row (sum, sum^2) of the stored values
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (TM * 16) + i * 16 + fr;
        if constexpr (ST == 2) { rs1[i] = 0.f; rs2[i] = 0.f; }
        if (m >= p.M) continue;
        bf16_t* yp = (bf16_t*)p.y + (size_t)m * p.y_ld;
        auto four = [&](const f32x4& a, const float4& b, unsigned r0, unsigned r1, float* t1, float* t2, int jn) {
          (void)jn;
          float v0 = a[0] * p.alpha + b.x, v1 = a[1] * p.alpha + b.y, v2 = a[2] * p.alpha + b.z, v3 = a[3] * p.alpha + b.w;
          if (fl & CF_RES) {
            v0 += __uint_as_float(r0 << 16); v1 += __uint_as_float(r0 & 0xffff0000u);
            v2 += __uint_as_float(r1 << 16); v3 += __uint_as_float(r1 & 0xffff0000u);
          }
          if (fl & CF_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
          if (ST == 1 && (fl & CF_STATS)) {
            t1[0] += v0; t1[1] += v1; t1[2] += v2; t1[3] += v3;
            t2[0] = __builtin_fmaf(v0, v0, t2[0]); t2[1] = __builtin_fmaf(v1, v1, t2[1]);
            t2[2] = __builtin_fmaf(v2, v2, t2[2]); t2[3] = __builtin_fmaf(v3, v3, t2[3]);
          }
          if constexpr (ST == 2) {
            rs1[i] += (v0 + v1) + (v2 + v3);
            rs2[i] = __builtin_fmaf(v0, v0, __builtin_fmaf(v1, v1, __builtin_fmaf(v2, v2, __builtin_fmaf(v3, v3, rs2[i]))));
          }
          return make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        };
#pragma unroll
        for (int t = 0; t < TN / 2; ++t) {
          const uint2 lo = four(acc[2 * t][i], bv[2 * t], rvp[i][t].x, rvp[i][t].y, s1[2 * t], s2[2 * t], 2 * t);
          const uint2 hi = four(acc[2 * t + 1][i], bv[2 * t + 1], rvp[i][t].z, rvp[i][t].w, s1[2 * t + 1], s2[2 * t + 1], 2 * t + 1);
          *(uint4*)(yp + cp + 32 * t) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        if constexpr (TN & 1) *(uint2*)(yp + co) = four(acc[TN - 1][i], bv[TN - 1], rvo[i].x, rvo[i].y, s1[TN - 1], s2[TN - 1], TN - 1);
      }
      if (ST == 1 && (fl & CF_STATS)) emit_stats(s1, s2, m0 + wm * (TM * 16), wb);
      if constexpr (ST == 2) {
        // a row's TN * 16 columns of this wave sit in the 4 lanes fr, fr + 16, fr + 32, fr + 48: two xor-shuffles, lanes 0-15 store
        // (sum, sum^2) of their rows into span (n-tile * WN + wn) of the row
        const int span = (n0 / BN) * WN + wn;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float a = rs1[i], q = rs2[i];
          a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
          a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
          const int m = m0 + wm * (TM * 16) + i * 16 + fr;
          if (fq == 0 && m < p.M) *(float2*)(p.rowpart + ((size_t)m * p.rowpart_ld + span) * 2) = make_float2(a, q);
        }
      }
      return;
    }
    if constexpr ((TN & 1) == 0) {
      if (fold_ok && tile_full && (fl & CF_GEGLU) && !(fl & (CF_MASK | CF_RES | CF_OUT_F32)) && !(p.raw_ld & 3)) {
        float4 bh[TN / 2], bg[TN / 2];
        float4 ch[ST == 3 ? TN / 2 : 1], cg[ST == 3 ? TN / 2 : 1];
        float rs[ST == 3 ? TM : 1], nm[ST == 3 ? TM : 1];      // ST == 3: rstd and -rstd * mean of this lane's rows
#pragma unroll
        for (int t = 0; t < TN / 2; ++t) {
          bh[t] = *(const float4*)(smem + BIAS_OFF + c_slot * 1024 + (ncol0 - n0 + 2 * t * 16) * 4);
          bg[t] = *(const float4*)(smem + BIAS_OFF + c_slot * 1024 + (ncol0 - n0 + 2 * t * 16 + 16) * 4);
          if constexpr (ST == 3) {
            ch[t] = *(const float4*)(smem + C1_OFF + c_slot2 * 1024 + (ncol0 - n0 + 2 * t * 16) * 4);
            cg[t] = *(const float4*)(smem + C1_OFF + c_slot2 * 1024 + (ncol0 - n0 + 2 * t * 16 + 16) * 4);
          }
        }
        if constexpr (ST == 3) {
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const float2 st = *(const float2*)(smem + LNS_OFF + c_slot2 * 1024 + (wm * (TM * 16) + i * 16 + fr) * 8);
            rs[i] = st.y * p.alpha; nm[i] = -st.y * st.x;
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + wm * (TM * 16) + i * 16 + fr;
          if (m >= p.M) continue;
#pragma unroll
          for (int t = 0; t < TN / 2; ++t) {
            const int nb = ncol0 + 2 * t * 16;
            float h0, h1, h2, h3, g0, g1, g2, g3;
            if constexpr (ST == 3) {
              // LayerNorm folded into the GEGLU projection: rstd * (acc - mean * c1) + b' on both halves
              h0 = __builtin_fmaf(rs[i], acc[2 * t][i][0], __builtin_fmaf(nm[i], ch[t].x, bh[t].x)); h1 = __builtin_fmaf(rs[i], acc[2 * t][i][1], __builtin_fmaf(nm[i], ch[t].y, bh[t].y));
              h2 = __builtin_fmaf(rs[i], acc[2 * t][i][2], __builtin_fmaf(nm[i], ch[t].z, bh[t].z)); h3 = __builtin_fmaf(rs[i], acc[2 * t][i][3], __builtin_fmaf(nm[i], ch[t].w, bh[t].w));
              g0 = __builtin_fmaf(rs[i], acc[2 * t + 1][i][0], __builtin_fmaf(nm[i], cg[t].x, bg[t].x)); g1 = __builtin_fmaf(rs[i], acc[2 * t + 1][i][1], __builtin_fmaf(nm[i], cg[t].y, bg[t].y));
              g2 = __builtin_fmaf(rs[i], acc[2 * t + 1][i][2], __builtin_fmaf(nm[i], cg[t].z, bg[t].z)); g3 = __builtin_fmaf(rs[i], acc[2 * t + 1][i][3], __builtin_fmaf(nm[i], cg[t].w, bg[t].w));
            } else {
              h0 = acc[2 * t][i][0] * p.alpha + bh[t].x; h1 = acc[2 * t][i][1] * p.alpha + bh[t].y;
              h2 = acc[2 * t][i][2] * p.alpha + bh[t].z; h3 = acc[2 * t][i][3] * p.alpha + bh[t].w;
              g0 = acc[2 * t + 1][i][0] * p.alpha + bg[t].x; g1 = acc[2 * t + 1][i][1] * p.alpha + bg[t].y;
              g2 = acc[2 * t + 1][i][2] * p.alpha + bg[t].z; g3 = acc[2 * t + 1][i][3] * p.alpha + bg[t].w;
            }
            if (fl & CF_GEGLU_RAW) {
              bf16_t* rp = p.raw + (size_t)m * p.raw_ld + nb;
              uint2 a, b;
              a.x = pack2bf(h0, h1); a.y = pack2bf(h2, h3); b.x = pack2bf(g0, g1); b.y = pack2bf(g2, g3);
              *(uint2*)rp = a; *(uint2*)(rp + 16) = b;
            }
            uint2 o;
            o.x = pack2bf(h0 * gelu_f(g0), h1 * gelu_f(g1)); o.y = pack2bf(h2 * gelu_f(g2), h3 * gelu_f(g3));
            *(uint2*)((bf16_t*)p.y + (size_t)m * p.y_ld + (nb >> 5) * 16 + (nb & 15)) = o;
          }
        }
        return;
      }
    }
    }  // FE
    float s1[TN][4], s2[TN][4];
    if (ST == 1 && (p.flags & CF_STATS)) {
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[jn][r] = 0.f; s2[jn][r] = 0.f; }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * (TM * 16) + i * 16 + fr;
      if (m >= p.M) continue;
      if (p.flags & CF_GEGLU) {
        if constexpr ((TN & 1) == 0) {
#pragma unroll
          for (int t = 0; t < TN / 2; ++t) {
            const int nb = n0 + wn * (TN * 16) + (2 * t) * 16 + fq * 4;
            if (nb >= p.N) continue;
            float h[4], g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { h[r] = acc[2 * t][i][r]; g[r] = acc[2 * t + 1][i][r]; }
            Epi::apply(p, bias, m, nb, h, g, nb + 16);
          }
        }
      } else {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int nb = n0 + wn * (TN * 16) + col_of(jn);
          if (nb >= p.N) continue;
          float h[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = acc[jn][i][r];
          Epi::apply(p, bias, m, nb, h, h, 0);     // leaves the stored values (before the bf16 rounding) in h
          if (ST == 1 && (p.flags & CF_STATS)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[jn][r] += h[r]; s2[jn][r] = __builtin_fmaf(h[r], h[r], s2[jn][r]); }
          }
        }
      }
    }
    if (ST == 1 && (p.flags & CF_STATS)) emit_stats(s1, s2, m0 + wm * (TM * 16), n0 + wn * (TN * 16));
  };

  // ---- pipeline: NS = 3 LDS stages, software-pipelined through registers.  K-step s reads its two 32-wide halves as
  // fragment sets F0 / F1:
  //     ds_read F1(s)            | hidden by
  //     MFMA    F0(s)            |
  //     s_waitcnt vmcnt ; s_barrier     -> stage s+1 has landed for every wave; every wave has issued all reads of stage s
  //     ds_read F0(s+1)          | hidden by
  //     DMA     step s+3 -> the stage of step s (waves 0-3 before, waves 4-7 after the MFMAs: SIMD partners alternate)
  //     MFMA    F1(s)            |
  // so no LDS latency is exposed to the MFMA stream, the stage of step s is re-filled as soon as its reads are issued (the loader
  // runs three K-steps ahead, across work items) and there is one barrier per K-step.  vmcnt counts every VMEM op of the wave in
  // issue order, so after an epilogue (whose loads/stores interleave with the DMAs) the next wait drains with vmcnt(0).
  // NS = 2 (the 4-wave form, two workgroups per CU): the stage of step s is re-filled with step s+2 and every wait is vmcnt(0);
  // the second workgroup of the CU covers the shorter prefetch distance and, above all, the other one's epilogue.
  auto wait_dma = [&](bool keep_one_stage) {
    if (keep_one_stage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  int cw = w_first;                                   // compute iterator
  int c_left = min(ksteps, (cw % p.ksplit) * per + per) - (cw % p.ksplit) * per;
  setup_loader(lw);
  prefetch_tap();
  issue_step(0);
  bool more = advance_loader();
  prefetch_tap();
  int ahead = 0;                                      // DMA stages in flight (after the first wait: including the one needed next)
  if (NS == 3 && more) { issue_step(1); more = advance_loader(); prefetch_tap(); ahead = 1; }
  wait_dma(ahead == 1);
  if (more) { issue_step(NS - 1); more = advance_loader(); prefetch_tap(); ahead = NS - 1; }
  zero_acc();
  Frags F0, F1;
  load_frags(F0, 0, 0);
  const bool issue_first = NW == 4 || wave < 4;       // 8 waves: SIMD partners run DMA issue and MFMAs in opposite order
  int cur = 0;
  bool drained = false;
#ifdef DD_TRACE
  trace_stamp(tn, 0);
#endif
  // Every LDS read of the loop body is unconditional (the last K-step of the work list reads a stale stage into F0 and takes one
  // more barrier; `ahead` may go negative there, which only selects the vmcnt(0) form of the wait): a conditional ds_read makes the compiler's waitcnt pass fall back to lgkmcnt(0) at the join, which would
  // expose the F0 read latency in front of every MFMA(F1) block.
  while (true) {
    // K loop of one work item (the loader is up to three K-steps ahead, possibly already in the next item)
    for (int left = c_left; left > 0; --left) {
      load_frags(F1, cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(F0);
      __builtin_amdgcn_sched_barrier(0);
      const int nxt = cur == NS - 1 ? 0 : cur + 1;
      // stage s+1 must have landed; `ahead` counts the stages in flight including it (steady state: two)
      wait_dma(ahead >= 2 && !drained);
      drained = false;
      --ahead;
      load_frags(F0, nxt, 0);
      const bool refill = more;
      if (refill && issue_first) issue_step(cur);
      __builtin_amdgcn_sched_barrier(0);
      mma(F1);
      __builtin_amdgcn_sched_barrier(0);
      if (refill) {
        if (!issue_first) issue_step(cur);
        more = advance_loader();
        ++ahead;
      }
      prefetch_tap();
      cur = nxt;
    }
#ifdef DD_TRACE
    trace_stamp(tn, 1);
#endif
    epilogue(cw);
    c_slot = c_slot == 2 ? 0 : c_slot + 1;
    if constexpr (ST == 3) c_slot2 ^= 1;
#ifdef DD_TRACE
    trace_stamp(tn, 2);
#endif
    drained = true;
    cw += Gx;
    if (cw >= w_end) break;
    c_left = min(ksteps, (cw % p.ksplit) * per + per) - (cw % p.ksplit) * per;
    zero_acc();
  }
#ifdef DD_TRACE
  trace_stamp(tn, 4);
#endif
}

template <int WM, int WN, int TM, int TN, int NS, bool FE, int FM, int ST>
hipError_t run_big_fe3(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  constexpr int lds = NS * (BM + BN) * 128 + 256 + (ST == 3 ? 7 : 3) * 1024;   // + per-tap tables (<= 32 taps: packed (dy,dx) and byte offsets), bias (+ c1) ring
  static_assert(lds <= 163840, "LDS budget");
  static bool attr = false;
  if (!attr) { hipFuncSetAttribute((const void*)conv_gemm_big_kernel<WM, WN, TM, TN, NS, FE, FM, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  const int W = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.ksplit;
  constexpr int WGS = 256 * ((WM * WN == 4) ? 2 : 1);   // persistent: one 8-wave or two 4-wave workgroups per CU
  int G = W < WGS ? (W + 7) / 8 * 8 : WGS;
  hipLaunchKernelGGL((conv_gemm_big_kernel<WM, WN, TM, TN, NS, FE, FM, ST>), dim3(G), dim3(WM * WN * 64), lds, stream, p);
  return hipGetLastError();
}
template <int WM, int WN, int TM, int TN, int NS, bool FE, int FM>
hipError_t run_big_fe2(const ConvGemmParams& p, hipStream_t stream) {
  // the LayerNorm forms only exist for the batched epilogue of the two-workgroup (shallow-K, pointwise) tiles: conv_gemm_ln_form()
  if constexpr (WM * WN == 4 && FE && FM == 1) {
    if ((p.flags & CF_LNFOLD) && p.K >= 192 && !(p.M & 1)) return run_big_fe3<WM, WN, TM, TN, NS, FE, FM, 3>(p, stream);
    if (p.flags & CF_ROWSTATS) return run_big_fe3<WM, WN, TM, TN, NS, FE, FM, 2>(p, stream);
  }
  return (p.flags & CF_STATS) ? run_big_fe3<WM, WN, TM, TN, NS, FE, FM, 1>(p, stream) : run_big_fe3<WM, WN, TM, TN, NS, FE, FM, 0>(p, stream);
}

template <int WM, int WN, int TM, int TN, int NS, bool FE>
hipError_t run_big_fe(const ConvGemmParams& p, hipStream_t stream) {
  // the fast staging path addresses the input through 32-bit BYTE offsets of a buffer resource (plus the tap bias): inputs of
  // 3.75 GB and more take the general path (32-bit ELEMENT offsets; launch_conv_gemm rejects inputs beyond those)
  const size_t x_bytes = (size_t)p.B * p.H * p.W * (size_t)p.x_ld * 2;
  const bool fast = (p.cin & 63) == 0 && p.shift == 0 && p.ntaps <= 32 && x_bytes < 0xF0000000ull;
  return fast ? run_big_fe2<WM, WN, TM, TN, NS, FE, 1>(p, stream) : run_big_fe2<WM, WN, TM, TN, NS, FE, 0>(p, stream);
}
template <int WM, int WN, int TM, int TN, int NS>
hipError_t run_big(const ConvGemmParams& p, hipStream_t stream) {
  // Two instantiations per tile.  The batched epilogue (FE: bias from the LDS ring, all residual loads issued together, 16-byte
  // stores) saves ~15 us per work item over the generic one; its instantiation used to run the K loop ~0.18 us per K-step slower
  // (register allocation), which made the generic form the better one beyond ~100 K-steps per item.  With the bias staged through
  // LDS the batched form wins at every depth (same-device tools/ab_ops.sh, 32-image batch: limit 60 -> 1975 ms of conv, 100 -> 1949,
  // none -> 1936), so only split-K items (fp32 partial stores, no epilogue work) take the generic instantiation.
  const int steps_per_item = (p.K / 64 + p.ksplit - 1) / p.ksplit;
#ifndef DD_FE_LIMIT
#define DD_FE_LIMIT 1000000
#endif
  // split-K items only store fp32 partials (same code in both instantiations): take the faster loop
  return (steps_per_item <= DD_FE_LIMIT && p.ksplit == 1) ? run_big_fe<WM, WN, TM, TN, NS, true>(p, stream) : run_big_fe<WM, WN, TM, TN, NS, false>(p, stream);
}

}  // namespace

#ifdef DD_TRACE
extern "C" int dd_debug_clear_trace() {
  static unsigned long long z[512];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), z, sizeof(z));
}
extern "C" int dd_debug_read_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n);
}
#endif

// tile choice for the big kernel: 0 = not applicable (use the 128x128 kernel), else config id
int conv_gemm_big_config(int M, int N, int K, int flags) {
  if (M < 1024 || N < 64 || K < 128) return 0;
  const bool geglu = (flags & CF_GEGLU) != 0;
  // Items of up to ~100 K-steps run as two independent 4-wave workgroups per CU (NS = 2, half the tile): one workgroup's epilogue
  // and pipeline refill overlap the other's MFMAs, which a single 8-wave workgroup in lock step cannot do.  Same-device A/B on
  // the bench workload (tools/ab_ops.sh): conv 1078 -> 1029 ms per step; deeper items and the big VAE shapes prefer the larger
  // tile (arithmetic intensity).  DD_CONV_2WG / DD_CONV_2WG128 override the K-step limits (0 disables) for A/B runs.
  static const int two_wg = getenv("DD_CONV_2WG") ? atoi(getenv("DD_CONV_2WG")) : 100;
  if (two_wg > 0 && K / 64 <= two_wg && !geglu && N % 160 == 0) return 4;              // 128 x 160, 4 waves, 2 per CU
  static const int two_wg128 = getenv("DD_CONV_2WG128") ? atoi(getenv("DD_CONV_2WG128")) : 24;
  if (two_wg128 > 0 && K / 64 <= two_wg128 && N % 128 == 0 && (geglu || N % 160 != 0)) return 5;   // 128 x 128, 4 waves, 2 per CU
  if (N <= 128) return 3;                          // 256 x 128
  if (!geglu && N % 160 == 0 && (N % 256 != 0 || N == 1280)) return 2;   // 256 x 160
  if (N % 256 == 0 || N >= 1024) return 1;         // 128 x 256
  if (!geglu && N % 160 == 0) return 2;
  return 0;
}

// columns per wave of a tile (the span of one CF_ROWSTATS partial): 0 unless the configuration has a row-statistics form
int conv_gemm_big_rowstat_span(int cfg) { return cfg == 4 ? 80 : cfg == 5 ? 64 : 0; }

void conv_gemm_big_tile(int cfg, int* bm, int* bn) {
  *bm = (cfg == 1 || cfg == 4 || cfg == 5) ? 128 : 256;
  *bn = cfg == 1 ? 256 : (cfg == 2 || cfg == 4) ? 160 : 128;
}

hipError_t launch_conv_gemm_big(const ConvGemmParams& p, int cfg, hipStream_t stream) {
  switch (cfg) {
    // three LDS stages + counted vmcnt + partner-wave stagger: +0..7 % over two stages in same-device A/B (tools/ab_conv.sh)
    case 1: return run_big<2, 4, 4, 4, 3>(p, stream);    // 128 x 256, 8 waves
    case 2: return run_big<4, 2, 4, 5, 3>(p, stream);    // 256 x 160, 8 waves
    case 3: return run_big<4, 2, 4, 4, 3>(p, stream);    // 256 x 128, 8 waves
    case 4: return run_big<2, 2, 4, 5, 2>(p, stream);    // 128 x 160, 4 waves, two workgroups per CU (shallow K)
    case 5: return run_big<2, 2, 4, 4, 2>(p, stream);    // 128 x 128, 4 waves, two workgroups per CU (GEGLU / N % 128 == 0)
    default: return hipErrorInvalidValue;
  }
}
