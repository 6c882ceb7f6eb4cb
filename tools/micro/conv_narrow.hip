// K1 (narrow outputs) -- 3x3 / stride-1 convolutions with at most 16 output channels: the AutoencoderKL decoder's conv_out (128 -> 3 at
// 512 x 512: generate_data.py:701 / :1223 vae.decode) and the UNet's conv_out (320 -> 4 at 64 x 64: :112), bf16 MFMA, gfx950.
//
// Why: the tiled implicit-GEMM kernels give a wave 64 output columns; with N = 3 / 4 they spend 60 of 64 MFMA columns on zero weights
// and run at 24 - 31 TFLOP/s -- 7.3 + 5.3 ms of a bench step for layers that only have to stream their input once (2.1 GB at 512 x 512:
// 0.4 ms at the HBM rate).  Here a wave owns 16 output pixels of one image row and ONE 16-column MFMA tile (4x less matrix work); the
// whole packed weight matrix (16 rows x K, 36 - 92 KB) sits in LDS for the lifetime of a persistent workgroup; the input is not staged at
// all: every (tap, 32-channel chunk) fragment -- 16 pixels x 64 B -- is one 16-byte buffer load per lane straight into the MFMA's B
// operand, zero padding by the buffer range check.  The nine taps re-read the same rows from L1 / L2 (9x the input bytes on the
// L1 side, ~1x on the HBM side); nothing is written but the N valid channels.
// Same packed weights / tap table / NHWC rows / bias / fp32-or-bf16 output semantics as launch_conv_gemm's other kernels.
//
// ROUND 5: BUILT, PARITY GREEN (tests/test_kernels_gpu.py conv cases n3_vae_conv_out / n4_unet_conv_out and the fp32 padded-row case ran
// through it), MEASURED, NOT KEPT -- moved out of the library.  tools/bench_narrow.py on one MI355X: 128 -> 3 at 32 x 512 x 512 2528 us
// against 2420 us for the tiled small kernel (256 x 64 tiles), 320 -> 4 at 64 x 64 x 64 204 against 196 us.  Fragment-shaped loads (16
// rows x 64 B per instruction) straight into registers are latency-bound here: hipcc keeps two to four of them in flight per wave
// whatever the source order (it hoists the weight fragments into 144 VGPRs or, held back, re-uses the same few destination registers),
// and 16 waves per CU do not cover an L2 round trip per tap.  What this layer needs is the halo-resident staging of conv_halo.hip with a
// 16-column wave tile, not direct loads.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// CH32 = cin / 32 (4: cin 128, 10: cin 320); NW waves per workgroup, WPS waves per SIMD (the register budget: 512 / WPS VGPRs)
template <int CH32, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void conv_narrow_kernel(ConvGemmParams p) {
  constexpr int CIN = CH32 * 32, K = 9 * CIN, ROWB = K * 2 + 16;      // LDS row of one output channel: K bf16 + 16 B (bank spread)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  // ---- weights: rows n < N of the packed [N][K] matrix, zero rows above
  for (int i = tid; i < 16 * (K / 8); i += NW * 64) {
    const int n = i / (K / 8), c = i - n * (K / 8);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < p.N) v = *(const uint4*)(p.w + (size_t)n * p.K + c * 8);
    *(uint4*)(smem + n * ROWB + c * 16) = v;
  }
  int dyx[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) dyx[t] = p.taptab[t];
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = ((p.flags & CF_BIAS) && fq * 4 + r < p.N) ? p.bias[fq * 4 + r] : 0.f;
  __syncthreads();

  const int W = p.W, H = p.H, gpr = W >> 4;                 // 16-pixel groups per image row
  const int groups = p.M >> 4;
  const size_t img_elems = (size_t)H * W * p.x_ld;
  const unsigned char* wl = smem + fr * ROWB + fq * 16;
  for (int g = blockIdx.x * NW + wave; g < groups; g += gridDim.x * NW) {
    const int rowi = g / gpr, x0 = (g - rowi * gpr) << 4;   // rowi = image * H + y
    const int b = rowi / H, y = rowi - b * H;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * img_elems), 0, (unsigned)(img_elems * 2), 0x00020000);
    // (the weight fragments are re-read from LDS per group: hoisted out of this loop they take 36 x 4 VGPRs and spill)
    const unsigned char* wlg = wl;
    asm volatile("" : "+v"(wlg));
    auto tap_off = [&](int t) {
      const int dy = ((dyx[t] >> 6) & 63) - 32, dx = (dyx[t] & 63) - 32;
      const int yy = y + dy, xx = x0 + fr + dx;
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      return ok ? ((unsigned)(yy * W + xx) * (unsigned)p.x_ld + (unsigned)(fq * 8)) * 2u : 0xfffffff0u;
    };
    // the fragments of tap t + 1 are requested before the MFMAs of tap t (two register sets): with 16 waves per CU that keeps
    // 2 x CH32 KB per wave in flight, enough to cover the L2 latency of the re-read rows
    u32x4 xa[CH32], xb[CH32];
    {
      const unsigned off = tap_off(0);
#pragma unroll
      for (int c = 0; c < CH32; ++c) xa[c] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, c * 64, 0);
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      u32x4* cur = (t & 1) ? xb : xa;
      u32x4* nxt = (t & 1) ? xa : xb;
      if (t + 1 < 9) {
        const unsigned off = tap_off(t + 1);
#pragma unroll
        for (int c = 0; c < CH32; ++c) nxt[c] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, c * 64, 0);
      }
#pragma unroll
      for (int c = 0; c < CH32; ++c) {
        // packed K order for cin % 64 == 0 (weights.cpp): (64-channel chunk, tap, channel in chunk)
        const bf16x8 wf = *(const bf16x8*)(wlg + ((((c >> 1) * 9 + t) * 64) + (c & 1) * 32) * 2);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, cur[c]), acc, 0, 0, 0);
      }
    }
    // lane (fr, fq): output pixel x0 + fr, channels fq * 4 .. + 3
    const size_t m = (size_t)rowi * W + x0 + fr;
    const int n0 = fq * 4;
    if (n0 < p.N) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[r] * p.alpha + bias[r];
      if (p.flags & CF_OUT_F32) {
        float* yp = (float*)p.y + m * p.y_ld + n0;
        if (n0 + 4 <= p.N && !(p.y_ld & 3)) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < p.N) yp[r] = v[r];
      } else {
        bf16_t* yp = (bf16_t*)p.y + m * p.y_ld + n0;
        if (n0 + 4 <= p.N && !(p.y_ld & 3)) *(uint2*)yp = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        else
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < p.N) yp[r] = f2bf(v[r]);
      }
    }
  }
}

template <int CH32, int NW, int WPS>
hipError_t run_narrow(const ConvGemmParams& p, hipStream_t stream) {
  const int lds = 16 * (9 * CH32 * 32 * 2 + 16);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)conv_narrow_kernel<CH32, NW, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr = true; }
  static const int cus = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
  const int per_cu = WPS * 4 / NW;                         // workgroups per CU the register / LDS budget admits
  const int groups = p.M >> 4;
  int grid = cus * per_cu;
  if (grid * NW > groups) grid = (groups + NW - 1) / NW;
  hipLaunchKernelGGL((conv_narrow_kernel<CH32, NW, WPS>), dim3(grid), dim3(NW * 64), lds, stream, p);
  return hipGetLastError();
}

}  // namespace

// 0 = not eligible, else cin / 32
int conv_narrow_config(const ConvGemmParams& p) {
  static const int on = getenv("DD_CONV_NARROW") ? atoi(getenv("DD_CONV_NARROW")) : 1;
  if (!on || p.force_small || p.N > 16 || p.ntaps != 9 || p.stride != 1 || p.shift || p.parity || p.H != p.Ho || p.W != p.Wo) return 0;
  if ((p.cin != 128 && p.cin != 320) || p.K != 9 * p.cin || (p.W & 15) || p.M != p.B * p.H * p.W || p.M < 65536 || p.ksplit > 1 || p.bias_sel) return 0;
  if (p.flags & ~(CF_BIAS | CF_OUT_F32)) return 0;
  if ((p.x_ld & 7) || (size_t)p.H * p.W * p.x_ld * 2 >= 0xF0000000ull) return 0;
  return p.cin / 32;
}
hipError_t launch_conv_narrow(const ConvGemmParams& p, int ch32, hipStream_t stream) {
  // cin 128: 37 KB of weights, four 4-wave workgroups per CU (128 VGPRs); cin 320: 92 KB, one 8-wave workgroup (256 VGPRs)
  return ch32 == 4 ? run_narrow<4, 4, 4>(p, stream) : run_narrow<10, 8, 2>(p, stream);
}
