// Micro-benchmark for BASELINE.json configs[4]'s "fp8 MFMA attention": what can an fp8 P.V buy a d = 64 flash-attention forward?
// Two register-only inner loops of one wave (32 queries, one 128-key tile per iteration; no LDS, no global traffic: an UPPER bound on
// the gain), identical up to the P.V product:
//   bf16:  S^T = K Q^T  32 x v_mfma_f32_16x16x32_bf16 | online softmax on 64 scores per lane | P -> bf16 | O += V^T P   32 x 16x16x32_bf16
//   fp8 :  the same QK^T and softmax                                                        | P -> e4m3 |  O += V^T P    8 x
//          v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, unit block scales: 2x the bf16 rate per MAC, MI355X_MICROARCH.md)
// The fp8 form contracts over 128 KEYS per instruction, so the d = 64 head dim is no obstacle for P.V (it is for QK^T, whose
// contraction is over d).  V would have to be stored as e4m3 too (3 mantissa bits).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o pv_fp8 tools/micro/pv_fp8.hip && ./pv_fp8
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <bool FP8>
__global__ __launch_bounds__(256) void pv_kernel(const int* in, float* out, int tiles) {
  const int lane = threadIdx.x & 63;
  // operands that would come from LDS: a few distinct register sets, seeded from memory so that nothing folds
  bf16x8 kf[2], qf[2][2], vf[4];
  i32x8 vf8[4];
  for (int i = 0; i < 2; ++i) {
    uint4 u = *(const uint4*)(in + (lane * 4 + i * 256) % 4096);
    kf[i] = __builtin_bit_cast(bf16x8, u);
    for (int q = 0; q < 2; ++q) { uint4 w = *(const uint4*)(in + (lane * 4 + i * 512 + q * 128 + 1024) % 4096); qf[q][i] = __builtin_bit_cast(bf16x8, w); }
  }
  for (int i = 0; i < 4; ++i) {
    uint4 u = *(const uint4*)(in + (lane * 4 + i * 64 + 2048) % 4096);
    vf[i] = __builtin_bit_cast(bf16x8, u);
    for (int e = 0; e < 8; ++e) vf8[i][e] = in[(lane * 8 + e + i * 512) % 4096] & 0x3f3f3f3f;      // finite e4m3 bytes
  }
  f32x4 o[2][4];
  float mrun[2] = {-1e30f, -1e30f}, lsum[2] = {0.f, 0.f};
  for (int q = 0; q < 2; ++q)
    for (int d = 0; d < 4; ++d) o[q][d] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float sl2 = 0.18f;
  for (int t = 0; t < tiles; ++t) {
    f32x4 st[2][8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
      for (int q = 0; q < 2; ++q) st[q][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q) st[q][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[q][ks], st[q][kt], 0, 0, 0);
    }
    bf16x8 pf[2][4];
    i32x8 pf8[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float mx = st[q][0][0];
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[q][kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[q], mx * sl2);
      const float alpha = __builtin_amdgcn_exp2f(mrun[q] - mnew);
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[q][kt][r], sl2, -mnew));
          st[q][kt][r] = e;
          ps += e;
        }
      lsum[q] = lsum[q] * alpha + ps;
      mrun[q] = mnew;
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int d = 0; d < 4; ++d) o[q][d] *= alpha;
      }
      if (FP8) {
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
          int w = __builtin_amdgcn_cvt_pk_fp8_f32(st[q][kt][0], st[q][kt][1], 0, false);
          pf8[q][kt] = __builtin_amdgcn_cvt_pk_fp8_f32(st[q][kt][2], st[q][kt][3], w, true);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          uint4 u;
          u.x = pack2bf(st[q][2 * c][0], st[q][2 * c][1]); u.y = pack2bf(st[q][2 * c][2], st[q][2 * c][3]);
          u.z = pack2bf(st[q][2 * c + 1][0], st[q][2 * c + 1][1]); u.w = pack2bf(st[q][2 * c + 1][2], st[q][2 * c + 1][3]);
          pf[q][c] = __builtin_bit_cast(bf16x8, u);
        }
      }
    }
    if (FP8) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          o[q][d] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vf8[d], pf8[q], o[q][d], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 2; ++q) o[q][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[(c + d) & 3], pf[q][c], o[q][d], 0, 0, 0);
    }
  }
  float acc = lsum[0] + lsum[1];
  for (int q = 0; q < 2; ++q)
    for (int d = 0; d < 4; ++d) acc += o[q][d][0] + o[q][d][3];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <bool FP8>
static float run(int wgs_per_cu, int tiles) {
  int* in; float* out;
  hipMalloc(&in, 4096 * 4 + 64); hipMalloc(&out, 256 * 8 * 256 * 4);
  int h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0x3c003c00 + (i * 2654435761u & 0x00ff00ff);
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  const int grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((pv_kernel<FP8>), dim3(grid), dim3(256), 0, 0, in, out, tiles);
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pv_kernel<FP8>), dim3(grid), dim3(256), 0, 0, in, out, tiles);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(in); hipFree(out);
  return ms / 3;
}

int main() {
  const int tiles = 2048;
  for (int w = 1; w <= 3; ++w) {
    const float b = run<false>(w, tiles), f = run<true>(w, tiles);
    // per SIMD: w waves, each `tiles` iterations of 32 queries x 128 keys x d 64: 4 * 32 * 128 * 64 FLOP
    const double fl = 4.0 * 32 * 128 * 64 * tiles * 4.0 * w * 256;
    printf("%d wave(s) per SIMD: bf16 P.V %.3f ms (%.0f TFLOP/s)  fp8 P.V %.3f ms (%.0f TFLOP/s)  ratio %.3f\n", w, b, fl / b / 1e9, f, fl / f / 1e9, b / f);
  }
  return 0;
}
