// Micro-benchmark: back-to-back issue rate of v_mfma_f32_16x16x16_bf16 (the K = 16 legacy form) against v_mfma_f32_16x16x32_bf16 on gfx950,
// one wave per SIMD, 4 independent accumulator chains.  Question behind it: the d = 40 attention forward pads the QK^T
// contraction to 64; would a K = 16 instruction for dims 32..47 save MFMA cycles?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_k16 tools/micro/mfma_k16.hip && ./mfma_k16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* in, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  uint4 u = *(const uint4*)(in + lane * 4);
  bf16x8 a8 = __builtin_bit_cast(bf16x8, u), b8 = a8;
  s16x4 a4 = {(short)u.x, (short)u.y, (short)u.z, (short)u.w}, b4 = a4;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  int* in; float* out;
  hipMalloc(&in, 4096 * 4); hipMemset(in, 0, 4096 * 4);
  hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 20000, blocks = 256;       // one 4-wave workgroup per CU
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 32;
    printf("%s: %.3f ms, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", mode == 0 ? "16x16x32_bf16" : "16x16x16_bf16", ms, ms * 1e6 / mfmas, ms * 1e6 / mfmas * 2.4);
  }
  return 0;
}
