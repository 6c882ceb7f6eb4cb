// Micro-benchmark of K-loop forms with ONE wave per SIMD and large per-wave tiles (accumulators in AccVGPRs), against the production
// 8-wave form of conv_gemm2.hip (mfma_loop.hip k2).  Same LDS layout (128-byte rows, XOR swizzle), same LDS-DMA staging with the conv
// gather pattern on random data.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_loop2 mfma_loop2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// WM x WN waves, each TM x TN MFMA tiles of 16x16; NS LDS stages of one 64-wide K-step
template <int WM, int WN, int TM, int TN, int NS, int IL>
__global__ __launch_bounds__(WM* WN * 64, 1) void kw(const unsigned char* src, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = WM * WN, NT = NW * 64, BM = WM * TM * 16, BN = WN * TN * 16, BUF = (BM + BN) * 128;
  constexpr int PIECES = (BM + BN) / 8 / NW;     // 1 KB pieces per wave per stage
  constexpr int APIECES = BM / 8 / NW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, fr = lane & 15, fq = lane >> 4;
  {
    unsigned st = 1234567u + tid * 7919u + blockIdx.x * 104729u;
    for (int i = tid; i < NS * BUF / 4; i += NT) {
      st = st * 1664525u + 1013904223u;
      const unsigned a = 0x3f800000u | (st & 0x807fffffu), b = (st * 2654435761u);
      ((unsigned*)smem)[i] = (a >> 16) | ((0x3f80u | (b & 0x807f)) << 16);
    }
  }
  __syncthreads();
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* gsrc2 = src + (size_t)(blockIdx.x & 127) * (256 * 640) + (size_t)(lane >> 3) * 640 + (lane & 7) * 16 + 65 * 640;
  struct Fr { bf16x8 wf[TN]; bf16x8 xf[TM]; };
  auto loadf = [&](Fr& F, int buf, int ks) {
    const unsigned char* A = smem + buf * BUF;
    const unsigned char* Bt = A + BM * 128;
    const int slot = fq + 4 * ks;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) { const int row = wn * (TN * 16) + jn * 16 + fr; F.wf[jn] = *(const bf16x8*)(Bt + row * 128 + ((slot ^ (row & 7)) << 4)); }
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * (TM * 16) + i * 16 + fr; F.xf[i] = *(const bf16x8*)(A + row * 128 + ((slot ^ (row & 7)) << 4)); }
  };
  auto mma = [&](const Fr& F) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.wf[jn], F.xf[i], acc[jn][i], 0, 0, 0);
  };
  auto dma = [&](int s, int buf) {
    const int tap = s % 9, chunk = (s / 9) % 5;
    const long toff = ((tap / 3 - 1) * 64 + (tap % 3 - 1)) * 640 + chunk * 128;
#pragma unroll
    for (int i = 0; i < APIECES; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc2 + toff + (size_t)((i * NW + wave) * 8 % 256) * 640),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + (i * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = APIECES; i < PIECES; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (48u << 20) + (size_t)((((i - APIECES) * NW + wave) * 8 + (lane >> 3)) * 5760 + (s % 45) * 128 + (lane & 7) * 16)),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + (i * NW + wave) * 1024), 16, 0, 0);
  };
  Fr F0, F1;
  for (int s = 0; s < NS - 1; ++s) dma(s, s + 1 == NS ? 0 : s + 1);
  loadf(F0, 0, 0);
  int cur = 0;
  for (int s = 0; s < steps; ++s) {
    const int nxt = cur == NS - 1 ? 0 : cur + 1;
    loadf(F1, cur, 1);
    if (!IL) __builtin_amdgcn_sched_barrier(0);
    mma(F0);
    if (IL) {
#pragma unroll
      for (int q = 0; q < TM + TN; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, IL, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - IL * (TM + TN), 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    loadf(F0, nxt, 0);
    dma(s + NS - 1, cur);
    if (!IL) __builtin_amdgcn_sched_barrier(0);
    mma(F1);
    if (IL) {
#pragma unroll
      for (int q = 0; q < TM + TN; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, IL, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
      for (int q = 0; q < PIECES; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - IL * (TM + TN) - PIECES, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[(size_t)blockIdx.x * NT + tid] = sum;
}

template <int WM, int WN, int TM, int TN, int NS, int IL>
void run(const unsigned char* src, float* out, int steps, const char* name) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  const int lds = NS * (BM + BN) * 128;
  auto f = kw<WM, WN, TM, TN, NS, IL>;
  hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(f, dim3(256), dim3(WM * WN * 64), lds, 0, src, out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(f, dim3(256), dim3(WM * WN * 64), lds, 0, src, out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = 256.0 * steps * 2.0 * BM * BN * 64;
  printf("%-44s %8.1f us  %7.1f TF/s  %6.0f ns/K-step  %5.0f ns per 5.24 MFLOP  (%s)\n", name, ms * 1e3, flops / ms / 1e9, ms * 1e6 / steps,
         ms * 1e6 / steps * 5.24288e6 / (2.0 * BM * BN * 64), hipGetErrorString(hipGetLastError()));
}

int main() {
  unsigned char* src; float* out;
  hipMalloc(&src, 64 << 20);
  {
    std::vector<unsigned short> h(32 << 20);
    unsigned st = 42;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (unsigned short)(0x3f80u | ((st >> 9) & 0x807f)); }
    hipMemcpy(src, h.data(), 64 << 20, hipMemcpyHostToDevice);
  }
  hipMalloc(&out, 256 * 512 * 4);
  const int steps = 2000;
  run<4, 2, 4, 5, 3, 0>(src, out, steps, "8 waves 256x160 (64x80 per wave), 3 stages");
  run<4, 2, 4, 5, 3, 1>(src, out, steps, "8 waves 256x160, interleaved 1:1");
  run<2, 2, 8, 5, 3, 0>(src, out, steps, "4 waves 256x160 (128x80 per wave), 3 stages");
  run<2, 2, 8, 5, 3, 1>(src, out, steps, "4 waves 256x160 (128x80), interleaved 1:1");
  run<2, 2, 6, 10, 2, 0>(src, out, steps, "4 waves 192x320 (96x160 per wave), 2 stages");
  run<2, 2, 6, 10, 2, 1>(src, out, steps, "4 waves 192x320, interleaved 1:1");
  run<2, 2, 6, 10, 2, 2>(src, out, steps, "4 waves 192x320, interleaved 2:1");
  run<2, 2, 8, 8, 2, 0>(src, out, steps, "4 waves 256x256 (128x128 per wave), 2 stages");
  run<2, 2, 8, 8, 2, 1>(src, out, steps, "4 waves 256x256, interleaved 1:1");
  run<2, 2, 8, 8, 2, 2>(src, out, steps, "4 waves 256x256, interleaved 2:1");
  run<4, 1, 8, 8, 2, 1>(src, out, steps, "4 waves 512x128 (128x128), interleaved 1:1");
  return 0;
}
