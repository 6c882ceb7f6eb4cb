// Micro-benchmark of the conv_gemm main-loop ingredients on gfx950 (one 512-thread workgroup per CU):
//   mode bit 0: ds_read_b128 fragment loads (18 per wave per K-step, same addresses as conv_gemm2.hip)
//   mode bit 1: s_barrier per K-step
//   mode bit 2: LDS-DMA of the stage (7 x 1 KB pieces per wave per K-step from an L2-resident buffer)
// Always: 40 v_mfma_f32_16x16x32_bf16 per wave per K-step (TN=5 x TM=4 x 2).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_loop mfma_loop.hip ; run: ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* src, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TN = 5, TM = 4, BM = 256, BN = 160, BUF = (BM + BN) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / 2, wn = wave % 2, fr = lane & 15, fq = lane >> 4;
  if (MODE & 8) {   // random bf16 payload (power / clock effect of real data)
    unsigned st = 1234567u + tid * 7919u + blockIdx.x * 104729u;
    for (int i = tid; i < 3 * BUF / 4; i += 512) {
      st = st * 1664525u + 1013904223u;
      const unsigned a = 0x3f800000u | (st & 0x807fffffu), b = (st * 2654435761u);
      ((unsigned*)smem)[i] = (a >> 16) | ((0x3f80u | (b & 0x807f)) << 16);
    }
  } else
    for (int i = tid; i < 3 * BUF / 4; i += 512) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 wf[TN], xf[TM];
#pragma unroll
  for (int a = 0; a < TN; ++a) wf[a] = *(const bf16x8*)(smem + (a * 16 + fr) * 128 + fq * 16);
#pragma unroll
  for (int b = 0; b < TM; ++b) xf[b] = *(const bf16x8*)(smem + 8192 + (b * 16 + fr) * 128 + fq * 16);
  const unsigned char* gsrc = src + ((size_t)blockIdx.x * 64 + lane) * 16;
  // bit 4: conv-like gather: a piece = 8 pixel rows x 128 B at a 640 B pixel stride (320-channel NHWC), 9 taps re-read the
  // same rows shifted by (dy*64+dx) pixels, chunk = 128 B column; per block a 256-pixel tile of a 64x64 image
  const unsigned char* gsrc2 = src + (size_t)(blockIdx.x & 127) * (256 * 640) + (size_t)(lane >> 3) * 640 + (lane & 7) * 16 + 65 * 640;
  int cur = 0;
  for (int s = 0; s < steps; ++s) {
    const unsigned char* A = smem + cur * BUF;
    const unsigned char* Bt = A + BM * 128;
    if (MODE & 4) {
      if (MODE & 16) {
        const int tap = s % 9, chunk = (s / 9) % 5;
        const long toff = ((tap / 3 - 1) * 64 + (tap % 3 - 1)) * 640 + chunk * 128;
        if (!(MODE & 32))   // bit 5: no A pieces (what a halo slab reused by the nine taps would leave: ~0.6 of 4 per K-step)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc2 + toff + (size_t)(i * 64 + wave * 8) * 640),
                                           (__attribute__((address_space(3))) void*)(smem + ((cur + 2) % 3) * BUF + (i * 8 + wave) * 1024), 16, 0, 0);
        if (!(MODE & 64))   // bit 6: no weight pieces
#pragma unroll
        for (int i = 4; i < 7; ++i)   // weights: rows of 128 B at a 5760 B stride, shared by all blocks
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (48u << 20) + (size_t)(((i - 4) * 64 + wave * 8 + (lane >> 3)) * 5760 + (s % 45) * 128 + (lane & 7) * 16)),
                                           (__attribute__((address_space(3))) void*)(smem + ((cur + 2) % 3) * BUF + (i * 8 + wave) * 1024), 16, 0, 0);
      } else
#pragma unroll
      for (int i = 0; i < 7; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)((s * 7 + i) & 1023) * 1024),
                                         (__attribute__((address_space(3))) void*)(smem + ((cur + 2) % 3) * BUF + (i * 8 + wave) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (MODE & 1) {
        const int slot = fq + 4 * ks;
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int row = wn * (TN * 16) + jn * 16 + fr;
          wf[jn] = *(const bf16x8*)(Bt + row * 128 + ((slot ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * (TM * 16) + i * 16 + fr;
          xf[i] = *(const bf16x8*)(A + row * 128 + ((slot ^ (row & 7)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) acc[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jn], xf[i], acc[jn][i], 0, 0, 0);
    }
    if (MODE & 4) { if (MODE & 32) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else if (MODE & 64) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
    if (MODE & 2) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    cur = cur == 2 ? 0 : cur + 1;
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[(size_t)blockIdx.x * 512 + tid] = sum;
}


// Production-like structure (conv_gemm2.hip): F1 reads | MFMA F0 | vmcnt + barrier | F0' reads | DMA (waves 0-3) | MFMA F1 | DMA (waves 4-7)
template <int STAGGER, int DMA_FIRST>
__global__ __launch_bounds__(512) void k2(const unsigned char* src, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TN = 5, TM = 4, BM = 256, BN = 160, BUF = (BM + BN) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / 2, wn = wave % 2, fr = lane & 15, fq = lane >> 4;
  {
    unsigned st = 1234567u + tid * 7919u + blockIdx.x * 104729u;
    for (int i = tid; i < 3 * BUF / 4; i += 512) {
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
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc2 + toff + (size_t)(i * 64 + wave * 8) * 640),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + (i * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 4; i < 7; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (48u << 20) + (size_t)(((i - 4) * 64 + wave * 8 + (lane >> 3)) * 5760 + (s % 45) * 128 + (lane & 7) * 16)),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + (i * 8 + wave) * 1024), 16, 0, 0);
  };
  Fr F0, F1;
  dma(0, 1); dma(1, 2);
  loadf(F0, 0, 0);
  const bool first = STAGGER ? wave < 4 : (DMA_FIRST != 0);
  int cur = 0;
  for (int s = 0; s < steps; ++s) {
    const int nxt = cur == 2 ? 0 : cur + 1;
    loadf(F1, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(F0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    loadf(F0, nxt, 0);
    if (first) dma(s + 2, cur);
    __builtin_amdgcn_sched_barrier(0);
    mma(F1);
    __builtin_amdgcn_sched_barrier(0);
    if (!first) dma(s + 2, cur);
    cur = nxt;
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[(size_t)blockIdx.x * 512 + tid] = sum;
}

template <int STAGGER, int DMA_FIRST>
void run2(const unsigned char* src, float* out, int steps, const char* name) {
  const int lds = 3 * (256 + 160) * 128;
  hipFuncSetAttribute((const void*)k2<STAGGER, DMA_FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<STAGGER, DMA_FIRST>), dim3(256), dim3(512), lds, 0, src, out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k2<STAGGER, DMA_FIRST>), dim3(256), dim3(512), lds, 0, src, out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = 256.0 * 8 * steps * 40 * 16 * 16 * 32 * 2;
  printf("%-34s %8.1f us  %7.1f TF/s  %6.0f ns/K-step\n", name, ms * 1e3, flops / ms / 1e9, ms * 1e6 / steps);
}

template <int MODE>
void run(const unsigned char* src, float* out, int steps, const char* name) {
  const int lds = 3 * (256 + 160) * 128;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, src, out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, src, out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = 256.0 * 8 * steps * 40 * 16 * 16 * 32 * 2;
  printf("%-34s %8.1f us  %7.1f TF/s  %6.0f ns/K-step\n", name, ms * 1e3, flops / ms / 1e9, ms * 1e6 / steps);
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
  run<0>(src, out, steps, "mfma only");
  run<1>(src, out, steps, "mfma + ds_read");
  run<2>(src, out, steps, "mfma + barrier");
  run<3>(src, out, steps, "mfma + ds_read + barrier");
  run<4>(src, out, steps, "mfma + dma");
  run<5>(src, out, steps, "mfma + ds_read + dma");
  run<7>(src, out, steps, "mfma + ds_read + dma + barrier");
  run<8>(src, out, steps, "mfma only, random data");
  run<8 + 3>(src, out, steps, "mfma+ds_read+barrier, random");
  run<8 + 7>(src, out, steps, "all, random data");
  run<16 + 8 + 7>(src, out, steps, "all, random, conv gather pattern");
  run<16 + 8 + 4>(src, out, steps, "mfma + dma gather, random");
  run<32 + 16 + 8 + 7>(src, out, steps, "all, conv gather, weight pieces only (3)");
  run<64 + 16 + 8 + 7>(src, out, steps, "all, conv gather, A pieces only (4)");
  run2<1, 0>(src, out, steps, "swp structure, stagger");
  run2<0, 1>(src, out, steps, "swp structure, dma before mma");
  run2<0, 0>(src, out, steps, "swp structure, dma after mma");
  return 0;
}
