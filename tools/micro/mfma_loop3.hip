// Micro-benchmark: the two-workgroup (4 waves, 128x160) conv K loop with
//   (a) two 64-wide LDS stages, DMA one K-step ahead (production: conv_gemm2.hip NS = 2)
//   (b) four 32-wide half-stages in the same LDS, one barrier per half-step, DMA three half-steps ahead
// 512 workgroups of 256 threads (two per CU), conv gather pattern on random data out of L2 / MALL.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_loop3 mfma_loop3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int TM = 4, TN = 5, BM = 128, BN = 160;

__device__ __forceinline__ void fill(unsigned char* smem, int bytes, int tid) {
  unsigned st = 1234567u + tid * 7919u + blockIdx.x * 104729u;
  for (int i = tid; i < bytes / 4; i += 256) {
    st = st * 1664525u + 1013904223u;
    const unsigned a = 0x3f800000u | (st & 0x807fffffu), b = (st * 2654435761u);
    ((unsigned*)smem)[i] = (a >> 16) | ((0x3f80u | (b & 0x807f)) << 16);
  }
}

// (a) production structure
__global__ __launch_bounds__(256, 2) void ka(const unsigned char* src, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BUF = (BM + BN) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / 2, wn = wave % 2, fr = lane & 15, fq = lane >> 4;
  fill(smem, 2 * BUF, tid);
  __syncthreads();
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* gA = src + (size_t)(blockIdx.x & 255) * (128 * 640) + (size_t)(lane >> 3) * 640 + (lane & 7) * 16 + 65 * 640;
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
    for (int i = 0; i < 4; ++i)     // A: 128 rows = 16 pieces of 8 rows
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + toff + (size_t)(i * 32 + wave * 8) * 640),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 5; ++i)     // W: 160 rows = 20 pieces
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (48u << 20) + (size_t)((i * 32 + wave * 8 + (lane >> 3)) * 5760 + (s % 45) * 128 + (lane & 7) * 16)),
                                       (__attribute__((address_space(3))) void*)(smem + buf * BUF + BM * 128 + (i * 4 + wave) * 1024), 16, 0, 0);
  };
  Fr F0, F1;
  dma(0, 0); dma(1, 1);
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  __syncthreads();
  loadf(F0, 0, 0);
  int cur = 0;
  for (int s = 0; s < steps; ++s) {
    const int nxt = cur ^ 1;
    loadf(F1, cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(F0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    loadf(F0, nxt, 0);
    dma(s + 2, cur);
    __builtin_amdgcn_sched_barrier(0);
    mma(F1);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
}

// (b) half-stages: LDS rows of 64 B, 16-byte slot swizzled by (row >> 2) & 3; a 1 KB DMA piece = 16 rows x 64 B
template <int DEPTH>   // half-stages in the ring (4 = same LDS as (a))
__global__ __launch_bounds__(256, 2) void kb(const unsigned char* src, float* out, int steps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HB = (BM + BN) * 64;
  constexpr int PIECES = 2 + 3;     // per wave: A 128 rows = 8 pieces (2 each), W 160 rows = 10 pieces (3 each: 2 over-fetched)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / 2, wn = wave % 2, fr = lane & 15, fq = lane >> 4;
  fill(smem, DEPTH * HB + 2048, tid);
  __syncthreads();
  f32x4 acc[TN][TM];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int prow = lane >> 2, pslot = (lane & 3) ^ ((prow >> 2) & 3);       // source-side swizzle
  const unsigned char* gA = src + (size_t)(blockIdx.x & 255) * (128 * 640) + (size_t)prow * 640 + pslot * 16 + 65 * 640;
  struct Fr { bf16x8 wf[TN]; bf16x8 xf[TM]; };
  auto loadf = [&](Fr& F, int buf) {
    const unsigned char* A = smem + buf * HB;
    const unsigned char* Bt = A + BM * 64;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) { const int row = wn * (TN * 16) + jn * 16 + fr; F.wf[jn] = *(const bf16x8*)(Bt + row * 64 + ((fq ^ ((row >> 2) & 3)) << 4)); }
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * (TM * 16) + i * 16 + fr; F.xf[i] = *(const bf16x8*)(A + row * 64 + ((fq ^ ((row >> 2) & 3)) << 4)); }
  };
  auto mma = [&](const Fr& F) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) acc[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.wf[jn], F.xf[i], acc[jn][i], 0, 0, 0);
  };
  auto dma = [&](int h, int buf) {     // half-step h = (K-step h / 2, half h & 1)
    const int s = h >> 1, tap = s % 9, chunk = (s / 9) % 5;
    const long toff = ((tap / 3 - 1) * 64 + (tap % 3 - 1)) * 640 + chunk * 128 + (h & 1) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + toff + (size_t)(i * 64 + wave * 16) * 640),
                                       (__attribute__((address_space(3))) void*)(smem + buf * HB + (i * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = i * 4 + wave;          // 12 pieces issued, 10 real: the last two go to a dump area behind the ring
      unsigned char* dst = piece < 10 ? smem + buf * HB + BM * 64 + piece * 1024 : smem + DEPTH * HB + (piece - 10) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (48u << 20) + (size_t)(((piece % 10) * 16 + prow) * 5760 + (s % 45) * 128 + (h & 1) * 64 + pslot * 16)),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  Fr F0, F1;
#pragma unroll
  for (int h = 0; h < DEPTH; ++h) dma(h, h);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * PIECES) : "memory");
  __syncthreads();
  loadf(F0, 0);
  int cur = 0;       // buffer of half-step j
  auto half = [&](Fr& Fc, Fr& Fn, int j) {
    // half j+1 must have landed (halves j+2 .. j+DEPTH-1 stay in flight); after the barrier every wave has issued its reads of half j,
    // so its buffer (cur) takes half j+DEPTH
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 2) * PIECES) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nxt = cur == DEPTH - 1 ? 0 : cur + 1;
    loadf(Fn, nxt);
    dma(j + DEPTH, cur);
    __builtin_amdgcn_sched_barrier(0);
    mma(Fc);
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  };
  for (int s = 0; s < steps; ++s) {
    half(F0, F1, 2 * s);
    half(F1, F0, 2 * s + 1);
  }
  float sum = 0.f;
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <typename K>
void run(K f, int lds, const unsigned char* src, float* out, int steps, const char* name) {
  hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(f, dim3(512), dim3(256), lds, 0, src, out, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(f, dim3(512), dim3(256), lds, 0, src, out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = 512.0 * steps * 2.0 * BM * BN * 64;
  printf("%-56s %8.1f us  %7.1f TF/s  %6.0f ns per K-step per workgroup  (%s)\n", name, ms * 1e3, flops / ms / 1e9, ms * 1e6 / steps,
         hipGetErrorString(hipGetLastError()));
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
  hipMalloc(&out, 512 * 256 * 4);
  const int steps = 2000;
  run(ka, 2 * (BM + BN) * 128, src, out, steps, "(a) 2 stages x 64, DMA one step ahead");
  run(kb<4>, 4 * (BM + BN) * 64 + 2048, src, out, steps, "(b) 4 half-stages x 32, DMA three half-steps ahead");
  run(kb<3>, 3 * (BM + BN) * 64 + 2048, src, out, steps, "(b') 3 half-stages x 32, DMA two half-steps ahead");
  run(kb<2>, 2 * (BM + BN) * 64 + 2048, src, out, steps, "(b'') 2 half-stages x 32, DMA one half-step ahead");
  return 0;
}
