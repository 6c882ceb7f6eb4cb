// Semantics probe: buffer_load_dwordx4 ... lds with an out-of-range voffset (does the LDS get zeros?) and with soffset.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ void k(const unsigned* src, unsigned* out, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)nbytes, 0x00020000);
  unsigned voff = lane * 16;
  if (lane & 1) voff = 0xfffffff0u;           // odd lanes: far out of range
  if (lane == 62) voff = nbytes - 8;          // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, /*soffset*/ 1024, /*imm*/ 0, /*aux*/ 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

// EXEC-masked lanes: do they leave their 16-byte LDS slot alone, and do the active lanes still write at base + lane * 16?
__global__ void kmask(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0xffffff00, 0x00020000);
  if (lane % 6 != 5)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, (unsigned)lane * 16u, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
  unsigned *src, *out;
  const unsigned n = 4096;   // dwords
  hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 4);
  unsigned h[4096];
  for (unsigned i = 0; i < n; ++i) h[i] = i;
  hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 2048u);   // num_records = 2048 bytes: soffset 1024 + voff
  unsigned o[256];
  hipMemcpy(o, out, 256 * 4, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 3, 60, 62, 63}) printf("lane %2d: %08x %08x %08x %08x\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  hipLaunchKernelGGL(kmask, dim3(1), dim3(64), 0, 0, src, out);
  hipMemcpy(o, out, 256 * 4, hipMemcpyDeviceToHost);
  printf("masked lanes (lane %% 6 == 5 inactive):\n");
  for (int l : {0, 4, 5, 6, 11, 12, 63}) printf("lane %2d: %08x %08x %08x %08x\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  return 0;
}
