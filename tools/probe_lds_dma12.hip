#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* lds_ptr;
__device__ void buf_load_lds(i32x4 rsrc, lds_ptr lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  i32x4 v = __builtin_bit_cast(i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]); v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]); v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}
__global__ void k(const uint32_t* src, uint32_t* out) {
  __shared__ uint32_t s[512];
  for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeef;
  __syncthreads();
  i32x4 r = make_rsrc(src, 4096);
  buf_load_lds(r, (lds_ptr)s, 12, threadIdx.x * 12, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
  uint32_t h[1024], *d, *o, ho[512];
  for (int i = 0; i < 1024; ++i) h[i] = i;
  hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; ++i) printf("%x%c", ho[i], (i % 16 == 15) ? '\n' : ' ');
  return 0;
}
