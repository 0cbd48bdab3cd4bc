// Probe (gfx950): do ds_read_b128 / ds_read_b64 / ds_write_b64 work at 4-byte aligned LDS addresses, and at what rate?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_unaligned.hip -o /tmp/probe_lds && /tmp/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void check(uint32_t* out, int pitch) {
  __shared__ __attribute__((aligned(16))) uint32_t s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0x1000000u + i;
  __syncthreads();
  const int lane = threadIdx.x;
  const char* base = reinterpret_cast<const char*>(s);
  const int off = lane * pitch;                      // pitch = 20: 4-byte aligned, not 8 / 16
  u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
  u32x2 w = *reinterpret_cast<const u32x2*>(base + off + 4);
  out[lane * 8 + 0] = v[0]; out[lane * 8 + 1] = v[1]; out[lane * 8 + 2] = v[2]; out[lane * 8 + 3] = v[3];
  out[lane * 8 + 4] = w[0]; out[lane * 8 + 5] = w[1];
  __syncthreads();
  // unaligned 8-byte write, read back with dword reads
  *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(s) + 4096 + lane * pitch) = u32x2{0xAA000000u + lane, 0xBB000000u + lane};
  __syncthreads();
  out[lane * 8 + 6] = s[(4096 + lane * pitch) / 4];
  out[lane * 8 + 7] = s[(4096 + lane * pitch) / 4 + 1];
}

template <int MODE> __global__ void rate(uint32_t* out, int iters, int pitch) {
  __shared__ __attribute__((aligned(16))) uint32_t s[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // the x-column kernel's operand read: 16 voxels (2 rows x 8 along z) x 4 K-quarters of 16 bytes
  const int vox = (lane & 7) + (lane & 8 ? 10 : 0);
  int off = vox * pitch + (lane >> 4) * 16 + (threadIdx.x >> 6) * 4096;
  if (MODE == 0) off &= ~15;
  const uint32_t addr = (uint32_t)(uintptr_t)reinterpret_cast<char*>(s) + off;      // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    u32x4 v0, v1, v2, v3;
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:400\n ds_read_b128 %2, %4 offset:800\n ds_read_b128 %3, %4 offset:1200\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
    acc += v0 + v1 + v2 + v3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  uint32_t* d; hipMalloc(&d, 1 << 22);
  std::vector<uint32_t> h(64 * 8);
  for (int pitch : {16, 20, 24, 12}) {
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d, pitch);
    if (hipDeviceSynchronize() != hipSuccess) { printf("pitch %d: FAULT\n", pitch); return 1; }
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const uint32_t e0 = 0x1000000u + l * pitch / 4;
      for (int q = 0; q < 4; ++q) bad += h[l * 8 + q] != e0 + q;
      bad += h[l * 8 + 4] != e0 + 1; bad += h[l * 8 + 5] != e0 + 2;
      bad += h[l * 8 + 6] != 0xAA000000u + l; bad += h[l * 8 + 7] != 0xBB000000u + l;
    }
    printf("pitch %2d bytes: %s (%d mismatches)  lane1 read %08x %08x %08x %08x\n", pitch, bad ? "WRONG" : "correct", bad, h[8], h[9], h[10], h[11]);
  }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
    for (int pitch : {16, 20, 24, 32}) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, d, 2000, pitch);
        else hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, d, 2000, pitch);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      const double bytes = 1024.0 * 256 * 2000 * 4 * 16;
      printf("mode %s pitch %2d: %.3f ms  %.1f TB/s of LDS reads\n", mode ? "unaligned" : "aligned  ", pitch, best, bytes / best / 1e9);
    }
  return 0;
}
