// Probe (gfx950): which lanes of a wave share an LDS cycle?  Base pattern: lane l reads 16 (or 8) bytes at l * 16 (l * 8) — conflict
// free.  Variant k: lane k is moved onto lane 0's banks in another row (+4096 bytes).  If lanes 0 and k are served in the same cycle
// the instruction takes one more pass.  Prints the relative time per k for ds_read_b128, ds_read_b64 and ds_write_b64.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_lane_groups.hip -o /tmp/probe_lanes && /tmp/probe_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void rate(const int* offs, uint32_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t addr = (uint32_t)(uintptr_t)reinterpret_cast<char*>(s) + offs[lane] + (threadIdx.x >> 6) * 8192 * 0;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      u32x4 v0, v1, v2, v3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
      acc += v0 + v1 + v2 + v3;
    } else if (MODE == 1) {
      u32x2 v0, v1, v2, v3;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n s_waitcnt lgkmcnt(0)"
                   : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
      acc[0] += v0[0] + v1[1] + v2[0] + v3[1];
    } else {
      u32x2 w = {acc[0], (uint32_t)i};
      asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n ds_write_b64 %0, %1\n s_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(w) : "memory");
      acc[0] += i;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  uint32_t* d; (void)hipMalloc(&d, 1 << 22);
  int* doff; (void)hipMalloc(&doff, 256);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const char* names[3] = {"ds_read_b128", "ds_read_b64", "ds_write_b64"};
  for (int mode = 0; mode < 3; ++mode) {
    const int w = mode == 0 ? 16 : 8;
    float base = 0.f;
    printf("%s (lane l at l * %d; lane k moved onto lane 0's banks): time relative to the conflict-free pattern\n", names[mode], w);
    for (int k = 0; k < 64; ++k) {
      int h[64];
      for (int l = 0; l < 64; ++l) h[l] = l * w;
      if (k > 0) h[k] = 4096;          // same banks as lane 0, another row
      (void)hipMemcpy(doff, h, 256, hipMemcpyHostToDevice);
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(1024), dim3(256), 0, 0, doff, d, 1000);
        else if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(1024), dim3(256), 0, 0, doff, d, 1000);
        else hipLaunchKernelGGL(rate<2>, dim3(1024), dim3(256), 0, 0, doff, d, 1000);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      if (k == 0) base = best;
      printf("%s%4.2f", (k % 16 == 0) ? "\n  " : " ", best / base);
    }
    printf("\n");
  }
  return 0;
}
