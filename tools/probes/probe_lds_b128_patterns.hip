// Probe (gfx950): ds_read_b128 throughput for candidate lane -> address patterns of a 16-column x 32-k weight fragment read
// (lane = q4 * 16 + r16; r16 = fragment row, q4 = 16-byte k quarter).  Which row pitch / swizzle is bank-conflict free?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_lds_b128_patterns.hip -o /tmp/probe_b128 && /tmp/probe_b128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <functional>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void rate(const int* offs, uint32_t* out, int iters) {
  __shared__ __attribute__((aligned(16))) uint32_t s[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t addr = (uint32_t)(uintptr_t)reinterpret_cast<char*>(s) + offs[lane] + (threadIdx.x >> 6) * 8192;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    u32x4 v0, v1, v2, v3;
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:4096\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4 offset:4096\n s_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "memory");
    acc += v0 + v1 + v2 + v3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  uint32_t* d; hipMalloc(&d, 1 << 22);
  int* doff; hipMalloc(&doff, 256);
  struct Pat { const char* name; std::function<int(int, int)> f; };
  auto sw01 = [](int r) { return (r & 12) | ((r & 1) << 1) | ((r >> 1) & 1); };
  std::vector<Pat> pats = {
      {"linear lane*16", [](int r, int q) { return (q * 16 + r) * 16; }},
      {"pitch128 q^(r>>1)&7   (conv_halo_sw)", [](int r, int q) { return r * 128 + ((q ^ ((r >> 1) & 7)) << 4); }},
      {"pitch128 no swizzle", [](int r, int q) { return r * 128 + (q << 4); }},
      {"pitch64 q^(r>>2)&3    (first up8)", [](int r, int q) { return r * 64 + ((q ^ ((r >> 2) & 3)) << 4); }},
      {"pitch64 swap01 rows, q^(r>>2)&3", [=](int r, int q) { return sw01(r) * 64 + ((q ^ ((r >> 2) & 3)) << 4); }},
      {"pitch64 q^(r>>1)&3", [](int r, int q) { return r * 64 + ((q ^ ((r >> 1) & 3)) << 4); }},
      {"pitch64 q^(r&3)", [](int r, int q) { return r * 64 + ((q ^ (r & 3)) << 4); }},
      {"pitch64 no swizzle", [](int r, int q) { return r * 64 + (q << 4); }},
      {"pitch64 q^((r>>1)&1 | (r>>2)&2)", [](int r, int q) { return r * 64 + ((q ^ (((r >> 1) & 1) | ((r >> 2) & 2))) << 4); }},
      {"pitch80 (64 + 16 pad) no swizzle", [](int r, int q) { return r * 80 + (q << 4); }},
      {"pitch64 rows (r&7)*2+(r>>3), q^(r>>1)&3", [](int r, int q) { return ((r & 7) * 2 + (r >> 3)) * 64 + ((q ^ ((r >> 1) & 3)) << 4); }},
      {"pitch64 q^(r>>1)&3 ^ (r>>3)", [](int r, int q) { return r * 64 + ((q ^ ((r >> 1) & 3) ^ ((r >> 3) & 1)) << 4); }},
      {"pitch64 q^((r>>1)&1)*2 ^ ((r>>2)&1)", [](int r, int q) { return r * 64 + ((q ^ (((r >> 1) & 1) * 2) ^ ((r >> 2) & 1)) << 4); }},
  };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (auto& p : pats) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = p.f(l & 15, l >> 4);
    hipMemcpy(doff, h, 256, hipMemcpyHostToDevice);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, doff, d, 2000);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    const double bytes = 1024.0 * 256 * 2000 * 4 * 16;
    printf("%-44s %.3f ms  %6.1f TB/s\n", p.name, best, bytes / best / 1e9);
  }
  return 0;
}
