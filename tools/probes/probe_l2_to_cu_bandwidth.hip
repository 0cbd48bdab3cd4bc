// Probe (gfx950): how many bytes per clock does ONE CU get out of its XCD's L2 (and out of the memory-side cache), through
//   (a) global_load_dwordx4 into registers,
//   (b) buffer_load_dwordx4 ... lds (direct-to-LDS, what every ring / halo kernel here stages with),
//   (c) buffer_load_dword ... lds,
// with every CU busy?  The weight-gradient GEMMs re-read their operands from L2 once per workgroup tile; DESIGN.md 3.2k derives from
// the per-layer times that they stop at ~24 B/clk/CU of staging traffic whatever the tile — this measures that ceiling directly.
// Each workgroup streams a region that its XCD keeps resident (region = blockIdx % 8, 2 MiB each: 8 XCDs x 4 MiB L2), or a 96 MiB
// buffer that only the 256 MiB memory-side cache holds.  Wave-instruction footprint: 1 KiB contiguous (SEG = 1024) or 64-byte
// pieces of rows 256 bytes apart (SEG = 64: the gather of a 32-channel layer or of a 32-wide K stage); "halves paired": two
// consecutive loads of a wave fetch bytes 0-63 and 64-127 of the SAME 128-byte lines (does the vector cache keep the line?).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_l2_to_cu_bandwidth.hip -o /tmp/p && /tmp/p [MHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* lds_u32_ptr;
__device__ void raw_buffer_load_lds(i32x4 rsrc, lds_u32_ptr lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  i32x4 v = __builtin_bit_cast(i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]); v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]); v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

// MODE 0: global_load_dwordx4 -> VGPR; 1: dwordx4 -> LDS; 2: dword -> LDS.  INFL loads in flight per wave.
template <int MODE, int SEG, int INFL, int PAIR = 0> __global__ __launch_bounds__(512) void stream(const char* buf, uint32_t region_bytes, int regions, int iters, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const char* reg = buf + (size_t)(blockIdx.x % regions) * region_bytes;
  const i32x4 rs = make_rsrc(reg, region_bytes);
  constexpr int W = MODE == 2 ? 4 : 16;                 // bytes per lane
  // byte offset of this lane inside a wave-instruction footprint
  int loff;
  if (SEG == 1024) loff = lane * W;
  else loff = (lane * W / SEG) * 256 + (lane * W % SEG);          // SEG-byte pieces, 256 bytes apart
  constexpr int FOOT = SEG == 1024 ? 64 * W : (64 * W / SEG) * 256;   // address span of one wave instruction
  // a workgroup walks its region; wave w of workgroup b starts at a different place so that the CUs of an XCD do not hit one channel
  uint32_t pos = ((blockIdx.x / regions) * nw + wave) * FOOT * 7u % region_bytes;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
      u32x4 v[INFL];
#pragma unroll
      for (int k = 0; k < INFL; ++k) {
        uint32_t o = pos + k * FOOT * nw;
        o = o >= region_bytes ? o - region_bytes : o;
        v[k] = *reinterpret_cast<const u32x4*>(reg + o + loff);
      }
#pragma unroll
      for (int k = 0; k < INFL; ++k) acc ^= v[k];
    } else {
#pragma unroll
      for (int k = 0; k < INFL; ++k) {
        uint32_t o = PAIR ? pos + (k >> 1) * FOOT * nw + (k & 1) * 64 : pos + k * FOOT * nw;     // PAIR: loads 2j, 2j+1 = the two halves of the same lines
        o = o >= region_bytes ? o - region_bytes : o;
        raw_buffer_load_lds(rs, (lds_u32_ptr)(smem + wave * (INFL * 64 * W) + k * 64 * W), W, (int)(o + loff), 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    pos += (PAIR ? INFL / 2 : INFL) * FOOT * nw;
    pos = pos >= region_bytes ? pos - region_bytes : pos;
  }
  if constexpr (MODE != 0) { __syncthreads(); acc[0] = *reinterpret_cast<uint32_t*>(smem + tid * 4); }
  out[blockIdx.x * blockDim.x + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

// Ring-shaped variant (what conv_igemm_ring / conv_wgrad_ring do): 512 threads, a ring of D slots of STAGE bytes in LDS; per step every
// wave waits until all but its loads of the last D-2 steps have landed, the workgroup meets at a barrier (BAR = 1), and every wave
// requests its share of the stage D-1 steps ahead (STAGE / 8 waves / 1 KiB direct-to-LDS loads).  Contiguous 1 KiB footprints.
template <int STAGE, int D, int BAR> __global__ __launch_bounds__(512) void ring(const char* buf, uint32_t region_bytes, int regions, int steps, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) char smem[D * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int CNT = STAGE / 8 / 1024;                      // loads per wave and stage
  const char* reg = buf + (size_t)(blockIdx.x % regions) * region_bytes;
  const i32x4 rs = make_rsrc(reg, region_bytes);
  uint32_t pos = ((blockIdx.x / regions) * 8 + wave) * 1024u * 7u % region_bytes;
  auto request = [&](int slot) {
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
      uint32_t o = pos + k * 8192;
      o = o >= region_bytes ? o - region_bytes : o;
      raw_buffer_load_lds(rs, (lds_u32_ptr)(smem + slot * STAGE + (wave * CNT + k) * 1024), 16, (int)(o + lane * 16), 0, 0, 0);
    }
    pos += CNT * 8192;
    pos = pos >= region_bytes ? pos - region_bytes : pos;
  };
#pragma unroll
  for (int t = 0; t < D - 1; ++t) request(t);
  int sl = D - 1;
  for (int s = 0; s < steps; ++s) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * CNT) : "memory");
    if (BAR) __builtin_amdgcn_s_barrier();
    request(sl);
    sl = sl + 1 == D ? 0 : sl + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * blockDim.x + tid] = *reinterpret_cast<uint32_t*>(smem + tid * 4);
}

template <int STAGE, int D, int BAR> static void run_ring(const char* name, const char* buf, uint32_t region, int regions, uint32_t* out, double mhz) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int steps = 2000;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((ring<STAGE, D, BAR>), dim3(256), dim3(512), 0, 0, buf, region, regions, steps, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  const double bytes = (double)256 * STAGE * steps;
  printf("%-44s stage %2d KiB x %d slots, barrier %d: %7.2f TB/s  %6.1f B/clk/CU  (%.0f clk per step)\n", name, STAGE / 1024, D, BAR, bytes / best * 1e-9,
         bytes / 256 / (best * 1e-3 * mhz * 1e6), best * 1e-3 * mhz * 1e6 / steps);
}

template <int MODE, int SEG, int INFL, int PAIR = 0> static void run(const char* name, const char* buf, uint32_t region, int regions, int threads, int wg_per_cu, uint32_t* out, double mhz) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 2000;
  const int W = MODE == 2 ? 4 : 16;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((stream<MODE, SEG, INFL, PAIR>), dim3(256 * wg_per_cu), dim3(threads), 0, 0, buf, region, regions, iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
  }
  const double bytes = (double)256 * wg_per_cu * threads * W * INFL * iters;
  printf("%-44s %4d thr x %d/CU, %d in flight: %7.2f TB/s  %6.1f B/clk/CU\n", name, threads, wg_per_cu, INFL, bytes / best * 1e-9, bytes / 256 / (best * 1e-3 * mhz * 1e6));
}

int main(int argc, char** argv) {
  const double mhz = argc > 1 ? atof(argv[1]) : 2400.0;
  char* buf; (void)hipMalloc(&buf, (96u << 20) + (1u << 20)); (void)hipMemset(buf, 1, (96u << 20) + (1u << 20));   // (+ slack: a footprint may start just below a region's end)
  uint32_t* out; (void)hipMalloc(&out, 4u << 20);
  printf("L2-resident (8 regions of 2 MiB, region = workgroup %% 8):\n");
  run<0, 1024, 4>("global_load_dwordx4 -> VGPR, 1 KiB rows", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<0, 1024, 8>("global_load_dwordx4 -> VGPR, 1 KiB rows", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<0, 1024, 8>("global_load_dwordx4 -> VGPR, 1 KiB rows", buf, 2u << 20, 8, 256, 2, out, mhz);
  run<0, 1024, 8>("global_load_dwordx4 -> VGPR, 1 KiB rows", buf, 2u << 20, 8, 256, 4, out, mhz);
  run<0, 64, 8>("global_load_dwordx4 -> VGPR, 64 B pieces", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<1, 1024, 4>("buffer_load_dwordx4 -> LDS, 1 KiB rows", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<1, 1024, 8>("buffer_load_dwordx4 -> LDS, 1 KiB rows", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<1, 1024, 8>("buffer_load_dwordx4 -> LDS, 1 KiB rows", buf, 2u << 20, 8, 256, 2, out, mhz);
  run<1, 64, 8>("buffer_load_dwordx4 -> LDS, 64 B pieces", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<1, 64, 8, 1>("dwordx4 -> LDS, 64 B pieces, halves paired", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<1, 64, 4, 1>("dwordx4 -> LDS, 64 B pieces, halves paired", buf, 2u << 20, 8, 512, 1, out, mhz);
  run<2, 1024, 8>("buffer_load_dword -> LDS, 256 B rows", buf, 2u << 20, 8, 512, 1, out, mhz);
  printf("ring-shaped staging, L2-resident (the structure of conv_igemm_ring / conv_wgrad_ring, loads only):\n");
  run_ring<32768, 4, 1>("ring, 1 KiB rows", buf, 2u << 20, 8, out, mhz);
  run_ring<32768, 4, 0>("ring, 1 KiB rows", buf, 2u << 20, 8, out, mhz);
  run_ring<24576, 5, 1>("ring, 1 KiB rows", buf, 2u << 20, 8, out, mhz);
  run_ring<24576, 6, 1>("ring, 1 KiB rows", buf, 2u << 20, 8, out, mhz);
  run_ring<16384, 8, 1>("ring, 1 KiB rows", buf, 2u << 20, 8, out, mhz);
  run_ring<8192, 16, 1>("ring, 1 KiB rows", buf, 2u << 20, 8, out, mhz);
  run_ring<32768, 4, 1>("ring, 1 KiB rows, one 24 MiB region", buf, 24u << 20, 1, out, mhz);
  printf("memory-side cache (one 96 MiB region):\n");
  run<0, 1024, 8>("global_load_dwordx4 -> VGPR, 1 KiB rows", buf, 96u << 20, 1, 512, 1, out, mhz);
  run<1, 1024, 8>("buffer_load_dwordx4 -> LDS, 1 KiB rows", buf, 96u << 20, 1, 512, 1, out, mhz);
  return 0;
}
