// Probe (gfx950): what does the memory path deliver for the halo-staging access pattern of the 10-channel head kernels?
// Tensor [2][512][512][48][12] bf16 (24-byte voxel rows, 604 MB), persistent workgroups walking 4 x 8 x 8 tiles (x, y, z; z contiguous)
// in the order of conv_wgrad_head2 / conv_halo_x, each tile = the 6 x 10 x 10 halo (600 voxels, 14.4 KB).  Loads only: one register set
// per tile in flight (or two: DEPTH 2), consumed by an XOR.  Patterns:
//   0  8-byte pieces, 3 per voxel, lane = consecutive pieces along z (what ships)
//   1  16-byte pieces of whole z rows, rows widened to 12 voxels from z0 - 2 so that every piece is 16-byte aligned (18 per row)
//   2  12-byte pieces (dwordx3), 2 per voxel
//   3  8-byte pieces, no halo (the 4 x 8 x 8 tile only: 256 voxels) — the dY operand's pattern
// Prints ms and useful TB/s (halo bytes / time) per pattern, workgroups per CU and prefetch depth.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/probe_halo_load_pattern.hip -o tools/probes/probe_halo_load_pattern.bin && tools/probes/probe_halo_load_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int X = 512, Y = 512, Z = 48, N = 2, ROW = 24;
constexpr int TXN = X / 4, TYN = Y / 8, TZN = Z / 8, TILES = TXN * TYN * TZN;

template <int PAT, int NTHR> struct Cfg;
template <int NTHR> struct Cfg<0, NTHR> { static constexpr int PIECES = 1800, J = (PIECES + NTHR - 1) / NTHR; };
template <int NTHR> struct Cfg<1, NTHR> { static constexpr int PIECES = 60 * 18, J = (PIECES + NTHR - 1) / NTHR; };
template <int NTHR> struct Cfg<2, NTHR> { static constexpr int PIECES = 1200, J = (PIECES + NTHR - 1) / NTHR; };
template <int NTHR> struct Cfg<3, NTHR> { static constexpr int PIECES = 768, J = (PIECES + NTHR - 1) / NTHR; };

template <int PAT, int NTHR, int DEPTH>
__global__ __launch_bounds__(NTHR) void probe(const char* in, uint32_t* out, int total_tiles) {
  constexpr int J = Cfg<PAT, NTHR>::J, PIECES = Cfg<PAT, NTHR>::PIECES;
  const int tid = threadIdx.x;
  const int YZ = Y * Z;
  const int sample_bytes = X * Y * Z * ROW;
  const int bias = (YZ + Z + 2) * ROW;          // the halo origin of a tile at the origin lies (1,1,1) (+1 voxel for pattern 1) before the sample
  int off[J];
  bool live[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int idx = tid + j * NTHR;
    live[j] = idx < PIECES;
    if (PAT == 0) { const int hv = idx / 3, part = idx % 3, hx = hv / 100, hy = (hv / 10) % 10, hz = hv % 10; off[j] = (hx * YZ + hy * Z + hz + 1) * ROW + part * 8; }
    else if (PAT == 1) { const int row = idx / 18, k = idx % 18, hx = row / 10, hy = row % 10; off[j] = (hx * YZ + hy * Z) * ROW + k * 16; }
    else if (PAT == 2) { const int hv = idx / 2, part = idx % 2, hx = hv / 100, hy = (hv / 10) % 10, hz = hv % 10; off[j] = (hx * YZ + hy * Z + hz + 1) * ROW + part * 12; }
    else { const int tv = idx / 3, part = idx % 3, tx = tv >> 6, ty = (tv >> 3) & 7, tz = tv & 7; off[j] = ((tx + 1) * YZ + (ty + 1) * Z + tz + 2) * ROW + part * 8; }
  }
  int t = blockIdx.x, stride = gridDim.x, last = total_tiles;
  if ((gridDim.x & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * chunk + (blockIdx.x >> 3); stride = gridDim.x >> 3;
    last = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  }
  uint32_t acc = 0;
  u32x4 r[DEPTH][J];
  auto gload = [&](int tt, u32x4 (&reg)[J]) {
    const int n = tt / TILES; int q = tt - n * TILES;
    const int tz = q % TZN; q /= TZN; const int ty = q % TYN, tx = q / TYN;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(in) + (int64_t)n * sample_bytes - bias, 0, sample_bytes + bias, 0x00020000);
    const int soff = ((tx * 4 * Y + ty * 8) * Z + tz * 8) * ROW;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int vo = live[j] ? off[j] : (int)0x80000000;
      if (PAT == 1) reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0);
      else if (PAT == 2) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, vo, soff, 0); reg[j] = u32x4{v[0], v[1], v[2], 0u}; }
      else { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo, soff, 0); reg[j] = u32x4{v[0], v[1], 0u, 0u}; }
    }
  };
  auto consume = [&](const u32x4 (&reg)[J]) {
#pragma unroll
    for (int j = 0; j < J; ++j) acc ^= reg[j][0] ^ reg[j][1] ^ reg[j][2] ^ reg[j][3];
  };
  if (DEPTH == 1) {
    if (t < last) gload(t, r[0]);
    for (; t < last; t += stride) {
      u32x4 cur[J];
#pragma unroll
      for (int j = 0; j < J; ++j) cur[j] = r[0][j];
      consume(cur);
      if (t + stride < last) gload(t + stride, r[0]);
      __syncthreads();
    }
  } else {
    if (t < last) gload(t, r[0]);
    if (t + stride < last) gload(t + stride, r[1 % DEPTH]);
    int k = 0;
    for (; t < last; t += stride, k ^= 1) {
      if (k == 0) { consume(r[0]); if (t + 2 * stride < last) gload(t + 2 * stride, r[0]); }
      else { consume(r[1 % DEPTH]); if (t + 2 * stride < last) gload(t + 2 * stride, r[1 % DEPTH]); }
      __syncthreads();
    }
  }
  out[blockIdx.x * NTHR + tid] = acc;
}

template <int PAT, int NTHR, int DEPTH> static void run(const char* in, uint32_t* out, int wg_per_cu, const char* name) {
  const int total = TILES * N, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<PAT, NTHR, DEPTH>), dim3(grid), dim3(NTHR), 0, 0, in, out, total);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<PAT, NTHR, DEPTH>), dim3(grid), dim3(NTHR), 0, 0, in, out, total);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double useful = (double)total * (PAT == 3 ? 256 : 600) * ROW;
  printf("%-46s threads %4d  wg/CU %d  depth %d : %.4f ms  %.2f TB/s useful (%.0f MB staged)\n", name, NTHR, wg_per_cu, DEPTH, ms, useful / ms / 1e9, useful / 1e6);
}

int main() {
  // the probe does not mask halo voxels outside the volume (the kernels do): a guard in front of the tensor keeps the reads of the
  // first tiles (x = -1, y = -1, z = -2) inside the allocation; past the end the buffer range check delivers zeros
  const size_t guard = 2u << 20, bytes = (size_t)N * X * Y * Z * ROW + guard + (1 << 20);
  static_assert((size_t)(Y * Z + Z + 2) * ROW < (2u << 20), "guard covers the halo origin shift");
  char* alloc; uint32_t* out;
  if (hipMalloc(&alloc, bytes) != hipSuccess || hipMalloc(&out, 1 << 24) != hipSuccess) { printf("alloc failed\n"); return 1; }
  char* in = alloc + guard;
  std::vector<uint32_t> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u);
  for (size_t o = 0; o + (4u << 20) <= bytes; o += (4u << 20)) (void)hipMemcpy(alloc + o, h.data(), 4u << 20, hipMemcpyHostToDevice);
  (void)hipDeviceSynchronize();
  run<0, 512, 1>(in, out, 1, "8-byte pieces (shipped pattern)");
  run<0, 512, 2>(in, out, 1, "8-byte pieces (shipped pattern)");
  run<0, 256, 1>(in, out, 2, "8-byte pieces (shipped pattern)");
  run<0, 256, 2>(in, out, 2, "8-byte pieces (shipped pattern)");
  run<0, 256, 2>(in, out, 4, "8-byte pieces (shipped pattern)");
  run<1, 512, 1>(in, out, 1, "16-byte pieces of aligned 12-voxel z rows");
  run<1, 512, 2>(in, out, 1, "16-byte pieces of aligned 12-voxel z rows");
  run<1, 256, 1>(in, out, 2, "16-byte pieces of aligned 12-voxel z rows");
  run<1, 256, 2>(in, out, 2, "16-byte pieces of aligned 12-voxel z rows");
  run<1, 256, 2>(in, out, 4, "16-byte pieces of aligned 12-voxel z rows");
  run<2, 512, 1>(in, out, 1, "12-byte pieces");
  run<2, 512, 2>(in, out, 1, "12-byte pieces");
  run<2, 256, 2>(in, out, 2, "12-byte pieces");
  run<3, 512, 1>(in, out, 1, "8-byte pieces, tile only (no halo)");
  run<3, 512, 2>(in, out, 1, "8-byte pieces, tile only (no halo)");
  return 0;
}
