// LDS-halo weight gradient of the stride-2 transposed convolution of the head (bf16, gfx950): ConvTranspose3d 64 -> <= 16, k3 s2.
//
//   R[tap*16 + a][b] = sum_{coarse voxel i} dOut[2 i + d(tap)][a] * x[i][b]        a < 16 fine channels, b < 64 coarse channels
//
// (the weight-gradient GEMM of ctseg_conv_wgrad with sin = 2: the gathered operand is the FINE tensor dOut, the row operand the coarse
// input x).  The generic split-K kernel re-gathers dOut once per 128 K rows — 27 taps x 16 channels = 4 K blocks — through L2; this
// one stages, per tile of 4 x 4 x 8 coarse voxels, the 9 x 9 x 17 fine halo (44 KB) and the coarse tile (16 KB) ONCE with LDS-DMA
// (raw.buffer.load.lds, 1 KB per instruction, no registers) and keeps all 27 x 16 x 64 fp32 accumulators in the registers of its four
// waves across all of the workgroup's tiles.
//
// Stride 2 makes every tap read the fine tensor at ONE parity per axis, so
//  * z: the 17 fine z of a halo row are stored parity-split (9 even, then 8 odd): the 8 coarse z of a k-step read 8 CONSECUTIVE
//    32-byte entries whatever the tap — the conflict-free 256-byte pattern of conv_wgrad_halo.hip's transposed reads;
//  * x: a k-step is the 4 x 8 (y, z) voxels of one coarse x plane p, so the fine fragment of (p, tap x = 2) is the fragment of
//    (p + 1, tap x = 0): a wave reads the 9 fine-plane fragments of a (tap y, tap z) pair once and feeds 12 tap-plane products.
// Four waves = two halves of the 64 coarse channels x two groups of (tap y, tap z) pairs {0..4} / {5..8}; the coarse fragments of a
// wave's channel half are read once per tile.  388 ds_read_b64_tr_b16 and 432 MFMAs per 128-voxel tile.
// Coarse voxels outside the volume (partial tiles) are zero in LDS; fine voxels before the volume (the -1 taps of the first
// tile along an axis) are zero; fine voxels past a partial tile meet zero coarse operands (finite x 0).
// Row 27 * 16 of the result is the column sum of the row operand (ctseg_conv_wgrad's bias-gradient row): the same geometry serves a
// plain stride-2 Conv3d 16 -> 64.  One fp32 slab per workgroup -> ctseg_conv_wgrad_reduce.  Two workgroups (60 KB of LDS each) per CU alternate load and multiply.
#include "ctseg_dev.h"

namespace ctseg {

typedef int32_t wu_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* wu_lds_ptr;
__device__ void wu_buffer_load_lds(wu_i32x4 rsrc, wu_lds_ptr lds, int size, int voffset, int soffset, int offset,
                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ wu_i32x4 wu_make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  wu_i32x4 v = __builtin_bit_cast(wu_i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]);
  v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]);
  v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

struct WgradUpArgs {
  const char* fine;     // dOut [N][2X][2Y][2Z][16] bf16
  const char* coarse;   // x    [N][X][Y][Z][64]    bf16
  float* ws;
  int N, X, Y, Z;       // coarse extents
  int kpad_w, cn_pad;
  int tiles, tyn, tzn;  // tiles per sample, along y / z
  int fine_sample_bytes, coarse_sample_bytes;
};

constexpr int WU_TX = 4, WU_TY = 4, WU_TZ = 8;                       // coarse tile
constexpr int WU_HX = 2 * WU_TX + 1, WU_HY = 2 * WU_TY + 1, WU_HZ = 2 * WU_TZ + 1;   // 9 x 9 x 17 fine halo
constexpr int WU_FPIECES = WU_HX * WU_HY * WU_HZ * 2;                // half-row pieces of the fine halo (2754): 16 or 12 bytes each
constexpr int WU_FINSTR = (WU_FPIECES + 63) / 64;                    // 44 DMA instructions
constexpr int WU_FBYTES = WU_FINSTR * 1024;                          // (16-wide rows; 12-wide rows use 44 x 768 of it)
constexpr int WU_CINSTR = 16;                                        // 4 channel blocks x 4 coarse planes, 32 voxels x 32 B each
constexpr int WU_CBYTES = WU_CINSTR * 1024;
constexpr int WU_FJ = WU_FINSTR / 4, WU_CJ = WU_CINSTR / 4;          // per wave
static_assert(WU_FINSTR % 4 == 0, "every wave issues the same number of DMA instructions");

// GW = elements of a fine row in memory: 16, or 12 (24-byte rows of the <= 12-channel gradient).  A 12-byte LDS-DMA piece lands
// in a 16-byte LDS slot like a 16-byte one (lane l writes bytes [16 l, 16 l + 12): measured, tools/probe_lds_dma12.hip), so the LDS
// image keeps its 32-byte voxel slots with channels 0-5 | 4 bytes never written | channels 6-11 | 4 bytes never written: the
// transposed reads are unchanged and operand row a' of the MFMA holds channel a' (a' < 6) or a' - 2 (8 <= a' < 14); rows 6, 7, 14,
// 15 multiply the zeroed gaps and are not written to the slab.
template <int GW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_up_kernel(const WgradUpArgs P, int total_tiles) {
  constexpr int FS = 32, PB = GW == 16 ? 16 : 12, GB = GW * 2;     // LDS voxel slot, bytes per DMA piece, bytes of a row in memory
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef s16x4 __attribute__((address_space(3)))* lds_s16x4;
  __shared__ __attribute__((aligned(16))) char smem[WU_FBYTES + WU_CBYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int tq = r16 >> 2, tp = r16 & 3;
  const int lz = 4 * (q4 & 1) + tq, ly = q4 >> 1;          // this lane's coarse (y, z) inside the first half of a k-step; + 2 y for the second
  const int nh = wave & 1, grp = wave >> 1;                // coarse channel half, (tap y, tap z) group
  const int c0 = grp * 5, nco = grp == 0 ? 5 : 4;

  const int Yf = 2 * P.Y, Zf = 2 * P.Z;
  // fine DMA pieces of this wave: byte offset from the halo origin (low 4 bits: "first halo plane along x / y / z", "no such piece")
  int fpk[WU_FJ];
#pragma unroll
  for (int j = 0; j < WU_FJ; ++j) {
    const int idx = (wave + 4 * j) * 64 + lane, s = idx >> 1, half = idx & 1;
    const int hx = s / (WU_HY * WU_HZ), rem = s - hx * (WU_HY * WU_HZ), hy = rem / WU_HZ, e = rem - hy * WU_HZ;
    const int hz = e < 9 ? 2 * e : 2 * (e - 9) + 1;
    const int off = ((hx * Yf + hy) * Zf + hz) * GB + half * PB;
    fpk[j] = idx < WU_FPIECES ? (off * 4 + ((hx == 0 ? 1 : 0) | (hy == 0 ? 2 : 0) | (hz == 0 ? 4 : 0))) : 8;      // (12-byte offsets have no free low bits)
  }
  // coarse pieces: instruction i = (channel block nb = i >> 2, plane p = i & 3); lane -> (voxel (ly_c, lz_c), 16-byte half)
  const int cly = lane >> 4, clz = (lane >> 1) & 7;
  const int coff = (cly * P.Z + clz) * 128 + (lane & 1) * 16;
  const uint32_t chot = (1u << cly) | (1u << (4 + clz));
  const int fbias = (Yf * Zf + Zf + 1) * GB;

  auto range_mask = [](int hi, int nbits) -> uint32_t {     // bits 0..hi (clamped)
    hi = hi > nbits - 1 ? nbits - 1 : hi;
    return hi < 0 ? 0u : ((2u << hi) - 1u);
  };
  auto dma = [&](int t) {
    const int n = t / P.tiles;
    int r = t - n * P.tiles;
    const int tz = r % P.tzn; r /= P.tzn;
    const int ty = r % P.tyn, tx = r / P.tyn;
    const int x0 = tx * WU_TX, y0 = ty * WU_TY, z0 = tz * WU_TZ;
    const int fm = (x0 == 0 ? 1 : 0) | (y0 == 0 ? 2 : 0) | (z0 == 0 ? 4 : 0) | 8;
    const wu_i32x4 fr = wu_make_rsrc(P.fine + (int64_t)n * P.fine_sample_bytes - fbias, (uint32_t)(P.fine_sample_bytes + fbias));
    const int fso = ((2 * x0 * Yf + 2 * y0) * Zf + 2 * z0) * GB;
#pragma unroll
    for (int j = 0; j < WU_FJ; ++j) {
      const int vo = (fpk[j] & fm) == 0 ? (int)((unsigned)fpk[j] >> 4 << 2) : (int)0x80000000;
      wu_buffer_load_lds(fr, (wu_lds_ptr)(smem + (wave + 4 * j) * 1024), PB, vo, fso, 0, 0);
    }
    const wu_i32x4 cr = wu_make_rsrc(P.coarse + (int64_t)n * P.coarse_sample_bytes, (uint32_t)P.coarse_sample_bytes);
    const uint32_t cnot = ~(range_mask(P.Y - y0 - 1, 4) | (range_mask(P.Z - z0 - 1, 8) << 4));
    const int cvo = (chot & cnot) == 0u ? coff : (int)0x80000000;
    const int cso = ((x0 * P.Y + y0) * P.Z + z0) * 128;
#pragma unroll
    for (int j = 0; j < WU_CJ; ++j) {
      const int i = wave + 4 * j, nb = i >> 2, p = i & 3;
      const int vo = (x0 + p < P.X) ? cvo : (int)0x80000000;
      wu_buffer_load_lds(cr, (wu_lds_ptr)(smem + WU_FBYTES + i * 1024), 16, vo, cso + p * P.Y * P.Z * 128 + nb * 32, 0, 0);
    }
  };

  f32x4 acc[5][3][2];
#pragma unroll
  for (int ci = 0; ci < 5; ++ci)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[ci][tx][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tr_frag = [&](const char* p0, int hi_off) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + hi_off));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  s16x8 ones;
  {
    const short o = (r16 == 0) ? (short)0x3f80 : (short)0;
    ones = s16x8{o, o, o, o, o, o, o, o};
  }
  const char* abase = smem + ((2 * ly) * WU_HZ + lz) * FS + tp * 8;                      // fine halo (plane 0, row 2 ly, entry lz)
  const char* bbase = smem + WU_FBYTES + (2 * nh) * 4096 + (ly * 8 + lz) * 32 + tp * 8;   // coarse block 2 nh, plane 0
  auto compute = [&]() {
    bf16x8 bf[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) bf[p][nb] = tr_frag(bbase + nb * 4096 + p * 1024, 2 * 8 * 32);
#pragma unroll
    for (int ci = 0; ci < 5; ++ci) {
      if (ci < nco) {                                  // wave-uniform
        const int c = c0 + ci, ty = c / 3, tz = c - 3 * ty;
        const int ez = tz == 0 ? 0 : (tz == 1 ? 9 : 1);                                   // even[lz], odd[lz], even[lz + 1]
        const char* ab = abase + (ty * WU_HZ + ez) * FS;
#pragma unroll
        for (int f = 0; f < WU_HX; ++f) {
          const bf16x8 af = tr_frag(ab + f * (WU_HY * WU_HZ * FS), 4 * WU_HZ * FS);       // second half: coarse y + 2 = fine row + 4
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            if ((f - tx) % 2 != 0 || f - tx < 0 || f - tx > 6) continue;                  // fine plane f = 2 p + tx
            const int p = (f - tx) / 2;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[ci][tx][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[p][nb], acc[ci][tx][nb], 0, 0, 0);
          }
        }
      }
    }
    if (grp == 1) {        // all-ones pseudo tap (row 27 * 16 = sum over voxels of the row operand: the bias gradient of a plain stride-2 conv)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[4][0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), bf[p][nb], acc[4][0][nb], 0, 0, 0);
    }
  };

  int t = blockIdx.x, tstride = gridDim.x, tlast = total_tiles;
  if ((gridDim.x & 7) == 0) {          // each XCD walks one contiguous eighth of the tiles: neighbouring halos meet in its L2
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * chunk + (blockIdx.x >> 3);
    tstride = gridDim.x >> 3;
    tlast = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  }
  if constexpr (GW == 12) {       // the 4-byte gaps behind every 12-byte piece are never written again
    for (int i = tid; i < WU_FBYTES / 16; i += 256) *reinterpret_cast<uint32_t*>(smem + i * 16 + 12) = 0u;
    __syncthreads();
  }
  if (t < tlast) dma(t);
  for (; t < tlast; t += tstride) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    compute();
    __syncthreads();
    if (t + tstride < tlast) dma(t + tstride);
  }

  float* slab = P.ws + (int64_t)blockIdx.x * P.kpad_w * P.cn_pad;
#pragma unroll
  for (int ci = 0; ci < 5; ++ci) {
    if (ci >= nco) continue;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int tap = tx * 9 + c0 + ci;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ar = 4 * q4 + e;                                         // operand row -> gathered channel
          const int ch = GW == 16 ? ar : ((ar & 7) < 6 ? ar - (ar >> 3) * 2 : -1);
          if (ch >= 0) slab[(int64_t)(tap * 16 + ch) * P.cn_pad + (2 * nh + nb) * 16 + r16] = acc[ci][tx][nb][e];
        }
    }
  }
  if (grp == 1) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (27 * 16 + 4 * q4 + e < P.kpad_w) slab[(int64_t)(27 * 16 + 4 * q4 + e) * P.cn_pad + (2 * nh + nb) * 16 + r16] = acc[4][0][nb][e];
  }
}

bool wgrad_up_eligible(const ctseg_wgrad_desc* d) {
  if (getenv("CTSEG_NO_WGRAD_UP") != nullptr) return false;
  if (d->dtype != CTSEG_BF16 || d->sin != 2 || d->ntaps != 27 || d->Cg != 16 || (d->g_ld != 16 && d->g_ld != 12) || d->Cn != 64 || d->d_ld != 64) return false;
  if (d->Xi != 2 * d->Xr || d->Yi != 2 * d->Yr || d->Zi != 2 * d->Zr) return false;
  if (d->kpad_w < 27 * 16 || d->cn_pad < 64) return false;
  if (d->in != nullptr && ((uintptr_t)d->in % 16) != 0) return false;       // (12-wide rows: 4-byte aligned pieces)
  if (d->dy != nullptr && ((uintptr_t)d->dy % 16) != 0) return false;
  const int64_t fb = ((int64_t)d->Xi * d->Yi * d->Zi + (int64_t)d->Yi * d->Zi + d->Zi + 1) * d->g_ld * 2, cb = (int64_t)d->Xr * d->Yr * d->Zr * 128;
  if (fb >= (1ll << 29)) return false;      // piece offsets are kept shifted left by 2 beside their flags
  if (fb >= (1ll << 31) || cb >= (1ll << 31)) return false;
  for (int j = 0; j < 27; ++j) {       // taps are indexed tx * 9 + ty * 3 + tz
    const int tp = d->taps[j];
    if ((int)(int8_t)(tp & 0xff) != j / 9 - 1 || (int)(int8_t)((tp >> 8) & 0xff) != (j / 3) % 3 - 1 || (int)(int8_t)((tp >> 16) & 0xff) != j % 3 - 1)
      return false;
  }
  return true;
}

static int wgrad_up_grid(const ctseg_wgrad_desc* d) {
  const int tiles = ((d->Xr + WU_TX - 1) / WU_TX) * ((d->Yr + WU_TY - 1) / WU_TY) * ((d->Zr + WU_TZ - 1) / WU_TZ) * d->N;
  int per_cu = 2;
  if (const char* e = getenv("CTSEG_WU_PER_CU")) per_cu = atoi(e);
  return persistent_grid(CTSEG_NUM_CU * per_cu, tiles);
}

int wgrad_up_slabs(const ctseg_wgrad_desc* d) { return wgrad_up_grid(d); }

void launch_wgrad_up(const ctseg_wgrad_desc* d, hipStream_t st) {
  WgradUpArgs a;
  a.fine = (const char*)d->in; a.coarse = (const char*)d->dy; a.ws = d->ws;
  a.N = d->N; a.X = d->Xr; a.Y = d->Yr; a.Z = d->Zr;
  a.kpad_w = d->kpad_w; a.cn_pad = d->cn_pad;
  a.tyn = (d->Yr + WU_TY - 1) / WU_TY; a.tzn = (d->Zr + WU_TZ - 1) / WU_TZ;
  a.tiles = ((d->Xr + WU_TX - 1) / WU_TX) * a.tyn * a.tzn;
  a.fine_sample_bytes = (int)((int64_t)d->Xi * d->Yi * d->Zi * d->g_ld * 2);
  a.coarse_sample_bytes = (int)((int64_t)d->Xr * d->Yr * d->Zr * 128);
  if (d->g_ld == 12) hipLaunchKernelGGL(conv_wgrad_up_kernel<12>, dim3(wgrad_up_grid(d)), dim3(256), 0, st, a, a.tiles * d->N);
  else hipLaunchKernelGGL(conv_wgrad_up_kernel<16>, dim3(wgrad_up_grid(d)), dim3(256), 0, st, a, a.tiles * d->N);
}

}  // namespace ctseg
