// LDS-halo weight-gradient kernel (bf16) for 3x3x3 stride-1 layers with few channels (gfx950).
//
//   R[tap*Cg + a][b] = sum_voxels x[voxel + d(tap)][a] * dy[voxel][b],   R[27*Cg][b] = sum_voxels dy[voxel][b]
//
// The generic split-K kernel re-gathers x once per 128 K-rows (27x through L2 for a 3x3x3 layer).  Here a persistent
// workgroup stages the 6x10x10 halo of x and the 4x8x8 tile of dy ONCE per tile (double-buffered), and the four waves
// split the 27 taps (+ one all-ones pseudo tap = bias gradient): wave w owns taps w, w+4, ... and keeps their
// [Cg x Cn] fp32 accumulators in registers across ALL of the workgroup's tiles; dy fragments are read once per
// 32-voxel k-step and reused for the wave's 7 taps.  Both MFMA operands contract over voxels, the slow axis of the
// channels-last tensors: ds_read_b64_tr_b16 transposes 4 voxels x 16 channels per 16-lane group; the LDS images are
// planes of 16 channels (32 B per voxel) and a k-step is 4 z-rows of 8 voxels, so each 32-lane half reads 8
// contiguous 32-byte rows (256 B): bank-conflict free.  One fp32 slab per workgroup -> ctseg_conv_wgrad_reduce.
#include "ctseg_dev.h"

namespace ctseg {

struct WgradHaloArgs {
  const char* in;
  const char* dy;
  float* ws;
  int N, X, Y, Z, Cg, g_ld, d_ld;
  int kpad_w, cn_pad;
  int tiles, txn, tyn, tzn;   // tiles per sample and per axis
  int delta[28];              // halo voxel delta of each tap (in 32-byte rows), entry 27 unused
};

constexpr int WH_HV = 600, WH_TV = 256;

template <int VB, int DBY>
__global__ __launch_bounds__(256) void conv_wgrad_halo_kernel(const WgradHaloArgs P, int total_tiles) {
  constexpr int PA = VB / 32, PB = DBY / 32;
  constexpr int XBYTES = PA * WH_HV * 32, DBYTES = PB * WH_TV * 32, BUF = XBYTES + DBYTES;
  constexpr int XCH = WH_HV * (VB / 16), DCH = WH_TV * (DBY / 16);
  constexpr int JX = (XCH + 255) / 256, JD = (DCH + 255) / 256;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef s16x4 __attribute__((address_space(3)))* lds_s16x4;

  // ONE staging buffer: the next tile's operands wait in registers (gload) while this tile is consumed and are written after
  // the barrier that ends it.  Half the LDS of a double buffer -> twice the workgroups per CU, whose phases interleave.
  __shared__ __attribute__((aligned(16))) char smem[BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tap selection stays on the scalar unit
  const int r16 = lane & 15, q4 = lane >> 4;
  const int tq = r16 >> 2, tp = r16 & 3;

  // per-thread constants of the staging slots (only a scalar tile base changes from tile to tile)
  const int YZ = P.Y * P.Z;
  int gx_byte[JX], gx_hxyz[JX], gx_lds[JX], gd_byte[JD], gd_xyz[JD], gd_lds[JD];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int idx = tid + j * 256;
    const int c = idx % (VB / 16), hv = idx / (VB / 16);
    const int hx = hv / 100, rem = hv - hx * 100, hy = rem / 10, hz = rem - hy * 10;
    gx_byte[j] = (((hx - 1) * YZ + (hy - 1) * P.Z + (hz - 1)) * P.g_ld + c * 8) * 2;
    gx_hxyz[j] = (idx < XCH) ? (hx | (hy << 8) | (hz << 16)) : 0x7f7f7f;
    gx_lds[j] = (c >> 1) * (WH_HV * 32) + hv * 32 + (c & 1) * 16;
  }
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int idx = tid + j * 256;
    const int c = idx % (DBY / 16), tv = idx / (DBY / 16);
    const int tx = tv >> 6, ty = (tv >> 3) & 7, tz = tv & 7;
    gd_byte[j] = ((tx * YZ + ty * P.Z + tz) * P.d_ld + c * 8) * 2;
    gd_xyz[j] = (idx < DCH) ? (tx | (ty << 8) | (tz << 16)) : 0x7f7f7f;
    gd_lds[j] = (c >> 1) * (WH_TV * 32) + tv * 32 + (c & 1) * 16;
  }
  u32x4 rx[JX], rd[JD];
  // 12-wide bf16 rows (24 bytes): chunk 1 is channels 8..11 + zero fill (32-byte planes only)
  const bool x12 = VB == 32 && (P.g_ld & 7) != 0, d12 = DBY == 32 && (P.d_ld & 7) != 0;
  auto load_row_chunk = [](const char* p, bool w12, bool second) -> u32x4 {
    if (!w12) return *reinterpret_cast<const u32x4*>(p);
    return load_row12_chunk(p, second);
  };
  auto origin = [&](int t, int& n, int& x0, int& y0, int& z0) {
    n = t / P.tiles;
    int r = t - n * P.tiles;
    const int tz = r % P.tzn; r /= P.tzn;
    const int ty = r % P.tyn; const int tx = r / P.tyn;
    x0 = tx * 4; y0 = ty * 8; z0 = tz * 8;
  };
  auto gload = [&](int t) {
    int n, x0, y0, z0;
    origin(t, n, x0, y0, z0);
    const int64_t vb = (((int64_t)n * P.X + x0) * P.Y + y0) * P.Z + z0;
    const char* xb = P.in + vb * P.g_ld * 2;
    const char* db = P.dy + vb * P.d_ld * 2;
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const int xi = x0 - 1 + (gx_hxyz[j] & 0xff), yi = y0 - 1 + ((gx_hxyz[j] >> 8) & 0xff), zi = z0 - 1 + (gx_hxyz[j] >> 16);
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((unsigned)xi < (unsigned)P.X && (unsigned)yi < (unsigned)P.Y && (unsigned)zi < (unsigned)P.Z)
        v = load_row_chunk(xb + gx_byte[j], x12, (gx_lds[j] & 16) != 0);
      rx[j] = v;
    }
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      const int xi = x0 + (gd_xyz[j] & 0xff), yi = y0 + ((gd_xyz[j] >> 8) & 0xff), zi = z0 + (gd_xyz[j] >> 16);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xi < P.X && yi < P.Y && zi < P.Z) v = load_row_chunk(db + gd_byte[j], d12, (gd_lds[j] & 16) != 0);
      rd[j] = v;
    }
  };
  auto sstore = [&](int) {
    char* xs = smem;
    char* ds = xs + XBYTES;
#pragma unroll
    for (int j = 0; j < JX; ++j)
      if (JX * 256 == XCH || tid + j * 256 < XCH) *reinterpret_cast<u32x4*>(xs + gx_lds[j]) = rx[j];
#pragma unroll
    for (int j = 0; j < JD; ++j)
      if (JD * 256 == DCH || tid + j * 256 < DCH) *reinterpret_cast<u32x4*>(ds + gd_lds[j]) = rd[j];
  };

  f32x4 acc[7][PA][PB];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int a = 0; a < PA; ++a)
#pragma unroll
      for (int b = 0; b < PB; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's voxel inside a k-step for the two transposed reads: y-row 2r + (q4>>1), z = 4*(q4&1) + tq
  const int lz = 4 * (q4 & 1) + tq, ly = q4 >> 1;
  s16x8 ones;
  {
    const short o = (r16 == 0) ? (short)0x3f80 : (short)0;
    ones = s16x8{o, o, o, o, o, o, o, o};
  }

  // tile sequence: each XCD (blockIdx % 8 shares one) owns a contiguous range of tiles and its workgroups walk it round-robin,
  // so the tiles in flight on one XCD are neighbours and share their halos through that XCD's L2
  int t = blockIdx.x, tstride = gridDim.x, tlast = total_tiles;
  if ((gridDim.x & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * chunk + (blockIdx.x >> 3);
    tstride = gridDim.x >> 3;
    tlast = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  }
  if (t < tlast) {
    gload(t);
    sstore(0);
  }
  __syncthreads();
  for (; t < tlast; t += tstride) {
    const int tn = t + tstride;
    if (tn < tlast) gload(tn);
    const char* xs = smem;
    const char* ds = xs + XBYTES;
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const int x = s >> 1, yb = 4 * (s & 1);
      // dy fragments of this k-step (shared by the wave's 7 taps)
      bf16x8 df[PB];
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        const char* p0 = ds + b * (WH_TV * 32) + (((x * 8) + (yb + ly)) * 8 + lz) * 32 + tp * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 8 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        df[b] = __builtin_bit_cast(bf16x8, v);
      }
      const int hbase = (((x + 1) * 10) + (yb + ly + 1)) * 10 + (lz + 1);
      // taps wave, wave+4, ..., wave+20 always exist: fetch all their operands first, then issue the MFMAs back to back
      // (branch-free, so the LDS latency of one tap hides behind the others); tap wave+24 may be the pseudo tap 27
      bf16x8 af[6][PA];
#pragma unroll
      for (int ti = 0; ti < 6; ++ti) {
        const int h = hbase + P.delta[wave + 4 * ti];
#pragma unroll
        for (int a = 0; a < PA; ++a) {
          const char* p0 = xs + a * (WH_HV * 32) + h * 32 + tp * 8;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 10 * 32));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[ti][a] = __builtin_bit_cast(bf16x8, v);
        }
      }
#pragma unroll
      for (int ti = 0; ti < 6; ++ti)
#pragma unroll
        for (int a = 0; a < PA; ++a)
#pragma unroll
          for (int b = 0; b < PB; ++b)
            acc[ti][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ti][a], df[b], acc[ti][a][b], 0, 0, 0);
      {
        const int tap = wave + 24;
        if (tap < 27) {
          const int h = hbase + P.delta[tap];
#pragma unroll
          for (int a = 0; a < PA; ++a) {
            const char* p0 = xs + a * (WH_HV * 32) + h * 32 + tp * 8;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 10 * 32));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const bf16x8 a6 = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[6][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6, df[b], acc[6][a][b], 0, 0, 0);
          }
        } else {  // pseudo tap 27: x == 1 on channel row 0 -> row 0 of the tile accumulates sum(dy) (bias gradient)
#pragma unroll
          for (int b = 0; b < PB; ++b)
            acc[6][0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), df[b], acc[6][0][b], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (tn < tlast) sstore(0);
    __syncthreads();
  }

  // ---- one slab per workgroup: lane holds column c16 = dy channel, rows 4*q4 + e = x channel ------------------------
  float* slab = P.ws + (int64_t)blockIdx.x * P.kpad_w * P.cn_pad;
#pragma unroll
  for (int ti = 0; ti < 7; ++ti) {
    const int tap = wave + 4 * ti;
#pragma unroll
    for (int a = 0; a < PA; ++a) {
      if (tap == 27 && a > 0) continue;
#pragma unroll
      for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = tap * P.Cg + a * 16 + 4 * q4 + e;
          if (row < P.kpad_w) slab[(int64_t)row * P.cn_pad + b * 16 + r16] = acc[ti][a][b][e];
        }
    }
  }
}

bool wgrad_halo_eligible(const ctseg_wgrad_desc* d) {
  if (d->dtype != CTSEG_BF16 || d->ntaps != 27 || d->sin != 1) return false;
  const int vb = d->Cg * 2, db = ((d->Cn + 15) / 16) * 32;
  if (!(vb == 32 || vb == 64) || !(db == 32 || db == 64)) return false;
  if (d->Xr != d->Xi || d->Yr != d->Yi || d->Zr != d->Zi || d->Zr < 4) return false;
  // rows 16-byte chunked, or 12 wide (24 bytes) on a 32-byte-plane operand
  const bool x_ok = d->g_ld % 8 == 0 || (vb == 32 && d->g_ld == 12), d_ok = d->d_ld % 8 == 0 || (db == 32 && d->d_ld == 12);
  if (!x_ok || !d_ok || d->d_ld < d->Cn || ((uintptr_t)d->in % 16) || ((uintptr_t)d->dy % 16)) return false;
  if (d->d_ld % 8 == 0 && d->d_ld < db / 2) return false;
  if (d->cn_pad < db / 2) return false;
  if ((int64_t)d->Xi * d->Yi * d->Zi * (d->g_ld > d->d_ld ? d->g_ld : d->d_ld) * 2 >= (1ll << 31)) return false;  // 32-bit per-sample byte offsets
  for (int j = 0; j < 27; ++j)
    for (int s = 0; s < 24; s += 8) {
      const int v = (int)(int8_t)((d->taps[j] >> s) & 0xff);
      if (v < -1 || v > 1) return false;
    }
  return true;
}

static int wgrad_halo_grid(const ctseg_wgrad_desc* d) {
  const int tiles = ((d->Xr + 3) / 4) * ((d->Yr + 7) / 8) * ((d->Zr + 7) / 8) * d->N;
  const int vb = d->Cg * 2, db = ((d->Cn + 15) / 16) * 32;
  // more workgroups per CU than this measured no faster with the single staging buffer (0.245 / 0.250 / 0.255 ms at 1 / 2 / 3
  // per CU for the 32->32 layer); the smaller LDS footprint is kept for what it leaves to the main stream's kernels
  const int per_cu = (vb + db <= 64) ? 2 : 1;
  int g = 256 * per_cu;
  return g < tiles ? g : tiles;
}

int wgrad_halo_slabs(const ctseg_wgrad_desc* d) { return wgrad_halo_grid(d); }

void launch_wgrad_halo(const ctseg_wgrad_desc* d, hipStream_t st) {
  WgradHaloArgs a;
  a.in = (const char*)d->in; a.dy = (const char*)d->dy; a.ws = d->ws;
  a.N = d->N; a.X = d->Xr; a.Y = d->Yr; a.Z = d->Zr; a.Cg = d->Cg; a.g_ld = d->g_ld; a.d_ld = d->d_ld;
  a.kpad_w = d->kpad_w; a.cn_pad = d->cn_pad;
  a.txn = (d->Xr + 3) / 4; a.tyn = (d->Yr + 7) / 8; a.tzn = (d->Zr + 7) / 8;
  a.tiles = a.txn * a.tyn * a.tzn;
  for (int j = 0; j < 28; ++j) {
    int v = 0;
    if (j < 27) {
      const int tp = d->taps[j];
      v = ((int)(int8_t)(tp & 0xff) * 10 + (int)(int8_t)((tp >> 8) & 0xff)) * 10 + (int)(int8_t)((tp >> 16) & 0xff);
    }
    a.delta[j] = v;
  }
  const int total = a.tiles * d->N, grid = wgrad_halo_grid(d);
  const int vb = d->Cg * 2, db = ((d->Cn + 15) / 16) * 32;
  if (vb == 32 && db == 32) hipLaunchKernelGGL((conv_wgrad_halo_kernel<32, 32>), dim3(grid), dim3(256), 0, st, a, total);
  else if (vb == 32) hipLaunchKernelGGL((conv_wgrad_halo_kernel<32, 64>), dim3(grid), dim3(256), 0, st, a, total);
  else if (db == 32) hipLaunchKernelGGL((conv_wgrad_halo_kernel<64, 32>), dim3(grid), dim3(256), 0, st, a, total);
  else hipLaunchKernelGGL((conv_wgrad_halo_kernel<64, 64>), dim3(grid), dim3(256), 0, st, a, total);
}

}  // namespace ctseg
