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

#ifndef WH_ABL
#define WH_ABL 0     // timing-only ablation of conv_wgrad_head_kernel: 1 = no MFMAs, 2 = no LDS operand reads, 4 = no global loads, 8 = no LDS staging stores
#endif

namespace ctseg {

struct WgradHaloArgs {
  const char* in;
  const char* dy;
  float* ws;
  int N, X, Y, Z, Cg, g_ld, d_ld;
  int kpad_w, cn_pad;
  int tiles, txn, tyn, tzn;   // tiles per sample and per axis
  int delta[28];              // halo voxel delta of each tap (in 32-byte rows), entry 27 unused
  // InstanceNorm + PReLU of the gathered operand on load (ctseg_wgrad_desc::in_mean_rstd): conv_wgrad_head2_kernel only
  const float* in_mr;
  const float* in_alpha;
  int in_C;
};

constexpr int WH_NRM_MAXN = 16;

constexpr int WH_HV = 600, WH_TV = 256;

// NW waves split the 27 taps (+ the all-ones pseudo tap): 4 waves x 7 taps, or 8 waves x 3-4 taps.  With 64-byte operands the 7-tap
// accumulators (112 registers) left one wave per SIMD (302 registers): 8 waves with <= 4 taps each fit two per SIMD.
template <int VB, int DBY, int NW>
__global__ __launch_bounds__(64 * NW) void conv_wgrad_halo_kernel(const WgradHaloArgs P, int total_tiles) {
  constexpr int NTH = 64 * NW, TPW = (28 + NW - 1) / NW, TFULL = 27 / NW;    // taps per wave; taps every wave owns (the last may not exist)
  constexpr int PA = VB / 32, PB = DBY / 32;
  // plane pitches padded by 64 B: with two planes the 8 lanes of a 16-byte staging store hold 2 voxels x 2 planes, and pitches
  // that are multiples of the 256-byte bank period (19 200 / 8 192 B) put both planes of a voxel on the same banks
  constexpr int XPL = WH_HV * 32 + (PA > 1 ? 64 : 0), DPL = WH_TV * 32 + (PB > 1 ? 64 : 0);
  constexpr int XBYTES = PA * XPL, DBYTES = PB * DPL, BUF = XBYTES + DBYTES;
  constexpr int XCH = WH_HV * (VB / 16), DCH = WH_TV * (DBY / 16);
  constexpr int JX = (XCH + NTH - 1) / NTH, JD = (DCH + NTH - 1) / NTH;
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
    const int idx = tid + j * NTH;
    const int c = idx % (VB / 16), hv = idx / (VB / 16);
    const int hx = hv / 100, rem = hv - hx * 100, hy = rem / 10, hz = rem - hy * 10;
    gx_byte[j] = (((hx - 1) * YZ + (hy - 1) * P.Z + (hz - 1)) * P.g_ld + c * 8) * 2;
    gx_hxyz[j] = (idx < XCH) ? (hx | (hy << 8) | (hz << 16)) : 0x7f7f7f;
    gx_lds[j] = (c >> 1) * XPL + hv * 32 + (c & 1) * 16;
  }
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int idx = tid + j * NTH;
    const int c = idx % (DBY / 16), tv = idx / (DBY / 16);
    const int tx = tv >> 6, ty = (tv >> 3) & 7, tz = tv & 7;
    gd_byte[j] = ((tx * YZ + ty * P.Z + tz) * P.d_ld + c * 8) * 2;
    gd_xyz[j] = (idx < DCH) ? (tx | (ty << 8) | (tz << 16)) : 0x7f7f7f;
    gd_lds[j] = (c >> 1) * DPL + tv * 32 + (c & 1) * 16;
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
      if (JX * NTH == XCH || tid + j * NTH < XCH) *reinterpret_cast<u32x4*>(xs + gx_lds[j]) = rx[j];
#pragma unroll
    for (int j = 0; j < JD; ++j)
      if (JD * NTH == DCH || tid + j * NTH < DCH) *reinterpret_cast<u32x4*>(ds + gd_lds[j]) = rd[j];
  };

  f32x4 acc[TPW][PA][PB];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
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
        const char* p0 = ds + b * DPL + (((x * 8) + (yb + ly)) * 8 + lz) * 32 + tp * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 8 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        df[b] = __builtin_bit_cast(bf16x8, v);
      }
      const int hbase = (((x + 1) * 10) + (yb + ly + 1)) * 10 + (lz + 1);
      // taps wave, wave+4, ..., wave+20 always exist: fetch all their operands first, then issue the MFMAs back to back
      // (branch-free, so the LDS latency of one tap hides behind the others); tap wave+24 may be the pseudo tap 27
      bf16x8 af[TFULL][PA];
#pragma unroll
      for (int ti = 0; ti < TFULL; ++ti) {
        const int h = hbase + P.delta[wave + NW * ti];
#pragma unroll
        for (int a = 0; a < PA; ++a) {
          const char* p0 = xs + a * XPL + h * 32 + tp * 8;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 10 * 32));
          const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[ti][a] = __builtin_bit_cast(bf16x8, v);
        }
      }
#pragma unroll
      for (int ti = 0; ti < TFULL; ++ti)
#pragma unroll
        for (int a = 0; a < PA; ++a)
#pragma unroll
          for (int b = 0; b < PB; ++b)
            acc[ti][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ti][a], df[b], acc[ti][a][b], 0, 0, 0);
      if constexpr (TPW > TFULL) {
        const int tap = wave + NW * TFULL;      // wave-uniform: a real tap, the pseudo tap 27, or nothing
        if (tap < 27) {
          const int h = hbase + P.delta[tap];
#pragma unroll
          for (int a = 0; a < PA; ++a) {
            const char* p0 = xs + a * XPL + h * 32 + tp * 8;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 10 * 32));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const bf16x8 a6 = __builtin_bit_cast(bf16x8, v);
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[TFULL][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a6, df[b], acc[TFULL][a][b], 0, 0, 0);
          }
        } else if (tap == 27) {  // pseudo tap 27: x == 1 on channel row 0 -> row 0 of the tile accumulates sum(dy) (bias gradient)
#pragma unroll
          for (int b = 0; b < PB; ++b)
            acc[TFULL][0][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), df[b], acc[TFULL][0][b], 0, 0, 0);
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
  for (int ti = 0; ti < TPW; ++ti) {
    const int tap = wave + NW * ti;
    if (tap > 27) continue;
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The 16 -> <= 16 channel layer at full resolution (the logits convolution of the head: 25 M voxels, 1.2 GB of operands, 348 GFLOP of
// padded MFMA work) is HBM-latency bound in the kernel above: a tile's loads are issued one tile (~900 cycles of multiplies) before
// they are written to LDS, against ~3 us of memory latency, and two workgroups per CU keep 54 KB in flight (0.65 ms; 1.10 ms with one
// workgroup per CU).  This variant keeps TWO tiles of loads in flight per workgroup in two register sets (tile t+2 is requested at
// the top of iteration t and written to LDS at the end of iteration t+1), moves rows in 8-byte pieces with raw buffer loads — 12-wide
// (24-byte) and 16-wide rows alike, a voxel outside the volume is an out-of-range offset selected by one v_and / v_cmp / v_cndmask —
// and has no per-load branches.  The multiplies are those of conv_wgrad_halo_kernel<32, 32>.
template <int NPX, int NPD>      // 8-byte pieces per x / dy voxel row: 3 (12 wide) or 4 (16 wide)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_wgrad_head_kernel(const WgradHaloArgs P, int total_tiles, int x_sample_bytes, int d_sample_bytes) {
  constexpr int XBYTES = WH_HV * 32, DBYTES = WH_TV * 32, BUF = XBYTES + DBYTES;
  constexpr int XN = WH_HV * NPX, DN = WH_TV * NPD, JX = (XN + 255) / 256, JD = (DN + 255) / 256;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef s16x4 __attribute__((address_space(3)))* lds_s16x4;
  __shared__ __attribute__((aligned(16))) char smem[BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int tq = r16 >> 2, tp = r16 & 3;
  const int YZ = P.Y * P.Z;

  int xo[JX], xl[JX], dof[JD], dl[JD];
  uint32_t xh[JX], dh[JD];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int idx = tid + j * 256, hv = idx / NPX, part = idx - hv * NPX;
    const int hx = hv / 100, rem = hv - hx * 100, hy = rem / 10, hz = rem - hy * 10;
    xo[j] = (hx * YZ + hy * P.Z + hz) * P.g_ld * 2 + part * 8;
    xh[j] = idx < XN ? ((1u << hx) | (1u << (6 + hy)) | (1u << (16 + hz))) : 0x80000000u;
    xl[j] = hv * 32 + part * 8;
  }
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int idx = tid + j * 256, tv = idx / NPD, part = idx - tv * NPD;
    const int tx = tv >> 6, ty = (tv >> 3) & 7, tz = tv & 7;
    dof[j] = (tx * YZ + ty * P.Z + tz) * P.d_ld * 2 + part * 8;
    dh[j] = idx < DN ? ((1u << tx) | (1u << (4 + ty)) | (1u << (12 + tz))) : 0x80000000u;
    dl[j] = XBYTES + tv * 32 + part * 8;
  }
  const int bias_bytes = (YZ + P.Z + 1) * P.g_ld * 2;
  auto range_mask = [](int lo, int hi, int nbits) -> uint32_t {
    lo = lo < 0 ? 0 : lo;
    hi = hi > nbits - 1 ? nbits - 1 : hi;
    return hi < lo ? 0u : ((2u << hi) - (1u << lo));
  };
  auto gload = [&](int t, u32x2 (&rx)[JX], u32x2 (&rd)[JD]) {
    const int n = t / P.tiles;
    int r = t - n * P.tiles;
    const int tz = r % P.tzn; r /= P.tzn;
    const int ty = r % P.tyn, tx = r / P.tyn;
    const int x0 = tx * 4, y0 = ty * 8, z0 = tz * 8;
    const uint32_t xm = ~(range_mask(1 - x0, P.X - x0, 6) | (range_mask(1 - y0, P.Y - y0, 10) << 6) | (range_mask(1 - z0, P.Z - z0, 10) << 16));
    const uint32_t dm = ~(range_mask(0, P.X - x0 - 1, 4) | (range_mask(0, P.Y - y0 - 1, 8) << 4) | (range_mask(0, P.Z - z0 - 1, 8) << 12));
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.in) + (int64_t)n * x_sample_bytes - bias_bytes, 0,
                                                                         x_sample_bytes + bias_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.dy) + (int64_t)n * d_sample_bytes, 0, d_sample_bytes,
                                                                         0x00020000);
    const int xs = ((x0 * P.Y + y0) * P.Z + z0) * P.g_ld * 2, dsf = ((x0 * P.Y + y0) * P.Z + z0) * P.d_ld * 2;
#pragma unroll
    for (int j = 0; j < JX; ++j)
      rx[j] = (WH_ABL & 4) ? u32x2{(uint32_t)xs, 0u} : __builtin_amdgcn_raw_buffer_load_b64(xr, (xh[j] & xm) == 0u ? xo[j] : (int)0x80000000, xs, 0);
#pragma unroll
    for (int j = 0; j < JD; ++j)
      rd[j] = (WH_ABL & 4) ? u32x2{(uint32_t)dsf, 0u} : __builtin_amdgcn_raw_buffer_load_b64(dr, (dh[j] & dm) == 0u ? dof[j] : (int)0x80000000, dsf, 0);
  };
  auto sstore = [&](const u32x2 (&rx)[JX], const u32x2 (&rd)[JD]) {
    if (WH_ABL & 8) { if (rx[0][0] == 0x12345u && rd[0][0] == 0x54321u) smem[tid] = 1; return; }
#pragma unroll
    for (int j = 0; j < JX; ++j)
      if (JX * 256 == XN || tid + j * 256 < XN) *reinterpret_cast<u32x2*>(smem + xl[j]) = rx[j];
#pragma unroll
    for (int j = 0; j < JD; ++j)
      if (JD * 256 == DN || tid + j * 256 < DN) *reinterpret_cast<u32x2*>(smem + dl[j]) = rd[j];
  };

  f32x4 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lz = 4 * (q4 & 1) + tq, ly = q4 >> 1;
  s16x8 ones;
  {
    const short o = (r16 == 0) ? (short)0x3f80 : (short)0;
    ones = s16x8{o, o, o, o, o, o, o, o};
  }
  auto compute = [&]() {
    const char* xs = smem;
    const char* ds = xs + XBYTES;
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      const int x = s >> 1, yb = 4 * (s & 1);
      bf16x8 df;
      {
        const char* p0 = ds + (((x * 8) + (yb + ly)) * 8 + lz) * 32 + tp * 8;
        const s16x4 lo = (WH_ABL & 2) ? s16x4{(short)s, 1, 2, 3} : __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = (WH_ABL & 2) ? s16x4{(short)s, 1, 2, 3} : __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 8 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        df = __builtin_bit_cast(bf16x8, v);
      }
      const int hbase = (((x + 1) * 10) + (yb + ly + 1)) * 10 + (lz + 1);
      bf16x8 af[6];
#pragma unroll
      for (int ti = 0; ti < 6; ++ti) {
        const char* p0 = xs + (hbase + P.delta[wave + 4 * ti]) * 32 + tp * 8;
        const s16x4 lo = (WH_ABL & 2) ? s16x4{(short)(s + tp), 1, 2, 3} : __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = (WH_ABL & 2) ? s16x4{(short)(s + tp), 1, 2, 3} : __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 10 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[ti] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int ti = 0; ti < 6; ++ti) {
        if constexpr ((WH_ABL & 1) != 0) acc[ti][0] += __builtin_bit_cast(f32x4, af[ti])[0] * __builtin_bit_cast(f32x4, df)[1];
        else acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ti], df, acc[ti], 0, 0, 0);
      }
      const int tap = wave + 24;
      if (tap < 27) {
        const char* p0 = xs + (hbase + P.delta[tap]) * 32 + tp * 8;
        const s16x4 lo = (WH_ABL & 2) ? s16x4{(short)(s + tp), 1, 2, 3} : __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = (WH_ABL & 2) ? s16x4{(short)(s + tp), 1, 2, 3} : __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 10 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), df, acc[6], 0, 0, 0);
      } else {   // pseudo tap 27: row 0 accumulates sum(dy) (bias gradient)
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), df, acc[6], 0, 0, 0);
      }
    }
  };

  int t = blockIdx.x, tstride = gridDim.x, tlast = total_tiles;
  if ((gridDim.x & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * chunk + (blockIdx.x >> 3);
    tstride = gridDim.x >> 3;
    tlast = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  }
  // rows narrower than 32 bytes: the pad bytes of every slot are zero for the whole launch
  if (NPX < 4) for (int i = tid; i < WH_HV; i += 256) *reinterpret_cast<u32x2*>(smem + i * 32 + 24) = u32x2{0u, 0u};
  if (NPD < 4) for (int i = tid; i < WH_TV; i += 256) *reinterpret_cast<u32x2*>(smem + XBYTES + i * 32 + 24) = u32x2{0u, 0u};
  // ONE register set of loads in flight per workgroup, FOUR workgroups per CU (128 registers each): the phases of a workgroup
  // (loads -> multiplies -> barrier -> LDS stores -> barrier) do not overlap inside it; they overlap with the other three's.
  u32x2 rxA[JX], rdA[JD];
  if (t < tlast) {
    gload(t, rxA, rdA);
    sstore(rxA, rdA);
  }
  __syncthreads();
  for (; t < tlast; t += tstride) {
    if (t + tstride < tlast) gload(t + tstride, rxA, rdA);
    compute();
    __syncthreads();
    if (t + tstride < tlast) sstore(rxA, rdA);
    __syncthreads();
  }

  float* slab = P.ws + (int64_t)blockIdx.x * P.kpad_w * P.cn_pad;
#pragma unroll
  for (int ti = 0; ti < 7; ++ti) {
    const int tap = wave + 4 * ti;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = tap * P.Cg + 4 * q4 + e;
      if (row < P.kpad_w) slab[(int64_t)row * P.cn_pad + r16] = acc[ti][e];
    }
  }
}

// x-column reuse for the transposed operand reads (round 2): a k-step is 32 voxels of ONE x plane (4 y rows x 8 z), so the x^T
// fragment of output plane p for tap (dx,dy,dz) IS the fragment of halo plane p+1+dx for tap (0,dy,dz).  Eight waves = two k halves
// (y rows 0-3 / 4-7 of every plane) x four groups of (dy,dz) combos {0,1,2} {3,4} {5,6} {7,8 + the all-ones pseudo tap}: per combo a
// wave reads the six halo-plane fragments of its half once (12 ds_read_b64_tr_b16) and feeds 12 MFMAs (3 dx x 4 planes) against the
// four dy fragments of its half — 280 transposed reads per tile instead of 512.  The two k halves accumulate into two slabs.
template <int NPX, int NPD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_wgrad_head2_kernel(const WgradHaloArgs P, int total_tiles, int x_sample_bytes, int d_sample_bytes) {
  constexpr int XBYTES = WH_HV * 32, DBYTES = WH_TV * 32, BUF = XBYTES + DBYTES;
  constexpr int XN = WH_HV * NPX, DN = WH_TV * NPD, JX = (XN + 511) / 512, JD = (DN + 511) / 512;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef s16x4 __attribute__((address_space(3)))* lds_s16x4;
  __shared__ __attribute__((aligned(16))) char smem[BUF + WH_NRM_MAXN * 128];
  float* const sPar = reinterpret_cast<float*>(smem + BUF);     // per sample: 4 channel quads x (mean x 4, rstd x 4)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int tq = r16 >> 2, tp = r16 & 3;
  const int YZ = P.Y * P.Z;

  int xo[JX], xl[JX], dof[JD], dl[JD];
  uint32_t xh[JX], dh[JD];
#pragma unroll
  for (int j = 0; j < JX; ++j) {
    const int idx = tid + j * 512, hv = idx / NPX, part = idx - hv * NPX;
    const int hx = hv / 100, rem = hv - hx * 100, hy = rem / 10, hz = rem - hy * 10;
    xo[j] = (hx * YZ + hy * P.Z + hz) * P.g_ld * 2 + part * 8;
    xh[j] = idx < XN ? ((1u << hx) | (1u << (6 + hy)) | (1u << (16 + hz))) : 0x80000000u;
    xl[j] = hv * 32 + part * 8;
  }
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int idx = tid + j * 512, tv = idx / NPD, part = idx - tv * NPD;
    const int tx = tv >> 6, ty = (tv >> 3) & 7, tz = tv & 7;
    dof[j] = (tx * YZ + ty * P.Z + tz) * P.d_ld * 2 + part * 8;
    dh[j] = idx < DN ? ((1u << tx) | (1u << (4 + ty)) | (1u << (12 + tz))) : 0x80000000u;
    dl[j] = XBYTES + tv * 32 + part * 8;
  }
  const int bias_bytes = (YZ + P.Z + 1) * P.g_ld * 2;
  auto range_mask = [](int lo, int hi, int nbits) -> uint32_t {
    lo = lo < 0 ? 0 : lo;
    hi = hi > nbits - 1 ? nbits - 1 : hi;
    return hi < lo ? 0u : ((2u << hi) - (1u << lo));
  };
  auto gload = [&](int t, u32x2 (&rx)[JX], u32x2 (&rd)[JD]) {
    const int n = t / P.tiles;
    int r = t - n * P.tiles;
    const int tz = r % P.tzn; r /= P.tzn;
    const int ty = r % P.tyn, tx = r / P.tyn;
    const int x0 = tx * 4, y0 = ty * 8, z0 = tz * 8;
    const uint32_t xm = ~(range_mask(1 - x0, P.X - x0, 6) | (range_mask(1 - y0, P.Y - y0, 10) << 6) | (range_mask(1 - z0, P.Z - z0, 10) << 16));
    const uint32_t dm = ~(range_mask(0, P.X - x0 - 1, 4) | (range_mask(0, P.Y - y0 - 1, 8) << 4) | (range_mask(0, P.Z - z0 - 1, 8) << 12));
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.in) + (int64_t)n * x_sample_bytes - bias_bytes, 0,
                                                                         x_sample_bytes + bias_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.dy) + (int64_t)n * d_sample_bytes, 0, d_sample_bytes,
                                                                         0x00020000);
    const int xs = ((x0 * P.Y + y0) * P.Z + z0) * P.g_ld * 2, dsf = ((x0 * P.Y + y0) * P.Z + z0) * P.d_ld * 2;
#pragma unroll
    for (int j = 0; j < JX; ++j)
      rx[j] = (WH_ABL & 4) ? u32x2{(uint32_t)xs, 0u} : __builtin_amdgcn_raw_buffer_load_b64(xr, (xh[j] & xm) == 0u ? xo[j] : (int)0x80000000, xs, 0);
#pragma unroll
    for (int j = 0; j < JD; ++j)
      rd[j] = (WH_ABL & 4) ? u32x2{(uint32_t)dsf, 0u} : __builtin_amdgcn_raw_buffer_load_b64(dr, (dh[j] & dm) == 0u ? dof[j] : (int)0x80000000, dsf, 0);
  };
  // operand normalisation on load (the arithmetic of instnorm_prelu_fwd_kernel, rounded to bf16 as that pass rounds): x pieces only;
  // halo voxels outside the volume and channels >= in_C stay 0.  Piece j holds channel quad (tid + j * 512) % NPX.
  const bool nrm = P.in_mr != nullptr;
  const float nrm_al = nrm ? P.in_alpha[0] : 1.f;
  if (nrm)
    for (int i = tid; i < P.N * 32; i += 512) {
      const int n = i >> 5, k = i & 31, c = (k >> 3) * 4 + (k & 3);
      sPar[i] = c < P.in_C ? P.in_mr[((int64_t)n * P.in_C + c) * 2 + ((k >> 2) & 1)] : 0.f;
    }
  auto xmask = [&](int t, int& n) -> uint32_t {
    n = t / P.tiles;
    int r = t - n * P.tiles;
    const int tz = r % P.tzn; r /= P.tzn;
    const int ty = r % P.tyn, tx = r / P.tyn;
    const int x0 = tx * 4, y0 = ty * 8, z0 = tz * 8;
    return ~(range_mask(1 - x0, P.X - x0, 6) | (range_mask(1 - y0, P.Y - y0, 10) << 6) | (range_mask(1 - z0, P.Z - z0, 10) << 16));
  };
  auto sstore = [&](int t, const u32x2 (&rx)[JX], const u32x2 (&rd)[JD]) {
    if (!nrm) {
#pragma unroll
      for (int j = 0; j < JX; ++j)
        if ((JX * 512 == XN || tid + j * 512 < XN) && (!(WH_ABL & 8) || rx[j][0] == 0x12345u)) *reinterpret_cast<u32x2*>(smem + xl[j]) = rx[j];
    } else {
      int n;
      const uint32_t xm = xmask(t, n);
      const float* par = sPar + n * 32;
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        const int part = (tid + j * 512) % NPX;
        const f32x4 mean = *reinterpret_cast<const f32x4*>(par + part * 8), rstd = *reinterpret_cast<const f32x4*>(par + part * 8 + 4);
        float v[4] = {h2f<BF16>(rx[j][0] & 0xffffu), h2f<BF16>(rx[j][0] >> 16), h2f<BF16>(rx[j][1] & 0xffffu), h2f<BF16>(rx[j][1] >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = (v[e] - mean[e]) * rstd[e];
          v[e] = a > 0.f ? a : nrm_al * a;
        }
        const u32x2 w = (xh[j] & xm) == 0u ? u32x2{pack2<BF16>(v[0], v[1]), pack2<BF16>(v[2], v[3])} : u32x2{0u, 0u};
        if (JX * 512 == XN || tid + j * 512 < XN) *reinterpret_cast<u32x2*>(smem + xl[j]) = w;
      }
    }
#pragma unroll
    for (int j = 0; j < JD; ++j)
      if (JD * 512 == DN || tid + j * 512 < DN) *reinterpret_cast<u32x2*>(smem + dl[j]) = rd[j];
  };
  const int kh = wave >> 2, cg = wave & 3;               // k half, combo group
  const int c0 = cg == 0 ? 0 : 2 * cg + 1, nco = cg == 0 ? 3 : 2;      // first (dy,dz) combo of the group, how many
  f32x4 acc[3][3], acc_ps = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ci = 0; ci < 3; ++ci)
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) acc[ci][dxi] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lz = 4 * (q4 & 1) + tq, ly = q4 >> 1;
  s16x8 ones;
  {
    const short o = (r16 == 0) ? (short)0x3f80 : (short)0;
    ones = s16x8{o, o, o, o, o, o, o, o};
  }
  auto tr_frag = [&](const char* p0, int hi_off) -> bf16x8 {
    if constexpr ((WH_ABL & 2) != 0) { const short v0 = (short)(uintptr_t)p0; return __builtin_bit_cast(bf16x8, s16x8{v0, v0, v0, v0, v0, v0, v0, v0}); }   // (timing only)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + hi_off));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&]() {
    const char* xs = smem;
    const char* ds = xs + XBYTES;
    const int yb = 4 * kh;
    bf16x8 df[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) df[p] = tr_frag(ds + (((p * 8) + (yb + ly)) * 8 + lz) * 32 + tp * 8, 2 * 8 * 32);
    const int hb0 = (yb + ly + 1) * 10 + (lz + 1);          // halo voxel of this lane in plane 0, before the (dy,dz) shift
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
      if (ci < nco) {                                        // wave-uniform
        const int dyz = P.delta[9 + c0 + ci];                // taps 9..17 are the dx = 0 group: delta = dy * 10 + dz
        bf16x8 xf[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) xf[q] = tr_frag(xs + (q * 100 + hb0 + dyz) * 32 + tp * 8, 2 * 10 * 32);
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            if constexpr ((WH_ABL & 1) != 0) { acc[ci][dxi][0] += __builtin_bit_cast(f32x4, xf[p + dxi])[0] * __builtin_bit_cast(f32x4, df[p])[1]; continue; }   // (timing only)
            acc[ci][dxi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[p + dxi], df[p], acc[ci][dxi], 0, 0, 0);
          }
      }
    }
    if (cg == 3) {       // pseudo tap 27: row 0 accumulates sum(dy) (bias gradient)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc_ps = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones), df[p], acc_ps, 0, 0, 0);
    }
  };

  int t = blockIdx.x, tstride = gridDim.x, tlast = total_tiles;
  if ((gridDim.x & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    t = xcd * chunk + (blockIdx.x >> 3);
    tstride = gridDim.x >> 3;
    tlast = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  }
  // rows narrower than 32 bytes: the pad bytes of every slot are zero for the whole launch
  if (NPX < 4) for (int i = tid; i < WH_HV; i += 512) *reinterpret_cast<u32x2*>(smem + i * 32 + 24) = u32x2{0u, 0u};
  if (NPD < 4) for (int i = tid; i < WH_TV; i += 512) *reinterpret_cast<u32x2*>(smem + XBYTES + i * 32 + 24) = u32x2{0u, 0u};
  // ONE register set of loads in flight per workgroup, FOUR workgroups per CU (128 registers each): the phases of a workgroup
  // (loads -> multiplies -> barrier -> LDS stores -> barrier) do not overlap inside it; they overlap with the other three's.
  u32x2 rxA[JX], rdA[JD];
  __syncthreads();          // the operand-normalisation table is written
  if (t < tlast) {
    gload(t, rxA, rdA);
    sstore(t, rxA, rdA);
  }
  __syncthreads();
  for (; t < tlast; t += tstride) {
    if (t + tstride < tlast) gload(t + tstride, rxA, rdA);
    compute();
    __syncthreads();
    if (t + tstride < tlast) sstore(t + tstride, rxA, rdA);
    __syncthreads();
  }

  float* slab = P.ws + ((int64_t)blockIdx.x * 2 + kh) * P.kpad_w * P.cn_pad;       // one slab per k half
#pragma unroll
  for (int ci = 0; ci < 3; ++ci) {
    if (ci >= nco) continue;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int tap = dxi * 9 + c0 + ci;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = tap * P.Cg + 4 * q4 + e;
        if (row < P.kpad_w) slab[(int64_t)row * P.cn_pad + r16] = acc[ci][dxi][e];
      }
    }
  }
  if (cg == 3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 27 * P.Cg + 4 * q4 + e;
      if (row < P.kpad_w) slab[(int64_t)row * P.cn_pad + r16] = acc_ps[e];
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

static bool wgrad_head2(const ctseg_wgrad_desc* d);
static int wgrad_halo_grid(const ctseg_wgrad_desc* d) {
  const int tiles = ((d->Xr + 3) / 4) * ((d->Yr + 7) / 8) * ((d->Zr + 7) / 8) * d->N;
  const int vb = d->Cg * 2, db = ((d->Cn + 15) / 16) * 32;
  // more workgroups per CU than this measured no faster with the single staging buffer (0.245 / 0.250 / 0.255 ms at 1 / 2 / 3
  // per CU for the 32->32 layer); the smaller LDS footprint is kept for what it leaves to the main stream's kernels
  int per_cu = (vb + db <= 64) ? 2 : 1;
  if (vb == 32 && db == 32 && getenv("CTSEG_WGRAD_HEAD_OLD") == nullptr && (d->g_ld == 12 || d->g_ld == 16) && (d->d_ld == 12 || d->d_ld == 16))
    per_cu = wgrad_head2(d) ? 1 : 4;      // 27 KB of LDS, 128 registers: four 4-wave or two 8-wave workgroups per CU would fit.
  // ONE 8-wave workgroup per CU for the head kernel: alone it is slower that way (0.43 -> 0.50 ms), inside the step — it runs on the
  // side stream beside the main stream's HBM-bound head passes — faster: 9.94 / 9.90 -> 9.86 / 9.82 ms/step (same box, round 2)
  if (const char* e = getenv("CTSEG_WH_PER_CU")) per_cu = atoi(e);
  return persistent_grid(CTSEG_NUM_CU * per_cu, tiles);
}

static bool wgrad_head2(const ctseg_wgrad_desc* d) {
  const int vb = d->Cg * 2, db = ((d->Cn + 15) / 16) * 32;
  for (int j = 0; j < 27; ++j) {      // the x-column reuse indexes taps as dx * 9 + (dy, dz): canonical order only
    const int tp = d->taps[j];
    if ((int)(int8_t)(tp & 0xff) != j / 9 - 1 || (int)(int8_t)((tp >> 8) & 0xff) != (j / 3) % 3 - 1 || (int)(int8_t)((tp >> 16) & 0xff) != j % 3 - 1)
      return false;
  }
  return vb == 32 && db == 32 && getenv("CTSEG_WGRAD_HEAD_OLD") == nullptr && getenv("CTSEG_WGRAD_HEAD_V1") == nullptr &&
         (d->g_ld == 12 || d->g_ld == 16) && (d->d_ld == 12 || d->d_ld == 16);
}

bool wgrad_halo_in_norm_ok(const ctseg_wgrad_desc* d) { return wgrad_halo_eligible(d) && wgrad_head2(d) && d->N <= WH_NRM_MAXN && d->in_norm_C <= 12; }

int wgrad_halo_slabs(const ctseg_wgrad_desc* d) { return wgrad_head2(d) ? 2 * wgrad_halo_grid(d) : wgrad_halo_grid(d); }

void launch_wgrad_halo(const ctseg_wgrad_desc* d, hipStream_t st) {
  WgradHaloArgs a;
  a.in = (const char*)d->in; a.dy = (const char*)d->dy; a.ws = d->ws;
  a.N = d->N; a.X = d->Xr; a.Y = d->Yr; a.Z = d->Zr; a.Cg = d->Cg; a.g_ld = d->g_ld; a.d_ld = d->d_ld;
  a.kpad_w = d->kpad_w; a.cn_pad = d->cn_pad;
  a.in_mr = d->in_mean_rstd; a.in_alpha = d->in_alpha; a.in_C = d->in_norm_C;
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
  if (vb == 32 && db == 32 && getenv("CTSEG_WGRAD_HEAD_OLD") == nullptr && (d->g_ld == 12 || d->g_ld == 16) && (d->d_ld == 12 || d->d_ld == 16)) {
    const int xsb = (int)((int64_t)d->Xi * d->Yi * d->Zi * d->g_ld * 2), dsb = (int)((int64_t)d->Xi * d->Yi * d->Zi * d->d_ld * 2);
    const bool v2 = wgrad_head2(d);
#define WH_GO(NX, ND)                                                                                                       \
  do {                                                                                                                      \
    if (v2) hipLaunchKernelGGL((conv_wgrad_head2_kernel<NX, ND>), dim3(grid), dim3(512), 0, st, a, total, xsb, dsb);         \
    else hipLaunchKernelGGL((conv_wgrad_head_kernel<NX, ND>), dim3(grid), dim3(256), 0, st, a, total, xsb, dsb);             \
  } while (0)
    if (d->g_ld == 12) { if (d->d_ld == 12) WH_GO(3, 3); else WH_GO(3, 4); }
    else { if (d->d_ld == 12) WH_GO(4, 3); else WH_GO(4, 4); }
#undef WH_GO
  } else if (vb == 32 && db == 32) hipLaunchKernelGGL((conv_wgrad_halo_kernel<32, 32, 4>), dim3(grid), dim3(256), 0, st, a, total);
  else if (vb == 32) hipLaunchKernelGGL((conv_wgrad_halo_kernel<32, 64, 4>), dim3(grid), dim3(256), 0, st, a, total);
  else if (db == 32) hipLaunchKernelGGL((conv_wgrad_halo_kernel<64, 32, 4>), dim3(grid), dim3(256), 0, st, a, total);
  else if (getenv("CTSEG_WGRAD_HALO_4W") != nullptr) hipLaunchKernelGGL((conv_wgrad_halo_kernel<64, 64, 4>), dim3(grid), dim3(256), 0, st, a, total);
  else hipLaunchKernelGGL((conv_wgrad_halo_kernel<64, 64, 8>), dim3(grid), dim3(512), 0, st, a, total);
}

}  // namespace ctseg
