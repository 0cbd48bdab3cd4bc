// Weight-gradient pass for gfx950: R[slot*Cg + a][b] = sum_rows in[row*sin + d(slot)][a] * dy[row][b]
// (autograd's Conv3d / ConvTranspose3d weight + bias backward in the reference's training step,
// capstone/volumetric/base_trainer.py:80-82 -> loss.backward()).
//
// The contraction runs over voxels (rows), which is the SLOW axis of both channels-last operands, so
// the MFMA fragments need a transpose: bf16 uses ds_read_b64_tr_b16 (hardware transposed LDS read,
// 4 voxels x 16 channels per 16-lane group); fp32 feeds v_mfma_f32_16x16x4_f32 one element per lane,
// which needs none.  A workgroup owns 128 K-rows x BNW columns and a contiguous row range of one
// sample (split-K); partial tiles go to fp32 slabs that ctseg_conv_wgrad_reduce sums in fixed order
// (deterministic) straight into the torch weight layout.  K index ntaps*Cg is a virtual all-ones
// gathered channel, so its row of R is the bias gradient.
#include "ctseg_dev.h"

namespace ctseg {

struct WgradKArgs {
  const char* in;
  const char* dy;
  float* ws;
  int N, Xi, Yi, Zi, Xr, Yr, Zr;
  int Cg, Cn, g_ld, d_ld, sin, ntaps;
  int rows, splits, rows_per_split;
  int kpad_w, cn_pad, d_valid;
  int sx, sy, sz;  // mixed-radix decomposition of a 32-row step
  int kblocks, cblocks;   // > 0: 1-D grid, workgroup -> (K block, column block, slab) decoded so that one XCD owns a slab
  int addr64;             // 1: a sample of the gathered operand is >= 2 GiB (or CTSEG_WGRAD_ADDR64): 64-bit address chain per chunk
  int taps[CTSEG_MAX_TAPS];
};

// GLDS (bf16, 16-byte-chunked gather, >= 64 columns): operands go global -> LDS directly (global_load_lds, no VGPR staging,
// no ds_write pass).  The LDS image is then lane-linear (no row padding), so the bank spread the padding gave the transposed
// reads comes from an XOR of the 32-byte column slot with the row instead, applied on the SOURCE side (which chunk a lane
// fetches) and in the read address: 256-byte rows: slot ^ (row & 7); 128-byte rows: slot ^ ((row >> 1) & 3).
__device__ __attribute__((aligned(16))) unsigned short g_wg_zero16[8] = {0, 0, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) unsigned short g_wg_one16[8] = {0x3f80, 0, 0, 0, 0, 0, 0, 0};   // bf16 1.0 on channel 0

typedef int32_t wg_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* wg_lds_u32_ptr;
__device__ void wg_raw_buffer_load_lds(wg_i32x4 rsrc, wg_lds_u32_ptr lds, int size, int voffset, int soffset, int offset,
                                       int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
__device__ __forceinline__ wg_i32x4 wg_make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  wg_i32x4 v = __builtin_bit_cast(wg_i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]);
  v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]);
  v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

template <typename T, int BNW, bool SMALLC> struct WgradCfg {
  static constexpr int SZ = TT<T>::SZ;
  static constexpr bool GLDS = SZ == 2 && !SMALLC && BNW >= 64;
  static constexpr int PA = GLDS ? 256 : 128 * SZ + (SZ == 2 ? 32 : 64);
  static constexpr int PD = GLDS ? BNW * 2 : BNW * SZ + ((SZ == 2) ? (BNW == 16 ? 64 : 32) : 64);
  static constexpr int STAGE = 32 * (PA + PD);
};

template <typename T, int BNW, int WK, int WC, bool SMALLC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradKArgs P) {
  constexpr int SZ = TT<T>::SZ, EPC = TT<T>::EPC;
  using CF = WgradCfg<T, BNW, SMALLC>;
  constexpr int PA = CF::PA, PD = CF::PD;
  constexpr bool GLDS = CF::GLDS;
  constexpr int KT = 128 / WK / 16, CT = BNW / WC / 16;
  constexpr int ACPR = 128 / EPC;            // 16-byte chunks per gathered row (16 bf16 / 32 fp32)
  constexpr int AJ = 32 * ACPR / 256;        // gathered chunks per thread per stage (2 / 4)
  constexpr int ARS = 256 / ACPR;            // row stride between a thread's chunks (16 / 8)
  constexpr int DCPR = BNW / EPC;            // chunks per dy row
  constexpr int DJ = (32 * DCPR + 255) / 256;
  static_assert(WK * WC == 4 && KT >= 1 && CT >= 1, "4 waves");

  __shared__ __attribute__((aligned(16))) char smem[2 * CF::STAGE];
  __shared__ int sTap[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 32) sTap[tid] = tid < P.ntaps ? P.taps[tid] : 0;
  __syncthreads();
  const int wk = wave / WC, wc = wave % WC;
  const int r16 = lane & 15, q4 = lane >> 4;
  // Workgroups go to the 8 XCDs round-robin by linear id.  With the plain 3-D grid the K / column blocks of one slab (the
  // same rows of `in` and `dy`) land on different XCDs and each XCD's L2 fetches its own copy.  1-D grid: ids L, L+8, L+16, ...
  // inside a group of 8 * kblocks * cblocks are the blocks of ONE slab -> same XCD, dispatched together, one fetch.
  int kblock, cblock, zslab;
  if (P.kblocks > 0) {
    const int kc = P.kblocks * P.cblocks, G = 8 * kc;
    const int L = blockIdx.x, g = L / G, r = L - g * G;
    const int q = r >> 3;
    zslab = g * 8 + (r & 7);
    kblock = q % P.kblocks;
    cblock = q / P.kblocks;
  } else {
    kblock = blockIdx.x; cblock = blockIdx.y; zslab = blockIdx.z;
  }
  const int col0 = cblock * BNW;
  const int n = zslab / P.splits, sp = zslab % P.splits;
  const int mstart = sp * P.rows_per_split;
  int mend = mstart + P.rows_per_split;
  if (mend > P.rows) mend = P.rows;
  const int nst = (mend > mstart) ? (mend - mstart + 31) / 32 : 0;

  // ---- fixed K position of this thread's gathered chunks -----------------------------------------
  const int arow0 = tid / ACPR;
  // GLDS: the thread's LDS position is (row, tid % 16); it holds logical chunk ((slot32 ^ (row & 7)) << 1) | low
  const int acc_c = GLDS ? (((((tid % ACPR) >> 1) ^ (arow0 & 7)) << 1) | (tid & 1)) : tid % ACPR;
  const int kpos = kblock * 128 + acc_c * EPC;
  const int ktot = P.ntaps * P.Cg;
  const int slot = kpos / P.Cg, ci = kpos - slot * P.Cg;
  const bool kvalid = kpos < ktot, kones = kpos == ktot;
  int dx = 0, dy_ = 0, dz = 0;
  if (kvalid) {
    const int tp = sTap[slot & 31];
    dx = (int)(int8_t)(tp & 0xff); dy_ = (int)(int8_t)((tp >> 8) & 0xff); dz = (int)(int8_t)((tp >> 16) & 0xff);
  }
  int cx[AJ], cy[AJ], cz[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    int m = mstart + arow0 + j * ARS;
    cz[j] = m % P.Zr; int t = m / P.Zr;
    cy[j] = t % P.Yr; cx[j] = t / P.Yr;
  }
  const int64_t nbase = (int64_t)n * P.Xi;
  const char* dbase = P.dy + ((int64_t)n * P.rows * P.d_ld + col0) * SZ;

  u32x4 ra[AJ], rd[DJ];
  u32x4 ones = {0u, 0u, 0u, 0u};
  ones[0] = (SZ == 4) ? 0x3f800000u : 0x3f80u;
  auto gload = [&](int s) {
    const int mb = mstart + s * 32;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = mb + arow0 + j * ARS;
      u32x4 v = {0u, 0u, 0u, 0u};
      if constexpr (SMALLC) {
        // gathered channel count is not a multiple of the 16-byte chunk (the Cin = 1 stem): element-wise gather
        if (m < mend) {
          uint32_t el[EPC];
#pragma unroll
          for (int t = 0; t < EPC; ++t) {
            const int kp = kpos + t;
            uint32_t val = 0u;
            if (kp < ktot) {
              const int sl = kp / P.Cg, c = kp - sl * P.Cg;
              const int tp = sTap[sl & 31];
              const int xi = cx[j] * P.sin + (int)(int8_t)(tp & 0xff), yi = cy[j] * P.sin + (int)(int8_t)((tp >> 8) & 0xff),
                        zi = cz[j] * P.sin + (int)(int8_t)((tp >> 16) & 0xff);
              if ((unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi) {
                const int64_t vox = ((nbase + xi) * P.Yi + yi) * P.Zi + zi;
                const char* p = P.in + (vox * P.g_ld + c) * SZ;
                if constexpr (SZ == 4) val = *reinterpret_cast<const uint32_t*>(p);
                else val = *reinterpret_cast<const unsigned short*>(p);
              }
            } else if (kp == ktot) {
              val = (SZ == 4) ? 0x3f800000u : 0x3f80u;
            }
            el[t] = val;
          }
          if constexpr (SZ == 4) { v[0] = el[0]; v[1] = el[1]; v[2] = el[2]; v[3] = el[3]; }
          else {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = el[2 * t] | (el[2 * t + 1] << 16);
          }
        }
      } else if (m < mend) {
        if (kvalid) {
          const int xi = cx[j] * P.sin + dx, yi = cy[j] * P.sin + dy_, zi = cz[j] * P.sin + dz;
          if ((unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi) {
            const int64_t vox = ((nbase + xi) * P.Yi + yi) * P.Zi + zi;
            v = *reinterpret_cast<const u32x4*>(P.in + (vox * P.g_ld + ci) * SZ);
          }
        } else if (kones) {
          v = ones;
        }
      }
      ra[j] = v;
      // advance this chunk's row by 32 (mixed radix z,y,x)
      cz[j] += P.sz; if (cz[j] >= P.Zr) { cz[j] -= P.Zr; ++cy[j]; }
      cy[j] += P.sy; if (cy[j] >= P.Yr) { cy[j] -= P.Yr; ++cx[j]; }
      cx[j] += P.sx;
    }
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
      const int idx = tid + j * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < 32 * DCPR) {
        const int r = idx / DCPR, dc = idx - r * DCPR;
        const int m = mb + r;
        if (m < mend && col0 + dc * EPC < P.d_valid)
          v = *reinterpret_cast<const u32x4*>(dbase + ((int64_t)m * P.d_ld + dc * EPC) * SZ);
      }
      rd[j] = v;
    }
  };
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // GLDS: dy chunk this thread's LDS position holds (source-side swizzle, see WgradCfg)
  const int d_row0 = tid / DCPR, d_pos = tid % DCPR;
  const int d_c32 = (BNW == 128) ? ((d_pos >> 1) ^ (d_row0 & 7)) : ((d_pos >> 1) ^ ((d_row0 >> 1) & 3));
  const int d_chunk = (d_c32 << 1) | (d_pos & 1);
  // GLDS addressing without multiplies (the 64-bit voxel -> byte chain of gload() above is ~12 quarter-rate multiplies per chunk and
  // stage; with two chunks and 16 MFMAs per stage the address code, not the matrix pipe, bounded the kernel at ~30 % MFMA):
  // a chunk keeps its row's SCALED coordinates (x sin, y sin, z sin), its 32-bit byte offset inside the sample and advances all
  // four by uniform per-stage constants plus two carry corrections.  Samples >= 2 GiB keep the 64-bit chain (P.addr64).
  const int gl = P.g_ld * SZ;
  const char* inb = P.in + (int64_t)n * P.Xi * P.Yi * P.Zi * gl;
  const int zrs = P.Zr * P.sin, yrs = P.Yr * P.sin;                         // scaled extents of the row grid
  const int szs = P.sz * P.sin, sys_ = P.sy * P.sin, sxs = P.sx * P.sin;    // scaled 32-row step
  const int o_step = (szs + (sys_ + sxs * P.Yi) * P.Zi) * gl;               // byte step of 32 rows without carries
  const int o_cz = P.sin * gl * (P.Zi - P.Zr), o_cy = P.sin * gl * P.Zi * (P.Yi - P.Yr);   // a z carry / a y carry
  int aoff[AJ];
  if constexpr (GLDS) {       // (this path keeps the rows' SCALED coordinates in cx / cy / cz)
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      cx[j] *= P.sin; cy[j] *= P.sin; cz[j] *= P.sin;
      aoff[j] = (((cx[j] + dx) * P.Yi + cy[j] + dy_) * P.Zi + cz[j] + dz) * gl + ci * SZ;
    }
  }
  const bool dcol_ok = col0 + d_chunk * EPC < P.d_valid;
  int doff = (d_row0 * P.d_ld + d_chunk * EPC) * SZ;                     // byte offset of this thread's first dy chunk in the stage
  const char* dstage = dbase + (int64_t)mstart * P.d_ld * SZ;              // (uniform) first row of the split
  const int d_step = 32 * P.d_ld * SZ, d_jstep = (256 / DCPR) * P.d_ld * SZ;
  auto gload_lds = [&](int s, int buf) {
    const int mb = mstart + s * 32;
    char* a = smem + buf * CF::STAGE + wave * 1024;
    char* d = smem + buf * CF::STAGE + 32 * PA + wave * 1024;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = mb + arow0 + j * ARS;
      const char* src = reinterpret_cast<const char*>(g_wg_zero16);
      const int xi = cx[j] + dx, yi = cy[j] + dy_, zi = cz[j] + dz;
      const bool inside = (unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi;
      if (m < mend) {
        if (kvalid) {
          if (inside) src = P.addr64 ? P.in + ((((nbase + xi) * P.Yi + yi) * P.Zi + zi) * P.g_ld + ci) * SZ : inb + (uint32_t)aoff[j];
        } else if (kones) {
          src = reinterpret_cast<const char*>(g_wg_one16);
        }
      }
      cz[j] += szs;
      const bool carry_z = cz[j] >= zrs;
      cz[j] -= carry_z ? zrs : 0;
      cy[j] += sys_ + (carry_z ? P.sin : 0);
      const bool carry_y = cy[j] >= yrs;
      cy[j] -= carry_y ? yrs : 0;
      cx[j] += sxs + (carry_y ? P.sin : 0);
      aoff[j] += o_step + (carry_z ? o_cz : 0) + (carry_y ? o_cy : 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a + j * (ARS * PA)), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
      const int m = mb + d_row0 + j * (256 / DCPR);
      const char* src = (m < mend && dcol_ok) ? dstage + (uint32_t)(doff + j * d_jstep) : reinterpret_cast<const char*>(g_wg_zero16);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(d + j * 4096), 16, 0, 0);
    }
    doff += d_step;
  };
  // The same stage through range-checked buffer loads (every K block but the one holding the all-ones bias row, samples < 2 GiB):
  // a voxel outside the volume, a row past the split or a column past d_valid is an out-of-range offset (the hardware delivers
  // zeros) -- no zero page, no 64-bit pointer arithmetic, one select per chunk.
  const bool use_buf = GLDS && !P.addr64 && !(kblock * 128 <= ktot && ktot < kblock * 128 + 128);
  const wg_i32x4 rsA = wg_make_rsrc(inb, (uint32_t)((int64_t)P.Xi * P.Yi * P.Zi * gl));
  const wg_i32x4 rsD = wg_make_rsrc(dstage, (uint32_t)((int64_t)(mend > mstart ? mend - mstart : 0) * P.d_ld * SZ));
  int doffb = dcol_ok ? doff : (int)0x80000000;          // (stays out of range: rows * row bytes < 2^31)
  auto gload_lds_buf = [&](int s, int buf) {
    const int mb = mstart + s * 32;
    char* a = smem + buf * CF::STAGE + wave * 1024;
    char* d = smem + buf * CF::STAGE + 32 * PA + wave * 1024;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = mb + arow0 + j * ARS;
      const bool ok = m < mend && kvalid && (unsigned)(cx[j] + dx) < (unsigned)P.Xi && (unsigned)(cy[j] + dy_) < (unsigned)P.Yi &&
                      (unsigned)(cz[j] + dz) < (unsigned)P.Zi;
      const int vo = ok ? aoff[j] : (int)0x80000000;
      cz[j] += szs;
      const bool carry_z = cz[j] >= zrs;
      cz[j] -= carry_z ? zrs : 0;
      cy[j] += sys_ + (carry_z ? P.sin : 0);
      const bool carry_y = cy[j] >= yrs;
      cy[j] -= carry_y ? yrs : 0;
      cx[j] += sxs + (carry_y ? P.sin : 0);
      aoff[j] += o_step + (carry_z ? o_cz : 0) + (carry_y ? o_cy : 0);
      wg_raw_buffer_load_lds(rsA, (wg_lds_u32_ptr)(a + j * (ARS * PA)), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < DJ; ++j) wg_raw_buffer_load_lds(rsD, (wg_lds_u32_ptr)(d + j * 4096), 16, doffb + j * d_jstep, 0, 0, 0);
    doffb += d_step;
  };
  auto sstore = [&](int buf) {
    char* a = smem + buf * CF::STAGE;
    char* d = a + 32 * PA;
#pragma unroll
    for (int j = 0; j < AJ; ++j) *reinterpret_cast<u32x4*>(a + (arow0 + j * ARS) * PA + acc_c * 16) = ra[j];
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
      const int idx = tid + j * 256;
      if (idx < 32 * DCPR) {
        const int r = idx / DCPR, dc = idx - r * DCPR;
        *reinterpret_cast<u32x4*>(d + r * PD + dc * 16) = rd[j];
      }
    }
  };

  f32x4 acc[KT][CT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nst > 0) {
    if constexpr (GLDS) { if (use_buf) gload_lds_buf(0, 0); else gload_lds(0, 0); }
    else { gload(0); sstore(0); }
  }
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    if (s + 1 < nst) {
      if constexpr (GLDS) { if (use_buf) gload_lds_buf(s + 1, buf ^ 1); else gload_lds(s + 1, buf ^ 1); }   // streams into the other buffer while this one feeds the MFMAs
      else gload(s + 1);
    }
    const char* a = smem + buf * CF::STAGE;
    const char* d = a + 32 * PA;
    if constexpr (SZ == 2) {
      // transposed reads: lane supplies row (4*q4 + (r16>>2)) [+16], columns 4*(r16&3).. of its 16-column block
      const int mrow = 4 * q4 + (r16 >> 2), pc = (r16 & 3) * 4;
      bf16x8 af[KT], df[CT];
#pragma unroll
      for (int i = 0; i < KT; ++i) {
        const char* p = GLDS ? a + mrow * PA + ((((wk * KT + i) ^ (mrow & 7)) << 5) + (pc << 1))
                             : a + mrow * PA + (((wk * KT + i) * 16 + pc) << 1);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 16 * PA));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const char* p = GLDS ? d + mrow * PD + ((((wc * CT + j) ^ (BNW == 128 ? (mrow & 7) : ((mrow >> 1) & 3))) << 5) + (pc << 1))
                             : d + mrow * PD + (((wc * CT + j) * 16 + pc) << 1);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 16 * PD));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        df[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], df[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int m = 4 * st + q4;
        float af[KT], df[CT];
#pragma unroll
        for (int i = 0; i < KT; ++i) af[i] = *reinterpret_cast<const float*>(a + m * PA + (((wk * KT + i) * 16 + r16) << 2));
#pragma unroll
        for (int j = 0; j < CT; ++j) df[j] = *reinterpret_cast<const float*>(d + m * PD + (((wc * CT + j) * 16 + r16) << 2));
#pragma unroll
        for (int i = 0; i < KT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], df[j], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (!GLDS) { if (s + 1 < nst) sstore(buf ^ 1); }
    __syncthreads();
  }

  float* slab = P.ws + ((int64_t)zslab * P.kpad_w + kblock * 128) * P.cn_pad + col0;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (wk * KT + i) * 16 + 4 * q4 + e, col = (wc * CT + j) * 16 + r16;
        slab[(int64_t)row * P.cn_pad + col] = acc[i][j][e];
      }
}

template <int NSUB>
__global__ __launch_bounds__(32 * NSUB) void wgrad_reduce_kernel(const float* __restrict__ ws, int nslabs, int kpad_w, int cn_pad, int A, int AS, int T,
                                    int col0, int nb, float* __restrict__ dw, float* __restrict__ db) {
  // block = 32 (k,b) elements x NSUB strided sub-sums over the slabs, combined in fixed order (deterministic);
  // consecutive threads walk b (contiguous in the slab).  NSUB = 32 for the many-slab / few-element case (the stem: 1024
  // slabs x 896 elements ran 28 blocks of 128 serial loads each)
  __shared__ float s_part[NSUB][32];
  const int64_t total = (int64_t)(T * AS + 1) * nb;
  const int64_t slab = (int64_t)kpad_w * cn_pad;
  const int el = threadIdx.x & 31, sub = threadIdx.x >> 5;
  for (int64_t base = (int64_t)blockIdx.x * 32; base < total; base += (int64_t)gridDim.x * 32) {
    const int64_t i = base + el;
    int k = 0, b = 0;
    float s = 0.f;
    if (i < total) {
      k = (int)(i / nb);
      b = (int)(i - (int64_t)k * nb);
      const float* p = ws + (int64_t)k * cn_pad + col0 + b;
      for (int q = sub; q < nslabs; q += NSUB) s += p[q * slab];
    }
    s_part[sub][el] = s;
    __syncthreads();
    if (sub == 0 && i < total) {
      float t8 = 0.f;
#pragma unroll
      for (int q = 0; q < NSUB; ++q) t8 += s_part[q][el];
      if (k == T * AS) {
        if (db != nullptr) db[b] = t8;
      } else {
        const int t = k / AS, a = k - t * AS;
        if (a < A) dw[((int64_t)b * A + a) * T + t] = t8;
      }
    }
    __syncthreads();
  }
}

// Vectorised reduce: 8 lanes x float4 cover up to 128 contiguous bytes of a slab row, 32 strided sub-sums per block (256 threads),
// combined in fixed order; ceil(rows * ceil(nb / 4) / 8) blocks.  (The scalar kernel above moves 128 B per half wave with 8
// sub-sums: 1 - 1.5 TB/s on the 30 - 67 MB of slabs of a persistent LDS-halo weight-gradient kernel, 12 + 6 launches per step of
// the reference's network.)  col0 a multiple of 4; columns nb .. roundup(nb, 4) are pad columns of the slab (read, not written).
// <= 32 registers (modest unrolling): a block then fits beside the persistent one-workgroup-per-CU halo kernels of the main stream
// (2 x 240 of a SIMD's 512 registers, <= 154 KB of LDS) instead of waiting for them to end; same summation order.
template <int LW, int NSUB>      // LW lanes x float4 along a slab row, NSUB strided sub-sums: (8, 32) for many slabs, (32, 8) for few
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void wgrad_reduce4_kernel(const float* __restrict__ ws, int nslabs, int kpad_w, int cn_pad, int A, int AS, int T,
                                                            int col0, int nb, float* __restrict__ dw, float* __restrict__ db) {
  static_assert(LW * NSUB == 256, "256 threads");
  __shared__ f32x4 s_part[NSUB][LW];
  const int nb4 = (nb + 3) >> 2, total4 = (T * AS + 1) * nb4;
  const int64_t slab = (int64_t)kpad_w * cn_pad;
  const int el = threadIdx.x % LW, sub = threadIdx.x / LW;
  const int i4 = blockIdx.x * LW + el;
  const int k = i4 / nb4, b0 = (i4 - k * nb4) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
    const float* p = ws + (int64_t)k * cn_pad + col0 + b0;
#pragma unroll 2
    for (int q = sub; q < nslabs; q += NSUB) s += *reinterpret_cast<const f32x4*>(p + q * slab);
  }
  s_part[sub][el] = s;
  __syncthreads();
  if (sub == 0 && i4 < total4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = 0; q < NSUB; ++q) t += s_part[q][el];
    if (k == T * AS) {
      if (db != nullptr)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b0 + e < nb) db[b0 + e] = t[e];
    } else {
      const int tt = k / AS, a = k - tt * AS;
      if (a < A)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b0 + e < nb) dw[((int64_t)(b0 + e) * A + a) * T + tt] = t[e];
    }
  }
}

// wgrad_reduce4_kernel for a table of passes: block -> job by a scan of the (few) block0 entries, then exactly the arithmetic of
// wgrad_reduce4_kernel<LW, 256 / LW> with LW = job.lanes (8 or 32): the same strided sub-sums, the same combine order.
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) void wgrad_reduce_batch_kernel(const ctseg_reduce_job* __restrict__ jobs, int n_jobs) {
  __shared__ f32x4 s_part[256];
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].block0) ++j;       // (uniform: scalar loads)
  const ctseg_reduce_job J = jobs[j];
  const int LW = J.lanes, NSUB = 256 / LW;
  const int nb4 = (J.nb + 3) >> 2, total4 = (J.T * J.Astride + 1) * nb4;
  const int64_t slab = (int64_t)J.kpad_w * J.cn_pad;
  const int el = threadIdx.x % LW, sub = threadIdx.x / LW;
  const int i4 = ((int)blockIdx.x - J.block0) * LW + el;
  const int k = i4 / nb4, b0 = (i4 - k * nb4) * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < total4) {
    const float* p = J.ws + (int64_t)k * J.cn_pad + J.col0 + b0;
#pragma unroll 2
    for (int q = sub; q < J.nslabs; q += NSUB) s += *reinterpret_cast<const f32x4*>(p + q * slab);
  }
  s_part[sub * LW + el] = s;
  __syncthreads();
  if (sub == 0 && i4 < total4) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int q = 0; q < NSUB; ++q) t += s_part[q * LW + el];
    if (k == J.T * J.Astride) {
      if (J.db != nullptr)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b0 + e < J.nb) J.db[b0 + e] = t[e];
    } else {
      const int tt = k / J.Astride, a = k - tt * J.Astride;
      if (a < J.A)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (b0 + e < J.nb) J.dw[((int64_t)(b0 + e) * J.A + a) * J.T + tt] = t[e];
    }
  }
}

bool wgrad_up_eligible(const ctseg_wgrad_desc* d);
int wgrad_up_slabs(const ctseg_wgrad_desc* d);
void launch_wgrad_up(const ctseg_wgrad_desc* d, hipStream_t st);
bool wgrad_stem_eligible(const ctseg_wgrad_desc* d);
bool wgrad_stem_dyn_ok(const ctseg_wgrad_desc* d);
int wgrad_stem_slabs(const ctseg_wgrad_desc* d);
void launch_wgrad_stem(const ctseg_wgrad_desc* d, hipStream_t st);
bool wgrad_halo_eligible(const ctseg_wgrad_desc* d);
int wgrad_halo_slabs(const ctseg_wgrad_desc* d);
bool wgrad_halo_in_norm_ok(const ctseg_wgrad_desc* d);
void launch_wgrad_halo(const ctseg_wgrad_desc* d, hipStream_t st);
bool wgrad_ring_eligible(const ctseg_wgrad_desc* d);
int wgrad_ring_wgs_per_slab(const ctseg_wgrad_desc* d, int32_t* stage_bytes);
int launch_wgrad_ring(const ctseg_wgrad_desc* d, hipStream_t st);

template <typename T, bool SMALLC> static void launch_wgrad(WgradKArgs& a, hipStream_t st) {
  const int bnw = ctseg_wgrad_tile_cols(a.Cn);
  const int kb = a.kpad_w / 128, cb = a.cn_pad / bnw, zs = a.N * a.splits;
  static const bool remap = !(getenv("CTSEG_WGRAD_XCD") && atoi(getenv("CTSEG_WGRAD_XCD")) == 0);
  dim3 grid((unsigned)kb, (unsigned)cb, (unsigned)zs);
  a.kblocks = a.cblocks = 0;
  a.addr64 = ((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * (int64_t)sizeof(T) >= ((int64_t)1 << 31) - 4096 || getenv("CTSEG_WGRAD_ADDR64") != nullptr) ? 1 : 0;
  if (remap && zs % 8 == 0 && kb * cb > 1) {
    a.kblocks = kb; a.cblocks = cb;
    grid = dim3((unsigned)(kb * cb * zs), 1u, 1u);
  }
  if (bnw == 16) hipLaunchKernelGGL((conv_wgrad_kernel<T, 16, 4, 1, SMALLC>), grid, dim3(256), 0, st, a);
  else if (bnw == 32) hipLaunchKernelGGL((conv_wgrad_kernel<T, 32, 2, 2, SMALLC>), grid, dim3(256), 0, st, a);
  else if (bnw == 64) hipLaunchKernelGGL((conv_wgrad_kernel<T, 64, 2, 2, SMALLC>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<T, 128, 2, 2, SMALLC>), grid, dim3(256), 0, st, a);
}

}  // namespace ctseg

using namespace ctseg;

extern "C" int ctseg_wgrad_tile_cols(int32_t Cn) { return Cn <= 16 ? 16 : Cn <= 32 ? 32 : Cn <= 64 ? 64 : 128; }

// number of fp32 slabs [kpad_w][cn_pad] a ctseg_conv_wgrad call with this descriptor writes into `ws`
extern "C" int ctseg_conv_wgrad_slabs(const ctseg_wgrad_desc* d) {
  if (!desc_ok(d)) return -1;
  if (wgrad_halo_eligible(d)) return wgrad_halo_slabs(d);
  if (wgrad_up_eligible(d)) return wgrad_up_slabs(d);
  if (wgrad_stem_eligible(d)) return wgrad_stem_slabs(d);
  return d->N * d->splits;
}

extern "C" int ctseg_conv_wgrad_wgs_per_slab(const ctseg_wgrad_desc* d, int32_t* per_cu, int32_t* stage_bytes) {
  if (per_cu) *per_cu = 0;
  if (stage_bytes) *stage_bytes = 0;
  if (!desc_ok(d)) return -1;
  if (wgrad_halo_eligible(d) || wgrad_up_eligible(d) || wgrad_stem_eligible(d)) return 0;
  const int SZ = d->dtype == CTSEG_F32 ? 4 : 2, EPC = 16 / SZ;
  const bool smallc = (d->Cg % EPC) != 0 || (d->g_ld % EPC) != 0 || ((uintptr_t)d->in % 16) != 0;
  if (!smallc && wgrad_ring_eligible(d)) {
    if (per_cu) *per_cu = 1;
    return wgrad_ring_wgs_per_slab(d, stage_bytes);
  }
  const int bnw = ctseg_wgrad_tile_cols(d->Cn);
  if (per_cu) *per_cu = 4;
  if (stage_bytes) *stage_bytes = 32 * (128 + bnw) * SZ;
  return (d->kpad_w / 128) * (d->cn_pad / bnw);
}

extern "C" int ctseg_wgrad_in_norm_ok(const ctseg_wgrad_desc* d) { return (desc_ok(d) && d->dtype == CTSEG_BF16 && wgrad_halo_in_norm_ok(d)) ? 1 : 0; }

extern "C" int ctseg_wgrad_dy_norm_ok(const ctseg_wgrad_desc* d) {
  if (!desc_ok(d) || d->dyn_col0 <= 0 || d->in_mean_rstd != nullptr) return 0;
  if (wgrad_halo_eligible(d) || wgrad_up_eligible(d)) return 0;
  return (wgrad_stem_eligible(d) && wgrad_stem_dyn_ok(d)) ? 1 : 0;
}

extern "C" int ctseg_wgrad_narrow_ok(const ctseg_wgrad_desc* d) {
  if (!desc_ok(d)) return 0;
  if (wgrad_halo_eligible(d)) return 1;
  return (d->d_ld != 12 && wgrad_up_eligible(d)) ? 1 : 0;      // the stride-2 transposed-conv kernel takes a 12-wide gathered operand
}

extern "C" int ctseg_conv_wgrad(const ctseg_wgrad_desc* d, void* stream) {
  CTSEG_REQUIRE_DESC(d, "conv_wgrad");
  CTSEG_REQUIRE(d->in && d->dy && d->ws, "conv_wgrad: null pointer");
  CTSEG_REQUIRE(d->dtype == CTSEG_F32 || d->dtype == CTSEG_BF16, "conv_wgrad: bad dtype");
  const int SZ = d->dtype == CTSEG_F32 ? 4 : 2, EPC = 16 / SZ;
  const bool halo = wgrad_halo_eligible(d);     // also moves 12-wide bf16 rows (ctseg_wgrad_narrow_ok)
  if (d->dyn_g != nullptr)
    CTSEG_REQUIRE(ctseg_wgrad_dy_norm_ok(d) == 1 && d->dyn_y && d->dyn_mean_rstd && d->dyn_alpha && d->dyn_sums &&
                      ((uintptr_t)d->dyn_g % 16) == 0 && ((uintptr_t)d->dyn_y % 16) == 0,
                  "conv_wgrad: dyn_* (dY formed on load) is not implemented for this pass (ask ctseg_wgrad_dy_norm_ok)");
  else
    CTSEG_REQUIRE(d->dyn_col0 == 0, "conv_wgrad: dyn_col0 without dyn_g");
  if (d->in_mean_rstd != nullptr)
    CTSEG_REQUIRE(halo && wgrad_halo_in_norm_ok(d) && d->in_alpha != nullptr,
                  "conv_wgrad: in_mean_rstd (normalise the operand on load) is not implemented for this pass (ask ctseg_wgrad_in_norm_ok)");
  CTSEG_REQUIRE((d->d_ld % EPC == 0 || halo) && ((uintptr_t)d->dy % 16) == 0, "conv_wgrad: dy must be 16-byte chunked");
  CTSEG_REQUIRE(halo || wgrad_up_eligible(d) || d->dtype != CTSEG_BF16 || d->g_ld != 12 || d->Cg != 16,
                "conv_wgrad: 12-wide rows need an LDS-halo kernel");
  const bool smallc = (d->Cg % EPC) != 0 || (d->g_ld % EPC) != 0 || ((uintptr_t)d->in % 16) != 0;
  CTSEG_REQUIRE(d->ntaps >= 1 && d->ntaps <= CTSEG_MAX_TAPS && d->splits >= 1, "conv_wgrad: ntaps/splits");
  const int bnw = ctseg_wgrad_tile_cols(d->Cn);
  const int ktot = d->ntaps * d->Cg;
  if (halo) {
    CTSEG_REQUIRE(d->kpad_w >= ktot + 16 && d->cn_pad >= ((d->Cn + 15) / 16) * 16, "conv_wgrad: slab too small for the halo kernel");
    launch_wgrad_halo(d, (hipStream_t)stream);
    CTSEG_LAUNCH_CHECK("conv_wgrad_halo");
    return 0;
  }
  if (wgrad_up_eligible(d)) {
    launch_wgrad_up(d, (hipStream_t)stream);
    CTSEG_LAUNCH_CHECK("conv_wgrad_up");
    return 0;
  }
  if (wgrad_stem_eligible(d)) {
    launch_wgrad_stem(d, (hipStream_t)stream);
    CTSEG_LAUNCH_CHECK("conv_wgrad_stem");
    return 0;
  }
  CTSEG_REQUIRE(d->kpad_w % 128 == 0 && d->kpad_w >= ktot + 1, "conv_wgrad: kpad_w %d (K=%d)", d->kpad_w, ktot);
  CTSEG_REQUIRE(d->cn_pad % bnw == 0 && d->cn_pad >= d->Cn, "conv_wgrad: cn_pad %d", d->cn_pad);
  if (!smallc && wgrad_ring_eligible(d)) {
    const int rc = launch_wgrad_ring(d, (hipStream_t)stream);
    CTSEG_REQUIRE(rc == 0, "conv_wgrad: the ring kernel cannot take this row grid (%d)", rc);
    CTSEG_LAUNCH_CHECK("conv_wgrad_ring");
    return 0;
  }
  WgradKArgs a;
  a.in = (const char*)d->in; a.dy = (const char*)d->dy; a.ws = d->ws;
  a.N = d->N; a.Xi = d->Xi; a.Yi = d->Yi; a.Zi = d->Zi; a.Xr = d->Xr; a.Yr = d->Yr; a.Zr = d->Zr;
  a.Cg = d->Cg; a.Cn = d->Cn; a.g_ld = d->g_ld; a.d_ld = d->d_ld; a.sin = d->sin; a.ntaps = d->ntaps;
  const int64_t rows64 = (int64_t)d->Xr * d->Yr * d->Zr;
  CTSEG_REQUIRE(rows64 < (1ll << 31) - 4096, "conv_wgrad: row grid too large");
  a.rows = (int)rows64; a.splits = d->splits;
  int rps = (int)((rows64 + d->splits - 1) / d->splits);
  rps = ((rps + 31) / 32) * 32;
  a.rows_per_split = rps;
  CTSEG_REQUIRE((int64_t)(rps + 64) * d->d_ld * SZ < ((int64_t)1 << 31), "conv_wgrad: rows per split * row bytes exceed 32-bit offsets (raise splits)");
  a.kpad_w = d->kpad_w; a.cn_pad = d->cn_pad;
  int dv = ((d->Cn + EPC - 1) / EPC) * EPC;
  a.d_valid = dv < d->d_ld ? dv : d->d_ld;
  int step = 32;
  a.sz = step % d->Zr; step /= d->Zr;
  a.sy = step % d->Yr; a.sx = step / d->Yr;
  for (int i = 0; i < CTSEG_MAX_TAPS; ++i) a.taps[i] = i < d->ntaps ? d->taps[i] : 0;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == CTSEG_F32) { if (smallc) launch_wgrad<float, true>(a, st); else launch_wgrad<float, false>(a, st); }
  else { if (smallc) launch_wgrad<BF16, true>(a, st); else launch_wgrad<BF16, false>(a, st); }
  CTSEG_LAUNCH_CHECK("conv_wgrad");
  return 0;
}

extern "C" int ctseg_conv_wgrad_reduce(const float* ws, int32_t nslabs, int32_t kpad_w, int32_t cn_pad, int32_t A, int32_t AS,
                                       int32_t T, int32_t col0, int32_t nb, float* dw, float* db, void* stream) {
  CTSEG_REQUIRE(ws && dw && nslabs >= 1 && A <= AS && T * AS + 1 <= kpad_w && col0 + nb <= cn_pad, "wgrad_reduce: bad arguments");
  const int64_t total = (int64_t)(T * AS + 1) * nb;
  if ((col0 & 3) == 0 && (cn_pad & 3) == 0 && col0 + ((nb + 3) & ~3) <= cn_pad && ((uintptr_t)ws & 15) == 0 &&
      getenv("CTSEG_WGRAD_REDUCE_SCALAR") == nullptr) {
    const int total4 = (T * AS + 1) * ((nb + 3) / 4);
    if (nslabs >= 64)
      hipLaunchKernelGGL((wgrad_reduce4_kernel<8, 32>), dim3((unsigned)((total4 + 7) / 8)), dim3(256), 0, (hipStream_t)stream, ws, nslabs, kpad_w,
                         cn_pad, A, AS, T, col0, nb, dw, db);
    else
      hipLaunchKernelGGL((wgrad_reduce4_kernel<32, 8>), dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, (hipStream_t)stream, ws, nslabs, kpad_w,
                         cn_pad, A, AS, T, col0, nb, dw, db);
    CTSEG_LAUNCH_CHECK("wgrad_reduce4");
    return 0;
  }
  int blocks = (int)((total + 31) / 32);
  if (blocks > 8192) blocks = 8192;
  if (nslabs >= 256 && blocks <= 1024)
    hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, ws, nslabs, kpad_w, cn_pad, A, AS, T,
                       col0, nb, dw, db);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, nslabs, kpad_w, cn_pad, A, AS, T,
                       col0, nb, dw, db);
  CTSEG_LAUNCH_CHECK("wgrad_reduce");
  return 0;
}

extern "C" int ctseg_conv_wgrad_reduce_batch_ok(const float* ws, int32_t cn_pad, int32_t col0, int32_t nb) {
  return ((col0 & 3) == 0 && (cn_pad & 3) == 0 && col0 + ((nb + 3) & ~3) <= cn_pad && ((uintptr_t)ws & 15) == 0 &&
          getenv("CTSEG_WGRAD_REDUCE_SCALAR") == nullptr) ? 1 : 0;
}

extern "C" int ctseg_conv_wgrad_reduce_batch(const ctseg_reduce_job* jobs, int32_t n_jobs, int32_t total_blocks, void* stream) {
  CTSEG_REQUIRE(jobs && n_jobs >= 1 && total_blocks >= 1, "wgrad_reduce_batch: bad arguments");
  hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, n_jobs);
  CTSEG_LAUNCH_CHECK("wgrad_reduce_batch");
  return 0;
}
