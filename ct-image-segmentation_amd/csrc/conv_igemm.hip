// Implicit-GEMM convolution pass for gfx950 (MI355X): Conv3d / ConvTranspose3d forward and
// input-gradient of the MONAI UNet the reference trains (capstone/volumetric/base_trainer.py:65-72).
//
//   out[voxel(row)][n] = bias[n] + add[voxel(row)][n] + sum_{slot,c} in[row*sin + d(slot)][c] * W[n][slot*Cg + c]
//
// One workgroup (4 waves) owns BM consecutive rows of one sample's row grid x BN output channels.
// Per 128-byte K stage: the im2col rows (bounds-checked 16-byte gathers of channels-last voxels)
// and the K-contiguous weight rows are staged global -> registers -> LDS (double buffered, one
// barrier per stage), XOR-swizzled so that every ds_read_b128 of an MFMA fragment is bank-conflict
// free, and consumed by v_mfma_f32_16x16x32_bf16 (bf16 storage) or v_mfma_f32_16x16x4_f32 (fp32
// storage: bit-for-bit an fmaf chain, the parity mode).  The weight tile is the FIRST MFMA operand,
// so each lane ends up with 4 consecutive output channels of one voxel -> channels-last epilogue.
// Epilogue: + bias, per-(tile, channel) sum / sum-of-squares partials for InstanceNorm, transpose
// through LDS, optional addend (residual / gradient accumulation), 16-byte coalesced stores.
#include <utility>

#include "conv_common.h"
#include <type_traits>

namespace ctseg {

// CSZ = largest output element the epilogue may have to stage (4 unless the configuration never writes fp32)
template <int BM, int BN, int CSZ = 4> struct ConvSmem {
  static constexpr int BKB = 128;
  static constexpr int STAGE = (BM + BN) * BKB;
  static constexpr int CROW = BN * CSZ + 16;  // epilogue row pitch
  static constexpr int MAIN = (2 * STAGE > BM * CROW) ? 2 * STAGE : BM * CROW;
  static constexpr int STATS = 4 * 2 * BN * 4;
  static constexpr int ROWTAB = BM * 8;
  static constexpr int TOTAL = MAIN + STATS + ROWTAB + 64 * 4;
};

// 16 zero bytes in device memory: the source of padded (out-of-tensor) operand chunks for direct-to-LDS loads
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

template <typename T, int BM, int BN, int WGM, int WGN, bool SMALLC, bool BST = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_kernel(const ConvKArgs P) {
  constexpr int NTHR = 64 * WGM * WGN;     // 4 waves, or 8 for the 192x256 bf16 tile (never fp32 output: CSZ = 2)
  constexpr int SZ = TT<T>::SZ, EPC = TT<T>::EPC;
  constexpr int BKB = 128, BK = BKB / SZ, CH = BKB / 16, RPR = NTHR / CH;
  constexpr int AR = BM / RPR;
  constexpr int BR = (BN + RPR - 1) / RPR;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MT = WTM / 16, NT = WTN / 16;
  using SM = ConvSmem<BM, BN, (NTHR == 512 ? TT<T>::SZ : 4)>;
  static_assert((WGM * WGN == 4 || WGM * WGN == 8) && MT >= 1 && NT >= 1 && BM % RPR == 0, "4 or 8 waves per workgroup");

  __shared__ __attribute__((aligned(16))) char smem[SM::TOTAL];
  char* const sA0 = smem;                    // [2][BM][128] then [2][BN][128]
  char* const sB0 = smem + 2 * BM * BKB;
  float* const sStats = reinterpret_cast<float*>(smem + SM::MAIN);
  int* const sRow = reinterpret_cast<int*>(smem + SM::MAIN + SM::STATS);  // [BM][2]
  int* const sTap = sRow + BM * 2;                                        // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, q4 = lane >> 4;
  int bx = blockIdx.x;
  if (P.xcd_order) {
    bx = xcd_tile(bx, P.tiles * P.N);
    if (bx < 0) return;
  }
  const int tile = bx % P.tiles, n = bx / P.tiles;
  const int col0 = blockIdx.y * BN;
  const ctseg_conv_class& K = P.cls[blockIdx.z];
  const int ntaps = K.ntaps, kpad = K.kpad;

  // ---- per-tile row table: gathered base coordinates of each GEMM row -------------------------
  for (int r = tid; r < BM; r += NTHR) {
    int ri = tile * BM + r;
    int xy = 0, z = -(1 << 24);
    if (ri < P.rows) {
      int zr = ri % P.Zr, t = ri / P.Zr;
      int yr = t % P.Yr, xr = t / P.Yr;
      xy = xr | (yr << 16);
      z = zr;
    }
    sRow[2 * r] = xy;
    sRow[2 * r + 1] = z;
  }
  if (tid < 32) {
    const int tp = (tid < ntaps) ? K.taps[tid] : 0;
    sTap[tid] = tp;
    // byte delta of the tap inside the gathered tensor (offsets are in [-1,1] per axis, checked on the host)
    sTap[32 + tid] = (((int)(int8_t)(tp & 0xff) * P.Yi + (int)(int8_t)((tp >> 8) & 0xff)) * P.Zi + (int)(int8_t)((tp >> 16) & 0xff)) *
                     P.g_ld * SZ;
  }
  __syncthreads();

  constexpr bool GLDS = !SMALLC;   // global_load_lds: no VGPR staging, no ds_write pass (element-wise stem gather excepted)
  const int q8s = tid % CH, r0 = tid / CH;
  // with direct-to-LDS loads the LDS image is lane-linear (slot = tid % 8), so the XOR swizzle is applied to WHICH K chunk a
  // lane fetches instead (rule: linear destination + swizzled source + swizzled read); RPR is a multiple of 16 -> same for all j
  const int q8 = GLDS ? (q8s ^ ((r0 >> 1) & 7)) : q8s;
  int rxy[AR], rz[AR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    int xy = sRow[2 * (r0 + j * RPR)], z = sRow[2 * (r0 + j * RPR) + 1];
    rxy[j] = ((xy & 0xffff) * P.sin) | (((xy >> 16) * P.sin) << 16);
    rz[j] = (z < 0) ? z : z * P.sin;
  }
  // K position of this thread's 16-byte chunk: slot (tap) and channel, advanced stage by stage
  int slot = (q8 * EPC) / P.Cg, ci = (q8 * EPC) % P.Cg;
  const int64_t nbase = (int64_t)n * P.Xi;
  // per-row constants: byte offset of the row's base voxel, and which of the offsets -1/0/+1 stay inside the tensor
  // per axis (bits 0-2 x, 3-5 y, 6-8 z; 0 for rows past the end) -> a tap is two shifts and an AND per row per stage
  int64_t rowb[AR];
  int rmask[AR], soffA[AR], soffB[BR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int xb = rxy[j] & 0xffff, yb = rxy[j] >> 16, zb = rz[j];
    rowb[j] = (((nbase + xb) * P.Yi + yb) * P.Zi + (zb < 0 ? 0 : zb)) * P.g_ld * SZ;
    int m = 0;
    if (zb >= 0) {
#pragma unroll
      for (int d = -1; d <= 1; ++d) {
        m |= ((unsigned)(xb + d) < (unsigned)P.Xi) << (d + 1);
        m |= ((unsigned)(yb + d) < (unsigned)P.Yi) << (d + 4);
        m |= ((unsigned)(zb + d) < (unsigned)P.Zi) << (d + 7);
      }
    }
    rmask[j] = m;
    const int r = r0 + j * RPR;
    soffA[j] = r * BKB + ((q8s ^ ((r >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int r = r0 + j * RPR;
    soffB[j] = r * BKB + ((q8s ^ ((r >> 1) & 7)) << 4);
  }
  const char* wrow = P.w + (K.w_off + (int64_t)(col0 + r0) * kpad + q8 * EPC) * SZ;

  u32x4 ra0[AR], rb0[BR];
  auto gload = [&](int s, u32x4 (&ra)[AR], u32x4 (&rb)[BR]) {
    if constexpr (!SMALLC) {
      const bool sv = slot < ntaps;
      const int tp = sv ? sTap[slot & 31] : 0;
      const int tb = (sv ? sTap[32 + (slot & 31)] : 0) + ci * SZ;
      const int sx = (int)(int8_t)(tp & 0xff) + 1, sy = (int)(int8_t)((tp >> 8) & 0xff) + 4, sz = (int)(int8_t)((tp >> 16) & 0xff) + 7;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const bool ok = sv && (((rmask[j] >> sx) & (rmask[j] >> sy) & (rmask[j] >> sz) & 1) != 0);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4*>(P.in + rowb[j] + tb);
        ra[j] = v;
      }
      ci += BK;
      while (ci >= P.Cg) { ci -= P.Cg; ++slot; }
    } else {
      // channel count not a multiple of the 16-byte chunk (Cin = 1 stem): element-wise gather
      const int kp0 = s * BK + q8 * EPC;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        uint32_t e[EPC];
#pragma unroll
        for (int t = 0; t < EPC; ++t) {
          int kp = kp0 + t, sl = kp / P.Cg, c = kp - sl * P.Cg;
          uint32_t val = 0u;
          if (sl < ntaps) {
            int tp = sTap[sl & 31];
            int xi = (rxy[j] & 0xffff) + (int)(int8_t)(tp & 0xff), yi = (rxy[j] >> 16) + (int)(int8_t)((tp >> 8) & 0xff),
                zi = rz[j] + (int)(int8_t)((tp >> 16) & 0xff);
            if ((unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi) {
              int64_t vox = ((nbase + xi) * P.Yi + yi) * P.Zi + zi;
              const char* p = P.in + (vox * P.g_ld + c) * SZ;
              if constexpr (SZ == 4) val = *reinterpret_cast<const uint32_t*>(p);
              else val = *reinterpret_cast<const unsigned short*>(p);
            }
          }
          e[t] = val;
        }
        u32x4 v;
        if constexpr (SZ == 4) { v[0] = e[0]; v[1] = e[1]; v[2] = e[2]; v[3] = e[3]; }
        else {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = e[2 * t] | (e[2 * t + 1] << 16);
        }
        ra[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (BN >= RPR || r0 + j * RPR < BN)
        v = *reinterpret_cast<const u32x4*>(wrow + ((int64_t)j * RPR * kpad + (int64_t)s * BK) * SZ);
      rb[j] = v;
    }
  };
  auto sstore = [&](int buf, const u32x4 (&ra)[AR], const u32x4 (&rb)[BR]) {
    char* a = sA0 + buf * BM * BKB;
    char* b = sB0 + buf * BN * BKB;
#pragma unroll
    for (int j = 0; j < AR; ++j) *reinterpret_cast<u32x4*>(a + soffA[j]) = ra[j];
#pragma unroll
    for (int j = 0; j < BR; ++j)
      if (BN >= RPR || r0 + j * RPR < BN) *reinterpret_cast<u32x4*>(b + soffB[j]) = rb[j];
  };

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto gload_lds = [&](int s, int buf) {   // GLDS only: each wave instruction fills 8 rows x 128 B = 1 KiB of LDS, lane-linear
    const bool sv = slot < ntaps;
    const int tp = sv ? sTap[slot & 31] : 0;
    const int tb = (sv ? sTap[32 + (slot & 31)] : 0) + ci * SZ;
    const int sx = (int)(int8_t)(tp & 0xff) + 1, sy = (int)(int8_t)((tp >> 8) & 0xff) + 4, sz = (int)(int8_t)((tp >> 16) & 0xff) + 7;
    char* a = sA0 + buf * BM * BKB + (wave * 8) * BKB;
    char* b = sB0 + buf * BN * BKB + (wave * 8) * BKB;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const bool ok = sv && (((rmask[j] >> sx) & (rmask[j] >> sy) & (rmask[j] >> sz) & 1) != 0);
      const char* src = ok ? (P.in + rowb[j] + tb) : reinterpret_cast<const char*>(g_zero16);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a + j * RPR * BKB), 16, 0, 0);
    }
    ci += BK;
    while (ci >= P.Cg) { ci -= P.Cg; ++slot; }
#pragma unroll
    for (int j = 0; j < BR; ++j)
      if (BN >= RPR || wave * 8 + j * RPR < BN)   // wave-uniform: a 16-column tile only has rows for the first two waves
        __builtin_amdgcn_global_load_lds((gptr_t)(wrow + ((int64_t)j * RPR * kpad + (int64_t)s * BK) * SZ), (lptr_t)(b + j * RPR * BKB), 16, 0, 0);
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nst = kpad / BK;
  const int swz = (r16 >> 1) & 7;
  auto compute = [&](int buf) {
    const char* a = sA0 + buf * BM * BKB + (wm * WTM + r16) * BKB;
    const char* b = sB0 + buf * BN * BKB + (wn * WTN + r16) * BKB;
#pragma unroll
    for (int c = 0; c < BKB / 64; ++c) {
      const int off = ((4 * c + q4) ^ swz) << 4;
      u32x4 xf[MT], wf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const u32x4*>(a + i * 16 * BKB + off);
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(b + j * 16 * BKB + off);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) mma16<T>(acc[j][i], wf[j], xf[i]);
    }
  };
  if constexpr (GLDS) {
    // stage s+1 streams straight into the other LDS buffer while stage s computes;
    // __syncthreads() waits for the DMA (hipcc emits vmcnt(0) for pending LDS-DMA) and for every wave's reads of `buf`
    gload_lds(0, 0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
      const int buf = s & 1;
      if (s + 1 < nst) gload_lds(s + 1, buf ^ 1);
      compute(buf);
      __syncthreads();
    }
  } else {
    // 4-wave tiles run 2+ workgroups per CU; there a second register set measured SLOWER (0.295 -> 0.362 ms on the
    // 256->256 layer with the 128x128 tile): one stage of register prefetch
    gload(0, ra0, rb0);
    sstore(0, ra0, rb0);
    __syncthreads();
    for (int s = 0; s < nst; ++s) {
      const int buf = s & 1;
      if (s + 1 < nst) gload(s + 1, ra0, rb0);
      compute(buf);
      if (s + 1 < nst) sstore(buf ^ 1, ra0, rb0);
      __syncthreads();
    }
  }

  conv_epilogue<T, BM, BN, WGM, WGN, BST>(P, K, smem, sStats, sRow, acc, n, tile, (int)blockIdx.z, col0);
}

template <typename T, int BM, int BN, int WGM, int WGN>
static int launch_cfg(const ConvKArgs& a, bool smallc, int nclass, hipStream_t st) {
  const int gx = a.xcd_order ? 8 * ((a.tiles * a.N + 7) / 8) : a.tiles * a.N;
  dim3 grid((unsigned)gx, (unsigned)((a.Cn + BN - 1) / BN), (unsigned)nclass);
  if constexpr (std::is_same<T, BF16>::value) {       // backward statistics (training storage, 16-byte chunked operands: host-checked)
    if (a.bst.part != nullptr) {
      hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, false, true>), grid, dim3(64 * WGM * WGN), 0, st, a);
      return 0;
    }
  }
  if (smallc) hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, true>), grid, dim3(64 * WGM * WGN), 0, st, a);
  else hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WGM, WGN, false>), grid, dim3(64 * WGM * WGN), 0, st, a);
  return 0;
}

// rows of the row grid one workgroup tile covers.  Cn > 128 in bf16 (never fp32 output there): 192 x 256 tile, 8 waves —
// the 128 x 128 tile is bound by the L2 -> LDS operand traffic per CU (64 FLOP/B); 192 x 256 moves 110 FLOP/B, and the
// M of every layer of the reference's volumes (48 / 24 / 12 / 6 deep) is a multiple of 192.
// 65..128 columns take the 192 x 128 tile of the ring-pipelined kernel where it is eligible (Cg a multiple of 32 ...): 256
// tiles for the 49152-row layers instead of 384 tiles of 128 rows (1.5 rounds over the 256 CUs).
static int tile_rows_for(const ConvKArgs& a, int dtype, bool smallc, int nclass) {
  if (is16(dtype) && !a.out_f32 && !smallc) {
    if (a.Cn > 128) return 192;
    // (not for 32 gathered channels: those passes are operand-traffic bound with short K loops; measured 0.21 -> 0.24 ms)
    if (a.Cn > 64 && a.Cg >= 64 && conv_ring_eligible(a, dtype, nclass)) return 192;
  }
  return a.Cn <= 32 ? 256 : 128;
}

// partial rows per sample the pass taking this geometry fills for ConvKArgs::bst (the same kernel selection as the launch); 0: it cannot
static int bst_slots_for(const ConvKArgs& a, int dtype, int nclass, bool smallc) {
  if (dtype != CTSEG_BF16 || a.out_f32 || a.bst.C <= 0) return 0;      // (training storage: bf16; fp32 keeps its pinned summation order)
  if (conv_halo_eligible(a, dtype, nclass)) return conv_halo_x_eligible(a, dtype, nclass) ? conv_halo_x_bst_slots(a) : 0;
  if (conv_up_eligible(a, dtype, nclass) || conv_stem_eligible(a, dtype, nclass)) return 0;
  if (conv_halo_sw_eligible(a, dtype, nclass)) return conv_halo_sw_bst_slots(a, nclass);
  if (conv_down_halo_eligible(a, dtype, nclass)) return conv_down_halo_bst_slots(a);
  if (conv_down_r_eligible(a, dtype, nclass)) return conv_down_r_bst_slots(a);
  if (conv_up8_eligible(a, dtype, nclass)) return 0;
  { const char* e = getenv("CTSEG_BST_GENERIC"); if (e != nullptr && e[0] == '0') return 0; }   // (A/B switch)
  // generic / ring kernels (conv_epilogue): one partial row per (class, tile); whole 16-byte chunks of y beside those of the output
  if (smallc || a.out2 != nullptr || a.o_ld % 8 != 0 || a.Cn_store % 8 != 0 || a.bst.col0 % 8 != 0 || a.bst.y_ld % 8 != 0 ||
      ((uintptr_t)a.bst.y % 16) != 0 || a.bst.col0 + a.bst.C > a.Cn_store)
    return 0;
  const int bm = tile_rows_for(a, dtype, smallc, nclass);
  // (not the 128 x 64 / 256 x 32 / 256 x 16 tiles: the sums cost them a workgroup per CU — 108 -> 180 registers — which is what the
  // reduce pass they would remove costs: 0.121 vs 0.092 + 0.024 ms on the 8-class 256 -> 64 pass)
  if (bm != 192 && a.Cn <= 64) return 0;
  return ((a.rows + bm - 1) / bm) * nclass;
}

template <typename T> static int launch_dtype(ConvKArgs& a, bool smallc, int nclass, hipStream_t st) {
  const int bm = tile_rows_for(a, TT<T>::DT, smallc, nclass);
  a.tiles = (a.rows + bm - 1) / bm;
  // XCD-contiguous tile ranges: measured neutral for single-class passes (their halo re-reads already hit L2 / Infinity Cache),
  // 7-9 % on the 8-class passes (384->64 and 256->64), where every class re-gathers the same input tile
  a.xcd_order = nclass > 1 ? 1 : 0;
  // classes ride on grid.z, dispatched in index order: longest K loop (most taps) first, so the tail of the launch is made of the
  // 1-tap classes' short workgroups instead of the 8-tap class's long ones
  if (nclass > 1 && getenv("CTSEG_CLASS_ORDER_KEEP") == nullptr)
    for (int i = 1; i < nclass; ++i)
      for (int j = i; j > 0 && a.cls[j].ntaps > a.cls[j - 1].ntaps; --j) std::swap(a.cls[j], a.cls[j - 1]);
  if constexpr (TT<T>::SZ == 2) {
    if (bm == 192) {
      if (conv_ring_eligible(a, TT<T>::DT, nclass)) { launch_conv_ring(a, nclass, st); return 0; }
      return launch_cfg<T, 192, 256, 2, 4>(a, false, nclass, st);
    }
  }
  if (a.Cn <= 16) return launch_cfg<T, 256, 16, 4, 1>(a, smallc, nclass, st);
  if (a.Cn <= 32) return launch_cfg<T, 256, 32, 4, 1>(a, smallc, nclass, st);
  if (a.Cn <= 64) return launch_cfg<T, 128, 64, 2, 2>(a, smallc, nclass, st);
  return launch_cfg<T, 128, 128, 2, 2>(a, smallc, nclass, st);
}

}  // namespace ctseg

using namespace ctseg;

extern "C" int ctseg_conv_tile_rows(int32_t Cn) { return Cn <= 32 ? 256 : 128; }
extern "C" int ctseg_conv_tile_cols(int32_t Cn) { return Cn <= 16 ? 16 : Cn <= 32 ? 32 : Cn <= 64 ? 64 : 128; }

extern "C" int ctseg_conv_igemm(const ctseg_conv_desc* d, void* stream) {
  CTSEG_REQUIRE_DESC(d, "conv_igemm");
  CTSEG_REQUIRE(d->in && d->w && d->out, "conv_igemm: null pointer");
  CTSEG_REQUIRE(d->dtype == CTSEG_F32 || is16(d->dtype), "conv_igemm: bad dtype %d", d->dtype);
  const int SZ = d->dtype == CTSEG_F32 ? 4 : 2, EPC = 16 / SZ, BK = 128 / SZ;
  const int OSZ = d->out_f32 ? 4 : SZ, EPO = 16 / OSZ;
  CTSEG_REQUIRE(d->nclass >= 1 && d->nclass <= CTSEG_MAX_CLASSES, "conv_igemm: nclass %d", d->nclass);
  CTSEG_REQUIRE(d->N > 0 && d->Cg > 0 && d->Cn > 0 && d->Xr > 0 && d->Yr > 0 && d->Zr > 0, "conv_igemm: empty dims");
  CTSEG_REQUIRE(d->Xr < 65536 && d->Yr < 32768 && d->sin >= 1 && d->sin <= 2 && d->sout >= 1 && d->sout <= 2,
                "conv_igemm: grid/stride out of range");
  CTSEG_REQUIRE((int64_t)d->Xr * d->sin < 65536 && (int64_t)d->Yr * d->sin < 32768, "conv_igemm: coordinates overflow 16 bits");
  // "narrow" bf16 rows: 12 elements (24 bytes) for tensors of 9..12 channels (the 10 classes of the reference).  Only the
  // LDS-halo passes move them (8-byte pieces); everything else wants 16-byte chunked rows (ctseg_conv_narrow_ok tells).
  const bool n_out = SZ == 2 && !d->out_f32 && (d->o_ld % EPO != 0 || d->Cn_store % EPO != 0);
  const bool n_in = SZ == 2 && d->g_ld == 12 && d->Cg == 16;
  const bool n_add = SZ == 2 && d->add != nullptr && !d->add_f32 && d->add_ld % EPO != 0;
  const int EPOv = n_out ? 4 : EPO;
  CTSEG_REQUIRE(d->Cn_store >= d->Cn && d->Cn_store % EPOv == 0 && (d->out2 ? d->out2_col0 : d->Cn_store) <= d->o_ld,
                "conv_igemm: Cn_store %d", d->Cn_store);
  CTSEG_REQUIRE(d->o_ld % EPOv == 0 && ((uintptr_t)d->out % 16) == 0, "conv_igemm: out not 16-byte chunked");
  const bool smallc = (d->Cg % EPC) != 0 || (d->g_ld % EPC) != 0 || ((uintptr_t)d->in % 16) != 0;
  CTSEG_REQUIRE(d->g_ld >= d->Cg || n_in, "conv_igemm: g_ld < Cg");
  if (d->add) {
    const int ASZ = d->add_f32 ? 4 : SZ;
    const int EPA = n_add ? 4 : EPO;
    CTSEG_REQUIRE((d->add_ld * ASZ) % (EPA * ASZ) == 0 && ((uintptr_t)d->add % (EPO * ASZ)) == 0 && d->add_ld >= d->Cn_store,
                  "conv_igemm: addend layout");
  }
  CTSEG_REQUIRE((int64_t)d->Xr * d->Yr * d->Zr < (1ll << 31), "conv_igemm: row grid too large");
  int mox = 0, moy = 0, moz = 0;
  for (int c = 0; c < d->nclass; ++c) {
    const ctseg_conv_class& k = d->cls[c];
    mox = k.ox > mox ? k.ox : mox; moy = k.oy > moy ? k.oy : moy; moz = k.oz > moz ? k.oz : moz;
    CTSEG_REQUIRE(k.ntaps >= 1 && k.ntaps <= CTSEG_MAX_TAPS, "conv_igemm: class %d ntaps %d", c, k.ntaps);
    CTSEG_REQUIRE(k.kpad % BK == 0 && k.kpad >= k.ntaps * d->Cg, "conv_igemm: class %d kpad %d", c, k.kpad);
    CTSEG_REQUIRE(((k.w_off * SZ) % 16) == 0, "conv_igemm: class %d weight offset unaligned", c);
    for (int j = 0; j < k.ntaps; ++j)
      for (int sh = 0; sh < 24; sh += 8) {
        const int dd = (int)(int8_t)((k.taps[j] >> sh) & 0xff);
        CTSEG_REQUIRE(dd >= -1 && dd <= 1, "conv_igemm: class %d tap %d offset %d outside [-1,1]", c, j, dd);
      }
    CTSEG_REQUIRE(k.ox >= 0 && k.oy >= 0 && k.oz >= 0 && k.ox < d->sout && k.oy < d->sout && k.oz < d->sout,
                  "conv_igemm: class %d output parity", c);
  }
  CTSEG_REQUIRE((int64_t)(d->Xr - 1) * d->sout + mox < d->Xo && (int64_t)(d->Yr - 1) * d->sout + moy < d->Yo &&
                    (int64_t)(d->Zr - 1) * d->sout + moz < d->Zo,
                "conv_igemm: row grid * sout exceeds written dims");
  ConvKArgs a;
  a.in = (const char*)d->in; a.w = (const char*)d->w; a.bias = d->bias; a.out = (char*)d->out;
  a.add = (const char*)d->add; a.stats = d->stats;
  a.N = d->N; a.Xi = d->Xi; a.Yi = d->Yi; a.Zi = d->Zi; a.Xr = d->Xr; a.Yr = d->Yr; a.Zr = d->Zr;
  a.Xo = d->Xo; a.Yo = d->Yo; a.Zo = d->Zo;
  a.Cg = d->Cg; a.Cn = d->Cn; a.Cn_store = d->Cn_store; a.g_ld = d->g_ld; a.o_ld = d->o_ld; a.add_ld = d->add_ld;
  a.sin = d->sin; a.sout = d->sout; a.rows = d->Xr * d->Yr * d->Zr; a.tiles = 0;
  a.out_f32 = d->out_f32; a.add_f32 = d->add_f32;
  a.stats_ld = d->stats_ld; a.stats_tiles = d->stats_tiles; a.stats_tile0 = d->stats_tile0;
  for (int c = 0; c < CTSEG_MAX_CLASSES; ++c) a.cls[c] = d->cls[c < d->nclass ? c : 0];
  a.out2 = (char*)d->out2; a.out2_col0 = d->out2_col0; a.o2_ld = d->o2_ld; a.xcd_order = 0;
  a.dtype = d->dtype;
  a.in_mr = d->in_mean_rstd; a.in_alpha = d->in_alpha; a.in_C = d->in_norm_C;
  fill_bst(d, a);
  if (d->bst_partials != nullptr) {
    const int slots = bst_slots_for(a, d->dtype, d->nclass, smallc);
    CTSEG_REQUIRE(slots > 0, "conv_igemm: bst_* (backward statistics in the epilogue) is not implemented for this pass (ask ctseg_conv_bwd_stats_slots)");
    CTSEG_REQUIRE(d->bst_y && d->bst_mean_rstd && d->bst_alpha && d->bst_P >= slots && d->bst_ld >= d->bst_C && d->bst_C > 0 &&
                      ((uintptr_t)d->bst_y % 8) == 0,
                  "conv_igemm: bst_* layout (need bst_P >= %d partial rows per sample)", slots);
  }
  const bool halo = conv_halo_eligible(a, d->dtype, d->nclass);
  if (d->in_mean_rstd != nullptr)
    CTSEG_REQUIRE(halo && conv_halo_x_in_norm_ok(a, d->dtype, d->nclass) && d->in_alpha != nullptr,
                  "conv_igemm: in_mean_rstd (normalise the operand on load) is not implemented for this pass (ask ctseg_conv_in_norm_ok)");
  if (halo && d->stats && conv_halo_x_eligible(a, d->dtype, d->nclass))
    CTSEG_REQUIRE(conv_halo_x_stats_ok(a), "conv_igemm: InstanceNorm partials with an addend / fp32 output / input-gradient taps "
                                           "are not implemented on the x-column halo pass");
  const bool up = !halo && conv_up_eligible(a, d->dtype, d->nclass);
  const bool stem = !halo && !up && conv_stem_eligible(a, d->dtype, d->nclass);
  const bool sw = !halo && !up && !stem && conv_halo_sw_eligible(a, d->dtype, d->nclass);
  const bool down = !halo && !up && !stem && !sw && conv_down_halo_eligible(a, d->dtype, d->nclass);
  const bool downr = !halo && !up && !stem && !sw && !down && conv_down_r_eligible(a, d->dtype, d->nclass);
  const bool up8 = !halo && !up && !stem && !sw && !down && !downr && conv_up8_eligible(a, d->dtype, d->nclass);
  if (d->stats && up8) {
    CTSEG_REQUIRE(d->stats_tile0 + conv_up8_slots(a) <= d->stats_tiles && d->stats_ld >= d->Cn,
                  "conv_igemm: stats partial layout (many-channel 8-class pass)");
  } else if (d->stats && downr) {
    CTSEG_REQUIRE(d->stats_tile0 + conv_down_r_slots(a) <= d->stats_tiles && d->stats_ld >= d->Cn,
                  "conv_igemm: stats partial layout (stride-2 register-weight pass)");
  } else if (d->stats && down) {
    CTSEG_REQUIRE(d->stats_tile0 + conv_down_halo_slots(a) <= d->stats_tiles && d->stats_ld >= d->Cn,
                  "conv_igemm: stats partial layout (stride-2 halo pass)");
  } else if (d->stats && sw) {
    CTSEG_REQUIRE(d->stats_tile0 + conv_halo_sw_slots(a) <= d->stats_tiles && d->stats_ld >= d->Cn,
                  "conv_igemm: stats partial layout (streamed-weight halo pass)");
  } else if (d->stats && stem) {
    CTSEG_REQUIRE(d->stats_tile0 + conv_stem_slots(a) <= d->stats_tiles && d->stats_ld >= ((d->Cn + 15) / 16) * 16,
                  "conv_igemm: stats partial layout (stem pass)");
  } else if (d->stats && up) {
    CTSEG_REQUIRE(d->stats_tile0 + conv_up_slots(a) <= d->stats_tiles && d->stats_ld >= 16, "conv_igemm: stats partial layout (up pass)");
  } else if (d->stats) {
    const int bm = tile_rows_for(a, d->dtype, smallc, d->nclass);
    const int tiles = halo ? conv_halo_slots(a, d->dtype) : (a.rows + bm - 1) / bm;
    const int bn = bm == 192 ? (d->Cn > 128 ? 256 : 128) : ctseg_conv_tile_cols(d->Cn);
    CTSEG_REQUIRE(d->stats_tile0 + tiles * d->nclass <= d->stats_tiles && d->stats_ld >= ((d->Cn + bn - 1) / bn) * bn,
                  "conv_igemm: stats partial layout (need stats_ld >= roundup(Cn, tile cols))");
  }
  if (n_out || n_in || n_add)
    CTSEG_REQUIRE(halo || (up && !n_in) || (down && !n_out && !n_add),
                  "conv_igemm: 12-wide bf16 rows are moved by the LDS-halo passes only (ask ctseg_conv_narrow_ok)");
  if (d->out2 != nullptr)
    CTSEG_REQUIRE((stem || down || downr) && d->add == nullptr && d->out2_col0 > 0 && d->out2_col0 % 4 == 0 && d->out2_col0 < d->Cn_store &&
                      d->o2_ld >= d->Cn_store - d->out2_col0 && d->o2_ld % 4 == 0 && ((uintptr_t)d->out2 % 16) == 0,
                  "conv_igemm: out2 (split output) is not supported for this pass (ask ctseg_conv_split_ok first)");
  hipStream_t st = (hipStream_t)stream;
  if (halo) launch_conv_halo(a, d->dtype, st);
  else if (up) launch_conv_up(a, st);
  else if (stem) launch_conv_stem(a, st);
  else if (sw) launch_conv_halo_sw(a, d->nclass, st);
  else if (down) launch_conv_down_halo(a, st);
  else if (downr) launch_conv_down_r(a, st);
  else if (up8) launch_conv_up8(a, st);
  else if (d->dtype == CTSEG_F32) launch_dtype<float>(a, smallc, d->nclass, st);
  else if (d->dtype == CTSEG_F16) launch_dtype<F16>(a, smallc, d->nclass, st);
  else launch_dtype<BF16>(a, smallc, d->nclass, st);
  CTSEG_LAUNCH_CHECK("conv_igemm");
  return 0;
}

static void fill_args(const ctseg_conv_desc* d, ConvKArgs& a) {
  fill_bst(d, a);
  a.out2 = nullptr; a.out2_col0 = 0; a.o2_ld = 0; a.xcd_order = 0; a.dtype = d->dtype;
  a.in_mr = d->in_mean_rstd; a.in_alpha = d->in_alpha; a.in_C = d->in_norm_C;
  a.w = (const char*)d->w; a.Cn_store = d->Cn_store;
  a.in = (const char*)d->in; a.N = d->N; a.Xi = d->Xi; a.Yi = d->Yi; a.Zi = d->Zi; a.Xr = d->Xr; a.Yr = d->Yr; a.Zr = d->Zr;
  a.Cg = d->Cg; a.Cn = d->Cn; a.g_ld = d->g_ld; a.sin = d->sin; a.sout = d->sout; a.rows = d->Xr * d->Yr * d->Zr;
  // everything a kernel-selection predicate may look at (the same answer at sizing time and at launch time)
  a.out = (char*)d->out; a.add = (const char*)d->add; a.bias = d->bias; a.stats = nullptr;
  a.Xo = d->Xo; a.Yo = d->Yo; a.Zo = d->Zo; a.o_ld = d->o_ld; a.add_ld = d->add_ld; a.out_f32 = d->out_f32; a.add_f32 = d->add_f32;
  a.tiles = 0; a.stats_ld = 0; a.stats_tiles = 0; a.stats_tile0 = 0; a.o2_ld = 0;
  for (int c = 0; c < CTSEG_MAX_CLASSES; ++c) a.cls[c] = d->cls[c < d->nclass ? c : 0];
}

extern "C" int ctseg_conv_split_ok(const ctseg_conv_desc* d) {
  if (!desc_ok(d) || d->nclass != 1 || d->add != nullptr) return 0;
  ConvKArgs a;
  fill_args(d, a);
  a.out_f32 = d->out_f32; a.Xo = d->Xo; a.Yo = d->Yo; a.Zo = d->Zo; a.add = nullptr; a.o_ld = d->o_ld;
  if (conv_halo_eligible(a, d->dtype, d->nclass) || conv_up_eligible(a, d->dtype, d->nclass)) return 0;
  if (conv_stem_eligible(a, d->dtype, d->nclass)) return 1;
  if (conv_halo_sw_eligible(a, d->dtype, d->nclass)) return 0;
  if (conv_down_halo_eligible(a, d->dtype, d->nclass)) return 1;
  return (d->out2_col0 % 16 == 0 && conv_down_r_eligible(a, d->dtype, d->nclass)) ? 1 : 0;
}

extern "C" int ctseg_conv_bwd_stats_slots(const ctseg_conv_desc* d) {
  if (!desc_ok(d) || !is16(d->dtype) || d->nclass < 1 || d->nclass > CTSEG_MAX_CLASSES || d->bst_C <= 0 || d->out_f32) return 0;
  ConvKArgs a;
  fill_args(d, a);
  a.out2 = (char*)d->out2; a.out2_col0 = d->out2_col0; a.o2_ld = d->o2_ld;
  const bool smallq = (d->Cg % 8) != 0 || (d->g_ld % 8) != 0 || ((uintptr_t)d->in % 16) != 0;
  return bst_slots_for(a, d->dtype, d->nclass, smallq);
}

extern "C" int ctseg_conv_in_norm_ok(const ctseg_conv_desc* d) {
  if (!desc_ok(d) || !is16(d->dtype) || d->nclass < 1 || d->nclass > CTSEG_MAX_CLASSES) return 0;
  ConvKArgs a;
  fill_args(d, a);
  return (conv_halo_eligible(a, d->dtype, d->nclass) && conv_halo_x_in_norm_ok(a, d->dtype, d->nclass)) ? 1 : 0;
}

// 1 when this pass may read / write 12-wide bf16 rows (g_ld, o_ld / Cn_store, add_ld of the descriptor): it is taken by
// the resident-weight LDS-halo kernel (any of them narrow), by the stride-2 "up" kernel (narrow output / addend only) or by the
// stride-2 "down" halo kernel (narrow gathered operand only)
extern "C" int ctseg_conv_narrow_ok(const ctseg_conv_desc* d) {
  if (!desc_ok(d) || !is16(d->dtype) || d->nclass < 1 || d->nclass > CTSEG_MAX_CLASSES) return 0;
  ConvKArgs a;
  fill_args(d, a);
  a.out_f32 = d->out_f32; a.Xo = d->Xo; a.Yo = d->Yo; a.Zo = d->Zo; a.add = (const char*)d->add; a.o_ld = d->o_ld;
  if (conv_halo_eligible(a, d->dtype, d->nclass)) return 1;
  if (d->g_ld % 8 != 0) {
    const bool wide_out = d->out_f32 || (d->o_ld % 8 == 0 && d->Cn_store % 8 == 0), wide_add = d->add == nullptr || d->add_f32 || d->add_ld % 8 == 0;
    return (d->g_ld == 12 && wide_out && wide_add && !conv_up_eligible(a, d->dtype, d->nclass) && !conv_stem_eligible(a, d->dtype, d->nclass) &&
            !conv_halo_sw_eligible(a, d->dtype, d->nclass) && conv_down_halo_eligible(a, d->dtype, d->nclass)) ? 1 : 0;
  }
  return conv_up_eligible(a, d->dtype, d->nclass) ? 1 : 0;
}

// tiles per sample (all classes) a pass with this geometry writes InstanceNorm partials for
extern "C" int ctseg_conv_num_tiles(const ctseg_conv_desc* d) {
  if (!desc_ok(d) || d->nclass < 1 || d->nclass > CTSEG_MAX_CLASSES) return -1;
  ConvKArgs a;
  fill_args(d, a);
  if (conv_halo_eligible(a, d->dtype, d->nclass)) return conv_halo_slots(a, d->dtype);
  a.out_f32 = d->out_f32; a.Xo = d->Xo; a.Yo = d->Yo; a.Zo = d->Zo;
  if (conv_up_eligible(a, d->dtype, d->nclass)) return conv_up_slots(a);
  a.add = (const char*)d->add; a.o_ld = d->o_ld;
  if (conv_stem_eligible(a, d->dtype, d->nclass)) return conv_stem_slots(a);
  if (conv_halo_sw_eligible(a, d->dtype, d->nclass)) return conv_halo_sw_slots(a);
  if (conv_down_halo_eligible(a, d->dtype, d->nclass)) return conv_down_halo_slots(a);
  if (conv_down_r_eligible(a, d->dtype, d->nclass)) return conv_down_r_slots(a);
  if (conv_up8_eligible(a, d->dtype, d->nclass)) return conv_up8_slots(a);
  const int SZq = d->dtype == CTSEG_F32 ? 4 : 2, EPCq = 16 / SZq;
  const bool smallq = (d->Cg % EPCq) != 0 || (d->g_ld % EPCq) != 0 || ((uintptr_t)d->in % 16) != 0;
  const int bm = tile_rows_for(a, d->dtype, smallq, d->nclass);
  return ((a.rows + bm - 1) / bm) * d->nclass;
}
