// Stem kernels (bf16): the single-channel 3x3x3 stride-2 convolution at the top of the reference's U-Net
// (Conv3d(1, 32, k3, s2) twice — residual branch and unit0 — fused to 64 columns, capstone/volumetric/base_trainer.py:65-72
// with in_channels = 1) and its weight gradient.  K = 27: the generic kernel's element-wise gather spends ~1000 VALU
// instructions per stage on addresses; here a persistent workgroup stages the 9x17x17 input patch of a 4x8x8 output tile
// in LDS (5 KB), builds each MFMA operand from 8 ds_read_u16, keeps the 27x64 weights in registers, and stores from the
// accumulators.  Both passes are bound by the 128-byte-per-voxel output / dY stream.
//
//   forward : out[v][n] = bias[n] + sum_t in[2v + t - 1] * W[n][t]            (+ InstanceNorm partial sums per workgroup)
//   wgrad   : R[t][b] = sum_v in[2v + t - 1] * dy[v][b],  R[27][b] = sum_v dy[v][b]   -> 4 slabs per workgroup
#include "conv_common.h"

#ifndef STEM_ABL
#define STEM_ABL 0     // timing-only ablation bits (tools/ablate_stem.sh): 1 no MFMAs, 2 no LDS operand reads, 4 no patch loads, 8 no stores, 16 no stats
#endif

namespace ctseg {

constexpr int S_IX = 9, S_IY = 17, S_IZ = 17, S_PZ = 18;          // input patch and its padded z pitch
constexpr int S_INB = (S_IX * S_IY * S_PZ * 2 + 2 + 15) / 16 * 16;   // bytes of the LDS image (+ the one-element shift), 16-byte aligned
// The patch is staged in ALIGNED 4-byte pairs: z runs from 2 z0 - 2 (one element before the first tap's, even) for 18 elements =
// 9 dwords = the padded pitch.  A pair is inside or outside the volume as a whole (Zi is even), so one bounds test and one load
// move two elements: 6 loads per thread and tile instead of 11.  Readers see the image through a pointer advanced by one element.
constexpr int S_DW = S_PZ / 2, S_NDW = S_IX * S_IY * S_DW;          // 1377 dwords
constexpr int S_J = (S_NDW + 255) / 256;                           // 6 staging slots per thread
constexpr int S_SHIFT = 2;                                         // byte offset of patch element (0,0,0) inside the image

__device__ __forceinline__ void stem_patch_voxel(int r16, int& dy, int& z) {
  dy = (0xEF80u >> r16) & 1;
  z = (int)((0x2104765437653210ull >> (4 * r16)) & 7ull);
}

struct StemGeom {
  int tiles, tyn, tzn;
};

// shared by both kernels: per-thread staging constants and the patch loader
struct StemStage {
  int g_off[S_J], g_xyz[S_J], g_lds[S_J];
  __device__ __forceinline__ void init(int tid, int Yi, int Zi) {
#pragma unroll
    for (int j = 0; j < S_J; ++j) {
      const int idx = tid + j * 256;
      const int row = idx / S_DW, dw = idx - row * S_DW, ix = row / S_IY, iy = row - ix * S_IY;
      g_off[j] = ((ix - 1) * Yi + (iy - 1)) * Zi + 2 * dw - 2;
      g_xyz[j] = idx < S_NDW ? (ix | (iy << 8) | ((2 * dw) << 16)) : 0x7f7f7f;
      g_lds[j] = ((ix * S_IY + iy) * S_PZ + 2 * dw) * 2;
    }
  }
  __device__ __forceinline__ void load(const unsigned short* base, int x0, int y0, int z0, int Xi, int Yi, int Zi,
                                       uint32_t (&r)[S_J]) const {
    // base = element pointer of input voxel (2x0, 2y0, 2z0) of the sample
#pragma unroll
    for (int j = 0; j < S_J; ++j) {
      const int xi = 2 * x0 - 1 + (g_xyz[j] & 0xff), yi = 2 * y0 - 1 + ((g_xyz[j] >> 8) & 0xff), zi = 2 * z0 - 2 + (g_xyz[j] >> 16);
      uint32_t v = 0;
      if (!(STEM_ABL & 4) && (unsigned)xi < (unsigned)Xi && (unsigned)yi < (unsigned)Yi && (unsigned)zi < (unsigned)Zi)
        v = *reinterpret_cast<const uint32_t*>(base + g_off[j]);
      r[j] = v;
    }
  }
  __device__ __forceinline__ void store(char* lds, const uint32_t (&r)[S_J]) const {
#pragma unroll
    for (int j = 0; j < S_J; ++j)
      if ((g_xyz[j] & 0xff) != 0x7f) *reinterpret_cast<uint32_t*>(lds + g_lds[j]) = r[j];
  }
};

// LDS element offset of tap t relative to the patch element of output voxel (x,y,z) at tap (0,0,0)
__device__ __forceinline__ int stem_tap_off(int t) {
  const int tx = t / 9, ty = (t / 3) % 3, tz = t % 3;
  return (tx * S_IY + ty) * S_PZ + tz;
}

template <typename H, int NT, bool STATS>     // H = 16-bit storage kind (BF16 / F16)
__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(const ConvKArgs P, int total_tiles, StemGeom G) {
  constexpr int BN = 16 * NT;
  __shared__ __attribute__((aligned(16))) char smem[2 * S_INB + 4 * 2 * BN * 4];
  float* const sStats = reinterpret_cast<float*>(smem + 2 * S_INB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;
  const ctseg_conv_class& K = P.cls[0];

  // weights [rows][kpad = 64]: lane (row r16 of column tile j, quarter q4) holds K elements 8*q4 .. 8*q4+7 (taps; >= 27 are zero)
  u32x4 wf[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
    wf[j] = *reinterpret_cast<const u32x4*>(P.w + (K.w_off + (int64_t)(j * 16 + r16) * K.kpad) * 2 + q4 * 16);
  float bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = j * 16 + 4 * q4 + e;
      bias[j][e] = (P.bias != nullptr && ch < P.Cn) ? P.bias[ch] : 0.f;
    }
  StemStage stg;
  stg.init(tid, P.Yi, P.Zi);
  int pdy, pz;
  stem_patch_voxel(r16, pdy, pz);
  int toff[8], abase[4], ovox[4];
#pragma unroll
  for (int k = 0; k < 8; ++k) toff[k] = (8 * q4 + k < 27) ? stem_tap_off(8 * q4 + k) * 2 : -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    abase[i] = (((2 * wave) * S_IY + 2 * (2 * i + pdy)) * S_PZ + 2 * pz) * 2;
    ovox[i] = (wave * P.Yo + 2 * i + pdy) * P.Zo + pz;
  }
  auto origin = [&](int t, int& n, int& x0, int& y0, int& z0) {
    n = t / G.tiles;
    int r = t - n * G.tiles;
    const int tz = r % G.tzn; r /= G.tzn;
    const int ty = r % G.tyn; const int tx = r / G.tyn;
    x0 = tx * 4; y0 = ty * 8; z0 = tz * 8;
  };
  auto in_base = [&](int n, int x0, int y0, int z0) {
    return reinterpret_cast<const unsigned short*>(P.in) + (((int64_t)n * P.Xi + 2 * x0) * P.Yi + 2 * y0) * P.Zi + 2 * z0;
  };
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 wsum2[NT][2], wsq2[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 2; ++e) { wsum2[j][e] = f32x2{0.f, 0.f}; wsq2[j][e] = f32x2{0.f, 0.f}; }
  int stat_n = -1;
  auto flush_stats = [&](int n) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = wsum2[j][e >> 1][e & 1], b = wsq2[j][e >> 1][e & 1];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          sStats[(wave * 2 + 0) * BN + j * 16 + 4 * q4 + e] = a;
          sStats[(wave * 2 + 1) * BN + j * 16 + 4 * q4 + e] = b;
        }
        wsum2[j][e >> 1][e & 1] = 0.f;
        wsq2[j][e >> 1][e & 1] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid % BN;
      const float a = sStats[(0 * 2 + which) * BN + c] + sStats[(1 * 2 + which) * BN + c] + sStats[(2 * 2 + which) * BN + c] +
                      sStats[(3 * 2 + which) * BN + c];
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      P.stats[(slot_t * 2 + which) * P.stats_ld + c] = a;
    }
    __syncthreads();
  };

  uint32_t rg[S_J];
  int t = blockIdx.x, cur = 0;
  {
    int n, x0, y0, z0;
    if (t < total_tiles) {
      origin(t, n, x0, y0, z0);
      stg.load(in_base(n, x0, y0, z0), x0, y0, z0, P.Xi, P.Yi, P.Zi, rg);
      stg.store(smem, rg);
    }
  }
  __syncthreads();
  for (; t < total_tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    int n, x0, y0, z0;
    if (tn < total_tiles) {
      origin(tn, n, x0, y0, z0);
      stg.load(in_base(n, x0, y0, z0), x0, y0, z0, P.Xi, P.Yi, P.Zi, rg);
    }
    origin(t, n, x0, y0, z0);
    if (STATS && n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = n;
    }
    const char* h = smem + cur * S_INB + S_SHIFT;
    const int64_t vb = (((int64_t)n * P.Xo + x0) * P.Yo + y0) * P.Zo + z0;
    const bool xok = x0 + wave < P.Xr, zok = z0 + pz < P.Zr;
    char* ob = P.out + vb * P.o_ld * 2;
    char* ob2 = P.out2 != nullptr ? P.out2 + vb * P.o2_ld * 2 : nullptr;
    const bool wide = NT % 2 == 0 && P.Cn_store == 16 * NT && (P.o_ld & 7) == 0 &&
                      (P.out2 == nullptr || ((P.out2_col0 & 31) == 0 && (P.o2_ld & 7) == 0)) && P.xcd_order == 0;
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {   // K = 32 is ONE MFMA per 16x16 tile: operand -> MFMA -> store, row tile by row tile
      u32x4 xf;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (STEM_ABL & 2) { xf[k] = (uint32_t)(abase[i] + k + t); continue; }
        const uint32_t lo = toff[2 * k] >= 0 ? *reinterpret_cast<const unsigned short*>(h + abase[i] + toff[2 * k]) : 0u;
        const uint32_t hi = toff[2 * k + 1] >= 0 ? *reinterpret_cast<const unsigned short*>(h + abase[i] + toff[2 * k + 1]) : 0u;
        xf[k] = lo | (hi << 16);
      }
      const bool rv = xok && zok && (y0 + 2 * i + pdy < P.Yr);
      u32x2 o2[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 acc = f32x4{bias[j][0], bias[j][1], bias[j][2], bias[j][3]};     // the bias rides in the accumulator
        if (STEM_ABL & 1) acc = f32x4{__uint_as_float(xf[0] ^ wf[j][0]), __uint_as_float(xf[1]), __uint_as_float(xf[2]), __uint_as_float(xf[3])};
        else mma16<H>(acc, wf[j], xf);
        const float v[4] = {acc[0], acc[1], acc[2], acc[3]};
        if (STATS && !(STEM_ABL & 16) && rv) {      // two-wide fp32 adds / FMAs (v_pk_add_f32, v_pk_fma_f32)
          const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
          wsum2[j][0] += lo; wsum2[j][1] += hi;
          wsq2[j][0] = lo * lo + wsq2[j][0]; wsq2[j][1] = hi * hi + wsq2[j][1];
        }
        o2[j] = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
        if (!wide) {
          const int ch = j * 16 + 4 * q4;
          if (rv && ch < P.Cn_store) {
            char* op = (ob2 != nullptr && ch >= P.out2_col0) ? ob2 + ((int64_t)ovox[i] * P.o2_ld + (ch - P.out2_col0)) * 2
                                                            : ob + ((int64_t)ovox[i] * P.o_ld + ch) * 2;
            *reinterpret_cast<u32x2*>(op) = o2[j];
          }
        }
      }
      if constexpr (NT % 2 == 0) {
        if (wide) {
          // 16-byte stores: v_permlane16_swap pairs two 16-column blocks so that a lane holds 8 consecutive channels (chunk
          // {0, 2, 1, 3}[q4] of the pair's 32 channels): one store instruction writes whole 64-byte rows instead of half rows
#pragma unroll
          for (int jp = 0; jp < NT / 2; ++jp) {
            const auto s0 = __builtin_amdgcn_permlane16_swap(o2[2 * jp][0], o2[2 * jp + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(o2[2 * jp][1], o2[2 * jp + 1][1], false, false);
            const int ch = jp * 32 + ((q4 & 1) * 2 + (q4 >> 1)) * 8;
            if ((STEM_ABL & 8) ? (rv && s0[0] == 0x12345678u) : rv) {
              char* op = (ob2 != nullptr && ch >= P.out2_col0) ? ob2 + ((int64_t)ovox[i] * P.o2_ld + (ch - P.out2_col0)) * 2
                                                              : ob + ((int64_t)ovox[i] * P.o_ld + ch) * 2;
              *reinterpret_cast<u32x4*>(op) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
          }
        }
      }
    }
    if (tn < total_tiles) stg.store(smem + (cur ^ 1) * S_INB, rg);
    __syncthreads();
    cur ^= 1;
  }
  if (STATS && stat_n >= 0) flush_stats(stat_n);
}

bool conv_stem_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (!is16(dtype) || nclass != 1 || a.Cg != 1 || a.g_ld != 1 || a.sin != 2 || a.sout != 1 || a.out_f32 || a.add) return false;
  if (((uintptr_t)a.in % 4) != 0) return false;       // the patch is staged in aligned 4-byte pairs
  if (a.cls[0].ntaps != 27 || a.cls[0].kpad != 64 || a.Cn > 64 || a.Cn % 16 != 0 || a.Zr < 4) return false;
  if (a.Xi != 2 * a.Xr || a.Yi != 2 * a.Yr || a.Zi != 2 * a.Zr) return false;
  if ((int64_t)a.Xi * a.Yi * a.Zi >= (1ll << 31) || (int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * 2 >= (1ll << 31)) return false;
  for (int j = 0; j < 27; ++j) {   // taps must be the plain conv's, in torch order: offset = t - 1 per axis
    const int tp = a.cls[0].taps[j];
    const int ex = j / 9 - 1, ey = (j / 3) % 3 - 1, ez = j % 3 - 1;
    if ((int)(int8_t)(tp & 0xff) != ex || (int)(int8_t)((tp >> 8) & 0xff) != ey || (int)(int8_t)((tp >> 16) & 0xff) != ez) return false;
  }
  return true;
}

static int stem_tiles(int Xr, int Yr, int Zr) { return ((Xr + 3) / 4) * ((Yr + 7) / 8) * ((Zr + 7) / 8); }
static int stem_grid(int total) { return persistent_grid(4 * CTSEG_NUM_CU, total); }
int conv_stem_slots(const ConvKArgs& a) { return stem_grid(stem_tiles(a.Xr, a.Yr, a.Zr) * a.N); }

void launch_conv_stem(ConvKArgs& a, hipStream_t st) {
  a.xcd_order = getenv("CTSEG_STEM_STORE_B64") != nullptr;      // timing switch: 8-byte half-row stores (field unused otherwise here)
  StemGeom g;
  g.tiles = stem_tiles(a.Xr, a.Yr, a.Zr); g.tyn = (a.Yr + 7) / 8; g.tzn = (a.Zr + 7) / 8;
  a.tiles = g.tiles;
  const int total = g.tiles * a.N, grid = stem_grid(total);
  const int nt = (a.Cn + 15) / 16;
#define CTSEG_STEM(NTV)                                                                                               \
  do {                                                                                                              \
    if (a.dtype == CTSEG_F16) {                                                                                       \
      if (a.stats != nullptr) hipLaunchKernelGGL((conv_stem_fwd_kernel<F16, NTV, true>), dim3(grid), dim3(256), 0, st, a, total, g); \
      else hipLaunchKernelGGL((conv_stem_fwd_kernel<F16, NTV, false>), dim3(grid), dim3(256), 0, st, a, total, g);    \
    } else if (a.stats != nullptr) hipLaunchKernelGGL((conv_stem_fwd_kernel<BF16, NTV, true>), dim3(grid), dim3(256), 0, st, a, total, g); \
    else hipLaunchKernelGGL((conv_stem_fwd_kernel<BF16, NTV, false>), dim3(grid), dim3(256), 0, st, a, total, g);     \
  } while (0)
  if (nt == 1) CTSEG_STEM(1);
  else if (nt == 2) CTSEG_STEM(2);
  else if (nt == 3) CTSEG_STEM(3);
  else CTSEG_STEM(4);
#undef CTSEG_STEM
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient: R[32 rows = taps (row 27 = ones)][16*CT columns]; wave w owns k-steps 2w, 2w+1 of every tile
// ---------------------------------------------------------------------------------------------------------------------------
struct StemWgradArgs {
  const char* in;
  const char* dy;
  float* ws;
  int N, Xi, Yi, Zi, Xr, Yr, Zr, d_ld, kpad_w, cn_pad;
  StemGeom G;
  // DYN (ctseg_wgrad_desc::dyn_*): the upper half of the dY columns is formed on load from (g, y) of the norm behind it
  const char* dyn_g;
  const char* dyn_y;
  const float* dyn_mr;
  const float* dyn_alpha;
  const float* dyn_sums;
  int dyn_g_ld, dyn_y_ld;
};

constexpr int S_DYN_MAXN = 8;       // samples whose norm-backward constants fit the LDS table of the DYN variant

// DYN: columns [8 CT, 16 CT) of dY (the unit0 half of the fused [residual | unit0] first layer) are not read but formed from the
// gradient g behind that half's InstanceNorm + PReLU and the forward conv output y: the apply pass of the LAST norm of the backward,
// its 201 MB write and this kernel's read of it disappear from the tail of the step, and the lower half is read from the residual
// gradient where it lies (no copy into a fused operand).  Chunk c of staging slot j is c_thread ^ (CT (j & 1)): every thread then
// owns as many lower-half chunks (plain loads) as upper-half ones (two loads + the arithmetic), whatever its lane.
template <int CT, bool DYN = false>
__global__ __launch_bounds__(256) void conv_stem_wgrad_kernel(const StemWgradArgs P, int total_tiles) {
  // dy tile: planes of 16 channels, 32 B per voxel.  The plane pitch is 8 KiB + 32 B: the 8 lanes that store one voxel's 8 chunks
  // (4 planes x 32 B) then cover 128 distinct bytes of the bank period; with a pitch of exactly 8 KiB the four planes fell on the
  // same banks and every staging store was a 4-way conflict (768 conflict cycles per tile = 52 % of the kernel's LDS cycles)
  constexpr int DPL = 256 * 32 + 32;
  constexpr int DBYTES = CT * DPL, DCH = 256 * CT * 2, JD = DCH / 256;
  constexpr int BUF = S_INB + DBYTES;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef s16x4 __attribute__((address_space(3)))* lds_s16x4;
  static_assert(!DYN || ((CT == 2 || CT == 4) && JD % 2 == 0), "DYN: a power-of-two chunk count that divides the thread count");
  constexpr int DYNC = 8 * CT;                         // channels of the norm (the upper half of the columns)
  constexpr int TABB = DYN ? S_DYN_MAXN * DYNC * 16 : 0;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF + 32 + TABB];
  float* const sDyn = reinterpret_cast<float*>(smem + 2 * BUF + 32);     // [n][channel][mean, rstd, s1, s2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q4 = lane >> 4, tq = r16 >> 2, tp = r16 & 3;

  StemStage stg;
  stg.init(tid, P.Yi, P.Zi);
  const int YZ = P.Yr * P.Zr;
  const int c_thr = tid % (2 * CT);
  const bool flip = DYN && c_thr >= CT;                // this thread's EVEN slots are upper-half chunks (else its odd ones)
  int gd_byte[JD], gd_xyz[JD], gd_lds[JD], gy_byte[DYN ? JD / 2 : 1];
#pragma unroll
  for (int j = 0; j < JD; ++j) {
    const int idx = tid + j * 256;
    const int c0 = idx % (2 * CT), tv = idx / (2 * CT);
    const int c = DYN ? (c0 ^ (CT * (j & 1))) : c0;
    const int tx = tv >> 6, ty = (tv >> 3) & 7, tz = tv & 7;
    const int vox = tx * YZ + ty * P.Zr + tz;
    if (DYN && c >= CT) {
      gd_byte[j] = (vox * P.dyn_g_ld + (c - CT) * 8) * 2;
      gy_byte[j >> 1] = (vox * P.dyn_y_ld + (c - CT) * 8) * 2;
    } else {
      gd_byte[j] = (vox * P.d_ld + c * 8) * 2;
    }
    gd_xyz[j] = tx | (ty << 8) | (tz << 16);
    gd_lds[j] = (c >> 1) * DPL + tv * 32 + (c & 1) * 16;
  }
  if constexpr (DYN) {
    for (int i = tid; i < P.N * DYNC; i += 256) {
      *reinterpret_cast<f32x4*>(sDyn + 4 * i) = f32x4{P.dyn_mr[2 * i], P.dyn_mr[2 * i + 1], P.dyn_sums[2 * i], P.dyn_sums[2 * i + 1]};
    }
  }
  const float dyn_al = DYN ? P.dyn_alpha[0] : 1.f;
  auto origin = [&](int t, int& n, int& x0, int& y0, int& z0) {
    n = t / P.G.tiles;
    int r = t - n * P.G.tiles;
    const int tz = r % P.G.tzn; r /= P.G.tzn;
    const int ty = r % P.G.tyn; const int tx = r / P.G.tyn;
    x0 = tx * 4; y0 = ty * 8; z0 = tz * 8;
  };
  uint32_t rg[S_J];
  u32x4 rd[JD], ry[DYN ? JD / 2 : 1];
  auto gload = [&](int t) {
    int n, x0, y0, z0;
    origin(t, n, x0, y0, z0);
    const unsigned short* ib = reinterpret_cast<const unsigned short*>(P.in) + (((int64_t)n * P.Xi + 2 * x0) * P.Yi + 2 * y0) * P.Zi + 2 * z0;
    stg.load(ib, x0, y0, z0, P.Xi, P.Yi, P.Zi, rg);
    const int64_t tv0 = (((int64_t)n * P.Xr + x0) * P.Yr + y0) * P.Zr + z0;
    const char* db = P.dy + tv0 * P.d_ld * 2;
    const char* bE = db;                                // source of the even / odd slots of this thread
    const char* bO = db;
    if constexpr (DYN) {
      const char* gb = P.dyn_g + tv0 * P.dyn_g_ld * 2;
      bE = flip ? gb : db;
      bO = flip ? db : gb;
    }
#pragma unroll
    for (int j = 0; j < JD; ++j) {
      const int xi = x0 + (gd_xyz[j] & 0xff), yi = y0 + ((gd_xyz[j] >> 8) & 0xff), zi = z0 + (gd_xyz[j] >> 16);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (xi < P.Xr && yi < P.Yr && zi < P.Zr) v = *reinterpret_cast<const u32x4*>(((j & 1) ? bO : bE) + gd_byte[j]);
      rd[j] = v;
    }
    if constexpr (DYN) {                                // y of the thread's upper-half slots (slot 2k or 2k + 1)
      const char* yb = P.dyn_y + tv0 * P.dyn_y_ld * 2;
#pragma unroll
      for (int k = 0; k < JD / 2; ++k) {
        const int xyz = flip ? gd_xyz[2 * k] : gd_xyz[2 * k + 1];
        const int xi = x0 + (xyz & 0xff), yi = y0 + ((xyz >> 8) & 0xff), zi = z0 + (xyz >> 16);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (xi < P.Xr && yi < P.Yr && zi < P.Zr) v = *reinterpret_cast<const u32x4*>(yb + gy_byte[k]);
        ry[k] = v;
      }
    }
  };
  // DYN: the staged (g, y) chunks of tile t -> the dY chunks the apply pass would have stored (bf16 rounding included)
  auto convert = [&](int t) {
    int n, x0, y0, z0;
    origin(t, n, x0, y0, z0);
    const float* tab = sDyn + (n * DYNC + (c_thr & (CT - 1)) * 8) * 4;
    f32x4 k4[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) k4[e] = *reinterpret_cast<const f32x4*>(tab + 4 * e);
#pragma unroll
    for (int k = 0; k < JD / 2; ++k) {
      const u32x4 gq = flip ? rd[2 * k] : rd[2 * k + 1];
      const int xyz = flip ? gd_xyz[2 * k] : gd_xyz[2 * k + 1];
      const int xi = x0 + (xyz & 0xff), yi = y0 + ((xyz >> 8) & 0xff), zi = z0 + (xyz >> 16);
      const bool ok = xi < P.Xr && yi < P.Yr && zi < P.Zr;
      u32x4 o;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float r0 = inorm_prelu_bwd_value(bf2f(gq[h] & 0xffffu), bf2f(ry[k][h] & 0xffffu), k4[2 * h][0], k4[2 * h][1], k4[2 * h][2], k4[2 * h][3], dyn_al);
        const float r1 = inorm_prelu_bwd_value(bf2f(gq[h] >> 16), bf2f(ry[k][h] >> 16), k4[2 * h + 1][0], k4[2 * h + 1][1], k4[2 * h + 1][2], k4[2 * h + 1][3], dyn_al);
        o[h] = ok ? pack2bf(r0, r1) : 0u;
      }
      rd[2 * k] = flip ? o : rd[2 * k];
      rd[2 * k + 1] = flip ? rd[2 * k + 1] : o;
    }
  };
  auto sstore = [&](int buf) {
    char* b = smem + buf * BUF;
    stg.store(b, rg);
#pragma unroll
    for (int j = 0; j < JD; ++j) *reinterpret_cast<u32x4*>(b + S_INB + gd_lds[j]) = rd[j];
  };

  // A' operand: row = tap (r16 [+16]), k = voxel (y-row 2r + (q4>>1), z = 4*(q4&1) + j) of the k-step, as in conv_wgrad_halo
  int toffA[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int tap = a * 16 + r16;
    toffA[a] = tap < 27 ? stem_tap_off(tap) * 2 : (tap == 27 ? -2 : -1);     // -2: all-ones row (bias gradient), -1: zero row
  }
  f32x4 acc[2][CT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  int t = blockIdx.x, cur = 0;
  if constexpr (DYN) __syncthreads();       // constant table
  if (t < total_tiles) {
    gload(t);
    if constexpr (DYN) convert(t);
    sstore(0);
  }
  __syncthreads();
  for (; t < total_tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < total_tiles) gload(tn);
    const char* xs = smem + cur * BUF + S_SHIFT;
    const char* ds = xs - S_SHIFT + S_INB;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int s = 2 * wave + ss, x = s >> 1, yb = 4 * (s & 1);
      const int ly = q4 >> 1, lz0 = 4 * (q4 & 1);
      bf16x8 df[CT];
#pragma unroll
      for (int b = 0; b < CT; ++b) {
        const char* p0 = ds + b * DPL + (((x * 8) + (yb + ly)) * 8 + lz0 + tq) * 32 + tp * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 2 * 8 * 32));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        df[b] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        s16x8 av;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // voxel of k index 8*q4 + j: y = yb + 2*(j>>2) + ly, z = lz0 + (j&3)
          const int y = yb + 2 * (j >> 2) + ly, z = lz0 + (j & 3);
          short val = 0;
          if (toffA[a] >= 0) val = *reinterpret_cast<const short*>(xs + (((2 * x) * S_IY + 2 * y) * S_PZ + 2 * z) * 2 + toffA[a]);
          else if (toffA[a] == -2) val = (short)0x3f80;
          av[j] = val;
        }
        const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
        for (int b = 0; b < CT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, df[b], acc[a][b], 0, 0, 0);
      }
    }
    if constexpr (DYN) { if (tn < total_tiles) convert(tn); }      // (registers only: ahead of the barrier)
    __syncthreads();
    if (tn < total_tiles) sstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  // one slab per wave: rows 4*q4+e (+16a) = tap, column b*16 + r16
  float* slab = P.ws + ((int64_t)blockIdx.x * 4 + wave) * P.kpad_w * P.cn_pad;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < CT; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) slab[(int64_t)(a * 16 + 4 * q4 + e) * P.cn_pad + b * 16 + r16] = acc[a][b][e];
}

// ctseg_wgrad_desc::dyn_*: the upper half of the columns formed on load (DYN variant): 32 or 64 columns, <= S_DYN_MAXN samples
bool wgrad_stem_dyn_ok(const ctseg_wgrad_desc* d) {
  if (d->dyn_col0 * 2 != d->Cn || (d->Cn != 32 && d->Cn != 64) || d->N > S_DYN_MAXN) return false;
  if (d->dyn_g_ld % 8 != 0 || d->dyn_y_ld % 8 != 0 || d->dyn_g_ld < d->dyn_col0 || d->dyn_y_ld < d->dyn_col0) return false;
  if ((int64_t)d->Xr * d->Yr * d->Zr * d->dyn_g_ld * 2 >= (1ll << 31) || (int64_t)d->Xr * d->Yr * d->Zr * d->dyn_y_ld * 2 >= (1ll << 31)) return false;
  return true;
}

bool wgrad_stem_eligible(const ctseg_wgrad_desc* d) {
  if (d->dtype != CTSEG_BF16 || d->ntaps != 27 || d->sin != 2 || d->Cg != 1 || d->g_ld != 1) return false;
  const int dcols = d->dyn_col0 > 0 ? d->dyn_col0 : d->Cn;       // columns `dy` itself holds
  if (d->Cn > 64 || d->Cn % 16 != 0 || d->d_ld % 8 != 0 || d->d_ld < dcols || ((uintptr_t)d->dy % 16) != 0) return false;
  if (((uintptr_t)d->in % 4) != 0) return false;      // the patch is staged in aligned 4-byte pairs
  if (d->Xi != 2 * d->Xr || d->Yi != 2 * d->Yr || d->Zi != 2 * d->Zr || d->Zr < 4) return false;
  if (d->kpad_w < 32 || d->cn_pad < d->Cn) return false;
  if ((int64_t)d->Xi * d->Yi * d->Zi >= (1ll << 31) || (int64_t)d->Xr * d->Yr * d->Zr * d->d_ld * 2 >= (1ll << 31)) return false;
  for (int j = 0; j < 27; ++j) {
    const int tp = d->taps[j];
    const int ex = j / 9 - 1, ey = (j / 3) % 3 - 1, ez = j % 3 - 1;
    if ((int)(int8_t)(tp & 0xff) != ex || (int)(int8_t)((tp >> 8) & 0xff) != ey || (int)(int8_t)((tp >> 16) & 0xff) != ez) return false;
  }
  return true;
}

static int wgrad_stem_grid(const ctseg_wgrad_desc* d) {
  return persistent_grid(2 * CTSEG_NUM_CU, stem_tiles(d->Xr, d->Yr, d->Zr) * d->N);
}
int wgrad_stem_slabs(const ctseg_wgrad_desc* d) { return 4 * wgrad_stem_grid(d); }

void launch_wgrad_stem(const ctseg_wgrad_desc* d, hipStream_t st) {
  StemWgradArgs a;
  a.in = (const char*)d->in; a.dy = (const char*)d->dy; a.ws = d->ws;
  a.N = d->N; a.Xi = d->Xi; a.Yi = d->Yi; a.Zi = d->Zi; a.Xr = d->Xr; a.Yr = d->Yr; a.Zr = d->Zr;
  a.d_ld = d->d_ld; a.kpad_w = d->kpad_w; a.cn_pad = d->cn_pad;
  a.G.tiles = stem_tiles(d->Xr, d->Yr, d->Zr); a.G.tyn = (d->Yr + 7) / 8; a.G.tzn = (d->Zr + 7) / 8;
  const int total = a.G.tiles * d->N, grid = wgrad_stem_grid(d);
  const int ct = d->Cn / 16;
  a.dyn_g = (const char*)d->dyn_g; a.dyn_y = (const char*)d->dyn_y; a.dyn_mr = d->dyn_mean_rstd; a.dyn_alpha = d->dyn_alpha;
  a.dyn_sums = d->dyn_sums; a.dyn_g_ld = d->dyn_g_ld; a.dyn_y_ld = d->dyn_y_ld;
  if (d->dyn_g != nullptr) {      // (wgrad_stem_dyn_ok checked by the entry point)
    if (ct == 2) hipLaunchKernelGGL((conv_stem_wgrad_kernel<2, true>), dim3(grid), dim3(256), 0, st, a, total);
    else hipLaunchKernelGGL((conv_stem_wgrad_kernel<4, true>), dim3(grid), dim3(256), 0, st, a, total);
    return;
  }
  if (ct == 1) hipLaunchKernelGGL((conv_stem_wgrad_kernel<1>), dim3(grid), dim3(256), 0, st, a, total);
  else if (ct == 2) hipLaunchKernelGGL((conv_stem_wgrad_kernel<2>), dim3(grid), dim3(256), 0, st, a, total);
  else if (ct == 3) hipLaunchKernelGGL((conv_stem_wgrad_kernel<3>), dim3(grid), dim3(256), 0, st, a, total);
  else hipLaunchKernelGGL((conv_stem_wgrad_kernel<4>), dim3(grid), dim3(256), 0, st, a, total);
}

}  // namespace ctseg
