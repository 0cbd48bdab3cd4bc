// LDS-halo kernel for the 8-class stride-2 "up" pass with few output channels (gfx950, bf16):
// the top-level ConvTranspose3d(64 -> 10, k3 s2 p1 op1) of the reference's U-Net at 256x256x24 -> 512x512x48.
//
// Output parity class (px,py,pz) of a transposed conv is a small conv over the INPUT grid with taps at offsets
// {0} (parity 0) or {+1, 0} (parity 1) per axis: 1+2+2+2+4+4+4+8 = 27 taps over the 8 classes.  The generic kernel
// runs the classes as independent workgroups, each re-gathering its taps through the vector-memory path (27 gathers
// of a 128-byte voxel per input voxel).  Here one persistent workgroup stages the (4+1)x(8+1)x(8+1) input tile ONCE,
// keeps the packed weights of all 8 classes in LDS (27*Cg*16*2 B = 55 KB), and produces all 8 x 256 outputs of the
// tile from LDS.  LDS halo image, lane<->voxel permutation and the register epilogue are those of conv_halo.hip
// (same 6x10x10 slot geometry, so every ds_read_b128 operand fetch is bank-conflict free); InstanceNorm partial sums
// are accumulated over the 8 classes in registers -> one partial slot per tile.
#include "conv_common.h"
#include <type_traits>

#ifndef UP_DELTA
#define UP_DELTA 1   // 1: delta-major schedule (shifted fragments read once per K chunk for all classes of a wave half); 0: class by class
#endif
#ifndef UP_ABL
#define UP_ABL 0     // timing-only ablation: 1 no MFMAs, 2 no LDS operand reads, 8 no output stores, 16 no halo loads
#endif

namespace ctseg {

constexpr int U_HY = 10, U_HZ = 10, U_HV = 600, U_PLANE = U_HV * 16;
constexpr int U_FV = 5 * 9 * 9;   // voxels actually filled

__device__ __forceinline__ void up_patch_voxel(int r16, int& dy, int& z) {
  dy = (0xEF80u >> r16) & 1;
  z = (int)((0x2104765437653210ull >> (4 * r16)) & 7ull);
}

// tap index of shift delta = (dx,dy,dz) in {0,1}^3 inside parity class c = px*4 + py*2 + pz, in the order the host lists a class's taps
// (capstone_amd/engine.py classes_up: x outermost; a parity-1 axis lists offset +1 before 0) — checked by conv_up_eligible; -1: the
// class has no such tap
constexpr int up_tap_index(int c, int delta) {
  const int px = (c >> 2) & 1, py = (c >> 1) & 1, pz = c & 1, dx = (delta >> 2) & 1, dy = (delta >> 1) & 1, dz = delta & 1;
  if (dx > px || dy > py || dz > pz) return -1;
  const int ix = px ? 1 - dx : 0, iy = py ? 1 - dy : 0, iz = pz ? 1 - dz : 0;
  return (ix * (1 + py) + iy) * (1 + pz) + iz;
}

template <int VB> struct UpCfg {
  static constexpr int NPL = VB / 16;
  static constexpr int HALO = NPL * U_PLANE;
  static constexpr int WBYTES = 27 * (VB / 64) * 16 * 64;   // 27 taps x (VB/64) 64-byte chunks x 16 rows  (= 27*Cg*16*2)
  static constexpr int TOTAL = WBYTES + 2048 + HALO + 8 * 2 * 16 * 4 + 64 * 4 + 16 * 4;
};

template <typename H, int VB>     // H = 16-bit storage kind (BF16 / F16)
__global__ __launch_bounds__(512) void conv_up_halo_kernel(const ConvKArgs P, int total_tiles, int tyn, int tzn) {
  using CF = UpCfg<VB>;
  constexpr int NPL = CF::NPL, CPT = VB / 64;           // 64-byte K chunks per tap
  constexpr int NTHR = 512;      // 8 waves: waves 0-3 / 4-7 split the 8 parity classes (13 / 14 taps) over the same 4x8x8 tile
  constexpr int NCH = U_FV * NPL, J = (NCH + NTHR - 1) / NTHR;
  __shared__ __attribute__((aligned(16))) char smem[CF::TOTAL];
  char* const sW = smem;                                 // per class: [stages][16 rows][128 B] swizzled, stage padded
  char* const sH = smem + CF::WBYTES + 2048;
  float* const sStats = reinterpret_cast<float*>(sH + CF::HALO);
  int* const sDelta = reinterpret_cast<int*>(sH + CF::HALO + 8 * 2 * 16 * 4);   // [8 classes][8 taps]
  int* const sWst = sDelta + 64;                                                  // first weight stage of each class

  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6, wave = wave8 & 3, half = wave8 >> 2;
  const int r16 = lane & 15, q4 = lane >> 4;

  // ---- weights of all classes -> LDS; class c starts at stage wstage[c] ------------------------------------------------
  int wstage[9];
  wstage[0] = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) wstage[c + 1] = wstage[c] + P.cls[c].kpad * 2 / 128;
  for (int c = 0; c < 8; ++c) {
    const int nst = wstage[c + 1] - wstage[c];
    for (int idx = tid; idx < 16 * nst * 8; idx += NTHR) {
      const int q8 = idx & 7, row = (idx >> 3) & 15, s = idx >> 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(P.w + (P.cls[c].w_off + (int64_t)row * P.cls[c].kpad) * 2 + s * 128 + q8 * 16);
      *reinterpret_cast<u32x4*>(sW + ((wstage[c] + s) * 16 + row) * 128 + ((q8 ^ ((row >> 1) & 7)) << 4)) = v;
    }
  }
  if (tid < 9) sWst[tid] = wstage[tid];
  if (tid < 64) {
    const int c = tid >> 3, tp_i = tid & 7;
    int d = 0;
    if (tp_i < P.cls[c].ntaps) {
      const int tp = P.cls[c].taps[tp_i];
      d = ((int)(int8_t)(tp & 0xff) * U_HY + (int)(int8_t)((tp >> 8) & 0xff)) * U_HZ + (int)(int8_t)((tp >> 16) & 0xff);
    }
    sDelta[tid] = d * 16;
  }

  // ---- staging slots: the 5x9x9 voxels at halo coordinates 1..5 x 1..9 x 1..9 --------------------------------------------
  const int YZ = P.Yi * P.Zi;
  int g_byte[J], g_hxyz[J], g_lds[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int idx = tid + j * NTHR;
    const int pl = (idx >> 3) % NPL, fv = (idx / (8 * NPL)) * 8 + (idx & 7);
    const int fx = fv / 81, rem = fv - fx * 81, fy = rem / 9, fz = rem - fy * 9;     // offsets 0..4, 0..8, 0..8
    g_byte[j] = ((fx * YZ + fy * P.Zi + fz) * P.g_ld + pl * 8) * 2;
    g_hxyz[j] = (fv < U_FV) ? (fx | (fy << 8) | (fz << 16)) : 0x7f7f7f;   // 405 is not a multiple of 8: test fv, not idx
    g_lds[j] = pl * U_PLANE + (((fx + 1) * U_HY + (fy + 1)) * U_HZ + (fz + 1)) * 16;
  }
  const int tiles_per_sample = P.tiles;
  auto tile_origin = [&](int t, int& n, int& x0, int& y0, int& z0) {
    n = t / tiles_per_sample;
    int r = t - n * tiles_per_sample;
    const int tz = r % tzn; r /= tzn;
    const int ty = r % tyn; const int tx = r / tyn;
    x0 = tx * 4; y0 = ty * 8; z0 = tz * 8;
  };
  u32x4 rh[J];
  // branch-free: raw buffer loads with a per-sample descriptor, voxels outside the volume get an out-of-range offset (zeros).  (Plain
  // loads under `if (inside)` made the compiler wait for each one before the next branch: 0.10 of this pass's 0.37 ms at 2 x 256 x
  // 256 x 24 was the exposed latency of these seven loads.)
  const int in_sample_bytes = (int)((int64_t)P.Xi * P.Yi * P.Zi * P.g_ld * 2);
  auto gload = [&](int t) {
    int n, x0, y0, z0;
    tile_origin(t, n, x0, y0, z0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.in) + (int64_t)n * in_sample_bytes, 0, in_sample_bytes, 0x00020000);
    const int soff = ((x0 * P.Yi + y0) * P.Zi + z0) * P.g_ld * 2;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int xi = x0 + (g_hxyz[j] & 0xff), yi = y0 + ((g_hxyz[j] >> 8) & 0xff), zi = z0 + (g_hxyz[j] >> 16);
      const bool inside = !(UP_ABL & 16) && xi < P.Xi && yi < P.Yi && zi < P.Zi;
      rh[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, inside ? g_byte[j] : (int)0x80000000, soff, 0);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < J; ++j)
      if ((g_hxyz[j] & 0xff) != 0x7f) *reinterpret_cast<u32x4*>(sH + g_lds[j]) = rh[j];
  };

  int pdy, pz;
  up_patch_voxel(r16, pdy, pz);
  int abase[4], ovox[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    abase[i] = ((((wave + 1) * U_HY) + (2 * i + pdy + 1)) * U_HZ + (pz + 1)) * 16 + q4 * U_PLANE;
    ovox[i] = ((2 * wave) * P.Yo + 2 * (2 * i + pdy)) * P.Zo + 2 * pz;     // from the tile's first OUTPUT voxel (2x0,2y0,2z0)
  }
  const int wrow = r16 * 128, wswz = (r16 >> 1) & 7;
  const bool af32 = P.add_f32 != 0;
  const int ASZ = af32 ? 4 : 2;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = (P.bias != nullptr && 4 * q4 + e < P.Cn) ? P.bias[4 * q4 + e] : 0.f;
  const int ch = 4 * q4;

  // InstanceNorm partial sums stay in registers across the workgroup's tiles and are written once per (workgroup, sample):
  // one partial slot per workgroup instead of one per tile (12 288 tiles per sample at 256x256x24 -> 256 slots: no per-tile
  // shuffles / barrier, and the finalize that sits between this pass and its norm pass reads 48x fewer rows)
  float wsum[4] = {0.f, 0.f, 0.f, 0.f}, wsq[4] = {0.f, 0.f, 0.f, 0.f};
  int stat_n = -1;
  auto flush_stats = [&](int n) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = wsum[e], b = wsq[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (r16 == 0) {
        sStats[(wave8 * 2 + 0) * 16 + 4 * q4 + e] = a;
        sStats[(wave8 * 2 + 1) * 16 + 4 * q4 + e] = b;
      }
      wsum[e] = 0.f;
      wsq[e] = 0.f;
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, c = tid & 15;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += sStats[(w * 2 + which) * 16 + c];
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      P.stats[(slot_t * 2 + which) * P.stats_ld + c] = a;
    }
    __syncthreads();
  };

  auto compute_tile = [&](int t) {
    int n, x0, y0, z0;
    tile_origin(t, n, x0, y0, z0);
    if (P.stats != nullptr && n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = n;
    }
    const int64_t vb = (((int64_t)n * P.Xo + 2 * x0) * P.Yo + 2 * y0) * P.Zo + 2 * z0;
    const bool xok = x0 + wave < P.Xr, zok = z0 + pz < P.Zr;
    bool rv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rv[i] = xok && zok && (y0 + 2 * i + pdy < P.Yr);
    // one parity class, tap count known at compile time (class c has 2^popcount(c) taps — checked by conv_up_eligible): the
    // fully unrolled body lets the scheduler run the LDS operand reads of the next taps under the MFMAs of the current one,
    // which matters here because the 134 KB LDS image leaves a single wave per SIMD
    auto do_class = [&](auto ntaps_c, const int c) {
      constexpr int NTP = decltype(ntaps_c)::value;
      const ctseg_conv_class& K = P.cls[c];
      f32x4 acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* wc = sW + sWst[c] * (16 * 128) + wrow;
#pragma unroll
      for (int tp = 0; tp < NTP; ++tp) {
        const int delta = sDelta[c * 8 + tp];
#pragma unroll
        for (int kc = 0; kc < CPT; ++kc) {
          const int ci = tp * CPT + kc;                 // 64-byte chunk index inside the class
          u32x4 xf[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            xf[i] = (UP_ABL & 2) ? u32x4{(uint32_t)(delta + i), 1u, 2u, (uint32_t)kc} : *reinterpret_cast<const u32x4*>(sH + abase[i] + kc * 4 * U_PLANE + delta);
          const u32x4 wf = (UP_ABL & 2) ? u32x4{(uint32_t)ci, 3u, (uint32_t)tp, 5u}
                                        : *reinterpret_cast<const u32x4*>(wc + (ci >> 1) * (16 * 128) + (((4 * (ci & 1) + q4) ^ wswz) << 4));
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr ((UP_ABL & 1) != 0) acc[i][0] += __builtin_bit_cast(f32x4, wf)[0] * __builtin_bit_cast(f32x4, xf[i])[1];
            else mma16<H>(acc[i], wf, xf[i]);
          }
        }
      }
      // epilogue of this class: output voxel = 2*voxel + (ox,oy,oz)
      const int coff = (K.ox * P.Yo + K.oy) * P.Zo + K.oz;
      char* ob = P.out + (vb + coff) * P.o_ld * 2;
      const char* ab = (P.add != nullptr) ? P.add + (vb + coff) * P.add_ld * ASZ : nullptr;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][e] + bias[e];
          if (rv[i]) { wsum[e] += v[e]; wsq[e] += v[e] * v[e]; }
        }
        if (rv[i] && ch < P.Cn_store && (!(UP_ABL & 8) || v[0] + v[1] + v[2] + v[3] == 1.2345f)) {
          if (ab != nullptr) {
            const char* ap = ab + ((int64_t)ovox[i] * P.add_ld + ch) * ASZ;
            if (af32) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap); v[0] += a4[0]; v[1] += a4[1]; v[2] += a4[2]; v[3] += a4[3]; }
            else {
              const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap);
              v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
            }
          }
          *reinterpret_cast<u32x2*>(ob + ((int64_t)ovox[i] * P.o_ld + ch) * 2) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
        }
      }
    };
    using std::integral_constant;
#if UP_DELTA
    // "delta-major": the 27 (class, tap) pairs of a transposed conv read only the 8 shifted fragments x[v + delta], delta in {0,1}^3.
    // Each shifted fragment is read ONCE per 64-byte K chunk and feeds every class of this wave half that has the tap (64 operand
    // reads per tile and wave instead of 104 - 120); the accumulators of the half's four classes are live together (64 registers).
    auto do_half = [&](auto c0_, auto c1_, auto c2_, auto c3_) {
      constexpr int CL[4] = {decltype(c0_)::value, decltype(c1_)::value, decltype(c2_)::value, decltype(c3_)::value};
      f32x4 acc[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[k][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* wc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) wc[k] = sW + sWst[CL[k]] * (16 * 128) + wrow;
#pragma unroll
      for (int dl = 0; dl < 8; ++dl) {
        constexpr int dummy = 0; (void)dummy;
        const int delta = ((((dl >> 2) & 1) * U_HY + ((dl >> 1) & 1)) * U_HZ + (dl & 1)) * 16;
        const bool used = up_tap_index(CL[0], dl) >= 0 || up_tap_index(CL[1], dl) >= 0 || up_tap_index(CL[2], dl) >= 0 || up_tap_index(CL[3], dl) >= 0;
        if (!used) continue;
#pragma unroll
        for (int kc = 0; kc < CPT; ++kc) {
          u32x4 xf[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const u32x4*>(sH + abase[i] + kc * 4 * U_PLANE + delta);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int tp = up_tap_index(CL[k], dl);
            if (tp < 0) continue;
            const int ci = tp * CPT + kc;
            const u32x4 wf = *reinterpret_cast<const u32x4*>(wc[k] + (ci >> 1) * (16 * 128) + (((4 * (ci & 1) + q4) ^ wswz) << 4));
#pragma unroll
            for (int i = 0; i < 4; ++i) mma16<H>(acc[k][i], wf, xf[i]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const ctseg_conv_class& K = P.cls[CL[k]];
        const int coff = (K.ox * P.Yo + K.oy) * P.Zo + K.oz;
        char* ob = P.out + (vb + coff) * P.o_ld * 2;
        const char* ab = (P.add != nullptr) ? P.add + (vb + coff) * P.add_ld * ASZ : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[k][i][e] + bias[e];
            if (rv[i]) { wsum[e] += v[e]; wsq[e] += v[e] * v[e]; }
          }
          if (rv[i] && ch < P.Cn_store) {
            if (ab != nullptr) {
              const char* ap = ab + ((int64_t)ovox[i] * P.add_ld + ch) * ASZ;
              if (af32) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap); v[0] += a4[0]; v[1] += a4[1]; v[2] += a4[2]; v[3] += a4[3]; }
              else {
                const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap);
                v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
              }
            }
            *reinterpret_cast<u32x2*>(ob + ((int64_t)ovox[i] * P.o_ld + ch) * 2) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
          }
        }
      }
    };
    // (same split as below: classes that differ in the z parity share a wave half: their two partial line writes meet in L2)
    if (half == 0) do_half(integral_constant<int, 7>{}, integral_constant<int, 6>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
    else do_half(integral_constant<int, 5>{}, integral_constant<int, 4>{}, integral_constant<int, 3>{}, integral_constant<int, 2>{});
#else
    // classes that differ in the z parity write z-neighbouring voxels (two 24-byte rows = 48 contiguous bytes): they run back to back
    // in the same waves so that the two partial writes of a line meet in L2 (15 / 12 taps per wave half instead of 13 / 14)
    if (half == 0) {
      do_class(integral_constant<int, 8>{}, 7);
      do_class(integral_constant<int, 4>{}, 6);
      do_class(integral_constant<int, 2>{}, 1);
      do_class(integral_constant<int, 1>{}, 0);
    } else {
      do_class(integral_constant<int, 4>{}, 5);
      do_class(integral_constant<int, 2>{}, 4);
      do_class(integral_constant<int, 4>{}, 3);
      do_class(integral_constant<int, 2>{}, 2);
    }
#endif
  };

  const int G = gridDim.x;
  int first, stride, last;
  if ((G & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    first = xcd * chunk + (blockIdx.x >> 3);
    stride = G >> 3;
    last = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  } else {
    first = blockIdx.x; stride = G; last = total_tiles;
  }
  // the unused -1 border of the 6x10x10 image is never read (tap offsets are 0 / +1), no need to clear it
  int t = first;
  if (t < last) {
    gload(t);
    sstore();
  }
  __syncthreads();
  for (; t < last; t += stride) {
    const int tn = t + stride;
    if (tn < last) gload(tn);
    compute_tile(t);
    __syncthreads();          // every wave is done reading the halo (and the stats slots)
    if (tn < last) sstore();
    __syncthreads();
  }
  if (P.stats != nullptr && stat_n >= 0) flush_stats(stat_n);
}

bool conv_up_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (!is16(dtype) || nclass != 8 || a.sin != 1 || a.sout != 2 || a.out_f32) return false;
  const int vb = a.Cg * 2;
  if (!(vb == 64 || vb == 128) || a.Cn > 16) return false;
  if ((a.g_ld % 8) != 0 || ((uintptr_t)a.in % 16) != 0) return false;
  if (a.Xr != a.Xi || a.Yr != a.Yi || a.Zr != a.Zi || a.Zr < 4) return false;
  if (a.Xo != 2 * a.Xr || a.Yo != 2 * a.Yr || a.Zo != 2 * a.Zr) return false;
  if ((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2 >= (1ll << 31)) return false;
  int taps = 0;
  for (int c = 0; c < 8; ++c) {
    const ctseg_conv_class& k = a.cls[c];
    if (k.ntaps != (1 << __builtin_popcount(c)) || k.kpad != ((k.ntaps * a.Cg + 63) / 64) * 64) return false;
    taps += k.ntaps;
    for (int j = 0; j < k.ntaps; ++j)
      for (int s = 0; s < 24; s += 8) {
        const int d = (int)(int8_t)((k.taps[j] >> s) & 0xff);
        if (d < 0 || d > 1) return false;
      }
    // the delta-major schedule addresses a class's taps by (dx,dy,dz): class c = px*4 + py*2 + pz at parity (px,py,pz), taps in the
    // host's order (up_tap_index)
    if (k.ox != ((c >> 2) & 1) || k.oy != ((c >> 1) & 1) || k.oz != (c & 1)) return false;
    for (int dl = 0; dl < 8; ++dl) {
      const int tp = up_tap_index(c, dl);
      if (tp < 0) continue;
      if (tp >= k.ntaps) return false;
      const int t = k.taps[tp];
      if ((int)(int8_t)(t & 0xff) != ((dl >> 2) & 1) || (int)(int8_t)((t >> 8) & 0xff) != ((dl >> 1) & 1) || (int)(int8_t)((t >> 16) & 0xff) != (dl & 1)) return false;
    }
  }
  return taps == 27;
}

int conv_up_tiles(const ConvKArgs& a) { return ((a.Xr + 3) / 4) * ((a.Yr + 7) / 8) * ((a.Zr + 7) / 8); }

static int up_grid(const ConvKArgs& a) {
  const int total = conv_up_tiles(a) * a.N;
  return persistent_grid((a.Cg * 2 == 128) ? CTSEG_NUM_CU : 2 * CTSEG_NUM_CU, total);
}
// InstanceNorm partial slots per sample: one per workgroup
int conv_up_slots(const ConvKArgs& a) { return up_grid(a); }

void launch_conv_up(ConvKArgs& a, hipStream_t st) {
  const int tyn = (a.Yr + 7) / 8, tzn = (a.Zr + 7) / 8;
  a.tiles = conv_up_tiles(a);
  const int total = a.tiles * a.N;
  const int vb = a.Cg * 2;
  const int gx = up_grid(a);
  if (a.dtype == CTSEG_F16) {
    if (vb == 128) hipLaunchKernelGGL((conv_up_halo_kernel<F16, 128>), dim3(gx), dim3(512), 0, st, a, total, tyn, tzn);
    else hipLaunchKernelGGL((conv_up_halo_kernel<F16, 64>), dim3(gx), dim3(512), 0, st, a, total, tyn, tzn);
  } else {
    if (vb == 128) hipLaunchKernelGGL((conv_up_halo_kernel<BF16, 128>), dim3(gx), dim3(512), 0, st, a, total, tyn, tzn);
    else hipLaunchKernelGGL((conv_up_halo_kernel<BF16, 64>), dim3(gx), dim3(512), 0, st, a, total, tyn, tzn);
  }
}

}  // namespace ctseg
