// LDS-halo kernel for the STRIDE-2 3x3x3 passes with few gathered channels (gfx950, bf16):
//   * input gradient of the top-level ConvTranspose3d(64 -> 10) = a stride-2 conv 16 -> 64 over the full-resolution gradient
//   * Conv3d 32 -> 64+64 k3 s2 (fused [residual | unit0] of the second down block) and the level-1 transposed conv's input gradient
// of the reference's U-Net (MONAI UNet built at capstone/volumetric/base_trainer.py:65-72).
//
//   out[r][n] = bias[n] + add[r][n] + sum_{tap,c} in[2 r + d(tap)][c] * W[n][tap*Cg + c],   d in {-1,0,1}^3
//
// The generic kernel gathers 27 x (32 or 64) bytes per output row through L2 in 16-byte requests — request-rate bound
// (0.44 ms for the 16 -> 64 pass whose operands are 1.2 GB).  Here a workgroup stages the (2 TA + 1) x 17 x 17 input region
// of a TA x 8 x 8 output tile once (register prefetch of the next tile under the MFMAs), streams the K-contiguous packed
// weights through a two-slot LDS ring with direct-to-LDS loads (as conv_halo_sw.hip) and handles 64 output columns
// (blockIdx.y picks the column block).
//
// LDS halo image: planes of 16-byte channel chunks, [plane][(ha * 17 + hb) * 18 + (hc ^ ((hb >> 1) & 1))][16 B].  An MFMA
// operand row group = 2 (b) x 8 (c) output voxels = input slots two apart: 8 lanes of a b-row cover 256 B on the even (or
// odd) 16-byte slots, and the XOR puts the other b-row (hb differs by 2) on the opposite parity: every ds_read_b128 phase
// hits 16 distinct slots for every tap.
#include "conv_common.h"
#include <type_traits>

namespace ctseg {

constexpr int DH_NTHR = 512, DH_HB = 17, DH_HC = 18, DH_CN = 64, DH_KSB = DH_CN * 128;   // bytes of one 128-byte K stage of 64 rows

template <int VB> struct DownCfg {
  static constexpr int TA = 4;                      // 4 x 8 x 8 output tile, 8 waves (a 4-wave 2 x 8 x 8 variant with two workgroups per
                                                    // CU measured 0.63 ms against 0.40 ms: a barrier per K stage costs more than the overlap gains)
  static constexpr int HA = 2 * TA + 1;
  static constexpr int HV = HA * DH_HB * DH_HC;
  static constexpr int NPL = VB / 16, PLANE = HV * 16, HALO = NPL * PLANE;
  static constexpr int CG = VB / 2;
  static constexpr int G = 3;                       // 128-byte K stages per ring slot
  static constexpr int NKS_ = ((27 * CG + 31) / 32 + 1) / 2;   // 128-byte K stages of the whole tap list
  // RES: all K stages of the 64 columns stay in LDS for the whole launch (16 gathered channels: 7 stages = 56 KB beside the 88 KB
  // halo) instead of being streamed through a two-slot ring every tile.  The ring cost three stage barriers per tile, each a full
  // vmcnt(0) wait that also drained the register prefetch of the NEXT tile's halo issued just before it: 7.5 us per tile of which
  // 0.85 are multiplies.
  static constexpr bool RES = HALO + NKS_ * DH_KSB + 8 * 2 * DH_CN * 4 + 64 * 4 + 64 * 4 <= 160 * 1024;
  static constexpr int WRING = RES ? NKS_ * DH_KSB : 2 * G * DH_KSB;      // 57344 resident / 49152 ring
  static constexpr int TOTAL = HALO + WRING + 8 * 2 * DH_CN * 4 + 64 * 4 + 64 * 4;      // ... + tap table + BST constants
};

struct DownGeom {
  int da, db, dc;            // output (row grid) extents along the tile axes
  int xa, xb, xc;            // input extents along the tile axes
  int ia, ib, ic;            // input voxel strides of the tile axes
  int oa, ob, oc;            // output voxel strides of the tile axes
  int pa, pb, pc;            // volume axis of each tile axis
  int tbn, tcn, tiles;
};

// R12: the gathered rows are 12 elements wide (24 bytes: the <= 12-channel gradient of the head's transposed conv): staged in
// 8-byte pieces (3 per voxel) into the same two-plane image; the upper half of every plane-1 slot is zeroed once.
// BJ >= 0: backward InstanceNorm statistics (ConvKArgs::bst) of the 32 written columns 16 BJ .. 16 BJ + 31 (the half of the
// transposed conv's input gradient that belongs to the sub-block's output: its last norm consumes it).  A lane keeps the sums of
// its 8 channels (packed pairs) over both row tiles and every tile of a sample; the y values of a tile are requested at the top of
// the tile, ahead of the next tile's halo, and have the multiplies to arrive; combined per workgroup at every sample change.
template <typename H, int VB, bool STATS, bool R12 = false, int BJ = -1>     // H = 16-bit storage kind (BF16 / F16)
__global__ __launch_bounds__(DH_NTHR) void conv_down_halo_kernel(const ConvKArgs P, const DownGeom G, int total_tiles) {
  static_assert(!R12 || VB == 32, "12-wide rows: 16 gathered channels");
  constexpr bool BST = BJ >= 0;
  static_assert(!BST || (VB == 32 && !STATS && (BJ == 0 || BJ == 2)), "backward statistics: the 16-channel gradient pass, one column block, 32 aligned columns");
  using CF = DownCfg<VB>;
  constexpr int TA = CF::TA, HA = CF::HA, NPL = CF::NPL, PLANE = CF::PLANE, CG = CF::CG, RG = CF::G;
  constexpr int FV = HA * 17 * 17, NCH = FV * (R12 ? 3 : NPL), J = (NCH + DH_NTHR - 1) / DH_NTHR;
  using RH = std::conditional_t<R12, u32x2, u32x4>;
  constexpr int NU = (27 * CG + 31) / 32;                 // 32-wide k-steps that carry real taps
  constexpr int NKS = (NU + 1) / 2;                       // 128-byte K stages to stream
  constexpr int NRS = (NKS + RG - 1) / RG;                // ring stages per tile
  // wave layout: 16 row tiles of 16 voxels, 8 waves x (2 row tiles x 4 column tiles)
  constexpr int RT = 2, NT = VB == 32 ? 4 : 2;

  __shared__ __attribute__((aligned(16))) char smem[CF::TOTAL];
  char* const sH = smem;
  char* const sW = smem + CF::HALO;
  float* const sStats = reinterpret_cast<float*>(sW + CF::WRING);
  int* const sTab = reinterpret_cast<int*>(sStats + 8 * 2 * DH_CN);   // per tap: [0,32) halo byte offset, [32,64) XOR flags
  float* const sBt = reinterpret_cast<float*>(sTab + 64);              // BST: [0,32) rstd, [32,64) -mean * rstd of the sample's channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;
  const ctseg_conv_class& K = P.cls[0];
  const int col0 = blockIdx.y * DH_CN;
  const int kpad = K.kpad;

  if (tid < 32) {
    // slot of (lane voxel, tap) = lane part + tap part, except for the XOR: hc ^ s = hc + s * (hc even ? +1 : -1) with
    // s = ((2 vb + db1) >> 1) & 1 = (vb & 1) ^ (db1 >> 1) and hc parity = dc1 parity  ->  per tap: byte offset, (db1 >> 1), +-16
    int da1 = 1, db1 = 1, dc1 = 1;                          // phantom taps (zero weights) read the centre voxel
    if (tid < 27) {
      const int tp = K.taps[tid];
      const int dv[3] = {(int)(int8_t)(tp & 0xff), (int)(int8_t)((tp >> 8) & 0xff), (int)(int8_t)((tp >> 16) & 0xff)};
      da1 = dv[G.pa] + 1; db1 = dv[G.pb] + 1; dc1 = dv[G.pc] + 1;
    }
    sTab[tid] = ((da1 * DH_HB + db1) * DH_HC + dc1) * 16;
    sTab[32 + tid] = (db1 >> 1) | ((dc1 & 1) ? 0x100 : 0);     // bit 0: h, bit 8: hc odd (step -16 instead of +16)
  }

  // ---- halo staging slots -----------------------------------------------------------------------------------------------
  int g_byte[J], g_fabc[R12 ? 1 : J], g_lds[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int idx = tid + j * DH_NTHR;
    // 16-byte chunks: 8 consecutive threads take 8 consecutive voxels of one plane; 8-byte pieces (R12): voxel-major, 3 per voxel
    const int pl = R12 ? idx % 3 : (idx >> 3) % NPL, fv = R12 ? idx / 3 : (idx / (8 * NPL)) * 8 + (idx & 7);
    const int fa = fv / (17 * 17), rem = fv - fa * (17 * 17), fb = rem / 17, fc = rem - fb * 17;
    const int slot16 = ((fa * DH_HB + fb) * DH_HC + (fc ^ ((fb >> 1) & 1))) * 16;
    if constexpr (R12) {
      // byte offset from the halo origin (sign bit: no such piece); LDS offset (a multiple of 8) | "first halo plane along a / b / c"
      g_byte[j] = fv < FV ? (fa * G.ia + fb * G.ib + fc * G.ic) * P.g_ld * 2 + pl * 8 : (int)0x80000000;
      g_lds[j] = ((pl == 2 ? PLANE : 0) + slot16 + (pl == 1 ? 8 : 0)) | (fa == 0 ? 1 : 0) | (fb == 0 ? 2 : 0) | (fc == 0 ? 4 : 0);
    } else {
      g_byte[j] = ((fa * G.ia + fb * G.ib + fc * G.ic) * P.g_ld + pl * 8) * 2;
      g_fabc[j] = (fv < FV) ? (fa | (fb << 8) | (fc << 16)) : 0x7f7f7f;
      g_lds[j] = pl * PLANE + slot16;
    }
  }
  auto tile_origin = [&](int t, int& n, int& a0, int& b0, int& c0) {
    n = t / G.tiles;
    int r = t - n * G.tiles;
    const int tc = r % G.tcn; r /= G.tcn;
    const int tb = r % G.tbn; const int ta = r / G.tbn;
    a0 = ta * TA; b0 = tb * 8; c0 = tc * 8;
  };
  const int64_t in_sample = (int64_t)P.Xi * P.Yi * P.Zi, out_sample = (int64_t)P.Xo * P.Yo * P.Zo;
  RH rh[J];
  if constexpr (R12) {
    for (int i = tid; i < CF::HV; i += DH_NTHR) *reinterpret_cast<u32x2*>(sH + PLANE + i * 16 + 8) = u32x2{0u, 0u};
  }
  auto gload = [&](int t) {
    int n, a0, b0, c0;
    tile_origin(t, n, a0, b0, c0);
    if constexpr (R12) {
      // the input is exactly twice the row grid (host-checked): only the -1 taps of the first tile along an axis leave the volume;
      // what a partial tile reads past the grid feeds rows that are not stored (finite, or zero past the end of the sample)
      const int bias = (G.ia + G.ib + G.ic) * P.g_ld * 2, sb = (int)(in_sample * P.g_ld * 2);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.in) + (int64_t)n * sb - bias, 0, sb + bias, 0x00020000);
      const int soff = (2 * a0 * G.ia + 2 * b0 * G.ib + 2 * c0 * G.ic) * P.g_ld * 2;
      const int m = (a0 == 0 ? 1 : 0) | (b0 == 0 ? 2 : 0) | (c0 == 0 ? 4 : 0);
#pragma unroll
      for (int j = 0; j < J; ++j)
        rh[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, ((g_lds[j] & m) == 0 && g_byte[j] >= 0) ? g_byte[j] : (int)0x80000000, soff, 0);
    } else {
      const int ai0 = 2 * a0 - 1, bi0 = 2 * b0 - 1, ci0 = 2 * c0 - 1;       // input coordinate of halo (0,0,0)
      // branch-free (plain loads under `if (inside)` were waited for one by one): out-of-volume voxels get an out-of-range offset
      const int bias = (G.ia + G.ib + G.ic) * P.g_ld * 2, sb = (int)(in_sample * P.g_ld * 2);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.in) + (int64_t)n * sb - bias, 0, sb + bias, 0x00020000);
      const int soff = (2 * a0 * G.ia + 2 * b0 * G.ib + 2 * c0 * G.ic) * P.g_ld * 2;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int ai = ai0 + (g_fabc[j] & 0xff), bi = bi0 + ((g_fabc[j] >> 8) & 0xff), ci = ci0 + (g_fabc[j] >> 16);
        const bool inside = (unsigned)ai < (unsigned)G.xa && (unsigned)bi < (unsigned)G.xb && (unsigned)ci < (unsigned)G.xc;
        if constexpr (!R12) rh[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, inside ? g_byte[j] : (int)0x80000000, soff, 0);
      }
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if constexpr (R12) { if (g_byte[j] >= 0) *reinterpret_cast<RH*>(sH + (g_lds[j] & ~7)) = rh[j]; }
      else { if ((g_fabc[j] & 0xff) != 0x7f) *reinterpret_cast<RH*>(sH + g_lds[j]) = rh[j]; }
    }
  };

  // ---- weight stream ---------------------------------------------------------------------------------------------------
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int w_row = (tid >> 3) & (DH_NTHR / 8 - 1), w_q8 = (tid & 7) ^ ((w_row >> 1) & 7);
  const char* const wsrc = P.w + (K.w_off + (int64_t)(col0 + w_row) * kpad + w_q8 * 8) * 2;
  int wstage = 0;
  auto wload = [&](int rs, int slot) {
    char* dst = sW + slot * (RG * DH_KSB) + wave * 1024;
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const int ks = rs * RG + g;
      if (ks < NKS) {
#pragma unroll
        for (int rd = 0; rd < 512 / DH_NTHR; ++rd)         // 512 16-byte chunks per K stage of 64 rows: rows 32 rd .. 32 rd + 31
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ((int64_t)rd * 32 * kpad) * 2 + (int64_t)ks * 128),
                                           (lptr_t)(dst + g * DH_KSB + rd * (DH_NTHR * 16)), 16, 0, 0);
      }
    }
  };

  // ---- per-lane constants ------------------------------------------------------------------------------------------------
  const int pb = r16 >> 3, pc = r16 & 7;
  int va[RT], vb[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    if (VB == 32) { va[rt] = wave >> 1; vb[rt] = 2 * (2 * (wave & 1) + rt) + pb; }
    else { va[rt] = (wave >> 1) >> 1; vb[rt] = 2 * (2 * ((wave >> 1) & 1) + rt) + pb; }
  }
  int lbase[RT], lpar[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    lbase[rt] = ((2 * va[rt] * DH_HB + 2 * vb[rt]) * DH_HC + 2 * pc) * 16;
    lpar[rt] = vb[rt] & 1;
  }
  const int jn0 = (VB == 32) ? 0 : (wave & 1) * 2;           // first 16-column tile of this wave
  const int wrd = r16 * 128, wswz = (r16 >> 1) & 7;
  const bool af32 = P.add_f32 != 0;
  const int ASZ = af32 ? 4 : 2;
  float bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = col0 + (jn0 + j) * 16 + 4 * q4 + e;
      bias[j][e] = (!BST && P.bias != nullptr && ch < P.Cn) ? P.bias[ch] : 0.f;      // (BST: no bias, host-checked — 16 registers)
    }
  float wsum[NT][4], wsq[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { wsum[j][e] = 0.f; wsq[j][e] = 0.f; }
  int stat_n = -1;
  auto flush_stats = [&](int n) {
    // each (channel) column is owned by the waves with the same jn0: VB = 32 all 8 waves; VB = 64 the 4 waves of a column half
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = wsum[j][e], b = wsq[j][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          const int c = (jn0 + j) * 16 + 4 * q4 + e;
          sStats[(wave * 2 + 0) * DH_CN + c] = a;
          sStats[(wave * 2 + 1) * DH_CN + c] = b;
        }
        wsum[j][e] = 0.f;
        wsq[j][e] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * DH_CN) {
      const int which = tid / DH_CN, c = tid % DH_CN;
      float a = 0.f;
      if (VB == 32) {
#pragma unroll
        for (int w = 0; w < DH_NTHR / 64; ++w) a += sStats[(w * 2 + which) * DH_CN + c];
      } else {
        const int par = (c >> 5) & 1;                           // column half -> waves with (wave & 1) == par
#pragma unroll
        for (int w = 0; w < 4; ++w) a += sStats[((2 * w + par) * 2 + which) * DH_CN + c];
      }
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      if (col0 + c < P.stats_ld) P.stats[(slot_t * 2 + which) * P.stats_ld + col0 + c] = a;
    }
    __syncthreads();
  };

  // ---- backward statistics (BST) ---------------------------------------------------------------------------------------
  f32x2 q1[4], q2[4];
  float q3 = 0.f;
#pragma unroll
  for (int h = 0; h < 4; ++h) { q1[h] = f32x2{0.f, 0.f}; q2[h] = f32x2{0.f, 0.f}; }
  const int ychunk = (q4 & 1) * 2 + (q4 >> 1);          // the 8-channel chunk (of 32 columns) a lane holds behind the permlane swap
  const float q_al = BST ? P.bst.alpha[0] : 1.f;
  const int y_sample_bytes = BST ? (int)(out_sample * P.bst.y_ld * 2) : 0;       // < 2^31 (host-checked)
  int yoff[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) yoff[rt] = BST ? ((va[rt] * G.oa + vb[rt] * G.ob + pc * G.oc) * P.bst.y_ld + 8 * ychunk) * 2 : 0;
  u32x4 yq[RT];
  int bst_n = -1;
  auto y_issue = [&](int n, int a0, int b0, int c0) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.bst.y) + (int64_t)n * y_sample_bytes, 0, y_sample_bytes, 0x00020000);
    const int soff = (a0 * G.oa + b0 * G.ob + c0 * G.oc) * P.bst.y_ld * 2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rv = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
      yq[rt] = __builtin_amdgcn_raw_buffer_load_b128(rs, rv ? yoff[rt] : (int)0x80000000, soff, 0);
    }
  };
  auto bst_consts = [&](int n) {       // (every thread: ends with a barrier)
    if (tid < 64) {
      const int c = tid & 31, which = tid >> 5;
      const float mean = P.bst.mr[((int64_t)n * P.bst.C + c) * 2], rstd = P.bst.mr[((int64_t)n * P.bst.C + c) * 2 + 1];
      sBt[which * 32 + c] = which == 0 ? rstd : -mean * rstd;
    }
    __syncthreads();
  };
  auto flush_bst = [&](int n) {        // (every thread: contains barriers)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float a = q1[h][e], b = q2[h][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          sStats[(wave * 3 + 0) * 32 + ychunk * 8 + 2 * h + e] = a;
          sStats[(wave * 3 + 1) * 32 + ychunk * 8 + 2 * h + e] = b;
        }
        q1[h][e] = 0.f; q2[h][e] = 0.f;
      }
    {
      float c = q3;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      if (r16 == 0) sStats[(wave * 3 + 2) * 32 + ychunk * 8] = c;      // the slope term: only its total over the channels matters
      q3 = 0.f;
    }
    __syncthreads();
    if (tid < 96) {
      const int which = tid >> 5, c = tid & 31;
      float a = 0.f;
      if (which < 2 || (c & 7) == 0) {
#pragma unroll
        for (int w = 0; w < DH_NTHR / 64; ++w) a += sStats[(w * 3 + which) * 32 + c];
      }
      P.bst.part[(((int64_t)n * P.bst.P + blockIdx.x) * 3 + which) * P.bst.ld + c] = a;
    }
    __syncthreads();
  };

  const int GX = gridDim.x;
  int first, stride, last;
  if ((GX & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    first = xcd * chunk + (blockIdx.x >> 3);
    stride = GX >> 3;
    last = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  } else {
    first = blockIdx.x; stride = GX; last = total_tiles;
  }

  constexpr bool RES = CF::RES;
  __syncthreads();
  int t = first;
  if (t < last) {
    gload(t);
    if constexpr (RES) {          // every K stage, once
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int rd = 0; rd < 512 / DH_NTHR; ++rd)
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ((int64_t)rd * 32 * kpad) * 2 + (int64_t)ks * 128),
                                           (lptr_t)(sW + ks * DH_KSB + wave * 1024 + rd * (DH_NTHR * 16)), 16, 0, 0);
    } else {
      wload(0, 0);
    }
    sstore();
  }
  __syncthreads();
  for (; t < last; t += stride) {
    const int tn = t + stride;
    int n, a0, b0, c0;
    tile_origin(t, n, a0, b0, c0);
    if (STATS && n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = n;
    }
    if constexpr (BST) {
      if (n != bst_n) {
        if (bst_n >= 0) flush_bst(bst_n);
        bst_n = n;
        bst_consts(n);
      }
      y_issue(n, a0, b0, c0);
    }
    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RES) { if (tn < last) gload(tn); }        // the next tile's halo has this tile's multiplies AND stores to arrive
#pragma unroll 1
    for (int rs = 0; rs < NRS; ++rs) {
      const int slot = wstage & 1;
      if constexpr (!RES) {
        if (rs + 1 < NRS) wload(rs + 1, slot ^ 1);
        else if (tn < last) wload(0, slot ^ 1);
        if (rs == 0 && tn < last) gload(tn);
      }
      const char* wb = RES ? sW + rs * (RG * DH_KSB) + wrd : sW + slot * (RG * DH_KSB) + wrd;
#pragma unroll
      for (int g = 0; g < RG; ++g) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int u = (rs * RG + g) * 2 + half;               // 32-wide k-step: K = 32 u + 8 q4 .. + 7
          if (u < NU) {
            const int kq = 32 * u + 8 * q4;
            const int tap = kq / CG, plane = (kq % CG) >> 3;
            const int tbase = sTab[tap] + plane * PLANE, tflag = sTab[32 + tap];
            const int step = (tflag & 0x100) ? -16 : 16;
            u32x4 xf[RT], wf[NT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              xf[rt] = *reinterpret_cast<const u32x4*>(sH + lbase[rt] + tbase + (((lpar[rt] ^ tflag) & 1) ? step : 0));
            const char* wk = wb + g * DH_KSB + (((4 * half + q4) ^ wswz) << 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(wk + (jn0 + j) * 16 * 128);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
              for (int j = 0; j < NT; ++j) mma16<H>(acc[rt][j], wf[j], xf[rt]);
          }
        }
      }
      ++wstage;
      if constexpr (!RES) __syncthreads();
    }
    // ---- epilogue: lane = (voxel (va, vb, pc) of row tile rt, channels (jn0 + j) * 16 + 4 q4 .. + 3) -----------------------
    const int64_t obase = n * out_sample + (int64_t)a0 * G.oa + (int64_t)b0 * G.ob + (int64_t)c0 * G.oc;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rv = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
      const int64_t vox = obase + (int64_t)va[rt] * G.oa + (int64_t)vb[rt] * G.ob + (int64_t)pc * G.oc;
      u32x2 oc[BST ? NT : 1];        // BST: the packed quads of the row tile, re-dealt into 16-byte chunks below
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int ch = col0 + (jn0 + j) * 16 + 4 * q4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[rt][j][e] + bias[j][e];
          if (STATS && rv) { wsum[j][e] += v[e]; wsq[j][e] += v[e] * v[e]; }
        }
        if constexpr (BST) {         // (no bias, no addend: host-checked)
          oc[j] = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
          continue;
        }
        u32x2 o2 = {0u, 0u};
        if (rv && ch < P.Cn_store) {
          if (P.add != nullptr) {
            const char* ap = P.add + (vox * P.add_ld + ch) * ASZ;
            if (af32) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap); v[0] += a4[0]; v[1] += a4[1]; v[2] += a4[2]; v[3] += a4[3]; }
            else {
              const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap);
              v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
            }
          }
          char* op = (P.out2 != nullptr && ch >= P.out2_col0) ? P.out2 + (vox * P.o2_ld + (ch - P.out2_col0)) * 2
                                                               : P.out + (vox * P.o_ld + ch) * 2;
          o2 = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
          *reinterpret_cast<u32x2*>(op) = o2;
        }
      }
      if constexpr (BST) {
        // v_permlane16_swap (conv_halo_sw.hip): the lanes of q4 = 0 / 2 get the neighbouring 4 channels of their own 16-column block,
        // those of q4 = 1 / 3 the ones of the next block: a lane holds 8 consecutive channels = one 16-byte store per 32 columns
#pragma unroll
        for (int jp = 0; jp < NT / 2; ++jp) {
          const auto s0 = __builtin_amdgcn_permlane16_swap(oc[2 * jp][0], oc[2 * jp + 1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(oc[2 * jp][1], oc[2 * jp + 1][1], false, false);
          const u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
          const int ch = col0 + jp * 32 + ychunk * 8;
          const bool st = rv && ch < P.Cn_store;
          if (st) {
            char* op = (P.out2 != nullptr && ch >= P.out2_col0) ? P.out2 + (vox * P.o2_ld + (ch - P.out2_col0)) * 2
                                                                 : P.out + (vox * P.o_ld + ch) * 2;
            *reinterpret_cast<u32x4*>(op) = o4;
          }
          if (jp == BJ / 2) {                                  // (g = 0 for the rows the tile does not own)
            const u32x4 gz = st ? o4 : u32x4{0u, 0u, 0u, 0u};
            const float* const cb = sBt + ychunk * 8;
            const f32x4 rs0 = *reinterpret_cast<const f32x4*>(cb), rs1 = *reinterpret_cast<const f32x4*>(cb + 4);
            const f32x4 nm0 = *reinterpret_cast<const f32x4*>(cb + 32), nm1 = *reinterpret_cast<const f32x4*>(cb + 36);
            bst_pair_bf16(gz[0], yq[rt][0], f32x2{rs0[0], rs0[1]}, f32x2{nm0[0], nm0[1]}, q_al, q1[0], q2[0], q3);
            bst_pair_bf16(gz[1], yq[rt][1], f32x2{rs0[2], rs0[3]}, f32x2{nm0[2], nm0[3]}, q_al, q1[1], q2[1], q3);
            bst_pair_bf16(gz[2], yq[rt][2], f32x2{rs1[0], rs1[1]}, f32x2{nm1[0], nm1[1]}, q_al, q1[2], q2[2], q3);
            bst_pair_bf16(gz[3], yq[rt][3], f32x2{rs1[2], rs1[3]}, f32x2{nm1[2], nm1[3]}, q_al, q1[3], q2[3], q3);
          }
        }
      }
    }
    if (tn < last) {
      if constexpr (RES) __syncthreads();                    // resident weights: no ring barrier behind the last K stage — every wave
                                                             // must have left the multiplies before the (single) halo image is overwritten
      sstore();                                              // (ring: every wave passed the last ring barrier: halo is free)
      __syncthreads();
    }
  }
  if (STATS && stat_n >= 0) flush_stats(stat_n);
  if constexpr (BST) { if (bst_n >= 0) flush_bst(bst_n); }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static void down_geom(const ConvKArgs& a, int vb, DownGeom& g) {
  const int ta = 4;
  const int od[3] = {a.Xr, a.Yr, a.Zr}, id[3] = {a.Xi, a.Yi, a.Zi};
  const int istr[3] = {a.Yi * a.Zi, a.Zi, 1}, ostr[3] = {a.Yo * a.Zo, a.Zo, 1};
  auto waste = [&](int pa, int pb, int pc) {
    const double padded = (double)((od[pa] + ta - 1) / ta * ta) * ((od[pb] + 7) / 8 * 8) * ((od[pc] + 7) / 8 * 8);
    return padded / ((double)od[0] * od[1] * od[2]);
  };
  int pa = 0, pb = 1, pc = 2;
  if (waste(2, 0, 1) < waste(0, 1, 2) - 1e-9) { pa = 2; pb = 0; pc = 1; }
  g.pa = pa; g.pb = pb; g.pc = pc;
  g.da = od[pa]; g.db = od[pb]; g.dc = od[pc];
  g.xa = id[pa]; g.xb = id[pb]; g.xc = id[pc];
  g.ia = istr[pa]; g.ib = istr[pb]; g.ic = istr[pc];
  g.oa = ostr[pa]; g.ob = ostr[pb]; g.oc = ostr[pc];
  g.tbn = (g.db + 7) / 8; g.tcn = (g.dc + 7) / 8;
  g.tiles = ((g.da + ta - 1) / ta) * g.tbn * g.tcn;
}

bool conv_down_halo_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (!is16(dtype) || a.out_f32 || nclass != 1 || a.sin != 2 || a.sout != 1) return false;
  const int vb = a.Cg * 2;
  // 32 gathered channels -> 128 columns was measured SLOWER than the generic kernel (128-row tiles do not amortise the 221 KB
  // weight stream): only the 16-channel case runs here
  if (vb != 32 || a.Cn < 48 || (a.Cn_store % 4) != 0) return false;
  if (((a.g_ld % 8) != 0 && a.g_ld != 12) || ((uintptr_t)a.in % 16) != 0 || ((uintptr_t)a.w % 16) != 0) return false;   // 12: 8-byte pieces
  if (a.g_ld == 12 && (a.Xi != 2 * a.Xr || a.Yi != 2 * a.Yr || a.Zi != 2 * a.Zr)) return false;        // (only the -1 taps leave the volume)
  if (a.Xo != a.Xr || a.Yo != a.Yr || a.Zo != a.Zr) return false;
  if (2 * a.Xr - 1 > a.Xi + 1 || 2 * a.Yr - 1 > a.Yi + 1 || 2 * a.Zr - 1 > a.Zi + 1) return false;   // rows whose centre lies outside
  if ((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2 >= (1ll << 31)) return false;
  if ((int64_t)a.Xr * a.Yr * a.Zr < 2048) return false;
  const ctseg_conv_class& k = a.cls[0];
  if (k.ntaps != 27 || k.kpad < 27 * a.Cg || (k.kpad % 64) != 0 || (k.w_off % 8) != 0) return false;
  for (int j = 0; j < 27; ++j)
    for (int s = 0; s < 24; s += 8) {
      const int d = (int)(int8_t)((k.taps[j] >> s) & 0xff);
      if (d < -1 || d > 1) return false;
    }
  return true;
}

static int down_grid(const ConvKArgs& a, const DownGeom& g) {
  return persistent_grid(CTSEG_NUM_CU, g.tiles * a.N);
}

int conv_down_halo_slots(const ConvKArgs& a) {
  DownGeom g;
  down_geom(a, a.Cg * 2, g);
  return down_grid(a, g);
}

// ConvKArgs::bst on this pass: the second 32 of its 64 written columns (the sub-block half of the level-0 concat gradient), bf16
int conv_down_halo_bst_slots(const ConvKArgs& a) {
  { const char* e = getenv("CTSEG_BST_DOWN"); if (e != nullptr && e[0] == '0') return 0; }   // (A/B switch)
  // (12-wide gathered rows only: the variant staging 16-byte chunks is at the 256-register line without the sums)
  if (a.dtype != CTSEG_BF16 || a.g_ld != 12 || a.stats != nullptr || a.bias != nullptr || a.add != nullptr || a.Cn > DH_CN || a.bst.C != 32 || a.bst.col0 != 32 || a.Cn_store < 64) return 0;
  // 16-byte chunks of 8 channels: the written tensor(s) and y
  if ((a.bst.y_ld % 8) != 0 || ((uintptr_t)a.bst.y % 16) != 0 || (a.o_ld % 8) != 0 || ((uintptr_t)a.out % 16) != 0 || (a.Cn_store % 8) != 0) return 0;
  if (a.out2 != nullptr && ((a.o2_ld % 8) != 0 || (a.out2_col0 % 8) != 0 || ((uintptr_t)a.out2 % 16) != 0)) return 0;
  if ((int64_t)a.Xo * a.Yo * a.Zo * a.bst.y_ld * 2 >= (1ll << 31)) return 0;
  return conv_down_halo_slots(a);
}

void launch_conv_down_halo(ConvKArgs& a, hipStream_t st) {
  DownGeom g;
  const int vb = a.Cg * 2;
  down_geom(a, vb, g);
  a.tiles = g.tiles;
  const int total = g.tiles * a.N;
  const dim3 grid((unsigned)down_grid(a, g), (unsigned)((a.Cn + DH_CN - 1) / DH_CN)), blk(DH_NTHR);
  const bool stats = a.stats != nullptr;
#define CTSEG_DH_GO(H, ST, R) hipLaunchKernelGGL((conv_down_halo_kernel<H, 32, ST, R>), grid, blk, 0, st, a, g, total)
  const bool r12 = a.g_ld == 12;
  if (a.bst.part != nullptr) {        // (bf16, 12-wide gathered rows, no forward statistics, columns 32..63: conv_down_halo_bst_slots)
    hipLaunchKernelGGL((conv_down_halo_kernel<BF16, 32, false, true, 2>), grid, blk, 0, st, a, g, total);
    return;
  }
  if (a.dtype == CTSEG_F16) {
    if (stats) { if (r12) CTSEG_DH_GO(F16, true, true); else CTSEG_DH_GO(F16, true, false); }
    else { if (r12) CTSEG_DH_GO(F16, false, true); else CTSEG_DH_GO(F16, false, false); }
  } else {
    if (stats) { if (r12) CTSEG_DH_GO(BF16, true, true); else CTSEG_DH_GO(BF16, true, false); }
    else { if (r12) CTSEG_DH_GO(BF16, false, true); else CTSEG_DH_GO(BF16, false, false); }
  }
#undef CTSEG_DH_GO
}

}  // namespace ctseg
