// LDS-halo implicit-GEMM kernel with STREAMED weights for the 64- and 128-channel 3x3x3 passes (gfx950, bf16):
//   * Conv3d 64 -> 64 k3 s1 forward / input gradient at 128x128x12            (one class, 27 taps, Cg = 64,  Cn = 64)
//   * the 8-parity-class stride-2 passes with Cg = 128 -> Cn = 32 at 128x128x12 (ConvTranspose3d 128 -> 32 forward and the
//     input gradient of the fused stride-2 [residual | unit0] convolution 32 -> 64+64): 1+2+2+2+4+4+4+8 = 27 taps
// of the reference's U-Net (MONAI UNet built at capstone/volumetric/base_trainer.py:65-72).
//
// The generic kernel (conv_igemm.hip) re-gathers every im2col row chunk through L2 for each of its 27 taps; at these channel
// counts it is bound by L2 -> LDS operand traffic (43 FLOP/B with the 128x64 tile, 29 FLOP/B with 256x32), not by MFMA or HBM.
// Here one 8-wave workgroup per CU stages the input halo of a TXx8x8 tile ONCE (conv_halo.hip's plane layout and lane<->voxel
// permutation: conflict-free ds_read_b128 for every tap) and streams the packed weights of the 27 (class, tap) pairs through a
// two-slot LDS ring, three taps (24 KB) per stage, with direct-to-LDS loads (global_load_lds, swizzle applied on the source
// side).  Every workgroup walks the same weight sequence, so the stream is an L2 hit; per tile the CU moves 77-102 KB of halo +
// 221 KB of weights for 57 (28) MFLOP: >100 FLOP per operand byte.
//
// The tile's long axes can be mapped to any pair of volume axes (`perm`): the reference's 12-deep level would waste a third of
// an 8-deep z tile, so there the short tile axis (4 or 2) runs along z and the 8x8 face covers (x, y).
#include "conv_common.h"

#ifndef SW_STORE8
#define SW_STORE8 0
#endif
#ifndef SW_FULL_WAIT
#define SW_FULL_WAIT 0
#endif
#ifndef SW_ABL
#define SW_ABL 0     // timing-only ablation: 1 no MFMAs, 2 no LDS operand reads, 4 no weight stream, 8 no output stores (and no addend loads), 16 no halo loads, 32 no addend loads, 64 BST without the y loads, 128 BST without its arithmetic
#endif

namespace ctseg {

constexpr int SW_NTHR = 512, SW_TAPB = 8192;

// The 128-channel "up" pass (UP, VB = 256) stores only the halo region its taps (offsets 0 / +1) can read — (TA + 1) x 9 rows of 10
// slots — which lets the 4-deep tile (two row tiles per wave: every weight fragment read from LDS feeds twice the MFMAs, half the
// weight stream and half the stage barriers per voxel) fit next to a two-tap ring stage: 16 planes x 456 slots + 32 KB = 149 KB.
// Planes are padded to 8 (mod 16) slots so that the two planes a ds_read_b128 lane group mixes sit on opposite halves of the
// 256-byte bank row (the 400-slot planes of the 2-deep tile are 0 (mod 16): 2-way conflicts on every operand read).
template <int VB, bool UP> struct SwCfg {
  static constexpr bool DENSE = UP && VB == 256;        // halo image holds coordinates 1.. only
  static constexpr int TA = (VB == 128 || DENSE) ? 4 : 2;          // short tile axis
  static constexpr int O = DENSE ? 1 : 0;               // first stored halo coordinate
  static constexpr int LB = 10 - O;                     // stored rows per plane of the short axis
  static constexpr int HV = DENSE ? 456 : (TA + 2) * 100;   // halo slots per 16-byte channel plane
  static constexpr int NPL = VB / 16, PLANE = HV * 16, HALO = NPL * PLANE;
  static constexpr int CG = VB / 2, CN = SW_TAPB / VB;  // 64 -> 64, 128 -> 32
  static constexpr int NT = CN / 16;
  static constexpr int RT = TA / 2;                     // row tiles (16 voxels) per wave
  static constexpr int KS = CG / 32;                    // 32-wide k-steps per tap
  static constexpr int G = DENSE ? 2 : 3;               // taps per ring stage
  static constexpr int NSTAGE = (27 + G - 1) / G;
  static constexpr int WRING = 2 * G * SW_TAPB;
  static constexpr int TOTAL = HALO + WRING + 8 * 3 * CN * 4 + 128 * 4 + 2 * CN * 4;     // ... + partial sums of the waves + tap tables + BST constants
  static_assert(TOTAL <= 160 * 1024, "LDS");
  __host__ __device__ static constexpr int slot(int ha, int hb, int hc) { return ((ha - O) * LB + (hb - O)) * 10 + hc; }
};

struct SwGeom {
  int da, db, dc;            // extents of the row grid along the tile axes (a = short axis)
  int ia, ib, ic;            // gathered-tensor voxel strides of the tile axes
  int oa, ob, oc;            // written-tensor voxel strides of the tile axes (already multiplied by sout)
  int pa, pb, pc;            // which volume axis (0 = x, 1 = y, 2 = z) each tile axis runs along
  int tbn, tcn, tiles;       // tiles along b, c; tiles per sample
};

__device__ __forceinline__ void sw_patch_voxel(int r16, int& db, int& c) {
  db = (0xEF80u >> r16) & 1;
  c = (int)((0x2104765437653210ull >> (4 * r16)) & 7ull);
}

// BST: backward InstanceNorm statistics of the written gradient (ConvKArgs::bst) on the input-gradient passes (8-class 128 -> 32,
// single-class 64 -> 64).  After the v_permlane16_swap of the store a lane holds 16-byte chunks (8 consecutive channels) of one
// voxel: the matching y chunks are requested one class ahead, beside the addend of that class (8 classes), or at the top of the
// tile's epilogue (one class: they arrive under the bias / addend / pack arithmetic), and the lane keeps the sums of its 8 or 16
// channels (packed pairs) over all classes and tiles of a sample; combined per workgroup at every sample change, like the forward
// statistics.
template <typename H, int VB, bool UP, bool STATS, bool BST = false>     // H = 16-bit storage kind (BF16 / F16)
__global__ __launch_bounds__(SW_NTHR) void conv_halo_sw_kernel(const ConvKArgs P, const SwGeom G, int total_tiles) {
  static_assert(!BST || (!STATS && ((UP && VB == 256) || (!UP && VB == 128))), "backward statistics: the input-gradient passes this kernel takes");
  using CF = SwCfg<VB, UP>;
  constexpr int TA = CF::TA, NPL = CF::NPL, PLANE = CF::PLANE, CN = CF::CN, NT = CF::NT, RT = CF::RT, KS = CF::KS;
  constexpr int SW_G = CF::G, SW_NSTAGE = CF::NSTAGE;
  constexpr int O = UP ? 1 : 0;                                        // first halo coordinate that is ever read
  constexpr int FA = TA + 2 - O, FB = 10 - O, FV = FA * FB * FB;       // filled region of the halo image
  constexpr int NCH = FV * NPL, J = (NCH + SW_NTHR - 1) / SW_NTHR;

  __shared__ __attribute__((aligned(16))) char smem[CF::TOTAL];
  char* const sH = smem;
  char* const sW = smem + CF::HALO;
  float* const sStats = reinterpret_cast<float*>(sW + CF::WRING);
  int* const sTab = reinterpret_cast<int*>(sStats + 8 * 3 * CN);       // [0,32) halo delta, [32,64) weight offset, [64,96) kpad, [96,128) class | last<<8
  float* const sBt = reinterpret_cast<float*>(sTab + 128);             // BST: [0,CN) rstd, [CN,2 CN) -mean * rstd of the sample's channels

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;

  if (tid < 32) {
    int d = 0, wo = 0, kp = 0, cl = 0;
    if (tid < 27) {
      int c = 0, first = 0;
      while (tid >= first + P.cls[c].ntaps) { first += P.cls[c].ntaps; ++c; }
      const int ti = tid - first;
      const int tp = P.cls[c].taps[ti];
      int dv[3] = {(int)(int8_t)(tp & 0xff), (int)(int8_t)((tp >> 8) & 0xff), (int)(int8_t)((tp >> 16) & 0xff)};
      d = ((dv[G.pa] * CF::LB + dv[G.pb]) * 10 + dv[G.pc]) * 16;
      wo = (int)(P.cls[c].w_off + (int64_t)ti * CF::CG);
      kp = P.cls[c].kpad;
      cl = c | ((ti == P.cls[c].ntaps - 1) ? 256 : 0);
    }
    sTab[tid] = d; sTab[32 + tid] = wo; sTab[64 + tid] = kp; sTab[96 + tid] = cl;
  }

  // ---- halo staging slots: per-thread constants, only a scalar tile base changes ---------------------------------------
  // LEAN (the BST variant, which sits at the 256-register line): only the packed halo coordinates stay in registers; the source and
  // LDS offsets (2 x J registers) are recomputed from them for every tile (a dozen multiply-adds per slot against a 40 us tile).
  // The channel plane of a slot does not depend on j there (512 threads = 64 voxel octets x 16 planes ... SW_NTHR / 8 % NPL == 0).
  constexpr bool LEAN = BST;
  static_assert(!LEAN || (SW_NTHR / 8) % NPL == 0, "lean staging: the plane of a slot must not depend on j");
  int g_byte[LEAN ? 1 : J], g_habc[J], g_lds[LEAN ? 1 : J];
  const int pl0 = (tid >> 3) % NPL;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int idx = tid + j * SW_NTHR;
    const int pl = (idx >> 3) % NPL, fv = (idx / (8 * NPL)) * 8 + (idx & 7);
    const int fa = fv / (FB * FB), rem = fv - fa * (FB * FB), fb = rem / FB, fc = rem - fb * FB;
    const int ha = fa + O, hb = fb + O, hc = fc + O;                   // halo coordinates, tile origin = (1,1,1)
    g_habc[j] = (fv < FV) ? (ha | (hb << 8) | (hc << 16)) : 0x7f7f7f;  // sentinel fails every bounds test
    if constexpr (!LEAN) {
      g_byte[j] = (((ha - 1) * G.ia + (hb - 1) * G.ib + (hc - 1) * G.ic) * P.g_ld + pl * 8) * 2;
      g_lds[j] = pl * PLANE + CF::slot(ha, hb, hc) * 16;
    }
  }
  auto lean_habc = [&](int j, int& ha, int& hb, int& hc) {     // (opaque to the optimiser: otherwise it hoists the offsets back into registers)
    int h = g_habc[j];
    asm volatile("" : "+v"(h));
    ha = h & 0xff; hb = (h >> 8) & 0xff; hc = h >> 16;
  };
  auto tile_origin = [&](int t, int& n, int& a0, int& b0, int& c0) {
    n = t / G.tiles;
    int r = t - n * G.tiles;
    const int tc = r % G.tcn; r /= G.tcn;
    const int tb = r % G.tbn; const int ta = r / G.tbn;
    a0 = ta * TA; b0 = tb * 8; c0 = tc * 8;
  };
  const int64_t in_sample = (int64_t)P.Xi * P.Yi * P.Zi, out_sample = (int64_t)P.Xo * P.Yo * P.Zo;
  u32x4 rh[J];
  // (branch-free raw buffer loads measured no faster here: 0.171 -> 0.182 ms on the 128 -> 32 up pass, 0.094 -> 0.103 on 64 -> 64)
  auto gload = [&](int t) {
    int n, a0, b0, c0;
    tile_origin(t, n, a0, b0, c0);
    const char* base = P.in + (n * in_sample + (int64_t)a0 * G.ia + (int64_t)b0 * G.ib + (int64_t)c0 * G.ic) * P.g_ld * 2;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int ai = a0 - 1 + (g_habc[j] & 0xff), bi = b0 - 1 + ((g_habc[j] >> 8) & 0xff), ci = c0 - 1 + (g_habc[j] >> 16);
      u32x4 v = {0u, 0u, 0u, 0u};
      int gb;
      if constexpr (LEAN) {
        int ha, hb, hc;
        lean_habc(j, ha, hb, hc);
        gb = (((ha - 1) * G.ia + (hb - 1) * G.ib + (hc - 1) * G.ic) * P.g_ld + pl0 * 8) * 2;
      } else {
        gb = g_byte[j];
      }
      if (!(SW_ABL & 16) && (unsigned)ai < (unsigned)G.da && (unsigned)bi < (unsigned)G.db && (unsigned)ci < (unsigned)G.dc)
        v = *reinterpret_cast<const u32x4*>(base + gb);
      rh[j] = v;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      int gl;
      if constexpr (LEAN) {
        int ha, hb, hc;
        lean_habc(j, ha, hb, hc);
        gl = pl0 * PLANE + CF::slot(ha, hb, hc) * 16;
      } else {
        gl = g_lds[j];
      }
      if ((g_habc[j] & 0xff) != 0x7f) *reinterpret_cast<u32x4*>(sH + gl) = rh[j];
    }
  };

  // ---- weight stream: thread = one 16-byte chunk of a tap's [CN][CG] block, lane-linear LDS image, source-side swizzle -------
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int w_s64 = tid / (CN * 8), w_row = (tid >> 3) % CN, w_q8 = (tid & 7) ^ ((w_row >> 1) & 7);
  const int w_koff = w_s64 * 64 + w_q8 * 8;
  int wstage = 0;                                    // running stage counter: ring slot = wstage & 1
  auto wload = [&](int stage_in_tile, int slot) {
    char* dst = sW + slot * (SW_G * SW_TAPB) + wave * 1024;
#pragma unroll
    for (int g = 0; g < SW_G; ++g) {
      const int tap = stage_in_tile * SW_G + g;
      if (SW_G * SW_NSTAGE > 27 && tap >= 27) continue;      // the last stage of the two-tap ring holds one tap
      if (SW_ABL & 4) continue;
      const char* src = P.w + ((int64_t)sTab[32 + tap] + (int64_t)w_row * sTab[64 + tap] + w_koff) * 2;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + g * SW_TAPB), 16, 0, 0);
    }
  };

  // ---- per-lane constants of the MFMA operands and the epilogue -------------------------------------------------------
  int pdb, pc;
  sw_patch_voxel(r16, pdb, pc);
  // row tile rt of wave w: a = wa, b in {2i, 2i+1}, c 0..7 (permuted inside the 2x8 patch)
  int abase[RT], va[RT], vb[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int wa = (TA == 4) ? (wave >> 1) : (wave >> 2);
    const int i = (TA == 4) ? (2 * (wave & 1) + rt) : (wave & 3);
    va[rt] = wa;
    vb[rt] = 2 * i + pdb;
    abase[rt] = CF::slot(wa + 1, vb[rt] + 1, pc + 1) * 16 + q4 * PLANE;
  }
  const int wrd = r16 * 128, wswz = (r16 >> 1) & 7;
  const bool af32 = P.add_f32 != 0;
  const int ASZ = af32 ? 4 : 2;
  float bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[j][e] = (!BST && P.bias != nullptr) ? P.bias[j * 16 + 4 * q4 + e] : 0.f;     // (BST: no bias, host-checked)

  float wsum[NT][4], wsq[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { wsum[j][e] = 0.f; wsq[j][e] = 0.f; }
  int stat_n = -1;
  auto flush_stats = [&](int n) {                    // called by every thread (contains barriers)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = wsum[j][e], b = wsq[j][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          const int c = j * 16 + 4 * q4 + e;
          sStats[(wave * 2 + 0) * CN + c] = a;
          sStats[(wave * 2 + 1) * CN + c] = b;
        }
        wsum[j][e] = 0.f;
        wsq[j][e] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * CN) {
      const int which = tid / CN, c = tid % CN;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += sStats[(w * 2 + which) * CN + c];
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      P.stats[(slot_t * 2 + which) * P.stats_ld + c] = a;
    }
    __syncthreads();
  };

  f32x4 acc[RT][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // Epilogue of one class straight from the accumulators: lane = (voxel of row tile rt, channels j*16 + 4*q4 .. +3).
  // Stores and addend loads are raw buffer operations with a per-sample descriptor: lanes outside the volume get an out-of-range
  // offset, so every wave issues the SAME number of vector-memory operations per epilogue and the stage barrier can wait for the
  // weight DMA alone (s_waitcnt vmcnt(<operations issued after it>)) instead of draining the stores it has just issued.  The bf16 /
  // fp16 addend of the NEXT class is requested in the epilogue of the current one (of the first class: at the top of the tile) and
  // has a whole class of multiplies to arrive.
  const int out_sample_bytes = (int)(out_sample * P.o_ld * 2), add_sample_bytes = (int)(out_sample * P.add_ld * ASZ);
  const bool apf = UP && !STATS && P.add != nullptr && !af32;    // prefetched 16-bit addend (the statistics variants have no registers to spare)
  int ooff[RT], aoff[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int vox = va[rt] * G.oa + vb[rt] * G.ob + pc * G.oc;
    ooff[rt] = (vox * P.o_ld + 4 * q4) * 2;
    aoff[rt] = (vox * P.add_ld + 4 * q4) * ASZ;
  }
  u32x2 apre[RT][NT];
  // ---- backward statistics (BST) --------------------------------------------------------------------------------------------
  constexpr int NQ = NT / 2;                            // 16-byte chunks (of 32 channels each ... chunk jp = channels 32 jp + 8 ychunk ..) per lane
  f32x2 q1[NQ][4], q2[NQ][4];
  float q3 = 0.f;
#pragma unroll
  for (int jp = 0; jp < NQ; ++jp)
#pragma unroll
    for (int h = 0; h < 4; ++h) { q1[jp][h] = f32x2{0.f, 0.f}; q2[jp][h] = f32x2{0.f, 0.f}; }
  const float q_al = BST ? P.bst.alpha[0] : 1.f;
  const int ychunk = (q4 & 1) * 2 + (q4 >> 1);          // the 8-channel chunk a lane holds behind the permlane swap
  const int y_sample_bytes = BST ? (int)(out_sample * P.bst.y_ld * 2) : 0;     // < 2^31 (host-checked)
  int yoffc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) yoffc[rt] = BST ? ((va[rt] * G.oa + vb[rt] * G.ob + pc * G.oc) * P.bst.y_ld + ychunk * 8) * 2 : 0;
  u32x4 ypre[RT];
  int bst_n = -1;
  auto bst_consts = [&](int n) {       // (every thread: ends with a barrier)
    if (tid < 2 * CN) {
      const int c = tid % CN, which = tid / CN;
      const float mean = P.bst.mr[((int64_t)n * P.bst.C + c) * 2], rstd = P.bst.mr[((int64_t)n * P.bst.C + c) * 2 + 1];
      sBt[which * CN + c] = which == 0 ? rstd : -mean * rstd;
    }
    __syncthreads();
  };
  auto bst_chunk = [&](int jp, const u32x4& gz, const u32x4& yv) {      // 8 channels of one voxel: stored gradient, forward y
    if (SW_ABL & 128) { q3 += __uint_as_float(gz[0] ^ yv[3]); return; }
    const float* const cb = sBt + jp * 32 + ychunk * 8;
    const f32x4 rs0 = *reinterpret_cast<const f32x4*>(cb), rs1 = *reinterpret_cast<const f32x4*>(cb + 4);
    const f32x4 nm0 = *reinterpret_cast<const f32x4*>(cb + CN), nm1 = *reinterpret_cast<const f32x4*>(cb + CN + 4);
    bst_pair_bf16(gz[0], yv[0], f32x2{rs0[0], rs0[1]}, f32x2{nm0[0], nm0[1]}, q_al, q1[jp][0], q2[jp][0], q3);
    bst_pair_bf16(gz[1], yv[1], f32x2{rs0[2], rs0[3]}, f32x2{nm0[2], nm0[3]}, q_al, q1[jp][1], q2[jp][1], q3);
    bst_pair_bf16(gz[2], yv[2], f32x2{rs1[0], rs1[1]}, f32x2{nm1[0], nm1[1]}, q_al, q1[jp][2], q2[jp][2], q3);
    bst_pair_bf16(gz[3], yv[3], f32x2{rs1[2], rs1[3]}, f32x2{nm1[2], nm1[3]}, q_al, q1[jp][3], q2[jp][3], q3);
  };
  auto flush_bst = [&](int n) {        // (every thread: contains barriers)
#pragma unroll
    for (int jp = 0; jp < NQ; ++jp)
#pragma unroll
      for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float a = q1[jp][h][e], b = q2[jp][h][e];
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
          if (r16 == 0) {
            sStats[(wave * 3 + 0) * CN + jp * 32 + ychunk * 8 + 2 * h + e] = a;
            sStats[(wave * 3 + 1) * CN + jp * 32 + ychunk * 8 + 2 * h + e] = b;
          }
          q1[jp][h][e] = 0.f; q2[jp][h][e] = 0.f;
        }
    {
      float c = q3;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      if (r16 == 0) sStats[(wave * 3 + 2) * CN + ychunk * 8] = c;      // the slope term: only its total over the channels matters
      q3 = 0.f;
    }
    __syncthreads();
    if (tid < 3 * CN) {
      const int which = tid / CN, c = tid % CN;
      float a = 0.f;
      if (which < 2 || ((c & 7) == 0 && c < 32)) {
#pragma unroll
        for (int w = 0; w < 8; ++w) a += sStats[(w * 3 + which) * CN + c];
      }
      P.bst.part[(((int64_t)n * P.bst.P + blockIdx.x) * 3 + which) * P.bst.ld + c] = a;
    }
    __syncthreads();
  };
  int pend = 0;                                         // vector-memory operations issued since this stage's weight DMA (wave-uniform)
  auto class_base = [&](int cls, int a0, int b0, int c0) -> int {
    const ctseg_conv_class& K = P.cls[cls];
    return a0 * G.oa + b0 * G.ob + c0 * G.oc + (K.ox * P.Yo + K.oy) * P.Zo + K.oz;
  };
  auto add_issue = [&](int cls, int n, int a0, int b0, int c0) {        // cls < 0: nothing to fetch (same operation count)
    const int cb = cls >= 0 ? class_base(cls, a0, b0, c0) : 0;
    if (apf) {
      const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.add) + (int64_t)n * add_sample_bytes, 0, add_sample_bytes, 0x00020000);
      const int soff = cb * P.add_ld * ASZ;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool rv = cls >= 0 && (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
#pragma unroll
        for (int j = 0; j < NT; ++j)
          apre[rt][j] = (SW_ABL & 32) ? u32x2{0u, 0u} : __builtin_amdgcn_raw_buffer_load_b64(ars, rv ? aoff[rt] + j * 16 * ASZ : (int)0x80000000, soff, 0);
      }
      pend += RT * NT;
    }
    if constexpr (BST) {             // the y chunks of the class's voxels
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.bst.y) + (int64_t)n * y_sample_bytes, 0, y_sample_bytes, 0x00020000);
      const int soff = cb * P.bst.y_ld * 2;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const bool rv = cls >= 0 && (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
        ypre[rt] = (SW_ABL & 64) ? u32x4{1u, 2u, 3u, (uint32_t)soff} : __builtin_amdgcn_raw_buffer_load_b128(yrs, rv ? yoffc[rt] : (int)0x80000000, soff, 0);
      }
      pend += RT;
    }
  };
  auto epilogue = [&](int cls, int next_cls, int n, int a0, int b0, int c0) {
    const int cb = class_base(cls, a0, b0, c0);
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(P.out + (int64_t)n * out_sample_bytes, 0, out_sample_bytes, 0x00020000);
    const int soff = cb * P.o_ld * 2;
    // 16-byte stores: v_permlane16_swap hands the lanes of q4 = 0 / 2 the neighbouring 4 channels of their own 16-column block and the
    // lanes of q4 = 1 / 3 those of the next block, so a lane holds 8 consecutive channels (16-byte chunk {0, 2, 1, 3}[q4] of 32 channels)
    const int cbq = ((q4 & 1) * 2 + (q4 >> 1)) * 16;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rv = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
      u32x2 o2[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[rt][j][e] + bias[j][e];
          if (STATS && rv) { wsum[j][e] += v[e]; wsq[j][e] += v[e] * v[e]; }
        }
        if (apf) {
          const u32x2 w2 = apre[rt][j];
          v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
        } else if (P.add != nullptr && rv) {      // fp32 addend, or a statistics pass with an addend: fetched here
          const char* ap = P.add + (int64_t)n * add_sample_bytes + (int64_t)cb * P.add_ld * ASZ + aoff[rt] + j * 16 * ASZ;
          if (af32) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap); v[0] += a4[0]; v[1] += a4[1]; v[2] += a4[2]; v[3] += a4[3]; }
          else {
            const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap);
            v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
          }
        }
        o2[j] = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
      }
#if SW_STORE8
#pragma unroll
      for (int j = 0; j < NT; ++j)
        __builtin_amdgcn_raw_buffer_store_b64(o2[j], ors, rv ? ooff[rt] + j * 32 : (int)0x80000000, soff, 0);
      continue;
#endif
#pragma unroll
      for (int jp = 0; jp < NT / 2; ++jp) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(o2[2 * jp][0], o2[2 * jp + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(o2[2 * jp][1], o2[2 * jp + 1][1], false, false);
        const u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
        const bool keep = rv && (!(SW_ABL & 8) || o4[0] == 0x12345u);
        __builtin_amdgcn_raw_buffer_store_b128(o4, ors, keep ? ooff[rt] - 4 * q4 * 2 + jp * 64 + cbq : (int)0x80000000, soff, 0);
        // The data registers of a 16-byte buffer store must not be rewritten in the next two issue slots (the ">64-bit store data"
        // hazard): the compiler does not pad it when the store has an SGPR offset, and in the statistics variant it scheduled a
        // v_mov into the first data register right behind the store — a few 16-bit elements per launch came out as garbage.
        // Keeping the registers live across two wait states closes the window.
        asm volatile("s_nop 1" ::"v"(o4));
        if constexpr (BST) bst_chunk(jp, rv ? o4 : u32x4{0u, 0u, 0u, 0u}, ypre[rt]);   // (NT = 2: one chunk per row tile; g = 0 for voxels the tile does not own)
      }
    }
    pend += RT * NT / 2;
    if (apf || BST) add_issue(next_cls, n, a0, b0, c0);
  };
  // stage barrier: the next stage's weight DMA (issued first in this stage) has landed, later stores / addend loads may still fly
  auto stage_barrier = [&]() {
    // one epilogue without / with the next addend request (BST: + the y chunks, with or without an addend)
    constexpr int YL = BST ? RT : 0, E1 = RT * NT / 2 + YL, E2 = RT * NT / 2 + RT * NT + YL;
    constexpr int T1 = YL, T2 = RT * NT + YL;                        // BST: the first class's request at the top of a tile (no stores)
    if (SW_FULL_WAIT || pend == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (BST && pend == T1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(T1) : "memory");
    else if (BST && pend == T2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(T2) : "memory");
    else if (BST && pend == T1 + E1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(T1 + E1) : "memory");
    else if (BST && pend == T2 + E2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(T2 + E2) : "memory");
    else if (pend == E1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(E1) : "memory");
    else if (pend == 2 * E1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * E1) : "memory");
    else if (pend == E2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(E2) : "memory");
    else if (pend == 2 * E2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * E2) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pend = 0;
  };

  // single-class passes (one epilogue per tile): plain global stores, addend fetched in place.  lane: lane = (voxel of row tile rt, channels j*16 + 4*q4 .. +3)
  auto epilogue_plain = [&](int cls, int n, int a0, int b0, int c0) {
    const ctseg_conv_class& K = P.cls[cls];
    int ov[3] = {K.ox, K.oy, K.oz};
    const int64_t obase = n * out_sample + (int64_t)a0 * G.oa + (int64_t)b0 * G.ob + (int64_t)c0 * G.oc +
                          ((int64_t)ov[0] * P.Yo + ov[1]) * P.Zo + ov[2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rv = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
      const int64_t vox = obase + (int64_t)va[rt] * G.oa + (int64_t)vb[rt] * G.ob + (int64_t)pc * G.oc;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int ch = j * 16 + 4 * q4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[rt][j][e] + bias[j][e];
          if (STATS && rv) { wsum[j][e] += v[e]; wsq[j][e] += v[e] * v[e]; }
        }
        if (rv) {
          if (P.add != nullptr) {
            const char* ap = P.add + (vox * P.add_ld + ch) * ASZ;
            if (af32) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap); v[0] += a4[0]; v[1] += a4[1]; v[2] += a4[2]; v[3] += a4[3]; }
            else {
              const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap);
              v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
            }
          }
          *reinterpret_cast<u32x2*>(P.out + (vox * P.o_ld + ch) * 2) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
        }
      }
    }
  };

  // single-class pass with BST: 16-byte chunks as above (buffer stores with a per-sample descriptor), the y chunks requested first
  auto epilogue_plain_bst = [&](int n, int a0, int b0, int c0) {
    const ctseg_conv_class& K = P.cls[0];
    const int cb = a0 * G.oa + b0 * G.ob + c0 * G.oc + (K.ox * P.Yo + K.oy) * P.Zo + K.oz;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(P.out + (int64_t)n * out_sample_bytes, 0, out_sample_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.bst.y) + (int64_t)n * y_sample_bytes, 0, y_sample_bytes, 0x00020000);
    u32x4 yq[RT][NQ];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rv = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
#pragma unroll
      for (int jp = 0; jp < NQ; ++jp) yq[rt][jp] = __builtin_amdgcn_raw_buffer_load_b128(yrs, rv ? yoffc[rt] + jp * 64 : (int)0x80000000, cb * P.bst.y_ld * 2, 0);
    }
    const int cbq = ychunk * 16;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool rv = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
      u32x2 o2[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[rt][j][e];
        if (P.add != nullptr && rv) {
          const char* ap = P.add + (int64_t)n * add_sample_bytes + (int64_t)cb * P.add_ld * ASZ + aoff[rt] + j * 16 * ASZ;
          if (af32) { const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap); v[0] += a4[0]; v[1] += a4[1]; v[2] += a4[2]; v[3] += a4[3]; }
          else {
            const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap);
            v[0] += h2f<H>(w2[0] & 0xffffu); v[1] += h2f<H>(w2[0] >> 16); v[2] += h2f<H>(w2[1] & 0xffffu); v[3] += h2f<H>(w2[1] >> 16);
          }
        }
        o2[j] = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
      }
#pragma unroll
      for (int jp = 0; jp < NQ; ++jp) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(o2[2 * jp][0], o2[2 * jp + 1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(o2[2 * jp][1], o2[2 * jp + 1][1], false, false);
        const u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(o4, ors, rv ? ooff[rt] - 4 * q4 * 2 + jp * 64 + cbq : (int)0x80000000, cb * P.o_ld * 2, 0);
        asm volatile("s_nop 1" ::"v"(o4));            // (the store-data hazard of epilogue() above)
        bst_chunk(jp, rv ? o4 : u32x4{0u, 0u, 0u, 0u}, yq[rt][jp]);
      }
    }
  };

  // tile sequence: each XCD owns a contiguous range of tiles (neighbouring halos share that XCD's L2)
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

  __syncthreads();                                   // tap tables visible
  int t = first;
  if (t < last) {
    gload(t);
    wload(0, 0);
    sstore();
  }
  __syncthreads();                                   // halo + stage 0 of the weights landed
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
    }
    zero_acc();
#pragma unroll 1
    for (int s = 0; s < SW_NSTAGE; ++s) {
      const int slot = wstage & 1;
      if (s + 1 < SW_NSTAGE) wload(s + 1, slot ^ 1);
      else if (tn < last) wload(0, slot ^ 1);        // first stage of the next tile
      if (s == 0 && tn < last) gload(tn);            // next halo rides in registers until this tile is done
      if ((apf || BST) && s == 0) add_issue(sTab[96] & 255, n, a0, b0, c0);
      const char* wb = sW + slot * (SW_G * SW_TAPB) + wrd;
#pragma unroll
      for (int g = 0; g < SW_G; ++g) {
        const int tap = s * SW_G + g;
        if (SW_G * SW_NSTAGE > 27 && tap >= 27) continue;
        const int delta = sTab[tap];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          u32x4 xf[RT], wf[NT];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            xf[rt] = (SW_ABL & 2) ? u32x4{(uint32_t)(delta + rt), 1u, 2u, (uint32_t)ks} : *reinterpret_cast<const u32x4*>(sH + abase[rt] + ks * 4 * PLANE + delta);
          const char* wk = wb + g * SW_TAPB + (ks >> 1) * (CN * 128) + (((4 * (ks & 1) + q4) ^ wswz) << 4);
#pragma unroll
          for (int j = 0; j < NT; ++j) wf[j] = (SW_ABL & 2) ? u32x4{(uint32_t)(j + g), 3u, (uint32_t)tap, 5u} : *reinterpret_cast<const u32x4*>(wk + j * 16 * 128);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if constexpr ((SW_ABL & 1) != 0) acc[rt][j][0] += __builtin_bit_cast(f32x4, wf[j])[0] * __builtin_bit_cast(f32x4, xf[rt])[1];
              else mma16<H>(acc[rt][j], wf[j], xf[rt]);
            }
        }
        if constexpr (UP) {
          const int cl = sTab[96 + tap];
          if (cl & 256) {                            // wave-uniform: last tap of a parity class
            epilogue(cl & 255, tap + 1 < 27 ? (sTab[96 + tap + 1] & 255) : -1, n, a0, b0, c0);
            zero_acc();
          }
        }
      }
      ++wstage;
      if constexpr (UP) stage_barrier();             // next stage's DMA landed, this slot is free
      else __syncthreads();
    }
    if constexpr (!UP && BST) epilogue_plain_bst(n, a0, b0, c0);
    else if constexpr (!UP) epilogue_plain(0, n, a0, b0, c0);
    if (tn < last) {
      // every wave passed the last stage barrier => nobody reads the halo any more
      sstore();
      __syncthreads();
    }
  }
  if (STATS && stat_n >= 0) flush_stats(stat_n);
  if constexpr (BST) { if (bst_n >= 0) flush_bst(bst_n); }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static bool sw_geom(const ConvKArgs& a, int vb, SwGeom& g) {
  const int ta = (vb == 128 || a.sout == 2) ? 4 : 2;      // SwCfg<VB, UP>::TA
  const int dims[3] = {a.Xr, a.Yr, a.Zr};
  const int istr[3] = {a.Yi * a.Zi, a.Zi, 1};
  const int ostr[3] = {a.Yo * a.Zo * a.sout, a.Zo * a.sout, a.sout};
  auto waste = [&](int pa, int pb, int pc) {
    const double full = (double)dims[0] * dims[1] * dims[2];
    const double padded = (double)((dims[pa] + ta - 1) / ta * ta) * ((dims[pb] + 7) / 8 * 8) * ((dims[pc] + 7) / 8 * 8);
    return padded / full;
  };
  // (a,b,c) = (x,y,z) keeps z innermost; (z,x,y) puts the short tile axis along a shallow z
  int pa = 0, pb = 1, pc = 2;
  if (waste(2, 0, 1) < waste(0, 1, 2) - 1e-9) { pa = 2; pb = 0; pc = 1; }
  g.pa = pa; g.pb = pb; g.pc = pc;
  g.da = dims[pa]; g.db = dims[pb]; g.dc = dims[pc];
  g.ia = istr[pa]; g.ib = istr[pb]; g.ic = istr[pc];
  g.oa = ostr[pa]; g.ob = ostr[pb]; g.oc = ostr[pc];
  const int tan = (g.da + ta - 1) / ta;
  g.tbn = (g.db + 7) / 8; g.tcn = (g.dc + 7) / 8;
  g.tiles = tan * g.tbn * g.tcn;
  return true;
}

bool conv_halo_sw_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (!is16(dtype) || a.out_f32 || a.sin != 1) return false;
  const int vb = a.Cg * 2;
  if (!((vb == 128 && a.Cn == 64) || (vb == 256 && a.Cn == 32)) || a.Cn_store != a.Cn) return false;
  if ((a.g_ld % 8) != 0 || ((uintptr_t)a.in % 16) != 0 || ((uintptr_t)a.w % 16) != 0) return false;
  if (a.Xr != a.Xi || a.Yr != a.Yi || a.Zr != a.Zi) return false;
  if ((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2 >= (1ll << 31)) return false;       // 32-bit per-sample byte offsets
  if ((int64_t)a.Xr * a.Yr * a.Zr < 2048) return false;                            // too few tiles to fill the chip: generic kernel
  const bool up = nclass == 8;
  if (up) {
    if (a.sout != 2 || a.Xo != 2 * a.Xr || a.Yo != 2 * a.Yr || a.Zo != 2 * a.Zr) return false;
  } else {
    if (nclass != 1 || a.sout != 1 || a.Xo != a.Xr || a.Yo != a.Yr || a.Zo != a.Zr) return false;
  }
  int taps = 0;
  for (int c = 0; c < nclass; ++c) {
    const ctseg_conv_class& k = a.cls[c];
    if (up ? (k.ntaps != (1 << __builtin_popcount(c))) : (k.ntaps != 27)) return false;
    if (k.kpad < k.ntaps * a.Cg || (k.kpad % 8) != 0 || (k.w_off % 8) != 0) return false;
    if (k.w_off + (int64_t)a.Cn * k.kpad >= (1ll << 31)) return false;
    taps += k.ntaps;
    for (int j = 0; j < k.ntaps; ++j)
      for (int s = 0; s < 24; s += 8) {
        const int d = (int)(int8_t)((k.taps[j] >> s) & 0xff);
        if (d < (up ? 0 : -1) || d > 1) return false;
      }
  }
  return taps == 27;
}

static int sw_grid(const ConvKArgs& a, const SwGeom& g) {
  return persistent_grid(CTSEG_NUM_CU, g.tiles * a.N);
}

int conv_halo_sw_slots(const ConvKArgs& a) {
  SwGeom g;
  sw_geom(a, a.Cg * 2, g);
  return sw_grid(a, g);
}

// ConvKArgs::bst on this pass: the 8-class 128 -> 32 and the single-class 64 -> 64 input gradients in bf16, all written channels
int conv_halo_sw_bst_slots(const ConvKArgs& a, int nclass) {
  { const char* e = getenv("CTSEG_BST_SW"); if (e != nullptr && (e[0] == '0' || (e[0] == '8' && nclass != 8) || (e[0] == '1' && nclass != 1))) return 0; }   // (A/B switch)
  if (a.dtype != CTSEG_BF16 || a.stats != nullptr || a.bias != nullptr) return 0;
  if (!((nclass == 8 && a.Cg == 128) || (nclass == 1 && a.Cg == 64))) return 0;
  if (a.bst.C != a.Cn || a.bst.col0 != 0 || a.Cn_store != a.Cn || (a.o_ld % 8) != 0 || ((uintptr_t)a.out % 16) != 0) return 0;
  if ((a.bst.y_ld % 8) != 0 || ((uintptr_t)a.bst.y % 16) != 0) return 0;
  if ((int64_t)a.Xo * a.Yo * a.Zo * a.bst.y_ld * 2 >= (1ll << 31) || (int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * 2 >= (1ll << 31)) return 0;
  return conv_halo_sw_slots(a);
}

void launch_conv_halo_sw(ConvKArgs& a, int nclass, hipStream_t st) {
  SwGeom g;
  const int vb = a.Cg * 2;
  sw_geom(a, vb, g);
  a.tiles = g.tiles;
  const int total = g.tiles * a.N;
  const dim3 grid((unsigned)sw_grid(a, g)), blk(SW_NTHR);
  const bool up = nclass == 8, stats = a.stats != nullptr;
  if (a.bst.part != nullptr) {        // (conv_halo_sw_bst_slots: bf16, 8 classes, 128 -> 32, no forward statistics)
    if (up) hipLaunchKernelGGL((conv_halo_sw_kernel<BF16, 256, true, false, true>), grid, blk, 0, st, a, g, total);
    else hipLaunchKernelGGL((conv_halo_sw_kernel<BF16, 128, false, false, true>), grid, blk, 0, st, a, g, total);
    return;
  }
#define CTSEG_SW_LAUNCH(VB, UP, ST)                                                                                      \
  do {                                                                                                                  \
    if (a.dtype == CTSEG_F16) hipLaunchKernelGGL((conv_halo_sw_kernel<F16, VB, UP, ST>), grid, blk, 0, st, a, g, total); \
    else hipLaunchKernelGGL((conv_halo_sw_kernel<BF16, VB, UP, ST>), grid, blk, 0, st, a, g, total);                     \
  } while (0)
  if (vb == 128) {
    if (up) { if (stats) CTSEG_SW_LAUNCH(128, true, true); else CTSEG_SW_LAUNCH(128, true, false); }
    else { if (stats) CTSEG_SW_LAUNCH(128, false, true); else CTSEG_SW_LAUNCH(128, false, false); }
  } else {
    if (up) { if (stats) CTSEG_SW_LAUNCH(256, true, true); else CTSEG_SW_LAUNCH(256, true, false); }
    else { if (stats) CTSEG_SW_LAUNCH(256, false, true); else CTSEG_SW_LAUNCH(256, false, false); }
  }
#undef CTSEG_SW_LAUNCH
}

}  // namespace ctseg
