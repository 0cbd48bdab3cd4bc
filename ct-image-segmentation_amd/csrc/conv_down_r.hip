// Stride-2 3x3x3 pass 32 -> 128 channels with the weights in REGISTERS (gfx950, 16-bit storage):
//   * Conv3d 32 -> 64+64 k3 s2, the fused [residual | unit0] convolution of the second down block (forward, + InstanceNorm partials)
//   * the input gradient of the level-1 ConvTranspose3d(128 -> 32) = a stride-2 conv 32 -> 128 over its output gradient
// of the reference's U-Net (MONAI UNet built at capstone/volumetric/base_trainer.py:65-72).
//
//   out[r][n] = bias[n] + sum_{tap,c} in[2 r + d(tap)][c] * W[n][tap*32 + c],   d in {-1,0,1}^3
//
// The generic kernel gathers 27 x 64 bytes per output row through L2 (732 MB of reads for a 200 MB input, 0.21 ms); a halo kernel
// that STREAMS the 221 KB of weights through LDS per 128-row tile was slower still (conv_down_halo.hip).  Here each of the eight
// waves owns 16 output columns and keeps their 27 x 32 weights in 108 registers for the whole launch; the workgroup stages the
// 3 x 17 x 17 input halo of a 1 x 8 x 8 output tile with LDS-DMA into one of two buffers (60 KB each) while it multiplies from the
// other, and every wave reads the whole halo: 108 ds_read_b128 for 108 MFMAs per wave and tile.  One barrier per tile; the stage
// wait is counted (the four stores of the previous tile's epilogue stay in flight).  Measured (2 x 128 x 128 x 12 rows): forward
// 0.213 -> 0.146 ms, input gradient 0.155 -> 0.13 ms; zero LDS bank conflicts; the LDS port (864 KB of operand reads per tile at
// ~150 B/clk) and the halo stream (1.69 x the input for a 1-deep tile) are what is left.
//
// LDS halo image: VOXEL-major 64-byte rows, so that a DMA piece (1 KB, lane-linear) is 16 whole voxel rows of the input — a
// plane-major image made every lane of a piece fetch 16 bytes of a different row: 64 cache-line requests per instruction, 0.14 of
// 0.23 ms.  Slot of halo voxel (ha, hb, hc) = (ha * 17 + hb) * 17 + (hc ^ ((hc >> 2) & 1)); its 16-byte chunk q sits at position
// q ^ (2 * ((hb >> 1) & 1)).  An MFMA operand row group reads 2 (b) x 8 (c) output voxels = input slots two apart: the swap of
// neighbouring slots in every other group of four spreads the 8 c-voxels of a row over all four 64-byte positions of the 256-byte
// bank row (twice), the two chunks a ds_read_b128 lane group mixes separate those pairs, and the chunk XOR separates the two b rows:
// 16 distinct 16-byte slots per lane group for every tap.  Both swizzles are applied on the DMA's source side.
#include "conv_common.h"

#ifndef DR_ABL
#define DR_ABL 0     // timing-only ablation: 1 no MFMAs, 2 no LDS operand reads, 4 no DMA, 8 no output stores
#endif

namespace ctseg {

constexpr int DR_NTHR = 512, DR_HB = 17, DR_HC = 17, DR_HA = 3;
constexpr int DR_SLOTS = DR_HA * DR_HB * DR_HC;                 // 867 voxel slots of 64 bytes
constexpr int DR_NP = (DR_SLOTS + 15) / 16, DR_PJ = (DR_NP + 7) / 8;   // 55 DMA pieces (16 slots) per tile, <= 7 per wave
constexpr int DR_HALO = DR_NP * 1024;                           // 56 320 bytes

typedef int32_t dr_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* dr_lds_ptr;
__device__ void dr_buffer_load_lds(dr_i32x4 rsrc, dr_lds_ptr lds, int size, int voffset, int soffset, int offset,
                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
__device__ __forceinline__ dr_i32x4 dr_make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  dr_i32x4 v = __builtin_bit_cast(dr_i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]); v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]); v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

struct DownRGeom {
  int da, db, dc;            // output (row grid) extents along the tile axes (a = the 1-deep axis)
  int ia, ib, ic;            // input voxel strides of the tile axes
  int oa, ob, oc;            // output voxel strides of the tile axes
  int pa, pb, pc;            // volume axis of each tile axis
  int tbn, tcn, tiles;
  int in_sample_bytes, out_sample_bytes, out2_sample_bytes;
  int a_fast;                // tile order: the 1-deep axis fastest
};

// BST (round 4): backward InstanceNorm statistics of the written gradient (ConvKArgs::bst) for the waves whose 16 columns are channels
// of the norm — here the sub-block half of the level-1 concat gradient.  A wave owns its columns, so the three sums need neither LDS
// nor a barrier: a lane keeps them for its 4 channels over the workgroup's tiles and the 16 voxel lanes are combined by a butterfly at
// every sample change, one partial row per workgroup and sample (the layout ctseg_instnorm_prelu_bwd_finalize sums).  The y values of a
// tile are requested BEFORE the next tile's DMA pieces (the counted wait at the loop head covers exactly the stores behind the DMA) and
// consumed behind its 108 multiplies: a tile lasts ~4 us here, longer than the round trip.
template <typename H, bool STATS, bool BST = false>
__global__ __launch_bounds__(DR_NTHR) void conv_down_r_kernel(const ConvKArgs P, const DownRGeom G, int total_tiles) {
  static_assert(!(STATS && BST), "forward statistics or backward statistics");
  __shared__ __attribute__((aligned(16))) char smem[2 * DR_HALO + 64 * 4];
  int* const sTab = reinterpret_cast<int*>(smem + 2 * DR_HALO);   // per tap: [0,32) halo byte offset, [32,64) XOR flags
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const ctseg_conv_class& K = P.cls[0];
  const int col0 = wave * 16;

  if (tid < 32) {
    int da1 = 1, db1 = 1, dc1 = 1;
    if (tid < 27) {
      const int tp = K.taps[tid];
      const int dv[3] = {(int)(int8_t)(tp & 0xff), (int)(int8_t)((tp >> 8) & 0xff), (int)(int8_t)((tp >> 16) & 0xff)};
      da1 = dv[G.pa] + 1; db1 = dv[G.pb] + 1; dc1 = dv[G.pc] + 1;
    }
    sTab[tid] = (da1 * DR_HB + db1) * DR_HC * 64;              // row part of the tap (the c part goes through the slot swap)
    sTab[32 + tid] = dc1 | ((db1 >> 1) << 8);                  // c offset + 1; parity contribution of db1 to (hb >> 1)
  }

  // ---- this wave's weights: 27 taps x (16 columns x 32 channels), one MFMA A-operand fragment per tap ----------------------------
  u32x4 W[27];
  {
    const char* wrow = P.w + (K.w_off + (int64_t)(col0 + r16) * K.kpad) * 2 + q4 * 16;
#pragma unroll
    for (int j = 0; j < 27; ++j) W[j] = *reinterpret_cast<const u32x4*>(wrow + j * 64);
  }
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = P.bias != nullptr ? P.bias[col0 + 4 * q4 + e] : 0.f;

  // ---- DMA pieces of this wave: piece i = wave + 8 j fills physical slots [16 i, 16 i + 16), lane = (slot, chunk position) --------
  // per lane: source byte offset from the halo origin, low 4 bits = "first halo plane along a / b / c" and "no such slot"
  int dpk[DR_PJ];
#pragma unroll
  for (int j = 0; j < DR_PJ; ++j) {
    const int i = wave + 8 * j, ps = i * 16 + (lane >> 2), cp = lane & 3;
    const int ha = ps / (DR_HB * DR_HC), rem = ps - ha * (DR_HB * DR_HC), hb = rem / DR_HC, pp = rem - hb * DR_HC;
    const int hc = pp ^ ((pp >> 2) & 1), q = cp ^ (2 * ((hb >> 1) & 1));
    const bool ok = i < DR_NP && ps < DR_SLOTS;
    dpk[j] = ok ? (((ha * G.ia + hb * G.ib + hc * G.ic) * P.g_ld * 2 + q * 16) | (ha == 0 ? 1 : 0) | (hb == 0 ? 2 : 0) | (hc == 0 ? 4 : 0)) : 8;
  }
  const int hbias = (G.ia + G.ib + G.ic) * P.g_ld * 2;

  auto tile_origin = [&](int t, int& n, int& a0, int& b0, int& c0) {
    n = t / G.tiles;
    int r = t - n * G.tiles;
    // the 1-deep axis runs fastest: the workgroups of one XCD (consecutive tile ids) work on neighbours along a, which share one of
    // their three input planes through that XCD's L2
    int ta, tb, tc;
    if (G.a_fast) { ta = r % G.da; r /= G.da; tc = r % G.tcn; tb = r / G.tcn; }
    else { tc = r % G.tcn; r /= G.tcn; tb = r % G.tbn; ta = r / G.tbn; }
    a0 = ta; b0 = tb * 8; c0 = tc * 8;
  };
  auto dma = [&](int t, int buf) {
    int n, a0, b0, c0;
    tile_origin(t, n, a0, b0, c0);
    const dr_i32x4 rs = dr_make_rsrc(P.in + (int64_t)n * G.in_sample_bytes - hbias, (uint32_t)(G.in_sample_bytes + hbias));
    const int soff = (2 * a0 * G.ia + 2 * b0 * G.ib + 2 * c0 * G.ic) * P.g_ld * 2;
    const int m = (a0 == 0 ? 1 : 0) | (b0 == 0 ? 2 : 0) | (c0 == 0 ? 4 : 0) | 8;
    char* dst = smem + buf * DR_HALO;
#pragma unroll
    for (int j = 0; j < DR_PJ; ++j) {
      const int i = wave + 8 * j;
      if (i < DR_NP && !(DR_ABL & 4)) {       // wave-uniform
        const int vo = (dpk[j] & m) == 0 ? (dpk[j] & ~15) : (int)0x80000000;
        dr_buffer_load_lds(rs, (dr_lds_ptr)(dst + i * 1024), 16, vo, soff, 0, 0);
      }
    }
  };

  // ---- MFMA operand addressing: output voxel (b, c) of row tile rt -> halo slot (2 b, 2 c) + tap part (conv_down_halo.hip) ----------
  const int pb = r16 >> 3, pc = r16 & 7;
  int lbase[4], ooff[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int vb = 2 * rt + pb;
    lbase[rt] = (2 * vb) * DR_HC * 64;                   // row of output voxel b (tap b offset 0), before the tap's row part
    ooff[rt] = vb * G.ob + pc * G.oc;                    // output voxel offset from the tile's first voxel
  }
  // c part per c offset of a tap: swapped slot of hc = 2 pc + dc1; chunk position per parity of (hb >> 1) = pb ^ (db1 >> 1)
  int pz[3], cq[2];
#pragma unroll
  for (int d = 0; d < 3; ++d) { const int hc = 2 * pc + d; pz[d] = (hc ^ ((hc >> 2) & 1)) * 64; }
#pragma unroll
  for (int k = 0; k < 2; ++k) cq[k] = (q4 ^ (2 * (pb ^ k))) * 16;
  // output: columns of this wave go to `out` or, past out2_col0, to `out2`
  const bool second = P.out2 != nullptr && col0 >= P.out2_col0;
  char* const obase = second ? P.out2 : P.out;
  const int old_ = second ? P.o2_ld : P.o_ld, ocol = second ? col0 - P.out2_col0 : col0;
  const int osb = second ? G.out2_sample_bytes : G.out_sample_bytes;

  float wsum[4] = {0.f, 0.f, 0.f, 0.f}, wsq[4] = {0.f, 0.f, 0.f, 0.f};
  int stat_n = -1;
  auto flush_stats = [&](int n) {       // the wave owns its 16 columns: butterfly over the 16 voxel lanes, no LDS, no barrier
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = wsum[e], b = wsq[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
      if (r16 == 0) {
        const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
        P.stats[(slot_t * 2 + 0) * P.stats_ld + col0 + 4 * q4 + e] = a;
        P.stats[(slot_t * 2 + 1) * P.stats_ld + col0 + 4 * q4 + e] = b;
      }
      wsum[e] = 0.f;
      wsq[e] = 0.f;
    }
  };

  // ---- BST state ----------------------------------------------------------------------------------------------------------------
  const bool ywave = BST && col0 >= P.bst.col0 && col0 + 16 <= P.bst.col0 + P.bst.C;      // wave-uniform
  const int ych = BST ? col0 - P.bst.col0 + 4 * q4 : 0;                                    // the lane's first channel of the norm
  const float bal = BST ? P.bst.alpha[0] : 1.f;
  float brs[4] = {0.f, 0.f, 0.f, 0.f}, bnm[4] = {0.f, 0.f, 0.f, 0.f};                      // rstd, -mean * rstd of the lane's channels
  float b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f}, b3 = 0.f;
  int bst_n = -1;
  const int y_sample_bytes = BST ? (int)((int64_t)P.Xo * P.Yo * P.Zo * P.bst.y_ld * 2) : 0;   // < 2^31 (host-checked)
  auto bst_consts = [&](int n) {
    if (ywave) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float mean = P.bst.mr[((int64_t)n * P.bst.C + ych + e) * 2], rstd = P.bst.mr[((int64_t)n * P.bst.C + ych + e) * 2 + 1];
        brs[e] = rstd;
        bnm[e] = -mean * rstd;
      }
    }
  };
  auto flush_bst = [&](int n) {
    if (ywave) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = b1[e], b = b2[e], c = e == 0 ? b3 : 0.f;        // (one PReLU slope: only the total of the third sum matters)
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
        if (r16 == 0) {
          float* row = P.bst.part + ((int64_t)n * P.bst.P + blockIdx.x) * 3 * P.bst.ld + ych + e;
          row[0] = a; row[P.bst.ld] = b; row[2 * P.bst.ld] = c;
        }
        b1[e] = 0.f; b2[e] = 0.f;
      }
      b3 = 0.f;
    }
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

  int t = first, buf = 0;
  if (t < last) dma(t, 0);
  bool stores_in_flight = false;
  for (; t < last; t += stride, buf ^= 1) {
    // the DMA of this tile (issued one tile ago, BEFORE the previous epilogue's four stores) has landed; the stores may still fly
    if (stores_in_flight) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // ... for every wave; and every wave is done reading the other buffer
    asm volatile("" ::: "memory");
    int n, a0, b0, c0;
    tile_origin(t, n, a0, b0, c0);
    u32x2 yv[4] = {u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}, u32x2{0u, 0u}};
    if constexpr (BST) {
      if (n != bst_n) {
        if (bst_n >= 0) flush_bst(bst_n);
        bst_n = n;
        bst_consts(n);
      }
      // y of this tile's voxels, ahead of the next tile's DMA pieces (always issued: the counted wait at the loop head)
      const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.bst.y) + (int64_t)n * y_sample_bytes, 0, y_sample_bytes, 0x00020000);
      const int ysoff = (a0 * G.oa + b0 * G.ob + c0 * G.oc) * P.bst.y_ld * 2;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const bool rv = ywave && (b0 + 2 * rt + pb < G.db) && (c0 + pc < G.dc);
        yv[rt] = __builtin_amdgcn_raw_buffer_load_b64(yrs, rv ? (ooff[rt] * P.bst.y_ld + ych) * 2 : (int)0x80000000, ysoff, 0);
      }
    }
    if (t + stride < last) dma(t + stride, buf ^ 1);
    if (STATS && n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = n;
    }
    f32x4 acc[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* hb = smem + buf * DR_HALO;
    // (requesting the fragments of tap j + 1 before the multiplies of tap j — two register sets, 238 registers — measured no faster:
    // the LDS port is the limit here, 148 B/clk of ds_read_b128 for 864 KB per tile, not the read latency)
#pragma unroll
    for (int j = 0; j < 27; ++j) {
      const int tbase = __builtin_amdgcn_readfirstlane(sTab[j]), tflag = __builtin_amdgcn_readfirstlane(sTab[32 + j]);
      const int dc1 = tflag & 0xff;
      const int toff = tbase + (dc1 == 0 ? pz[0] : dc1 == 1 ? pz[1] : pz[2]) + ((tflag >> 8) ? cq[1] : cq[0]);
      u32x4 xf[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        xf[rt] = (DR_ABL & 2) ? u32x4{(uint32_t)(tbase + rt), 1u, 2u, (uint32_t)toff} : *reinterpret_cast<const u32x4*>(hb + lbase[rt] + toff);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        if constexpr ((DR_ABL & 1) != 0) acc[rt][0] += __builtin_bit_cast(f32x4, W[j])[0] * __builtin_bit_cast(f32x4, xf[rt])[1];
        else mma16<H>(acc[rt], W[j], xf[rt]);
      }
    }
    // ---- epilogue: lane = (voxel (b, c) of row tile rt, columns col0 + 4 q4 .. + 3); four buffer stores, always issued ---------
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(obase + (int64_t)n * osb, 0, osb, 0x00020000);
    const int soff = (a0 * G.oa + b0 * G.ob + c0 * G.oc) * old_ * 2;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const bool rv = (b0 + 2 * rt + pb < G.db) && (c0 + pc < G.dc);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[rt][e] + bias[e];
        if (STATS && rv) { wsum[e] += v[e]; wsq[e] += v[e] * v[e]; }
      }
      const u32x2 o2 = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
      __builtin_amdgcn_raw_buffer_store_b64(o2, ors, (rv && (!(DR_ABL & 8) || v[0] + v[1] == 1.2345f)) ? (ooff[rt] * old_ + ocol + 4 * q4) * 2 : (int)0x80000000, soff, 0);
      if constexpr (BST) {
        if (ywave && rv) {       // the sums are over the STORED gradient (rounded to the storage type), as the reduce pass reads it
          const float g4[4] = {h2f<H>(o2[0] & 0xffffu), h2f<H>(o2[0] >> 16), h2f<H>(o2[1] & 0xffffu), h2f<H>(o2[1] >> 16)};
          const float y4[4] = {h2f<H>(yv[rt][0] & 0xffffu), h2f<H>(yv[rt][0] >> 16), h2f<H>(yv[rt][1] & 0xffffu), h2f<H>(yv[rt][1] >> 16)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = fmaf(y4[e], brs[e], bnm[e]);
            const float dxh = g4[e] * (xh > 0.f ? 1.f : bal);
            b1[e] += dxh;
            b2[e] = fmaf(dxh, xh, b2[e]);
            b3 = fmaf(g4[e], fminf(xh, 0.f), b3);
          }
        }
      }
    }
    stores_in_flight = true;
  }
  if (STATS && stat_n >= 0) flush_stats(stat_n);
  if constexpr (BST) { if (bst_n >= 0) flush_bst(bst_n); }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static void down_r_geom(const ConvKArgs& a, DownRGeom& g) {
  const int od[3] = {a.Xr, a.Yr, a.Zr};
  const int istr[3] = {a.Yi * a.Zi, a.Zi, 1}, ostr[3] = {a.Yo * a.Zo, a.Zo, 1};
  auto waste = [&](int pb, int pc) { return (double)((od[pb] + 7) / 8 * 8) * ((od[pc] + 7) / 8 * 8) / ((double)od[pb] * od[pc]); };
  // the 1-deep tile axis takes the volume axis whose removal leaves the least padding in the 8 x 8 face; c stays the innermost of the rest
  int pa = 0, pb = 1, pc = 2;
  double best = waste(1, 2);
  if (waste(0, 1) < best - 1e-9) { pa = 2; pb = 0; pc = 1; best = waste(0, 1); }
  if (waste(0, 2) < best - 1e-9) { pa = 1; pb = 0; pc = 2; }
  g.pa = pa; g.pb = pb; g.pc = pc;
  g.da = od[pa]; g.db = od[pb]; g.dc = od[pc];
  g.ia = istr[pa]; g.ib = istr[pb]; g.ic = istr[pc];
  g.oa = ostr[pa]; g.ob = ostr[pb]; g.oc = ostr[pc];
  g.tbn = (g.db + 7) / 8; g.tcn = (g.dc + 7) / 8;
  g.tiles = g.da * g.tbn * g.tcn;
  g.in_sample_bytes = (int)((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2);
  g.out_sample_bytes = (int)((int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * 2);
  g.out2_sample_bytes = a.out2 != nullptr ? (int)((int64_t)a.Xo * a.Yo * a.Zo * a.o2_ld * 2) : 0;
  g.a_fast = getenv("CTSEG_DR_AFAST") != nullptr ? atoi(getenv("CTSEG_DR_AFAST")) : 1;
}

bool conv_down_r_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (getenv("CTSEG_NO_DOWN_R") != nullptr) return false;
  if (!is16(dtype) || a.out_f32 || nclass != 1 || a.sin != 2 || a.sout != 1 || a.add != nullptr) return false;
  if (a.Cg != 32 || a.Cn != 128 || a.Cn_store != 128) return false;
  if ((a.g_ld % 8) != 0 || ((uintptr_t)a.in % 16) != 0 || ((uintptr_t)a.w % 16) != 0) return false;
  if (a.Xo != a.Xr || a.Yo != a.Yr || a.Zo != a.Zr) return false;
  if (a.Xi != 2 * a.Xr || a.Yi != 2 * a.Yr || a.Zi != 2 * a.Zr) return false;      // (only the -1 taps of a first tile leave the volume)
  if (((int64_t)a.Xi * a.Yi * a.Zi + (int64_t)a.Yi * a.Zi + a.Zi + 1) * a.g_ld * 2 >= (1ll << 31)) return false;
  if ((int64_t)a.Xo * a.Yo * a.Zo * 128 * 2 >= (1ll << 31)) return false;
  if ((int64_t)a.Xr * a.Yr * a.Zr < 2048) return false;
  if (a.out2 != nullptr ? ((a.out2_col0 % 16) != 0 || (a.o2_ld % 4) != 0 || a.o2_ld < 128 - a.out2_col0 || a.o_ld < a.out2_col0) : a.o_ld < 128) return false;
  if ((a.o_ld % 4) != 0) return false;
  const ctseg_conv_class& k = a.cls[0];
  if (k.ntaps != 27 || k.kpad < 27 * 32 || (k.kpad % 8) != 0 || (k.w_off % 8) != 0) return false;
  for (int j = 0; j < 27; ++j)
    for (int s = 0; s < 24; s += 8) {
      const int d = (int)(int8_t)((k.taps[j] >> s) & 0xff);
      if (d < -1 || d > 1) return false;
    }
  return true;
}

static int down_r_grid(const ConvKArgs& a, const DownRGeom& g) {
  return persistent_grid(CTSEG_NUM_CU, g.tiles * a.N);
}

int conv_down_r_slots(const ConvKArgs& a) {
  DownRGeom g;
  down_r_geom(a, g);
  return down_r_grid(a, g);
}

// ConvKArgs::bst on this pass (bf16, no forward statistics, no bias): whole 16-column blocks of the written columns, y in 8-byte pieces
int conv_down_r_bst_slots(const ConvKArgs& a) {
  { const char* e = getenv("CTSEG_BST_DOWN_R"); if (e != nullptr && e[0] == '0') return 0; }   // (A/B switch)
  if (a.dtype != CTSEG_BF16 || a.stats != nullptr || a.bias != nullptr) return 0;
  if (a.bst.C <= 0 || (a.bst.C % 16) != 0 || (a.bst.col0 % 16) != 0 || a.bst.col0 + a.bst.C > 128) return 0;
  if ((a.bst.y_ld % 4) != 0 || a.bst.y_ld < a.bst.C || ((uintptr_t)a.bst.y % 8) != 0) return 0;
  if ((int64_t)a.Xo * a.Yo * a.Zo * a.bst.y_ld * 2 >= (1ll << 31)) return 0;
  return conv_down_r_slots(a);
}

void launch_conv_down_r(ConvKArgs& a, hipStream_t st) {
  DownRGeom g;
  down_r_geom(a, g);
  a.tiles = g.tiles;
  const int total = g.tiles * a.N;
  const dim3 grid((unsigned)down_r_grid(a, g)), blk(DR_NTHR);
  const bool stats = a.stats != nullptr;
  if (a.bst.part != nullptr) {        // (bf16, no forward statistics: conv_down_r_bst_slots)
    hipLaunchKernelGGL((conv_down_r_kernel<BF16, false, true>), grid, blk, 0, st, a, g, total);
    return;
  }
  if (a.dtype == CTSEG_F16) {
    if (stats) hipLaunchKernelGGL((conv_down_r_kernel<F16, true>), grid, blk, 0, st, a, g, total);
    else hipLaunchKernelGGL((conv_down_r_kernel<F16, false>), grid, blk, 0, st, a, g, total);
  } else {
    if (stats) hipLaunchKernelGGL((conv_down_r_kernel<BF16, true>), grid, blk, 0, st, a, g, total);
    else hipLaunchKernelGGL((conv_down_r_kernel<BF16, false>), grid, blk, 0, st, a, g, total);
  }
}

}  // namespace ctseg
