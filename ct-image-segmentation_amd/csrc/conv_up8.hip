// 8-parity-class stride-2 passes with MANY gathered channels and 64 output columns (gfx950, 16-bit storage):
//   * ConvTranspose3d 384 -> 64 k3 s2 forward at 64x64x6 -> 128x128x12
//   * the input gradient of the fused stride-2 [residual | unit0] convolution 64 -> 128+128 (Cg = 256 -> Cn = 64, + addend)
// of the reference's U-Net (MONAI UNet built at capstone/volumetric/base_trainer.py:65-72, level 2).
//
//   out[2 r + p][n] = bias[n] + add + sum_{delta <= p} sum_c in[r + delta][c] * W_p[n][tap(p, delta) * Cg + c],   p, delta in {0,1}^3
//
// The generic kernel runs the 8 classes as 8 independent GEMMs over 128-row tiles: per 128 coarse voxels it moves 27 tap gathers of
// 98 KB plus 1.33 MB of class weights from L2 to LDS (4 MB; 13 TB/s over the launch: L2-bandwidth bound at 22 % MFMA), and the 1- and
// 2-tap classes are K loops of 6-12 stages whose prologue is exposed.  Here:
//
//  * a workgroup owns a 3 x 8 x 8 coarse tile (192 voxels: the 6-deep level is two tiles deep, 256 tiles at the reference's size)
//    and ONE of two class groups — {0, 3, 7} (1 + 4 + 8 = 13 taps) or {1, 2, 4, 5, 6} (2 + 2 + 2 + 4 + 4 = 14 taps) — and keeps the
//    accumulators of all its classes in registers (a wave: 3 row tiles x 2 column tiles x <= 5 classes = 120 registers), walking K in
//    chunks of 32 channels; nothing is stored until the last chunk, so the input halo (4 x 9 x 9 voxels: the taps only reach offsets
//    0 / +1) is read once per tile, chunk and group instead of once per tap.  Per 192 coarse voxels and chunk a workgroup streams
//    <= 56 KB of weights + 23 KB of halo.  (A first version with all 8 classes in one workgroup of 128 voxels streamed 110 + 18 KB
//    per chunk, was bound by exactly that stream — 0.108 ms with or without its MFMAs — and left a quarter of the chip idle in its
//    second round of tiles.)
//  * per chunk the group's (class, tap) weight blocks [64 columns][32 k] stream through a five-slot LDS ring, five blocks per stage,
//    with direct-to-LDS loads FOUR stages ahead (counted vmcnt waits; every wave issues the same number of loads per stage);
//  * the (class, tap) pairs are walked DELTA-major: the shifted operand fragments x[r + delta] are read once per chunk and feed every
//    class of the group that has the tap;
//  * the halo chunk of the next K chunk is staged by LDS-DMA (raw buffer loads ... lds, out-of-range offsets deliver zeros) into the
//    other of two buffers; conv_halo_sw.hip's plane layout and lane <-> voxel permutation keep every ds_read_b128 conflict free;
//  * epilogue as conv_halo_sw.hip's: v_permlane16_swap -> 16-byte channels-last stores, InstanceNorm partials per workgroup.
#include "conv_common.h"
#include <type_traits>

#ifndef U8_ABL
#define U8_ABL 0     // timing-only ablation: 1 no MFMAs, 2 no LDS operand reads, 4 no weight stream, 8 no halo loads, 16 no x operand reads, 32 no weight operand reads, 64 no stage barriers, 128 no DMA instructions at all
#endif

namespace ctseg {

constexpr int U8_NTHR = 512, U8_CN = 64, U8_TA = 3;
constexpr int U8_HV = 360;                                  // (TA + 1) x 9 rows of 10 slots (= 8 mod 16: conv_halo_sw.hip)
constexpr int U8_PLANE = U8_HV * 16;
constexpr int U8_HBUF = 1472 * 16;                          // 4 planes (1440 slots) + the overhang of the last DMA piece's wave
constexpr int U8_TAPB = U8_CN * 64;                         // one (class, tap) block of a 32-channel chunk: 64 rows x 64 B
constexpr int U8_ST = 5, U8_NS = 3;                         // blocks per ring stage, stages per chunk (13 / 14 blocks: 5 + 5 + 3 / 4)
constexpr int U8_STAGE = U8_ST * U8_TAPB, U8_SLOTS = 5, U8_RING = U8_SLOTS * U8_STAGE;      // look-ahead: four stages
constexpr int U8_TOTAL = 2 * U8_HBUF + U8_RING + 8 * 2 * U8_CN * 4 + 64 * 4;
static_assert(U8_TOTAL <= 160 * 1024, "LDS");

typedef int32_t u8i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* u8_lds_u32_ptr;
__device__ void u8_raw_buffer_load_lds(u8i32x4 rsrc, u8_lds_u32_ptr lds, int size, int voffset, int soffset, int offset,
                                       int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ u8i32x4 u8_make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  u8i32x4 v = __builtin_bit_cast(u8i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]);
  v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]);
  v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

// tap index of shift delta inside parity class c in the host's order (conv_up_halo.hip's up_tap_index; checked by conv_up8_eligible)
constexpr int u8_tap_index(int c, int delta) {
  const int px = (c >> 2) & 1, py = (c >> 1) & 1, pz = c & 1, dx = (delta >> 2) & 1, dy = (delta >> 1) & 1, dz = delta & 1;
  if (dx > px || dy > py || dz > pz) return -1;
  const int ix = px ? 1 - dx : 0, iy = py ? 1 - dy : 0, iz = pz ? 1 - dz : 0;
  return (ix * (1 + py) + iy) * (1 + pz) + iz;
}
// class groups and their (class, tap) pairs in delta-major order: position -> delta / class slot (index into the group)
constexpr int u8_ncls(int grp) { return grp == 0 ? 3 : 5; }
constexpr int u8_gclass(int grp, int i) { return grp == 0 ? (i == 0 ? 0 : i == 1 ? 3 : 7) : (i == 0 ? 1 : i == 1 ? 2 : i == 2 ? 4 : i == 3 ? 5 : 6); }
constexpr int u8_npos(int grp) { return grp == 0 ? 13 : 14; }
constexpr int u8_pos_find(int grp, int pos, int what) {     // what: 0 delta, 1 class slot
  int n = 0;
  for (int d = 0; d < 8; ++d)
    for (int i = 0; i < u8_ncls(grp); ++i)
      if ((u8_gclass(grp, i) & d) == d) { if (n == pos) return what == 0 ? d : i; ++n; }
  return -1;
}
static_assert(u8_pos_find(0, 12, 0) == 7 && u8_pos_find(0, 12, 1) == 2 && u8_pos_find(0, 13, 0) == -1, "group 0: 13 pairs");
static_assert(u8_pos_find(1, 13, 0) == 6 && u8_pos_find(1, 13, 1) == 4 && u8_pos_find(1, 14, 0) == -1, "group 1: 14 pairs");

struct Up8Geom {
  int da, db, dc;            // extents of the coarse (row) grid along the tile axes (a = short axis)
  int ia, ib, ic;            // gathered-tensor voxel strides of the tile axes
  int oa, ob, oc;            // written-tensor voxel strides of the tile axes (already times 2)
  int pa, pb, pc;            // which volume axis (0 = x, 1 = y, 2 = z) each tile axis runs along
  int tbn, tcn, tiles;       // tiles along b, c; tiles per sample
  int in_sample_bytes, out_sample_bytes, add_sample_bytes;
  int gxs;                   // spatial workgroups: the grid is 2 * gxs, workgroup L takes class group L / gxs
};

__device__ __forceinline__ void u8_patch_voxel(int r16, int& db, int& c) {
  db = (0xEF80u >> r16) & 1;
  c = (int)((0x2104765437653210ull >> (4 * r16)) & 7ull);
}

// ZXY: the tile axes (a, b, c) run along the volume axes (z, x, y) instead of (x, y, z) — compile time, so that the halo offset of a
// shift is an instruction immediate; GRP: the class group.  (The first version of this kernel spent 165 non-MFMA instructions per
// stage and wave around 36 MFMAs — 64-bit source addresses of the DMA pieces, bounds tests per chunk, operand address adds.)
template <typename H, bool STATS, bool ZXY, int GRP>     // H = 16-bit storage kind (BF16 / F16)
__device__ __forceinline__ void conv_up8_body(const ConvKArgs& P, const Up8Geom& G, int total_tiles, char* smem) {
  constexpr int NCLS = u8_ncls(GRP), NPOS = u8_npos(GRP);
  char* const sH = smem;
  char* const sW = smem + 2 * U8_HBUF;
  float* const sStats = reinterpret_cast<float*>(sW + U8_RING);
  int* const sTab = reinterpret_cast<int*>(sStats + 8 * 2 * U8_CN);       // [0,16) weight element offset of position, [16,32) its kpad

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, q4 = lane >> 4;
  const int NKC = P.Cg / 32;
  const int sid = blockIdx.x % G.gxs;                         // spatial workgroup index inside the group

  if (tid < 16) {
    int wo = 0, kp = 0;
    if (tid < NPOS) {
      const int c = u8_gclass(GRP, u8_pos_find(GRP, tid, 1)), ti = u8_tap_index(c, u8_pos_find(GRP, tid, 0));
      wo = (int)(P.cls[c].w_off + (int64_t)ti * P.Cg);
      kp = P.cls[c].kpad;
    }
    sTab[tid] = wo; sTab[16 + tid] = kp;
  }

  // ---- halo DMA: three rounds of 512 16-byte pieces cover the 4 planes x 360 slots of a chunk (the third: waves 0-6; wave 7 repeats
  // its second piece so that every wave issues the same number of loads) ----------------------------------------------------------
  int h_off[3], h_abc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int idx = (r == 2 && wave == 7) ? tid + 512 : tid + r * 512;
    const int pl = idx / U8_HV, s = idx - pl * U8_HV;
    const int ha = s / 90, rem = s - ha * 90, hb = rem / 10, hc = rem - hb * 10 - 1;     // halo coordinates minus one: 0 .. TA / 8 / 8
    const bool ok = idx < 4 * U8_HV && hc >= 0;
    h_off[r] = (ha * G.ia + hb * G.ib + hc * G.ic) * P.g_ld * 2 + pl * 16;
    h_abc[r] = ok ? (ha | (hb << 8) | (hc << 16)) : 0x7f7f7f;
  }
  auto tile_origin = [&](int t, int& n, int& a0, int& b0, int& c0) {
    n = t / G.tiles;
    int r = t - n * G.tiles;
    const int tc = r % G.tcn; r /= G.tcn;
    const int tb = r % G.tbn; const int ta = r / G.tbn;
    a0 = ta * U8_TA; b0 = tb * 8; c0 = tc * 8;
  };
  // per tile: the sample's descriptor, the byte offset of the tile origin and the three per-lane offsets (out of range where the halo
  // voxel lies outside the volume, or there is no such tile); per chunk only the scalar offset moves
  u8i32x4 h_rs = u8_make_rsrc(P.in, (uint32_t)G.in_sample_bytes);
  int h_soff = 0, h_v[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  auto halo_tile = [&](int t, bool live) {
    int n, a0, b0, c0;
    tile_origin(live ? t : 0, n, a0, b0, c0);
    h_rs = u8_make_rsrc(P.in + (int64_t)n * G.in_sample_bytes, (uint32_t)G.in_sample_bytes);
    h_soff = (a0 * G.ia + b0 * G.ib + c0 * G.ic) * P.g_ld * 2;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ai = a0 + (h_abc[r] & 0xff), bi = b0 + ((h_abc[r] >> 8) & 0xff), ci = c0 + (h_abc[r] >> 16);
      h_v[r] = (live && !(U8_ABL & 8) && ai < G.da && bi < G.db && ci < G.dc) ? h_off[r] : (int)0x80000000;
    }
  };
  auto issue_halo = [&](int kc, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int base = ((r == 2 && wave == 7) ? 512 : r * 512) + wave * 64;
      if (U8_ABL & 128) continue;
      u8_raw_buffer_load_lds(h_rs, (u8_lds_u32_ptr)(sH + buf * U8_HBUF + base * 16), 16, h_v[r], h_soff + kc * 64, 0, 0);
    }
  };

  // ---- weight stream: stage s of a chunk = positions 5 s .. 5 s + 4 (the last stage: 3 / 4 real ones); piece = tid + 512 g, g < 3:
  // block (tid >> 8) + 2 g of the stage, the sixth of waves 4-7 repeats their g = 1 piece; LDS image lane-linear, the swizzle
  // (16-byte slot ^ ((row >> 1) & 3)) on the source side.  The byte offset of every piece from the packed operand's base is a
  // per-thread constant (out of range for the positions a group does not have); the chunk moves a scalar --------------------------
  // swizzle: 16-byte slot ^ ((row >> 1) & 3).  (tools/probes/probe_lds_b128_patterns.hip: with the 64-byte row pitch of a 32-k block
  // this is the conflict-free one — 116 TB/s of ds_read_b128 over the chip, as the linear pattern; ^ ((row >> 2) & 3), ^ (row & 3) or
  // none: 75-77 TB/s.  The first layout used (row >> 2): 47 % of its LDS cycles were bank conflicts, all on the weight fragments.)
  const int w_row = (tid & 255) >> 2, w_src = ((tid & 3) ^ ((w_row >> 1) & 3)) * 8;
  const u8i32x4 w_rs = u8_make_rsrc(P.w, 0x7fffffffu);
  __syncthreads();                                   // tables visible
  int w_v[U8_NS][3];
#pragma unroll
  for (int s = 0; s < U8_NS; ++s)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int ge = (g == 2 && wave >= 4) ? 1 : g;
      const int pos = U8_ST * s + (tid >> 8) + 2 * ge;
      w_v[s][g] = (pos < NPOS && !(U8_ABL & 4)) ? (sTab[pos] + w_row * sTab[16 + pos] + w_src) * 2 : (int)0x80000000;        // < 2^31 (host-checked)
    }
  auto issue_w = [&](int kc, auto SC, int slot) __attribute__((always_inline)) {       // stage s of chunk kc -> ring slot
    constexpr int s = decltype(SC)::value;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int ge = (g == 2 && wave >= 4) ? 1 : g;
      if (U8_ABL & 128) continue;
      u8_raw_buffer_load_lds(w_rs, (u8_lds_u32_ptr)(sW + slot * U8_STAGE + (wave * 64 + ge * 512) * 16), 16, w_v[s][g], kc * 64, 0, 0);
    }
  };

  // ---- per-lane constants -----------------------------------------------------------------------------------------------------
  int pdb, pc;
  u8_patch_voxel(r16, pdb, pc);
  const int trip = wave >> 1, ch = wave & 1;                // row-tile triple, column half
  int abase[3], va[3], vb[3];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int r = 3 * trip + rt;                            // row tile 0..11 = (a plane, b pair)
    va[rt] = r >> 2;
    vb[rt] = 2 * (r & 3) + pdb;
    abase[rt] = ((va[rt] * 9 + vb[rt]) * 10 + pc + 1) * 16 + q4 * U8_PLANE;
  }
  // halo byte offset of shift delta = (dx,dy,dz) bits: its components along the tile axes (a, b, c) = (z, x, y) or (x, y, z)
  auto doff = [](int d) constexpr -> int {
    const int dx = (d >> 2) & 1, dy = (d >> 1) & 1, dz = d & 1;
    return ZXY ? ((dz * 9 + dx) * 10 + dy) * 16 : ((dx * 9 + dy) * 10 + dz) * 16;
  };
  int wro[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) wro[j] = ((ch * 2 + j) * 16 + r16) * 64 + ((q4 ^ ((r16 >> 1) & 3)) << 4);
  const bool af32 = P.add_f32 != 0;
  const int ASZ = af32 ? 4 : 2;
  float bias[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[j][e] = (P.bias != nullptr) ? P.bias[(ch * 2 + j) * 16 + 4 * q4 + e] : 0.f;
  float wsum[2][4], wsq[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { wsum[j][e] = 0.f; wsq[j][e] = 0.f; }
  int stat_n = -1;
  auto flush_stats = [&](int n) {                    // called by every thread (contains barriers)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = wsum[j][e], b = wsq[j][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          const int c = (ch * 2 + j) * 16 + 4 * q4 + e;
          sStats[(wave * 2 + 0) * U8_CN + c] = a;
          sStats[(wave * 2 + 1) * U8_CN + c] = b;
        }
        wsum[j][e] = 0.f;
        wsq[j][e] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * U8_CN) {
      const int which = tid / U8_CN, c = tid % U8_CN, half = c >> 5;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) a += sStats[((2 * k + half) * 2 + which) * U8_CN + c];      // the four waves of the column half
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      P.stats[(slot_t * 2 + which) * P.stats_ld + c] = a;
    }
    __syncthreads();
  };

  f32x4 acc[NCLS][3][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
      for (int rt = 0; rt < 3; ++rt)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[c][rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // one ring stage: positions 5 S .. 5 S + 4 of the group against the halo chunk in buffer hb, weights in ring slot `slot`
  auto compute = [&](auto SC, int hb, int slot) __attribute__((always_inline)) {
    constexpr int S = decltype(SC)::value;
    const char* hs = sH + hb * U8_HBUF;
    const char* ws = sW + slot * U8_STAGE;
    u32x4 xf[3];
    auto tap = [&](auto TC) __attribute__((always_inline)) {        // (position, shift and class are compile-time: acc[C] is a register name)
      constexpr int tpi = decltype(TC)::value, pos = U8_ST * S + tpi;
      if constexpr (pos < NPOS) {
        constexpr int D = u8_pos_find(GRP, pos, 0), C = u8_pos_find(GRP, pos, 1);
        if constexpr (tpi == 0 || D != u8_pos_find(GRP, pos > 0 ? pos - 1 : 0, 0)) {
#pragma unroll
          for (int rt = 0; rt < 3; ++rt)
            xf[rt] = (U8_ABL & (2 | 16)) ? u32x4{(uint32_t)(D + rt), 1u, 2u, 3u} : *reinterpret_cast<const u32x4*>(hs + abase[rt] + doff(D));
        }
        u32x4 wf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wf[j] = (U8_ABL & (2 | 32)) ? u32x4{(uint32_t)(j + tpi), 3u, (uint32_t)C, 5u} : *reinterpret_cast<const u32x4*>(ws + tpi * U8_TAPB + wro[j]);
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr ((U8_ABL & 1) != 0) acc[C][rt][j][0] += __builtin_bit_cast(f32x4, wf[j])[0] * __builtin_bit_cast(f32x4, xf[rt])[1];
            else mma16<H>(acc[C][rt][j], wf[j], xf[rt]);
          }
      }
    };
    tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{}); tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{}); tap(std::integral_constant<int, 4>{});
  };
  // end of a stage: everything but the newest NB vector-memory operations has landed
  auto stage_end = [&](auto NBC) __attribute__((always_inline)) {
    constexpr int NB = decltype(NBC)::value;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NB) : "memory");
    if (!(U8_ABL & 64)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- epilogue of one tile: class c, lane = (voxel of row tile rt, channels (2 ch + j) * 16 + 4 q4 .. + 3) -------------------------
  const int ychunk = (q4 & 1) * 2 + (q4 >> 1);          // the 8-channel chunk (of the wave's 32 columns) a lane holds behind the swap
  int ooff[3], aoff[3];
#pragma unroll
  for (int rt = 0; rt < 3; ++rt) {
    const int vox = va[rt] * G.oa + vb[rt] * G.ob + pc * G.oc;
    ooff[rt] = (vox * P.o_ld + ch * 32 + ychunk * 8) * 2;
    aoff[rt] = (vox * P.add_ld + ch * 32 + 4 * q4) * ASZ;
  }
  auto epilogue = [&](int n, int a0, int b0, int c0) {
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(P.out + (int64_t)n * G.out_sample_bytes, 0, G.out_sample_bytes, 0x00020000);
    bool rv[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) rv[rt] = (a0 + va[rt] < G.da) && (b0 + vb[rt] < G.db) && (c0 + pc < G.dc);
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const ctseg_conv_class& K = P.cls[u8_gclass(GRP, c)];
      const int cb = a0 * G.oa + b0 * G.ob + c0 * G.oc + (K.ox * P.Yo + K.oy) * P.Zo + K.oz;
      // the class's four addend pieces are requested together (buffer loads: out-of-range offsets for voxels the tile does not own),
      // so their round trips overlap instead of following one another
      f32x4 ad[3][2];
      if (P.add != nullptr) {
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.add) + (int64_t)n * G.add_sample_bytes, 0, G.add_sample_bytes, 0x00020000);
        const int asoff = cb * P.add_ld * ASZ;
#pragma unroll
        for (int rt = 0; rt < 3; ++rt)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int vo = rv[rt] ? aoff[rt] + j * 16 * ASZ : (int)0x80000000;
            if (af32) ad[rt][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, asoff, 0));
            else {
              const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(ars, vo, asoff, 0);
              ad[rt][j] = f32x4{h2f<H>(w2[0] & 0xffffu), h2f<H>(w2[0] >> 16), h2f<H>(w2[1] & 0xffffu), h2f<H>(w2[1] >> 16)};
            }
          }
      }
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) {
        u32x2 o2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[c][rt][j][e] + bias[j][e];
            if (STATS && rv[rt]) { wsum[j][e] += v[e]; wsq[j][e] += v[e] * v[e]; }
          }
          if (P.add != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += ad[rt][j][e];
          }
          o2[j] = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
        }
        // v_permlane16_swap (conv_halo_sw.hip): a lane ends up with 8 consecutive channels -> one 16-byte store per row tile
        const auto s0 = __builtin_amdgcn_permlane16_swap(o2[0][0], o2[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(o2[0][1], o2[1][1], false, false);
        const u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(o4, ors, rv[rt] ? ooff[rt] : (int)0x80000000, cb * P.o_ld * 2, 0);
        asm volatile("s_nop 1" ::"v"(o4));              // (the >64-bit store-data hazard: conv_halo_sw.hip)
      }
    }
  };

  // tile sequence: each XCD owns a contiguous range of tiles (neighbouring halos share that XCD's L2); both class groups walk the
  // same sequence (workgroups sid and gxs + sid sit on the same XCD when gxs is a multiple of 8: the second reads the halo from L2)
  const int GX = G.gxs;
  int first, stride, last;
  if ((GX & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = sid & 7;
    first = xcd * chunk + (sid >> 3);
    stride = GX >> 3;
    last = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  } else {
    first = sid; stride = GX; last = total_tiles;
  }

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using N9 = std::integral_constant<int, 9>; using N12 = std::integral_constant<int, 12>;
  // Stage S of the global sequence (3 per chunk, chunks of consecutive tiles back to back) lives in ring slot S % 5 and is requested at
  // the start of stage S - 4; the halo chunk of chunk k + 1 is requested at the start of chunk k's first stage, AHEAD of that stage's
  // weight request.  At the end of a stage the next one must have landed: the newest operations then are three weight requests
  // (9 operations) plus, at the end of a chunk's first and second stage, the halo request (12); the end of a chunk's last stage also
  // needs that halo: 9.
  int t = first, hcur = 0, slot = 0;                   // slot of the stage about to be computed
  auto slot_of = [&](int ahead) { const int s = slot + ahead; return s >= U8_SLOTS ? s - U8_SLOTS : s; };
  if (t < last) {
    halo_tile(t, true);
    issue_halo(0, 0);
    issue_w(0, I0{}, 0);
    issue_w(0, I1{}, 1);
    issue_w(0, I2{}, 2);
    issue_w(1, I0{}, 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; t < last; t += stride) {
    const int tn = t + stride;
    int n, a0, b0, c0;
    tile_origin(t, n, a0, b0, c0);
    if (STATS && n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = n;
    }
    zero_acc();
#pragma unroll 1
    for (int kc = 0; kc < NKC; ++kc) {
      const int k1 = kc + 1 < NKC ? kc + 1 : kc + 1 - NKC, k2 = kc + 2 < NKC ? kc + 2 : kc + 2 - NKC;     // chunks of the stages 4 ahead (NKC >= 4)
      // stage 0: the next chunk's halo (of the next tile behind the last chunk), then stage 3 kc + 4 = (k1, 1)
      if (kc + 1 == NKC) halo_tile(tn, tn < last);
      issue_halo(k1, hcur ^ 1);
      issue_w(k1, I1{}, slot_of(4));
      compute(I0{}, hcur, slot);
      stage_end(N12{});
      slot = slot_of(1);
      issue_w(k1, I2{}, slot_of(4));                  // stage 3 kc + 5 = (k1, 2)
      compute(I1{}, hcur, slot);
      stage_end(N12{});
      slot = slot_of(1);
      issue_w(k2, I0{}, slot_of(4));                  // stage 3 kc + 6 = (k2, 0)
      compute(I2{}, hcur, slot);
      stage_end(N9{});
      slot = slot_of(1);
      hcur ^= 1;
    }
    epilogue(n, a0, b0, c0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last prefetches (never read) land before the workgroup's LDS is released
  if (STATS && stat_n >= 0) flush_stats(stat_n);
}

template <typename H, bool STATS, bool ZXY>     // H = 16-bit storage kind (BF16 / F16)
__global__ __launch_bounds__(U8_NTHR) void conv_up8_kernel(const ConvKArgs P, const Up8Geom G, int total_tiles) {
  __shared__ __attribute__((aligned(16))) char smem[U8_TOTAL];
  if ((int)blockIdx.x < G.gxs) conv_up8_body<H, STATS, ZXY, 0>(P, G, total_tiles, smem);     // (workgroup-uniform)
  else conv_up8_body<H, STATS, ZXY, 1>(P, G, total_tiles, smem);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static void up8_geom(const ConvKArgs& a, Up8Geom& g) {
  const int ta = U8_TA;
  const int dims[3] = {a.Xr, a.Yr, a.Zr};
  const int istr[3] = {a.Yi * a.Zi, a.Zi, 1};
  const int ostr[3] = {a.Yo * a.Zo * 2, a.Zo * 2, 2};
  auto waste = [&](int pa, int pb, int pc) {
    const double full = (double)dims[0] * dims[1] * dims[2];
    const double padded = (double)((dims[pa] + ta - 1) / ta * ta) * ((dims[pb] + 7) / 8 * 8) * ((dims[pc] + 7) / 8 * 8);
    return padded / full;
  };
  int pa = 0, pb = 1, pc = 2;
  if (waste(2, 0, 1) < waste(0, 1, 2) - 1e-9) { pa = 2; pb = 0; pc = 1; }
  g.pa = pa; g.pb = pb; g.pc = pc;
  g.da = dims[pa]; g.db = dims[pb]; g.dc = dims[pc];
  g.ia = istr[pa]; g.ib = istr[pb]; g.ic = istr[pc];
  g.oa = ostr[pa]; g.ob = ostr[pb]; g.oc = ostr[pc];
  const int tan = (g.da + ta - 1) / ta;
  g.tbn = (g.db + 7) / 8; g.tcn = (g.dc + 7) / 8;
  g.tiles = tan * g.tbn * g.tcn;
  g.in_sample_bytes = (int)((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2);
  g.out_sample_bytes = (int)((int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * 2);
  g.add_sample_bytes = (int)((int64_t)a.Xo * a.Yo * a.Zo * a.add_ld * (a.add_f32 ? 4 : 2));
  const int total = g.tiles * a.N;
  g.gxs = persistent_grid(CTSEG_NUM_CU / 2, total);   // 2 x 128 workgroups = one per CU, the reference's 256 tiles twice each
}

bool conv_up8_eligible(const ConvKArgs& a, int dtype, int nclass) {
  { const char* e = getenv("CTSEG_UP8"); if (e != nullptr && e[0] == '0') return false; }   // (A/B switch)
  if (!is16(dtype) || nclass != 8 || a.sin != 1 || a.sout != 2 || a.out_f32) return false;
  if (a.Cn != U8_CN || a.Cn_store != U8_CN || a.Cg < 128 || (a.Cg % 32) != 0 || a.out2 != nullptr) return false;
  if ((a.g_ld % 8) != 0 || ((uintptr_t)a.in % 16) != 0 || ((uintptr_t)a.w % 16) != 0 || (a.o_ld % 8) != 0 || ((uintptr_t)a.out % 16) != 0) return false;
  if (a.add != nullptr && ((a.add_ld % 4) != 0 || ((uintptr_t)a.add % 16) != 0)) return false;
  if (a.Xr != a.Xi || a.Yr != a.Yi || a.Zr != a.Zi) return false;
  if (a.Xo != 2 * a.Xr || a.Yo != 2 * a.Yr || a.Zo != 2 * a.Zr) return false;
  if ((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2 >= (1ll << 31) || (int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * 2 >= (1ll << 31)) return false;
  if (a.add != nullptr && (int64_t)a.Xo * a.Yo * a.Zo * a.add_ld * (a.add_f32 ? 4 : 2) >= (1ll << 31)) return false;
  if ((int64_t)a.Xr * a.Yr * a.Zr < 2048) return false;                            // too few tiles to fill the chip: generic kernel
  int taps = 0;
  for (int c = 0; c < 8; ++c) {
    const ctseg_conv_class& k = a.cls[c];
    if (k.ntaps != (1 << __builtin_popcount(c)) || k.kpad < k.ntaps * a.Cg || (k.kpad % 8) != 0 || (k.w_off % 8) != 0) return false;
    if (k.w_off + (int64_t)a.Cn * k.kpad >= (1ll << 30)) return false;          // 32-bit byte offsets from the operand's base
    taps += k.ntaps;
    if (k.ox != ((c >> 2) & 1) || k.oy != ((c >> 1) & 1) || k.oz != (c & 1)) return false;
    for (int dl = 0; dl < 8; ++dl) {       // the delta-major schedule addresses a class's taps by (dx,dy,dz), in the host's order
      const int tp = u8_tap_index(c, dl);
      if (tp < 0) continue;
      if (tp >= k.ntaps) return false;
      const int t = k.taps[tp];
      if ((int)(int8_t)(t & 0xff) != ((dl >> 2) & 1) || (int)(int8_t)((t >> 8) & 0xff) != ((dl >> 1) & 1) || (int)(int8_t)((t >> 16) & 0xff) != (dl & 1)) return false;
    }
  }
  return taps == 27;
}

// InstanceNorm partial slots per sample: one per workgroup (both class groups)
int conv_up8_slots(const ConvKArgs& a) {
  Up8Geom g;
  up8_geom(a, g);
  return 2 * g.gxs;
}

void launch_conv_up8(ConvKArgs& a, hipStream_t st) {
  Up8Geom g;
  up8_geom(a, g);
  a.tiles = g.tiles;
  const int total = g.tiles * a.N;
  const dim3 grid((unsigned)(2 * g.gxs)), blk(U8_NTHR);
  const bool stats = a.stats != nullptr;
  const bool zxy = g.pa == 2;
#define CTSEG_U8_GO(H, ST)                                                                                        \
  do {                                                                                                            \
    if (zxy) hipLaunchKernelGGL((conv_up8_kernel<H, ST, true>), grid, blk, 0, st, a, g, total);                    \
    else hipLaunchKernelGGL((conv_up8_kernel<H, ST, false>), grid, blk, 0, st, a, g, total);                       \
  } while (0)
  if (a.dtype == CTSEG_F16) { if (stats) CTSEG_U8_GO(F16, true); else CTSEG_U8_GO(F16, false); }
  else { if (stats) CTSEG_U8_GO(BF16, true); else CTSEG_U8_GO(BF16, false); }
#undef CTSEG_U8_GO
}

}  // namespace ctseg
