// Weight gradient of the many-channel layers as a ring-pipelined GEMM over voxels (bf16, gfx950):
//   R[slot * Cg + a][b] = sum_rows in[row * sin + d(slot)][a] * dy[row][b]        (same contract and slab layout as conv_wgrad.hip;
// autograd's Conv3d / ConvTranspose3d weight + bias backward in the reference's training step, capstone/volumetric/base_trainer.py:80-82).
//
// Why a second kernel (measurements: DESIGN.md 3.2k).  These GEMMs are bound by the bytes a workgroup stages from L2 per flop
// (tools/probes/probe_l2_to_cu_bandwidth.hip: a CU pulls ~50 B/clk of contiguous, ~30 B/clk of 64-byte pieces), so the tile is
// 256 x 256 / 256 x 128 / 512 x 64 instead of conv_wgrad_kernel's 128 x 128; and tools/probes/probe_mfma_lds.hip says what the matrix
// pipe tolerates beside a v_mfma_f32_16x16x32_bf16: two ds_read_b64_tr_b16 per multiply cost nothing when they are software-pipelined
// (90-97 % of the MFMA rate with two waves per SIMD against 76-80 % when a stage is read, waited for and multiplied in turn), up to
// two VALU instructions per multiply are free, four cost 40 %.  conv_wgrad_kernel runs ~6 VALU per multiply, reads and waits in front
// of its multiplies, and hipcc drains its direct-to-LDS prefetch (vmcnt(0)) in front of the first fragment read.  This kernel:
//   * 512 threads, ONE workgroup per CU, wave tile KT x CT blocks of 16 x 16 (8 x 4 or 4 x 4): 32 or 16 multiplies per 32-voxel stage;
//   * a stage is 32 voxels of both operands in 128-column panels of [32 rows][256 B] (conv_wgrad.hip's XOR swizzle, applied on the
//     source side); four stages live in an LDS ring, stage s+3 is requested during step s (raw.buffer.load.lds, counted vmcnt);
//   * per-row addressing comes from a table in LDS (byte offset + 27-neighbour mask of 512 rows at a time): a request costs an add, a
//     bit-field extract and an OR instead of a chain of coordinate carries and bounds tests;
//   * fragments of stage s+1 are read while stage s multiplies (two register sets; the 8 x 4 tile refills one gathered set in halves);
//     the reads are inline assembly, waited for by the step itself (see wr_tr2);
//   * the bias row (a virtual all-ones gathered channel at K index ntaps*Cg) is a fragment constant ORed into the K-padding row.
// Rows past the end of a split or of the sample are zero rows of dy (its buffer range IS the split), so a workgroup runs a uniform
// number of stages whatever its split holds.
#include <type_traits>

#include "ctseg_dev.h"

#ifndef WR_ABL
#define WR_ABL 0     // timing-only ablation builds (tools/ablate_wgrad_ring.sh; results are garbage): 1 no MFMAs, 2 no transposed fragment reads, 4 no gathered-operand loads, 8 no dy loads, 16 no step barrier, 32 no bounds tests / coordinate carries
#endif

namespace ctseg {

struct WRingKArgs {
  const char* in;
  const char* dy;
  float* ws;
  int N, Xi, Yi, Zi, Xr, Yr, Zr;
  int Cg, Cn, g_ld, d_ld, sin, ntaps;
  int rows, splits, rows_per_split, nst;   // nst: stages every workgroup runs (even, >= 4)
  int kpad_w, cn_pad, d_valid;
  int cx5, cy5, cz5;                       // mixed-radix decomposition of a 512-row step (one chunk of the row table)
  int ktiles, ctiles;                      // workgroup -> (tile, slab): ids L, L+8, ... of a group of 8*ktiles*ctiles share a slab (one XCD)
  int xcd;                                 // 1: that order; 0: tile = blockIdx.x, slab = blockIdx.y
  int taps[CTSEG_MAX_TAPS];
};

typedef int32_t wr_i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* wr_lds_u32_ptr;
__device__ void wr_raw_buffer_load_lds(wr_i32x4 rsrc, wr_lds_u32_ptr lds, int size, int voffset, int soffset, int offset,
                                       int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
__device__ __forceinline__ wr_i32x4 wr_make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  wr_i32x4 v = __builtin_bit_cast(wr_i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]);
  v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]);
  v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

// Two transposed LDS reads = one bf16 MFMA operand (rows r and r + 16 of the stage).
// hipcc waits for EVERY outstanding direct-to-LDS load (vmcnt(0)) in front of a __builtin_amdgcn_ds_read_tr16_b64 (the builtin carries
// no memory operand it could tell apart from the ring slots being filled), which would empty the ring every step.  The reads are
// therefore inline assembly: the compiler neither orders them against the loads nor waits for their data -- the step does, with
// lgkmcnt(0) behind its multiplies and a scheduling barrier on either side.  tests/test_isa_hazards.py disassembles the shipped
// library and checks, over each kernel's control-flow graph, that no register such a read writes is consumed before that wait.
template <int HI> __device__ __forceinline__ bf16x8 wr_tr2(uint32_t addr) {
  u32x2 lo, hi;
  if constexpr ((WR_ABL & 2) != 0) {
    asm volatile("" : "=v"(lo), "=v"(hi) : "v"(addr));
  } else {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(HI) : "memory");
  }
  u32x4 t = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8, t);
}

template <int WK_, int WC_, int KT_, int CT_, int D_> struct WRingCfg {
  static constexpr int WK = WK_, WC = WC_, KT = KT_, CT = CT_, D = D_;
  static constexpr int KB = WK * KT * 16, BN = WC * CT * 16, NTHR = 512;
  static_assert(WK * WC == 8 && KB % 128 == 0 && (BN % 128 == 0 || BN == 64), "8 waves; 128-row K panels; 128- or 64-column dy panels");
  static constexpr int AP = KB / 128;                           // K panels of [32 rows][256 B]
  static constexpr int DP = BN >= 128 ? BN / 128 : 1;           // dy panels
  static constexpr int DROW = BN >= 128 ? 256 : 128;            // bytes per dy panel row
  static constexpr int DOFF = AP * 8192, STAGE = DOFF + DP * 32 * DROW;
  static constexpr int RING = D * STAGE;
  static constexpr int TAB = 2 * 512 * 8;                       // row table: two chunks of 512 rows x {byte offset, out-of-volume mask}
  static constexpr int TOTAL = RING + TAB;
  static_assert(TOTAL <= 160 * 1024, "LDS");
};

// ND: dy panels THIS wave stages (BN = 64: one 4 KiB panel = waves 0-3 only)
template <typename C, int ND>
__device__ __forceinline__ void wring_main(const WRingKArgs& P, char* smem, f32x4 (&acc)[C::KT][C::CT], int n, int sp, int kb0, int col0,
                                           int wave, int lane, int tid) {
  constexpr int KT = C::KT, CT = C::CT, D = C::D, AP = C::AP, STAGE = C::STAGE, DOFF = C::DOFF;
  const int wk = wave / C::WC, wc = wave % C::WC;
  const int r16 = lane & 15, q4 = lane >> 4;

  // ---- this thread's staging position: voxel row tid / 16 of a stage, 16-byte slot tid % 16 of every panel -----------------------
  const int srow = tid >> 4, spos = tid & 15;
  const int gl = P.g_ld * 2;
  const int ktot = P.ntaps * P.Cg;
  // Per panel: byte delta of this slot's tap and channel, and WHICH bit of a row's out-of-volume mask decides whether the chunk is
  // read: bit (dx+1)*9 + (dy+1)*3 + (dz+1) for tap (dx, dy, dz) (every tap offset is -1, 0 or +1: host-checked); bit 31 -- always
  // set -- for K padding and the bias row.
  int tapoff[AP], tapbit[AP];
#pragma unroll
  for (int p = 0; p < AP; ++p) {
    const int chunk = ((((spos >> 1) ^ (srow & 7)) << 1) | (spos & 1));          // logical chunk this LDS slot holds (source-side swizzle)
    const int kpos = kb0 + p * 128 + chunk * 8;
    const bool kval = kpos < ktot;
    const int slot = kval ? kpos / P.Cg : 0, ci = kval ? kpos - slot * P.Cg : 0;
    int t = 0;      // (a select chain over the kernel-argument table: once per thread and panel, and the table stays in SGPRs)
#pragma unroll
    for (int q = 0; q < CTSEG_MAX_TAPS; ++q) t = (q == slot) ? P.taps[q] : t;
    const int dx = (int)(int8_t)(t & 0xff), dy = (int)(int8_t)((t >> 8) & 0xff), dz = (int)(int8_t)((t >> 16) & 0xff);
    tapoff[p] = ((dx * P.Yi + dy) * P.Zi + dz) * gl + ci * 2;
    tapbit[p] = kval ? (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1) : 31;
  }
  const int mstart = sp * P.rows_per_split;
  int mend = mstart + P.rows_per_split;
  if (mend > P.rows) mend = P.rows;

  // ---- row table.  What a gathered chunk needs per voxel row -- the row's byte offset inside the sample and whether each of its 27
  // neighbours lies inside the volume -- used to be recomputed per thread and stage as a chain of coordinate carries and bounds tests
  // (~40 dependent VALU / SALU instructions; with them compiled out the launch runs 24-34 % faster: tools/ablate_wgrad_ring.sh,
  // WR_ABL = 32).  Now the workgroup keeps it in LDS for 512 rows (16 stages) at a time: thread t owns row 512 c + t of chunk c, walks its
  // coordinates from chunk to chunk (carries, no division) and writes {offset, ~mask27}; mask27 = the product of three 3-bit axis
  // masks spread to bit strides 9, 3 and 1 (no carries between them).  A request then costs a bit-field extract, an add and an OR.
  u32x2* const tab = reinterpret_cast<u32x2*>(smem + C::RING);
  int tx, ty, tz, toff;          // scaled coordinates and byte offset of row mstart + 512 c + tid
  {
    const int m = mstart + tid;
    tz = m % P.Zr; const int t = m / P.Zr;
    ty = t % P.Yr; tx = t / P.Yr;
    tx *= P.sin; ty *= P.sin; tz *= P.sin;
    toff = ((tx * P.Yi + ty) * P.Zi + tz) * gl;
  }
  const int zrs = P.Zr * P.sin, yrs = P.Yr * P.sin, xrs = P.Xr * P.sin;
  const int c5z = P.cz5 * P.sin, c5y = P.cy5 * P.sin, c5x = P.cx5 * P.sin;
  const int o5 = (c5z + (c5y + c5x * P.Yi) * P.Zi) * gl;
  const int o_cz = P.sin * gl * (P.Zi - P.Zr), o_cy = P.sin * gl * P.Zi * (P.Yi - P.Yr);
  auto table_chunk = [&](int c) {
    auto axis = [&](int v, int lim, int stride) -> uint32_t {           // bits 0, stride, 2 stride: v-1, v, v+1 inside [0, lim)
      return ((unsigned)(v - 1) < (unsigned)lim ? 1u : 0u) | ((unsigned)v < (unsigned)lim ? 1u << stride : 0u) |
             ((unsigned)(v + 1) < (unsigned)lim ? 1u << (2 * stride) : 0u);
    };
    uint32_t m27 = axis(tz, P.Zi, 1) * axis(ty, P.Yi, 3) * axis(tx, P.Xi, 9);
    if (tx >= xrs) m27 = 0u;                                            // a row past the end of the sample
    tab[(c & 1) * 512 + tid] = u32x2{(uint32_t)toff, ~m27};
    tz += c5z;
    const bool carry_z = tz >= zrs;
    tz -= carry_z ? zrs : 0;
    ty += c5y + (carry_z ? P.sin : 0);
    const bool carry_y = ty >= yrs;
    ty -= carry_y ? yrs : 0;
    tx += c5x + (carry_y ? P.sin : 0);
    toff += o5 + (carry_z ? o_cz : 0) + (carry_y ? o_cy : 0);
  };
  // the entry of this thread's row in stage q (inline assembly like the fragment reads: the step's own lgkmcnt(0) covers it)
  const uint32_t tab0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem) + C::RING + srow * 8;
  auto entry = [&](int q) -> u32x2 {
    u32x2 e;
    const uint32_t a = tab0 + (((q >> 4) & 1) * 512 + (q & 15) * 32) * 8;
    asm volatile("ds_read_b64 %0, %1" : "=v"(e) : "v"(a) : "memory");
    return e;
  };
  const char* inb = P.in + (int64_t)n * P.Xi * P.Yi * P.Zi * gl;
  const wr_i32x4 rsA = wr_make_rsrc(inb, (uint32_t)((int64_t)P.Xi * P.Yi * P.Zi * gl));
  const char* dstage = P.dy + ((int64_t)n * P.rows + mstart) * P.d_ld * 2;
  const wr_i32x4 rsD = wr_make_rsrc(dstage, (uint32_t)((int64_t)(mend > mstart ? mend - mstart : 0) * P.d_ld * 2));
  // dy: panel p, this thread's row and chunk; a column past d_valid starts (and stays) out of range
  int doff[ND > 0 ? ND : 1];
#pragma unroll
  for (int p = 0; p < ND; ++p) {
    int drow, dcol;
    if constexpr (C::BN >= 128) {
      drow = srow;
      dcol = col0 + p * 128 + ((((spos >> 1) ^ (srow & 7)) << 1) | (spos & 1)) * 8;
    } else {
      drow = tid >> 3;                                                              // 8 slots per 128-byte row, threads 0..255
      const int pos = tid & 7;
      dcol = col0 + ((((pos >> 1) ^ ((drow >> 1) & 3)) << 1) | (pos & 1)) * 8;
    }
    doff[p] = dcol < P.d_valid ? (drow * P.d_ld + dcol) * 2 : (int)0x80000000;
  }
  const int d_step = 32 * P.d_ld * 2;

  auto dma = [&](int sl, const u32x2& e) {
    char* dst = smem + sl * STAGE + wave * 1024;
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int vo = ((int)e[0] + tapoff[p]) | __builtin_amdgcn_sbfe((int)e[1], tapbit[p], 1);     // outside the volume: 0xffffffff -> zeros
      if constexpr ((WR_ABL & 4) == 0) wr_raw_buffer_load_lds(rsA, (wr_lds_u32_ptr)(dst + p * 8192), 16, vo, 0, 0, 0);
      else asm volatile("" ::"v"(vo));
    }
#pragma unroll
    for (int p = 0; p < ND; ++p) {
      if constexpr ((WR_ABL & 8) == 0) wr_raw_buffer_load_lds(rsD, (wr_lds_u32_ptr)(dst + DOFF + p * 8192), 16, doff[p], 0, 0, 0);
      doff[p] += d_step;
    }
  };

  // ---- fragment addresses: lane supplies row 4*q4 + (r16>>2) [+16], columns 4*(r16&3).. of a 16-column block ------------------------
  const int mrow = 4 * q4 + (r16 >> 2), pc2 = (r16 & 3) * 8;
  int fa[KT], fd[CT];     // byte offset of this wave's block i / j inside a stage (panel, swizzled 32-byte slot, lane's 8 bytes)
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    const int gb = wk * KT + i;
    fa[i] = (gb >> 3) * 8192 + mrow * 256 + (((gb & 7) ^ (mrow & 7)) << 5) + pc2;
  }
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int gb = wc * CT + j;
    if constexpr (C::BN >= 128) fd[j] = DOFF + (gb >> 3) * 8192 + mrow * 256 + (((gb & 7) ^ (mrow & 7)) << 5) + pc2;
    else fd[j] = DOFF + mrow * 128 + (((gb & 3) ^ ((mrow >> 1) & 3)) << 5) + pc2;
  }
  // ---- bias row: K index ktot lives in block bias_i of this wave (or in none).  The gathered operand's row there is K padding (all
  // zeros), so the wave that owns it ORs a constant fragment -- 1.0 in that row, every voxel -- into the fragment it has just read.
  int bias_i = -1;
  uint32_t one2;       // two bf16 1.0 in the lanes that hold row (ktot - kb0) % 16 of a block, 0 elsewhere
  {
    const int rel = ktot - kb0 - wk * KT * 16;
    if (rel >= 0 && rel < KT * 16) bias_i = rel >> 4;
    one2 = (r16 == (rel & 15)) ? 0x3f803f80u : 0u;
  }
  bias_i = __builtin_amdgcn_readfirstlane(bias_i);
  auto add_bias = [&](bf16x8 (&af)[KT], int i0, int i1) {       // blocks i0 .. i1-1 of `af` have landed
    if (bias_i >= 0) {
#pragma unroll
      for (int i = 0; i < KT; ++i)
        if (i >= i0 && i < i1 && i == bias_i) {
          u32x4 t = __builtin_bit_cast(u32x4, af[i]);
          t[0] |= one2; t[1] |= one2; t[2] |= one2; t[3] |= one2;
          af[i] = __builtin_bit_cast(bf16x8, t);
        }
    }
  };

  constexpr int CNT = AP + ND;      // this wave's loads per stage
  const int nst = P.nst;            // even, >= D + 2 (host)
  using T_ = std::true_type;
  using F_ = std::false_type;
  static_assert(D >= 3, "stage s+D-1 is requested while stage s multiplies and stage s+1 is read");
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

  // prologue shared by both schedules: chunk 0 of the row table, the entries of stages 0 .. D-1, the requests of stages 0 .. D-2
  u32x2 ecur;
  {
    table_chunk(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    u32x2 e[D];
#pragma unroll
    for (int t = 0; t < D; ++t) e[t] = entry(t);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < D - 1; ++t) dma(t, e[t]);
    ecur = e[D - 1];
  }
  // per step s: chunk c of the table is written at step 16 c - 12 (its first entry is read at step 16 c - D, its buffer's previous
  // tenant was last read at step 16 (c - 1) - D - 1); the entry of stage s + D is read one step before the request that uses it
  auto step_table = [&](int st) -> u32x2 {
    if (((st + 12) & 15) == 0) table_chunk((st + 12) >> 4);
    return entry(st + D);
  };

  if constexpr (KT == 8) {
    // 8 x 4 wave tile: 128 accumulators leave room for ONE set of gathered fragments (8) beside two sets of dy fragments (2 x 4).
    // The gathered set is refilled in halves behind the multiplies that used it: while blocks 0-3 of stage s multiply, blocks 4-7 of
    // stage s are read; while blocks 4-7 multiply, blocks 0-3 and the dy fragments of stage s+1 are read.
    bf16x8 af[KT], d0[CT], d1[CT];
    auto rd_a = [&](int sl, int i0, int i1) {
      const uint32_t base = lds0 + sl * STAGE;
#pragma unroll
      for (int i = 0; i < KT; ++i)
        if (i >= i0 && i < i1) af[i] = wr_tr2<16 * 256>(base + fa[i]);
    };
    auto rd_d = [&](int sl, bf16x8 (&df)[CT]) {
      const uint32_t base = lds0 + sl * STAGE;
#pragma unroll
      for (int j = 0; j < CT; ++j) df[j] = wr_tr2<16 * C::DROW>(base + fd[j]);
    };
    auto mm = [&](int i0, int i1, const bf16x8 (&df)[CT]) {
#pragma unroll
      for (int i = 0; i < KT; ++i)
        if (i >= i0 && i < i1)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            if constexpr ((WR_ABL & 1) == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], df[j], acc[i][j], 0, 0, 0);
            else asm volatile("" : "+v"(acc[i][j]) : "v"(af[i]), "v"(df[j]));
          }
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * CNT) : "memory");
    __builtin_amdgcn_s_barrier();
    rd_a(0, 0, 4);
    rd_d(0, d0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    add_bias(af, 0, 4);
    auto step = [&](auto dma_c, auto fr_c, int st, int sl, bf16x8 (&xd)[CT], bf16x8 (&yd)[CT]) {
      constexpr bool DMA = decltype(dma_c)::value, FR = decltype(fr_c)::value;
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 3) * CNT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr ((WR_ABL & 16) == 0) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const int sl1 = sl + 1 == D ? 0 : sl + 1, slp = sl == 0 ? D - 1 : sl - 1;
      u32x2 enext = ecur;
      if constexpr (DMA) enext = step_table(st);
      rd_a(sl, 4, 8);                         // the other half of THIS stage (its slot is not requested into before step s+1)
      if constexpr (DMA) dma(slp, ecur);
      mm(0, 4, xd);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      add_bias(af, 4, 8);
      if constexpr (FR) { rd_a(sl1, 0, 4); rd_d(sl1, yd); }
      mm(4, 8, xd);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (FR) add_bias(af, 0, 4);
      ecur = enext;
    };
    int sl = 0;
    auto next = [&]() { sl = (sl + 1 == D) ? 0 : sl + 1; };
    static_assert(D == 4, "the drain below is written for a four-deep ring (nst - 3 request steps: one ahead of the two-step loop)");
    step(T_{}, T_{}, 0, sl, d0, d1); next();
    for (int s = 1; s < nst - D + 1; s += 2) {
      step(T_{}, T_{}, s, sl, d1, d0); next();
      step(T_{}, T_{}, s + 1, sl, d0, d1); next();
    }
    step(F_{}, T_{}, 0, sl, d1, d0); next();
    step(F_{}, T_{}, 0, sl, d0, d1); next();
    step(F_{}, F_{}, 0, sl, d1, d0);
  } else {
    bf16x8 a0[KT], d0[CT], a1[KT], d1[CT];
    auto frags = [&](int sl, bf16x8 (&af)[KT], bf16x8 (&df)[CT]) {
      const uint32_t base = lds0 + sl * STAGE;
#pragma unroll
      for (int i = 0; i < KT; ++i) af[i] = wr_tr2<16 * 256>(base + fa[i]);
#pragma unroll
      for (int j = 0; j < CT; ++j) df[j] = wr_tr2<16 * C::DROW>(base + fd[j]);
    };
    auto mmas = [&](const bf16x8 (&af)[KT], const bf16x8 (&df)[CT]) {
#pragma unroll
      for (int i = 0; i < KT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) {
          if constexpr ((WR_ABL & 1) == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], df[j], acc[i][j], 0, 0, 0);
          else asm volatile("" : "+v"(acc[i][j]) : "v"(af[i]), "v"(df[j]));
        }
    };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * CNT) : "memory");
    __builtin_amdgcn_s_barrier();
    frags(0, a0, d0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    add_bias(a0, 0, KT);

    auto step = [&](auto dma_c, auto fr_c, int st, int sl, bf16x8 (&xa)[KT], bf16x8 (&xd)[CT], bf16x8 (&ya)[KT], bf16x8 (&yd)[CT]) {
      constexpr bool DMA = decltype(dma_c)::value, FR = decltype(fr_c)::value;
      if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 3) * CNT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr ((WR_ABL & 16) == 0) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const int sl1 = sl + 1 == D ? 0 : sl + 1, slp = sl == 0 ? D - 1 : sl - 1;
      u32x2 enext = ecur;
      if constexpr (DMA) enext = step_table(st);
      if constexpr (FR) frags(sl1, ya, yd);
      if constexpr (DMA) dma(slp, ecur);
      mmas(xa, xd);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the fragments of stage s+1 and the table entry (read by inline assembly: nobody else waits for them)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (FR) add_bias(ya, 0, KT);
      ecur = enext;
    };
    int sl = 0;
    auto next = [&]() { sl = (sl + 1 == D) ? 0 : sl + 1; };
    {
      static_assert(D == 4, "the drain below is written for a four-deep ring");
      // stages 0 .. nst-1; steps 0 .. nst-D request stage s+D-1, the last D-1 steps only drain.  nst - D + 1 = nst - 3 is odd for even
      // nst, so one step runs ahead of the two-step loop.
      step(T_{}, T_{}, 0, sl, a0, d0, a1, d1); next();
      for (int s = 1; s < nst - D + 1; s += 2) {
        step(T_{}, T_{}, s, sl, a1, d1, a0, d0); next();
        step(T_{}, T_{}, s + 1, sl, a0, d0, a1, d1); next();
      }
      step(F_{}, T_{}, 0, sl, a1, d1, a0, d0); next();
      step(F_{}, T_{}, 0, sl, a0, d0, a1, d1); next();
      step(F_{}, F_{}, 0, sl, a1, d1, a0, d0);
    }
  }
}

template <typename C>
__global__ __launch_bounds__(512) void conv_wgrad_ring_kernel(const WRingKArgs P) {
  constexpr int KT = C::KT, CT = C::CT;
  __shared__ __attribute__((aligned(16))) char smem[C::TOTAL];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, zslab;
  if (P.xcd) {
    const int kc = P.ktiles * P.ctiles, G = 8 * kc;
    const int L = blockIdx.x, g = L / G, r = L - g * G;
    zslab = g * 8 + (r & 7);
    tile = r >> 3;
  } else {
    tile = blockIdx.x; zslab = blockIdx.y;
  }
  const int kt = tile % P.ktiles, ctile = tile / P.ktiles;
  const int kb0 = kt * C::KB, col0 = ctile * C::BN;
  const int n = zslab / P.splits, sp = zslab % P.splits;

  f32x4 acc[KT][CT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (C::BN >= 128) {
    wring_main<C, C::DP>(P, smem, acc, n, sp, kb0, col0, wave, lane, tid);
  } else {
    if (wave < 4) wring_main<C, 1>(P, smem, acc, n, sp, kb0, col0, wave, lane, tid);
    else wring_main<C, 0>(P, smem, acc, n, sp, kb0, col0, wave, lane, tid);
  }

  const int wk = wave / C::WC, wc = wave % C::WC;
  const int r16 = lane & 15, q4 = lane >> 4;
  float* slab = P.ws + (int64_t)zslab * P.kpad_w * P.cn_pad;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = kb0 + (wk * KT + i) * 16 + 4 * q4 + e;
      if (row < P.kpad_w)
#pragma unroll
        for (int j = 0; j < CT; ++j) slab[(int64_t)row * P.cn_pad + col0 + (wc * CT + j) * 16 + r16] = acc[i][j][e];
    }
}

using WRingA = WRingCfg<2, 4, 8, 4, 4>;    // 256 K rows x 256 columns
using WRingB = WRingCfg<4, 2, 4, 4, 4>;    // 256 x 128
using WRingC = WRingCfg<8, 1, 4, 4, 4>;    // 512 x 64

// which tile serves this descriptor: 0 none, 1 = A, 2 = B, 3 = C
static int wring_kind(const ctseg_wgrad_desc* d) {
  const char* e = getenv("CTSEG_WGRAD_RING");          // (read per call: the A/B tools and the parity test flip it inside one process)
  if ((e != nullptr && atoi(e) == 0) || d->dtype != CTSEG_BF16) return 0;
  if (d->Cg % 8 != 0 || d->g_ld % 8 != 0 || d->d_ld % 8 != 0 || ((uintptr_t)d->in % 16) != 0 || ((uintptr_t)d->dy % 16) != 0) return 0;
  if (d->Cg < 32 || d->Cn < 64 || d->ntaps > 32) return 0;
  for (int t = 0; t < d->ntaps; ++t)                        // the row table keeps one bit per neighbour in {-1, 0, 1}^3
    for (int sh = 0; sh < 24; sh += 8) {
      const int o = (int)(int8_t)((d->taps[t] >> sh) & 0xff);
      if (o < -1 || o > 1) return 0;
    }
  if (d->dyn_g != nullptr || d->in_mean_rstd != nullptr) return 0;
  if ((int64_t)d->Xi * d->Yi * d->Zi * d->g_ld * 2 >= ((int64_t)1 << 31) - 4096) return 0;      // 32-bit buffer offsets inside a sample
  const char* ea = getenv("CTSEG_WGRAD_RING_A");       // (A/B: 0 keeps the 256 x 128 tile for the 256-column layers)
  if (d->cn_pad % 256 == 0 && !(ea != nullptr && atoi(ea) == 0)) return 1;
  if (d->cn_pad % 128 == 0) return 2;
  if (d->cn_pad == 64) return 3;
  return 0;
}

bool wgrad_ring_eligible(const ctseg_wgrad_desc* d) { return wring_kind(d) != 0; }

// workgroups one slab (sample x split) takes; bytes a workgroup stages per 32 rows
int wgrad_ring_wgs_per_slab(const ctseg_wgrad_desc* d, int32_t* stage_bytes) {
  const int kind = wring_kind(d);
  if (kind == 0) return 0;
  const int KB = kind == 3 ? 512 : 256, BN = kind == 1 ? 256 : kind == 2 ? 128 : 64;
  const int ktot = d->ntaps * d->Cg;
  if (stage_bytes) *stage_bytes = 32 * (KB + BN) * 2;
  return ((ktot + 1 + KB - 1) / KB) * (d->cn_pad / BN);
}

template <typename C> static void launch_wring(WRingKArgs& a, hipStream_t st) {
  const int ktot = a.ntaps * a.Cg;
  a.ktiles = (ktot + 1 + C::KB - 1) / C::KB;
  a.ctiles = a.cn_pad / C::BN;
  const int zs = a.N * a.splits, tiles = a.ktiles * a.ctiles;
  static const bool remap = !(getenv("CTSEG_WGRAD_XCD") && atoi(getenv("CTSEG_WGRAD_XCD")) == 0);
  a.xcd = (remap && zs % 8 == 0 && tiles > 1) ? 1 : 0;
  const dim3 grid = a.xcd ? dim3((unsigned)(tiles * zs), 1u, 1u) : dim3((unsigned)tiles, (unsigned)zs, 1u);
  hipLaunchKernelGGL((conv_wgrad_ring_kernel<C>), grid, dim3(512), 0, st, a);
}

int launch_wgrad_ring(const ctseg_wgrad_desc* d, hipStream_t st) {
  const int kind = wring_kind(d);
  WRingKArgs a;
  a.in = (const char*)d->in; a.dy = (const char*)d->dy; a.ws = d->ws;
  a.N = d->N; a.Xi = d->Xi; a.Yi = d->Yi; a.Zi = d->Zi; a.Xr = d->Xr; a.Yr = d->Yr; a.Zr = d->Zr;
  a.Cg = d->Cg; a.Cn = d->Cn; a.g_ld = d->g_ld; a.d_ld = d->d_ld; a.sin = d->sin; a.ntaps = d->ntaps;
  const int64_t rows64 = (int64_t)d->Xr * d->Yr * d->Zr;
  if (rows64 >= (1ll << 31) - 4096) return -1;
  a.rows = (int)rows64; a.splits = d->splits;
  int rps = (int)((rows64 + d->splits - 1) / d->splits);
  rps = ((rps + 31) / 32) * 32;
  a.rows_per_split = rps;
  if ((int64_t)(rps + 64) * d->d_ld * 2 >= ((int64_t)1 << 31)) return -2;
  int nst = rps / 32;
  if (nst < 8) nst = 8;          // >= D + 2 for every ring depth instantiated below
  nst += nst & 1;
  a.nst = nst;
  a.kpad_w = d->kpad_w; a.cn_pad = d->cn_pad;
  const int dv = ((d->Cn + 7) / 8) * 8;
  a.d_valid = dv < d->d_ld ? dv : d->d_ld;
  int step = 512;
  a.cz5 = step % d->Zr; step /= d->Zr;
  a.cy5 = step % d->Yr; a.cx5 = step / d->Yr;
  for (int i = 0; i < CTSEG_MAX_TAPS; ++i) a.taps[i] = i < d->ntaps ? d->taps[i] : 0;
  if (kind == 1) launch_wring<WRingA>(a, st);
  else if (kind == 2) launch_wring<WRingB>(a, st);
  else launch_wring<WRingC>(a, st);
  return 0;
}

}  // namespace ctseg
