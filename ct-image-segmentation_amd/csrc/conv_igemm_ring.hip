// 192 x 256 implicit-GEMM tile with a five-deep LDS ring of 32-wide K stages (bf16, gfx950): the many-channel passes
// of the UNet's encoder bottom (Conv3d 64->256 s2, 128->256, 256->256 and their input-gradient passes;
// capstone/volumetric/base_trainer.py:65-72 builds them through monai.networks.nets.UNet).
//
// Same GEMM, tables and epilogue as conv_igemm_kernel<BF16,192,256,2,4>.  What differs is the operand pipeline.
// There a 64-wide stage is fetched while the previous one is consumed and the barrier waits for it: one stage of
// direct-to-LDS traffic in flight, its L2 round trip paid once per stage, and ~100 instructions of branchy address
// code per stage that both waves of a SIMD execute in step while the matrix pipe idles.  Here
//   * a stage is 32 K (28 KiB for the whole tile) and five fit in LDS; stage s+4 is requested during step s, so two
//     to three stages are always in flight and the per-step wait is a counted vmcnt covering only the oldest;
//   * the loads are buffer_load_dwordx4 ... lds: the per-row, per-tap bounds test is one bit of a precomputed mask
//     that selects an out-of-range offset (the hardware then delivers zeros), so a step is ONE basic block
//     (24 MFMA + 10 ds_read_b128 + 3-4 loads + ~20 scalar/vector address ops) that the scheduler interleaves freely;
//   * MFMA fragments of stage s+1 are read while stage s multiplies (two register sets); the slot a load overwrites
//     was last read a full step earlier, so no manual LDS wait is needed and the compiler's counted lgkmcnt stands.
// Measured on the 256->256 layer (2x64x64x6 voxels): 0.147 ms against 0.158 ms for the double-buffered kernel on
// the same box; alone, the loads take 0.113 ms, the multiplies with their fragment reads 0.110 ms.
//
// LDS image of a stage: [192 + 256 rows][64 B].  A direct-to-LDS load is lane-linear (lane l -> bytes 16 l of a 1 KiB
// block = row l/4, slot l%4), so the swizzle is applied to WHICH K chunk a lane fetches: slot = chunk ^ (3 if row&8).
// With it the four 16-lane groups of a ds_read_b128 of an MFMA fragment (16 rows x one chunk per quarter wave) each
// cover all 64 banks once.
#include <type_traits>

#include "conv_common.h"

namespace ctseg {

// BN = 256: waves 2 x 4, wave tile 96 x 64.  BN = 128 (passes with 65..128 columns): waves 4 x 2, wave tile 48 x 64.
template <int BN_> struct RingCfgT {
  static constexpr int BM = 192, BN = BN_, WGM = BN_ == 256 ? 2 : 4, WGN = BN_ == 256 ? 4 : 2, NW = WGM * WGN, NTHR = 64 * NW, D = 5;
  static constexpr int RB = 64, KS = 32;                 // LDS row bytes / K per stage
  static constexpr int BOFF = BM * RB, STAGE = (BM + BN) * RB;
  static constexpr int CROW = BN * 2 + 16;
  static constexpr int MAIN = (D * STAGE > BM * CROW) ? D * STAGE : BM * CROW;
  static constexpr int STATS = WGM * 2 * BN * 4;
  static constexpr int ROWTAB = BM * 8;
  static constexpr int TOTAL = MAIN + STATS + ROWTAB + 64 * 4;
};
using RingCfg = RingCfgT<256>;

typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* lds_u32_ptr;
// buffer_load_dwordx4 ... lds: per-lane 32-bit byte offset into a range-checked buffer; an offset past the range
// delivers zeros (the padding voxels of a tap), so the stage body needs no branch and no zero page
__device__ void llvm_amdgcn_raw_buffer_load_lds(i32x4 rsrc, lds_u32_ptr lds, int size, int voffset, int soffset, int offset,
                                                int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ i32x4 make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  i32x4 v = __builtin_bit_cast(i32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]);
  v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]);
  v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

// NA = 16-row blocks of the gathered operand this wave fetches per stage (12 blocks over 8 waves: 2 for waves 0-3, 1 for 4-7)
template <typename H, int BN, int NA>
__device__ __forceinline__ void ring_main(const ConvKArgs& P, const ctseg_conv_class& K, char* smem, const int* sRow, const int* sTap,
                                          f32x4 (&acc)[4][RingCfgT<BN>::BM / RingCfgT<BN>::WGM / 16], int n, int col0, int wave,
                                          int lane) {
  using C = RingCfgT<BN>;
  constexpr int BM = C::BM, D = C::D, RB = C::RB, STAGE = C::STAGE, BOFF = C::BOFF, NW = C::NW, WGN = C::WGN;
  constexpr int WTM = BM / C::WGM, WTN = BN / WGN, MT = WTM / 16, NT = WTN / 16, NB = BN / 16 / NW;
  static_assert(NT == 4 && NB >= 1, "wave tile is 64 columns wide");
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, q4 = lane >> 4;
  const int ntaps = K.ntaps, kpad = K.kpad;

  // ---- loader role: row lane/4 of a 16-row block, 16-byte slot lane%4 ------------------------------------------
  const int lrow = lane >> 2;
  const int kc = (lane & 3) ^ ((lane & 32) ? 3 : 0);     // K chunk (8 bf16) this lane fetches
  const uint32_t in_bytes = (uint32_t)((int64_t)P.N * P.Xi * P.Yi * P.Zi * P.g_ld * 2);
  const i32x4 rsA = make_rsrc(P.in, in_bytes);
  const i32x4 rsB = make_rsrc(P.w + (K.w_off + (int64_t)col0 * kpad) * 2, 0x7fffffffu);
  const int tapd = sTap[32 + (lane & 31)];   // lane t keeps tap t's byte delta: a stage's tap is one v_readlane
  int rowoff[NA];
  uint32_t vmask[NA];          // bit t: tap t of this row reads inside the tensor
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int r = (wave + j * NW) * 16 + lrow;
    const int xy = sRow[2 * r], z = sRow[2 * r + 1];
    const int xb = (xy & 0xffff) * P.sin, yb = (xy >> 16) * P.sin, zb = (z < 0) ? z : z * P.sin;
    rowoff[j] = (int)(((((int64_t)n * P.Xi + xb) * P.Yi + yb) * P.Zi + (zb < 0 ? 0 : zb)) * P.g_ld * 2) + kc * 16;
    uint32_t m = 0;
    for (int t = 0; t < ntaps; ++t) {
      const int tp = sTap[t];
      const int xi = xb + (int)(int8_t)(tp & 0xff), yi = yb + (int)(int8_t)((tp >> 8) & 0xff), zi = zb + (int)(int8_t)((tp >> 16) & 0xff);
      const bool ok = zb >= 0 && (unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi;
      m |= (ok ? 1u : 0u) << t;
    }
    vmask[j] = m;
  }
  int woff[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) woff[j] = (((wave + j * NW) * 16 + lrow) * kpad + kc * 8) * 2;

  int dslot = 0, dci = 0;    // tap and channel position of the next stage to be fetched (wave-uniform)
  auto dma = [&](int sl) {
    const int tb = __builtin_amdgcn_readlane(tapd, dslot & 31) + dci * 2;
    char* dst = smem + sl * STAGE + wave * 16 * RB;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool ok = ((vmask[j] >> (dslot & 31)) & 1u) != 0;
      const int vo = ok ? rowoff[j] + tb : (int)0x80000000;
      llvm_amdgcn_raw_buffer_load_lds(rsA, (lds_u32_ptr)(dst + j * NW * 16 * RB), 16, vo, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      llvm_amdgcn_raw_buffer_load_lds(rsB, (lds_u32_ptr)(dst + BOFF + j * NW * 16 * RB), 16, woff[j], 0, 0, 0);
      woff[j] += C::KS * 2;
    }
    dci += C::KS;
    const bool wrap = dci >= P.Cg;       // Cg is a multiple of 32: a stage never straddles two taps
    dci = wrap ? 0 : dci;
    dslot += wrap ? 1 : 0;               // taps past ntaps (K padding) have no mask bit: zeros
  };

  u32x4 xf[2][MT], wf[2][NT];
  const int foff = ((q4 ^ ((r16 & 8) ? 3 : 0)) << 4);
  const int faoff = (wm * WTM + r16) * RB + foff, fboff = BOFF + (wn * WTN + r16) * RB + foff;
  auto frags = [&](int sl, u32x4 (&x)[MT], u32x4 (&w)[NT]) {
    const char* base = smem + sl * STAGE;
#pragma unroll
    for (int i = 0; i < MT; ++i) x[i] = *reinterpret_cast<const u32x4*>(base + faoff + i * 16 * RB);
#pragma unroll
    for (int j = 0; j < NT; ++j) w[j] = *reinterpret_cast<const u32x4*>(base + fboff + j * 16 * RB);
  };
  auto mmas = [&](const u32x4 (&x)[MT], const u32x4 (&w)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) mma16<H>(acc[j][i], w[j], x[i]);
  };
  constexpr int CNT = NA + NB;     // this wave's loads per stage

  const int nst = kpad / C::KS;   // even and >= 8 (host-checked)
  // Schedule.  Step s multiplies stage s (fragments read during step s-1), reads the fragments of stage s+1 and requests
  // stage s+D-1 into the slot stage s-1 occupied: every wave's multiplies of stage s-1 -- which waited for its own
  // fragment reads -- come before the barrier of step s, so that slot is free without any manual LDS wait.
#pragma unroll
  for (int t = 0; t < D - 1; ++t) dma(t);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * CNT) : "memory");   // stage 0 landed, stages 1..D-2 in flight
  __builtin_amdgcn_s_barrier();
  frags(0, xf[0], wf[0]);

  auto step = [&](auto dma_c, auto fr_c, int sl, u32x4 (&x0)[MT], u32x4 (&w0)[NT], u32x4 (&x1)[MT], u32x4 (&w1)[NT]) {
    constexpr bool DMA = decltype(dma_c)::value, FR = decltype(fr_c)::value;
    // this wave's share of stage s+1 has landed (stages s+2 .. s+D-2 may still be in flight)
    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 3) * CNT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // ... and everyone else's
    __builtin_amdgcn_sched_barrier(0);
    const int sl1 = sl + 1 == D ? 0 : sl + 1, slp = sl == 0 ? D - 1 : sl - 1;
    // (Round 4, same box, 256 -> 256: static s_setprio 1 for waves 4-7 0.1442 -> 0.1450 ms; waves 4-7 reading their fragments BEHIND
    // their multiplies instead of ahead of them 0.1465; both 0.1507 — the two waves of a SIMD are not helped by being told apart.)
    if constexpr (FR) frags(sl1, x1, w1);
    if constexpr (DMA) dma(slp);
    mmas(x0, w0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  int sl = 0;
  auto next = [&]() { sl = (sl + 1 == D) ? 0 : sl + 1; };
  for (int s = 0; s < nst - D + 1; s += 2) {      // nst - D + 1 is even
    step(T_{}, T_{}, sl, xf[0], wf[0], xf[1], wf[1]); next();
    step(T_{}, T_{}, sl, xf[1], wf[1], xf[0], wf[0]); next();
  }
  step(F_{}, T_{}, sl, xf[0], wf[0], xf[1], wf[1]); next();
  step(F_{}, T_{}, sl, xf[1], wf[1], xf[0], wf[0]); next();
  step(F_{}, T_{}, sl, xf[0], wf[0], xf[1], wf[1]); next();
  step(F_{}, F_{}, sl, xf[1], wf[1], xf[0], wf[0]);
  static_assert(D == 5, "the drain above is written for a five-deep ring");
}

template <typename H, int BN, bool BST = false>     // H = 16-bit storage kind (BF16 / F16): the only difference is the MFMA and the epilogue's conversions
__global__ __launch_bounds__(512) void conv_igemm_ring_kernel(const ConvKArgs P) {
  using C = RingCfgT<BN>;
  constexpr int BM = C::BM, MT = BM / C::WGM / 16;
  static_assert(C::TOTAL <= 160 * 1024, "LDS");

  __shared__ __attribute__((aligned(16))) char smem[C::TOTAL];
  float* const sStats = reinterpret_cast<float*>(smem + C::MAIN);
  int* const sRow = reinterpret_cast<int*>(smem + C::MAIN + C::STATS);   // [BM][2]
  int* const sTap = sRow + BM * 2;                                         // [32] packed offsets, [32] byte deltas

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx = blockIdx.x;
  if (P.xcd_order) {
    bx = xcd_tile(bx, P.tiles * P.N);
    if (bx < 0) return;
  }
  const int tile = bx % P.tiles, n = bx / P.tiles;
  const int col0 = blockIdx.y * BN;
  const ctseg_conv_class& K = P.cls[blockIdx.z];
  const int ntaps = K.ntaps;

  for (int r = tid; r < BM; r += C::NTHR) {
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
    sTap[32 + tid] = (((int)(int8_t)(tp & 0xff) * P.Yi + (int)(int8_t)((tp >> 8) & 0xff)) * P.Zi + (int)(int8_t)((tp >> 16) & 0xff)) *
                     P.g_ld * 2;
  }
  __syncthreads();

  f32x4 acc[4][MT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the two halves of the workgroup run the same schedule with a different (compile-time) number of loads per stage
  if (wave < 4) ring_main<H, BN, 2>(P, K, smem, sRow, sTap, acc, n, col0, wave, lane);
  else ring_main<H, BN, 1>(P, K, smem, sRow, sTap, acc, n, col0, wave, lane);
  __syncthreads();

  conv_epilogue<H, BM, BN, C::WGM, C::WGN, BST>(P, K, smem, sStats, sRow, acc, n, tile, (int)blockIdx.z, col0);
}

bool conv_ring_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (!is16(dtype) || a.out_f32 || a.Cn <= 64 || a.Cg % 32 != 0) return false;
  if ((int64_t)a.N * a.Xi * a.Yi * a.Zi * a.g_ld * 2 >= (int64_t)1 << 31) return false;   // 32-bit buffer offsets
  for (int c = 0; c < nclass; ++c)
    if (a.cls[c].kpad % 64 != 0 || a.cls[c].kpad < 256 || a.cls[c].ntaps > 32 || a.cls[c].kpad < a.cls[c].ntaps * a.Cg ||
        (int64_t)256 * a.cls[c].kpad * 2 >= (int64_t)1 << 31)
      return false;
  return true;
}

void launch_conv_ring(const ConvKArgs& a, int nclass, hipStream_t st) {
  const int gx = a.xcd_order ? 8 * ((a.tiles * a.N + 7) / 8) : a.tiles * a.N;
  if (a.Cn > 128) {
    dim3 grid((unsigned)gx, (unsigned)((a.Cn + 255) / 256), (unsigned)nclass);
    if (a.dtype == CTSEG_F16) hipLaunchKernelGGL((conv_igemm_ring_kernel<F16, 256>), grid, dim3(512), 0, st, a);
    else if (a.bst.part != nullptr) hipLaunchKernelGGL((conv_igemm_ring_kernel<BF16, 256, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_ring_kernel<BF16, 256>), grid, dim3(512), 0, st, a);
  } else {
    dim3 grid((unsigned)gx, 1u, (unsigned)nclass);
    if (a.dtype == CTSEG_F16) hipLaunchKernelGGL((conv_igemm_ring_kernel<F16, 128>), grid, dim3(512), 0, st, a);
    else if (a.bst.part != nullptr) hipLaunchKernelGGL((conv_igemm_ring_kernel<BF16, 128, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_ring_kernel<BF16, 128>), grid, dim3(512), 0, st, a);
  }
}

}  // namespace ctseg
