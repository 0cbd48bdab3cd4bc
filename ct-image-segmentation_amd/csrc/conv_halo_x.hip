// x-column LDS-halo kernel (gfx950): the 3x3x3 stride-1 passes with few channels in 16-bit storage — 32->32 at
// 256x256x24, the 10->10 logits convolution at 512x512x48 and their input gradients (reference: the MONAI UNet built at
// capstone/volumetric/base_trainer.py:65-72, levels 0 / top).
//
// Same LDS halo image, MFMA operand layout and epilogue contract as conv_halo.hip, whose phase ablation
// (profiles/r02_halo_kernel_phase_ablation.txt) showed the time going to ~1700 bookkeeping instructions per tile and wave
// around 56 MFMAs, with no overlap between staging, multiplies and stores.  What is different here:
//
//  * staging is LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction fills 64 consecutive 16-byte slots of one channel-chunk
//    plane; the per-lane source offset is a per-thread constant and a halo voxel outside the volume is ONE v_and + v_cmp +
//    v_cndmask (one-hot coordinate bits of the lane against the tile's valid-coordinate mask) selecting an offset past the
//    buffer range, for which the hardware delivers zeros.  No VGPR round trip, no ds_write pass, no branches;
//  * a wave owns an x-COLUMN of the tile: the four 2x8 (y,z) patches at x0..x0+3 of one y pair.  The operand fragment of
//    patch x for tap (dx,dy,dz) IS the fragment of patch x+dx for tap (0,dy,dz), so per (dy,dz) the wave reads the six
//    fragments x0-1..x0+4 once and feeds twelve MFMAs from them: 0.5 ds_read_b128 per MFMA instead of 1.0-1.25;
//  * the packed weights of the wave's 16 output columns stay in registers for the whole launch (27 or 15 fragments): nothing
//    but the halo lives in LDS, so the 32-byte-voxel variant still fits three workgroups per CU;
//  * every LDS read offset is an instruction immediate (the tap order is one of two canonical ones, checked on the host);
//    stores are buffer stores with a per-thread constant offset, a scalar tile base and out-of-range offsets for the voxels a
//    ragged tile does not own.
#include "conv_common.h"

#ifndef X_ABL
#define X_ABL 0      // timing-only ablation builds (tools/ablate_halo_x.sh): 1 = no MFMAs, 2 = no stores, 4 = no DMA, 128 = no 8-byte staging stores (12-wide rows), 256 = no tile barrier, 1024 = no operand reads, 2048 = no logits exchange (fused head; tools/ablate_head_ce_r4.sh -> profiles/r04_fused_head_ablation.txt); results are garbage
#endif

namespace ctseg {

constexpr int X_TX = 4, X_TY = 8, X_TZ = 8;
constexpr int X_NRM_MAXN = 16;      // samples whose operand-normalisation constants fit the LDS table
constexpr int X_HX = X_TX + 2, X_HY = X_TY + 2, X_HZ = X_TZ + 2, X_HV = X_HX * X_HY * X_HZ;   // 600 halo voxels
constexpr int X_PIECES = (X_HV + 63) / 64;                                                     // 10 DMA pieces per plane
// plane stride in 16-byte slots: >= 64 * X_PIECES (a DMA piece writes 64 slots) and = 8 (mod 16), so that the two planes an
// MFMA operand read combines in one 16-lane LDS group fall on disjoint halves of the 256-byte bank row (conv_halo.hip header)
constexpr int X_HVP = 648;
constexpr int X_PLANE = X_HVP * 16;
constexpr int X_XSTRIDE = X_HY * X_HZ * 16;   // bytes between x planes of the halo

typedef int32_t xi32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3)))* x_lds_u32_ptr;
__device__ void x_raw_buffer_load_lds(xi32x4 rsrc, x_lds_u32_ptr lds, int size, int voffset, int soffset, int offset,
                                      int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

__device__ __forceinline__ xi32x4 x_make_rsrc(const void* p, uint32_t bytes) {
  struct __attribute__((packed)) { const void* ptr; uint32_t range; uint32_t config; } r{p, bytes, 0x00020000u};
  xi32x4 v = __builtin_bit_cast(xi32x4, r);
  v[0] = __builtin_amdgcn_readfirstlane(v[0]);
  v[1] = __builtin_amdgcn_readfirstlane(v[1]);
  v[2] = __builtin_amdgcn_readfirstlane(v[2]);
  v[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return v;
}

// r16 -> (dy, z) inside a 2x8 patch (same permutation as conv_halo.hip: every ds_read_b128 lane group is conflict free)
__device__ __forceinline__ void x_patch_voxel(int r16, int& dy, int& z) {
  dy = (0xEF80u >> r16) & 1;
  z = (int)((0x2104765437653210ull >> (4 * r16)) & 7ull);
}

template <int VB> struct XCfg {
  static constexpr int NPL = VB / 16;                  // 16-byte channel chunks per voxel = LDS planes
  static constexpr int HALO = NPL * X_PLANE;           // one halo buffer
  static constexpr int TYPES = VB == 64 ? 9 : 5;       // K steps per dx: one (dy,dz) tap of 32 channels, or a pair of taps of 16
  static constexpr int NPIECE = NPL * X_PIECES;        // DMA pieces per tile
};

struct XGeom {
  int tiles, tyn, tzn;        // tiles per sample, tiles along y / z
  int in_sample_bytes;        // < 2^31 (host-checked)
  int out_sample_bytes;
  int add_sample_bytes;
  int y_sample_bytes;         // BST: the norm's forward tensor (ConvKArgs::bst.y)
};

// Cross-entropy fused into the epilogue of the logits convolution (CE = true): the fp32 logits never leave the registers.  What
// ctseg_seg_loss's cross-entropy-only path computes per voxel (capstone/models/losses.py:45-68 F.cross_entropy [weighted],
// training/utils.py:19-20 softmax -> argmax, models/metrics.py:15-21 Dice counts) is computed here with the same arithmetic.
struct XCe {
  const uint8_t* labels;            // [N][S]
  const float* class_weight;        // [C] or NULL
  const float* coef;                // per sample: coef[n * coef_stride] = d(loss)/d(weighted NLL sum) (1 / denominator)
  int coef_stride;
  char* dlogits;                    // [N][S][g_ld] 16-bit storage of the pass
  int g_ld;
  double* part;                     // [N][P][R]: entries 0 (weighted NLL sum) and 1 (weight sum) of slot blockIdx.x; the rest zeroed
  int P, R;
  unsigned long long* cnt;          // [N][3][C] += (|pred==c & true==c|, |pred==c|, |true==c|)
  int C;
};

// ADD: 0 none, 1 the addend is the INPUT tensor (identity residual: taken from the centre voxel of the LDS halo), 2 a tensor in
// global memory (bf16/half like the input, or fp32 when P.add_f32) loaded by the epilogue, 3 a 16-bit tensor staged by LDS-DMA
// beside the halo of its tile (ordinary loads inside the DMA pipeline make the compiler drain it: 0.27 vs 0.17 ms on 32 -> 32).
// A wave owns the x-column of one y pair for NT 16-column blocks of the output; NS such wave groups (4 waves each) split the
// column blocks of a wider output between them.  Measured on 32 -> 32 at 2 x 256 x 256 x 24 (ms per launch, register-staged kernel
// 0.22): NT = 1, NS = 2 with the weights in registers 0.165 (220-246 VGPRs: the operand reads sit right in front of their MFMAs);
// NT = 2, NS = 1 (one wave per SIMD) with the weights in registers 0.257 (216 weight registers leave ONE operand quad: MFMA A/B
// operands come from the 256 architectural VGPRs) and with the weights in LDS 0.22 (a lone wave per SIMD exposes every LDS wait the
// scheduler leaves); NT = 1, NS = 2 with the weights in LDS is the configuration launched.
// R12: the gathered rows are 12 elements wide (24 bytes: the 10-class tensors of the head): such a row cannot be moved in 16-byte DMA
// pieces, so the halo is staged through registers in 8-byte pieces (buffer loads with the same out-of-range trick, then
// ds_write_b64 into the same two-plane LDS image; the upper half of every plane-1 slot is zeroed once and never written again).
// BST: backward InstanceNorm statistics of the written gradient (ConvKArgs::bst).  Plane i of a tile's epilogue reads the y values of
// its four channels from registers that were loaded ONE TILE AHEAD (right behind the previous tile's store of the same plane: a
// load consumed in the tile it is issued in puts an HBM round trip on every tile), and accumulates the three sums per lane; they
// are combined per workgroup at every sample change, as the forward statistics are.
template <typename H, int VB, int NT, int NS, bool FLIP, bool STATS, int ADD, bool OF32, bool R12, bool CE = false, bool BST = false>
// (Round 4: a third workgroup per CU for the 32-byte-voxel DMA-staged variants — __launch_bounds__(256, 3): 168 registers, 12-64 B of
// scratch — bought nothing: logits conv on 16-wide rows 0.592 ms at 2 / CU, 0.616 at 3 / CU with the 2-per-CU build, 0.649 / 0.638 with
// the 168-register build.  Occupancy is not what these passes wait for.)
__global__ __launch_bounds__(256 * NS, (BST && VB == 32 && NS == 1) ? 2 : 1) void conv_halo_x_kernel(const ConvKArgs P, const XGeom G, int total_tiles, const XCe E) {
  static_assert(!BST || (!STATS && !OF32 && !CE && ADD != 2), "backward statistics: 16-bit gradient passes without a global-memory addend");
  static_assert(!R12 || (VB == 32 && NS == 1), "12-wide rows: 16 gathered channels, one wave group");
  static_assert(!CE || (VB == 32 && NS == 1 && NT == 1 && !FLIP && !STATS && ADD < 2), "fused cross-entropy: the logits convolution");
  static_assert(ADD != 3 || (!R12 && !CE), "DMA-staged addend: the DMA pipeline only");
  using CF = XCfg<VB>;
  // WL: the packed weights live in LDS in FRAGMENT order (1 KB per (column block, dx, K-step group): lane l reads its 16 bytes at
  // l * 16, conflict free by construction) instead of in registers: 54 fragments = 216 registers per lane do not fit beside the
  // operand prefetch (MFMA A/B operands come from the 256 architectural VGPRs)
  // Weights in registers (see above: the same speed, and the LDS goes to the addend staging of ADD == 3) — except with BST at
  // 64-byte voxels and two wave groups: that variant needs 220-246 registers without the three sums, their constants and the y
  // values, so its 54 weight fragments live in LDS.  A1: its DMA-staged addend then has ONE buffer, laid out per wave (a wave
  // stages exactly the 64 voxels x 2 chunks it reads itself), read into registers before the wave issues the next tile's pieces.
  constexpr bool WL = BST && VB == 64 && NS == 2;
  constexpr bool A1 = WL && ADD == 3;
  constexpr int NAB = A1 ? 1 : 2;
  constexpr int NTA = NT * NS;                           // column blocks of the workgroup
  constexpr int WBYTES = WL ? NTA * 3 * CF::TYPES * 1024 : 0;
  constexpr int NPL = CF::NPL, TYPES = CF::TYPES, NW = 4 * NS, NTHR = 64 * NW;
  constexpr int NWD = CF::NPIECE % NW == 0 ? NW : 4;    // waves that issue DMA pieces (20 pieces over 8 waves: the first four)
  constexpr int PPW = CF::NPIECE / NWD;                 // DMA pieces per such wave
  static_assert(CF::NPIECE % NWD == 0, "every DMA wave issues the same number of pieces (counted vmcnt)");
  constexpr int OSZ = OF32 ? 4 : 2;

  // ADD == 3: the addend tile (X_TX x 8 x 8 voxels, the workgroup's 16 * NTA channels) in 16-byte chunks, voxel-major, two buffers
  constexpr int ACH = 2 * NTA;                               // chunks per voxel
  constexpr int ABUF = ADD == 3 ? X_TX * 64 * ACH * 16 : 0;
  constexpr int APIECES = X_TX * 64 * ACH / 64, APW = ADD == 3 ? APIECES / NW : 0;
  static_assert(ADD != 3 || APIECES % NW == 0, "every wave issues the same number of addend pieces");
  constexpr int CE_T = CE ? NW * 2 * 16 * 48 : 0;           // per-wave exchange scratch: 12 fp32 logits per voxel, two x planes at a time
  constexpr int CE_B = CE ? 3 * 16 * 4 + NW * 2 * 8 : 0;     // Dice counters, per-wave loss sums
  constexpr int NRM_B = R12 ? X_NRM_MAXN * 96 : 0;           // operand normalisation: per sample 3 channel quads x (mean x 4, rstd x 4)
  constexpr int CE_L = CE ? NW * 3 * 6 * 64 * 4 : 0;         // per-LANE Dice counters: [wave][kind][class pair][lane], two 16-bit fields per word
  constexpr int ST_B = NW * (BST ? 3 : 2) * 16 * NT * 4;     // per-wave statistics slots (forward: sum, sum of squares; BST: three sums)
  constexpr int BT_B = BST ? 16 * NTA * 2 * 4 : 0;           // BST: (mean x 4, rstd x 4) per channel quad of the current sample
  constexpr int CAD_B = (BST && R12 && ADD == 1 && !CE) ? 2 * NT * X_TX * 256 * NS * 8 : 0;   // lane-private centre voxels of two tiles
  __shared__ __attribute__((aligned(16))) char smem[2 * CF::HALO + WBYTES + ST_B + CE_T + CE_B + NAB * ABUF + NRM_B + CE_L + BT_B + CAD_B];
  char* const sCad = smem + 2 * CF::HALO + WBYTES + ST_B + CE_T + CE_B + NAB * ABUF + NRM_B + CE_L + BT_B;
  float* const sBt = reinterpret_cast<float*>(smem + 2 * CF::HALO + WBYTES + ST_B + CE_T + CE_B + NAB * ABUF + NRM_B + CE_L);
  char* const sLaneBase = smem + 2 * CF::HALO + WBYTES + ST_B + CE_T + CE_B + NAB * ABUF + NRM_B;
  char* const sA = smem + 2 * CF::HALO + WBYTES + ST_B + CE_T + CE_B;
  float* const sPar = reinterpret_cast<float*>(smem + 2 * CF::HALO + WBYTES + ST_B + CE_T + CE_B + NAB * ABUF);
  char* const sW = smem + 2 * CF::HALO;
  float* const sStats = reinterpret_cast<float*>(smem + 2 * CF::HALO + WBYTES);     // per-wave statistics slots
  char* const sT = smem + 2 * CF::HALO + WBYTES + ST_B;
  unsigned int* const sCnt = reinterpret_cast<unsigned int*>(sT + CE_T);
  double* const sSum = reinterpret_cast<double*>(sT + CE_T + 3 * 16 * 4);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned int* const sLane = reinterpret_cast<unsigned int*>(sLaneBase) + wave * (3 * 6 * 64) + lane;     // (CE only)
  const int yp = wave & 3, ns = wave >> 2;               // y pair of the column, wave group
  const int r16 = lane & 15, q4 = lane >> 4;
  const ctseg_conv_class& K = P.cls[0];
  const int col0 = blockIdx.y * (16 * NTA) + ns * 16 * NT;   // first output column of this wave
  int pdy, pz;
  x_patch_voxel(r16, pdy, pz);

  // ---- weights -> registers, canonical order W[column block][dx + 1][type] whatever the pass (FLIP: the input-gradient pass
  //      lists its taps with every offset negated, so its group of dx = -1 is the LAST nine taps and type t holds -(canonical)) ----
  u32x4 W[WL ? 1 : NT][WL ? 1 : 3][WL ? 1 : TYPES];
#pragma unroll
  for (int j = 0; j < (WL ? NTA : NT); ++j) {
    const int cb = WL ? blockIdx.y * (16 * NTA) + j * 16 : col0 + j * 16;
    const char* wrow = P.w + (K.w_off + (int64_t)(cb + r16) * K.kpad) * 2;
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
      for (int t = 0; t < TYPES; ++t) {
        const int fi = (j * 3 + dxi) * TYPES + t;       // fragment index
        if (WL && (fi % NW) != wave) continue;          // each wave stages its share of the fragments
        const int g = FLIP ? 2 - dxi : dxi;
        u32x4 v = {0u, 0u, 0u, 0u};
        if constexpr (VB == 64) {
          const int tt = FLIP ? 8 - t : t;              // canonical (dy,dz) index t -> position inside the group
          v = *reinterpret_cast<const u32x4*>(wrow + ((9 * g + tt) * 32 + q4 * 8) * 2);
        } else {
          const int tc = 2 * t + (q4 >> 1);             // canonical (dy,dz) index of this lane's half of the K step
          if (tc < 9) {
            const int tt = FLIP ? 8 - tc : tc;
            v = *reinterpret_cast<const u32x4*>(wrow + ((9 * g + tt) * 16 + (q4 & 1) * 8) * 2);
          }
        }
        if constexpr (WL) *reinterpret_cast<u32x4*>(sW + fi * 1024 + lane * 16) = v;
        else W[j][dxi][t] = v;
      }
  }

  // ---- per-lane constants -------------------------------------------------------------------------------------------------
  // LDS read address of the fragment (plane h = 0, canonical type 0 = (dy,dz) = (-1,-1)): the rest are immediates
  int rbase[VB == 64 ? 1 : TYPES];
  {
    const int vox = ((2 * yp + pdy + 1) * X_HZ + (pz + 1)) * 16;
    if constexpr (VB == 64) {
      rbase[0] = q4 * X_PLANE + vox;
    } else {
#pragma unroll
      for (int t = 0; t < TYPES; ++t) {
        const int tc = 2 * t + (q4 >> 1);
        const int tcc = tc < 9 ? tc : 4;                // the empty half of the last K step reads the centre tap (weights are 0)
        rbase[t] = (q4 & 1) * X_PLANE + vox + ((tcc / 3 - 1) * X_HZ + (tcc % 3 - 1)) * 16;
      }
    }
  }
  // DMA pieces of this wave: source byte offset from the halo origin voxel and one-hot coordinate bits
  int poff[PPW];
  uint32_t phot[PPW];
  const int YZ = P.Yi * P.Zi;
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int piece = (wave % NWD) + j * NWD;
    const int pl = piece / X_PIECES, hv = (piece % X_PIECES) * 64 + lane;
    const int hx = hv / (X_HY * X_HZ), rem = hv - hx * (X_HY * X_HZ);
    const int hy = rem / X_HZ, hz = rem - hy * X_HZ;
    poff[j] = ((hx * YZ + hy * P.Zi + hz) * P.g_ld + pl * 8) * 2;
    phot[j] = hv < X_HV ? ((1u << hx) | (1u << (6 + hy)) | (1u << (16 + hz))) : 0x80000000u;
  }
  int apoff[APW > 0 ? APW : 1];
  uint32_t aphot[APW > 0 ? APW : 1];
  if constexpr (ADD == 3) {
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      if constexpr (A1) {
        // piece j of this wave: lane (x plane l >> 4, patch voxel l & 15 of the wave's y pair), chunk 2 ns + j — what the wave's own
        // lanes read back (lane (r16, q4): plane i, chunk 2 ns + (q4 >> 1) at lane slot i * 16 + r16 of piece q4 >> 1)
        static_assert(!A1 || APW == 2, "two chunks per wave group");
        int ady, az;
        x_patch_voxel(lane & 15, ady, az);
        const int ix = lane >> 4, iy = 2 * yp + ady, iz = az;
        apoff[j] = ((ix * P.Yo + iy) * P.Zo + iz) * P.add_ld * 2 + (blockIdx.y * ACH + 2 * ns + j) * 16;
        aphot[j] = (1u << ix) | (1u << (4 + iy)) | (1u << (12 + iz));
        continue;
      }
      const int ci = (wave + j * NW) * 64 + lane, tv = ci / ACH, ch = ci - tv * ACH;
      const int ix = tv >> 6, iy = (tv >> 3) & 7, iz = tv & 7;
      // 64-byte voxels (ACH = 4): the 16-byte chunk at LDS position ch holds channel chunk ch ^ ((voxel >> 2) & 3) — the 16 voxels x
      // 2 halves a 32-lane group reads (8 bytes each, 64 bytes apart) then cover all 64 banks instead of 8 slots four times
      const int lch = ACH == 4 ? (ch ^ ((tv >> 2) & 3)) : ch;
      apoff[j] = ((ix * P.Yo + iy) * P.Zo + iz) * P.add_ld * 2 + (blockIdx.y * ACH + lch) * 16;
      aphot[j] = (1u << ix) | (1u << (4 + iy)) | (1u << (12 + iz));
    }
  }
  // R12 staging: 3 eight-byte pieces per halo voxel (bytes 0-7, 8-15 -> plane 0; 16-23 -> plane 1), voxel-major so that a wave's
  // loads run along the 240-byte z rows of the volume
  constexpr int R_N = X_HV * 3, R_J = R12 ? (R_N + NTHR - 1) / NTHR : 1;
  // the constants of the last round (8 pieces of 1800 at 256 threads) are recomputed where they are used instead of held in three
  // registers by every lane through the multiplies (BST: the kernel sits at the 256-register line of two waves per SIMD)
  constexpr int R_JC = (R12 && BST && R_N % NTHR != 0) ? R_J - 1 : R_J;
  int roff[R_JC], rlds[R_JC];
  uint32_t rhot[R_JC];
  auto piece_consts = [&](int idx, int& off, uint32_t& hot, int& lds) {
    const int hv = idx / 3, part = idx - hv * 3;
    const int hx = hv / (X_HY * X_HZ), rem = hv - hx * (X_HY * X_HZ);
    const int hy = rem / X_HZ, hz = rem - hy * X_HZ;
    off = (hx * YZ + hy * P.Zi + hz) * P.g_ld * 2 + part * 8;
    hot = idx < R_N ? ((1u << hx) | (1u << (6 + hy)) | (1u << (16 + hz))) : 0x80000000u;
    lds = (part == 2 ? X_PLANE : 0) + hv * 16 + (part == 1 ? 8 : 0);
  };
  if constexpr (R12) {
#pragma unroll
    for (int j = 0; j < R_JC; ++j) piece_consts(tid + j * NTHR, roff[j], rhot[j], rlds[j]);
  }
  auto piece = [&](int j, int& off, uint32_t& hot, int& lds) {
    if (j < R_JC) { off = roff[j]; hot = rhot[j]; lds = rlds[j]; return; }
    int t = tid;
    asm volatile("" : "+v"(t));          // (opaque: keeps the recomputation from being hoisted out of the tile loop into registers)
    piece_consts(t + j * NTHR, off, hot, lds);
  };
  const int bias_bytes = (YZ + P.Zi + 1) * P.g_ld * 2;    // the halo origin of a tile at x0 = y0 = z0 = 0 lies this far before the sample
  // output: per-lane byte offset of (plane 0, this lane's voxel, its 4 channels of column block 0) from the tile's first voxel
  const int chn = col0 + 4 * q4;
  bool ch_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) ch_ok[j] = chn + 16 * j < P.Cn_store;
  const int ooff = ((2 * yp + pdy) * P.Zo + pz) * P.o_ld * OSZ + chn * OSZ;
  const int oplane = P.Yo * P.Zo * P.o_ld * OSZ;
  const int ASZ = P.add_f32 ? 4 : 2;
  const int aoff = ADD == 2 ? ((2 * yp + pdy) * P.Zo + pz) * P.add_ld * ASZ + chn * ASZ : 0;
  const int aplane = ADD == 2 ? P.Yo * P.Zo * P.add_ld * ASZ : 0;
  // centre voxel of the halo for the identity-residual addend: the lane's 4 channels inside their 16-byte chunk (column block j: + 2 planes)
  const int cbase = ADD == 1 ? ((chn * 2) >> 4) * X_PLANE + ((1 * X_HY + (2 * yp + pdy + 1)) * X_HZ + (pz + 1)) * 16 + ((chn * 2) & 15) : 0;
  float bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bias[j][e] = (!BST && P.bias != nullptr && chn + 16 * j + e < P.Cn) ? P.bias[chn + 16 * j + e] : 0.f;   // (BST: a gradient pass, no bias: host-checked)
      if (CE && chn + 16 * j + e >= E.C) bias[j][e] = -3.0e38f;       // fused cross-entropy: a column that is not a class never wins the maximum, e = 0
    }

  float wsum[NT][4], wsq[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { wsum[j][e] = 0.f; wsq[j][e] = 0.f; }
  int stat_n = -1;
  auto flush_stats = [&](int n) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = wsum[j][e], b = wsq[j][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          sStats[(wave * 2 + 0) * 16 * NT + j * 16 + 4 * q4 + e] = a;
          sStats[(wave * 2 + 1) * 16 * NT + j * 16 + 4 * q4 + e] = b;
        }
        wsum[j][e] = 0.f;
        wsq[j][e] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * 16 * NTA) {
      const int which = tid / (16 * NTA), c = tid % (16 * NTA), grp = c / (16 * NT), cc = c % (16 * NT);
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) a += sStats[((grp * 4 + w) * 2 + which) * 16 * NT + cc];
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      P.stats[(slot_t * 2 + which) * P.stats_ld + blockIdx.y * (16 * NTA) + c] = a;
    }
    __syncthreads();
  };

  // ---- BST: per-lane sums of the lane's 4 channels (x NT column blocks), constants of the current sample -----------------------
  // (the third sum feeds the gradient of the ONE PReLU slope: only its total over channels matters, so a lane keeps one accumulator
  // for its four channels and reports it under the first of them)
  float b1[NT][4], b2[NT][4], b3[NT];
  u32x2 yreg[NT][X_TX];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    b3[j] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { b1[j][e] = b2[j][e] = 0.f; }
#pragma unroll
    for (int i = 0; i < X_TX; ++i) yreg[j][i] = u32x2{0u, 0u};
  }
  const float bal = BST ? P.bst.alpha[0] : 1.f;
  int bst_n = -1;
  bool ych_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) ych_ok[j] = BST && chn + 16 * j >= P.bst.col0 && chn + 16 * j - P.bst.col0 < P.bst.C;
  // the constants of the current sample live in LDS (a lane's 4 channels: two broadcast 16-byte reads per slice), not in registers
  // the multiplies' operand prefetch needs; rstd = 0 marks a column that is not a channel of the norm
  auto bst_consts = [&](int n) {       // (called by every thread, between the barriers of flush_bst or before the first tile's)
    if (tid < 16 * NTA * 2) {
      const int col = tid >> 1, which = tid & 1;
      const int c = blockIdx.y * (16 * NTA) + col - P.bst.col0;
      const bool ok = c >= 0 && c < P.bst.C;
      // (rstd x 4, -mean * rstd x 4) per channel quad: xhat = fma(y, rstd, -mean * rstd)
      const float mean = ok ? P.bst.mr[((int64_t)n * P.bst.C + c) * 2] : 0.f, rstd = ok ? P.bst.mr[((int64_t)n * P.bst.C + c) * 2 + 1] : 0.f;
      sBt[(col >> 2) * 8 + which * 4 + (col & 3)] = which == 0 ? rstd : -mean * rstd;
    }
    __syncthreads();
  };
  auto flush_bst = [&](int n) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = b1[j][e], b = b2[j][e], c = e == 0 ? b3[j] : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); if (e == 0) c += __shfl_xor(c, o, 64); }
        if (r16 == 0) {
          sStats[(wave * 3 + 0) * 16 * NT + j * 16 + 4 * q4 + e] = a;
          sStats[(wave * 3 + 1) * 16 * NT + j * 16 + 4 * q4 + e] = b;
          sStats[(wave * 3 + 2) * 16 * NT + j * 16 + 4 * q4 + e] = c;
        }
        b1[j][e] = 0.f; b2[j][e] = 0.f;
        if (e == 0) b3[j] = 0.f;
      }
    __syncthreads();
    if (tid < 3 * 16 * NTA) {
      const int which = tid / (16 * NTA), c = tid % (16 * NTA), grp = c / (16 * NT), cc = c % (16 * NT);
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) a += sStats[((grp * 4 + w) * 3 + which) * 16 * NT + cc];
      const int ch = blockIdx.y * (16 * NTA) + c - P.bst.col0;
      if (ch >= 0 && ch < P.bst.C) P.bst.part[(((int64_t)n * P.bst.P + blockIdx.x) * 3 + which) * P.bst.ld + ch] = a;
    }
    __syncthreads();
  };

  struct Org { int n, x0, y0, z0; };
  auto tile_origin = [&](int t) -> Org {
    Org o;
    o.n = t / G.tiles;
    int r = t - o.n * G.tiles;
    const int tz = r % G.tzn; r /= G.tzn;
    const int ty = r % G.tyn; const int tx = r / G.tyn;
    o.x0 = tx * X_TX; o.y0 = ty * X_TY; o.z0 = tz * X_TZ;
    return o;
  };
  auto range_mask = [](int lo, int hi, int nbits) -> uint32_t {   // bits lo..hi (clamped to 0..nbits-1)
    lo = lo < 0 ? 0 : lo;
    hi = hi > nbits - 1 ? nbits - 1 : hi;
    return hi < lo ? 0u : ((2u << hi) - (1u << lo));
  };
  // live = false (BST only): the same vector-memory instructions with every offset out of range.  The compiler's counted waits for
  // the y registers are the MINIMUM over all paths of the operations issued since their load: with the staging of the next tile
  // skipped on the last-tile path, every slice of EVERY tile waited for the staging loads just issued (0.40 -> 0.64 ms, head pass)
  auto dma = [&](const Org& o, int buf, bool live = true) {
    // halo coordinate h of an axis is voxel x0 - 1 + h: inside the volume for h in [1 - x0, Xi - x0]
    const uint32_t m = range_mask(1 - o.x0, P.Xi - o.x0, X_HX) | (range_mask(1 - o.y0, P.Yi - o.y0, X_HY) << 6) |
                       (range_mask(1 - o.z0, P.Zi - o.z0, X_HZ) << 16);
    const uint32_t notm = ~m;
    const xi32x4 rs = x_make_rsrc(P.in + (int64_t)o.n * G.in_sample_bytes - bias_bytes, (uint32_t)(G.in_sample_bytes + bias_bytes));
    const int soff = ((o.x0 * P.Yi + o.y0) * P.Zi + o.z0) * P.g_ld * 2;    // halo origin = tile origin - (1,1,1) = this + (rsrc base shift)
    char* dst = smem + buf * CF::HALO;
    if (NWD < NW && wave >= NWD) return;
    if (X_ABL & 4) return;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + j * NWD;
      const int vo = (live && (phot[j] & notm) == 0u) ? poff[j] : (int)0x80000000;
      x_raw_buffer_load_lds(rs, (x_lds_u32_ptr)(dst + (piece / X_PIECES) * X_PLANE + (piece % X_PIECES) * 1024), 16, vo, soff, 0, 0);
    }
    if constexpr (ADD == 3) {      // the addend of the same tile, consumed (into registers) at the end of its multiplies
      const uint32_t am = range_mask(0, P.Xr - o.x0 - 1, X_TX) | (range_mask(0, P.Yr - o.y0 - 1, 8) << 4) | (range_mask(0, P.Zr - o.z0 - 1, 8) << 12);
      const uint32_t anot = ~am;
      const xi32x4 ars = x_make_rsrc(P.add + (int64_t)o.n * G.add_sample_bytes, (uint32_t)G.add_sample_bytes);
      const int asoff = ((o.x0 * P.Yo + o.y0) * P.Zo + o.z0) * P.add_ld * 2;
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const int vo = (live && (aphot[j] & anot) == 0u) ? apoff[j] : (int)0x80000000;
        if constexpr (A1) x_raw_buffer_load_lds(ars, (x_lds_u32_ptr)(sA + (wave * APW + j) * 1024), 16, vo, asoff, 0, 0);
        else x_raw_buffer_load_lds(ars, (x_lds_u32_ptr)(sA + buf * ABUF + (wave + j * NW) * 1024), 16, vo, asoff, 0, 0);
      }
    }
  };

  u32x2 rg[R_J];
  auto gload = [&](const Org& o, bool live = true) {
    const uint32_t m = range_mask(1 - o.x0, P.Xi - o.x0, X_HX) | (range_mask(1 - o.y0, P.Yi - o.y0, X_HY) << 6) |
                       (range_mask(1 - o.z0, P.Zi - o.z0, X_HZ) << 16);
    const uint32_t notm = ~m;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.in) + (int64_t)o.n * G.in_sample_bytes - bias_bytes, 0,
                                                                         G.in_sample_bytes + bias_bytes, 0x00020000);
    const int soff = ((o.x0 * P.Yi + o.y0) * P.Zi + o.z0) * P.g_ld * 2;
#pragma unroll
    for (int j = 0; j < R_J; ++j) {
      int ro, rl; uint32_t rh;
      piece(j, ro, rh, rl);
      const int vo = (live && (rh & notm) == 0u) ? ro : (int)0x80000000;
      rg[j] = (X_ABL & 4) ? u32x2{0u, 0u} : __builtin_amdgcn_raw_buffer_load_b64(rs, vo, soff, 0);
    }
  };
  // Operand normalisation on load (ctseg_conv_desc::in_mean_rstd): the staged 8-byte piece (4 channels of one voxel) becomes
  // prelu((x - mean) * rstd) — the arithmetic of instnorm_prelu_fwd_kernel, rounded to the storage type as that pass rounds its
  // output — before it is written to LDS; voxels outside the volume (zero padding of the ACTIVATION) and channels >= in_C stay 0.
  // Piece j of this thread holds channel quad (tid + j) % 3: three table addresses, rotated.
  const bool nrm = R12 && P.in_mr != nullptr;
  const float nrm_al = nrm ? P.in_alpha[0] : 1.f;
  int prot[3] = {0, 0, 0};
  if constexpr (R12) {
    if (nrm) {
      for (int i = tid; i < P.N * 24; i += NTHR) {
        const int n = i / 24, k = i - n * 24, c = (k >> 3) * 4 + (k & 3);
        sPar[i] = c < P.in_C ? P.in_mr[((int64_t)n * P.in_C + c) * 2 + ((k >> 2) & 1)] : 0.f;     // rstd = 0 keeps pad channels at 0
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) prot[k] = ((tid + k) % 3) * 8;
    }
  }
  auto sstore = [&](int buf, const Org& o) {
    char* dst = smem + buf * CF::HALO;
    if (!nrm) {
#pragma unroll
      for (int j = 0; j < R_J; ++j) {
        int ro, rl; uint32_t rh;
        piece(j, ro, rh, rl);
        if ((R_J * NTHR == R_N || tid + j * NTHR < R_N) && (!(X_ABL & 128) || rg[j][0] == 0x12345678u)) *reinterpret_cast<u32x2*>(dst + rl) = rg[j];
      }
      return;
    }
    const uint32_t notm = ~(range_mask(1 - o.x0, P.Xi - o.x0, X_HX) | (range_mask(1 - o.y0, P.Yi - o.y0, X_HY) << 6) |
                            (range_mask(1 - o.z0, P.Zi - o.z0, X_HZ) << 16));
    const float* par = sPar + o.n * 24;
#pragma unroll
    for (int j = 0; j < R_J; ++j) {
      const f32x4 mean = *reinterpret_cast<const f32x4*>(par + prot[j % 3]), rstd = *reinterpret_cast<const f32x4*>(par + prot[j % 3] + 4);
      float v[4] = {h2f<H>(rg[j][0] & 0xffffu), h2f<H>(rg[j][0] >> 16), h2f<H>(rg[j][1] & 0xffffu), h2f<H>(rg[j][1] >> 16)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = (v[e] - mean[e]) * rstd[e];
        v[e] = a > 0.f ? a : nrm_al * a;
      }
      int ro, rl; uint32_t rh;
      piece(j, ro, rh, rl);
      const bool inside = (rh & notm) == 0u;
      const u32x2 w = inside ? u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])} : u32x2{0u, 0u};
      if (R_J * NTHR == R_N || tid + j * NTHR < R_N) *reinterpret_cast<u32x2*>(dst + rl) = w;
    }
  };

  // ---- epilogue of one tile, in a head (per-tile scalars, addend loads) and X_TX slices (one x plane each) so that the slices
  //      can be issued between the K steps of the NEXT tile's multiplies ---------------------------------------------------------
  struct Ep {
    int nx, sbase;
    bool lane_ok;
    __amdgpu_buffer_rsrc_t ors;
    u32x4 gadd[NT][X_TX];
  };
  struct Yl { int nx, sbase; bool lane_ok; __amdgpu_buffer_rsrc_t rs; };
  auto y_head = [&](const Org& o, Yl& y) {           // the tile whose y values the slices fetch (one tile ahead of their use)
    y.nx = P.Xr - o.x0;
    y.lane_ok = (o.y0 + 2 * yp + pdy < P.Yr) && (o.z0 + pz < P.Zr);
    y.sbase = (int)((((int64_t)o.x0 * P.Yo + o.y0) * P.Zo + o.z0) * P.o_ld * 2);       // y is laid out like the written tensor (host-checked)
    y.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.bst.y) + (int64_t)o.n * G.y_sample_bytes, 0, G.y_sample_bytes, 0x00020000);
  };
  auto y_load = [&](const Yl& y, int i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int vo = (y.lane_ok && ych_ok[j] && i < y.nx) ? ooff + 32 * j : (int)0x80000000;
      yreg[j][i] = __builtin_amdgcn_raw_buffer_load_b64(y.rs, vo, y.sbase + i * oplane, 0);
    }
  };
  auto ep_head = [&](const Org& o, Ep& e) {
    if (STATS && o.n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = o.n;
    }
    if (BST && o.n != bst_n) {
      if (bst_n >= 0) flush_bst(bst_n);
      bst_n = o.n;
      bst_consts(o.n);
    }
    e.nx = P.Xr - o.x0;                                          // planes of the tile inside the row grid (may exceed X_TX)
    e.lane_ok = (o.y0 + 2 * yp + pdy < P.Yr) && (o.z0 + pz < P.Zr);
    const int64_t tvox = ((int64_t)o.x0 * P.Yo + o.y0) * P.Zo + o.z0;
    e.sbase = (int)(tvox * P.o_ld * OSZ);                        // < 2^31 (host-checked per sample)
    e.ors = __builtin_amdgcn_make_buffer_rsrc(P.out + (int64_t)o.n * G.out_sample_bytes, 0, G.out_sample_bytes, 0x00020000);
    if constexpr (ADD == 2) {
      const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(P.add) + (int64_t)o.n * G.add_sample_bytes, 0,
                                                                            G.add_sample_bytes, 0x00020000);
      const int abase = (int)(tvox * P.add_ld * ASZ);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < X_TX; ++i) {
          const int vo = (e.lane_ok && ch_ok[j] && i < e.nx) ? aoff + 16 * j * ASZ : (int)0x80000000;
          if (P.add_f32) e.gadd[j][i] = __builtin_amdgcn_raw_buffer_load_b128(ars, vo, abase + i * aplane, 0);
          else {
            const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(ars, vo, abase + i * aplane, 0);
            e.gadd[j][i] = u32x4{w2[0], w2[1], 0u, 0u};
          }
        }
    }
  };
  Yl ynext;
  bool y_more = false;          // workgroup-uniform: the slices fetch the y values of the tile being multiplied
  // BST && R12 && ADD == 1 (the head's input gradient, at the 256-register line of two waves per SIMD): the identity-residual
  // addend of a tile (the centre voxels of its halo) does not ride through the next tile's multiplies in 2 x 8 registers but in a
  // lane-private LDS slot, written when the tile's multiplies end and read back by the slices.  (Reading the halo itself from the
  // slices RACED: a wave that finishes its multiplies early stores the tile after next over it while a slower wave's slices still
  // read — 2 % of the logits off at 2 x 512 x 512 x 48, nothing at the small test shapes.)
  constexpr bool CADD_LATE = BST && R12 && ADD == 1 && !CE;
  auto ep_slice = [&](const Ep& e, const f32x4 (&acc)[NT][X_TX], const u32x2 (&cadd_)[NT][X_TX], int i, int hbuf) {
    const bool ok = e.lane_ok && i < e.nx;
    u32x2 cadd[NT][X_TX];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (CADD_LATE) cadd[j][i] = *reinterpret_cast<const u32x2*>(sCad + (((hbuf * NT + j) * X_TX + i) * NTHR + tid) * 8);
      else cadd[j][i] = cadd_[j][i];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = acc[j][i][q] + bias[j][q];
        if (STATS) {
          const float sv = ok ? v[q] : 0.f;
          wsum[j][q] += sv;
          wsq[j][q] += sv * sv;
        }
      }
      if constexpr (ADD == 1 || ADD == 3) {
        v[0] += h2f<H>(cadd[j][i][0] & 0xffffu); v[1] += h2f<H>(cadd[j][i][0] >> 16);
        v[2] += h2f<H>(cadd[j][i][1] & 0xffffu); v[3] += h2f<H>(cadd[j][i][1] >> 16);
      } else if constexpr (ADD == 2) {
        if (P.add_f32) {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] += __uint_as_float(e.gadd[j][i][q]);
        } else {
          v[0] += h2f<H>(e.gadd[j][i][0] & 0xffffu); v[1] += h2f<H>(e.gadd[j][i][0] >> 16);
          v[2] += h2f<H>(e.gadd[j][i][1] & 0xffffu); v[3] += h2f<H>(e.gadd[j][i][1] >> 16);
        }
      }
      const int vo = (ok && ch_ok[j] && !(X_ABL & 2)) ? ooff + 16 * j * OSZ : (int)0x80000000;
      if constexpr (OF32) {
        const u32x4 o4 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(o4, e.ors, vo, e.sbase + i * oplane, 0);
        // >64-bit store data hazard: the compiler does not pad it for a buffer store with an SGPR offset (conv_halo_sw.hip's class
        // epilogue lost 16-bit elements to a v_mov scheduled right behind such a store); keep the data registers live for two wait states
        asm volatile("s_nop 1" ::"v"(o4));
      } else {
        const u32x2 o2 = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
        __builtin_amdgcn_raw_buffer_store_b64(o2, e.ors, vo, e.sbase + i * oplane, 0);
        if constexpr (BST) {
          // The sums are over the STORED gradient (rounded to the storage type), as the stand-alone reduce pass reads it.  This pass
          // is MFMA-bound (the matrix pipe is ~85 % busy at two waves per SIMD), so the arithmetic is packed two channels per
          // instruction: xhat = fma(y, rstd, -mean rstd); dxhat = g * (xhat > 0 ? 1 : slope); the sums of dxhat, dxhat * xhat and
          // g * min(xhat, 0).  Columns that are not channels of the norm have g = 0 (zero weight rows, no addend) and rstd = 0.
          if (ok && ych_ok[j]) {
            const float* bt = sBt + ((ns * NT + j) * 4 + q4) * 8;
            const f32x4 rs4 = *reinterpret_cast<const f32x4*>(bt), nm4 = *reinterpret_cast<const f32x4*>(bt + 4);
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
              const uint32_t gw = o2[hlf], yw = yreg[j][i][hlf];
              const f32x2 g2 = {__uint_as_float(gw << 16), __uint_as_float(gw & 0xffff0000u)};
              const f32x2 y2 = {__uint_as_float(yw << 16), __uint_as_float(yw & 0xffff0000u)};
              const f32x2 rs2 = {rs4[2 * hlf], rs4[2 * hlf + 1]}, nm2 = {nm4[2 * hlf], nm4[2 * hlf + 1]};
              const f32x2 xh = y2 * rs2 + nm2;
              const f32x2 sel = {xh[0] > 0.f ? 1.f : bal, xh[1] > 0.f ? 1.f : bal};
              const f32x2 dxh = g2 * sel;
              const f32x2 neg = {fminf(xh[0], 0.f), fminf(xh[1], 0.f)};
              f32x2 a1 = {b1[j][2 * hlf], b1[j][2 * hlf + 1]}, a2 = {b2[j][2 * hlf], b2[j][2 * hlf + 1]};
              a1 += dxh;
              a2 += dxh * xh;
              b3[j] = fmaf(g2[0], neg[0], fmaf(g2[1], neg[1], b3[j]));
              b1[j][2 * hlf] = a1[0]; b1[j][2 * hlf + 1] = a1[1];
              b2[j][2 * hlf] = a2[0]; b2[j][2 * hlf + 1] = a2[1];
            }
          }
        }
      }
    }
    if constexpr (BST) { if (y_more) y_load(ynext, i); }
  };

  // A1: the addend of the tile about to be multiplied, out of the wave's own part of the single addend buffer — complete before
  // the wave issues the pieces of the tile after it into the same slots
  auto cadd_early = [&](u32x2 (&cadd)[NT][X_TX]) {
    if constexpr (A1) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < X_TX; ++i)
          cadd[j][i] = *reinterpret_cast<const u32x2*>(sA + ((wave * 2 + (q4 >> 1)) * 64 + i * 16 + r16) * 16 + (q4 & 1) * 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  // multiplies of one tile from halo[buf] into `acc`; with PREV the epilogue slices of the previous tile (accumulators `pacc`)
  // are spread over the K steps: their conversions, statistics and stores issue under this tile's MFMAs
  auto compute_tile = [&](auto prev_c, int buf, f32x4 (&acc)[NT][X_TX], u32x2 (&cadd)[NT][X_TX], const Ep& pe,
                          const f32x4 (&pacc)[NT][X_TX], const u32x2 (&pcadd)[NT][X_TX]) {
    constexpr bool PREV = decltype(prev_c)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < X_TX; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* hb = smem + buf * CF::HALO;
    auto frag_base = [&](int t) -> const char* {
      if constexpr (VB == 64) return hb + rbase[0] + ((t / 3) * X_HZ + (t % 3)) * 16 - (X_HZ + 1) * 16;
      else return hb + rbase[t];
    };
    // the fragments of K-step group t+1 (operand + LDS-resident weights) are requested before the multiplies of group t: one
    // group of LDS latency is always covered by 12 * NT MFMAs
    u32x4 F[2][X_HX], Wb[2][WL ? NT : 1][WL ? 3 : 1];
    auto load_group = [&](int t, u32x4 (&f)[X_HX], u32x4 (&w)[WL ? NT : 1][WL ? 3 : 1]) {
#pragma unroll
      for (int h = 0; h < X_HX; ++h) {
        if constexpr ((X_ABL & 1024) != 0) f[h] = u32x4{(uint32_t)tid, (uint32_t)h, (uint32_t)t, 0u};     // (timing only: no operand reads)
        else f[h] = *reinterpret_cast<const u32x4*>(frag_base(t) + h * X_XSTRIDE);
      }
      if constexpr (WL) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int dxi = 0; dxi < 3; ++dxi)
            w[j][dxi] = *reinterpret_cast<const u32x4*>(sW + (((ns * NT + j) * 3 + dxi) * TYPES + t) * 1024 + lane * 16);
      }
    };
    load_group(0, F[0], Wb[0]);
#pragma unroll
    for (int t = 0; t < TYPES; ++t) {
      if (t + 1 < TYPES) load_group(t + 1, F[(t + 1) & 1], Wb[(t + 1) & 1]);
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int i = 0; i < X_TX; ++i) {
            if constexpr ((X_ABL & 1) != 0) {      // keep the operand reads alive without the matrix pipe
              acc[j][i][0] += __uint_as_float((WL ? Wb[t & 1][j][dxi][1] : W[WL ? 0 : j][WL ? 0 : dxi][WL ? 0 : t][1]) ^ F[t & 1][i + dxi][0]);
              continue;
            }
            if constexpr (WL) mma16<H>(acc[j][i], Wb[t & 1][j][dxi], F[t & 1][i + dxi]);
            else mma16<H>(acc[j][i], W[j][dxi][t], F[t & 1][i + dxi]);
          }
      if constexpr (PREV) {
        constexpr int every = TYPES >= 2 * X_TX ? 2 : 1;          // 9 K-step groups: slices after groups 0,2,4,6; 5 groups: after 0..3
        if (t % every == 0 && t / every < X_TX) ep_slice(pe, pacc, pcadd, t / every, buf ^ 1);
      }
    }
    if constexpr (ADD == 1 && !CADD_LATE) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < X_TX; ++i) cadd[j][i] = *reinterpret_cast<const u32x2*>(hb + cbase + 2 * j * X_PLANE + i * X_XSTRIDE);
    } else if constexpr (CADD_LATE) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < X_TX; ++i)
          *reinterpret_cast<u32x2*>(sCad + (((buf * NT + j) * X_TX + i) * NTHR + tid) * 8) =
              *reinterpret_cast<const u32x2*>(hb + cbase + 2 * j * X_PLANE + i * X_XSTRIDE);
    } else if constexpr (ADD == 3 && !A1) {
      const int av = (2 * yp + pdy) * 8 + pz, asw = ACH == 4 ? ((av >> 2) & 3) : 0;       // (the swizzle of the DMA source side)
      const char* ab = sA + buf * ABUF + av * ACH * 16 + (q4 & 1) * 8;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < X_TX; ++i)
          cadd[j][i] = *reinterpret_cast<const u32x2*>(ab + i * 64 * ACH * 16 + (((ns * NT + j) * 2 + (q4 >> 1)) ^ asw) * 16);
    }
  };

  // ---- fused cross-entropy epilogue ------------------------------------------------------------------------------------------
  float ce_nll = 0.f, ce_w = 0.f;
  // Dice counters (|pred & true|, |pred|, |true| per class): per LANE, two 16-bit fields per register (class c in word c / 2), one
  // voxel per lane and tile, flushed at every sample change (a field holds 65 535 tiles of one workgroup and sample).  (Wave ballots per class put ~100 scalar instructions per tile on
  // the CU's single scalar unit: 0.36 ms of this launch on a net whose predictions are not yet all background.)
  // ... kept in LDS, one word per (kind, class pair) and lane, updated with returnless ds_add_u32 (three per voxel): as per-lane
  // REGISTER fields the update was a dynamic index into 18 registers = ~60 compare / select / add instructions per voxel, 0.16 ms of
  // this launch on data where most waves hold foreground (the synthetic batch of the bench; real CT labels mostly take the
  // all-background shortcut below)
  if constexpr (CE) {
#pragma unroll
    for (int k = 0; k < 18; ++k) sLane[k * 64] = 0u;
  }
  unsigned int u_bg = 0u;                  // wave-uniform: voxels of all-background waves (they count for class 0 in all three kinds)
  auto cnt_flush = [&]() {                // per-lane fields -> wave sums -> the workgroup's LDS counters
    if (lane == 0 && u_bg != 0u) { atomicAdd(&sCnt[0], u_bg); atomicAdd(&sCnt[16], u_bg); atomicAdd(&sCnt[32], u_bg); }
    u_bg = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        unsigned int v = (sLane[(k * 6 + (c >> 1)) * 64] >> ((c & 1) * 16)) & 0xffffu;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && v != 0u) atomicAdd(&sCnt[k * 16 + c], v);
      }
#pragma unroll
    for (int k = 0; k < 18; ++k) sLane[k * 64] = 0u;
  };
  int ce_n = -1;
  float ce_coef = 0.f;
  auto ce_flush = [&](int n) {            // this workgroup's record of sample n (slot blockIdx.x) and its Dice counts
    const double a = wave_sum((double)ce_nll), b = wave_sum((double)ce_w);
    if (lane == 0) { sSum[2 * wave] = a; sSum[2 * wave + 1] = b; }
    cnt_flush();
    ce_nll = 0.f;
    ce_w = 0.f;
    __syncthreads();
    if (tid < E.R) {
      double v = 0.0;
      if (tid < 2) v = sSum[tid] + sSum[2 + tid] + sSum[4 + tid] + sSum[6 + tid];
      E.part[((int64_t)n * E.P + blockIdx.x) * E.R + tid] = v;
    }
    if (tid < 3 * E.C) {
      const int k = tid / E.C, c = tid % E.C;
      const unsigned int v = sCnt[k * 16 + c];
      if (v) atomicAdd(&E.cnt[((int64_t)n * 3 + k) * E.C + c], (unsigned long long)v);
    }
    __syncthreads();
    if (tid < 48) sCnt[tid] = 0u;
    __syncthreads();
  };
  // the label of this lane's voxel of a tile (x plane q4, patch voxel r16): requested ONE TILE AHEAD — loaded inside the epilogue it
  // put a full HBM round trip on every tile of the workgroup (0.9 ms per launch instead of ~0.5)
  // (Round 4: the voxel index of a lane = a per-lane constant + a scalar tile base inside a per-sample buffer range — the 64-bit
  // per-lane multiply chain of ((n Xo + x) Yo + y) Zo + z was eight quarter-rate instructions per tile and wave, here for the label
  // and again for the gradient store.  A sample's voxel count x row bytes is < 2^31: conv_halo_x_eligible checks the output tensor.)
  const int ce_vox = CE ? (q4 * P.Yo + 2 * yp + pdy) * P.Zo + pz : 0;       // this lane's voxel inside a tile at the origin
  const int ce_svox = P.Xo * P.Yo * P.Zo;
  auto ce_label = [&](const Org& o) -> int {
    const int xg = o.x0 + q4, yg = o.y0 + 2 * yp + pdy, zg = o.z0 + pz;
    const bool valid = xg < P.Xr && yg < P.Yr && zg < P.Zr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(E.labels) + (int64_t)o.n * ce_svox, 0, ce_svox, 0x00020000);
    const int tbase = (o.x0 * P.Yo + o.y0) * P.Zo + o.z0;
    return (int)__builtin_amdgcn_raw_buffer_load_b8(rs, valid ? ce_vox : (int)0x80000000, tbase, 0);     // out of range: 0
  };
  auto ce_epilogue = [&](const Org& o, f32x4 (&acc)[NT][X_TX], u32x2 (&cadd)[NT][X_TX], int t) {
    if (o.n != ce_n) {
      if (ce_n >= 0) ce_flush(ce_n);
      ce_n = o.n;
      // the sample's gradient scale, ONCE per sample: read inside the per-tile arithmetic it was a global load whose full wait
      // (vmcnt(0)) also drained the halo loads of the next tile, issued just before — every tile
      ce_coef = E.coef[(int64_t)o.n * E.coef_stride];
    }
    // logits = conv + bias (+ identity residual), then a wave-local exchange through LDS: lane (r16, q4) ends up with the 12 values
    // of voxel r16 of x plane q4 (it held channels 4 q4 .. 4 q4 + 3 of all four planes)
    // (two rounds of two planes: 1.5 KB of scratch per wave, so that three workgroups fit a CU; lanes q4 = 2r, 2r+1 read in round r)
    char* sc = sT + wave * (2 * 16 * 48);
    float x[12];
    if constexpr ((X_ABL & 2048) != 0) {      // (timing only: no exchange through LDS)
#pragma unroll
      for (int q = 0; q < 12; ++q) x[q] = acc[0][q & 3][q >> 2] + bias[0][q & 3];
    } else
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * r + ii;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[0][i][q] + bias[0][q];
        if constexpr (ADD == 1) {
          v[0] += h2f<H>(cadd[0][i][0] & 0xffffu); v[1] += h2f<H>(cadd[0][i][0] >> 16);
          v[2] += h2f<H>(cadd[0][i][1] & 0xffffu); v[3] += h2f<H>(cadd[0][i][1] >> 16);
        }
        if (q4 < 3) *reinterpret_cast<f32x4*>(sc + (ii * 16 + r16) * 48 + q4 * 16) = v;
      }
      if ((q4 >> 1) == r) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sc + ((q4 & 1) * 16 + r16) * 48 + q * 16);
          x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
        }
      }
    }
    const int C = E.C;
    const int xg = o.x0 + q4, yg = o.y0 + 2 * yp + pdy, zg = o.z0 + pz;
    const bool valid = xg < P.Xr && yg < P.Yr && zg < P.Zr;
    if constexpr ((X_ABL & 8) != 0) {
      const int64_t vox = (((int64_t)o.n * P.Xo + xg) * P.Yo + yg) * P.Zo + zg;        // timing-only: no loss arithmetic, one store
      if (valid) *reinterpret_cast<u32x2*>(E.dlogits + vox * E.g_ld * 2) = u32x2{__float_as_uint(x[0] + x[5] + x[9]), (uint32_t)t};
      return;
    }
    // ---- per-voxel cross-entropy.  seg_loss_kernel (loss_metric.hip) spends ~350 VALU instructions per voxel on ten expf, ten IEEE
    //      divisions and a logf; here that work is NOT hidden behind an HBM stream, so: exp through v_exp_f32, one reciprocal of the
    //      sum instead of ten divisions.  The prediction keeps the reference's exact semantics (softmax in fp32, THEN argmax, first
    //      maximal index, capstone/training/utils.py:19-20): the largest logit has e = exp(0) = 1 exactly, and another class can only
    //      tie with it after the division if its e is within an ulp of 1 — lanes that hold such a near-tie (rare; wave-uniform test)
    //      redo the softmax with expf and true divisions exactly as seg_loss_kernel does.
    // Packed (two classes per instruction) arithmetic: this part is NOT hidden behind an HBM stream (tools/ablate_head_ce.sh: 0.28 of
    // the launch's 0.79 ms), so the per-class selects of the first version (x[t], the one-hot of the gradient, the prediction scan,
    // the near-tie count: ~100 compare / select pairs) are indicator products: clamp(1 + (x - m) * 2^100) is 1 exactly where
    // x == m, clamp(1 - (c - t)^2) is the one-hot of the label.  Columns >= C carry a bias of -3e38 (set once, below): e = 0.
    f32x2 xp[6], ep[6], dp[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) xp[k] = f32x2{x[2 * k], x[2 * k + 1]};
    f32x2 mm = __builtin_elementwise_max(__builtin_elementwise_max(xp[0], xp[1]), __builtin_elementwise_max(xp[2], xp[3]));
    mm = __builtin_elementwise_max(mm, __builtin_elementwise_max(xp[4], xp[5]));
    const float m = fmaxf(mm[0], mm[1]);
    const f32x2 m2 = {m, m}, zero2 = {0.f, 0.f}, one2 = {1.f, 1.f};
    constexpr float BIG = 0x1p100f;
    f32x2 pf2 = zero2, nr2 = zero2, ss2 = zero2;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      dp[k] = xp[k] - m2;                                              // <= 0, = 0 exactly at the maximum (so e = 1 exactly there)
      ep[k] = f32x2{__expf(dp[k][0]), __expf(dp[k][1])};
      ss2 += ep[k];
      const f32x2 at_max = __builtin_elementwise_min(__builtin_elementwise_max(dp[k] * BIG + one2, zero2), one2);
      // within 0.001 of the maximum <=> e > 0.999 (the maximum itself counts once): the trigger of the exact path below
      const f32x2 close = __builtin_elementwise_min(__builtin_elementwise_max(dp[k] * BIG + (1.f + 0.0010005f * BIG), zero2), one2);
      pf2 += at_max * f32x2{(float)(2 * k), (float)(2 * k + 1)};
      nr2 += close;
    }
    const float ssum = ss2[0] + ss2[1];
    // no near-tie: exactly one class has x == m, the sum of c * [x_c == m] is its index
    int pred = (int)(pf2[0] + pf2[1]);
    const int near = (int)(nr2[0] + nr2[1]);
    float e[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) { e[2 * k] = ep[k][0]; e[2 * k + 1] = ep[k][1]; }
    if (!(X_ABL & 64) && __builtin_amdgcn_ballot_w64(near > 1) != 0ull) {
      if (near > 1) {
        float e2[12], s2 = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c) { e2[c] = (c < C) ? expf(x[c] - m) : 0.f; if (c < C) s2 += e2[c]; }
        float best = -1.f;
        pred = 0;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
          const float p2 = (c < C) ? e2[c] / s2 : 0.f;
          if (c < C && p2 > best) { best = p2; pred = c; }
        }
      }
    }
    const float rs = __builtin_amdgcn_rcpf(ssum);                  // (gradient only: 1 ulp)
    const float lse = m + __logf(ssum);
    // one-hot of the label as products: oh_c = clamp(1 - (c - t)^2); x_t = sum_c oh_c x_c (a column >= C has oh = 0 and a FINITE x)
    const float tf = (float)t;
    f32x2 oh[6], xt2 = zero2;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x2 dd = f32x2{(float)(2 * k), (float)(2 * k + 1)} - f32x2{tf, tf};
      oh[k] = __builtin_elementwise_min(__builtin_elementwise_max(one2 - dd * dd, zero2), one2);
      xt2 += oh[k] * xp[k];
    }
    const float xt = xt2[0] + xt2[1];
    const float w = (E.class_weight != nullptr && t < C) ? E.class_weight[t] : 1.f;
    if (valid) {
      ce_nll += w * (lse - xt);
      ce_w += w;
    }
    // a wave whose 64 voxels are all background in truth and prediction — the common case once the net has trained for a few
    // steps — adds one wave-uniform count to class 0; otherwise every lane updates its own packed fields
    const unsigned long long fgb = (X_ABL & 32) ? 0ull : __builtin_amdgcn_ballot_w64(valid && (t != 0 || pred != 0));
    if (fgb == 0ull) {
      if (!(X_ABL & 32)) u_bg += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(valid));
    } else if (valid) {
      const uint32_t it = 1u << ((t & 1) * 16), ip = 1u << ((pred & 1) * 16);
      const int wt = t < 12 ? (t >> 1) : 5, wp = pred >> 1;       // (labels are < C <= 12: host-checked; clamp keeps the address in range)
      atomicAdd(&sLane[(2 * 6 + wt) * 64], t < 12 ? it : 0u);
      atomicAdd(&sLane[(1 * 6 + wp) * 64], ip);
      if (pred == t) atomicAdd(&sLane[wp * 64], ip);
    }
    const float ce_scale = ce_coef * w;
    float d[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x2 g2 = (ep[k] * f32x2{rs, rs} - oh[k]) * f32x2{ce_scale, ce_scale};     // (columns >= C: e = 0, oh = 0)
      d[2 * k] = g2[0]; d[2 * k + 1] = g2[1];
    }
    {
      const bool st_ok = (X_ABL & 16) ? (valid && d[0] + d[5] == 123.f) : valid;
      const int row = E.g_ld * 2;
      const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(E.dlogits + (int64_t)o.n * ce_svox * row, 0, ce_svox * row, 0x00020000);
      const int gbase = ((o.x0 * P.Yo + o.y0) * P.Zo + o.z0) * row;
      const int gvo = st_ok ? ce_vox * row : (int)0x80000000;
      const u32x4 lo = {pack2<H>(d[0], d[1]), pack2<H>(d[2], d[3]), pack2<H>(d[4], d[5]), pack2<H>(d[6], d[7])};
      const u32x2 hi = {pack2<H>(d[8], d[9]), pack2<H>(d[10], d[11])};
      if (E.g_ld % 8 == 0) {       // 16-wide rows: 16-byte aligned
        __builtin_amdgcn_raw_buffer_store_b128(lo, grs, gvo, gbase, 0);
        asm volatile("s_nop 1" ::"v"(lo));        // (>64-bit store data with an SGPR offset: see the fp32 logits store above)
        __builtin_amdgcn_raw_buffer_store_b64(hi, grs, gvo + 16, gbase, 0);
      } else {                     // 12-wide rows (24 bytes, 8-byte aligned): three 8-byte stores
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo[0], lo[1]}, grs, gvo, gbase, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{lo[2], lo[3]}, grs, gvo + 8, gbase, 0);
        __builtin_amdgcn_raw_buffer_store_b64(hi, grs, gvo + 16, gbase, 0);
      }
    }
  };

  // ---- tile sequence of this workgroup (as conv_halo.hip: each XCD owns a contiguous range, walked round-robin) -----------------
  const int Gd = gridDim.x;
  int first, stride, last;
  if ((Gd & 7) == 0) {
    const int chunk = (total_tiles + 7) / 8, xcd = blockIdx.x & 7;
    first = xcd * chunk + (blockIdx.x >> 3);
    stride = Gd >> 3;
    last = (xcd + 1) * chunk < total_tiles ? (xcd + 1) * chunk : total_tiles;
  } else {
    first = blockIdx.x; stride = Gd; last = total_tiles;
  }
  // Per tile t: [wait for this wave's DMA pieces of tile t] [barrier: halo[buf] complete, everyone done with halo[buf^1]]
  // [DMA of tile t+1 into halo[buf^1]] [multiplies of tile t from halo[buf], with the epilogue of tile t-1 between its K steps].
  // The X_TX stores of that epilogue are the only vector-memory operations younger than the DMA, so the counted wait at the top
  // of the next tile (all but the X_TX youngest) covers exactly the DMA pieces; before the first epilogue it is a full wait.
  using T_ = std::true_type;
  using F_ = std::false_type;
  if (first >= last) {
    if constexpr (CE) {
      for (int n = 0; n < P.N; ++n)
        if (tid < E.R) E.part[((int64_t)n * E.P + blockIdx.x) * E.R + tid] = 0.0;
    }
    return;
  }
  f32x4 accA[NT][X_TX], accB[NT][X_TX];
  u32x2 caddA[NT][X_TX], caddB[NT][X_TX];
  Ep ep;
  Org ocur = tile_origin(first);
  Org onext = ocur;
  int t = first;
  if constexpr (CE) {
    // not software-pipelined: the loss arithmetic needs the registers the second accumulator set would take, and two workgroups per
    // CU overlap each other's phases
    if (tid < 48) sCnt[tid] = 0u;
    if constexpr (R12) {
      for (int i = tid; i < 2 * X_HVP; i += NTHR)
        *reinterpret_cast<u32x2*>(smem + (i / X_HVP) * CF::HALO + X_PLANE + (i % X_HVP) * 16 + 8) = u32x2{0u, 0u};
      __syncthreads();       // the operand-normalisation table is written
      gload(ocur);
      sstore(0, ocur);
    } else {
      dma(ocur, 0);
    }
    int buf = 0;
    int lab_cur = ce_label(ocur), lab_next = 0;
    for (;; t += stride, buf ^= 1) {
      const bool more = t + stride < last;
      if constexpr (R12) { if (!(X_ABL & 256)) __syncthreads(); }
      else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      if (more) {
        onext = tile_origin(t + stride);
        if constexpr (R12) gload(onext); else dma(onext, buf ^ 1);
        lab_next = ce_label(onext);
      }
      compute_tile(F_{}, buf, accA, caddA, ep, accA, caddA);
      ce_epilogue(ocur, accA, caddA, lab_cur);
      if constexpr (R12) { if (more) sstore(buf ^ 1, onext); }
      if (!more) break;
      ocur = onext;
      lab_cur = lab_next;
    }
    ce_flush(ce_n);
    // samples this workgroup never touched: zero records (the host sums every slot of every sample)
    const int n_first = first / G.tiles, n_last = (last - 1) / G.tiles;
    for (int n = 0; n < P.N; ++n)
      if ((n < n_first || n > n_last) && tid < E.R) E.part[((int64_t)n * E.P + blockIdx.x) * E.R + tid] = 0.0;
    return;
  }
  if constexpr (BST) {      // y values of the first tile (every later tile's are fetched by the slices of the tile before it)
    y_head(ocur, ynext);
#pragma unroll
    for (int i = 0; i < X_TX; ++i) y_load(ynext, i);
  }
  if constexpr (R12) {
    // zero the pad half (channels 12..15) of every plane-1 slot of both buffers once
    for (int i = tid; i < 2 * X_HVP; i += NTHR)
      *reinterpret_cast<u32x2*>(smem + (i / X_HVP) * CF::HALO + X_PLANE + (i % X_HVP) * 16 + 8) = u32x2{0u, 0u};
    __syncthreads();         // the operand-normalisation table is written
    gload(ocur);
    sstore(0, ocur);
    __syncthreads();
    if (t + stride < last) { onext = tile_origin(t + stride); gload(onext); }
    compute_tile(F_{}, 0, accA, caddA, ep, accA, caddA);
    if (t + stride < last) sstore(1, onext);
    __syncthreads();
  } else {
    dma(ocur, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // DMA pieces landed; the weight fragments this wave staged are written
    __builtin_amdgcn_s_barrier();
    cadd_early(caddA);
    if (t + stride < last) { onext = tile_origin(t + stride); dma(onext, 1); }
    compute_tile(F_{}, 0, accA, caddA, ep, accA, caddA);
  }
  // invariant at the loop head: set A holds the finished multiplies of tile t (origin ocur); tile t+stride (origin onext) is in
  // flight into halo[1] (R12: already stored there, behind a barrier)
  constexpr int NSTORE = X_TX * NT * (BST ? 2 : 1);      // vector-memory operations of one tile's epilogue (BST: a store and a y load per plane)
  auto advance = [&](int buf, f32x4 (&cur)[NT][X_TX], u32x2 (&ccur)[NT][X_TX], f32x4 (&prv)[NT][X_TX], u32x2 (&cprv)[NT][X_TX], bool counted) {
    // tile t+stride becomes the current one: its multiplies go to `cur` while the epilogue of tile t runs from `prv`
    if constexpr (!R12) {
      if (ADD == 2 || !counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
      __builtin_amdgcn_s_barrier();
    }
    const Org odone = ocur;
    ocur = onext;
    t += stride;
    const bool more = t + stride < last;
    cadd_early(ccur);
    if (more || BST) {
      onext = tile_origin(more ? t + stride : t);
      if constexpr (R12) gload(onext, more); else dma(onext, buf ^ 1, more);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BST) { y_head(ocur, ynext); y_more = true; }
    ep_head(odone, ep);
    compute_tile(T_{}, buf, cur, ccur, ep, prv, cprv);
    if constexpr (R12) {
      if (more) sstore(buf ^ 1, onext);     // halo[buf ^ 1] was last read by the multiplies of the previous tile, behind the last barrier
      __syncthreads();
    }
  };
  bool counted = false;
  while (true) {
    if (t + stride >= last) { y_more = false; ep_head(ocur, ep);
#pragma unroll
      for (int i = 0; i < X_TX; ++i) ep_slice(ep, accA, caddA, i, 0);
      break; }
    advance(1, accB, caddB, accA, caddA, counted);
    counted = true;
    if (t + stride >= last) { y_more = false; ep_head(ocur, ep);
#pragma unroll
      for (int i = 0; i < X_TX; ++i) ep_slice(ep, accB, caddB, i, 1);
      break; }
    advance(0, accA, caddA, accB, caddB, counted);
  }
  if (STATS && stat_n >= 0) flush_stats(stat_n);
  if (BST && bst_n >= 0) flush_bst(bst_n);
}

// -------------------------------------------------------------------------------------------------------------------------------------
static int x_tap_order(const ConvKArgs& a) {   // 0: canonical (offset = t - 1 per axis, x slowest), 1: every offset negated, -1: neither
  int kind = -1;
  for (int flip = 0; flip < 2 && kind < 0; ++flip) {
    bool ok = true;
    for (int j = 0; j < 27 && ok; ++j) {
      const int tp = a.cls[0].taps[j];
      const int s = flip ? -1 : 1;
      const int ex = s * (j / 9 - 1), ey = s * ((j / 3) % 3 - 1), ez = s * (j % 3 - 1);
      ok = (int)(int8_t)(tp & 0xff) == ex && (int)(int8_t)((tp >> 8) & 0xff) == ey && (int)(int8_t)((tp >> 16) & 0xff) == ez;
    }
    if (ok) kind = flip;
  }
  return kind;
}

bool conv_halo_x_eligible(const ConvKArgs& a, int dtype, int nclass) {
  if (getenv("CTSEG_NO_HALO_X") != nullptr) return false;
  if (!is16(dtype) || nclass != 1 || a.cls[0].ntaps != 27 || a.sin != 1 || a.sout != 1) return false;
  const int vb = a.Cg * 2;
  if (!(vb == 32 || vb == 64) || a.Cn > 32) return false;
  // gathered rows: 16-byte chunked, or 12 elements wide (24 bytes) with 16 gathered channels and at most 16 columns
  if (((a.g_ld * 2) % 16 != 0 && !(a.g_ld == 12 && vb == 32 && a.Cn <= 16)) || ((uintptr_t)a.in % 16) != 0) return false;
  if (a.Xr != a.Xi || a.Yr != a.Yi || a.Zr != a.Zi || a.Zr < 4) return false;
  const int64_t YZ = (int64_t)a.Yi * a.Zi;
  if (((int64_t)a.Xi * YZ + YZ + a.Zi + 1) * a.g_ld * 2 >= (1ll << 31) - 65536) return false;
  const int64_t osz = a.out_f32 ? 4 : 2;
  if ((int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * osz >= (1ll << 31) - 65536) return false;
  if (a.add != nullptr && (int64_t)a.Xo * a.Yo * a.Zo * a.add_ld * (a.add_f32 ? 4 : 2) >= (1ll << 31) - 65536) return false;
  if (a.out2 != nullptr) return false;
  if (a.cls[0].kpad < 27 * a.Cg) return false;
  const int order = x_tap_order(a);
  if (order < 0) return false;
  if (a.out_f32 && order == 1) return false;        // fp32 output exists for the forward tap order only (the logits)
  return true;                                      // (independent of a.stats: sizing queries run before the partial buffer exists)
}

// operand normalisation on load: the register-staged (12-wide rows) variant, constants of at most X_NRM_MAXN samples in LDS
bool conv_halo_x_in_norm_ok(const ConvKArgs& a, int dtype, int nclass) {
  return conv_halo_x_eligible(a, dtype, nclass) && a.g_ld == 12 && a.Cg == 16 && a.N <= X_NRM_MAXN && a.in_C <= 12;
}

// InstanceNorm partials exist for the forward tap order with 16-bit output and no addend (what the plans record)
bool conv_halo_x_stats_ok(const ConvKArgs& a) { return x_tap_order(a) == 0 && !a.out_f32 && a.add == nullptr;
}

static int x_tiles(const ConvKArgs& a) { return ((a.Xr + X_TX - 1) / X_TX) * ((a.Yr + X_TY - 1) / X_TY) * ((a.Zr + X_TZ - 1) / X_TZ); }

static int x_grid(const ConvKArgs& a) {
  const int vb = a.Cg * 2, total = x_tiles(a) * a.N;
  // LDS: 2 x 41.5 KB of halo (+ 27 / 54 KB of weight fragments) per workgroup at 64-byte voxels: one workgroup per CU;
  // 2 x 20.7 KB at 32-byte voxels with the 15 weight fragments of 16 columns in registers: two per CU
  const int per_cu = (vb == 64 || a.Cn > 16) ? 1 : 2;
  return persistent_grid(CTSEG_NUM_CU * per_cu, total);
}

int conv_halo_x_slots(const ConvKArgs& a) { return x_grid(a); }

static int x_add_kind(const ConvKArgs& a);
// backward statistics: the input-gradient tap order, 16-bit output, any addend but one the epilogue has to load from global memory
int conv_halo_x_bst_slots(const ConvKArgs& a) {
  if (x_tap_order(a) != 1 || a.out_f32 || a.stats != nullptr || a.bias != nullptr || x_add_kind(a) == 2) return 0;
  if (a.bst.col0 != 0 || a.bst.C > a.Cn_store || a.bst.y_ld != a.o_ld) return 0;      // y laid out like the written gradient
  { const char* e = getenv("CTSEG_BST_X64"); if (a.Cg * 2 == 64 && a.Cn > 16 && e != nullptr && e[0] == '0') return 0; }   // (A/B switch)
  if ((int64_t)a.Xo * a.Yo * a.Zo * a.bst.y_ld * 2 >= (1ll << 31) - 65536) return 0;
  return x_grid(a);
}

// how the pass takes its addend: 0 none, 1 centre voxel of the LDS halo, 2 loads in the epilogue, 3 LDS-DMA beside the halo
static int x_add_kind(const ConvKArgs& a) {
  if (a.add == nullptr) return 0;
  const int vb = a.Cg * 2, ncols = a.Cn > 16 ? 32 : 16;      // (the grid has one column of workgroups: they cover every stored channel)
  const bool stats = a.stats != nullptr, of32 = a.out_f32 != 0;
  // identity residual: the addend is the input tensor itself, every stored channel is present in the staged voxel
  const bool addc = a.add == a.in && a.add_ld == a.g_ld && a.add_f32 == 0 && a.Cn_store * 2 <= vb;
  // a 16-bit addend whose rows hold whole 16-byte chunks goes through the DMA pipeline (64-byte voxels: the LDS has room for it)
  const bool add_dma = vb == 64 && !of32 && !stats && !addc && a.add_f32 == 0 && (a.add_ld * 2) % 16 == 0 && ((uintptr_t)a.add % 16) == 0 &&
                       a.add_ld >= ncols && a.g_ld != 12;
  return addc ? 1 : (add_dma ? 3 : 2);
}

template <typename H, int VB, int NT, int NS, bool FLIP>
static void x_launch(ConvKArgs& a, const XGeom& g, int total, dim3 grid, hipStream_t st) {
  const bool stats = a.stats != nullptr, of32 = a.out_f32 != 0;
  const int add = x_add_kind(a);
  const dim3 blk(256 * NS);
  const bool r12 = a.g_ld == 12;
  if constexpr (FLIP && std::is_same<H, BF16>::value) {      // input-gradient passes: backward InstanceNorm statistics of the written gradient (training: bf16)
    if (a.bst.part != nullptr) {
#define X_BST(AD, R) hipLaunchKernelGGL((conv_halo_x_kernel<H, VB, NT, NS, true, false, AD, false, R, false, true>), grid, blk, 0, st, a, g, total, XCe{})
      if constexpr (VB == 32 && NS == 1) {
        if (r12) { if (add == 1) X_BST(1, true); else X_BST(0, true); return; }
      }
      if (add == 1) X_BST(1, false);
      else if (add == 3) { if constexpr (VB == 64) X_BST(3, false); }
      else X_BST(0, false);
#undef X_BST
      return;
    }
  }
#define X_GO(ST, AD, OF)                                                                                                  \
  do {                                                                                                                    \
    if constexpr (VB == 32 && NS == 1) {                                                                                  \
      if (r12) { hipLaunchKernelGGL((conv_halo_x_kernel<H, VB, NT, NS, FLIP, ST, AD, OF, true>), grid, blk, 0, st, a, g, total, XCe{}); break; } \
    }                                                                                                                     \
    hipLaunchKernelGGL((conv_halo_x_kernel<H, VB, NT, NS, FLIP, ST, AD, OF, false>), grid, blk, 0, st, a, g, total, XCe{});  \
  } while (0)
  if constexpr (!FLIP) {     // forward passes: InstanceNorm partials (16-bit output, no addend) or the fp32 logits
    if (stats) { X_GO(true, 0, false); return; }
    if (of32) {
      if (add == 1) X_GO(false, 1, true); else if (add == 2) X_GO(false, 2, true); else X_GO(false, 0, true);
      return;
    }
  }
  if (add == 1) X_GO(false, 1, false);
  else if (add == 2) X_GO(false, 2, false);
  else if (add == 3) { if constexpr (VB == 64) hipLaunchKernelGGL((conv_halo_x_kernel<H, VB, NT, NS, FLIP, false, 3, false, false>), grid, blk, 0, st, a, g, total, XCe{}); }
  else X_GO(false, 0, false);
#undef X_GO
}

void launch_conv_halo_x(ConvKArgs& a, hipStream_t st) {
  XGeom g;
  g.tyn = (a.Yr + X_TY - 1) / X_TY; g.tzn = (a.Zr + X_TZ - 1) / X_TZ;
  g.tiles = x_tiles(a);
  a.tiles = g.tiles;
  g.in_sample_bytes = (int)((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2);
  g.out_sample_bytes = (int)((int64_t)a.Xo * a.Yo * a.Zo * a.o_ld * (a.out_f32 ? 4 : 2));
  g.add_sample_bytes = a.add ? (int)((int64_t)a.Xo * a.Yo * a.Zo * a.add_ld * (a.add_f32 ? 4 : 2)) : 0;
  g.y_sample_bytes = a.bst.part ? (int)((int64_t)a.Xo * a.Yo * a.Zo * a.bst.y_ld * 2) : 0;
  const int total = g.tiles * a.N;
  const int vb = a.Cg * 2, nt = a.Cn > 16 ? 2 : 1;
  const dim3 grid((unsigned)x_grid(a), 1u, 1u);
  const bool flip = x_tap_order(a) == 1;
#define X_DT(H)                                                                                                         \
  do {                                                                                                                  \
    if (vb == 64) {                                                                                                     \
      if (nt == 2) { if (flip) x_launch<H, 64, 1, 2, true>(a, g, total, grid, st); else x_launch<H, 64, 1, 2, false>(a, g, total, grid, st); } \
      else { if (flip) x_launch<H, 64, 1, 1, true>(a, g, total, grid, st); else x_launch<H, 64, 1, 1, false>(a, g, total, grid, st); }         \
    } else {                                                                                                            \
      if (nt == 2) { if (flip) x_launch<H, 32, 1, 2, true>(a, g, total, grid, st); else x_launch<H, 32, 1, 2, false>(a, g, total, grid, st); } \
      else { if (flip) x_launch<H, 32, 1, 1, true>(a, g, total, grid, st); else x_launch<H, 32, 1, 1, false>(a, g, total, grid, st); }         \
    }                                                                                                                   \
  } while (0)
  if (a.dtype == CTSEG_F16) X_DT(F16); else X_DT(BF16);
#undef X_DT
}

// ---- logits convolution with the cross-entropy fused into its epilogue --------------------------------------------------------
bool conv_halo_x_ce_eligible(const ConvKArgs& a, int dtype, int nclass, int C) {
  if (!conv_halo_x_eligible(a, dtype, nclass)) return false;
  if (a.Cg != 16 || a.Cn != C || C > 12 || C < 2 || a.stats != nullptr || x_tap_order(a) != 0) return false;
  if (a.add != nullptr && !(a.add == a.in && a.add_ld == a.g_ld && a.add_f32 == 0 && a.Cn_store * 2 <= 32)) return false;   // identity residual only
  return true;
}

static int x_grid_ce(const ConvKArgs& a) {     // 48 KB of LDS but ~230 registers per lane (weights + the softmax's arrays): two workgroups per CU
  return persistent_grid(CTSEG_NUM_CU * 2, x_tiles(a) * a.N);
}

int conv_halo_x_ce_slots(const ConvKArgs& a) { return x_grid_ce(a); }

template <typename H> static void x_launch_ce(ConvKArgs& a, const XGeom& g, int total, dim3 grid, const XCe& e, hipStream_t st) {
  const bool r12 = a.g_ld == 12, addc = a.add != nullptr;
  const dim3 blk(256);
#define X_CE(AD, R)                                                                                                         \
  hipLaunchKernelGGL((conv_halo_x_kernel<H, 32, 1, 1, false, false, AD, true, R, true>), grid, blk, 0, st, a, g, total, e)
  if (addc) { if (r12) X_CE(1, true); else X_CE(1, false); }
  else { if (r12) X_CE(0, true); else X_CE(0, false); }
#undef X_CE
}

void launch_conv_halo_x_ce(ConvKArgs& a, const XCe& e, hipStream_t st) {
  XGeom g;
  g.tyn = (a.Yr + X_TY - 1) / X_TY; g.tzn = (a.Zr + X_TZ - 1) / X_TZ;
  g.tiles = x_tiles(a);
  a.tiles = g.tiles;
  g.in_sample_bytes = (int)((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * 2);
  g.out_sample_bytes = 0;
  g.add_sample_bytes = 0;
  g.y_sample_bytes = 0;
  const int total = g.tiles * a.N;
  const dim3 grid((unsigned)x_grid_ce(a), 1u, 1u);
  if (a.dtype == CTSEG_F16) x_launch_ce<F16>(a, g, total, grid, e, st); else x_launch_ce<BF16>(a, g, total, grid, e, st);
}

}  // namespace ctseg

using namespace ctseg;

static void x_fill(const ctseg_conv_desc* d, ConvKArgs& a) {
  a.in = (const char*)d->in; a.w = (const char*)d->w; a.bias = d->bias; a.out = (char*)d->out; a.add = (const char*)d->add;
  a.stats = d->stats;
  a.N = d->N; a.Xi = d->Xi; a.Yi = d->Yi; a.Zi = d->Zi; a.Xr = d->Xr; a.Yr = d->Yr; a.Zr = d->Zr; a.Xo = d->Xo; a.Yo = d->Yo; a.Zo = d->Zo;
  a.Cg = d->Cg; a.Cn = d->Cn; a.Cn_store = d->Cn_store; a.g_ld = d->g_ld; a.o_ld = d->o_ld; a.add_ld = d->add_ld;
  a.sin = d->sin; a.sout = d->sout; a.rows = d->Xr * d->Yr * d->Zr; a.tiles = 0; a.out_f32 = d->out_f32; a.add_f32 = d->add_f32;
  a.stats_ld = d->stats_ld; a.stats_tiles = d->stats_tiles; a.stats_tile0 = d->stats_tile0;
  for (int c = 0; c < CTSEG_MAX_CLASSES; ++c) a.cls[c] = d->cls[c < d->nclass ? c : 0];
  a.out2 = nullptr; a.out2_col0 = 0; a.o2_ld = 0; a.dtype = d->dtype; a.xcd_order = 0;
  a.in_mr = d->in_mean_rstd; a.in_alpha = d->in_alpha; a.in_C = d->in_norm_C;
  a.bst = BstArgs{};
}

extern "C" int ctseg_conv_logits_ce_slots(const ctseg_conv_desc* d, int32_t C) {
  if (!desc_ok(d) || d->nclass != 1) return 0;
  ConvKArgs a;
  x_fill(d, a);
  a.stats = nullptr;
  if (!conv_halo_x_ce_eligible(a, d->dtype, d->nclass, C)) return 0;
  return conv_halo_x_ce_slots(a);
}

extern "C" int ctseg_conv_logits_ce(const ctseg_conv_desc* d, const uint8_t* labels, int32_t C, const float* class_weight,
                                    const float* coef, int32_t coef_stride, void* dlogits, int32_t g_ld, double* part, int32_t P,
                                    int32_t R, int64_t* cnt, void* stream) {
  CTSEG_REQUIRE_DESC(d, "conv_logits_ce");
  CTSEG_REQUIRE(d->in && d->w && labels && coef && dlogits && part && cnt, "conv_logits_ce: null pointer");
  CTSEG_REQUIRE(d->nclass == 1, "conv_logits_ce: one tap class expected");
  ConvKArgs a;
  x_fill(d, a);
  CTSEG_REQUIRE(conv_halo_x_ce_eligible(a, d->dtype, d->nclass, C), "conv_logits_ce: pass not eligible (ask ctseg_conv_logits_ce_slots)");
  const int slots = conv_halo_x_ce_slots(a);
  CTSEG_REQUIRE(P >= slots && R >= 2 && R <= 256, "conv_logits_ce: part needs >= %d slots per sample (P = %d)", slots, P);
  CTSEG_REQUIRE(g_ld >= C && g_ld % 4 == 0 && ((uintptr_t)dlogits % 8) == 0 && d->Xo == d->Xr && d->Yo == d->Yr && d->Zo == d->Zr,
                "conv_logits_ce: dlogits layout");
  XCe e;
  e.labels = labels; e.class_weight = class_weight; e.coef = coef; e.coef_stride = coef_stride; e.dlogits = (char*)dlogits;
  e.g_ld = g_ld; e.part = part; e.P = P; e.R = R; e.cnt = (unsigned long long*)cnt; e.C = C;
  launch_conv_halo_x_ce(a, e, (hipStream_t)stream);
  CTSEG_LAUNCH_CHECK("conv_logits_ce");
  return 0;
}
