// Shared pieces of the implicit-GEMM convolution kernels (generic gather kernel + LDS-halo kernel).
#pragma once
#include "ctseg_dev.h"

namespace ctseg {

// Backward InstanceNorm statistics taken in the epilogue of the pass that writes the gradient (ctseg_conv_desc::bst_*): the three
// sums instnorm_prelu_bwd_reduce_kernel (norm_act.hip) computes in a pass of its own, term for term the same arithmetic.
struct BstArgs {
  const char* y;        // forward conv output the norm normalised, indexed like the written tensor; channel c <-> GEMM column col0 + c
  const float* mr;      // [N][C][2] (mean, rstd)
  const float* alpha;   // PReLU slope
  float* part;          // [N][P][3][ld] fp32, one row per workgroup (or tile) and sample; nullptr: off
  int y_ld, C, col0, P, ld;
};

__device__ __forceinline__ void bst_term(float g, float yv, float mean, float rstd, float al, float& a1, float& a2, float& a3) {
  const float xh = (yv - mean) * rstd;
  const float dxh = g * (xh > 0.f ? 1.f : al);
  a1 += dxh;
  a2 += dxh * xh;
  a3 += xh > 0.f ? 0.f : g * xh;
}

// the same three sums for two channels held as a packed bf16 pair (low half = first channel): full-rate packed fp32 arithmetic,
// xhat = fma(y, rstd, -mean * rstd).  a3 is ONE accumulator (the slope is one parameter: only the total over channels matters).
__device__ __forceinline__ void bst_pair_bf16(uint32_t gw, uint32_t yw, f32x2 rstd, f32x2 nmr, float al, f32x2& a1, f32x2& a2, float& a3) {
  const f32x2 g2 = {__uint_as_float(gw << 16), __uint_as_float(gw & 0xffff0000u)};
  const f32x2 y2 = {__uint_as_float(yw << 16), __uint_as_float(yw & 0xffff0000u)};
  const f32x2 xh = y2 * rstd + nmr;
  const f32x2 sel = {xh[0] > 0.f ? 1.f : al, xh[1] > 0.f ? 1.f : al};
  const f32x2 dxh = g2 * sel;
  a1 += dxh;
  a2 += dxh * xh;
  a3 = fmaf(g2[0], fminf(xh[0], 0.f), fmaf(g2[1], fminf(xh[1], 0.f), a3));
}

struct ConvKArgs {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  const char* add;
  float* stats;
  int N, Xi, Yi, Zi, Xr, Yr, Zr, Xo, Yo, Zo;
  int Cg, Cn, Cn_store, g_ld, o_ld, add_ld;
  int sin, sout;
  int rows, tiles;
  int out_f32, add_f32;
  int stats_ld, stats_tiles, stats_tile0;
  ctseg_conv_class cls[CTSEG_MAX_CLASSES];
  // split output (ctseg_conv_desc::out2): columns >= out2_col0 go to out2; stem and stride-2 halo kernels only
  char* out2;
  int out2_col0, o2_ld;
  int dtype;       // CTSEG_F32 / CTSEG_BF16 / CTSEG_F16 storage of the pass (the launchers pick the instantiation from it)
  int xcd_order;   // generic / ring kernels: grid.x = 8 * ceil(tiles*N / 8), workgroup L takes tile (L&7)*chunk + (L>>3)
  // InstanceNorm + PReLU of the gathered operand on load (ctseg_conv_desc::in_mean_rstd): x-column halo pass over 12-wide rows only
  const float* in_mr;
  const float* in_alpha;
  int in_C;
  BstArgs bst;
};

static inline void fill_bst(const ctseg_conv_desc* d, ConvKArgs& a) {
  a.bst.y = (const char*)d->bst_y; a.bst.mr = d->bst_mean_rstd; a.bst.alpha = d->bst_alpha; a.bst.part = d->bst_partials;
  a.bst.y_ld = d->bst_y_ld; a.bst.C = d->bst_C; a.bst.col0 = d->bst_col0; a.bst.P = d->bst_P; a.bst.ld = d->bst_ld;
}

// Workgroups reach the 8 XCDs round-robin by linear id.  Tile = (L & 7) * chunk + (L >> 3) gives every XCD one contiguous
// range of row tiles, so the halo rows neighbouring tiles share are fetched by ONE L2 instead of by all eight.  -1: no tile.
__device__ __forceinline__ int xcd_tile(int L, int total) {
  const int chunk = (total + 7) >> 3, i = L >> 3;
  const int t = (L & 7) * chunk + i;
  return (i < chunk && t < total) ? t : -1;
}

template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& wfrag, const u32x4& xfrag);
template <> __device__ __forceinline__ void mma16<BF16>(f32x4& acc, const u32x4& wfrag, const u32x4& xfrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfrag), __builtin_bit_cast(bf16x8, xfrag), acc,
                                                0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<F16>(f32x4& acc, const u32x4& wfrag, const u32x4& xfrag) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wfrag), __builtin_bit_cast(f16x8, xfrag), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& wfrag, const u32x4& xfrag) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wfrag[s]), __uint_as_float(xfrag[s]), acc, 0, 0, 0);
}

// Epilogue shared by both kernels.  Lane holds, for MFMA tile (j,i): voxel row = wm*WTM + i*16 + r16 of the
// workgroup tile, channels wn*WTN + j*16 + 4*q4 + {0..3}.  sRow[2r] = xr | yr<<16, sRow[2r+1] = zr (or < 0: no voxel).
// + bias, per-(tile, channel) sum / sumsq partials, transpose through LDS (`smem`, BM*(BN*OSZ+16) bytes),
// optional addend, 16-byte coalesced channels-last stores.  Ends with all LDS reads done but NO trailing barrier.
// BST (compile time): the instantiation that can take ConvKArgs::bst.  It is a kernel of its own because its epilogue holds 50 more
// registers than the main loop needs (128 x 64 tile: 108 -> 180, three -> two workgroups per CU: +29 % on the 8-class forward pass
// that never asks for the statistics, when the flag was a run-time one).
template <typename T, int BM, int BN, int WGM, int WGN, bool BST = false>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& P, const ctseg_conv_class& K, char* smem, float* sStats,
                                              const int* sRow, f32x4 (&acc)[BN / WGN / 16][BM / WGM / 16], int n, int tile,
                                              int cls_index, int col0) {
  constexpr int SZ = TT<T>::SZ, NTHR = 64 * WGM * WGN;
  using H = typename TT<T>::H;
  constexpr int WTM = BM / WGM, WTN = BN / WGN, MT = WTM / 16, NT = WTN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, q4 = lane >> 4;
  // lane holds, for MFMA tile (j,i): voxel row = wm*WTM + i*16 + r16, channels wn*WTN + j*16 + 4*q4 + {0..3}
  const bool of32 = P.out_f32 != 0;
  const int OSZ = of32 ? 4 : SZ;
  const int crow = BN * OSZ + 16;
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch = col0 + wn * WTN + j * 16 + 4 * q4;
    float bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bv[e] = (P.bias != nullptr && ch + e < P.Cn) ? P.bias[ch + e] : 0.f;
      ssum[j][e] = 0.f;
      ssq[j][e] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int r = wm * WTM + i * 16 + r16;
      const bool rv = sRow[2 * r + 1] >= 0;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[j][i][e] + bv[e];
        if (rv) { ssum[j][e] += v[e]; ssq[j][e] += v[e] * v[e]; }
      }
      char* cp = smem + r * crow + (wn * WTN + j * 16 + 4 * q4) * OSZ;
      if (of32 || SZ == 4) {
        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
        *reinterpret_cast<u32x2*>(cp) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
      }
    }
  }
  if (P.stats != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = ssum[j][e], b = ssq[j][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (r16 == 0) {
          const int c = wn * WTN + j * 16 + 4 * q4 + e;
          sStats[(wm * 2 + 0) * BN + c] = a;
          sStats[(wm * 2 + 1) * BN + c] = b;
        }
      }
  }
  __syncthreads();
  if (P.stats != nullptr && tid < 2 * BN) {
    const int which = tid / BN, c = tid % BN;
    float a = 0.f;
#pragma unroll
    for (int m = 0; m < WGM; ++m) a += sStats[(m * 2 + which) * BN + c];
    const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + (int64_t)cls_index * P.tiles + tile;
    P.stats[(slot_t * 2 + which) * P.stats_ld + col0 + c] = a;
  }
  // backward InstanceNorm statistics of the stored gradient (ConvKArgs::bst): a thread keeps ONE 8-channel chunk column through the
  // store loop below (NTHR is a multiple of the chunks per row), so its 8 + 8 + 1 sums stay in registers for the whole tile
  const bool bst = BST && SZ == 2 && !of32 && P.bst.part != nullptr;
  f32x2 q_rs[4], q_nm[4], q1[4], q2[4];
  float q3 = 0.f, q_al = 1.f;
  int q_ch0 = 0;                                  // channel of the norm the thread's chunk starts at (may be < 0 or >= C: not a channel)
  if (bst) {
    q_al = P.bst.alpha[0];
    q_ch0 = col0 + (tid % (BN / 8)) * 8 - P.bst.col0;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      q1[h] = f32x2{0.f, 0.f}; q2[h] = f32x2{0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = q_ch0 + 2 * h + e;
        const bool okc = c >= 0 && c < P.bst.C;
        const float mean = okc ? P.bst.mr[((int64_t)n * P.bst.C + c) * 2] : 0.f, rstd = okc ? P.bst.mr[((int64_t)n * P.bst.C + c) * 2 + 1] : 0.f;
        q_rs[h][e] = rstd;
        q_nm[h][e] = -mean * rstd;
      }
    }
  }
  {
    const int EPO = 16 / OSZ;               // output elements per 16-byte chunk
    const int cpr = BN / EPO;                // chunks per tile row
    const bool af32 = P.add_f32 != 0;
    const int ASZ = af32 ? 4 : SZ;
    // batches of 4 chunks per thread: all addend loads of a batch are issued before the first use, so their
    // latency overlaps (one exposed round trip per batch instead of one per chunk)
    constexpr int UN = 4;
    const int nadd = (af32 || SZ == 4) ? EPO * 4 : EPO * 2;   // addend bytes per chunk: 8, 16 or 32
    for (int base = tid; base < BM * cpr; base += NTHR * UN) {
      char* op[UN];
      const char* cp[UN];
      u32x4 a0[UN], a1[UN], yq[UN];
      bool ok[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = base + u * NTHR;
        const int r = (idx < BM * cpr) ? idx / cpr : 0, cc = idx - r * cpr;
        const int ch = col0 + cc * EPO;
        const int z = sRow[2 * r + 1];
        ok[u] = idx < BM * cpr && z >= 0 && ch < P.Cn_store;
        const int xy = sRow[2 * r];
        const int64_t vox = (((int64_t)n * P.Xo + (xy & 0xffff) * P.sout + K.ox) * P.Yo + (xy >> 16) * P.sout + K.oy) * P.Zo +
                            z * P.sout + K.oz;
        cp[u] = smem + r * crow + cc * 16;
        op[u] = P.out + (vox * P.o_ld + ch) * OSZ;
        a0[u] = u32x4{0u, 0u, 0u, 0u};
        a1[u] = a0[u];
        yq[u] = a0[u];
        if (bst && ok[u] && q_ch0 >= 0 && q_ch0 < P.bst.C)
          yq[u] = *reinterpret_cast<const u32x4*>(P.bst.y + (vox * P.bst.y_ld + q_ch0) * 2);
        if (ok[u] && P.add != nullptr) {
          const char* ap = P.add + (vox * P.add_ld + ch) * ASZ;
          if (nadd == 8) { const u32x2 t = *reinterpret_cast<const u32x2*>(ap); a0[u][0] = t[0]; a0[u][1] = t[1]; }
          else {
            a0[u] = *reinterpret_cast<const u32x4*>(ap);
            if (nadd == 32) a1[u] = *reinterpret_cast<const u32x4*>(ap + 16);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (!ok[u]) continue;
        if (P.add == nullptr) {
          const u32x4 t = *reinterpret_cast<const u32x4*>(cp[u]);
          *reinterpret_cast<u32x4*>(op[u]) = t;
          if (bst) {
#pragma unroll
            for (int h = 0; h < 4; ++h) bst_pair_bf16(t[h], yq[u][h], q_rs[h], q_nm[h], q_al, q1[h], q2[h], q3);
          }
        } else if (bst) {       // (16-bit output: the sums are over the values as stored)
          float v[8], a[8];
          load_n_as_float<H>(cp[u], false, 8, v);
          if (af32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = __uint_as_float(a0[u][e]); a[4 + e] = __uint_as_float(a1[u][e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[2 * e] = h2f<H>(a0[u][e] & 0xffffu); a[2 * e + 1] = h2f<H>(a0[u][e] >> 16); }
          }
          u32x4 t;
#pragma unroll
          for (int h = 0; h < 4; ++h) t[h] = pack2<H>(v[2 * h] + a[2 * h], v[2 * h + 1] + a[2 * h + 1]);
          *reinterpret_cast<u32x4*>(op[u]) = t;
#pragma unroll
          for (int h = 0; h < 4; ++h) bst_pair_bf16(t[h], yq[u][h], q_rs[h], q_nm[h], q_al, q1[h], q2[h], q3);
        } else {
          float v[8], a[8];
          load_n_as_float<H>(cp[u], of32 || SZ == 4, EPO, v);
          if (af32 || SZ == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = __uint_as_float(a0[u][e]); a[4 + e] = __uint_as_float(a1[u][e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[2 * e] = h2f<H>(a0[u][e] & 0xffffu); a[2 * e + 1] = h2f<H>(a0[u][e] >> 16); }
          }
          for (int e = 0; e < EPO; ++e) v[e] += a[e];
          if (of32 || SZ == 4) store_chunk<float>(op[u], v);
          else store_chunk<H>(op[u], v);
        }
      }
    }
  }
  if (bst) {
    // lanes l, l + cpr, l + 2 cpr ... of a wave hold the same chunk column (cpr = BN / 8 divides 64): butterflies, then the waves
    // in order through LDS (the transposed tile is dead once every thread has left the store loop) — fixed order, no atomics
    constexpr int CPR = BN / 8, NW = WGM * WGN;
    float qv[17];
#pragma unroll
    for (int h = 0; h < 4; ++h) { qv[2 * h] = q1[h][0]; qv[2 * h + 1] = q1[h][1]; qv[8 + 2 * h] = q2[h][0]; qv[9 + 2 * h] = q2[h][1]; }
    qv[16] = q3;
#pragma unroll
    for (int k = 0; k < 17; ++k)
#pragma unroll
      for (int o = 32; o >= CPR; o >>= 1) qv[k] += __shfl_xor(qv[k], o, 64);
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (lane < CPR) {
#pragma unroll
      for (int k = 0; k < 17; ++k) red[(wave * CPR + lane) * 17 + k] = qv[k];
    }
    __syncthreads();
    for (int i = tid; i < 17 * CPR; i += NTHR) {
      const int cc = i / 17, k = i - cc * 17;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * CPR + cc) * 17 + k];
      const int which = k < 8 ? 0 : (k < 16 ? 1 : 2);
      const int c = col0 + cc * 8 + (k & 7) * (k < 16 ? 1 : 0) - P.bst.col0;
      const int64_t slot = (int64_t)n * P.bst.P + (int64_t)cls_index * P.tiles + tile;
      if (c >= 0 && c < P.bst.C) P.bst.part[(slot * 3 + which) * P.bst.ld + c] = s;
    }
  }
}

// LDS-halo kernel (conv_halo.hip): 3x3x3 stride-1 passes with few channels, input tile staged once for all 27 taps
bool conv_halo_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_halo_tiles(const ConvKArgs& a);
int conv_halo_slots(const ConvKArgs& a, int dtype);   // InstanceNorm partial slots per sample (= workgroups)
void launch_conv_halo(ConvKArgs& a, int dtype, hipStream_t st);
// conv_halo_x.hip: the same passes in 16-bit storage with 16-byte-chunked rows: LDS-DMA staging, x-column fragment reuse,
// weights in registers (taken first where eligible)
bool conv_halo_x_eligible(const ConvKArgs& a, int dtype, int nclass);
bool conv_halo_x_stats_ok(const ConvKArgs& a);
int conv_halo_x_bst_slots(const ConvKArgs& a);   // 0: this pass cannot take ConvKArgs::bst
int conv_halo_x_slots(const ConvKArgs& a);
void launch_conv_halo_x(ConvKArgs& a, hipStream_t st);
// 8-class stride-2 "up" pass with <= 16 output channels (conv_up_halo.hip): one input tile for all parity classes
bool conv_up_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_up_tiles(const ConvKArgs& a);
int conv_up_slots(const ConvKArgs& a);   // InstanceNorm partial slots per sample (= workgroups)
void launch_conv_up(ConvKArgs& a, hipStream_t st);
// single-channel 3x3x3 stride-2 stem (conv_stem.hip)
bool conv_stem_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_stem_slots(const ConvKArgs& a);
void launch_conv_stem(ConvKArgs& a, hipStream_t st);
// conv_halo_sw.hip: LDS halo + streamed weights, Cg=64->Cn=64 (one class) and Cg=128->Cn=32 (8 parity classes), bf16
bool conv_halo_x_in_norm_ok(const ConvKArgs& a, int dtype, int nclass);
bool conv_halo_sw_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_halo_sw_slots(const ConvKArgs& a);
int conv_halo_sw_bst_slots(const ConvKArgs& a, int nclass);   // 0: this pass cannot take ConvKArgs::bst
void launch_conv_halo_sw(ConvKArgs& a, int nclass, hipStream_t st);
// conv_up8.hip: 8-class stride-2 passes with >= 128 gathered channels and 64 columns: all classes' accumulators live, K chunked
bool conv_up8_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_up8_slots(const ConvKArgs& a);
void launch_conv_up8(ConvKArgs& a, hipStream_t st);
// conv_down_halo.hip: stride-2 3x3x3 conv with 16 / 32 gathered channels, 64 columns per workgroup, bf16
bool conv_down_halo_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_down_halo_slots(const ConvKArgs& a);
int conv_down_halo_bst_slots(const ConvKArgs& a);   // 0: this pass cannot take ConvKArgs::bst
void launch_conv_down_halo(ConvKArgs& a, hipStream_t st);
bool conv_down_r_eligible(const ConvKArgs& a, int dtype, int nclass);
int conv_down_r_slots(const ConvKArgs& a);
int conv_down_r_bst_slots(const ConvKArgs& a);   // 0: this pass cannot take ConvKArgs::bst
void launch_conv_down_r(ConvKArgs& a, hipStream_t st);
// conv_igemm_ring.hip: the 192 x 256 bf16 tile with a five-deep ring of 32-wide K stages (a.tiles already set for 192 rows)
bool conv_ring_eligible(const ConvKArgs& a, int dtype, int nclass);
void launch_conv_ring(const ConvKArgs& a, int nclass, hipStream_t st);

}  // namespace ctseg
