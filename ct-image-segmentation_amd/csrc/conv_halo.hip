// LDS-halo implicit-GEMM kernel for 3x3x3 stride-1 passes with few channels (gfx950).
//
// The full- and half-resolution layers of the reference's U-Net (32->32 at 256x256x24, the 10->10 logits conv
// at 512x512x48 and their input gradients) are HBM-bound: per output voxel the generic kernel re-gathers 27
// neighbours through the vector-memory path.  Here a persistent workgroup keeps ALL packed weights in LDS, stages
// the (4+2)x(8+2)x(8+2) input halo of a 4x8x8 output tile ONCE (16-byte chunks, zero fill at the border) two
// tiles ahead of the MFMAs, and reads every tap's operand from LDS: 27x reuse.
//
// LDS image of the halo: one plane per 16-byte channel chunk, [plane][600 voxels][16 B].  600*16 B = 8 slots
// (mod 16), and inside each 2x8 (y,z) patch of 16 voxels the lane<->voxel map below is chosen so that every
// ds_read_b128 lane group of an MFMA operand hits 16 distinct 16-byte slots for every tap: conflict free
// (checked exhaustively on the host).
//
// These layers do ~430 MACs per output value, so per-tile bookkeeping is what has to be cheap: all gather / store
// offsets are per-thread constants computed once (only a scalar tile base changes), and the epilogue stores straight
// from the accumulators — with the weight tile as first MFMA operand a lane owns 4 consecutive channels of one voxel
// (8 B bf16 / 16 B fp32) — + bias, InstanceNorm partial sums, optional addend.  Same contract as conv_igemm.hip, so
// ctseg_conv_igemm picks the kernel transparently.
#include "conv_common.h"

namespace ctseg {

constexpr int H_TX = 4, H_TY = 8, H_TZ = 8;
constexpr int H_HX = H_TX + 2, H_HY = H_TY + 2, H_HZ = H_TZ + 2, H_HV = H_HX * H_HY * H_HZ;  // 600
constexpr int H_PLANE = H_HV * 16;

// r16 -> (dy, z) inside a 2x8 patch:  A = {0,1,2,3,12,13,14,15} -> row0 z0..4, row1 z0..2 ; B = {4..11} -> row0 z5..7, row1 z3..7
__device__ __forceinline__ void patch_voxel(int r16, int& dy, int& z) {
  dy = (0xEF80u >> r16) & 1;
  z = (int)((0x2104765437653210ull >> (4 * r16)) & 7ull);
}

template <int VB, int NT> struct HaloCfg {
  // 64-byte voxels: one workgroup per CU (134 KB of LDS) -> 8 waves, two row tiles each, so the MFMAs of one wave cover the
  // LDS latency of another; 32-byte voxels fit three 4-wave workgroups per CU
  static constexpr int NW = VB == 64 ? 8 : 4, NTHR = 64 * NW, RT = 16 / NW;
  static constexpr int NPL = VB / 16;
  static constexpr int HALO = NPL * H_PLANE;
  static constexpr int BN = 16 * NT;
  static constexpr int KC = (27 * VB + 63) / 64;   // 64-byte K chunks
  static constexpr int NSTG = (KC + 1) / 2;        // 128-byte weight stages
  static constexpr int WBYTES = NSTG * BN * 128;
  static constexpr int TOTAL = WBYTES + 2 * HALO + NW * 2 * BN * 4 + 32 * 4;
};

// ADDC: the addend IS the input tensor (identity residual: out = conv(x) + x, and its gradient) — taken from the centre
// voxel of the LDS halo instead of a second trip to HBM.
template <typename T, int VB, int NT, bool STATS, bool ADDC>
__global__ __launch_bounds__((HaloCfg<VB, NT>::NTHR)) void conv_halo_kernel(const ConvKArgs P, int total_tiles, int tyn, int tzn) {
  constexpr int SZ = TT<T>::SZ, EPC = TT<T>::EPC;
  using H = typename TT<T>::H;
  using CF = HaloCfg<VB, NT>;
  constexpr int NPL = CF::NPL, BN = CF::BN, KC = CF::KC, NSTG = CF::NSTG;
  constexpr int NW = CF::NW, NTHR = CF::NTHR, RT = CF::RT;
  constexpr int NCH = H_HV * NPL, J = (NCH + NTHR - 1) / NTHR;

  __shared__ __attribute__((aligned(16))) char smem[CF::TOTAL];
  char* const sW = smem;
  char* const sH = smem + CF::WBYTES;
  float* const sStats = reinterpret_cast<float*>(sH + 2 * CF::HALO);
  int* const sDelta = reinterpret_cast<int*>(sH + 2 * CF::HALO + NW * 2 * BN * 4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wx = wave & 3, wh = wave >> 2;      // x plane of the tile, which RT of its 4 (y-pair) row tiles
  const int r16 = lane & 15, q4 = lane >> 4;
  const ctseg_conv_class& K = P.cls[0];
  const int col0 = blockIdx.y * BN;
  const int kpad = K.kpad;

  // ---- weights -> LDS (once per workgroup), tap offsets -> halo index deltas ---------------------------------
  for (int idx = tid; idx < BN * NSTG * 8; idx += NTHR) {
    const int q8 = idx & 7, row = (idx >> 3) % BN, s = idx / (8 * BN);
    const u32x4 v = *reinterpret_cast<const u32x4*>(P.w + (K.w_off + (int64_t)(col0 + row) * kpad) * SZ + s * 128 + q8 * 16);
    *reinterpret_cast<u32x4*>(sW + (s * BN + row) * 128 + ((q8 ^ ((row >> 1) & 7)) << 4)) = v;
  }
  if (tid < 32) {
    int d = 0;
    if (tid < K.ntaps) {
      const int tp = K.taps[tid];
      d = ((int)(int8_t)(tp & 0xff) * H_HY + (int)(int8_t)((tp >> 8) & 0xff)) * H_HZ + (int)(int8_t)((tp >> 16) & 0xff);
    }
    sDelta[tid] = d * 16;
  }

  // ---- per-thread constants of the J staging slots (only the tile base changes from tile to tile) -----------------
  const int YZ = P.Yi * P.Zi;
  int g_byte[J], g_hxyz[J], g_lds[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int idx = tid + j * NTHR;
    const int pl = (idx >> 3) % NPL, hv = (idx / (8 * NPL)) * 8 + (idx & 7);
    const int hx = hv / (H_HY * H_HZ), rem = hv - hx * (H_HY * H_HZ);
    const int hy = rem / H_HZ, hz = rem - hy * H_HZ;
    g_byte[j] = (((hx - 1) * YZ + (hy - 1) * P.Zi + (hz - 1)) * P.g_ld + pl * EPC) * SZ;
    g_hxyz[j] = (idx < NCH) ? (hx | (hy << 8) | (hz << 16)) : 0x7f7f7f;   // sentinel fails every bounds test
    g_lds[j] = pl * H_PLANE + hv * 16;
  }
  const bool ld12 = VB == 32 && SZ == 2 && (P.g_ld & 7) != 0;   // workgroup-uniform
  const int tiles_per_sample = P.tiles;
  auto tile_origin = [&](int t, int& n, int& x0, int& y0, int& z0) {
    n = t / tiles_per_sample;
    int r = t - n * tiles_per_sample;
    const int tz = r % tzn; r /= tzn;
    const int ty = r % tyn; const int tx = r / tyn;
    x0 = tx * H_TX; y0 = ty * H_TY; z0 = tz * H_TZ;
  };
  auto gload = [&](int t, u32x4 (&rh)[J]) {
    int n, x0, y0, z0;
    tile_origin(t, n, x0, y0, z0);
    const char* base = P.in + ((((int64_t)n * P.Xi + x0) * P.Yi + y0) * P.Zi + z0) * P.g_ld * SZ;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int xi = x0 - 1 + (g_hxyz[j] & 0xff), yi = y0 - 1 + ((g_hxyz[j] >> 8) & 0xff), zi = z0 - 1 + (g_hxyz[j] >> 16);
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi) {
        if (!ld12) {
          v = *reinterpret_cast<const u32x4*>(base + g_byte[j]);
        } else {
          v = load_row12_chunk(base + g_byte[j], g_lds[j] >= H_PLANE);   // 24-byte rows: chunk 1 is channels 8..11 + zeros
        }
      }
      rh[j] = v;
    }
  };
  auto sstore = [&](int buf, const u32x4 (&rh)[J]) {
    char* h = sH + buf * CF::HALO;
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (J * NTHR == NCH || tid + j * NTHR < NCH) *reinterpret_cast<u32x4*>(h + g_lds[j]) = rh[j];
  };

  // ---- per-lane constants of the MFMA operands and of the epilogue ---------------------------------------------------
  // row-tile i of wave w covers x = w, y in {2i, 2i+1}, z 0..7 (permuted inside the 2x8 patch, see header)
  int pdy, pz;
  patch_voxel(r16, pdy, pz);
  int abase[RT], ovox[RT], yrow[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    yrow[i] = 2 * (wh * RT + i) + pdy;
    abase[i] = ((((wx + 1) * H_HY) + (yrow[i] + 1)) * H_HZ + (pz + 1)) * 16;
    ovox[i] = (wx * P.Yo + yrow[i]) * P.Zo + pz;       // voxel offset from the tile's first voxel
  }
  const int aplane = (VB == 64 ? q4 : (q4 & 1)) * H_PLANE;
  const int wrow = r16 * 128, wswz = (r16 >> 1) & 7;
  const bool of32 = P.out_f32 != 0, af32 = (P.add_f32 != 0) || SZ == 4;
  const int OSZ = (of32 || SZ == 4) ? 4 : 2, ASZ = af32 ? 4 : 2;
  float bias[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = col0 + j * 16 + 4 * q4 + e;
      bias[j][e] = (P.bias != nullptr && ch < P.Cn) ? P.bias[ch] : 0.f;
    }

  // InstanceNorm partial sums are kept in registers across the workgroup's tiles and written once per (workgroup, sample):
  // one partial slot per workgroup instead of one per tile (no per-tile barrier, 30x fewer partials to finalize)
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
          const int c = j * 16 + 4 * q4 + e;
          sStats[(wave * 2 + 0) * BN + c] = a;
          sStats[(wave * 2 + 1) * BN + c] = b;
        }
        wsum[j][e] = 0.f;
        wsq[j][e] = 0.f;
      }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid % BN;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += sStats[(w * 2 + which) * BN + c];
      const int64_t slot_t = (int64_t)n * P.stats_tiles + P.stats_tile0 + blockIdx.x;
      P.stats[(slot_t * 2 + which) * P.stats_ld + col0 + c] = a;
    }
    __syncthreads();
  };
  auto compute_tile = [&](int t, int buf) {
    int n, x0, y0, z0;
    tile_origin(t, n, x0, y0, z0);
    if (STATS && n != stat_n) {
      if (stat_n >= 0) flush_stats(stat_n);
      stat_n = n;
    }
    f32x4 acc[NT][RT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < RT; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* h = sH + buf * CF::HALO + aplane;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      int delta;
      if constexpr (VB == 64) {
        delta = sDelta[c];
      } else {
        const int tap = 2 * c + (q4 >> 1);
        delta = sDelta[tap < 27 ? tap : 0];
      }
      u32x4 xf[RT], wf[NT];
#pragma unroll
      for (int i = 0; i < RT; ++i) xf[i] = *reinterpret_cast<const u32x4*>(h + abase[i] + delta);
      const char* wb = sW + ((c >> 1) * BN) * 128 + wrow + (((4 * (c & 1) + q4) ^ wswz) << 4);
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * 128);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < RT; ++i) mma16<T>(acc[j][i], wf[j], xf[i]);
    }
    // ---- epilogue straight from the accumulators: lane = (voxel of row-tile i, channels j*16 + 4*q4 .. +3) --------
    const int64_t vb = (((int64_t)n * P.Xo + x0) * P.Yo + y0) * P.Zo + z0;
    const bool xok = x0 + wx < P.Xr, zok = z0 + pz < P.Zr;
    bool rv[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) rv[i] = xok && zok && (y0 + yrow[i] < P.Yr);
    u32x4 av[NT][RT];
    if constexpr (ADDC) {
      const char* hc = sH + buf * CF::HALO;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int cb = (j * 16 + 4 * q4) * SZ;          // byte offset of the lane's 4 channels inside the voxel
          const char* ap = hc + (cb >> 4) * H_PLANE + abase[i] + (cb & 15);
          u32x4 v = {0u, 0u, 0u, 0u};
          if (SZ == 4) v = *reinterpret_cast<const u32x4*>(ap);
          else { const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap); v[0] = w2[0]; v[1] = w2[1]; }
          av[j][i] = v;
        }
    } else if (P.add != nullptr) {
      const char* ab = P.add + vb * P.add_ld * ASZ;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int ch = col0 + j * 16 + 4 * q4;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (rv[i] && ch < P.Cn_store) {
            const char* ap = ab + ((int64_t)ovox[i] * P.add_ld + ch) * ASZ;
            if (af32) v = *reinterpret_cast<const u32x4*>(ap);
            else { const u32x2 w2 = *reinterpret_cast<const u32x2*>(ap); v[0] = w2[0]; v[1] = w2[1]; }
          }
          av[j][i] = v;
        }
    }
    char* ob = P.out + vb * P.o_ld * OSZ;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = col0 + j * 16 + 4 * q4;
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[j][i][e] + bias[j][e];
          if (STATS && rv[i]) { wsum[j][e] += v[e]; wsq[j][e] += v[e] * v[e]; }
        }
        if (P.add != nullptr) {
          if (af32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(av[j][i][e]);
          } else {
            v[0] += h2f<H>(av[j][i][0] & 0xffffu); v[1] += h2f<H>(av[j][i][0] >> 16);
            v[2] += h2f<H>(av[j][i][1] & 0xffffu); v[3] += h2f<H>(av[j][i][1] >> 16);
          }
        }
        if (rv[i] && ch < P.Cn_store) {
          char* op = ob + ((int64_t)ovox[i] * P.o_ld + ch) * OSZ;
          if (OSZ == 4) *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
          else *reinterpret_cast<u32x2*>(op) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
        }
      }
    }
  };

  // tile sequence of this workgroup: each XCD (blockIdx % 8 shares one) owns a contiguous range of tiles, its workgroups
  // walk it round-robin, so the ~64 tiles in flight on one XCD are neighbours and share halos through that XCD's L2
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
  u32x4 r0[J], r1[J];
  int ta = first, tb = first + stride;
  if (ta < last) {
    gload(ta, r0);
    sstore(0, r0);
  }
  if (tb < last) gload(tb, r1);
  __syncthreads();
  while (ta < last) {
    const int tc = tb + stride;       // tile after tb
    if (tc < last) gload(tc, r0);
    compute_tile(ta, 0);
    if (tb < last) sstore(1, r1);
    __syncthreads();                  // halo[1] complete; every wave is done reading halo[0] and the stats slots
    if (tb >= last) break;
    const int td = tc + stride;
    if (td < last) gload(td, r1);
    compute_tile(tb, 1);
    if (tc < last) sstore(0, r0);
    __syncthreads();
    ta = tc;
    tb = td;
  }
  if (STATS && stat_n >= 0) flush_stats(stat_n);
}

bool conv_halo_eligible(const ConvKArgs& a, int dtype, int nclass) {
  const int SZ = dtype == CTSEG_F32 ? 4 : 2;
  const int vb = a.Cg * SZ;
  if (nclass != 1 || a.cls[0].ntaps != 27 || a.sin != 1 || a.sout != 1) return false;
  if (!(vb == 32 || vb == 64) || a.Cn > 32) return false;
  // gathered rows: 16-byte chunked, or 12 bf16 wide (24 bytes) for the 32-byte-voxel kernel
  if (((a.g_ld * SZ) % 16 != 0 && !(vb == 32 && SZ == 2 && a.g_ld == 12)) || ((uintptr_t)a.in % 16) != 0) return false;
  if (a.Xr != a.Xi || a.Yr != a.Yi || a.Zr != a.Zi || a.Zr < 4) return false;
  if ((int64_t)a.Xi * a.Yi * a.Zi * a.g_ld * SZ >= (1ll << 31)) return false;   // per-sample byte offsets are 32-bit
  for (int j = 0; j < 27; ++j) {
    const int tp = a.cls[0].taps[j];
    for (int s = 0; s < 24; s += 8) {
      const int d = (int)(int8_t)((tp >> s) & 0xff);
      if (d < -1 || d > 1) return false;
    }
  }
  return true;
}

int conv_halo_tiles(const ConvKArgs& a) {
  return ((a.Xr + H_TX - 1) / H_TX) * ((a.Yr + H_TY - 1) / H_TY) * ((a.Zr + H_TZ - 1) / H_TZ);
}

template <int VB, int NT> static int halo_grid(int total) {
  const int per_cu = HaloCfg<VB, NT>::TOTAL > 80 * 1024 ? 1 : (HaloCfg<VB, NT>::TOTAL > 54000 ? 2 : 3);   // 160 KiB LDS per CU
  return persistent_grid(CTSEG_NUM_CU * per_cu, total);
}

// workgroups (= InstanceNorm partial slots per sample) a launch with this geometry uses
int conv_halo_slots(const ConvKArgs& a, int dtype) {
  if (conv_halo_x_eligible(a, dtype, 1)) return conv_halo_x_slots(a);
  const int vb = a.Cg * (dtype == CTSEG_F32 ? 4 : 2), total = conv_halo_tiles(a) * a.N;
  if (vb == 64) return a.Cn > 16 ? halo_grid<64, 2>(total) : halo_grid<64, 1>(total);
  return a.Cn > 16 ? halo_grid<32, 2>(total) : halo_grid<32, 1>(total);
}

template <typename T, int VB, int NT> static void launch_halo(ConvKArgs& a, hipStream_t st) {
  const int tyn = (a.Yr + H_TY - 1) / H_TY, tzn = (a.Zr + H_TZ - 1) / H_TZ;
  a.tiles = conv_halo_tiles(a);
  const int total = a.tiles * a.N;
  const int gx = halo_grid<VB, NT>(total);
  dim3 grid((unsigned)gx, (unsigned)((a.Cn + 16 * NT - 1) / (16 * NT)), 1);
  constexpr int SZ = TT<T>::SZ;
  // identity residual: same tensor, same storage type, every stored channel present in the staged voxel, single column block
  const bool addc = a.add == a.in && a.add_ld == a.g_ld && (a.add_f32 != 0) == (SZ == 4) && a.Cn_store * SZ <= VB && grid.y == 1 &&
                    a.stats == nullptr;
  if (a.stats != nullptr) hipLaunchKernelGGL((conv_halo_kernel<T, VB, NT, true, false>), grid, dim3(HaloCfg<VB, NT>::NTHR), 0, st, a, total, tyn, tzn);
  else if (addc) hipLaunchKernelGGL((conv_halo_kernel<T, VB, NT, false, true>), grid, dim3(HaloCfg<VB, NT>::NTHR), 0, st, a, total, tyn, tzn);
  else hipLaunchKernelGGL((conv_halo_kernel<T, VB, NT, false, false>), grid, dim3(HaloCfg<VB, NT>::NTHR), 0, st, a, total, tyn, tzn);
}

void launch_conv_halo(ConvKArgs& a, int dtype, hipStream_t st) {
  if (conv_halo_x_eligible(a, dtype, 1)) { launch_conv_halo_x(a, st); return; }
  const int SZ = dtype == CTSEG_F32 ? 4 : 2;
  const int vb = a.Cg * SZ;
  const bool n2 = a.Cn > 16;
  if (dtype == CTSEG_F32) {
    if (vb == 64) { if (n2) launch_halo<float, 64, 2>(a, st); else launch_halo<float, 64, 1>(a, st); }
    else { if (n2) launch_halo<float, 32, 2>(a, st); else launch_halo<float, 32, 1>(a, st); }
  } else if (dtype == CTSEG_F16) {
    if (vb == 64) { if (n2) launch_halo<F16, 64, 2>(a, st); else launch_halo<F16, 64, 1>(a, st); }
    else { if (n2) launch_halo<F16, 32, 2>(a, st); else launch_halo<F16, 32, 1>(a, st); }
  } else {
    if (vb == 64) { if (n2) launch_halo<BF16, 64, 2>(a, st); else launch_halo<BF16, 64, 1>(a, st); }
    else { if (n2) launch_halo<BF16, 32, 2>(a, st); else launch_halo<BF16, 32, 1>(a, st); }
  }
}

}  // namespace ctseg
