// LDS-halo implicit-GEMM kernel for 3x3x3 stride-1 passes with few channels (gfx950).
//
// The full- and half-resolution layers of the reference's U-Net (32->32 at 256x256x24, the 10->10 logits conv
// at 512x512x48 and their input gradients) are HBM-bound: per output voxel the generic kernel re-gathers 27
// neighbours through the vector-memory path.  Here a persistent workgroup keeps ALL packed weights in LDS, stages
// the (4+2)x(8+2)x(8+2) input halo of a 4x8x8 output tile ONCE (16-byte chunks, zero fill at the border),
// double-buffered against the previous tile's MFMAs, and reads every tap's operand from LDS: 27x reuse.
//
// LDS image of the halo: one plane per 16-byte channel chunk, [plane][600 voxels][16 B].  600*16 B = 8 slots
// (mod 16), and inside each 2x8 (y,z) patch of 16 voxels the lane<->voxel map below is chosen so that every
// ds_read_b128 lane group of an MFMA operand hits 16 distinct 16-byte slots for every tap: conflict free
// (checked exhaustively on the host).  Epilogue = conv_common.h (bias, InstanceNorm partials,
// addend, channels-last stores), so both kernels are interchangeable behind ctseg_conv_igemm.
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
  static constexpr int NPL = VB / 16;
  static constexpr int HALO = NPL * H_PLANE;
  static constexpr int BN = 16 * NT;
  static constexpr int KC = (27 * VB + 63) / 64;   // 64-byte K chunks
  static constexpr int NSTG = (KC + 1) / 2;        // 128-byte weight stages
  static constexpr int WBYTES = NSTG * BN * 128;
  static constexpr int CS = 256 * (BN * 4 + 16);
  static constexpr int BUF = HALO > CS ? HALO : CS;
  static constexpr int TOTAL = WBYTES + 2 * BUF + 4 * 2 * BN * 4 + 256 * 8 + 32 * 4;
};

template <typename T, int VB, int NT>
__global__ __launch_bounds__(256) void conv_halo_kernel(const ConvKArgs P, int total_tiles, int tyn, int tzn) {
  constexpr int SZ = TT<T>::SZ, EPC = TT<T>::EPC;
  using CF = HaloCfg<VB, NT>;
  constexpr int NPL = CF::NPL, BN = CF::BN, KC = CF::KC, NSTG = CF::NSTG;
  constexpr int NCH = H_HV * NPL, J = (NCH + 255) / 256;

  __shared__ __attribute__((aligned(16))) char smem[CF::TOTAL];
  char* const sW = smem;
  char* const sH = smem + CF::WBYTES;
  float* const sStats = reinterpret_cast<float*>(sH + 2 * CF::BUF);
  int* const sRow = reinterpret_cast<int*>(sH + 2 * CF::BUF + 4 * 2 * BN * 4);
  int* const sDelta = sRow + 512;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, q4 = lane >> 4;
  const ctseg_conv_class& K = P.cls[0];
  const int col0 = blockIdx.y * BN;
  const int kpad = K.kpad;

  // ---- weights -> LDS (once per workgroup), tap offsets -> halo index deltas ---------------------------------
  for (int idx = tid; idx < BN * NSTG * 8; idx += 256) {
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

  const int tiles_per_sample = P.tiles;
  // chunk -> (plane, halo voxel) for this thread's J staging slots (same for every tile)
  u32x4 rh[J];
  auto tile_origin = [&](int t, int& n, int& x0, int& y0, int& z0) {
    n = t / tiles_per_sample;
    int r = t - n * tiles_per_sample;
    const int tz = r % tzn; r /= tzn;
    const int ty = r % tyn; const int tx = r / tyn;
    x0 = tx * H_TX; y0 = ty * H_TY; z0 = tz * H_TZ;
  };
  auto gload = [&](int t) {
    int n, x0, y0, z0;
    tile_origin(t, n, x0, y0, z0);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int idx = tid + j * 256;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (idx < NCH) {
        const int pl = (idx >> 3) % NPL, hv = (idx / (8 * NPL)) * 8 + (idx & 7);
        const int hx = hv / (H_HY * H_HZ), rem = hv - hx * (H_HY * H_HZ);
        const int hy = rem / H_HZ, hz = rem - hy * H_HZ;
        const int xi = x0 - 1 + hx, yi = y0 - 1 + hy, zi = z0 - 1 + hz;
        if ((unsigned)xi < (unsigned)P.Xi && (unsigned)yi < (unsigned)P.Yi && (unsigned)zi < (unsigned)P.Zi) {
          const int64_t vox = (((int64_t)n * P.Xi + xi) * P.Yi + yi) * P.Zi + zi;
          v = *reinterpret_cast<const u32x4*>(P.in + (vox * P.g_ld + pl * EPC) * SZ);
        }
      }
      rh[j] = v;
    }
  };
  auto sstore = [&](int buf) {
    char* h = sH + buf * CF::BUF;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int idx = tid + j * 256;
      if (idx < NCH) {
        const int pl = (idx >> 3) % NPL, hv = (idx / (8 * NPL)) * 8 + (idx & 7);
        *reinterpret_cast<u32x4*>(h + pl * H_PLANE + hv * 16) = rh[j];
      }
    }
  };

  // lane's operand addresses: row-tile i of wave w covers x = w, y in {2i, 2i+1}, z 0..7 (permuted, see header)
  int pdy, pz;
  patch_voxel(r16, pdy, pz);
  int abase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) abase[i] = ((((wave + 1) * H_HY) + (2 * i + pdy + 1)) * H_HZ + (pz + 1)) * 16;
  const int aplane = (VB == 64 ? q4 : (q4 & 1)) * H_PLANE;
  const int wrow = r16 * 128, wswz = (r16 >> 1) & 7;

  int t = blockIdx.x;
  int cur = 0;
  if (t < total_tiles) {
    gload(t);
    sstore(0);
  }
  __syncthreads();
  for (; t < total_tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < total_tiles) gload(tn);
    int n, x0, y0, z0;
    tile_origin(t, n, x0, y0, z0);
    {  // row table of this tile (read by the epilogue after the barrier below)
      const int r = tid, w = r >> 6, i = (r >> 4) & 3;
      int dy, z;
      patch_voxel(r & 15, dy, z);
      const int x = x0 + w, y = y0 + 2 * i + dy, zz = z0 + z;
      const bool ok = x < P.Xr && y < P.Yr && zz < P.Zr;
      sRow[2 * r] = x | (y << 16);
      sRow[2 * r + 1] = ok ? zz : -(1 << 24);
    }
    f32x4 acc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* h = sH + cur * CF::BUF + aplane;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      int delta;
      if constexpr (VB == 64) {
        delta = sDelta[c];
      } else {
        const int tap = 2 * c + (q4 >> 1);
        delta = sDelta[tap < 27 ? tap : 0];
      }
      u32x4 xf[4], wf[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const u32x4*>(h + abase[i] + delta);
      const char* wb = sW + ((c >> 1) * BN) * 128 + wrow + (((4 * (c & 1) + q4) ^ wswz) << 4);
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * 128);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) mma16<T>(acc[j][i], wf[j], xf[i]);
    }
    __syncthreads();  // every wave is done with halo[cur]; sRow is visible
    conv_epilogue<T, 256, BN, 4, 1>(P, K, sH + cur * CF::BUF, sStats, sRow, acc, n,
                                    t - n * tiles_per_sample, 0, col0);
    if (tn < total_tiles) sstore(cur ^ 1);
    __syncthreads();  // halo[cur^1] complete, epilogue's LDS reads done
    cur ^= 1;
  }
}

bool conv_halo_eligible(const ConvKArgs& a, int dtype, int nclass) {
  const int SZ = dtype == CTSEG_F32 ? 4 : 2;
  const int vb = a.Cg * SZ;
  if (nclass != 1 || a.cls[0].ntaps != 27 || a.sin != 1 || a.sout != 1) return false;
  if (!(vb == 32 || vb == 64) || a.Cn > 32) return false;
  if ((a.g_ld * SZ) % 16 != 0 || ((uintptr_t)a.in % 16) != 0) return false;
  if (a.Xr != a.Xi || a.Yr != a.Yi || a.Zr != a.Zi || a.Zr < 4) return false;
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

template <typename T, int VB, int NT> static void launch_halo(ConvKArgs& a, hipStream_t st) {
  const int tyn = (a.Yr + H_TY - 1) / H_TY, tzn = (a.Zr + H_TZ - 1) / H_TZ;
  a.tiles = conv_halo_tiles(a);
  const int total = a.tiles * a.N;
  const int per_cu = HaloCfg<VB, NT>::TOTAL > 80 * 1024 ? 1 : (HaloCfg<VB, NT>::TOTAL > 52 * 1024 ? 2 : 3);
  int gx = 256 * per_cu;
  if (gx > total) gx = total;
  dim3 grid((unsigned)gx, (unsigned)((a.Cn + 16 * NT - 1) / (16 * NT)), 1);
  hipLaunchKernelGGL((conv_halo_kernel<T, VB, NT>), grid, dim3(256), 0, st, a, total, tyn, tzn);
}

void launch_conv_halo(ConvKArgs& a, int dtype, hipStream_t st) {
  const int SZ = dtype == CTSEG_F32 ? 4 : 2;
  const int vb = a.Cg * SZ;
  const bool n2 = a.Cn > 16;
  if (dtype == CTSEG_F32) {
    if (vb == 64) { if (n2) launch_halo<float, 64, 2>(a, st); else launch_halo<float, 64, 1>(a, st); }
    else { if (n2) launch_halo<float, 32, 2>(a, st); else launch_halo<float, 32, 1>(a, st); }
  } else {
    if (vb == 64) { if (n2) launch_halo<BF16, 64, 2>(a, st); else launch_halo<BF16, 64, 1>(a, st); }
    else { if (n2) launch_halo<BF16, 32, 2>(a, st); else launch_halo<BF16, 32, 1>(a, st); }
  }
}

}  // namespace ctseg
