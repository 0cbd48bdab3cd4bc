// InstanceNorm3d(affine=False) + PReLU forward/backward for channels-last activations (gfx950).
// MONAI's Convolution block = conv -> InstanceNorm -> PReLU; the reference reaches it
// through UNet.forward (capstone/volumetric/base_trainer.py:74-78) and autograd.
//
// Statistics arrive as per-tile (sum, sumsq) partials from the conv epilogue and are combined in
// fp64 in a fixed order (deterministic).  All passes are HBM-bound: 16-byte vector accesses,
// per-sample mean/rstd staged in LDS, wave/LDS reductions for the backward sums.
#include "ctseg_dev.h"

namespace ctseg {

constexpr int FIN_GROUPS = 64;

// level 1: partials [N][P][R*ld] fp32 -> scratch [N][64][R*ld] fp64 (group g sums tiles g*F .. g*F+F-1 in order)
// One launch: FIN_GROUPS blocks per sample sum their share of the P partial rows into `scratch` (fp64, fixed order); the block
// that finishes last for a sample (device counter, reset for the next call) combines the groups into mean / rstd.
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float* __restrict__ part, int P, int ld, int col0, int C,
                                                                double count, double eps, double* __restrict__ scratch,
                                                                unsigned int* __restrict__ counter, float* __restrict__ mean_rstd) {
  // gridDim.x = groups in use (<= FIN_GROUPS; few partial rows -> few groups, so the second level has less to walk)
  const int G = gridDim.x;
  // This launch sits between every conv and its norm pass with nothing to overlap it: what counts is the length of its
  // dependent-load chains, so both levels spread their sums over all 256 threads.
  __shared__ int s_last;
  __shared__ double s_sub[256];
  const int g = blockIdx.x, n = blockIdx.y, rowlen = 2 * ld, t = threadIdx.x;
  const int F = (P + G - 1) / G;
  const int p0 = g * F, p1 = (p0 + F < P) ? p0 + F : P;
  // level 1: nj columns x R row lanes; lane r sums rows p0 + r, p0 + r + R, ...; the R sub-sums combine in lane order
  const int nj = rowlen < 256 ? rowlen : 256, R = 256 / nj;
  const int r = t / nj, j0 = t - r * nj;
  for (int jb = 0; jb < rowlen; jb += nj) {
    const int j = jb + j0;
    double s = 0.0;
    if (r < R && j < rowlen)
      for (int p = p0 + r; p < p1; p += R) s += (double)part[((int64_t)n * P + p) * rowlen + j];
    if (R > 1) {
      s_sub[t] = s;
      __syncthreads();
      if (r == 0 && j < rowlen) {
        for (int k = 1; k < R; ++k) s += s_sub[k * nj + j0];
      }
      __syncthreads();
    }
    if (r == 0 && j < rowlen) scratch[((int64_t)n * FIN_GROUPS + g) * rowlen + j] = s;
  }
  __threadfence();
  __syncthreads();
  if (t == 0) s_last = (atomicAdd(&counter[n], 1u) == (unsigned)(G - 1));
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // level 2 (last block of the sample): L lanes per channel, lane l sums groups l, l + L, ...; xor butterfly over the L lanes
  int L = 1;
  while (L * 2 <= G && L * 2 * C <= 256) L *= 2;
  const int per = 256 / L;                       // channels per sweep
  for (int cb = 0; cb < C; cb += per) {
    const int c = cb + t / L, l = t & (L - 1);
    double sm = 0.0, q = 0.0;
    if (c < C)
      for (int gg = l; gg < G; gg += L) {
        sm += __builtin_nontemporal_load(&scratch[((int64_t)n * FIN_GROUPS + gg) * rowlen + col0 + c]);
        q += __builtin_nontemporal_load(&scratch[((int64_t)n * FIN_GROUPS + gg) * rowlen + ld + col0 + c]);
      }
    for (int o = L >> 1; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); q += __shfl_xor(q, o, 64); }
    if (c < C && l == 0) {
      const double mean = sm / count;
      double var = q / count - mean * mean;
      if (var < 0.0) var = 0.0;
      mean_rstd[((int64_t)n * C + c) * 2] = (float)mean;
      mean_rstd[((int64_t)n * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + eps));
    }
  }
  if (t == 0) counter[n] = 0u;
}

// Single-level finalize for P <= 8192 partial rows (the per-workgroup slots of the halo kernels: <= 512; the per-tile slots of the
// generic kernel: a few thousand): one block per (sample, 8 channels) — 16 columns (8 sums, 8 sums of squares) x RL row lanes; lane r
// adds rows r, r + RL, ... in fp64, the RL sub-sums combine in lane order: fixed order, no atomics, and no cross-block hand-off.  The two-level kernel above publishes its group sums
// with an agent-scope release per block (buffer_wbl2: an L2 write-back walk, serialised per XCD): 64 groups x N samples of them cost
// 21 us at N = 2 and 262 us at the 48 windows of a sliding-window forward, for 12 MB of partials.
template <int RL>      // row lanes: 16 (256 threads) up to 1024 partial rows, 64 (1024 threads) up to 8192
__global__ __launch_bounds__(16 * RL) void instnorm_finalize1_kernel(const float* __restrict__ part, int P, int ld, int col0, int C, double count,
                                                                     double eps, float* __restrict__ mean_rstd) {
  __shared__ double s_sub[RL][17];
  const int n = blockIdx.y, c0 = blockIdx.x * 8, t = threadIdx.x;
  const int col = t & 15, r = t >> 4;                 // col < 8: sum of channel c0 + col; col >= 8: sum of squares of channel c0 + col - 8
  const int c = c0 + (col & 7);
  double s = 0.0;
  if (c < C) {
    const float* p = part + (int64_t)n * P * 2 * ld + (col >> 3) * ld + col0 + c;
#pragma unroll 4
    for (int row = r; row < P; row += RL) s += (double)p[(int64_t)row * 2 * ld];
  }
  s_sub[r][col] = s;
  __syncthreads();
  if (t < 8 && c0 + t < C) {
    double sm = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < RL; ++k) { sm += s_sub[k][t]; q += s_sub[k][8 + t]; }
    const double mean = sm / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[((int64_t)n * C + c0 + t) * 2] = (float)mean;
    mean_rstd[((int64_t)n * C + c0 + t) * 2 + 1] = (float)(1.0 / sqrt(var + eps));
  }
}

template <typename T, int EPC>
__global__ __launch_bounds__(256) void instnorm_prelu_fwd_kernel(const char* __restrict__ y, int y_ld,
                                                                  const float* __restrict__ mean_rstd,
                                                                  const float* __restrict__ alpha, const char* __restrict__ res,
                                                                  int res_ld, char* __restrict__ out, int out_ld, int64_t S,
                                                                  int C, int Cv) {
  constexpr int SZ = TT<T>::SZ;   // EPC: elements per chunk (a row is Cv chunks of EPC elements)
  extern __shared__ float s_mr[];  // per chunk [EPC][2] + 1 pad word (consecutive lanes read consecutive chunks: conflict-free)
  constexpr int TB = 2 * EPC + 1;
  const int n = blockIdx.y;
  if (mean_rstd != nullptr)
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_mr[((i >> 1) / EPC) * TB + ((i >> 1) % EPC) * 2 + (i & 1)] = mean_rstd[(int64_t)n * C * 2 + i];
  __syncthreads();
  const float al = alpha != nullptr ? alpha[0] : 1.f;
  const int64_t total = S * Cv;
  auto body = [&](int64_t v, int cv) {
    const int64_t vox = (int64_t)n * S + v;
    float x[EPC], r[EPC];
    load_ep<T, EPC>(y + (vox * y_ld + cv * EPC) * SZ, x);
    if (res != nullptr) load_ep<T, EPC>(res + (vox * res_ld + cv * EPC) * SZ, r);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = cv * EPC + e;
      float o = 0.f;
      if (c < C) {
        o = x[e];
        if (mean_rstd != nullptr) {
          o = (o - s_mr[cv * TB + 2 * e]) * s_mr[cv * TB + 2 * e + 1];
          o = o > 0.f ? o : al * o;
        }
        if (res != nullptr) o += r[e];
      }
      x[e] = o;
    }
    store_ep<T, EPC>(out + (vox * out_ld + cv * EPC) * SZ, x);
  };
  // a thread keeps its channel chunk when the grid stride is a multiple of Cv (the launcher arranges it): no 64-bit
  // division per element, and the per-channel constants stay in registers
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (stride % Cv == 0) {
    int64_t v = i0 / Cv;
    const int cv = (int)(i0 - v * Cv);
    const int64_t vstep = stride / Cv;
    for (; v < S; v += vstep) body(v, cv);
  } else {
    for (int64_t i = i0; i < total; i += stride) {
      const int64_t v = i / Cv;
      body(v, (int)(i - v * Cv));
    }
  }
}

// backward pass 1: per block (p, n): rows [p*rows_per, ...) of sample n -> partials[n][p][3][ld]
template <typename T, int EPC>
__global__ __launch_bounds__(256) void instnorm_prelu_bwd_reduce_kernel(const char* __restrict__ g, int g_ld,
                                                                         const char* __restrict__ y, int y_ld,
                                                                         const float* __restrict__ mean_rstd,
                                                                         const float* __restrict__ alpha,
                                                                         float* __restrict__ partials, int P, int ld, int64_t S,
                                                                         int C, int Cv) {
  constexpr int SZ = TT<T>::SZ;   // EPC: elements per chunk (a row is Cv chunks of EPC elements)
  // dynamic LDS: [2*C] mean/rstd, then the cross-thread reduction scratch -- 4 waves x Cv x 3 x EPC floats when Cv is a power
  // of two <= 64 (wave butterflies first), else one slot per thread.  The footprint matters: this pass shares the CUs with
  // the weight-gradient kernels of the side stream, and at 24 KB per block few of its blocks found room beside them.
  extern __shared__ float s_mr[];
  constexpr int TB = 2 * EPC + 1;       // (mean, rstd) of a chunk's channels + a pad word: conflict-free across consecutive chunks
  float* const s_red = s_mr + Cv * TB;
  const bool pow2 = (Cv & (Cv - 1)) == 0 && Cv <= 64;
  const int p = blockIdx.x, n = blockIdx.y;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) s_mr[((i >> 1) / EPC) * TB + ((i >> 1) % EPC) * 2 + (i & 1)] = mean_rstd[(int64_t)n * C * 2 + i];
  __syncthreads();
  const float al = alpha[0];
  const int64_t rows_per = (S + P - 1) / P;
  const int64_t v0 = p * rows_per, v1 = (v0 + rows_per < S) ? v0 + rows_per : S;
  const int nrow_thr = 256 / Cv;  // threads along rows
  const int cv = threadIdx.x % Cv, rsub = threadIdx.x / Cv;
  float a1[EPC], a2[EPC], a3[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) a1[e] = a2[e] = a3[e] = 0.f;
  if (rsub < nrow_thr) {
    for (int64_t v = v0 + rsub; v < v1; v += nrow_thr) {
      const int64_t vox = (int64_t)n * S + v;
      float gv[EPC], yv[EPC];
      load_ep<T, EPC>(g + (vox * g_ld + cv * EPC) * SZ, gv);
      load_ep<T, EPC>(y + (vox * y_ld + cv * EPC) * SZ, yv);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = cv * EPC + e;
        if (c < C) {
          const float xh = (yv[e] - s_mr[cv * TB + 2 * e]) * s_mr[cv * TB + 2 * e + 1];
          const float dxh = gv[e] * (xh > 0.f ? 1.f : al);
          a1[e] += dxh;
          a2[e] += dxh * xh;
          a3[e] += xh > 0.f ? 0.f : gv[e] * xh;
        }
      }
    }
  }
  if (pow2) {
    // lanes l, l + Cv, l + 2 Cv ... of a wave hold the same channel chunk (64 % Cv == 0): xor butterfly, then 4 waves via LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < EPC; ++e)
      for (int o = 32; o >= Cv; o >>= 1) {
        a1[e] += __shfl_xor(a1[e], o, 64);
        a2[e] += __shfl_xor(a2[e], o, 64);
        a3[e] += __shfl_xor(a3[e], o, 64);
      }
    if (lane < Cv) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        s_red[((wave * Cv + lane) * 3 + 0) * EPC + e] = a1[e];
        s_red[((wave * Cv + lane) * 3 + 1) * EPC + e] = a2[e];
        s_red[((wave * Cv + lane) * 3 + 2) * EPC + e] = a3[e];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
      const int which = i / C, c = i - which * C;
      const int ccv = c / EPC, e = c - ccv * EPC;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += s_red[((w * Cv + ccv) * 3 + which) * EPC + e];
      partials[(((int64_t)n * P + p) * 3 + which) * ld + c] = s;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    s_red[(threadIdx.x * 3 + 0) * EPC + e] = a1[e];
    s_red[(threadIdx.x * 3 + 1) * EPC + e] = a2[e];
    s_red[(threadIdx.x * 3 + 2) * EPC + e] = a3[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
    const int which = i / C, c = i - which * C;
    const int ccv = c / EPC, e = c - ccv * EPC;
    float s = 0.f;
    for (int r = 0; r < nrow_thr; ++r) s += s_red[((r * Cv + ccv) * 3 + which) * EPC + e];
    partials[(((int64_t)n * P + p) * 3 + which) * ld + c] = s;
  }
}

__global__ __launch_bounds__(256) void instnorm_prelu_bwd_finalize_kernel(const float* __restrict__ partials, int P, int ld, int C,
                                                                           double S, float* __restrict__ sums,
                                                                           double* __restrict__ da_part, float* __restrict__ dalpha) {
  // one block per (n, c): 256 strided sub-sums over the P partial rows, combined by a fixed tree (deterministic): wave
  // butterflies, then the four wave sums in order -- a few bytes of LDS, so the block fits on a CU whatever else runs there
  __shared__ double s_acc[3][4];
  const int i = blockIdx.x, n = i / C, c = i - n * C, t = threadIdx.x;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int p = t; p < P; p += 256) {
    const float* q = partials + ((int64_t)n * P + p) * 3 * ld + c;
    s1 += (double)q[0];
    s2 += (double)q[ld];
    s3 += (double)q[2 * ld];
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  if ((t & 63) == 0) { s_acc[0][t >> 6] = s1; s_acc[1][t >> 6] = s2; s_acc[2][t >> 6] = s3; }
  __syncthreads();
  if (t == 0) {
    const double r1 = ((s_acc[0][0] + s_acc[0][1]) + s_acc[0][2]) + s_acc[0][3];
    const double r2 = ((s_acc[1][0] + s_acc[1][1]) + s_acc[1][2]) + s_acc[1][3];
    const double r3 = ((s_acc[2][0] + s_acc[2][1]) + s_acc[2][2]) + s_acc[2][3];
    sums[(int64_t)i * 2] = (float)(r1 / S);
    sums[(int64_t)i * 2 + 1] = (float)(r2 / S);
    da_part[i] = r3;
  }
  if (dalpha == nullptr) return;
  // PReLU slope gradient = fixed-order sum of every block's third partial: done by whichever block finishes last (a counter
  // in the slot behind the gridDim.x partials, reset for the next call) -- no extra launch.  A separate one-block kernel on
  // the side stream measured 360 us when it was dispatched beside the stride-2 halo pass (it waited for that kernel to end).
  __shared__ int s_last;
  const int NC = gridDim.x;
  unsigned int* counter = reinterpret_cast<unsigned int*>(da_part + NC);
  __threadfence();
  __syncthreads();
  if (t == 0) s_last = (atomicAdd(counter, 1u) == (unsigned)(NC - 1));
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  double a = 0.0;
  for (int k = t; k < NC; k += 256) a += __builtin_nontemporal_load(&da_part[k]);
  a = wave_sum(a);
  __syncthreads();
  if ((t & 63) == 0) s_acc[0][t >> 6] = a;
  __syncthreads();
  if (t == 0) { dalpha[0] = (float)(((s_acc[0][0] + s_acc[0][1]) + s_acc[0][2]) + s_acc[0][3]); *counter = 0u; }
}
// PReLU slope gradient = sum over every (n, c) of the third partial, fixed order
__global__ __launch_bounds__(256) void instnorm_prelu_dalpha_kernel(const double* __restrict__ da_part, int NC, float* __restrict__ dalpha) {
  __shared__ double s_da[256];
  const int t = threadIdx.x;
  double a = 0.0;
  for (int i = t; i < NC; i += 256) a += da_part[i];
  s_da[t] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) s_da[t] += s_da[t + w];
    __syncthreads();
  }
  if (t == 0) dalpha[0] = (float)s_da[0];
}

// COLSUM: also emit per-block column sums of dy (fp32, before storage rounding) -> cs_part[(n*gridDim.x + block)][pld]; the
// launcher picks gridDim.x so that a thread's channel chunk is the same in every grid-stride iteration.  This is the bias
// gradient of the ConvTranspose3d that feeds this norm (its dOut = dy), which used to cost a separate pass over dy.
template <typename T, int EPC, bool COLSUM>
__global__ __launch_bounds__(256) void instnorm_prelu_bwd_apply_kernel(const char* __restrict__ g, int g_ld,
                                                                        const char* __restrict__ y, int y_ld,
                                                                        const float* __restrict__ mean_rstd,
                                                                        const float* __restrict__ alpha,
                                                                        const float* __restrict__ sums, char* __restrict__ dy,
                                                                        int dy_ld, char* __restrict__ g_copy, int g_copy_ld,
                                                                        int64_t S, int C, int Cv, float* __restrict__ cs_part,
                                                                        int pld, const double* __restrict__ da_part, int n_da,
                                                                        float* __restrict__ dalpha) {
  constexpr int SZ = TT<T>::SZ;   // EPC: elements per chunk (a row is Cv chunks of EPC elements)
  extern __shared__ float s_tab[];  // per chunk [EPC][4] + 1 pad: mean, rstd, s1, s2  (+ [256][EPC] column-sum scratch when COLSUM)
  float cs[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) cs[e] = 0.f;
  // PReLU slope gradient = fixed-order sum of the finalize pass's per-(n, c) terms: one block of this launch does it on the
  // side (no launch of its own, no atomics -- a same-address counter in the finalize cost ~40 ns per block, 20 us at C = 256)
  if (da_part != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
    __shared__ double s_da[4];
    double a = 0.0;
    for (int k = threadIdx.x; k < n_da; k += 256) a += da_part[k];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) s_da[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) dalpha[0] = (float)(((s_da[0] + s_da[1]) + s_da[2]) + s_da[3]);
  }
  // the reduce pass that precedes this one streamed (g, y) front to back: walk BACKWARDS (last sample first, last voxel
  // first) so the most recently read part of both tensors is re-read while it still sits in L2 / Infinity Cache
  const int n = gridDim.y - 1 - blockIdx.y;
  // table layout: the 4 constants of the EPC channels of chunk cv at cv * (4 * EPC + 1): consecutive lanes read consecutive chunks,
  // and without the pad word their addresses are 4 * EPC floats apart — a 32-way bank conflict on every read at C = 256 (the
  // 25 MB bottom-level pass took 91 us, the 151 MB level-1 pass 47)
  constexpr int TB = 4 * EPC + 1;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    float* t = s_tab + (i / EPC) * TB + (i % EPC) * 4;
    t[0] = mean_rstd[((int64_t)n * C + i) * 2];
    t[1] = mean_rstd[((int64_t)n * C + i) * 2 + 1];
    t[2] = sums[((int64_t)n * C + i) * 2];
    t[3] = sums[((int64_t)n * C + i) * 2 + 1];
  }
  __syncthreads();
  const float al = alpha[0];
  const int64_t total = S * Cv;
  auto body = [&](int64_t v, int cv) {
    const float* tab = s_tab + cv * TB;
    const int64_t vox = (int64_t)n * S + v;
    float gv[EPC], yv[EPC], o[EPC];
    load_ep<T, EPC>(g + (vox * g_ld + cv * EPC) * SZ, gv);
    load_ep<T, EPC>(y + (vox * y_ld + cv * EPC) * SZ, yv);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = cv * EPC + e;
      float r = 0.f;
      if (c < C) r = inorm_prelu_bwd_value(gv[e], yv[e], tab[4 * e], tab[4 * e + 1], tab[4 * e + 2], tab[4 * e + 3], al);
      o[e] = r;
      if (COLSUM) cs[e] += r;
    }
    bool whole_row = false;
    if constexpr (EPC * SZ == 8) whole_row = cv == Cv - 1 && dy_ld == (Cv + 1) * EPC;
    if constexpr (EPC * SZ == 8) {
      if (whole_row) {
      // 8-byte chunks into rows one chunk wider than the channels (12-wide inputs, 16-wide dy): the last chunk and the
      // padding go out as ONE 16-byte store, so every 32-byte row is written whole (no partial sectors at the memory side)
      float o2[2 * EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { o2[e] = o[e]; o2[EPC + e] = 0.f; }
        store_ep<T, 2 * EPC>(dy + (vox * dy_ld + cv * EPC) * SZ, o2);
      }
    }
    if (!whole_row) store_ep<T, EPC>(dy + (vox * dy_ld + cv * EPC) * SZ, o);
    if (g_copy != nullptr) store_ep<T, EPC>(g_copy + (vox * g_copy_ld + cv * EPC) * SZ, gv);
  };
  {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, ir0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (stride % Cv == 0) {       // fixed channel chunk per thread: no 64-bit division per element
      if (ir0 < total) {
        const int64_t i = total - 1 - ir0;
        int64_t v = i / Cv;
        const int cv = (int)(i - v * Cv);
        const int64_t vstep = stride / Cv;
        for (; v >= 0; v -= vstep) body(v, cv);
      }
    } else {
      for (int64_t ir = ir0; ir < total; ir += stride) {
        const int64_t i = total - 1 - ir;
        const int64_t v = i / Cv;
        body(v, (int)(i - v * Cv));
      }
    }
  }
  if constexpr (COLSUM) {
    // (gridDim.x * 256) % Cv == 0: thread t handled the same chunk column in every iteration (the sweep runs backwards from
    // total - 1, total = S * Cv); fixed-order sums of the threads of each column
    float* s_cs = s_tab + Cv * TB;
    const int64_t lead = (int64_t)blockIdx.x * 256;
    if ((Cv & (Cv - 1)) == 0 && Cv <= 64) {
      // power-of-two Cv: lanes l, l + Cv, l + 2 Cv, ... of a wave share the column -> xor butterfly, then 4 waves through LDS
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        for (int o = 32; o >= Cv; o >>= 1) cs[e] += __shfl_xor(cs[e], o, 64);
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      if (lane < Cv) {
        const int cv = (int)((((total - 1 - lead - lane) % Cv) + Cv) % Cv);      // same for lane + 64 w: 64 % Cv == 0
#pragma unroll
        for (int e = 0; e < EPC; ++e) s_cs[(wave * Cv + cv) * EPC + e] = cs[e];
      }
      __syncthreads();
      if ((int)threadIdx.x < Cv * EPC) {
        const float a = s_cs[threadIdx.x] + s_cs[Cv * EPC + threadIdx.x] + s_cs[2 * Cv * EPC + threadIdx.x] +
                        s_cs[3 * Cv * EPC + threadIdx.x];
        cs_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * pld + threadIdx.x] = a;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) s_cs[threadIdx.x * EPC + e] = cs[e];
      __syncthreads();
      if ((int)threadIdx.x < Cv * EPC) {
        const int cv = threadIdx.x / EPC, e = threadIdx.x % EPC;
        float a = 0.f;
        for (int t = 0; t < 256; ++t) {
          const int tcv = (int)((((total - 1 - lead - t) % Cv) + Cv) % Cv);
          if (tcv == cv) a += s_cs[t * EPC + e];
        }
        cs_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * pld + cv * EPC + e] = a;
      }
    }
  }
}

// column sums of a channels-last tensor over all rows (ConvTranspose bias gradient): partials [P][ld] then fixed-order sum
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const char* __restrict__ x, int ld, int64_t rows, int C, int Cv,
                                                             float* __restrict__ partials, int P, int pld) {
  constexpr int SZ = TT<T>::SZ, EPC = TT<T>::EPC;
  __shared__ float s_red[256 * EPC];
  const int p = blockIdx.x;
  const int64_t rows_per = (rows + P - 1) / P;
  const int64_t v0 = p * rows_per, v1 = (v0 + rows_per < rows) ? v0 + rows_per : rows;
  const int nrow_thr = 256 / Cv, cv = threadIdx.x % Cv, rsub = threadIdx.x / Cv;
  float a[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) a[e] = 0.f;
  if (rsub < nrow_thr)
    for (int64_t v = v0 + rsub; v < v1; v += nrow_thr) {
      float xv[EPC];
      load_chunk<T>(x + (v * ld + cv * EPC) * SZ, xv);
#pragma unroll
      for (int e = 0; e < EPC; ++e) a[e] += xv[e];
    }
#pragma unroll
  for (int e = 0; e < EPC; ++e) s_red[threadIdx.x * EPC + e] = a[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < nrow_thr; ++r) s += s_red[(r * Cv + c / EPC) * EPC + (c % EPC)];
    partials[(int64_t)p * pld + c] = s;
  }
}
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partials, int P, int pld, int C,
                                                             float* __restrict__ out) {
  // thread = (sub, c): 1024 / pld strided sub-sums over the P partial rows, combined in fixed order
  __shared__ double s_acc[1024];
  const int lanes = pld, subs = 1024 / lanes;
  const int c = (int)threadIdx.x % lanes, sub = (int)threadIdx.x / lanes;
  double s = 0.0;
  if (sub < subs && c < C)
    for (int p = sub; p < P; p += subs) s += (double)partials[(int64_t)p * pld + c];
  s_acc[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0 && c < C) {
    double t = 0.0;
    for (int k = 0; k < subs; ++k) t += s_acc[k * lanes + c];
    out[c] = (float)t;
  }
}

static inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}
// grid whose stride (blocks * 256) is a multiple of Cv, so that a thread keeps one channel chunk (256 supplies the twos)
static inline int ew_blocks_for(int64_t total, int Cv) {
  int b = ew_blocks(total), step = Cv;
  while (step % 2 == 0) step /= 2;
  if (b >= step) b = b / step * step;
  return b;
}

}  // namespace ctseg

using namespace ctseg;

// EPC_ = elements per chunk the launch works in: 16-byte chunks, or 8-byte ones for bf16 when some tensor's channel stride
// is a multiple of 4 but not of 8 (10 classes stored 12 wide); every stride must then be a multiple of 4
#define CHECK_CL_(dtype, C, ALLOW_HALF, FWD_ONLY_OK, ...)                                                               \
  CTSEG_REQUIRE(dtype == CTSEG_F32 || dtype == CTSEG_BF16 || (FWD_ONLY_OK && dtype == CTSEG_F16),           \
                "bad dtype %d (CTSEG_F16 is accepted by the forward pass only)", dtype);                   \
  int EPC_ = dtype == CTSEG_F32 ? 4 : 8;                                                                   \
  {                                                                                                        \
    const int lds_[] = {__VA_ARGS__};                                                                      \
    if (ALLOW_HALF && is16(dtype))                                                                         \
      for (int ld_ : lds_) if (ld_ % 8 != 0) EPC_ = 4;                                                     \
  }                                                                                                        \
  const int Cv = (C + EPC_ - 1) / EPC_;                                                                    \
  {                                                                                                        \
    const int lds_[] = {__VA_ARGS__};                                                                      \
    for (int ld_ : lds_) CTSEG_REQUIRE(ld_ % EPC_ == 0 && ld_ >= Cv * EPC_, "channel stride %d not chunked for C=%d", ld_, C); \
  }
#define CHECK_CL(dtype, C, ...) CHECK_CL_(dtype, C, false, false, __VA_ARGS__)
#define CHECK_CL_HALF(dtype, C, ...) CHECK_CL_(dtype, C, true, false, __VA_ARGS__)
#define CHECK_CL_HALF_FWD(dtype, C, ...) CHECK_CL_(dtype, C, true, true, __VA_ARGS__)

extern "C" int ctseg_instnorm_finalize(const float* partials, int32_t N, int32_t P, int32_t ld, int32_t col0, int32_t C,
                                       double count, double eps, double* scratch, float* mean_rstd, void* stream) {
  CTSEG_REQUIRE(partials && scratch && mean_rstd && N > 0 && P > 0 && C > 0 && col0 >= 0 && col0 + C <= ld,
                "instnorm_finalize: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  // scratch: N * FIN_GROUPS * 2 * ld doubles of group sums, followed by N zero-initialised counters (one double slot each)
  unsigned int* counter = reinterpret_cast<unsigned int*>(scratch + (int64_t)N * FIN_GROUPS * 2 * ld);
  if (P <= 8192 && getenv("CTSEG_FINALIZE_TWO_LEVEL") == nullptr) {
    if (P <= 1024) hipLaunchKernelGGL(instnorm_finalize1_kernel<16>, dim3((C + 7) / 8, N), dim3(256), 0, st, partials, P, ld, col0, C, count, eps, mean_rstd);
    else hipLaunchKernelGGL(instnorm_finalize1_kernel<64>, dim3((C + 7) / 8, N), dim3(1024), 0, st, partials, P, ld, col0, C, count, eps, mean_rstd);
    CTSEG_LAUNCH_CHECK("instnorm_finalize");
    return 0;
  }
  int groups = P / 8;                      // >= 8 partial rows per first-level group
  groups = groups < 1 ? 1 : (groups > FIN_GROUPS ? FIN_GROUPS : groups);
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(groups, N), dim3(256), 0, st, partials, P, ld, col0, C, count, eps, scratch,
                     counter, mean_rstd);
  CTSEG_LAUNCH_CHECK("instnorm_finalize");
  return 0;
}

extern "C" int ctseg_instnorm_prelu_fwd(int32_t dtype, const void* y, int32_t y_ld, const float* mean_rstd, const float* alpha,
                                        const void* res, int32_t res_ld, void* out, int32_t out_ld, int32_t N, int64_t S,
                                        int32_t C, void* stream) {
  CTSEG_REQUIRE(y && out && N > 0 && S > 0 && C > 0, "instnorm_prelu_fwd: bad arguments");
  CHECK_CL_HALF_FWD(dtype, C, y_ld, out_ld, res ? res_ld : y_ld);
  CTSEG_REQUIRE(mean_rstd == nullptr || alpha != nullptr, "instnorm_prelu_fwd: alpha missing");
  dim3 grid(ew_blocks_for(S * Cv, Cv), N);
  const size_t sh = Cv * (2 * EPC_ + 1) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define CTSEG_FWD(T, EP)                                                                                                      \
  hipLaunchKernelGGL((instnorm_prelu_fwd_kernel<T, EP>), grid, dim3(256), sh, st, (const char*)y, y_ld, mean_rstd, alpha,     \
                     (const char*)res, res_ld, (char*)out, out_ld, S, C, Cv)
  if (dtype == CTSEG_F32) CTSEG_FWD(float, 4);
  else if (dtype == CTSEG_F16) { if (EPC_ == 8) CTSEG_FWD(F16, 8); else CTSEG_FWD(F16, 4); }
  else if (EPC_ == 8) CTSEG_FWD(BF16, 8);
  else CTSEG_FWD(BF16, 4);
#undef CTSEG_FWD
  CTSEG_LAUNCH_CHECK("instnorm_prelu_fwd");
  return 0;
}

extern "C" int ctseg_instnorm_prelu_bwd_reduce(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld,
                                               const float* mean_rstd, const float* alpha, float* partials, int32_t P,
                                               int32_t ld, int32_t N, int64_t S, int32_t C, void* stream) {
  CTSEG_REQUIRE(g && y && mean_rstd && alpha && partials && P > 0 && N > 0 && C <= ld, "instnorm_prelu_bwd_reduce: bad arguments");
  CHECK_CL_HALF(dtype, C, g_ld, y_ld);
  CTSEG_REQUIRE(Cv <= 256, "instnorm_prelu_bwd_reduce: too many channels");
  const bool pow2 = (Cv & (Cv - 1)) == 0 && Cv <= 64;
  // reduction scratch: 4 waves x Cv chunks (butterfly path) or one slot per thread; padding it to the old 24 KB measured
  // 12.45 -> 12.49 ms/step (fewer of these blocks fit beside the side stream's weight-gradient workgroups)
  const size_t sh = (Cv * (2 * EPC_ + 1) + (pow2 ? 4 * Cv * 3 * EPC_ : 256 * 3 * EPC_)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define CTSEG_RED(T, EP)                                                                                                     \
  hipLaunchKernelGGL((instnorm_prelu_bwd_reduce_kernel<T, EP>), dim3(P, N), dim3(256), sh, st, (const char*)g, g_ld,         \
                     (const char*)y, y_ld, mean_rstd, alpha, partials, P, ld, S, C, Cv)
  if (dtype == CTSEG_F32) CTSEG_RED(float, 4);
  else if (EPC_ == 8) CTSEG_RED(BF16, 8);
  else CTSEG_RED(BF16, 4);
#undef CTSEG_RED
  CTSEG_LAUNCH_CHECK("instnorm_prelu_bwd_reduce");
  return 0;
}

extern "C" int ctseg_instnorm_prelu_bwd_finalize(const float* partials, int32_t N, int32_t P, int32_t ld, int32_t C, double S,
                                                 double* scratch, float* sums, float* dalpha, void* stream) {
  CTSEG_REQUIRE(partials && scratch && sums && N > 0 && P > 0 && C > 0, "instnorm_prelu_bwd_finalize: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(instnorm_prelu_bwd_finalize_kernel, dim3(N * C), dim3(256), 0, st, partials, P, ld, C, S, sums, scratch, dalpha);
  CTSEG_LAUNCH_CHECK("instnorm_prelu_bwd_finalize");
  return 0;
}

extern "C" int ctseg_instnorm_prelu_dalpha(const double* scratch, int32_t NC, float* dalpha, void* stream) {
  CTSEG_REQUIRE(scratch && dalpha && NC > 0, "instnorm_prelu_dalpha: bad arguments");
  hipLaunchKernelGGL(instnorm_prelu_dalpha_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, NC, dalpha);
  CTSEG_LAUNCH_CHECK("instnorm_prelu_dalpha");
  return 0;
}

static int bwd_apply_launch(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld, const float* mean_rstd,
                            const float* alpha, const float* sums, void* dy, int32_t dy_ld, void* g_copy, int32_t g_copy_ld, int32_t N,
                            int64_t S, int32_t C, float* cs_part, int32_t P_cap, float* cs_out, const double* da_part, int32_t n_da,
                            float* dalpha, void* stream) {
  CTSEG_REQUIRE(g && y && mean_rstd && alpha && sums && dy && N > 0, "instnorm_prelu_bwd_apply: bad arguments");
  CTSEG_REQUIRE(da_part == nullptr || (dalpha != nullptr && n_da > 0), "instnorm_prelu_bwd_apply: slope-gradient arguments");
  CHECK_CL_HALF(dtype, C, g_ld, y_ld, dy_ld, g_copy ? g_copy_ld : dy_ld);
  int gx = ew_blocks_for(S * Cv, Cv);
  const bool colsum = cs_part != nullptr;
  const int pld = Cv * EPC_;
  if (colsum) {
    CTSEG_REQUIRE(cs_out != nullptr && Cv * EPC_ <= 256 && P_cap >= N, "instnorm_prelu_bwd_apply_colsum: partial buffer");
    if (gx > P_cap / N) gx = P_cap / N;
    // a thread must see the same channel chunk in every grid-stride iteration: (gx * 256) % Cv == 0
    int step = Cv;
    while (step % 2 == 0) step /= 2;          // 256 supplies every factor of two
    gx = gx / step * step;
    CTSEG_REQUIRE(gx >= 1, "instnorm_prelu_bwd_apply_colsum: partial buffer too small for C=%d", C);
  }
  dim3 grid(gx, N);
  const size_t sh = (Cv * (4 * EPC_ + 1) + (colsum ? 256 * EPC_ : 0)) * sizeof(float);       // padded constant table (+ column-sum scratch)
  hipStream_t st = (hipStream_t)stream;
#define CTSEG_APPLY(T, EP, CS)                                                                                                      \
  hipLaunchKernelGGL((instnorm_prelu_bwd_apply_kernel<T, EP, CS>), grid, dim3(256), sh, st, (const char*)g, g_ld, (const char*)y, y_ld, \
                     mean_rstd, alpha, sums, (char*)dy, dy_ld, (char*)g_copy, g_copy_ld, S, C, Cv, cs_part, pld, da_part, n_da, dalpha)
  if (dtype == CTSEG_F32) { if (colsum) CTSEG_APPLY(float, 4, true); else CTSEG_APPLY(float, 4, false); }
  else if (EPC_ == 8) { if (colsum) CTSEG_APPLY(BF16, 8, true); else CTSEG_APPLY(BF16, 8, false); }
  else { if (colsum) CTSEG_APPLY(BF16, 4, true); else CTSEG_APPLY(BF16, 4, false); }
#undef CTSEG_APPLY
  if (colsum) hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(1024), 0, st, cs_part, gx * N, pld, C, cs_out);
  CTSEG_LAUNCH_CHECK("instnorm_prelu_bwd_apply");
  return 0;
}

extern "C" int ctseg_instnorm_prelu_bwd_apply(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld,
                                              const float* mean_rstd, const float* alpha, const float* sums, void* dy,
                                              int32_t dy_ld, void* g_copy, int32_t g_copy_ld, int32_t N, int64_t S, int32_t C,
                                              const double* da_part, int32_t n_da, float* dalpha, void* stream) {
  return bwd_apply_launch(dtype, g, g_ld, y, y_ld, mean_rstd, alpha, sums, dy, dy_ld, g_copy, g_copy_ld, N, S, C, nullptr, 0, nullptr,
                          da_part, n_da, dalpha, stream);
}

extern "C" int ctseg_instnorm_prelu_bwd_apply_colsum(int32_t dtype, const void* g, int32_t g_ld, const void* y, int32_t y_ld,
                                                     const float* mean_rstd, const float* alpha, const float* sums, void* dy,
                                                     int32_t dy_ld, void* g_copy, int32_t g_copy_ld, int32_t N, int64_t S, int32_t C,
                                                     float* colsum_partials, int32_t P_cap, float* colsum_out,
                                                     const double* da_part, int32_t n_da, float* dalpha, void* stream) {
  CTSEG_REQUIRE(colsum_partials != nullptr, "instnorm_prelu_bwd_apply_colsum: partial buffer");
  return bwd_apply_launch(dtype, g, g_ld, y, y_ld, mean_rstd, alpha, sums, dy, dy_ld, g_copy, g_copy_ld, N, S, C, colsum_partials,
                          P_cap, colsum_out, da_part, n_da, dalpha, stream);
}

extern "C" int ctseg_colsum(int32_t dtype, const void* x, int32_t ld, int64_t rows, int32_t C, float* partials, int32_t P,
                            float* out, void* stream) {
  CTSEG_REQUIRE(x && partials && out && rows > 0 && C > 0 && P > 0, "colsum: bad arguments");
  CHECK_CL(dtype, C, ld);
  CTSEG_REQUIRE(Cv <= 256, "colsum: too many channels");
  const int pld = Cv * EPC_;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CTSEG_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(P), dim3(256), 0, st, (const char*)x, ld, rows, C, Cv, partials, P, pld);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<BF16>, dim3(P), dim3(256), 0, st, (const char*)x, ld, rows, C, Cv, partials, P, pld);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(1024), 0, st, partials, P, pld, C, out);
  CTSEG_LAUNCH_CHECK("colsum");
  return 0;
}
