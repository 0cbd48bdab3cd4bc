// Loss / metric kernels for the reference's 3-D training step (gfx950, HBM-bound, one voxel per lane):
//   ctseg_squash_masks  : _squash_masks_3D (capstone/volumetric/utils.py:4-7)
//   ctseg_seg_loss      : F.cross_entropy [+ class weights] (capstone/models/losses.py:45-68), softmax -> argmax
//                         (capstone/training/utils.py:19-20), the integer counts behind compute_meandice
//                         (capstone/models/temp.py:173-214), soft-Dice / focal sums (monai DiceLoss, FocalLoss as
//                         configured at capstone/volumetric/losses.py:72-77,107) and d(loss)/d(logits).
// The reference materialises softmax, two 1 GB one-hots and their product; here one pass reads the logits
// once, keeps everything in registers and reduces with wave shuffles -> per-workgroup fp64 partials
// (summed later in fixed order: deterministic) and exact int64 counts.
#include "ctseg_dev.h"

namespace ctseg {

constexpr int CMAX = 16;

__global__ __launch_bounds__(256) void squash_masks_kernel(const uint8_t* __restrict__ masks, int K, int64_t S,
                                                           uint8_t* __restrict__ labels, int64_t* __restrict__ labels_i64,
                                                           unsigned long long* __restrict__ hist) {
  __shared__ unsigned int s_h[32];
  const int b = blockIdx.y;
  if (threadIdx.x < 32) s_h[threadIdx.x] = 0;
  __syncthreads();
  const uint8_t* mb = masks + (int64_t)b * K * S;
  int bg = 0;
  const bool vec = (S % 16 == 0) && (((uintptr_t)masks % 16) == 0) && (((uintptr_t)labels % 16) == 0);
  const int64_t nchunk = (S + 15) / 16;
  for (int64_t ch = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; ch < nchunk; ch += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v0 = ch * 16;
    int lab[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) lab[i] = 0;
    if (vec) {
      for (int k = 0; k < K; ++k) {
        const u32x4 m = *reinterpret_cast<const u32x4*>(mb + (int64_t)k * S + v0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int val = (int)((m[i >> 2] >> (8 * (i & 3))) & 0xffu) * (k + 1);
          lab[i] = val > lab[i] ? val : lab[i];
        }
      }
      u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        o[w] = (uint32_t)(lab[4 * w] & 0xff) | ((uint32_t)(lab[4 * w + 1] & 0xff) << 8) | ((uint32_t)(lab[4 * w + 2] & 0xff) << 16) |
               ((uint32_t)(lab[4 * w + 3] & 0xff) << 24);
      *reinterpret_cast<u32x4*>(labels + (int64_t)b * S + v0) = o;
    } else {
      for (int i = 0; i < 16; ++i) {
        if (v0 + i >= S) break;
        int l = 0;
        for (int k = 0; k < K; ++k) {
          const int val = (int)mb[(int64_t)k * S + v0 + i] * (k + 1);
          l = val > l ? val : l;
        }
        lab[i] = l;
        labels[(int64_t)b * S + v0 + i] = (uint8_t)l;
      }
    }
    // histogram: background is ~99 % of a CT volume, and 64 lanes adding to ONE LDS word serialise — count it per thread
    int nbg = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (v0 + i < S) {
        if (labels_i64 != nullptr) labels_i64[(int64_t)b * S + v0 + i] = lab[i];
        if (lab[i] == 0) ++nbg;
        else if (lab[i] <= K) atomicAdd(&s_h[lab[i]], 1u);
      }
    }
    bg += nbg;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bg += __shfl_xor(bg, o, 64);
  if ((threadIdx.x & 63) == 0 && bg) atomicAdd(&s_h[0], (unsigned)bg);
  __syncthreads();
  if (threadIdx.x <= K && hist != nullptr && s_h[threadIdx.x] != 0)
    atomicAdd(&hist[(int64_t)b * (K + 1) + threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}

// One pass over fp32 channels-last logits.  Block (p, b) covers voxels [p*vp, (p+1)*vp) of sample b.
// SOFT = false: cross-entropy-only fast path (the reference's 3-D default): no soft-Dice / focal sums, CE-only gradient
// CP = extent of the per-class register arrays (12 when C <= 12 — the reference's 10 classes — else 16): fewer live registers,
// more waves per SIMD for this latency-bound streaming pass
template <typename GT, bool SOFT, int CP>
__global__ __launch_bounds__(256) void seg_loss_kernel(const float* __restrict__ logits, int ld, const uint8_t* __restrict__ labels,
                                                       int64_t S, int C, const float* __restrict__ class_weight, int do_stats,
                                                       double* __restrict__ part, int P, unsigned long long* __restrict__ cnt,
                                                       int do_grad, const float* __restrict__ coef, char* __restrict__ dlogits,
                                                       int g_ld, uint8_t* __restrict__ pred_out) {
  constexpr int GSZ = TT<GT>::SZ, GEPC = TT<GT>::EPC;
  __shared__ float s_coef[1 + 3 * CMAX];
  __shared__ float s_cw[CMAX];
  __shared__ double s_part[4][2 + 3 * CMAX];
  __shared__ unsigned int s_cnt[3 * CMAX];
  const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < CMAX) s_cw[tid] = (class_weight != nullptr && tid < C) ? class_weight[tid] : 1.f;
  if (tid < 1 + 3 * CMAX) s_coef[tid] = 0.f;
  if (tid < 3 * CMAX) s_cnt[tid] = 0u;
  __syncthreads();
  if (do_grad && tid < 1 + 3 * C) {
    // coef[b] = (ce_scale, a[C], b[C], f[C]) -> padded to CMAX per table
    const float v = coef[(int64_t)b * (1 + 3 * C) + tid];
    if (tid == 0) s_coef[0] = v;
    else { const int t = (tid - 1) / C, c = (tid - 1) % C; s_coef[1 + t * CMAX + c] = v; }
  }
  __syncthreads();
  const int64_t vp = (S + P - 1) / P;
  const int64_t v0 = p * vp, v1 = (v0 + vp < S) ? v0 + vp : S;
  const int nld4 = ld / 4;

  float a_ce = 0.f, a_w = 0.f;
  float a_p[CP], a_py[CP], a_fo[CP];
  unsigned int c_in[CP], c_pr[CP], c_tr[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) { a_p[c] = a_py[c] = a_fo[c] = 0.f; c_in[c] = c_pr[c] = c_tr[c] = 0u; }

  for (int64_t v = v0 + tid; v < v1; v += 256) {
    const int64_t vox = (int64_t)b * S + v;
    float x[CP];
    const f32x4* lp = reinterpret_cast<const f32x4*>(logits + vox * ld);
#pragma unroll
    for (int q = 0; q < CP / 4; ++q) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
      if (q < nld4) t = lp[q];
      x[4 * q] = t[0]; x[4 * q + 1] = t[1]; x[4 * q + 2] = t[2]; x[4 * q + 3] = t[3];
    }
    const int t = (int)labels[vox];
    float m = x[0];
#pragma unroll
    for (int c = 1; c < CP; ++c) if (c < C) m = fmaxf(m, x[c]);
    float e[CP], ssum = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) { e[c] = (c < C) ? expf(x[c] - m) : 0.f; if (c < C) ssum += e[c]; }
    // softmax THEN argmax, first maximal index (capstone/training/utils.py:19-20)
    float pr[CP], best = -1.f;
    int pred = 0;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      pr[c] = (c < C) ? e[c] / ssum : 0.f;
      if (c < C && pr[c] > best) { best = pr[c]; pred = c; }
    }
    const float lse = m + logf(ssum);
    float xt = 0.f, pt = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) if (c == t) { xt = x[c]; pt = pr[c]; }
    const float logpt = xt - lse;
    const float w = s_cw[t < CMAX ? t : 0];
    if (pred_out != nullptr) pred_out[vox] = (uint8_t)pred;
    if (do_stats) {
      a_ce += w * (lse - xt);
      a_w += w;
      const float om = 1.f - pt;
      const float fo = -om * om * logpt;
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        if (c < C) {
          if (SOFT) a_p[c] += pr[c];
          if (c == t) { if (SOFT) { a_py[c] += pr[c]; a_fo[c] += fo; } c_tr[c] += 1u; }
          if (c == pred) { c_pr[c] += 1u; if (c == t) c_in[c] += 1u; }
        }
      }
    }
    if (do_grad) {
      const float ce_scale = s_coef[0] * w;
      // soft-Dice: dL/dp_c = a_c*[c==t] + b_c ; through softmax: p_k (g_k - sum_j g_j p_j)
      float d[CMAX];
#pragma unroll
      for (int c = CP; c < CMAX; ++c) d[c] = 0.f;
      if constexpr (SOFT) {
        float gk[CP], dot = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          gk[c] = (c < C) ? (s_coef[1 + CMAX + c] + (c == t ? s_coef[1 + c] : 0.f)) : 0.f;
          dot += gk[c] * pr[c];
        }
        const float ft = s_coef[1 + 2 * CMAX + (t < CMAX ? t : 0)];
        const float om = 1.f - pt;
        const float fterm = ft * (2.f * om * pt * logpt - om * om);
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float ind = (c == t) ? 1.f : 0.f;
          d[c] = (c < C) ? (ce_scale * (pr[c] - ind) + pr[c] * (gk[c] - dot) + fterm * (ind - pr[c])) : 0.f;
        }
      } else {
#pragma unroll
        for (int c = 0; c < CP; ++c) d[c] = (c < C) ? ce_scale * (pr[c] - ((c == t) ? 1.f : 0.f)) : 0.f;
      }
      char* gp = dlogits + vox * g_ld * GSZ;
      if (GSZ == 2 && (g_ld & 7) != 0) {
        // bf16 rows 12 wide (24 bytes, 8-byte aligned): 8-byte pieces
        if constexpr (GSZ == 2) {
#pragma unroll
          for (int u = 0; u < CP / 4; ++u)
            if (u * 4 < g_ld) store_ep<GT, 4>(gp + u * 8, d + u * 4);
        }
      } else {
#pragma unroll
        for (int q = 0; q < CMAX / GEPC; ++q)
          if (q * GEPC < g_ld) store_chunk<GT>(gp + q * 16, d + q * GEPC);
      }
    }
  }

  if (do_stats) {
    // wave shuffle reduction (fp64) -> LDS -> one partial record per workgroup
    double rec[2 + 3 * CMAX];
    rec[0] = a_ce; rec[1] = a_w;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      rec[2 + c] = c < CP ? a_p[c < CP ? c : 0] : 0.f;
      rec[2 + CMAX + c] = c < CP ? a_py[c < CP ? c : 0] : 0.f;
      rec[2 + 2 * CMAX + c] = c < CP ? a_fo[c < CP ? c : 0] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2 + 3 * CMAX; ++i) {
      double s = 0.0;
      if (SOFT || i < 2) s = wave_sum(rec[i]);
      if (lane == 0) s_part[wave][i] = s;
    }
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      unsigned int a = c_in[c], bq = c_pr[c], cq = c_tr[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); bq += __shfl_xor(bq, o, 64); cq += __shfl_xor(cq, o, 64); }
      if (lane == 0 && c < C) {
        if (a) atomicAdd(&s_cnt[c], a);
        if (bq) atomicAdd(&s_cnt[CMAX + c], bq);
        if (cq) atomicAdd(&s_cnt[2 * CMAX + c], cq);
      }
    }
    __syncthreads();
    const int R = 2 + 3 * C;
    if (tid < R) {
      int src = tid;
      if (tid >= 2) { const int t = (tid - 2) / C, c = (tid - 2) % C; src = 2 + t * CMAX + c; }
      const double s = s_part[0][src] + s_part[1][src] + s_part[2][src] + s_part[3][src];
      part[((int64_t)b * P + p) * R + tid] = s;
    }
    if (tid < 3 * C) {
      const int t = tid / C, c = tid % C;
      const unsigned int v = s_cnt[t * CMAX + c];
      if (v) atomicAdd(&cnt[((int64_t)b * 3 + t) * C + c], (unsigned long long)v);
    }
  }
}

// Dice counts from two label maps (DiceMetricWrapper called on already-squashed predictions)
__global__ __launch_bounds__(256) void dice_counts_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ truth,
                                                          int64_t S, int C, unsigned long long* __restrict__ cnt) {
  __shared__ unsigned int s_cnt[3 * CMAX];
  const int b = blockIdx.y;
  if (threadIdx.x < 3 * CMAX) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < S; v += (int64_t)gridDim.x * blockDim.x) {
    const int p = pred[(int64_t)b * S + v], t = truth[(int64_t)b * S + v];
    if (p < C) atomicAdd(&s_cnt[CMAX + p], 1u);
    if (t < C) atomicAdd(&s_cnt[2 * CMAX + t], 1u);
    if (p == t && p < C) atomicAdd(&s_cnt[p], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 3 * C) {
    const int k = threadIdx.x / C, c = threadIdx.x % C;
    const unsigned int v = s_cnt[k * CMAX + c];
    if (v) atomicAdd(&cnt[((int64_t)b * 3 + k) * C + c], (unsigned long long)v);
  }
}

// part [B][P][R] -> out [B][R]; one block per (record entry r, b): 256 strided sub-sums over P, combined in fixed order
__global__ __launch_bounds__(256) void reduce_partials_f64_kernel(const double* __restrict__ part, int P, int R,
                                                                  double* __restrict__ out) {
  __shared__ double s[256];
  const int r = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  double a = 0.0;
  for (int p = t; p < P; p += 256) a += part[((int64_t)b * P + p) * R + r];
  s[t] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {          // fixed pairing => deterministic
    if (t < w) s[t] += s[t + w];
    __syncthreads();
  }
  if (t == 0) out[(int64_t)b * R + r] = s[0];
}

// The scalar bookkeeping of a cross-entropy training step in ONE launch instead of ~20 scalar-sized torch kernels:
// out[0] = sum_b red[b][0] / sum_b red[b][1] (mean cross-entropy), out[2 + k] = Dice of foreground class k+1 averaged over the
// samples whose ground truth holds it ("mean_batch" with NaN for an empty truth, capstone/models/metrics.py:15-31), 0 when
// no sample does; out[1] = their mean over the C-1 foreground classes.  fp32 arithmetic in the order of the torch expressions.
__global__ __launch_bounds__(64) void loss_dice_summary_kernel(const double* __restrict__ red, int B, int R,
                                                                 const long long* __restrict__ cnt, int C, float* __restrict__ out) {
  __shared__ float s_pc[CMAX];
  const int t = threadIdx.x;
  if (t < C - 1) {
    const int c = t + 1;
    float sum = 0.f, n_ok = 0.f;
    for (int b = 0; b < B; ++b) {
      const float inter = (float)cnt[((int64_t)b * 3 + 0) * C + c], pred = (float)cnt[((int64_t)b * 3 + 1) * C + c],
                  truth = (float)cnt[((int64_t)b * 3 + 2) * C + c];
      if (truth > 0.f) { sum += 2.0f * inter / (truth + pred); n_ok += 1.f; }
    }
    const float pc = n_ok > 0.f ? sum / n_ok : 0.f;
    s_pc[t] = pc;
    out[2 + t] = pc;
  }
  __syncthreads();
  if (t == 0) {
    double a = 0.0, w = 0.0;
    for (int b = 0; b < B; ++b) { a += red[(int64_t)b * R]; w += red[(int64_t)b * R + 1]; }
    out[0] = (float)(a / w);
    float m = 0.f;
    for (int k = 0; k < C - 1; ++k) m += s_pc[k];
    out[1] = m / (float)(C - 1);
  }
}

}  // namespace ctseg

using namespace ctseg;

extern "C" int ctseg_loss_dice_summary(const double* red, int32_t B, int32_t R, const int64_t* cnt, int32_t C, float* out,
                                       void* stream) {
  CTSEG_REQUIRE(red && cnt && out && B > 0 && R >= 2 && C >= 2 && C <= CMAX, "loss_dice_summary: bad arguments");
  hipLaunchKernelGGL(loss_dice_summary_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, red, B, R, (const long long*)cnt, C, out);
  CTSEG_LAUNCH_CHECK("loss_dice_summary");
  return 0;
}

extern "C" int ctseg_squash_masks(const uint8_t* masks, int32_t B, int32_t K, int64_t S, uint8_t* labels, int64_t* labels_i64,
                                  int64_t* hist, void* stream) {
  CTSEG_REQUIRE(masks && labels && B > 0 && K > 0 && K < 32 && S > 0, "squash_masks: bad arguments");
  const int64_t nchunk = (S + 15) / 16;
  int64_t blocks = (nchunk + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(squash_masks_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, masks, K, S, labels,
                     labels_i64, (unsigned long long*)hist);
  CTSEG_LAUNCH_CHECK("squash_masks");
  return 0;
}


// ---- 3-D input pipeline: nearest Resize3D + (D,H,W)->(H,W,D) + optional window + optional squash ----
// One workgroup = one h, a 64-wide w tile and a 32-deep d tile.  Phase 1 reads the source along w (its contiguous axis), phase 2
// writes the target along d (its contiguous axis); the transposition goes through LDS.  Index rule = torch's nearest:
// src = min((int)floorf(dst * ((float)in / out)), in - 1).
constexpr int RZ_TW = 64, RZ_TD = 32, RZ_KMAX = 15;
__device__ __forceinline__ int rz_src(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}
template <typename TI>
__global__ __launch_bounds__(256) void resize3d_hwd_kernel(const TI* __restrict__ image, const uint8_t* __restrict__ masks, int K, int D, int H,
                                                           int W, int Do, int Ho, int Wo, float sD, float sH, float sW, int window,
                                                           float win_lo, float win_hi, float* __restrict__ image_out,
                                                           uint8_t* __restrict__ masks_out, uint8_t* __restrict__ labels_out,
                                                           unsigned long long* __restrict__ hist) {
  __shared__ float s_img[RZ_TD][RZ_TW + 1];
  __shared__ uint8_t s_lab[RZ_TD][RZ_TW + 4];
  __shared__ unsigned short s_bits[RZ_TD][RZ_TW + 2];      // bit k = mask k at this voxel (masks_out only)
  __shared__ unsigned int s_h[RZ_KMAX + 1];
  const int t = threadIdx.x;
  const int ndt = (Do + RZ_TD - 1) / RZ_TD;
  const int d0 = (blockIdx.x % ndt) * RZ_TD, w0 = (blockIdx.x / ndt) * RZ_TW, h = blockIdx.y;
  if (t <= RZ_KMAX) s_h[t] = 0;
  __syncthreads();
  const int sh = rz_src(h, sH, H);
  {
    const int ww = t & 63, w = w0 + ww;
    const int sw = rz_src(w < Wo ? w : Wo - 1, sW, W);
    for (int r = 0; r < RZ_TD / 4; ++r) {
      const int dd = r * 4 + (t >> 6), d = d0 + dd;
      if (w >= Wo || d >= Do) continue;
      const int64_t src = ((int64_t)rz_src(d, sD, D) * H + sh) * W + sw;
      if (image) {
        float v = (float)image[src];
        if (window) {
          v = fminf(fmaxf(v, win_lo), win_hi);
          if (window == 2) v = (v - win_lo) / (float)((double)win_hi - (double)win_lo + 1e-8);   // numpy float32 arithmetic of apply_window
        }
        s_img[dd][ww] = v;
      }
      if (masks) {
        int lab = 0;
        unsigned bits = 0;
        for (int k = 0; k < K; ++k) {
          const int m = masks[(int64_t)k * D * H * W + src];
          bits |= (m ? 1u : 0u) << k;
          const int val = m * (k + 1);
          lab = val > lab ? val : lab;
        }
        s_lab[dd][ww] = (uint8_t)lab;
        s_bits[dd][ww] = (unsigned short)bits;
        if (hist) atomicAdd(&s_h[lab & RZ_KMAX], 1u);
      }
    }
  }
  __syncthreads();
  {
    const int dd = t & 31, d = d0 + dd;
    for (int r = 0; r < RZ_TW / 8; ++r) {
      const int ww = r * 8 + (t >> 5), w = w0 + ww;
      if (w >= Wo || d >= Do) continue;
      const int64_t dst = ((int64_t)h * Wo + w) * Do + d;
      if (image) image_out[dst] = s_img[dd][ww];
      if (masks) {
        if (labels_out) labels_out[dst] = s_lab[dd][ww];
        if (masks_out) {
          const unsigned bits = s_bits[dd][ww];
          for (int k = 0; k < K; ++k) masks_out[(int64_t)k * Ho * Wo * Do + dst] = (uint8_t)((bits >> k) & 1u);
        }
      }
    }
  }
  __syncthreads();
  if (hist && t <= K && s_h[t]) atomicAdd(&hist[t], (unsigned long long)s_h[t]);
}

extern "C" int ctseg_seg_loss(const float* logits, int32_t ld, const uint8_t* labels, int32_t B, int64_t S, int32_t C,
                              const float* class_weight, int32_t do_stats, double* part, int32_t P, int64_t* cnt,
                              int32_t do_grad, const float* coef, void* dlogits, int32_t g_ld, int32_t gdtype, uint8_t* pred_out,
                              void* stream) {
  CTSEG_REQUIRE(logits && labels && B > 0 && S > 0 && C >= 2 && C <= CMAX, "seg_loss: bad arguments (C <= 16)");
  CTSEG_REQUIRE(ld % 4 == 0 && ld >= C && ld <= CMAX && ((uintptr_t)logits % 16) == 0, "seg_loss: logits stride %d", ld);
  CTSEG_REQUIRE(P > 0 && (!do_stats || (part && cnt)), "seg_loss: stats buffers");
  if (do_grad) {
    CTSEG_REQUIRE(coef && dlogits && (gdtype == CTSEG_F32 || gdtype == CTSEG_BF16), "seg_loss: grad buffers");
    // bf16 gradients: 16-byte chunked rows, or 12 wide (8-byte pieces) for the <= 12 class case
    CTSEG_REQUIRE(g_ld % 4 == 0 && (gdtype == CTSEG_F32 || g_ld % 8 == 0 || g_ld == 12) && g_ld >= C && g_ld <= CMAX &&
                      ((uintptr_t)dlogits % 16) == 0,
                  "seg_loss: dlogits stride %d", g_ld);
  }
  hipStream_t st = (hipStream_t)stream;
  const bool lite = (do_stats == 2 || do_stats == 0) && (do_grad == 2 || do_grad == 0);   // cross-entropy only
  const bool gbf = do_grad && gdtype == CTSEG_BF16;
#define CTSEG_LOSS_LAUNCH2(GT, SOFT, CP)                                                                                         \
  hipLaunchKernelGGL((seg_loss_kernel<GT, SOFT, CP>), dim3(P, B), dim3(256), 0, st, logits, ld, labels, S, C, class_weight, do_stats, \
                     part, P, (unsigned long long*)cnt, do_grad, coef, (char*)dlogits, g_ld, pred_out)
#define CTSEG_LOSS_LAUNCH(GT, SOFT)                                       \
  do {                                                                    \
    if (C <= 12 && ld <= 12) CTSEG_LOSS_LAUNCH2(GT, SOFT, 12);            \
    else CTSEG_LOSS_LAUNCH2(GT, SOFT, 16);                                \
  } while (0)
  if (gbf) { if (lite) CTSEG_LOSS_LAUNCH(BF16, false); else CTSEG_LOSS_LAUNCH(BF16, true); }
  else { if (lite) CTSEG_LOSS_LAUNCH(float, false); else CTSEG_LOSS_LAUNCH(float, true); }
#undef CTSEG_LOSS_LAUNCH2
#undef CTSEG_LOSS_LAUNCH
  CTSEG_LAUNCH_CHECK("seg_loss");
  return 0;
}

extern "C" int ctseg_reduce_partials_f64(const double* part, int32_t B, int32_t P, int32_t R, double* out, void* stream) {
  CTSEG_REQUIRE(part && out && B > 0 && P > 0 && R > 0 && R <= 64, "reduce_partials_f64: bad arguments");
  hipLaunchKernelGGL(reduce_partials_f64_kernel, dim3(R, B), dim3(256), 0, (hipStream_t)stream, part, P, R, out);
  CTSEG_LAUNCH_CHECK("reduce_partials_f64");
  return 0;
}

extern "C" int ctseg_dice_counts(const uint8_t* pred, const uint8_t* truth, int32_t B, int64_t S, int32_t C, int64_t* cnt,
                                 void* stream) {
  CTSEG_REQUIRE(pred && truth && cnt && B > 0 && S > 0 && C >= 2 && C <= CMAX, "dice_counts: bad arguments");
  int64_t blocks = (S + 256 * 16 - 1) / (256 * 16);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(dice_counts_kernel, dim3((unsigned)blocks, B), dim3(256), 0, (hipStream_t)stream, pred, truth, S, C,
                     (unsigned long long*)cnt);
  CTSEG_LAUNCH_CHECK("dice_counts");
  return 0;
}

extern "C" int ctseg_resize3d_to_hwd(const void* image, int32_t image_dtype, const uint8_t* masks, int32_t K, int32_t D, int32_t H,
                                     int32_t W, int32_t Do, int32_t Ho, int32_t Wo, int32_t window, float win_lo, float win_hi,
                                     float* image_out, uint8_t* masks_out, uint8_t* labels_out, int64_t* hist, void* stream) {
  CTSEG_REQUIRE(D > 0 && H > 0 && W > 0 && Do > 0 && Ho > 0 && Wo > 0 && (image || masks), "resize3d_to_hwd: bad geometry");
  CTSEG_REQUIRE(!image || image_out, "resize3d_to_hwd: image without image_out");
  CTSEG_REQUIRE(!masks || (K > 0 && K <= RZ_KMAX && (masks_out || labels_out)), "resize3d_to_hwd: masks need K <= 15 and an output");
  CTSEG_REQUIRE(!hist || labels_out, "resize3d_to_hwd: hist needs labels_out");
  CTSEG_REQUIRE(window >= 0 && window <= 2 && (!window || win_hi > win_lo), "resize3d_to_hwd: bad window");
  CTSEG_REQUIRE(image_dtype == CTSEG_F32 || image_dtype == CTSEG_I16 || image_dtype == CTSEG_U8, "resize3d_to_hwd: image dtype %d", image_dtype);
  const float sD = (float)D / (float)Do, sH = (float)H / (float)Ho, sW = (float)W / (float)Wo;
  dim3 grid(((Do + RZ_TD - 1) / RZ_TD) * ((Wo + RZ_TW - 1) / RZ_TW), Ho);
  hipStream_t st = (hipStream_t)stream;
#define CTSEG_RZ_LAUNCH(TI)                                                                                                         \
  hipLaunchKernelGGL(resize3d_hwd_kernel<TI>, grid, dim3(256), 0, st, (const TI*)image, masks, K, D, H, W, Do, Ho, Wo, sD, sH, sW, window, \
                     win_lo, win_hi, image_out, masks_out, labels_out, (unsigned long long*)hist)
  if (image_dtype == CTSEG_F32) CTSEG_RZ_LAUNCH(float);
  else if (image_dtype == CTSEG_I16) CTSEG_RZ_LAUNCH(short);
  else CTSEG_RZ_LAUNCH(uint8_t);
#undef CTSEG_RZ_LAUNCH
  CTSEG_LAUNCH_CHECK("resize3d_to_hwd");
  return 0;
}
