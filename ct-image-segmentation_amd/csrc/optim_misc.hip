// Adam step, weight packing, dtype / layout plumbing and the C-ABI error channel (gfx950).
#include "ctseg_dev.h"

namespace ctseg {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// torch.optim.Adam (amsgrad=False, weight_decay=0) as torch applies it:
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float omb1, float b2, float omb2, float eps,
                                                   float step_size, float bc2_sqrt, float gscale) {
  // omb1 = 1 - beta1, omb2 = 1 - beta2 as torch forms them: in DOUBLE on the host, then rounded (1.f - 0.999f is 1.3e-5 off 0.001)
  const int64_t n4 = n / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      mm[e] = mm[e] + (gr - mm[e]) * omb1;  // torch: exp_avg.lerp_(grad, 1 - beta1)
      vv[e] = vv[e] * b2 + omb2 * gr * gr;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pp[e] = pp[e] - step_size * (mm[e] / denom);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const float gr = g[i] * gscale;
    const float mm = m[i] + (gr - m[i]) * omb1;
    const float vv = v[i] * b2 + omb2 * gr * gr;
    m[i] = mm; v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

template <typename TD> __global__ void gather_cast_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           TD* __restrict__ dst, int64_t n);
template <typename TS, typename TD> __device__ __forceinline__ TD cvt(TS x);
template <> __device__ __forceinline__ float cvt<float, float>(float x) { return x; }
template <> __device__ __forceinline__ unsigned short cvt<float, unsigned short>(float x) { return (unsigned short)f2bf(x); }
template <> __device__ __forceinline__ float cvt<unsigned short, float>(unsigned short x) { return bf2f(x); }
template <> __device__ __forceinline__ unsigned short cvt<unsigned short, unsigned short>(unsigned short x) { return x; }
// IEEE half storage (CTSEG_F16) travels as _Float16 here so that it is a distinct type from the raw-ushort bf16
template <> __device__ __forceinline__ _Float16 cvt<float, _Float16>(float x) {
  return __builtin_bit_cast(_Float16, (unsigned short)f2h<F16>(x));
}
template <> __device__ __forceinline__ float cvt<_Float16, float>(_Float16 x) { return (float)x; }

template <typename TD> __global__ void gather_cast_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                           TD* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = cvt<float, TD>(src[idx[i]]);
}

template <typename TS, typename TD> __global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    d[i] = cvt<TS, TD>(s[i]);
}

// structured weight re-layout (ctseg_pack_weights): one workgroup per (block, row)
template <typename TD> __global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ src,
                                                                                  const ctseg_pack_block* __restrict__ blocks,
                                                                                  const int32_t* __restrict__ rows, TD* __restrict__ dst) {
  __shared__ float s_w[CTSEG_PACK_LDS_FLOATS];     // the row's source region(s): (g_hi - g_lo) * T floats per part (caller-checked)
  const int b = rows[2 * blockIdx.x], n = rows[2 * blockIdx.x + 1];
  const ctseg_pack_block& B = blocks[b];
  const int T = B.T;
  int loff[2] = {0, 0};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (p < B.nparts) {
      const ctseg_pack_part& P = B.part[p];
      if (p == 1) loff[1] = (B.part[0].g_hi - B.part[0].g_lo) * T;
      if (n >= P.n_lo && n < P.n_hi) {
        const int cnt = (P.g_hi - P.g_lo) * T;
        const float* base = src + P.o + (int64_t)(n - P.n_lo) * P.SN;
        // (batches of 8 independent loads: the longest rows — 384 gathered channels x 27 taps — would otherwise pay ~40 dependent
        // round trips through this loop)
        for (int i0 = threadIdx.x; i0 < cnt; i0 += 256 * 8) {
          float r[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            const int g = i / T, t = i - g * T;
            r[u] = i < cnt ? base[(int64_t)g * P.SG + t] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (i0 + u * 256 < cnt) s_w[loff[p] + i0 + u * 256] = r[u];
        }
      }
    }
  }
  __syncthreads();
  TD* drow = dst + B.dst_off + (int64_t)n * B.kpad;
  const int used = B.ntaps * B.gs;
  // eight consecutive K slots per thread: when gs is a multiple of 8 (every pass with 16-byte chunked rows) they share their tap and
  // their gathered channels are consecutive: one division per 16 bytes stored (2-byte stores moved 128 bytes per wave instruction)
  if (B.gs % 8 == 0 && B.kpad % 8 == 0) {
    for (int k0 = threadIdx.x * 8; k0 < B.kpad; k0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if (k0 < used) {
        const int j = k0 / B.gs, g0 = k0 - j * B.gs, tp = B.tap[j];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if (p < B.nparts && n >= B.part[p].n_lo && n < B.part[p].n_hi) {
            const int g_lo = B.part[p].g_lo, g_hi = B.part[p].g_hi;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (g0 + e >= g_lo && g0 + e < g_hi) v[e] = s_w[loff[p] + (g0 + e - g_lo) * T + tp];
          }
      }
      if constexpr (sizeof(TD) == 4) {
        reinterpret_cast<f32x4*>(drow + k0)[0] = f32x4{v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4*>(drow + k0)[1] = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          o[e] = (uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(v[2 * e])) |
                 ((uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(v[2 * e + 1])) << 16);
        *reinterpret_cast<u32x4*>(drow + k0) = o;
      }
    }
    return;
  }
  for (int k = threadIdx.x; k < B.kpad; k += 256) {
    float v = 0.f;
    if (k < used) {
      const int j = k / B.gs, g = k - j * B.gs;
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if (p < B.nparts && n >= B.part[p].n_lo && n < B.part[p].n_hi && g >= B.part[p].g_lo && g < B.part[p].g_hi)
          v = s_w[loff[p] + (g - B.part[p].g_lo) * T + B.tap[j]];
    }
    drow[k] = cvt<float, TD>(v);
  }
}

// fp32 [N][C][S] -> [N][S][ld]; thread per (voxel, channel<ld): reads strided by S, writes contiguous (small C only)
template <typename TD> __global__ void nc_to_cl_kernel(const float* __restrict__ s, TD* __restrict__ d, int C, int64_t S, int ld) {
  const int n = blockIdx.y;
  const int64_t total = S * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / ld;
    const int c = (int)(i - v * ld);
    const float x = c < C ? s[((int64_t)n * C + c) * S + v] : 0.f;
    d[(int64_t)n * total + i] = cvt<float, TD>(x);
  }
}
// C == ld == 1 (the U-Net's one-channel input): a plain cast, 8 elements per thread (two 16-byte loads, one 16-byte store of
// 16-bit storage).  The general kernel above does a 64-bit division and a 2-byte store per element: 95 us for the 100 MB batch,
// on the critical path between two training steps (the next stem convolution waits for it).
template <typename TD> __global__ __launch_bounds__(256) void cast8_kernel(const float* __restrict__ s, TD* __restrict__ d, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 a = reinterpret_cast<const f32x4*>(s)[2 * i], b = reinterpret_cast<const f32x4*>(s)[2 * i + 1];
    if constexpr (sizeof(TD) == 4) {
      reinterpret_cast<f32x4*>(d)[2 * i] = a;
      reinterpret_cast<f32x4*>(d)[2 * i + 1] = b;
    } else {
      u32x4 o;
      o[0] = (uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(a[0])) | ((uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(a[1])) << 16);
      o[1] = (uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(a[2])) | ((uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(a[3])) << 16);
      o[2] = (uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(b[0])) | ((uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(b[1])) << 16);
      o[3] = (uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(b[2])) | ((uint32_t)__builtin_bit_cast(unsigned short, cvt<float, TD>(b[3])) << 16);
      reinterpret_cast<u32x4*>(d)[i] = o;
    }
  }
}
template <typename TS> __global__ void cl_to_nc_kernel(const TS* __restrict__ s, float* __restrict__ d, int C, int64_t S, int ld) {
  const int n = blockIdx.y;
  const int64_t total = S * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / S);
    const int64_t v = i - (int64_t)c * S;
    d[(int64_t)n * total + i] = cvt<TS, float>(s[((int64_t)n * S + v) * ld + c]);
  }
}

// ---- sliding-window inference (semantics of MONAI's sliding_window_inference) ----
// one window per launch: windows of a batch may overlap, stream order makes the accumulation deterministic
template <typename TD>
__global__ void window_gather_kernel(const float* __restrict__ vol, int Cin, int X, int Y, int Z, int x0, int y0, int z0, int rx,
                                     int ry, int rz, float cval, TD* __restrict__ dst, int ld) {
  const int64_t total = (int64_t)rx * ry * rz * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / ld;
    const int c = (int)(i - v * ld);
    const int z = (int)(v % rz), y = (int)((v / rz) % ry), x = (int)(v / ((int64_t)rz * ry));
    const int gx = x0 + x, gy = y0 + y, gz = z0 + z;
    float val = 0.f;
    if (c < Cin) {
      const bool in = (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z;
      val = in ? vol[(((int64_t)c * X + gx) * Y + gy) * Z + gz] : cval;
    }
    dst[i] = cvt<float, TD>(val);
  }
}
// every window of a forward batch in one launch (blockIdx.y = window, origins read from the device)
template <typename TD>
__global__ void window_gather_batch_kernel(const float* __restrict__ vol, int Cin, int X, int Y, int Z, const int* __restrict__ starts,
                                           int rx, int ry, int rz, float cval, TD* __restrict__ dst, int ld) {
  const int w = blockIdx.y;
  const int x0 = starts[3 * w], y0 = starts[3 * w + 1], z0 = starts[3 * w + 2];
  const int64_t total = (int64_t)rx * ry * rz * ld;
  TD* d = dst + (int64_t)w * total;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / ld;
    const int c = (int)(i - v * ld);
    const int z = (int)(v % rz), y = (int)((v / rz) % ry), x = (int)(v / ((int64_t)rz * ry));
    const int gx = x0 + x, gy = y0 + y, gz = z0 + z;
    float val = 0.f;
    if (c < Cin) {
      const bool in = (unsigned)gx < (unsigned)X && (unsigned)gy < (unsigned)Y && (unsigned)gz < (unsigned)Z;
      val = in ? vol[(((int64_t)c * X + gx) * Y + gy) * Z + gz] : cval;
    }
    d[i] = cvt<float, TD>(val);
  }
}
__global__ void window_blend_kernel(const float* __restrict__ logits, int ld, int C, int rx, int ry, int rz, int x0, int y0, int z0,
                                    const float* __restrict__ imp, const float* __restrict__ inv_count, float* __restrict__ out,
                                    int X, int Y, int Z, int out_ld) {
  const int64_t total = (int64_t)rx * ry * rz * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / C;
    const int c = (int)(i - v * C);
    const int z = (int)(v % rz), y = (int)((v / rz) % ry), x = (int)(v / ((int64_t)rz * ry));
    const int gx = x0 + x, gy = y0 + y, gz = z0 + z;
    if ((unsigned)gx >= (unsigned)X || (unsigned)gy >= (unsigned)Y || (unsigned)gz >= (unsigned)Z) continue;   // padded rim
    const int64_t d = ((int64_t)gx * Y + gy) * Z + gz;
    out[d * out_ld + c] += imp[v] * (inv_count ? inv_count[d] : 1.f) * logits[v * ld + c];
  }
}

// All windows of one forward batch in ONE launch, output-centric: a thread owns an output voxel and adds, in window order, the
// weighted logits of every window of the batch that covers it -- the same fp32 sums in the same order as one
// window_blend_kernel launch per window, but the output row is read and written once per batch instead of once per covering
// window, rows move as 16-byte vectors, and there is no per-window launch.
template <int V4>     // float4 groups per row (out_ld = ld = 4 * V4)
__global__ __launch_bounds__(256) void window_blend_batch_kernel(const float* __restrict__ logits, int C, int rx, int ry, int rz,
                                                                 const int* __restrict__ starts, int nw,
                                                                 const float* __restrict__ imp, const float* __restrict__ inv_count,
                                                                 float* __restrict__ out, int X, int Y, int Z, int bx0, int by0,
                                                                 int bz0, int bX, int bY, int bZ) {
  // threads sweep the bounding box (bx0.., extents bX x bY x bZ, inside the volume) of the batch's windows
  const int64_t total = (int64_t)bX * bY * bZ, wvox = (int64_t)rx * ry * rz;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int gz = bz0 + (int)(i % bZ), gy = by0 + (int)((i / bZ) % bY), gx = bx0 + (int)(i / ((int64_t)bZ * bY));
    const int64_t d = ((int64_t)gx * Y + gy) * Z + gz;
    f32x4 acc[V4];
    f32x4* o = reinterpret_cast<f32x4*>(out + d * (4 * V4));
#pragma unroll
    for (int q = 0; q < V4; ++q) acc[q] = o[q];
    const float inv = inv_count ? inv_count[d] : 1.f;
    bool any = false;
    for (int w = 0; w < nw; ++w) {
      const int x = gx - starts[3 * w], y = gy - starts[3 * w + 1], z = gz - starts[3 * w + 2];
      if ((unsigned)x >= (unsigned)rx || (unsigned)y >= (unsigned)ry || (unsigned)z >= (unsigned)rz) continue;
      const int64_t v = ((int64_t)x * ry + y) * rz + z;
      const float wgt = imp[v] * inv;
      const f32x4* l = reinterpret_cast<const f32x4*>(logits + ((int64_t)w * wvox + v) * (4 * V4));
#pragma unroll
      for (int q = 0; q < V4; ++q) {
        const f32x4 t = l[q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * q + e < C) acc[q][e] += wgt * t[e];
      }
      any = true;
    }
    if (any) {
#pragma unroll
      for (int q = 0; q < V4; ++q) o[q] = acc[q];
    }
  }
}

static inline unsigned nblocks(int64_t total, int cap = 4096) {
  int64_t b = (total + 255) / 256;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace ctseg

using namespace ctseg;

extern "C" int ctseg_abi_version(void) { return CTSEG_ABI_VERSION; }
extern "C" const char* ctseg_last_error(void) { return g_err; }

extern "C" int ctseg_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                               double eps, int32_t step, float grad_scale, void* stream) {
  CTSEG_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  CTSEG_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
                "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n / 4 + 1, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(1.0 - beta1),
                     (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt, grad_scale);
  CTSEG_LAUNCH_CHECK("adam_step");
  return 0;
}

// x *= host_scale * (dev_scale ? *dev_scale : 1); every block reads the factor and leaves at once when it is exactly 1 (the usual
// upstream gradient of ``loss.backward()``): the drop-in path pays a ~5 us launch, not a pass over d loss / d logits
template <typename T> __global__ __launch_bounds__(256) void scale_inplace_kernel(T* __restrict__ x, int64_t n, const float* __restrict__ dev_scale,
                                                                                    float host_scale) {
  const float f = host_scale * (dev_scale ? *dev_scale : 1.f);
  if (f == 1.f) return;
  constexpr int V = 16 / sizeof(T);
  const int64_t nv = n / V;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (sizeof(T) == 4) {
      f32x4 v = reinterpret_cast<f32x4*>(x)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= f;
      reinterpret_cast<f32x4*>(x)[i] = v;
    } else {
      u32x4 v = reinterpret_cast<u32x4*>(x)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned lo = (unsigned)f2bf(bf2f(v[e] & 0xffffu) * f), hi = (unsigned)f2bf(bf2f(v[e] >> 16) * f);
        v[e] = (lo & 0xffffu) | (hi << 16);
      }
      reinterpret_cast<u32x4*>(x)[i] = v;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - nv * V)) {
    const int64_t i = nv * V + threadIdx.x;
    if constexpr (sizeof(T) == 4) x[i] *= f;
    else x[i] = (T)f2bf(bf2f(x[i]) * f);
  }
}

extern "C" int ctseg_scale_inplace(void* x, int32_t dtype, int64_t n, const float* dev_scale, float host_scale, void* stream) {
  CTSEG_REQUIRE(x && n > 0 && (dtype == CTSEG_F32 || dtype == CTSEG_BF16) && ((uintptr_t)x % 16) == 0, "scale_inplace: bad arguments");
  dim3 grid(nblocks(n / 8 + 1, 2048)), blk(256);
  if (dtype == CTSEG_F32)
    hipLaunchKernelGGL(scale_inplace_kernel<float>, grid, blk, 0, (hipStream_t)stream, (float*)x, n, dev_scale, host_scale);
  else
    hipLaunchKernelGGL(scale_inplace_kernel<unsigned short>, grid, blk, 0, (hipStream_t)stream, (unsigned short*)x, n, dev_scale, host_scale);
  CTSEG_LAUNCH_CHECK("scale_inplace");
  return 0;
}

extern "C" int ctseg_gather_cast(const float* src, const int32_t* idx, void* dst, int32_t dtype, int64_t n, void* stream) {
  CTSEG_REQUIRE(src && idx && dst && n > 0 && (dtype == CTSEG_F32 || is16(dtype)), "gather_cast: bad arguments");
  if (dtype == CTSEG_F32)
    hipLaunchKernelGGL(gather_cast_kernel<float>, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, src, idx, (float*)dst, n);
  else if (dtype == CTSEG_F16)
    hipLaunchKernelGGL(gather_cast_kernel<_Float16>, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, src, idx, (_Float16*)dst, n);
  else
    hipLaunchKernelGGL(gather_cast_kernel<unsigned short>, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, src, idx,
                       (unsigned short*)dst, n);
  CTSEG_LAUNCH_CHECK("gather_cast");
  return 0;
}

extern "C" int ctseg_cast(const void* src, int32_t sd, void* dst, int32_t dd, int64_t n, void* stream) {
  CTSEG_REQUIRE(src && dst && n > 0, "cast: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nblocks(n)), blk(256);
  if (sd == CTSEG_F32 && dd == CTSEG_BF16)
    hipLaunchKernelGGL((cast_kernel<float, unsigned short>), grid, blk, 0, st, (const float*)src, (unsigned short*)dst, n);
  else if (sd == CTSEG_BF16 && dd == CTSEG_F32)
    hipLaunchKernelGGL((cast_kernel<unsigned short, float>), grid, blk, 0, st, (const unsigned short*)src, (float*)dst, n);
  else if (sd == CTSEG_F32 && dd == CTSEG_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, blk, 0, st, (const float*)src, (float*)dst, n);
  else if (sd == CTSEG_BF16 && dd == CTSEG_BF16)
    hipLaunchKernelGGL((cast_kernel<unsigned short, unsigned short>), grid, blk, 0, st, (const unsigned short*)src,
                       (unsigned short*)dst, n);
  else if (sd == CTSEG_F32 && dd == CTSEG_F16)
    hipLaunchKernelGGL((cast_kernel<float, _Float16>), grid, blk, 0, st, (const float*)src, (_Float16*)dst, n);
  else if (sd == CTSEG_F16 && dd == CTSEG_F32)
    hipLaunchKernelGGL((cast_kernel<_Float16, float>), grid, blk, 0, st, (const _Float16*)src, (float*)dst, n);
  else CTSEG_REQUIRE(false, "cast: bad dtypes %d -> %d", sd, dd);
  CTSEG_LAUNCH_CHECK("cast");
  return 0;
}

extern "C" int ctseg_nc_to_cl(const float* src, void* dst, int32_t dtype, int32_t N, int32_t C, int64_t S, int32_t ld,
                              void* stream) {
  CTSEG_REQUIRE(src && dst && N > 0 && C > 0 && S > 0 && ld >= C, "nc_to_cl: bad arguments");
  if (C == 1 && ld == 1 && ((int64_t)N * S) % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0) {
    const int64_t n8 = (int64_t)N * S / 8;
    const dim3 g8(nblocks(n8 * 256 / 256, 8192));
    if (dtype == CTSEG_F32) hipLaunchKernelGGL(cast8_kernel<float>, g8, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, n8);
    else if (dtype == CTSEG_F16) hipLaunchKernelGGL(cast8_kernel<_Float16>, g8, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, n8);
    else hipLaunchKernelGGL(cast8_kernel<unsigned short>, g8, dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n8);
    CTSEG_LAUNCH_CHECK("nc_to_cl");
    return 0;
  }
  dim3 grid(nblocks(S * ld), N);
  if (dtype == CTSEG_F32) hipLaunchKernelGGL(nc_to_cl_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, C, S, ld);
  else if (dtype == CTSEG_F16) hipLaunchKernelGGL(nc_to_cl_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, C, S, ld);
  else hipLaunchKernelGGL(nc_to_cl_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, C, S, ld);
  CTSEG_LAUNCH_CHECK("nc_to_cl");
  return 0;
}

extern "C" int ctseg_pack_weights(const float* src, const ctseg_pack_block* blocks, int32_t n_blocks, const int32_t* rows, int32_t n_rows,
                                  void* dst, int32_t dtype, void* stream) {
  CTSEG_REQUIRE(src && blocks && rows && dst && n_blocks > 0 && n_rows > 0, "pack_weights: bad arguments");
  CTSEG_REQUIRE(dtype == CTSEG_F32 || is16(dtype), "pack_weights: bad dtype %d", dtype);
  const dim3 grid((unsigned)n_rows);
  if (dtype == CTSEG_F32) hipLaunchKernelGGL(pack_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, blocks, rows, (float*)dst);
  else if (dtype == CTSEG_F16) hipLaunchKernelGGL(pack_weights_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, src, blocks, rows, (_Float16*)dst);
  else hipLaunchKernelGGL(pack_weights_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, src, blocks, rows, (unsigned short*)dst);
  CTSEG_LAUNCH_CHECK("pack_weights");
  return 0;
}

extern "C" int ctseg_cl_to_nc(const void* src, int32_t dtype, float* dst, int32_t N, int32_t C, int64_t S, int32_t ld,
                              void* stream) {
  CTSEG_REQUIRE(src && dst && N > 0 && C > 0 && S > 0 && ld >= C, "cl_to_nc: bad arguments");
  dim3 grid(nblocks(S * C), N);
  if (dtype == CTSEG_F32) hipLaunchKernelGGL(cl_to_nc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, C, S, ld);
  else if (dtype == CTSEG_F16) hipLaunchKernelGGL(cl_to_nc_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst, C, S, ld);
  else hipLaunchKernelGGL(cl_to_nc_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src, dst, C, S, ld);
  CTSEG_LAUNCH_CHECK("cl_to_nc");
  return 0;
}

extern "C" int ctseg_window_gather(const float* vol, int32_t Cin, int32_t X, int32_t Y, int32_t Z, int32_t x0, int32_t y0, int32_t z0,
                                   int32_t rx, int32_t ry, int32_t rz, float cval, void* dst, int32_t dtype, int32_t ld, void* stream) {
  CTSEG_REQUIRE(vol && dst && Cin > 0 && ld >= Cin && X > 0 && Y > 0 && Z > 0 && rx > 0 && ry > 0 && rz > 0, "window_gather: bad arguments");
  dim3 grid(nblocks((int64_t)rx * ry * rz * ld));
  if (dtype == CTSEG_F32)
    hipLaunchKernelGGL(window_gather_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cin, X, Y, Z, x0, y0, z0, rx, ry, rz, cval, (float*)dst, ld);
  else if (dtype == CTSEG_F16)
    hipLaunchKernelGGL(window_gather_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cin, X, Y, Z, x0, y0, z0, rx, ry, rz, cval,
                       (_Float16*)dst, ld);
  else
    hipLaunchKernelGGL(window_gather_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cin, X, Y, Z, x0, y0, z0, rx, ry, rz, cval,
                       (unsigned short*)dst, ld);
  CTSEG_LAUNCH_CHECK("window_gather");
  return 0;
}

extern "C" int ctseg_window_gather_batch(const float* vol, int32_t Cin, int32_t X, int32_t Y, int32_t Z, const int32_t* starts, int32_t nw,
                                         int32_t rx, int32_t ry, int32_t rz, float cval, void* dst, int32_t dtype, int32_t ld,
                                         void* stream) {
  CTSEG_REQUIRE(vol && dst && starts && nw > 0 && nw < 65536 && Cin > 0 && ld >= Cin && X > 0 && Y > 0 && Z > 0 && rx > 0 && ry > 0 &&
                    rz > 0, "window_gather_batch: bad arguments");
  dim3 grid(nblocks((int64_t)rx * ry * rz * ld, 1024), (unsigned)nw);
  if (dtype == CTSEG_F32)
    hipLaunchKernelGGL(window_gather_batch_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cin, X, Y, Z, starts, rx, ry, rz,
                       cval, (float*)dst, ld);
  else if (dtype == CTSEG_F16)
    hipLaunchKernelGGL(window_gather_batch_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cin, X, Y, Z, starts, rx,
                       ry, rz, cval, (_Float16*)dst, ld);
  else
    hipLaunchKernelGGL(window_gather_batch_kernel<unsigned short>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cin, X, Y, Z, starts, rx,
                       ry, rz, cval, (unsigned short*)dst, ld);
  CTSEG_LAUNCH_CHECK("window_gather_batch");
  return 0;
}

extern "C" int ctseg_window_blend(const float* logits, int32_t ld, int32_t C, int32_t rx, int32_t ry, int32_t rz, int32_t x0, int32_t y0,
                                  int32_t z0, const float* importance, const float* inv_count, float* out, int32_t X, int32_t Y, int32_t Z,
                                  int32_t out_ld, void* stream) {
  CTSEG_REQUIRE(logits && importance && out && C > 0 && ld >= C && out_ld >= C && X > 0 && Y > 0 && Z > 0 && rx > 0 && ry > 0 &&
                    rz > 0, "window_blend: bad arguments");
  hipLaunchKernelGGL(window_blend_kernel, dim3(nblocks((int64_t)rx * ry * rz * C)), dim3(256), 0, (hipStream_t)stream, logits, ld, C, rx, ry, rz,
                     x0, y0, z0, importance, inv_count, out, X, Y, Z, out_ld);
  CTSEG_LAUNCH_CHECK("window_blend");
  return 0;
}

extern "C" int ctseg_window_blend_batch(const float* logits, int32_t ld, int32_t C, int32_t rx, int32_t ry, int32_t rz,
                                        const int32_t* starts, int32_t nw, const float* importance, const float* inv_count,
                                        float* out, int32_t X, int32_t Y, int32_t Z, int32_t out_ld, const int32_t* bbox,
                                        void* stream) {
  CTSEG_REQUIRE(logits && starts && importance && out && C > 0 && nw > 0 && X > 0 && Y > 0 && Z > 0 && rx > 0 && ry > 0 && rz > 0,
                "window_blend_batch: bad arguments");
  int b[6] = {0, 0, 0, X, Y, Z};
  if (bbox != nullptr) {
    for (int k = 0; k < 6; ++k) b[k] = bbox[k];
    CTSEG_REQUIRE(b[0] >= 0 && b[1] >= 0 && b[2] >= 0 && b[3] > 0 && b[4] > 0 && b[5] > 0 && b[0] + b[3] <= X && b[1] + b[4] <= Y &&
                      b[2] + b[5] <= Z, "window_blend_batch: bounding box outside the volume");
  }
  CTSEG_REQUIRE(ld == out_ld && ld % 4 == 0 && ld >= C && ld <= 16 && ((uintptr_t)logits % 16) == 0 && ((uintptr_t)out % 16) == 0,
                "window_blend_batch: rows must be 16-byte vectors with ld == out_ld <= 16 (got %d / %d)", ld, out_ld);
  const unsigned nb = nblocks((int64_t)b[3] * b[4] * b[5], 16384);
  hipStream_t st = (hipStream_t)stream;
#define CTSEG_WBB(V) hipLaunchKernelGGL(window_blend_batch_kernel<V>, dim3(nb), dim3(256), 0, st, logits, C, rx, ry, rz, starts, nw, \
                                        importance, inv_count, out, X, Y, Z, b[0], b[1], b[2], b[3], b[4], b[5])
  switch (ld / 4) {
    case 1: CTSEG_WBB(1); break;
    case 2: CTSEG_WBB(2); break;
    case 3: CTSEG_WBB(3); break;
    default: CTSEG_WBB(4); break;
  }
#undef CTSEG_WBB
  CTSEG_LAUNCH_CHECK("window_blend_batch");
  return 0;
}
