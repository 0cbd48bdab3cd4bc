// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes, MFMA 16x16 family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "ctseg_hip.h"

#include <stdlib.h>

namespace ctseg {

// The one target: MI355X (gfx950) = 256 CUs in 8 XCDs, 160 KiB of LDS and 4 SIMDs x 512 VGPRs per CU.  Persistent kernels launch
// CTSEG_NUM_CU * k workgroups (k = workgroups that fit a CU) and walk tiles with stride gridDim.x; XCD-aware orders assume 8 XCDs.
constexpr int CTSEG_NUM_CU = 256;
constexpr int CTSEG_NUM_XCD = 8;

// Grid size of a persistent launch: min(wanted, tiles), optionally capped by the TEST-ONLY environment knob CTSEG_MAX_WG=n so that
// small test shapes make every workgroup walk several tiles (the inter-tile paths — prefetch of tile t+1, store of tile t-1, ring-slot
// rotation across an epilogue, statistics flush at a sample change — are otherwise reached only at 512x512x48).  Every sizing query
// (partial-slot / slab counts) goes through the same function as its launch, so the knob must not change between recording a plan
// and running it.  Unset in production.
static inline int persistent_grid(int wanted, int tiles) {
  int g = wanted < tiles ? wanted : tiles;
  if (const char* e = getenv("CTSEG_MAX_WG")) {
    const int cap = atoi(e);
    if (cap >= 1 && cap < g) g = cap;
  }
  return g < 1 ? 1 : g;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct BF16 {};  // storage tag: 16-bit brain float kept as raw ushort
struct F16 {};   // storage tag: IEEE half kept as raw ushort (forward / inference passes only)

template <typename T> struct TT;
// H = the 16-bit storage kind whose conversions a kernel body names in branches that are dead for fp32 storage
template <> struct TT<float> { static constexpr int SZ = 4, EPC = 4, DT = CTSEG_F32; using H = BF16; };
template <> struct TT<BF16> { static constexpr int SZ = 2, EPC = 8, DT = CTSEG_BF16; using H = BF16; };
template <> struct TT<F16> { static constexpr int SZ = 2, EPC = 8, DT = CTSEG_F16; using H = F16; };

__device__ __forceinline__ float bf2f(uint32_t u16) { return __uint_as_float(u16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

// 16-bit storage kind H (BF16 / F16): element <-> float.  Half stores saturate at +-65504 instead of overflowing to inf (a
// convolution output ahead of its InstanceNorm may be large; its statistics are taken from the fp32 accumulators anyway).
template <typename H> __device__ __forceinline__ float h2f(uint32_t u16);
template <> __device__ __forceinline__ float h2f<BF16>(uint32_t u16) { return bf2f(u16); }
template <> __device__ __forceinline__ float h2f<F16>(uint32_t u16) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)u16);
}
template <typename H> __device__ __forceinline__ uint32_t f2h(float x);
template <> __device__ __forceinline__ uint32_t f2h<BF16>(float x) { return f2bf(x); }
template <> __device__ __forceinline__ uint32_t f2h<F16>(float x) {
  const _Float16 h = (_Float16)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);   // RNE; NaN passes through fmed3 as NaN
  return (uint32_t)__builtin_bit_cast(unsigned short, h);
}
template <typename H> __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return f2h<H>(lo) | (f2h<H>(hi) << 16); }

// Load / store EPC<T> consecutive elements (one 16-byte chunk) as floats.
template <typename T> __device__ __forceinline__ void load_chunk(const char* p, float* v);
template <> __device__ __forceinline__ void load_chunk<float>(const char* p, float* v) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load_chunk<BF16>(const char* p, float* v) {
  u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = bf2f(t[i] & 0xffffu); v[2 * i + 1] = bf2f(t[i] >> 16); }
}
template <> __device__ __forceinline__ void load_chunk<F16>(const char* p, float* v) {
  u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = h2f<F16>(t[i] & 0xffffu); v[2 * i + 1] = h2f<F16>(t[i] >> 16); }
}
template <typename T> __device__ __forceinline__ void store_chunk(char* p, const float* v);
template <> __device__ __forceinline__ void store_chunk<float>(char* p, const float* v) {
  f32x4 t = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(p) = t;
}
template <> __device__ __forceinline__ void store_chunk<BF16>(char* p, const float* v) {
  u32x4 t;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = t;
}

template <> __device__ __forceinline__ void store_chunk<F16>(char* p, const float* v) {
  u32x4 t;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = pack2<F16>(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = t;
}

// EP consecutive elements as floats: EP = EPC<T> (one 16-byte chunk) or, for bf16 rows whose stride is an odd multiple of
// 4 elements (10 classes stored 12 wide: 24-byte rows), EP = 4 (an 8-byte chunk; rows are then only 8-byte aligned)
template <typename T, int EP> __device__ __forceinline__ void load_ep(const char* p, float* v) {
  if constexpr (EP == TT<T>::EPC) {
    load_chunk<T>(p, v);
  } else {
    static_assert(EP == 4 && TT<T>::SZ == 2, "half chunks exist for 16-bit storage only");
    using H = typename TT<T>::H;
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = h2f<H>(t[0] & 0xffffu); v[1] = h2f<H>(t[0] >> 16); v[2] = h2f<H>(t[1] & 0xffffu); v[3] = h2f<H>(t[1] >> 16);
  }
}
template <typename T, int EP> __device__ __forceinline__ void store_ep(char* p, const float* v) {
  if constexpr (EP == TT<T>::EPC) {
    store_chunk<T>(p, v);
  } else {
    static_assert(EP == 4 && TT<T>::SZ == 2, "half chunks exist for 16-bit storage only");
    using H = typename TT<T>::H;
    *reinterpret_cast<u32x2*>(p) = u32x2{pack2<H>(v[0], v[1]), pack2<H>(v[2], v[3])};
  }
}

// 16-byte chunk `second` (0 / 1) of a 12-wide bf16 row (24 bytes, 8-byte aligned) as 8-byte pieces: chunk 0 is two of
// them, chunk 1 is channels 8..11 + zero fill.  (One 16-byte load from the 8-byte aligned address measured 1.4x slower on
// the whole pass: the memory pipeline splits it.)
__device__ __forceinline__ u32x4 load_row12_chunk(const char* p, bool second) {
  const u32x2 lo = *reinterpret_cast<const u32x2*>(p);
  u32x2 hi = {0u, 0u};
  if (!second) hi = *reinterpret_cast<const u32x2*>(p + 8);
  return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// n consecutive elements, fp32 or 16-bit kind H by a runtime flag -> floats (n = 4 or 8; pointer aligned to n*size)
template <typename H = BF16> __device__ __forceinline__ void load_n_as_float(const char* p, bool is_f32, int n, float* v) {
  if (is_f32) {
    for (int i = 0; i < n; i += 4) load_chunk<float>(p + 4 * i, v + i);
  } else if (n == 8) {
    load_chunk<H>(p, v);
  } else {
    u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = h2f<H>(t[0] & 0xffffu); v[1] = h2f<H>(t[0] >> 16); v[2] = h2f<H>(t[1] & 0xffffu); v[3] = h2f<H>(t[1] >> 16);
  }
}

// One element of the InstanceNorm + PReLU backward (what instnorm_prelu_bwd_apply_kernel stores): shared by the apply pass and by
// the passes that form it on load, so that both round the same way
__device__ __forceinline__ float inorm_prelu_bwd_value(float g, float y, float mean, float rstd, float s1, float s2, float al) {
  const float xh = (y - mean) * rstd;
  const float dxh = g * (xh > 0.f ? 1.f : al);
  return rstd * (dxh - s1 - xh * s2);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- host-side error plumbing -------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define CTSEG_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ctseg::set_error(__VA_ARGS__);        \
      return -1;                            \
    }                                       \
  } while (0)
#define CTSEG_LAUNCH_CHECK(name)                                         \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      ctseg::set_error("%s: %s", name, hipGetErrorString(e_));          \
      return -2;                                                         \
    }                                                                    \
  } while (0)

// ABI 2: a descriptor names the size of the struct its caller was compiled against; anything else is a caller built against
// another header (round 2 grew both structs under version 1: the library would read past the end of the old, shorter struct)
template <class D> static inline bool desc_ok(const D* d) { return d != nullptr && d->struct_size == (int32_t)sizeof(D); }
#define CTSEG_REQUIRE_DESC(d, name)                                                                                         \
  CTSEG_REQUIRE(ctseg::desc_ok(d), "%s: descriptor struct_size %d != %d (caller compiled against another ctseg_hip.h; ABI %d)", \
                name, (d) ? (d)->struct_size : -1, (int)sizeof(*(d)), CTSEG_ABI_VERSION)

static inline bool is16(int dtype) { return dtype == CTSEG_BF16 || dtype == CTSEG_F16; }   // 16-bit storage kinds

static inline int ilog_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ctseg
