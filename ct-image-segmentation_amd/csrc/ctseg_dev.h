// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes, MFMA 16x16 family).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "ctseg_hip.h"

namespace ctseg {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct BF16 {};  // storage tag: 16-bit brain float kept as raw ushort

template <typename T> struct TT;
template <> struct TT<float> { static constexpr int SZ = 4, EPC = 4, DT = CTSEG_F32; };
template <> struct TT<BF16> { static constexpr int SZ = 2, EPC = 8, DT = CTSEG_BF16; };

__device__ __forceinline__ float bf2f(uint32_t u16) { return __uint_as_float(u16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

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

// EP consecutive elements as floats: EP = EPC<T> (one 16-byte chunk) or, for bf16 rows whose stride is an odd multiple of
// 4 elements (10 classes stored 12 wide: 24-byte rows), EP = 4 (an 8-byte chunk; rows are then only 8-byte aligned)
template <typename T, int EP> __device__ __forceinline__ void load_ep(const char* p, float* v) {
  if constexpr (EP == TT<T>::EPC) {
    load_chunk<T>(p, v);
  } else {
    static_assert(EP == 4 && TT<T>::SZ == 2, "half chunks exist for bf16 only");
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = bf2f(t[0] & 0xffffu); v[1] = bf2f(t[0] >> 16); v[2] = bf2f(t[1] & 0xffffu); v[3] = bf2f(t[1] >> 16);
  }
}
template <typename T, int EP> __device__ __forceinline__ void store_ep(char* p, const float* v) {
  if constexpr (EP == TT<T>::EPC) {
    store_chunk<T>(p, v);
  } else {
    static_assert(EP == 4 && TT<T>::SZ == 2, "half chunks exist for bf16 only");
    *reinterpret_cast<u32x2*>(p) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
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

// n consecutive elements of runtime dtype -> floats (n = 4 or 8; pointer aligned to n*size)
__device__ __forceinline__ void load_n_as_float(const char* p, bool is_f32, int n, float* v) {
  if (is_f32) {
    for (int i = 0; i < n; i += 4) load_chunk<float>(p + 4 * i, v + i);
  } else if (n == 8) {
    load_chunk<BF16>(p, v);
  } else {
    u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = bf2f(t[0] & 0xffffu); v[1] = bf2f(t[0] >> 16); v[2] = bf2f(t[1] & 0xffffu); v[3] = bf2f(t[1] >> 16);
  }
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

static inline int ilog_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace ctseg
