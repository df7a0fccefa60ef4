// common.h - shared helpers for the gfx950 kernels (error plumbing, dtype traits, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dss_hip.h"

namespace dss {

// ---- error plumbing (thread-local message, int status at the ABI) -------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);

#define DSS_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return ::dss::fail(DSS_ERR_BAD_ARG, __VA_ARGS__); \
  } while (0)

#define DSS_CHECK_LAUNCH(name)                                                          \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess)                                                              \
      return ::dss::fail(DSS_ERR_HIP, "%s launch failed: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---- types ---------------------------------------------------------------------------------------
typedef _Float16 f16;
typedef __bf16 bf16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class T> struct vec8;
template <> struct vec8<f16> { typedef f16x8 type; };
template <> struct vec8<bf16> { typedef bf16x8 type; };
template <class T> struct vec4;
template <> struct vec4<f16> { typedef f16x4 type; };
template <> struct vec4<bf16> { typedef bf16x4 type; };
template <> struct vec4<float> { typedef f32x4 type; };

template <class T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }
template <class T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }

__device__ __forceinline__ f32x16 mfma32x32x16(f16x8 a, f16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- wave64 reductions (butterfly; every lane ends with the result) -----------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

}  // namespace dss
