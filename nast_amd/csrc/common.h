// Shared helpers for libnast_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nast_hip.h"

#define NAST_WAVE 64

void nast_set_error(const char* fmt, ...);

#define NAST_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      nast_set_error(__VA_ARGS__);              \
      return NAST_EINVAL;                       \
    }                                           \
  } while (0)

#define NAST_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      nast_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return NAST_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

// ---- in-library profiling hooks (prof.hip) -------------------------------------------------
void nast_prof_before(int kind, hipStream_t s);
void nast_prof_after(int kind, hipStream_t s, double work, const char* sig = nullptr);
bool nast_prof_shapes(int kind);   // true when per-signature timing is wanted (NAST_PROF_SHAPES set and the family armed)

// ---- wave / block reductions (64-wide wavefronts) ------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum / max for blockDim.x == NT (multiple of 64); `red` is LDS scratch of NT/64 floats.
// All threads get the result. Contains two barriers.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}

__device__ __forceinline__ int64_t row_offset(int64_t r, nast_rows a) {
  return a.grp > 0 ? (r / a.grp) * a.grp_stride + (r % a.grp) * a.ld : r * a.ld;
}
