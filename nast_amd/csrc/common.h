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
// Every lane gets the result.  (Round 3: DPP butterflies inside a row of 16 lanes -- quad_perm xor 1, xor 2, then
// row_half_mirror and row_mirror, which pair quads / halves that already hold equal values -- and v_readlane across the four
// rows, instead of the six ds_bpermute that `__shfl_xor` compiles to: no LDS-crossbar round trips in the row kernels.)
template <int CTRL>
__device__ __forceinline__ float nast_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float nast_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += nast_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += nast_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += nast_dpp<0x141>(v);     // row_half_mirror
  v += nast_dpp<0x140>(v);     // row_mirror
  return (nast_lane(v, 0) + nast_lane(v, 16)) + (nast_lane(v, 32) + nast_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, nast_dpp<0xB1>(v));
  v = fmaxf(v, nast_dpp<0x4E>(v));
  v = fmaxf(v, nast_dpp<0x141>(v));
  v = fmaxf(v, nast_dpp<0x140>(v));
  return fmaxf(fmaxf(nast_lane(v, 0), nast_lane(v, 16)), fmaxf(nast_lane(v, 32), nast_lane(v, 48)));
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
