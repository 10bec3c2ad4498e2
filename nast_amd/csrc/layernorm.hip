// LayerNorm forward / backward, one 64-wide wavefront per row (d <= 1024 -> <= 16 values per lane,
// kept in registers; HBM-bound: 8d bytes/row forward).
#include <initializer_list>

#include <stdlib.h>

#include "common.h"

#define LN_MAX_PER_LANE 16

__device__ __forceinline__ int64_t ln_in_row(int r, int grp, int grp_stride, int grp_off) {
  return grp > 0 ? (int64_t)(r / grp) * grp_stride + grp_off + (r % grp) : r;
}

__global__ __launch_bounds__(256) void nast_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          int rows_out, int d, float eps, int grp, int grp_stride, int grp_off) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows_out) return;
  const float* xr = x + ln_in_row(r, grp, grp_stride, grp_off) * d;
  float v[LN_MAX_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < d ? xr[c] : 0.f;
    s += v[i];
  }
  const float mu = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    const float t = c < d ? v[i] - mu : 0.f;
    q += t * t;
  }
  const float rs = rsqrtf(wave_sum(q) / d + eps);
  float* yr = y + (int64_t)r * d;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    if (c < d) yr[c] = (v[i] - mu) * rs * gamma[c] + beta[c];
  }
  if (lane == 0) {
    mean[r] = mu;
    rstd[r] = rs;
  }
}

__global__ __launch_bounds__(256) void nast_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          const float* __restrict__ dx_add,
                                                          int rows_out, int d, int grp, int grp_stride, int grp_off) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][4][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gam[LN_MAX_PER_LANE], ag[LN_MAX_PER_LANE], ab[LN_MAX_PER_LANE];
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    gam[i] = c < d ? gamma[c] : 0.f;
    ag[i] = 0.f;
    ab[i] = 0.f;
  }
  for (int r = blockIdx.x * 4 + wave; r < rows_out; r += gridDim.x * 4) {
    const int64_t ir = ln_in_row(r, grp, grp_stride, grp_off);
    const float* xr = x + ir * d;
    const float* gr = dy + (int64_t)r * d;
    const float mu = mean[r], rs = rstd[r];
    float xh[LN_MAX_PER_LANE], g[LN_MAX_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < d;
      const float dyv = ok ? gr[c] : 0.f;
      xh[i] = ok ? (xr[c] - mu) * rs : 0.f;
      g[i] = dyv * gam[i];
      s1 += g[i];
      s2 += g[i] * xh[i];
      ag[i] += dyv * xh[i];
      ab[i] += dyv;
    }
    s1 = wave_sum(s1) / d;
    s2 = wave_sum(s2) / d;
    float* dxr = dx + ir * d;
    const float* addr = dx_add ? dx_add + ir * d : nullptr;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < d) dxr[c] = rs * (g[i] - s1 - xh[i] * s2) + (addr ? addr[c] : 0.f);
    }
  }
  // cross-wave reduction of the parameter gradients, then one atomic per column per workgroup
  float* rg = red;
  float* rb = red + 4 * d;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    if (c < d) {
      rg[wave * d + c] = ag[i];
      rb[wave * d + c] = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    if (dgamma) atomicAdd(dgamma + c, rg[c] + rg[d + c] + rg[2 * d + c] + rg[3 * d + c]);   // (NULL: frozen parameter)
    if (dbeta) atomicAdd(dbeta + c, rb[c] + rb[d + c] + rb[2 * d + c] + rb[3 * d + c]);
  }
}


// ---- 16-byte variants (d % 4 == 0, identity or mapped rows): lane owns columns 4*lane + 256*i .. +3 -----------
template <int NV>
__global__ __launch_bounds__(256) void nast_ln_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              float* __restrict__ mean, float* __restrict__ rstd, int rows_out,
                                                              int d, float eps, int grp, int grp_stride, int grp_off) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows_out) return;
  const float* xr = x + ln_in_row(r, grp, grp_stride, grp_off) * d;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    v[i] = c < d ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mu = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      const float a = v[i].x - mu, b = v[i].y - mu, e = v[i].z - mu, f = v[i].w - mu;
      q += (a * a + b * b) + (e * e + f * f);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / d + eps);
  float* yr = y + (int64_t)r * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x; o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z; o.w = (v[i].w - mu) * rs * g.w + b.w;
      *reinterpret_cast<float4*>(yr + c) = o;
    }
  }
  if (lane == 0) {
    mean[r] = mu;
    rstd[r] = rs;
  }
}

// 16 wavefronts per workgroup and at most one workgroup per CU: enough rows in flight to stream at HBM rate while
// each column of dgamma / dbeta receives <= 256 atomic adds (same-address fp32 atomics serialise at ~20 G/s)
#define LN_BWD_WAVES 16
template <int NV, int RB>
__global__ __launch_bounds__(1024) void nast_ln_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dx,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              const float* __restrict__ dx_add, int rows_out, int d, int grp,
                                                              int grp_stride, int grp_off) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][LN_BWD_WAVES][d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 gam[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    gam[i] = c < d ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // RB rows of a wavefront are in flight together (all their loads are requested before the first is used): at the training
  // step's ~10k rows a wavefront owns 2-3 rows, and one row at a time meant 2-3 dependent HBM round trips per launch
  const int stride = gridDim.x * LN_BWD_WAVES;
  for (int r0 = blockIdx.x * LN_BWD_WAVES + wave; r0 < rows_out; r0 += stride * RB) {
    float4 dyv[RB][NV], xv[RB][NV], ad[RB][NV];
    float mu[RB], rs[RB];
    int64_t ir[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int r = r0 + k * stride;
      const bool rok = r < rows_out;
      ir[k] = rok ? ln_in_row(r, grp, grp_stride, grp_off) : 0;
      mu[k] = rok ? mean[r] : 0.f;
      rs[k] = rok ? rstd[r] : 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        const bool ok = rok && c < d;
        dyv[k][i] = ok ? *reinterpret_cast<const float4*>(dy + (int64_t)r * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        xv[k][i] = ok ? *reinterpret_cast<const float4*>(x + ir[k] * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        ad[k][i] = (ok && dx_add) ? *reinterpret_cast<const float4*>(dx_add + ir[k] * d + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      if (r0 + k * stride >= rows_out) break;     // (uniform for the wavefront)
      float4 xh[NV], g[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const bool ok = 4 * lane + 256 * i < d;
        const float m = ok ? mu[k] : 0.f;          // (columns past d hold zeros and must stay zeros)
        xh[i].x = (xv[k][i].x - m) * rs[k]; xh[i].y = (xv[k][i].y - m) * rs[k];
        xh[i].z = (xv[k][i].z - m) * rs[k]; xh[i].w = (xv[k][i].w - m) * rs[k];
        g[i].x = dyv[k][i].x * gam[i].x; g[i].y = dyv[k][i].y * gam[i].y; g[i].z = dyv[k][i].z * gam[i].z; g[i].w = dyv[k][i].w * gam[i].w;
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
        ag[i].x += dyv[k][i].x * xh[i].x; ag[i].y += dyv[k][i].y * xh[i].y; ag[i].z += dyv[k][i].z * xh[i].z; ag[i].w += dyv[k][i].w * xh[i].w;
        ab[i].x += dyv[k][i].x; ab[i].y += dyv[k][i].y; ab[i].z += dyv[k][i].z; ab[i].w += dyv[k][i].w;
      }
      s1 = wave_sum(s1) / d;
      s2 = wave_sum(s2) / d;
      float* dxr = dx + ir[k] * d;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = 4 * lane + 256 * i;
        if (c < d) {
          float4 o;
          o.x = rs[k] * (g[i].x - s1 - xh[i].x * s2) + ad[k][i].x; o.y = rs[k] * (g[i].y - s1 - xh[i].y * s2) + ad[k][i].y;
          o.z = rs[k] * (g[i].z - s1 - xh[i].z * s2) + ad[k][i].z; o.w = rs[k] * (g[i].w - s1 - xh[i].w * s2) + ad[k][i].w;
          *reinterpret_cast<float4*>(dxr + c) = o;
        }
      }
    }
  }
  float* rg = red;
  float* rb = red + LN_BWD_WAVES * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = 4 * lane + 256 * i;
    if (c < d) {
      *reinterpret_cast<float4*>(rg + wave * d + c) = ag[i];
      *reinterpret_cast<float4*>(rb + wave * d + c) = ab[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += LN_BWD_WAVES * 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < LN_BWD_WAVES; ++w) {
      sg += rg[w * d + c];
      sb += rb[w * d + c];
    }
    if (dgamma) atomicAdd(dgamma + c, sg);     // (NULL: frozen parameter)
    if (dbeta) atomicAdd(dbeta + c, sb);
  }
}

static bool ln_vec_ok(int d, std::initializer_list<const void*> ptrs) {
  if (d & 3) return false;
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return false;
  return true;
}

extern "C" int nast_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                      float* rstd, int rows_out, int d, float eps, int grp, int grp_stride,
                                      int grp_off, void* stream) {
  NAST_REQUIRE(d >= 1 && d <= 64 * LN_MAX_PER_LANE, "nast_layernorm_fwd_f32: d=%d unsupported (max %d)", d, 64 * LN_MAX_PER_LANE);
  NAST_REQUIRE(rows_out >= 0, "nast_layernorm_fwd_f32: rows_out < 0");
  if (rows_out == 0) return NAST_OK;
  NAST_REQUIRE(x && gamma && beta && y && mean && rstd, "nast_layernorm_fwd_f32: NULL pointer");
  const dim3 grid((rows_out + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (ln_vec_ok(d, {x, gamma, beta, y}) && d <= 256)
    hipLaunchKernelGGL(nast_ln_fwd_vec_kernel<1>, grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows_out, d, eps, grp, grp_stride, grp_off);
  else if (ln_vec_ok(d, {x, gamma, beta, y}) && d <= 512)
    hipLaunchKernelGGL(nast_ln_fwd_vec_kernel<2>, grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows_out, d, eps, grp, grp_stride, grp_off);
  else if (ln_vec_ok(d, {x, gamma, beta, y}))
    hipLaunchKernelGGL(nast_ln_fwd_vec_kernel<4>, grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows_out, d, eps, grp, grp_stride, grp_off);
  else
    hipLaunchKernelGGL(nast_ln_fwd_kernel, grid, block, 0, st, x, gamma, beta, y, mean, rstd, rows_out, d, eps, grp, grp_stride, grp_off);
  NAST_CHECK_LAUNCH("nast_layernorm_fwd_f32");
  return NAST_OK;
}

extern "C" int nast_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, float* dgamma, float* dbeta, const float* dx_add,
                                      int rows_out, int rows_in, int d, int grp, int grp_stride, int grp_off, void* stream) {
  NAST_REQUIRE(dx_add == nullptr || grp == 0, "nast_layernorm_bwd_f32: dx_add needs the identity row map");
  NAST_REQUIRE(d >= 1 && d <= 64 * LN_MAX_PER_LANE, "nast_layernorm_bwd_f32: d=%d unsupported", d);
  NAST_REQUIRE(rows_out >= 0 && rows_in >= rows_out, "nast_layernorm_bwd_f32: bad row counts %d/%d", rows_out, rows_in);
  if (rows_in == 0) return NAST_OK;
  NAST_REQUIRE(dy && x && gamma && mean && rstd && dx, "nast_layernorm_bwd_f32: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (grp > 0) {
    if (hipMemsetAsync(dx, 0, (size_t)rows_in * d * sizeof(float), s) != hipSuccess) {
      nast_set_error("nast_layernorm_bwd_f32: memset failed");
      return NAST_ELAUNCH;
    }
  }
  if (rows_out == 0) return NAST_OK;
  const bool vec = ln_vec_ok(d, {dy, x, gamma, dx, dx_add});
  if (vec) {
    int blocks = (rows_out + LN_BWD_WAVES - 1) / LN_BWD_WAVES;
    // every workgroup ends with 2 d atomic adds into the same 2 d floats; measured at 9920 rows x 256 (tools/ln_bench.py): the
    // streaming part takes 9.0 us on 256 workgroups / 10.2 on 128, the atomic tail 4.0 / 1.5 us -- 128 wins until the rows double
    static const int forced = [] { const char* e = getenv("NAST_LN_BWD_BLOCKS"); return e ? atoi(e) : 0; }();   // tuning aid
    const int max_blocks = forced > 0 ? forced : (rows_out >= 16384 ? 256 : 128);
    if (blocks > max_blocks) blocks = max_blocks;
    const size_t lds = (size_t)2 * LN_BWD_WAVES * d * sizeof(float);
    const dim3 grid(blocks), block(LN_BWD_WAVES * 64);
    if (d <= 256) {
      hipLaunchKernelGGL((nast_ln_bwd_vec_kernel<1, 3>), grid, block, lds, s, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dx_add, rows_out, d, grp, grp_stride, grp_off);
    } else if (d <= 512) {
      hipLaunchKernelGGL((nast_ln_bwd_vec_kernel<2, 2>), grid, block, lds, s, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dx_add, rows_out, d, grp, grp_stride, grp_off);
    } else {
      static bool attr_set = false;
      if (!attr_set) {
        hipFuncSetAttribute((const void*)nast_ln_bwd_vec_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LN_BWD_WAVES * 1024 * 4);
        attr_set = true;
      }
      hipLaunchKernelGGL((nast_ln_bwd_vec_kernel<4, 1>), grid, block, lds, s, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, dx_add, rows_out, d, grp, grp_stride, grp_off);
    }
  } else {
    int blocks = (rows_out + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(nast_ln_bwd_kernel, dim3(blocks), dim3(256), 2 * 4 * d * sizeof(float), s, dy, x, gamma, mean, rstd,
                       dx, dgamma, dbeta, dx_add, rows_out, d, grp, grp_stride, grp_off);
  }
  NAST_CHECK_LAUNCH("nast_layernorm_bwd_f32");
  return NAST_OK;
}
