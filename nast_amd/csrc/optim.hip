// Multi-tensor gradient-norm + clip + Adam (seq2seq.py:536-537, :637-638): two launches for the whole
// parameter group instead of ~4 launches per tensor.  Work is dealt in chunks of `chunk_elems` consecutive elements of one
// tensor (table built by the caller: chunk -> tensor, first element), one workgroup per chunk, so the three 2.4 M-element
// embedding / generator matrices spread over the chip like the 260 small tensors do (a (32, n_tensors) grid left them on
// 32 workgroups each: 1 TB/s).  Without a table: grid = (32, n_tensors), grid-stride inside a tensor.
#include "common.h"

struct ChunkRange {
  int tensor;
  int64_t lo, hi, step;
};
__device__ __forceinline__ ChunkRange chunk_range(const int32_t* __restrict__ chunk_tensor, const int64_t* __restrict__ chunk_off,
                                                  int chunk_elems, const int64_t* __restrict__ sizes) {
  ChunkRange r;
  if (chunk_tensor != nullptr) {
    r.tensor = chunk_tensor[blockIdx.x];
    const int64_t n = sizes[r.tensor];
    r.lo = chunk_off[blockIdx.x] + threadIdx.x;
    r.hi = chunk_off[blockIdx.x] + chunk_elems < n ? chunk_off[blockIdx.x] + chunk_elems : n;
    r.step = 256;
  } else {
    r.tensor = blockIdx.y;
    r.lo = (int64_t)blockIdx.x * 256 + threadIdx.x;
    r.hi = sizes[r.tensor];
    r.step = (int64_t)gridDim.x * 256;
  }
  return r;
}

__global__ __launch_bounds__(256) void nast_multi_sqnorm_kernel(const float* const* __restrict__ grads,
                                                                const int64_t* __restrict__ sizes, float* __restrict__ out_sq,
                                                                const int32_t* __restrict__ chunk_tensor,
                                                                const int64_t* __restrict__ chunk_off, int chunk_elems) {
  __shared__ double red[4];
  const ChunkRange R = chunk_range(chunk_tensor, chunk_off, chunk_elems, sizes);
  const float* g = grads[R.tensor];
  double s = 0.0;                                // fp64 partial sums: the kernel is memory-bound, the adds are free
  if (g) {
    int64_t lo = R.lo;
    if (chunk_tensor != nullptr) {               // 16-byte loads over the chunk's 4-aligned body, scalar tail
      const int64_t c0 = R.lo - threadIdx.x;
      if ((reinterpret_cast<uintptr_t>(g + c0) & 15) == 0) {
        const int64_t n4 = (R.hi - c0) / 4;
        for (int64_t q = threadIdx.x; q < n4; q += 256) {
          const float4 x = *reinterpret_cast<const float4*>(g + c0 + 4 * q);
          s += (double)x.x * (double)x.x + (double)x.y * (double)x.y + (double)x.z * (double)x.z + (double)x.w * (double)x.w;
        }
        lo = c0 + 4 * n4 + threadIdx.x;
      }
    }
    for (int64_t i = lo; i < R.hi; i += R.step) s += (double)g[i] * (double)g[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = red[0] + red[1] + red[2] + red[3];
    if (t != 0.0) atomicAdd(out_sq, (float)t);
  }
}

__global__ __launch_bounds__(256) void nast_multi_adam_kernel(float* const* __restrict__ params, float* const* __restrict__ grads,
                                                              float* const* __restrict__ exp_avg, float* const* __restrict__ exp_avg_sq,
                                                              const int64_t* __restrict__ sizes, const float* __restrict__ total_sq,
                                                              float max_norm, float lr, float beta1, float beta2, float eps,
                                                              float weight_decay, float bias_c1, float bias_c2,
                                                              const float* __restrict__ bias_dev,
                                                              const int32_t* __restrict__ chunk_tensor,
                                                              const int64_t* __restrict__ chunk_off, int chunk_elems) {
  const ChunkRange R = chunk_range(chunk_tensor, chunk_off, chunk_elems, sizes);
  float* g = grads[R.tensor];
  if (!g) return;
  float* p = params[R.tensor];
  float* m = exp_avg[R.tensor];
  float* v = exp_avg_sq[R.tensor];
  float coef = 1.f;
  if (total_sq) {
    // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1)
    coef = fminf(max_norm / (sqrtf(*total_sq) + 1e-6f), 1.f);
  }
  // operation order of torch.optim.Adam (_single_tensor_adam): exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g,
  // value=1-b2); denom = exp_avg_sq.sqrt() / sqrt(bias_c2) + eps; p.addcdiv_(exp_avg, denom, value=-lr/bias_c1)
  if (bias_dev) { bias_c1 = bias_dev[0]; bias_c2 = bias_dev[1]; }   // a replayed HIP graph: the step count lives on the device
  const float step = lr / bias_c1;
  const float bc2_sqrt = sqrtf(bias_c2);
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  auto one = [&](float& gi, float& pi, float& mi, float& vi) {
    if (total_sq) gi *= coef;                    // clip_grad_norm_ scales .grad in place
    float ge = gi;
    if (weight_decay != 0.f) ge += weight_decay * pi;
    mi = mi + w1 * (ge - mi);
    vi = vi * beta2 + (w2 * ge) * ge;
    pi += -step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  };
  int64_t lo = R.lo;
  if (chunk_tensor != nullptr) {
    // chunk of consecutive elements: 16-byte accesses over its 4-aligned body (same arithmetic per element), scalar tail
    const int64_t c0 = R.lo - threadIdx.x;
    const bool al = ((reinterpret_cast<uintptr_t>(g + c0) | reinterpret_cast<uintptr_t>(p + c0) | reinterpret_cast<uintptr_t>(m + c0) |
                      reinterpret_cast<uintptr_t>(v + c0)) & 15) == 0;
    if (al) {
      const int64_t n4 = (R.hi - c0) / 4;
      for (int64_t q = threadIdx.x; q < n4; q += 256) {
        const int64_t i = c0 + 4 * q;
        float4 g4 = *reinterpret_cast<const float4*>(g + i), p4 = *reinterpret_cast<const float4*>(p + i);
        float4 m4 = *reinterpret_cast<const float4*>(m + i), v4 = *reinterpret_cast<const float4*>(v + i);
        one(g4.x, p4.x, m4.x, v4.x); one(g4.y, p4.y, m4.y, v4.y); one(g4.z, p4.z, m4.z, v4.z); one(g4.w, p4.w, m4.w, v4.w);
        if (total_sq) *reinterpret_cast<float4*>(g + i) = g4;
        *reinterpret_cast<float4*>(m + i) = m4;
        *reinterpret_cast<float4*>(v + i) = v4;
        *reinterpret_cast<float4*>(p + i) = p4;
      }
      lo = c0 + 4 * n4 + threadIdx.x;
    }
  }
  for (int64_t i = lo; i < R.hi; i += R.step) {
    float gi = g[i], pi = p[i], mi = m[i], vi = v[i];
    one(gi, pi, mi, vi);
    if (total_sq) g[i] = gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
  }
}

extern "C" int nast_multi_sqnorm_f32(const float* const* grads, const int64_t* sizes, int n, float* out_sq, const int32_t* chunk_tensor,
                                     const int64_t* chunk_off, int n_chunks, int chunk_elems, void* stream) {
  NAST_REQUIRE(n >= 0 && n <= 65535, "nast_multi_sqnorm_f32: n=%d out of range", n);
  if (n == 0) return NAST_OK;
  NAST_REQUIRE(grads && sizes && out_sq, "nast_multi_sqnorm_f32: NULL pointer");
  NAST_REQUIRE(chunk_tensor == nullptr || (chunk_off != nullptr && n_chunks > 0 && chunk_elems > 0), "nast_multi_sqnorm_f32: chunk table");
  const dim3 grid = chunk_tensor ? dim3(n_chunks) : dim3(32, n);
  hipLaunchKernelGGL(nast_multi_sqnorm_kernel, grid, dim3(256), 0, (hipStream_t)stream, grads, sizes, out_sq, chunk_tensor, chunk_off,
                     chunk_elems);
  NAST_CHECK_LAUNCH("nast_multi_sqnorm_f32");
  return NAST_OK;
}

extern "C" int nast_multi_adam_f32(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                   const int64_t* sizes, int n, const float* total_sqnorm, float max_norm, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, float bias_c1, float bias_c2, const float* bias_dev,
                                   const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_elems,
                                   void* stream) {
  NAST_REQUIRE(n >= 0 && n <= 65535, "nast_multi_adam_f32: n=%d out of range", n);
  if (n == 0) return NAST_OK;
  NAST_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes, "nast_multi_adam_f32: NULL pointer");
  NAST_REQUIRE(chunk_tensor == nullptr || (chunk_off != nullptr && n_chunks > 0 && chunk_elems > 0), "nast_multi_adam_f32: chunk table");
  const dim3 grid = chunk_tensor ? dim3(n_chunks) : dim3(32, n);
  hipLaunchKernelGGL(nast_multi_adam_kernel, grid, dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     sizes, total_sqnorm, max_norm, lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2, bias_dev, chunk_tensor, chunk_off,
                     chunk_elems);
  NAST_CHECK_LAUNCH("nast_multi_adam_f32");
  return NAST_OK;
}
