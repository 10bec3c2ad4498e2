// Multi-tensor gradient-norm + clip + Adam (seq2seq.py:536-537, :637-638): two launches for the whole
// parameter group instead of ~4 launches per tensor.  grid = (32, n_tensors), grid-stride inside a tensor.
#include "common.h"

__global__ __launch_bounds__(256) void nast_multi_sqnorm_kernel(const float* const* __restrict__ grads,
                                                                const int64_t* __restrict__ sizes, float* __restrict__ out_sq) {
  __shared__ double red[4];
  const float* g = grads[blockIdx.y];
  const int64_t n = sizes[blockIdx.y];
  double s = 0.0;                                // fp64 partial sums: the kernel is memory-bound, the adds are free
  if (g)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += (double)g[i] * (double)g[i];
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
                                                              float weight_decay, float bias_c1, float bias_c2) {
  float* g = grads[blockIdx.y];
  if (!g) return;
  float* p = params[blockIdx.y];
  float* m = exp_avg[blockIdx.y];
  float* v = exp_avg_sq[blockIdx.y];
  const int64_t n = sizes[blockIdx.y];
  float coef = 1.f;
  if (total_sq) {
    // torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max=1)
    coef = fminf(max_norm / (sqrtf(*total_sq) + 1e-6f), 1.f);
  }
  // operation order of torch.optim.Adam (_single_tensor_adam): exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g,
  // value=1-b2); denom = exp_avg_sq.sqrt() / sqrt(bias_c2) + eps; p.addcdiv_(exp_avg, denom, value=-lr/bias_c1)
  const float step = lr / bias_c1;
  const float bc2_sqrt = sqrtf(bias_c2);
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i];
    if (total_sq) {
      gi *= coef;
      g[i] = gi;                                 // clip_grad_norm_ scales .grad in place
    }
    if (weight_decay != 0.f) gi += weight_decay * p[i];
    const float m0 = m[i];
    const float mi = m0 + w1 * (gi - m0);
    const float vi = v[i] * beta2 + (w2 * gi) * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] += -step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

extern "C" int nast_multi_sqnorm_f32(const float* const* grads, const int64_t* sizes, int n, float* out_sq, void* stream) {
  NAST_REQUIRE(n >= 0 && n <= 65535, "nast_multi_sqnorm_f32: n=%d out of range", n);
  if (n == 0) return NAST_OK;
  NAST_REQUIRE(grads && sizes && out_sq, "nast_multi_sqnorm_f32: NULL pointer");
  hipLaunchKernelGGL(nast_multi_sqnorm_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream, grads, sizes, out_sq);
  NAST_CHECK_LAUNCH("nast_multi_sqnorm_f32");
  return NAST_OK;
}

extern "C" int nast_multi_adam_f32(float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                   const int64_t* sizes, int n, const float* total_sqnorm, float max_norm, float lr, float beta1,
                                   float beta2, float eps, float weight_decay, float bias_c1, float bias_c2, void* stream) {
  NAST_REQUIRE(n >= 0 && n <= 65535, "nast_multi_adam_f32: n=%d out of range", n);
  if (n == 0) return NAST_OK;
  NAST_REQUIRE(params && grads && exp_avg && exp_avg_sq && sizes, "nast_multi_adam_f32: NULL pointer");
  hipLaunchKernelGGL(nast_multi_adam_kernel, dim3(32, n), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq,
                     sizes, total_sqnorm, max_norm, lr, beta1, beta2, eps, weight_decay, bias_c1, bias_c2);
  NAST_CHECK_LAUNCH("nast_multi_adam_f32");
  return NAST_OK;
}
