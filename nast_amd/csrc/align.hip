// Learnable-alignment kernels (LengthPredictor.match, transformer.py:394-449).
//   nast_align_cost_f32     squared-distance cost tables, one wavefront per target row
//   nast_align_viterbi_f32  min-plus DP + backtrack, ONE WAVEFRONT PER SENTENCE, lane = source position,
//                           f[s-1..s-3] via __shfl_up (S <= 64) or an LDS ring (S <= 256); 1-byte backpointers in LDS
//   nast_align_lse_*        soft (log-sum-exp) version of the same recurrence, forward + backward (extension,
//                           no counterpart in the reference)
#include "common.h"

#define ALIGN_BIG 1e9f
#define ALIGN_MAXD_PER_LANE 16

// ---------------------------------------------------------------------------------------------
// cost tables
// ---------------------------------------------------------------------------------------------
// NU = channels per lane (d <= 64 NU), SU = source positions in flight per wavefront.  The kernel is a chain of dependent
// round trips (source id -> embedding row -> reduction) per source position; SU of them are requested together, which is what
// its time follows (512 sentences x 33 x 33: 132 us one at a time, 70 us with 8 in flight, 73 with 11, 90 with 17).  Per (target, source) pair the arithmetic is
// unchanged: the same per-lane accumulation over the lane's channels, the same wave reduction.
template <int NU, int SU>
__global__ __launch_bounds__(256) void nast_align_cost_kernel(const float* __restrict__ E, int d,
                                                              const int64_t* __restrict__ tgt_ids,
                                                              const int64_t* __restrict__ src_ids,
                                                              const float* __restrict__ src_emb,
                                                              const float* __restrict__ src_unk_w, int64_t unk_stride,
                                                              int64_t unk_id, float* __restrict__ cost0,
                                                              float* __restrict__ cost, int B, int T, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, i)
  if (row >= (int64_t)B * T) return;
  const int b = row / T;
  const float* tp = E + tgt_ids[row] * d;
  float t[NU], fix[NU];
  float n2 = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int c = lane + 64 * u;
    t[u] = c < d ? tp[c] : 0.f;
    fix[u] = (c < d && src_unk_w) ? E[c] - E[unk_id * d + c] : 0.f;   // E[0] - E[unk]
    n2 += t[u] * t[u];
  }
  n2 = wave_sum(n2);
  if (lane == 0) cost0[row] = n2 + 1e-6f;
  for (int s0 = 0; s0 < S; s0 += SU) {
    const float* op[SU];
    float w[SU];
#pragma unroll
    for (int k = 0; k < SU; ++k) {
      const int s = s0 + k < S ? s0 + k : S - 1;                      // (past the end: the last position again, not stored)
      const int64_t sr = (int64_t)b * S + s;
      w[k] = 0.f;
      if (src_ids) {
        int64_t id = src_ids[sr];
        if (id == unk_id) id = 0;
        op[k] = E + id * d;
      } else {
        op[k] = src_emb + sr * d;
        if (src_unk_w) w[k] = src_unk_w[sr * unk_stride];
      }
    }
    float x[SU][NU];
#pragma unroll
    for (int k = 0; k < SU; ++k)
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int c = lane + 64 * u;
        x[k][u] = c < d ? op[k][c] : 0.f;
      }
#pragma unroll
    for (int k = 0; k < SU; ++k) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (lane + 64 * u < d) {
          const float df = t[u] - (x[k][u] + w[k] * fix[u]);
          acc += df * df;
        }
      }
      acc = wave_sum(acc);
      if (lane == 0 && s0 + k < S) {
        const float r = sqrtf(acc);   // the reference squares torch.cdist's output (transformer.py:412)
        cost[row * S + s0 + k] = r * r;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Viterbi
// ---------------------------------------------------------------------------------------------
struct MinIdx {
  float v;
  int i;
};
__device__ __forceinline__ void min_upd(MinIdx& a, float v, int i) {
  if (v < a.v || (v == a.v && i < a.i)) {
    a.v = v;
    a.i = i;
  }
}

template <bool SHFL>
__global__ __launch_bounds__(64) void nast_align_viterbi_kernel(const float* __restrict__ cost0, const float* __restrict__ cost,
                                                                const int64_t* __restrict__ olen_, const int64_t* __restrict__ tlen_,
                                                                int64_t* __restrict__ allpos, int T, int S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // layout: float fbuf[2][S+3] (LDS path only) | int pos[T] | uint8 back[(T-1)*S]
  float* fbuf = reinterpret_cast<float*>(lds_raw);
  const int fstride = S + 3;
  int* pos = reinterpret_cast<int*>(fbuf + (SHFL ? 0 : 2 * fstride));
  unsigned char* back = reinterpret_cast<unsigned char*>(pos + T);

  const int b = blockIdx.x, lane = threadIdx.x;
  const int olen = (int)olen_[b], tlen = (int)tlen_[b];
  const float* c0 = cost0 + (int64_t)b * T;
  const float* cm = cost + (int64_t)b * T * S;
  constexpr int PPL = SHFL ? 1 : 4;   // source positions per lane: s = lane + 64*u
  float f[PPL];
#pragma unroll
  for (int u = 0; u < PPL; ++u) {
    const int s = lane + 64 * u;
    float c = (s < S) ? cm[s] : ALIGN_BIG;
    if (s >= olen) c = ALIGN_BIG;                       // transformer.py:413-414
    f[u] = (s < 3) ? c : ALIGN_BIG;                     // :415
  }
  if (!SHFL) {
    if (lane < 3) { fbuf[lane] = ALIGN_BIG; fbuf[fstride + lane] = ALIGN_BIG; }
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int s = lane + 64 * u;
      if (s < S) fbuf[3 + s] = f[u];
    }
    __syncthreads();
  }
  const int steps = tlen < T ? tlen : T;                // rows i >= target_length are frozen (:429-430)
  for (int i = 1; i < T; ++i) {
    const bool live = i < steps;
    const float ci0 = c0[i];
    float* fo = fbuf + ((i - 1) & 1) * fstride;
    float* fn = fbuf + (i & 1) * fstride;
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int s = lane + 64 * u;
      float f1, f2, f3;
      if (SHFL) {
        f1 = __shfl_up(f[u], 1, 64); if (lane < 1) f1 = ALIGN_BIG;
        f2 = __shfl_up(f[u], 2, 64); if (lane < 2) f2 = ALIGN_BIG;
        f3 = __shfl_up(f[u], 3, 64); if (lane < 3) f3 = ALIGN_BIG;
      } else {
        const int q = (s < S ? s : 0) + 3;
        f1 = fo[q - 1]; f2 = fo[q - 2]; f3 = fo[q - 3];
      }
      float cs = (s < S) ? cm[(int64_t)i * S + s] : ALIGN_BIG;
      if (s >= olen) cs = ALIGN_BIG;
      // first minimal j wins (torch.min over the stacked candidates, :425-427)
      float best = f[u] + ci0; int bj = 0;
      float x = f1 + cs; if (x < best) { best = x; bj = 1; }
      x = f2 + cs; if (x < best) { best = x; bj = 2; }
      x = f3 + cs; if (x < best) { best = x; bj = 3; }
      if (live) f[u] = best;
      if (s < S) back[(i - 1) * S + s] = live ? (unsigned char)bj : (unsigned char)0;
      if (!SHFL && s < S) fn[3 + s] = f[u];
    }
    if (!SHFL) __syncthreads();
  }
  // admissible final states: s >= origin_length - 3 (:433-434); first minimal s wins (:435)
  MinIdx mi{INFINITY, 0x7fffffff};
#pragma unroll
  for (int u = 0; u < PPL; ++u) {
    const int s = lane + 64 * u;
    if (s < S) min_upd(mi, (s >= olen - 3) ? f[u] : ALIGN_BIG, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mi.v, o, 64);
    const int oi = __shfl_xor(mi.i, o, 64);
    min_upd(mi, ov, oi);
  }
  __syncthreads();
  if (lane == 0) {
    int last = mi.i;
    pos[T - 1] = last;
    for (int i = T - 1; i >= 1; --i) {                 // :439-442
      last -= back[(i - 1) * S + last];
      pos[i - 1] = last;
    }
  }
  __syncthreads();
  int64_t* ap = allpos + (int64_t)b * (T + 1);
  for (int i = lane; i <= T; i += 64)
    ap[i] = (i == 0) ? 0 : (int64_t)((i - 1 >= tlen) ? olen : pos[i - 1]) + 1;   // :445-448
}

// ---------------------------------------------------------------------------------------------
// soft (log-sum-exp) alignment -- extension, "parity unpinned": checked against its own CPU restatement
// and against the Viterbi value in the tau -> 0 limit.
// alpha [B,T,S] holds the soft-min forward values (row i >= target_length copies the last live row).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float softmin4(const float (&x)[4], float inv_tau, float tau) {
  const float m = fminf(fminf(x[0], x[1]), fminf(x[2], x[3]));
  const float s = __expf(-(x[0] - m) * inv_tau) + __expf(-(x[1] - m) * inv_tau) + __expf(-(x[2] - m) * inv_tau) +
                  __expf(-(x[3] - m) * inv_tau);
  return m - tau * __logf(s);
}

__global__ __launch_bounds__(64) void nast_align_lse_fwd_kernel(const float* __restrict__ cost0, const float* __restrict__ cost,
                                                                const int64_t* __restrict__ olen_, const int64_t* __restrict__ tlen_,
                                                                float tau, float* __restrict__ value, float* __restrict__ alpha,
                                                                int T, int S) {
  extern __shared__ __attribute__((aligned(16))) float fb[];   // [2][S+3]
  const int fstride = S + 3;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int olen = (int)olen_[b], tlen = (int)tlen_[b];
  const float inv_tau = 1.f / tau;
  const float* c0 = cost0 + (int64_t)b * T;
  const float* cm = cost + (int64_t)b * T * S;
  float* al = alpha + (int64_t)b * T * S;
  if (lane < 3) { fb[lane] = ALIGN_BIG; fb[fstride + lane] = ALIGN_BIG; }
  for (int s = lane; s < S; s += 64) {
    float c = cm[s];
    if (s >= olen) c = ALIGN_BIG;
    const float v = s < 3 ? c : ALIGN_BIG;
    fb[3 + s] = v;
    al[s] = v;
  }
  __syncthreads();
  const int steps = tlen < T ? tlen : T;
  for (int i = 1; i < T; ++i) {
    const float* fo = fb + ((i - 1) & 1) * fstride;
    float* fn = fb + (i & 1) * fstride;
    const float ci0 = c0[i];
    for (int s = lane; s < S; s += 64) {
      float v = fo[3 + s];
      if (i < steps) {
        float cs = cm[(int64_t)i * S + s];
        if (s >= olen) cs = ALIGN_BIG;
        const float x[4] = {fo[3 + s] + ci0, fo[2 + s] + cs, fo[1 + s] + cs, fo[s] + cs};
        v = softmin4(x, inv_tau, tau);
      }
      fn[3 + s] = v;
      al[(int64_t)i * S + s] = v;
    }
    __syncthreads();
  }
  const float* fl = fb + ((T - 1) & 1) * fstride;
  float m = INFINITY;
  for (int s = lane; s < S; s += 64) m = fminf(m, (s >= olen - 3) ? fl[3 + s] : ALIGN_BIG);
  m = -wave_max(-m);
  float sum = 0.f;
  for (int s = lane; s < S; s += 64)
    if (s >= olen - 3) sum += __expf(-(fl[3 + s] - m) * inv_tau);
  sum = wave_sum(sum);
  if (lane == 0) value[b] = m - tau * __logf(sum);
}

__global__ __launch_bounds__(64) void nast_align_lse_bwd_kernel(const float* __restrict__ cost0, const float* __restrict__ cost,
                                                                const int64_t* __restrict__ olen_, const int64_t* __restrict__ tlen_,
                                                                float tau, const float* __restrict__ value,
                                                                const float* __restrict__ alpha, const float* __restrict__ dvalue,
                                                                float* __restrict__ dcost0, float* __restrict__ dcost, int T, int S) {
  extern __shared__ __attribute__((aligned(16))) float gb[];   // [2][S+3]  d value / d alpha_i[s], 3 trailing zeros
  const int gstride = S + 3;
  const int b = blockIdx.x, lane = threadIdx.x;
  const int olen = (int)olen_[b], tlen = (int)tlen_[b];
  const float inv_tau = 1.f / tau;
  const float* c0 = cost0 + (int64_t)b * T;
  const float* cm = cost + (int64_t)b * T * S;
  const float* al = alpha + (int64_t)b * T * S;
  float* dc0 = dcost0 + (int64_t)b * T;
  float* dc = dcost + (int64_t)b * T * S;
  const int steps = tlen < T ? tlen : T;
  const int last = steps >= 1 ? steps - 1 : 0;
  const float dv = dvalue[b], val = value[b];
  if (lane < 3) { gb[S + lane] = 0.f; gb[gstride + S + lane] = 0.f; }
  for (int i = lane; i < T; i += 64) dc0[i] = 0.f;
  for (int i = steps; i < T; ++i)
    for (int s = lane; s < S; s += 64) dc[(int64_t)i * S + s] = 0.f;
  float* g = gb + (last & 1) * gstride;
  for (int s = lane; s < S; s += 64) g[s] = (s >= olen - 3) ? dv * __expf(-(al[(int64_t)last * S + s] - val) * inv_tau) : 0.f;
  __syncthreads();
  for (int i = last; i >= 1; --i) {
    float* gi = gb + (i & 1) * gstride;
    float* gp = gb + ((i - 1) & 1) * gstride;
    const float ci0 = c0[i];
    const float* ai = al + (int64_t)i * S;
    const float* ap = al + (int64_t)(i - 1) * S;
    float acc0 = 0.f;
    for (int s = lane; s < S; s += 64) {
      // edges INTO (i, s): posterior weights w_j, j = 0..3
      float cs = cm[(int64_t)i * S + s];
      const bool dead = s >= olen;
      if (dead) cs = ALIGN_BIG;
      const float gs = gi[s], a = ai[s];
      const float w0 = __expf(-(ap[s] + ci0 - a) * inv_tau);
      float wsum = 0.f;
#pragma unroll
      for (int j = 1; j <= 3; ++j)
        if (s - j >= 0) wsum += __expf(-(ap[s - j] + cs - a) * inv_tau);
      acc0 += gs * w0;
      dc[(int64_t)i * S + s] = dead ? 0.f : gs * wsum;
      // edges OUT OF (i-1, s): gather form of the scatter
      float out = gs * w0;
#pragma unroll
      for (int j = 1; j <= 3; ++j) {
        const int s2 = s + j;
        if (s2 < S) {
          float c2 = cm[(int64_t)i * S + s2];
          if (s2 >= olen) c2 = ALIGN_BIG;
          out += gi[s2] * __expf(-(ap[s] + c2 - ai[s2]) * inv_tau);
        }
      }
      gp[s] = out;
    }
    acc0 = wave_sum(acc0);
    if (lane == 0) dc0[i] = acc0;
    __syncthreads();
  }
  const float* g0 = gb;   // row 0 lives in buffer 0
  for (int s = lane; s < S; s += 64) dc[s] = (s < 3 && s < olen) ? g0[s] : 0.f;
}

// ---------------------------------------------------------------------------------------------
extern "C" int nast_align_cost_f32(const float* E, int V, int d, const int64_t* tgt_ids, const int64_t* src_ids,
                                   const float* src_emb, const float* src_unk_w, int64_t src_unk_stride, int64_t unk_id,
                                   float* cost0, float* cost, int B, int T, int S, void* stream) {
  NAST_REQUIRE(B >= 0 && T >= 1 && S >= 1 && V >= 1, "nast_align_cost_f32: bad sizes");
  NAST_REQUIRE(d >= 1 && d <= 64 * ALIGN_MAXD_PER_LANE, "nast_align_cost_f32: d=%d unsupported", d);
  NAST_REQUIRE(unk_id >= 0 && unk_id < V, "nast_align_cost_f32: unk_id out of range");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(E && tgt_ids && cost0 && cost && (src_ids || src_emb), "nast_align_cost_f32: NULL pointer");
  const int64_t rows = (int64_t)B * T;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (d <= 256) hipLaunchKernelGGL((nast_align_cost_kernel<4, 8>), grid, block, 0, (hipStream_t)stream, E, d, tgt_ids, src_ids,
                     src_emb, src_unk_w, src_unk_stride, unk_id, cost0, cost, B, T, S);
  else hipLaunchKernelGGL((nast_align_cost_kernel<ALIGN_MAXD_PER_LANE, 2>), grid, block, 0, (hipStream_t)stream, E, d, tgt_ids, src_ids,
                     src_emb, src_unk_w, src_unk_stride, unk_id, cost0, cost, B, T, S);
  NAST_CHECK_LAUNCH("nast_align_cost_f32");
  return NAST_OK;
}

extern "C" int nast_align_viterbi_f32(const float* cost0, const float* cost, const int64_t* origin_length,
                                      const int64_t* target_length, int64_t* allpos, int B, int T, int S, void* stream) {
  NAST_REQUIRE(B >= 0 && T >= 1 && S >= 1 && T <= 256 && S <= 256, "nast_align_viterbi_f32: need 1 <= T,S <= 256 (T=%d S=%d)", T, S);
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(cost0 && cost && origin_length && target_length && allpos, "nast_align_viterbi_f32: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  nast_prof_before(3, s);
  if (S <= 64) {
    const size_t lds = (size_t)T * 4 + (size_t)(T - 1) * S + 16;
    hipLaunchKernelGGL(nast_align_viterbi_kernel<true>, dim3(B), dim3(64), lds, s, cost0, cost, origin_length, target_length,
                       allpos, T, S);
  } else {
    const size_t lds = (size_t)2 * (S + 3) * 4 + (size_t)T * 4 + (size_t)(T - 1) * S + 16;
    auto kern = nast_align_viterbi_kernel<false>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(64), lds, s, cost0, cost, origin_length, target_length, allpos, T, S);
  }
  nast_prof_after(3, s, (double)B * (4.0 * T * S + 4.0 * T + 8.0 * (T + 1) + 16.0));   // SURVEY 8d bytes per sentence
  NAST_CHECK_LAUNCH("nast_align_viterbi_f32");
  return NAST_OK;
}

extern "C" int nast_align_lse_fwd_f32(const float* cost0, const float* cost, const int64_t* origin_length,
                                      const int64_t* target_length, float tau, float* value, float* alpha, int B, int T, int S,
                                      void* stream) {
  NAST_REQUIRE(B >= 0 && T >= 1 && S >= 1 && S <= 4096, "nast_align_lse_fwd_f32: bad sizes");
  NAST_REQUIRE(tau > 0.f, "nast_align_lse_fwd_f32: tau must be > 0");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(cost0 && cost && origin_length && target_length && value && alpha, "nast_align_lse_fwd_f32: NULL pointer");
  nast_prof_before(5, (hipStream_t)stream);
  hipLaunchKernelGGL(nast_align_lse_fwd_kernel, dim3(B), dim3(64), (size_t)2 * (S + 3) * 4, (hipStream_t)stream, cost0, cost,
                     origin_length, target_length, tau, value, alpha, T, S);
  // algorithmic bytes per sentence: the cost table read once, the forward table alpha written once, cost0, value, 2 lengths
  nast_prof_after(5, (hipStream_t)stream, (double)B * (8.0 * T * S + 4.0 * T + 4.0 + 16.0));
  NAST_CHECK_LAUNCH("nast_align_lse_fwd_f32");
  return NAST_OK;
}

extern "C" int nast_align_lse_bwd_f32(const float* cost0, const float* cost, const int64_t* origin_length,
                                      const int64_t* target_length, float tau, const float* value, const float* alpha,
                                      const float* dvalue, float* dcost0, float* dcost, int B, int T, int S, void* stream) {
  NAST_REQUIRE(B >= 0 && T >= 1 && S >= 1 && S <= 4096, "nast_align_lse_bwd_f32: bad sizes");
  NAST_REQUIRE(tau > 0.f, "nast_align_lse_bwd_f32: tau must be > 0");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(cost0 && cost && origin_length && target_length && value && alpha && dvalue && dcost0 && dcost,
               "nast_align_lse_bwd_f32: NULL pointer");
  nast_prof_before(5, (hipStream_t)stream);
  hipLaunchKernelGGL(nast_align_lse_bwd_kernel, dim3(B), dim3(64), (size_t)2 * (S + 3) * 4, (hipStream_t)stream, cost0, cost,
                     origin_length, target_length, tau, value, alpha, dvalue, dcost0, dcost, T, S);
  // cost and alpha read once, dcost written once, cost0 / dcost0, value / dvalue, 2 lengths
  nast_prof_after(5, (hipStream_t)stream, (double)B * (12.0 * T * S + 8.0 * T + 8.0 + 16.0));
  NAST_CHECK_LAUNCH("nast_align_lse_bwd_f32");
  return NAST_OK;
}
