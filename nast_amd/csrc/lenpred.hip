// LengthPredictor greedy skip decoding, one step (transformer.py:549-567): 4-way skip logits, first-step
// masking, argmax, position update and the gather of the next decoder input -- one wavefront per sentence,
// replacing ~15 tiny ATen launches (gather/einsum/linear/max/min/compare) per decoding step.
#include <stdlib.h>

#include "common.h"

__global__ __launch_bounds__(256) void nast_lp_step_kernel(const float* __restrict__ qh, const float* __restrict__ hid,
                                                           const float* __restrict__ KC, const float* __restrict__ kunk,
                                                           const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                                                           const float* __restrict__ cand_in, const float* __restrict__ unk,
                                                           const float* __restrict__ pos_w, const int64_t* __restrict__ olen_,
                                                           int64_t* __restrict__ now_pos, int64_t* __restrict__ allpos,
                                                           float* __restrict__ x_next, const float* __restrict__ norm_w,
                                                           const float* __restrict__ norm_b, const float* __restrict__ sample_u,
                                                           int B, int S1, int d, int step, int ldp, float sqrt_d,
                                                           int write_next) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int NC = S1 + 5;                      // rows of the candidate table (transformer.py:472-473)
  const int64_t now = now_pos[b];
  const float* q = qh + (int64_t)b * d;
  const float* hrow = hid + (int64_t)b * d;
  float dot[4] = {0.f, 0.f, 0.f, 0.f}, lin[4] = {0.f, 0.f, 0.f, 0.f};
  float mu = 0.f, rs = 1.f;
  if (norm_w) {   // `hid` is the un-normalised decoder output: apply HiddenDecoder.norm here (transformer.py:376)
    float s1 = 0.f;
    for (int c = lane; c < d; c += 64) s1 += hrow[c];
    mu = wave_sum(s1) / d;
    float s2 = 0.f;
    for (int c = lane; c < d; c += 64) { const float t = hrow[c] - mu; s2 += t * t; }
    rs = rsqrtf(wave_sum(s2) / d + 1e-6f);
  }
  for (int c = lane; c < d; c += 64) {
    const float qv = q[c];
    const float hv = norm_w ? (hrow[c] - mu) * rs * norm_w[c] + norm_b[c] : hrow[c];
    dot[0] = fmaf(qv, kunk[c], dot[0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) dot[j] = fmaf(qv, KC[((int64_t)b * NC + now + j) * d + c], dot[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) lin[j] = fmaf(hv, lin_w[j * d + c], lin[j]);
  }
  float logit[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) logit[j] = wave_sum(dot[j]) / sqrt_d + (wave_sum(lin[j]) + lin_b[j]);
  if (step == 0) {                             // :553-555
    const int64_t ol = olen_[b];
    logit[0] = -1e9f;
#pragma unroll
    for (int j = 0; j < 4; ++j) logit[j] = logit[j] - ((j >= 1 && j - 1 >= ol) ? 1.f : 0.f) * 1e9f;
  }
  int delta = 0;
  float best = logit[0];
#pragma unroll
  for (int j = 1; j < 4; ++j)
    if (logit[j] > best) { best = logit[j]; delta = j; }   // first maximum wins (:559)
  if (sample_u) {                              // :560-561  multinomial(softmax(logits)) by inverse CDF on this step's uniform
    float p[4], tot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[j] = expf(logit[j] - best); tot += p[j]; }
    const float u = sample_u[b] * tot;
    float c = 0.f;
    int pick = -1, last = delta;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      c += p[j];
      if (p[j] > 0.f) last = j;
      if (pick < 0 && p[j] > 0.f && c > u) pick = j;
    }
    delta = pick < 0 ? last : pick;
  }
  int64_t nxt = now + delta;
  if (nxt > S1) nxt = S1;                      // :565  (origin_seqlen + 1)
  if (lane == 0) {
    now_pos[b] = nxt;
    allpos[(int64_t)b * ldp + step + 1] = nxt;
  }
  if (write_next) {
    const float* src = (delta == 0) ? unk : cand_in + ((int64_t)b * NC + nxt) * d;   // :535-537
    const float* pw = pos_w + (int64_t)(step + 1) * d;
    float* dst = x_next + (int64_t)b * d;
    for (int c = lane; c < d; c += 64) dst[c] = src[c] + pw[c];
  }
}

extern "C" int nast_lp_step_f32(const float* qh, const float* hid, const float* KC, const float* kunk, const float* lin_w,
                                const float* lin_b, const float* cand_in, const float* unk, const float* pos_w,
                                const int64_t* origin_length, int64_t* now_pos, int64_t* allpos, float* x_next,
                                const float* norm_w, const float* norm_b, const float* sample_u, int B, int S1,
                                int d, int step, int ldp, int write_next, void* stream) {
  NAST_REQUIRE(B >= 0 && S1 >= 1 && d >= 1 && step >= 0 && ldp >= step + 2, "nast_lp_step_f32: bad sizes");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(qh && hid && KC && kunk && lin_w && lin_b && cand_in && unk && pos_w && origin_length && now_pos && allpos,
               "nast_lp_step_f32: NULL pointer");
  NAST_REQUIRE(!write_next || x_next, "nast_lp_step_f32: x_next NULL");
  hipLaunchKernelGGL(nast_lp_step_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, qh, hid, KC, kunk, lin_w, lin_b,
                     cand_in, unk, pos_w, origin_length, now_pos, allpos, x_next, norm_w, norm_b, sample_u, B, S1, d, step, ldp, sqrtf((float)d),
                     write_next);
  NAST_CHECK_LAUNCH("nast_lp_step_f32");
  return NAST_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Skip logits of the teacher-forced branch (transformer.py:504-507):
//   out[b,t,0] = qh[b,t] . kunk / sqrt(d),   out[b,t,j] = qh[b,t] . KC[b, allpos[b,t] + j] / sqrt(d)   (j = 1..3)
// (wk is linear, so wk(candidates_output)[b,t,j] is a row of KC = wk(candidates_input) -- the [B,S+2,4,d] table of
// the reference is never built).  One wavefront per (sentence, position).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nast_lp_logits_fwd_kernel(const float* __restrict__ qh, const float* __restrict__ KC,
                                                                 const float* __restrict__ kunk, const int64_t* __restrict__ allpos,
                                                                 float* __restrict__ out, int B, int T1, int NC, int d, float inv) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * T1) return;
  const int b = row / T1;
  const int64_t a = allpos[row];
  const float* q = qh + row * d;
  float dot[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = lane; c < d; c += 64) {
    const float qv = q[c];
    dot[0] = fmaf(qv, kunk[c], dot[0]);
#pragma unroll
    for (int j = 1; j < 4; ++j) dot[j] = fmaf(qv, KC[((int64_t)b * NC + a + j) * d + c], dot[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) dot[j] = wave_sum(dot[j]) * inv;
  if (lane == 0) *reinterpret_cast<float4*>(out + row * 4) = make_float4(dot[0], dot[1], dot[2], dot[3]);
}

// workgroup = (position t, chunk of 32 sentences): dqh written, dKC scattered with atomics (distinct rows), the shared
// dkunk summed per wavefront first (one atomic per column per wavefront)
#define LPL_BPB 128
__global__ __launch_bounds__(256) void nast_lp_logits_bwd_kernel(const float* __restrict__ g, const float* __restrict__ qh,
                                                                 const float* __restrict__ KC, const float* __restrict__ kunk,
                                                                 const int64_t* __restrict__ allpos, float* __restrict__ dqh,
                                                                 float* __restrict__ dKC, float* __restrict__ dkunk, int B, int T1,
                                                                 int NC, int d, float inv) {
  __shared__ float red[4][64];      // the kunk gradient: one atomic per column per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int b0 = blockIdx.y * LPL_BPB;
  for (int c0 = 0; c0 < d; c0 += 64) {
    const int c = c0 + lane;
    float s_unk = 0.f;
    if (c < d) {
      const float ku = kunk[c];
      for (int b = b0 + wave; b < B && b < b0 + LPL_BPB; b += 4) {
        const int64_t row = (int64_t)b * T1 + t;
        const float4 gv = *reinterpret_cast<const float4*>(g + row * 4);
        if (gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f) {   // positions past the sentence carry no loss
          dqh[row * d + c] = 0.f;
          continue;
        }
        const int64_t a = allpos[row];
        const float qv = qh[row * d + c];
        const float* k1 = KC + ((int64_t)b * NC + a + 1) * d + c;
        dqh[row * d + c] = inv * (gv.x * ku + gv.y * k1[0] + gv.z * k1[d] + gv.w * k1[2 * d]);
        s_unk = fmaf(gv.x, qv, s_unk);
        float* dk1 = dKC + ((int64_t)b * NC + a + 1) * d + c;
        atomicAdd(dk1, inv * gv.y * qv);
        atomicAdd(dk1 + d, inv * gv.z * qv);
        atomicAdd(dk1 + 2 * d, inv * gv.w * qv);
      }
    }
    red[wave][lane] = s_unk;
    __syncthreads();
    if (wave == 0 && c < d) {
      const float v = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
      if (v != 0.f) atomicAdd(dkunk + c, inv * v);
    }
    __syncthreads();
  }
}

// d % 4 == 0: 16-B accesses (lane = 4 columns, one pass over 256 columns) and 64 sentences per workgroup -- twice the
// workgroups (T1 x B/128 = 136 leaves half of the CUs idle at B = 512) and four columns per dependent allpos -> KC load chain.
#define LPLV_BPB 64
__global__ __launch_bounds__(256) void nast_lp_logits_bwd_vec_kernel(const float* __restrict__ g, const float* __restrict__ qh,
                                                                     const float* __restrict__ KC, const float* __restrict__ kunk,
                                                                     const int64_t* __restrict__ allpos, float* __restrict__ dqh,
                                                                     float* __restrict__ dKC, float* __restrict__ dkunk, int B, int T1,
                                                                     int NC, int d, float inv) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int b0 = blockIdx.y * LPLV_BPB;
  for (int c0 = 0; c0 < d; c0 += 256) {
    const int c = c0 + 4 * lane;
    float4 s_unk = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < d) {
      const float4 ku = *reinterpret_cast<const float4*>(kunk + c);
#pragma unroll 4
      for (int b = b0 + wave; b < B && b < b0 + LPLV_BPB; b += 4) {
        const int64_t row = (int64_t)b * T1 + t;
        const float4 gv = *reinterpret_cast<const float4*>(g + row * 4);
        float4* dq = reinterpret_cast<float4*>(dqh + row * d + c);
        if (gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f) {   // positions past the sentence carry no loss
          *dq = make_float4(0.f, 0.f, 0.f, 0.f);
          continue;
        }
        const int64_t a = allpos[row];
        const float4 qv = *reinterpret_cast<const float4*>(qh + row * d + c);
        const float* k1 = KC + ((int64_t)b * NC + a + 1) * d + c;
        const float4 ka = *reinterpret_cast<const float4*>(k1), kb = *reinterpret_cast<const float4*>(k1 + d),
                     kc = *reinterpret_cast<const float4*>(k1 + 2 * d);
        float4 o;
        o.x = inv * (gv.x * ku.x + gv.y * ka.x + gv.z * kb.x + gv.w * kc.x);
        o.y = inv * (gv.x * ku.y + gv.y * ka.y + gv.z * kb.y + gv.w * kc.y);
        o.z = inv * (gv.x * ku.z + gv.y * ka.z + gv.z * kb.z + gv.w * kc.z);
        o.w = inv * (gv.x * ku.w + gv.y * ka.w + gv.z * kb.w + gv.w * kc.w);
        *dq = o;
        s_unk.x = fmaf(gv.x, qv.x, s_unk.x); s_unk.y = fmaf(gv.x, qv.y, s_unk.y);
        s_unk.z = fmaf(gv.x, qv.z, s_unk.z); s_unk.w = fmaf(gv.x, qv.w, s_unk.w);
        float* dk1 = dKC + ((int64_t)b * NC + a + 1) * d + c;
        const float q4[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          atomicAdd(dk1 + e, inv * gv.y * q4[e]);
          atomicAdd(dk1 + d + e, inv * gv.z * q4[e]);
          atomicAdd(dk1 + 2 * d + e, inv * gv.w * q4[e]);
        }
      }
    }
    red[wave][lane] = s_unk;
    __syncthreads();
    if (wave == 0 && c < d) {
      const float4 a = red[0][lane], b4 = red[1][lane], c4 = red[2][lane], e4 = red[3][lane];
      const float v[4] = {a.x + b4.x + c4.x + e4.x, a.y + b4.y + c4.y + e4.y, a.z + b4.z + c4.z + e4.z, a.w + b4.w + c4.w + e4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] != 0.f) atomicAdd(dkunk + c + e, inv * v[e]);
    }
    __syncthreads();
  }
}

// One workgroup per sentence, no atomics on dKC: the sentence's qh rows, cotangents and allpos are staged in LDS, dqh rows
// are written as before, and every candidate row p GATHERS its contributions (sum over the positions t with
// 1 <= p - allpos[t] <= 3, ascending t: deterministic) and is read-modified-written once by the one wavefront that owns it.
// The atomic versions issue 3 d float atomics per (sentence, position) into lines that are not in L2 (13 M at B = 512,
// T1 = 34, d = 256: 124 us); dkunk keeps one atomic per column per workgroup.
__global__ __launch_bounds__(256) void nast_lp_logits_bwd_gather_kernel(const float* __restrict__ g, const float* __restrict__ qh,
                                                                        const float* __restrict__ KC, const float* __restrict__ kunk,
                                                                        const int64_t* __restrict__ allpos, float* __restrict__ dqh,
                                                                        float* __restrict__ dKC, float* __restrict__ dkunk, int B, int T1,
                                                                        int NC, int d, float inv) {
  extern __shared__ __attribute__((aligned(16))) float lp_smem[];
  float* q_s = lp_smem;                          // [T1][d]
  float* g_s = q_s + (size_t)T1 * d;             // [T1][4]
  float* red = g_s + (size_t)T1 * 4;             // [4][d]
  int* a_s = reinterpret_cast<int*>(red + 4 * (size_t)d);   // [T1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int d4 = d >> 2;
  for (int i = threadIdx.x; i < T1 * d4; i += 256)
    reinterpret_cast<float4*>(q_s)[i] = reinterpret_cast<const float4*>(qh + (int64_t)b * T1 * d)[i];
  for (int i = threadIdx.x; i < T1; i += 256) {
    reinterpret_cast<float4*>(g_s)[i] = reinterpret_cast<const float4*>(g + (int64_t)b * T1 * 4)[i];
    a_s[i] = (int)allpos[(int64_t)b * T1 + i];
  }
  __syncthreads();
  // dqh[t] = inv * (g0 kunk + g1 KC[a+1] + g2 KC[a+2] + g3 KC[a+3]);  dkunk partial = sum_t g0 qh[t]
  for (int c0 = 0; c0 < d; c0 += 256) {
    const int c = c0 + 4 * lane;
    float4 s_unk = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < d) {
      const float4 ku = *reinterpret_cast<const float4*>(kunk + c);
      for (int t = wave; t < T1; t += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(g_s + 4 * t);
        float4* dq = reinterpret_cast<float4*>(dqh + ((int64_t)b * T1 + t) * d + c);
        if (gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f) {   // positions past the sentence carry no loss
          *dq = make_float4(0.f, 0.f, 0.f, 0.f);
          continue;
        }
        const float* k1 = KC + ((int64_t)b * NC + a_s[t] + 1) * d + c;
        const float4 ka = *reinterpret_cast<const float4*>(k1), kb = *reinterpret_cast<const float4*>(k1 + d),
                     kc = *reinterpret_cast<const float4*>(k1 + 2 * d);
        float4 o;
        o.x = inv * (gv.x * ku.x + gv.y * ka.x + gv.z * kb.x + gv.w * kc.x);
        o.y = inv * (gv.x * ku.y + gv.y * ka.y + gv.z * kb.y + gv.w * kc.y);
        o.z = inv * (gv.x * ku.z + gv.y * ka.z + gv.z * kb.z + gv.w * kc.z);
        o.w = inv * (gv.x * ku.w + gv.y * ka.w + gv.z * kb.w + gv.w * kc.w);
        *dq = o;
        const float4 qv = *reinterpret_cast<const float4*>(q_s + (size_t)t * d + c);
        s_unk.x = fmaf(gv.x, qv.x, s_unk.x); s_unk.y = fmaf(gv.x, qv.y, s_unk.y);
        s_unk.z = fmaf(gv.x, qv.z, s_unk.z); s_unk.w = fmaf(gv.x, qv.w, s_unk.w);
      }
      *reinterpret_cast<float4*>(red + (size_t)wave * d + c) = s_unk;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += 256) {
    const float v = red[c] + red[d + c] + red[2 * (size_t)d + c] + red[3 * (size_t)d + c];
    if (v != 0.f) atomicAdd(dkunk + c, inv * v);
  }
  // dKC[p] += inv * sum_{t : 1 <= p - a[t] <= 3} g[t][p - a[t]] qh[t]
  for (int p = wave; p < NC; p += 4) {
    for (int c0 = 0; c0 < d; c0 += 256) {
      const int c = c0 + 4 * lane;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      bool any = false;
      for (int t = 0; t < T1; ++t) {
        const int j = p - a_s[t];
        if (j < 1 || j > 3) continue;              // (uniform for the wavefront)
        const float gj = g_s[4 * t + j];
        if (gj == 0.f) continue;
        any = true;
        if (c < d) {
          const float4 qv = *reinterpret_cast<const float4*>(q_s + (size_t)t * d + c);
          acc.x = fmaf(gj, qv.x, acc.x); acc.y = fmaf(gj, qv.y, acc.y); acc.z = fmaf(gj, qv.z, acc.z); acc.w = fmaf(gj, qv.w, acc.w);
        }
      }
      if (any && c < d) {
        float4* dst = reinterpret_cast<float4*>(dKC + ((int64_t)b * NC + p) * d + c);
        float4 o = *dst;
        o.x += inv * acc.x; o.y += inv * acc.y; o.z += inv * acc.z; o.w += inv * acc.w;
        *dst = o;
      }
    }
  }
}

extern "C" int nast_lp_logits_fwd_f32(const float* qh, const float* KC, const float* kunk, const int64_t* allpos, float* out,
                                      int B, int T1, int NC, int d, void* stream) {
  NAST_REQUIRE(B >= 0 && T1 >= 1 && NC >= 4 && d >= 1, "nast_lp_logits_fwd_f32: bad sizes");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(qh && KC && kunk && allpos && out, "nast_lp_logits_fwd_f32: NULL pointer");
  const int64_t rows = (int64_t)B * T1;
  hipLaunchKernelGGL(nast_lp_logits_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, qh, KC, kunk, allpos, out,
                     B, T1, NC, d, 1.0f / sqrtf((float)d));
  NAST_CHECK_LAUNCH("nast_lp_logits_fwd_f32");
  return NAST_OK;
}

extern "C" int nast_lp_logits_bwd_f32(const float* g, const float* qh, const float* KC, const float* kunk, const int64_t* allpos,
                                      float* dqh, float* dKC, float* dkunk, int B, int T1, int NC, int d, void* stream) {
  NAST_REQUIRE(B >= 0 && T1 >= 1 && NC >= 4 && d >= 1, "nast_lp_logits_bwd_f32: bad sizes");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(g && qh && KC && kunk && allpos && dqh && dKC && dkunk, "nast_lp_logits_bwd_f32: NULL pointer");
  const bool vec = (d & 3) == 0 && ((reinterpret_cast<uintptr_t>(qh) | reinterpret_cast<uintptr_t>(KC) | reinterpret_cast<uintptr_t>(kunk) |
                                     reinterpret_cast<uintptr_t>(dqh) | reinterpret_cast<uintptr_t>(dKC) | reinterpret_cast<uintptr_t>(dkunk)) & 15) == 0;
  const size_t gather_lds = ((size_t)T1 * d + (size_t)T1 * 4 + 4 * (size_t)d) * sizeof(float) + (size_t)T1 * sizeof(int);
  static const bool gather_on = [] { const char* e = getenv("NAST_LP_GATHER_BWD"); return e ? atoi(e) != 0 : true; }();   // tuning aid
  if (vec && gather_on && gather_lds <= 64 * 1024 && (reinterpret_cast<uintptr_t>(g) & 15) == 0)
    hipLaunchKernelGGL(nast_lp_logits_bwd_gather_kernel, dim3(B), dim3(256), gather_lds, (hipStream_t)stream, g, qh, KC, kunk, allpos, dqh,
                       dKC, dkunk, B, T1, NC, d, 1.0f / sqrtf((float)d));
  else if (vec)
    hipLaunchKernelGGL(nast_lp_logits_bwd_vec_kernel, dim3(T1, (B + LPLV_BPB - 1) / LPLV_BPB), dim3(256), 0, (hipStream_t)stream, g, qh, KC,
                       kunk, allpos, dqh, dKC, dkunk, B, T1, NC, d, 1.0f / sqrtf((float)d));
  else
    hipLaunchKernelGGL(nast_lp_logits_bwd_kernel, dim3(T1, (B + LPL_BPB - 1) / LPL_BPB), dim3(256), 0, (hipStream_t)stream, g, qh, KC,
                       kunk, allpos, dqh, dKC, dkunk, B, T1, NC, d, 1.0f / sqrtf((float)d));
  NAST_CHECK_LAUNCH("nast_lp_logits_bwd_f32");
  return NAST_OK;
}
