// Row gather/scatter kernels: embedding assembly (style/cls rows + token + position) and the
// LengthPredictor candidate gathers.  One wavefront per d-float row, HBM-bound.
#include "common.h"

__device__ __forceinline__ void row_copy_add(float* __restrict__ dst, const float* __restrict__ a,
                                             const float* __restrict__ b, int d, int lane) {
  if ((d & 3) == 0) {
    for (int c = lane * 4; c < d; c += 256) {
      float4 t = *reinterpret_cast<const float4*>(a + c);
      if (b) {
        const float4 u = *reinterpret_cast<const float4*>(b + c);
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      *reinterpret_cast<float4*>(dst + c) = t;
    }
  } else {
    for (int c = lane; c < d; c += 64) dst[c] = a[c] + (b ? b[c] : 0.f);
  }
}

__global__ __launch_bounds__(256) void nast_embed_fwd_kernel(float* __restrict__ out, int B, int L, int d, int ne,
                                                             const int64_t* __restrict__ ids, const float* __restrict__ tokw,
                                                             const float* __restrict__ tok_rows, const float* __restrict__ posw,
                                                             const float* __restrict__ ew0, const int64_t* __restrict__ ei0,
                                                             const float* __restrict__ ew1, const int64_t* __restrict__ ei1) {
  const int lane = threadIdx.x & 63;
  const int n = ne + L;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * n) return;
  const int b = row / n, p = row % n;
  float* dst = out + row * d;
  if (p < ne) {
    const float* w = p == 0 ? ew0 : ew1;
    const int64_t* ix = p == 0 ? ei0 : ei1;
    row_copy_add(dst, w + (ix ? ix[b] : 0) * d, nullptr, d, lane);
  } else {
    const int i = p - ne;
    const float* src = ids ? tokw + ids[(int64_t)b * L + i] * d : tok_rows + ((int64_t)b * L + i) * d;
    row_copy_add(dst, src, posw + (int64_t)i * d, d, lane);
  }
}

// token rows: scatter-add into the embedding table (hard ids) or dense copy (soft tokens)
__global__ __launch_bounds__(256) void nast_embed_bwd_tok_kernel(const float* __restrict__ dout, int B, int L, int d, int ne,
                                                                 const int64_t* __restrict__ ids, float* __restrict__ d_tokw,
                                                                 float* __restrict__ d_tok_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)B * L) return;
  const int b = r / L, i = r % L;
  const float* g = dout + ((int64_t)b * (ne + L) + ne + i) * d;
  if (ids) {
    if (d_tokw) {
      // rows whose gradient is all zero add nothing: skipped.  Padded positions carry no gradient, and they all hold the
      // same id (<pad>), so their atomics would otherwise pile up on one table row (fp32 atomics on one address serialise).
      bool any = false;
      for (int c = lane; c < d; c += 64) any |= g[c] != 0.f;
      if (__ballot(any) == 0ull) return;
      float* dst = d_tokw + ids[r] * d;
      for (int c = lane; c < d; c += 64) atomicAdd(dst + c, g[c]);
    }
  } else if (d_tok_rows) {
    float* dst = d_tok_rows + r * d;
    for (int c = lane; c < d; c += 64) dst[c] = g[c];
  }
}

// position rows and the extra (style / cls) rows: workgroup = (row position p, chunk of EMB_BPB sentences); every
// wavefront sums its sentences in registers and issues one atomic per column, so an address sees
// B / (EMB_BPB / 4) adds instead of B (fp32 atomics on one address serialise: ~14x slower than spread ones)
#define EMB_BPB 32
__global__ __launch_bounds__(256) void nast_embed_bwd_pos_kernel(const float* __restrict__ dout, int B, int L, int d, int ne,
                                                                 float* __restrict__ d_pos, float* __restrict__ dw0,
                                                                 const int64_t* __restrict__ ei0, float* __restrict__ dw1,
                                                                 const int64_t* __restrict__ ei1) {
  const int p = blockIdx.x;   // 0 .. ne+L-1
  const int n = ne + L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b0 = blockIdx.y * EMB_BPB;
  const int b1 = min(B, b0 + EMB_BPB);
  float* dst = nullptr;
  const int64_t* ix = nullptr;
  if (p >= ne) dst = d_pos ? d_pos + (int64_t)(p - ne) * d : nullptr;
  else { dst = p == 0 ? dw0 : dw1; ix = p == 0 ? ei0 : ei1; }
  if (!dst) return;
  for (int c = lane; c < d; c += 64) {
    if (!ix) {
      float s = 0.f;
      for (int b = b0 + wave; b < b1; b += 4) s += dout[((int64_t)b * n + p) * d + c];
      atomicAdd(dst + c, s);
    } else {
      for (int b = b0 + wave; b < b1; b += 4) atomicAdd(dst + ix[b] * d + c, dout[((int64_t)b * n + p) * d + c]);
    }
  }
}

extern "C" int nast_embed_fwd_f32(float* out, int B, int L, int d, int n_extra, const int64_t* ids, const float* tok_weight,
                                  const float* tok_rows, const float* pos_weight, const float* extra_w0,
                                  const int64_t* extra_idx0, const float* extra_w1, const int64_t* extra_idx1, void* stream) {
  NAST_REQUIRE(B >= 0 && L >= 0 && d >= 1 && n_extra >= 0 && n_extra <= 2, "nast_embed_fwd_f32: bad sizes");
  if (B == 0 || n_extra + L == 0) return NAST_OK;
  NAST_REQUIRE(out && pos_weight, "nast_embed_fwd_f32: NULL pointer");
  NAST_REQUIRE((ids && tok_weight) || tok_rows || L == 0, "nast_embed_fwd_f32: need ids+tok_weight or tok_rows");
  NAST_REQUIRE(n_extra < 1 || extra_w0, "nast_embed_fwd_f32: extra_w0 NULL");
  NAST_REQUIRE(n_extra < 2 || extra_w1, "nast_embed_fwd_f32: extra_w1 NULL");
  const int64_t rows = (int64_t)B * (n_extra + L);
  hipLaunchKernelGGL(nast_embed_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, out, B, L, d, n_extra, ids,
                     tok_weight, tok_rows, pos_weight, extra_w0, extra_idx0, extra_w1, extra_idx1);
  NAST_CHECK_LAUNCH("nast_embed_fwd_f32");
  return NAST_OK;
}

extern "C" int nast_embed_bwd_f32(const float* dout, int B, int L, int d, int n_extra, const int64_t* ids, float* d_tok_weight,
                                  float* d_tok_rows, float* d_pos_weight, float* d_extra_w0, const int64_t* extra_idx0,
                                  float* d_extra_w1, const int64_t* extra_idx1, void* stream) {
  NAST_REQUIRE(B >= 0 && L >= 0 && d >= 1 && n_extra >= 0 && n_extra <= 2, "nast_embed_bwd_f32: bad sizes");
  if (B == 0 || n_extra + L == 0) return NAST_OK;
  NAST_REQUIRE(dout, "nast_embed_bwd_f32: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (L > 0 && ((ids && d_tok_weight) || (!ids && d_tok_rows))) {
    const int64_t rows = (int64_t)B * L;
    hipLaunchKernelGGL(nast_embed_bwd_tok_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, dout, B, L, d, n_extra, ids,
                       d_tok_weight, d_tok_rows);
  }
  hipLaunchKernelGGL(nast_embed_bwd_pos_kernel, dim3(n_extra + L, (B + EMB_BPB - 1) / EMB_BPB), dim3(256), 0, s, dout, B, L, d, n_extra, d_pos_weight,
                     d_extra_w0, extra_idx0, d_extra_w1, extra_idx1);
  NAST_CHECK_LAUNCH("nast_embed_bwd_f32");
  return NAST_OK;
}

// ---------------------------------------------------------------------------------------------
// LengthPredictor candidate gather (transformer.py:471-473, :488-491, :504-505)
//   cand(b,p) = origin_input[b,p] if p == 0 or (p <= S and p-1 < len[b]) else eos
//   out[b,t]  = (use_stay && allpos[b,t] == allpos[b,t-1]) ? unk : cand(b, allpos[b,t] + shift)   (+ pos[t])
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lp_source(const int64_t* __restrict__ allpos, const int64_t* __restrict__ olen, int b, int t,
                                         int T1, int S1, int shift, int use_stay) {
  // returns p >= 0: row of origin_input; -1: eos; -2: unk
  const int64_t a = allpos[(int64_t)b * T1 + t];
  if (use_stay) {
    const int64_t prev = t > 0 ? allpos[(int64_t)b * T1 + t - 1] : -1;
    if (a == prev) return -2;
  }
  const int64_t p = a + shift;
  if (p == 0) return 0;
  if (p > 0 && p < S1 && p - 1 < olen[b]) return (int)p;
  return -1;
}

__global__ __launch_bounds__(256) void nast_lp_gather_fwd_kernel(const float* __restrict__ oin, const int64_t* __restrict__ olen,
                                                                 const int64_t* __restrict__ allpos, const float* __restrict__ eos,
                                                                 const float* __restrict__ unk, const float* __restrict__ posw,
                                                                 float* __restrict__ out, int B, int S1, int T1, int d, int shift,
                                                                 int use_stay) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * T1) return;
  const int b = row / T1, t = row % T1;
  const int p = lp_source(allpos, olen, b, t, T1, S1, shift, use_stay);
  const float* src = p >= 0 ? oin + ((int64_t)b * S1 + p) * d : (p == -1 ? eos : unk);
  row_copy_add(out + row * d, src, posw ? posw + (int64_t)t * d : nullptr, d, lane);
}

// workgroup = (target position t, chunk of LP_BPB sentences).  Gradients of rows gathered from origin_input are
// scattered with atomics (distinct addresses); the shared destinations (eos, unk, pos[t]) are first summed per
// wavefront in registers -> one atomic per column per wavefront instead of one per sentence.
#define LP_BPB 128
__global__ __launch_bounds__(256) void nast_lp_gather_bwd_kernel(const float* __restrict__ dout, const int64_t* __restrict__ olen,
                                                                 const int64_t* __restrict__ allpos, float* __restrict__ d_oin,
                                                                 float* __restrict__ d_eos, float* __restrict__ d_unk,
                                                                 float* __restrict__ d_pos, int B, int S1, int T1, int d, int shift,
                                                                 int use_stay, int pos_skip_row0) {
  // the three shared destinations see ONE atomic per column per workgroup (wavefront sums meet in LDS first): with
  // 128 sentences per workgroup an address takes T1 * B / 128 adds instead of T1 * B / 8 (same-address atomics serialise)
  __shared__ float red[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int b0 = blockIdx.y * LP_BPB;
  constexpr int NB = LP_BPB / 4;
  int src[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int b = b0 + wave + 4 * k;
    src[k] = b < B ? lp_source(allpos, olen, b, t, T1, S1, shift, use_stay) : -3;
  }
  const bool want_pos = d_pos && !(pos_skip_row0 && t == 0);
  for (int c0 = 0; c0 < d; c0 += 64) {
    const int c = c0 + lane;
    float s_eos = 0.f, s_unk = 0.f, s_pos = 0.f;
    if (c < d) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int b = b0 + wave + 4 * k;
        if (src[k] == -3) continue;
        const float v = dout[((int64_t)b * T1 + t) * d + c];
        s_pos += v;
        if (src[k] >= 0) { if (d_oin && v != 0.f) atomicAdd(d_oin + ((int64_t)b * S1 + src[k]) * d + c, v); }   // (padded positions: 0)
        else if (src[k] == -1) s_eos += v;
        else s_unk += v;
      }
    }
    red[0][wave][lane] = s_eos; red[1][wave][lane] = s_unk; red[2][wave][lane] = s_pos;
    __syncthreads();
    if (wave < 3 && c < d) {
      const float v = red[wave][0][lane] + red[wave][1][lane] + red[wave][2][lane] + red[wave][3][lane];
      float* dst = wave == 0 ? d_eos + c : (wave == 1 ? d_unk + c : (want_pos ? d_pos + (int64_t)t * d + c : nullptr));
      if (dst != nullptr && v != 0.f) atomicAdd(dst, v);
    }
    __syncthreads();
  }
}

// d % 4 == 0: the same work with 16-B loads (lane = 4 columns, one pass over 256 columns) and 64 sentences per workgroup:
// 2x the workgroups (the scalar version leaves half of the CUs without one at T1 = 34, B = 512) and 16 independent
// float4 loads in flight per lane.
#define LPV_BPB 64
__global__ __launch_bounds__(256) void nast_lp_gather_bwd_vec_kernel(const float* __restrict__ dout, const int64_t* __restrict__ olen,
                                                                     const int64_t* __restrict__ allpos, float* __restrict__ d_oin,
                                                                     float* __restrict__ d_eos, float* __restrict__ d_unk,
                                                                     float* __restrict__ d_pos, int B, int S1, int T1, int d, int shift,
                                                                     int use_stay, int pos_skip_row0) {
  __shared__ float4 red[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x;
  const int b0 = blockIdx.y * LPV_BPB;
  constexpr int NB = LPV_BPB / 4;
  int src[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int b = b0 + wave + 4 * k;
    src[k] = b < B ? lp_source(allpos, olen, b, t, T1, S1, shift, use_stay) : -3;
  }
  const bool want_pos = d_pos && !(pos_skip_row0 && t == 0);
  for (int c0 = 0; c0 < d; c0 += 256) {
    const int c = c0 + 4 * lane;
    float4 s_eos = make_float4(0.f, 0.f, 0.f, 0.f), s_unk = s_eos, s_pos = s_eos;
    if (c < d) {
      float4 v[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int b = b0 + wave + 4 * k;
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src[k] != -3) v[k] = *reinterpret_cast<const float4*>(dout + ((int64_t)b * T1 + t) * d + c);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        if (src[k] == -3) continue;
        const float4 u = v[k];
        s_pos.x += u.x; s_pos.y += u.y; s_pos.z += u.z; s_pos.w += u.w;
        if (src[k] >= 0) {
          if (d_oin) {
            float* dst = d_oin + ((int64_t)(b0 + wave + 4 * k) * S1 + src[k]) * d + c;
            if (u.x != 0.f) atomicAdd(dst, u.x);          // (padded positions: 0)
            if (u.y != 0.f) atomicAdd(dst + 1, u.y);
            if (u.z != 0.f) atomicAdd(dst + 2, u.z);
            if (u.w != 0.f) atomicAdd(dst + 3, u.w);
          }
        } else if (src[k] == -1) {
          s_eos.x += u.x; s_eos.y += u.y; s_eos.z += u.z; s_eos.w += u.w;
        } else {
          s_unk.x += u.x; s_unk.y += u.y; s_unk.z += u.z; s_unk.w += u.w;
        }
      }
    }
    red[0][wave][lane] = s_eos; red[1][wave][lane] = s_unk; red[2][wave][lane] = s_pos;
    __syncthreads();
    if (wave < 3 && c < d) {
      const float4 a = red[wave][0][lane], b4 = red[wave][1][lane], c4 = red[wave][2][lane], e = red[wave][3][lane];
      const float v0 = a.x + b4.x + c4.x + e.x, v1 = a.y + b4.y + c4.y + e.y, v2 = a.z + b4.z + c4.z + e.z, v3 = a.w + b4.w + c4.w + e.w;
      float* dst = wave == 0 ? d_eos + c : (wave == 1 ? d_unk + c : (want_pos ? d_pos + (int64_t)t * d + c : nullptr));
      if (dst != nullptr) {
        if (v0 != 0.f) atomicAdd(dst, v0);
        if (v1 != 0.f) atomicAdd(dst + 1, v1);
        if (v2 != 0.f) atomicAdd(dst + 2, v2);
        if (v3 != 0.f) atomicAdd(dst + 3, v3);
      }
    }
    __syncthreads();
  }
}

extern "C" int nast_lp_gather_fwd_f32(const float* origin_input, const int64_t* origin_length, const int64_t* allpos,
                                      const float* eos, const float* unk, const float* pos_weight, float* out, int B, int S1,
                                      int T1, int d, int shift, int use_stay, void* stream) {
  NAST_REQUIRE(B >= 0 && S1 >= 1 && T1 >= 1 && d >= 1, "nast_lp_gather_fwd_f32: bad sizes");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(origin_input && origin_length && allpos && eos && unk && out, "nast_lp_gather_fwd_f32: NULL pointer");
  const int64_t rows = (int64_t)B * T1;
  hipLaunchKernelGGL(nast_lp_gather_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, origin_input,
                     origin_length, allpos, eos, unk, pos_weight, out, B, S1, T1, d, shift, use_stay);
  NAST_CHECK_LAUNCH("nast_lp_gather_fwd_f32");
  return NAST_OK;
}

extern "C" int nast_lp_gather_bwd_f32(const float* dout, const int64_t* origin_length, const int64_t* allpos,
                                      float* d_origin_input, float* d_eos, float* d_unk, float* d_pos, int B, int S1, int T1,
                                      int d, int shift, int use_stay, int pos_skip_row0, void* stream) {
  NAST_REQUIRE(B >= 0 && S1 >= 1 && T1 >= 1 && d >= 1, "nast_lp_gather_bwd_f32: bad sizes");
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(dout && origin_length && allpos && d_eos && d_unk, "nast_lp_gather_bwd_f32: NULL pointer");
  const int64_t rows = (int64_t)B * T1;
  (void)rows;
  const bool vec = (d & 3) == 0 && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(d_origin_input) | reinterpret_cast<uintptr_t>(d_eos) |
                                     reinterpret_cast<uintptr_t>(d_unk) | reinterpret_cast<uintptr_t>(d_pos)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(nast_lp_gather_bwd_vec_kernel, dim3(T1, (B + LPV_BPB - 1) / LPV_BPB), dim3(256), 0, (hipStream_t)stream, dout, origin_length,
                       allpos, d_origin_input, d_eos, d_unk, d_pos, B, S1, T1, d, shift, use_stay, pos_skip_row0);
  else
    hipLaunchKernelGGL(nast_lp_gather_bwd_kernel, dim3(T1, (B + LP_BPB - 1) / LP_BPB), dim3(256), 0, (hipStream_t)stream, dout, origin_length,
                       allpos, d_origin_input, d_eos, d_unk, d_pos, B, S1, T1, d, shift, use_stay, pos_skip_row0);
  NAST_CHECK_LAUNCH("nast_lp_gather_bwd_f32");
  return NAST_OK;
}
