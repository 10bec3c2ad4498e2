// Fused masked multi-head attention for short sequences on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// One workgroup per (sentence, head), operands staged once in LDS; no n x n tensor reaches HBM.
// scaled_attention (transformer.py:296-303) is  S = Q K^T * scale, masked_fill(-inf), P = softmax(S), O = P V.
//
// Orientation trick: the scores are produced TRANSPOSED, S^T[j][i] = sum_c K[j][c] Q[i][c]  (A = K rows,
// B = Q rows).  In the 16x16 accumulator layout (col = lane&15, row = 4*(lane>>4) + reg) a lane then owns ONE
// query i and 4 keys per tile, so
//   * the softmax over keys is a per-lane reduction plus two shuffles (xor 16, 32),
//   * the accumulator registers ARE the A operand of the next product O[i][c] = sum_j P[i][j] V[j][c]
//     (A[row = lane&15][k = lane>>4]): register r of key tile jt supplies k = 16 jt + 4 (lane>>4) + r, and the
//     B operand (V) is fetched in that same k order -- any k permutation shared by A and B is a valid product.
// Operand rows are read as 16-byte pieces: lane (row r = lane&15, g = lane>>4) holds columns 16 q + 4 g + s of
// its row for MFMA step 4 q + s (again a k permutation shared by both operands).
//
// Backward recomputes P from the saved log-sum-exp.  dQ uses the transposed orientation (lane = query);
// dK / dV need lane = key, so S and dP are recomputed with the operands swapped (A = Q / dO rows, B = K / V rows)
// instead of transposing the score tiles: 7 small products per (sentence, head), all on MFMA.
//
// Limits: nq, nk <= 64 and dk in {8, 16, 32, 64}; longer sequences use the LDS kernels in attention.hip.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int DK>
__device__ __forceinline__ void load_frag(float (&f)[DK / 4], const float* __restrict__ row, int g) {
  if constexpr (DK == 8) {
    const float2 t = *reinterpret_cast<const float2*>(row + 2 * g);
    f[0] = t.x; f[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < DK / 16; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
      f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
    }
  }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float xor_sum_g(float v) {   // over the 4 lane groups g = lane>>4
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float xor_max_g(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float xor_sum_15(float v) {  // over the 16 lanes of a group (lane&15)
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct AttnArgs {
  const float *q, *k, *v, *dout, *lse_in;
  const uint8_t* mask;
  float *out, *lse, *dq, *dk, *dv;
  int BH, h, nq, nk, ldq, ldk, ldv, ldo;
  int64_t q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs;
  float scale;
};

// ---------------------------------------------------------------------------------------------------------
// One 256-thread workgroup per (sentence, head).  K and V (backward: also Q and dO) are staged once in LDS as
// [16*NT][DK+4] images (rows past the sequence end zero-filled); every MFMA operand then comes from LDS:
//   * row fragments (A/B operands indexed [row = lane&15][k-slot = lane>>4]) with one ds_read_b128 per 4 steps
//     -- the DK+4 row stride puts the 16 rows of a read on 16 distinct 4-bank groups (conflict free),
//   * column operands (B[k = row 16 t + 4 g + r][n = channel 16 ct + lane&15]) with ds_read_b32, 16 consecutive
//     floats per lane group.
// The 16-row query tiles (forward, dQ) or key tiles (dK/dV) are dealt to the 4 wavefronts.
// ---------------------------------------------------------------------------------------------------------
template <int DK>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src, int64_t ld, int rows, int rows_pad) {
  constexpr int LD = DK + 4, V4 = DK / 4;
  for (int c = threadIdx.x; c < rows_pad * V4; c += 256) {
    const int r = c / V4, q = (c % V4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + q);
    *reinterpret_cast<float4*>(dst + r * LD + q) = v;
  }
}

template <int DK>
__device__ __forceinline__ void lds_frag(float (&f)[DK / 4], const float* __restrict__ row, int g) {
  if constexpr (DK == 8) {
    const float2 t = *reinterpret_cast<const float2*>(row + 2 * g);
    f[0] = t.x; f[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < DK / 16; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(row + 16 * q + 4 * g);
      f[4 * q] = t.x; f[4 * q + 1] = t.y; f[4 * q + 2] = t.z; f[4 * q + 3] = t.w;
    }
  }
}

template <int DK, int NT>
__global__ __launch_bounds__(256) void nast_mha_fwd_mfma_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16, LD = DK + 4;
  __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x;
  const int b = bh / P.h, head = bh % P.h;
  const int nq = P.nq, nk = P.nk;
  const float* Qb = P.q + b * P.q_bs + head * DK;
  float* Ob = P.out + b * P.o_bs + head * DK;
  const int nqt = (nq + 15) / 16;
  float qf[NS];                                  // the first query tile of this wavefront is fetched before the staging
  {
    const int i0 = 16 * wave + r15;
    load_frag<DK>(qf, Qb + (int64_t)(i0 < nq ? i0 : nq - 1) * P.ldq, g);
  }
  stage_rows<DK>(Ks, P.k + b * P.k_bs + head * DK, P.ldk, nk, NT * 16);
  stage_rows<DK>(Vs, P.v + b * P.v_bs + head * DK, P.ldv, nk, NT * 16);
  __syncthreads();
  for (int it = wave; it < nqt; it += 4) {
    const int i = 16 * it + r15;                 // this lane's query
    const int ic = i < nq ? i : nq - 1;
    if (it != wave) load_frag<DK>(qf, Qb + (int64_t)ic * P.ldq, g);
    f32x4 st[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      float kf[NS];
      lds_frag<DK>(kf, Ks + (16 * jt + r15) * LD, g);
      st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) st[jt] = mfma4(kf[s], qf[s], st[jt]);
    }
    // softmax over keys j = 16 jt + 4 g + r for query i
    const uint8_t* mrow = P.mask ? P.mask + b * P.mask_bs + (int64_t)ic * P.mask_qs : nullptr;
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * g + r;
        float s = st[jt][r] * P.scale;
        if (j >= nk || (mrow && mrow[j < nk ? j : 0])) s = -INFINITY;
        st[jt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = xor_max_g(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(st[jt][r] - mx);
        st[jt][r] = e;
        sum += e;
      }
    sum = xor_sum_g(sum);
    const float inv = 1.f / sum;
    if (g == 0 && i < nq) P.lse[(int64_t)bh * nq + i] = mx + __logf(sum);
    // O[i][c] = sum_j P[i][j] V[j][c]: A = the score registers, B = V[j = 16 jt + 4 g + r][16 ct + lane&15]
    f32x4 o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = Vs + (16 * jt + 4 * g + r) * LD;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          const float bv = (DK >= 16 || c < DK) ? vr[c < DK ? c : 0] : 0.f;
          o[ct] = mfma4(st[jt][r], bv, o[ct]);
        }
      }
    // the result tile: col = channel 16 ct + lane&15, row = query 16 it + 4 g + r; 1/sum belongs to the ROW's query
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = 16 * it + 4 * g + r;
      const float invr = __shfl(inv, 4 * g + r, 64);   // lane (4 g + r) of group 0 holds query 16 it + 4 g + r
      if (io < nq) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          if (DK >= 16 || c < DK) Ob[(int64_t)io * P.ldo + c] = o[ct][r] * invr;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
template <int DK, int NT>
__global__ __launch_bounds__(256) void nast_mha_bwd_mfma_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16, LD = DK + 4;
  __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Qs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Gs[NT * 16 * LD];
  __shared__ float dlt[NT * 16];                // delta[i] = sum_j P[i][j] dP[i][j]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x;
  const int b = bh / P.h, head = bh % P.h;
  const int nq = P.nq, nk = P.nk;
  float* dQb = P.dq + b * P.q_bs + head * DK;
  float* dKb = P.dk + b * P.k_bs + head * DK;
  float* dVb = P.dv + b * P.v_bs + head * DK;
  const float* lse = P.lse_in + (int64_t)bh * nq;
  const float scale = P.scale;
  const int nqt = (nq + 15) / 16, nkt = (nk + 15) / 16;
  stage_rows<DK>(Ks, P.k + b * P.k_bs + head * DK, P.ldk, nk, NT * 16);
  stage_rows<DK>(Vs, P.v + b * P.v_bs + head * DK, P.ldv, nk, NT * 16);
  stage_rows<DK>(Qs, P.q + b * P.q_bs + head * DK, P.ldq, nq, NT * 16);
  stage_rows<DK>(Gs, P.dout + b * P.o_bs + head * DK, P.ldo, nq, NT * 16);
  __syncthreads();

  // ---- phase 1: lane = query.  dQ[i][c] = scale * sum_j dS[i][j] K[j][c] -----------------------------------
  for (int it = wave; it < nqt; it += 4) {
    const int i = 16 * it + r15;
    const int ic = i < nq ? i : nq - 1;
    float qf[NS], gf[NS];
    lds_frag<DK>(qf, Qs + (16 * it + r15) * LD, g);
    lds_frag<DK>(gf, Gs + (16 * it + r15) * LD, g);
    const float l = lse[ic];
    const uint8_t* mrow = P.mask ? P.mask + b * P.mask_bs + (int64_t)ic * P.mask_qs : nullptr;
    f32x4 st[NT], dp[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      float kf[NS], vf[NS];
      lds_frag<DK>(kf, Ks + (16 * jt + r15) * LD, g);
      lds_frag<DK>(vf, Vs + (16 * jt + r15) * LD, g);
      st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        st[jt] = mfma4(kf[s], qf[s], st[jt]);
        dp[jt] = mfma4(vf[s], gf[s], dp[jt]);
      }
    }
    float dsum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * g + r;
        const bool dead = j >= nk || i >= nq || (mrow && mrow[j < nk ? j : 0]);
        const float p = dead ? 0.f : __expf(st[jt][r] * scale - l);
        st[jt][r] = p;
        dsum = fmaf(p, dp[jt][r], dsum);
      }
    dsum = xor_sum_g(dsum);
    if (g == 0) dlt[16 * it + r15] = dsum;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) st[jt][r] = st[jt][r] * (dp[jt][r] - dsum) * scale;   // dS * scale
    f32x4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* kr = Ks + (16 * jt + 4 * g + r) * LD;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          const float bv = (DK >= 16 || c < DK) ? kr[c < DK ? c : 0] : 0.f;
          acc[ct] = mfma4(st[jt][r], bv, acc[ct]);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = 16 * it + 4 * g + r;
      if (io < nq) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          if (DK >= 16 || c < DK) dQb[(int64_t)io * P.ldq + c] = acc[ct][r];
        }
      }
    }
  }
  __syncthreads();   // dlt is read by other wavefronts below

  // ---- phase 2: lane = key.  dK[j][c] = scale * sum_i dS[i][j] Q[i][c],  dV[j][c] = sum_i P[i][j] dO[i][c] ----
  for (int jt = wave; jt < nkt; jt += 4) {
    const int j = 16 * jt + r15;                 // this lane's key
    const int jc = j < nk ? j : nk - 1;
    float kf[NS], vf[NS];
    lds_frag<DK>(kf, Ks + (16 * jt + r15) * LD, g);
    lds_frag<DK>(vf, Vs + (16 * jt + r15) * LD, g);
    f32x4 accK[NCT], accV[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      accK[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      accV[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      if (it < nqt) {
        // S[i][j] and dP[i][j] with row = query 16 it + 4 g + r, col = key j
        float qf[NS], gf[NS];
        lds_frag<DK>(qf, Qs + (16 * it + r15) * LD, g);
        lds_frag<DK>(gf, Gs + (16 * it + r15) * LD, g);
        f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, d4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          s4 = mfma4(qf[s], kf[s], s4);
          d4 = mfma4(gf[s], vf[s], d4);
        }
        f32x4 p4, ds4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * it + 4 * g + r;
          const int ic = i < nq ? i : nq - 1;
          const bool dead = i >= nq || j >= nk || (P.mask && P.mask[b * P.mask_bs + (int64_t)ic * P.mask_qs + jc]);
          const float p = dead ? 0.f : __expf(s4[r] * scale - lse[ic]);
          p4[r] = p;
          ds4[r] = p * (d4[r] - dlt[16 * it + 4 * g + r]) * scale;
        }
        // A = dS / P registers (row = key lane, k = query 16 it + 4 g + r), B = Q / dO [query][channel]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* qr = Qs + (16 * it + 4 * g + r) * LD;
          const float* gr = Gs + (16 * it + 4 * g + r) * LD;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const int c = 16 * ct + r15;
            const bool cok = DK >= 16 || c < DK;
            const float bq = cok ? qr[c < DK ? c : 0] : 0.f;
            const float bg = cok ? gr[c < DK ? c : 0] : 0.f;
            accK[ct] = mfma4(ds4[r], bq, accK[ct]);
            accV[ct] = mfma4(p4[r], bg, accV[ct]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jo = 16 * jt + 4 * g + r;
      if (jo < nk) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          if (DK >= 16 || c < DK) {
            dKb[(int64_t)jo * P.ldk + c] = accK[ct][r];
            dVb[(int64_t)jo * P.ldv + c] = accV[ct][r];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Forward, one WAVEFRONT per (sentence, head), operands straight from global memory (no LDS, no barrier): measured
// faster than the workgroup-cooperative LDS variant above for the forward pass (15.7 vs 21.7 us at 2048 heads of
// 33 x 33 x 64), because a forward tile has too little arithmetic to amortise the staging round trip.
// ---------------------------------------------------------------------------------------------------------
template <int DK, int NKT>
__global__ __launch_bounds__(256) void nast_mha_fwd_wave_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16;
  const int lane = threadIdx.x & 63, r15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= P.BH) return;
  const int b = bh / P.h, head = bh % P.h;
  const int nq = P.nq, nk = P.nk;
  const float* Kb = P.k + b * P.k_bs + head * DK;
  const float* Vb = P.v + b * P.v_bs + head * DK;
  const float* Qb = P.q + b * P.q_bs + head * DK;
  float* Ob = P.out + b * P.o_bs + head * DK;
  // K rows of all key tiles stay in registers
  float kf[NKT][NS];
#pragma unroll
  for (int jt = 0; jt < NKT; ++jt) {
    int j = 16 * jt + r15;
    j = j < nk ? j : nk - 1;
    load_frag<DK>(kf[jt], Kb + (int64_t)j * P.ldk, g);
  }
  const int nqt = (nq + 15) / 16;
  for (int it = 0; it < nqt; ++it) {
    const int i = 16 * it + r15;                 // this lane's query
    const int ic = i < nq ? i : nq - 1;
    float qf[NS];
    load_frag<DK>(qf, Qb + (int64_t)ic * P.ldq, g);
    f32x4 st[NKT];
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt) {
      st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NS; ++s) st[jt] = mfma4(kf[jt][s], qf[s], st[jt]);
    }
    // softmax over keys j = 16 jt + 4 g + r for query i
    const uint8_t* mrow = P.mask ? P.mask + b * P.mask_bs + (int64_t)ic * P.mask_qs : nullptr;
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jt + 4 * g + r;
        float s = st[jt][r] * P.scale;
        if (j >= nk || (mrow && mrow[j < nk ? j : 0])) s = -INFINITY;
        st[jt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = xor_max_g(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(st[jt][r] - mx);
        st[jt][r] = e;
        sum += e;
      }
    sum = xor_sum_g(sum);
    const float inv = 1.f / sum;
    if (g == 0 && i < nq) P.lse[(int64_t)bh * nq + i] = mx + __logf(sum);
    // O[i][c] = sum_j P[i][j] V[j][c]: A = the score registers, B = V[j = 16 jt + 4 g + r][16 ct + lane&15]
    f32x4 o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NKT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = 16 * jt + 4 * g + r;
        j = j < nk ? j : nk - 1;
        const float* vr = Vb + (int64_t)j * P.ldv;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          const float bv = (DK >= 16 || c < DK) ? vr[c < DK ? c : 0] : 0.f;
          o[ct] = mfma4(st[jt][r], bv, o[ct]);
        }
      }
    // the result tile: col = channel 16 ct + lane&15, row = query 16 it + 4 g + r; 1/sum belongs to the ROW's query
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = 16 * it + 4 * g + r;
      const float invr = __shfl(inv, 4 * g + r, 64);   // lane (4 g + r) of group 0 holds query 16 it + 4 g + r
      if (io < nq) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          if (DK >= 16 || c < DK) Ob[(int64_t)io * P.ldo + c] = o[ct][r] * invr;
        }
      }
    }
  }
}


template <int DK>
int launch_fwd_nt(const AttnArgs& A, int nkt, hipStream_t s) {
  const dim3 grid((A.BH + 3) / 4), block(256);
  switch (nkt) {
    case 1: hipLaunchKernelGGL((nast_mha_fwd_wave_kernel<DK, 1>), grid, block, 0, s, A); break;
    case 2: hipLaunchKernelGGL((nast_mha_fwd_wave_kernel<DK, 2>), grid, block, 0, s, A); break;
    case 3: hipLaunchKernelGGL((nast_mha_fwd_wave_kernel<DK, 3>), grid, block, 0, s, A); break;
    default: hipLaunchKernelGGL((nast_mha_fwd_wave_kernel<DK, 4>), grid, block, 0, s, A); break;
  }
  return 0;
}
template <int DK>
int launch_bwd_nt(const AttnArgs& A, int nt, hipStream_t s) {
  const dim3 grid(A.BH), block(256);
  switch (nt) {
    case 1: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 1>), grid, block, 0, s, A); break;
    case 2: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 2>), grid, block, 0, s, A); break;
    case 3: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 3>), grid, block, 0, s, A); break;
    default: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 4>), grid, block, 0, s, A); break;
  }
  return 0;
}

}  // namespace

// Entry points used by attention.hip's dispatcher.  Return 1 when the shape is outside this file's limits.
int nast_mha_fwd_mfma(const float* q, const float* k, const float* v, const uint8_t* mask, float* out, float* lse, int B, int h, int nq,
                      int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                      int64_t mask_bs, int64_t mask_qs, float scale, hipStream_t s) {
  if (nq > 64 || nk > 64) return 1;
  AttnArgs A = {};
  A.q = q; A.k = k; A.v = v; A.mask = mask; A.out = out; A.lse = lse;
  A.BH = B * h; A.h = h; A.nq = nq; A.nk = nk; A.ldq = ldq; A.ldk = ldk; A.ldv = ldv; A.ldo = ldo;
  A.q_bs = q_bs; A.k_bs = k_bs; A.v_bs = v_bs; A.o_bs = o_bs; A.mask_bs = mask_bs; A.mask_qs = mask_qs; A.scale = scale;
  const int nkt = (nk + 15) / 16;
  switch (dk) {
    case 8: return launch_fwd_nt<8>(A, nkt, s);
    case 16: return launch_fwd_nt<16>(A, nkt, s);
    case 32: return launch_fwd_nt<32>(A, nkt, s);
    case 64: return launch_fwd_nt<64>(A, nkt, s);
  }
  return 1;
}

int nast_mha_bwd_mfma(const float* q, const float* k, const float* v, const uint8_t* mask, const float* dout, const float* lse, float* dq,
                      float* dk_, float* dv, int B, int h, int nq, int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs,
                      int64_t k_bs, int64_t v_bs, int64_t o_bs, int64_t mask_bs, int64_t mask_qs, float scale, hipStream_t s) {
  if (nq > 64 || nk > 64) return 1;
  AttnArgs A = {};
  A.q = q; A.k = k; A.v = v; A.mask = mask; A.dout = dout; A.lse_in = lse; A.dq = dq; A.dk = dk_; A.dv = dv;
  A.BH = B * h; A.h = h; A.nq = nq; A.nk = nk; A.ldq = ldq; A.ldk = ldk; A.ldv = ldv; A.ldo = ldo;
  A.q_bs = q_bs; A.k_bs = k_bs; A.v_bs = v_bs; A.o_bs = o_bs; A.mask_bs = mask_bs; A.mask_qs = mask_qs; A.scale = scale;
  const int nt = ((nq > nk ? nq : nk) + 15) / 16;
  switch (dk) {
    case 8: return launch_bwd_nt<8>(A, nt, s);
    case 16: return launch_bwd_nt<16>(A, nt, s);
    case 32: return launch_bwd_nt<32>(A, nt, s);
    case 64: return launch_bwd_nt<64>(A, nt, s);
  }
  return 1;
}
