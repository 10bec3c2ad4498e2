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
// Limits: nq, nk <= 64 and dk in {8, 16, 32, 64}; 65 .. 144 positions (dk 16 / 32 / 64) take the *_long kernels further down;
// anything else uses the LDS VALU kernels in attention.hip.
#include <type_traits>

#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef ATTN_STAMPS   // debug build only (tools/attn_stamps.py): per-workgroup s_memtime stamps of the backward kernel's segments
__device__ unsigned long long g_attn_stamps[8192 * 8];
#define ATTN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_attn_stamps[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int nast_attn_stamps_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * n);
}
#else
#define ATTN_STAMP(k) do { } while (0)
#endif

#ifndef ATTN_LD_PAD
#define ATTN_LD_PAD 4      // (lab builds: 8 makes the 16-byte row reads conflict-free and the 4-byte column reads 2-way)
#endif

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
  const float *q, *k, *v, *dout, *lse_in, *o_in;   // o_in: forward output for the backward pass, or NULL
  const uint8_t* mask;
  float *out, *lse, *dq, *dk, *dv;
  int BH, h, nq, nk, ldq, ldk, ldv, ldo;
  int64_t q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs;
  float scale;
  // packed ("varlen") rows: sentence b owns rows q_off[b] .. q_off[b]+q_len[b]-1 of q / out / dq and rows
  // k_off[b] .. +k_len[b]-1 of k / v / dk / dv; nq / nk are then the maxima (they size the tiles) and mask is NULL
  const int32_t *q_off, *k_off, *q_len, *k_len;
  int causal;                                   // packed rows only: key j is visible to query i iff j <= i
  // packed rows only: rows [q_total, q_alloc) of out / dq and [k_total, k_alloc) of dk / dv belong to no sentence (buffer
  // padding); the kernels write zeros there so that the caller needs no separate fill
  int q_total, q_alloc, k_total, k_alloc;
};

template <int DK>
__device__ __forceinline__ void zero_pad_rows(float* __restrict__ base, int ld, int head, int row0, int row1, int tid, int nthreads) {
  const int n = (row1 - row0) * DK;
  for (int c = tid; c < n; c += nthreads) base[(int64_t)(row0 + c / DK) * ld + head * DK + c % DK] = 0.f;
}

// per-sentence geometry: padded batches use the batch strides, packed ones the offset arrays
struct SeqGeom {
  int nq, nk;
  int64_t q0, k0, v0, o0;
};
__device__ __forceinline__ SeqGeom seq_geom(const AttnArgs& P, int b) {
  SeqGeom g;
  if (P.q_off != nullptr) {
    g.nq = P.q_len[b]; g.nk = P.k_len[b];
    g.q0 = (int64_t)P.q_off[b] * P.ldq; g.o0 = (int64_t)P.q_off[b] * P.ldo;
    g.k0 = (int64_t)P.k_off[b] * P.ldk; g.v0 = (int64_t)P.k_off[b] * P.ldv;
  } else {
    g.nq = P.nq; g.nk = P.nk;
    g.q0 = b * P.q_bs; g.o0 = b * P.o_bs; g.k0 = b * P.k_bs; g.v0 = b * P.v_bs;
  }
  return g;
}

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
// Backward.  Two exact short cuts cut the padded work of a batch of ragged sentences to about a third:
//   * key tiles that are masked for every query (a key-padding mask, mask_qs == 0, hides the tail of the
//     sentence) have P == 0: they are not staged, not multiplied, and their dK / dV rows are written as zeros;
//   * query tiles whose dO rows are all zero (padded target positions carry no loss) have dP == delta == dS == 0:
//     they add nothing to dK / dV and their dQ rows are zeros.  The test is made on the data while dO is staged.
// ---------------------------------------------------------------------------------------------------------
template <int DK>
__device__ __forceinline__ void stage_rows_flag(float* __restrict__ dst, const float* __restrict__ src, int64_t ld, int rows, int rows_pad,
                                                int* __restrict__ tile_flag) {
  constexpr int LD = DK + 4, V4 = DK / 4;
  for (int c = threadIdx.x; c < rows_pad * V4; c += 256) {
    const int r = c / V4, q = (c % V4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) v = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + q);
    *reinterpret_cast<float4*>(dst + r * LD + q) = v;
    if (v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f) tile_flag[r >> 4] = 1;   // benign race: every writer stores 1
  }
}

// SPLIT (the caller passed the forward output O): delta[i] = sum_j P[i][j] dP[i][j] = sum_c dO[i][c] O[i][c] is then taken
// from the rows while they are staged, so phase 2 no longer waits for phase 1: there is no barrier between them and the
// 16-row tiles of both are dealt to the four wavefronts together (sentences of up to 32 positions: two wavefronts on dQ, two
// on dK / dV at the same time, where the two-phase form leaves two of the four idle in each phase).
// (Tried on top of SPLIT and dropped, round 2: a streaming form -- 2 workgroups per CU walking the items with the NEXT item's
// rows requested into registers before the current item's arithmetic -- needs > 256 VGPRs (11-52 spilled) and with 8 wavefronts
// per CU runs 35.6 / 62.8 us where this form takes 30.9 / 41.8 at 32 / 34 positions, 512 sentences x 4 heads.)
template <int DK, int NT, bool SPLIT>
__global__ __launch_bounds__(256, NT <= 3 ? 3 : 2) void nast_mha_bwd_mfma_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16, LD = DK + ATTN_LD_PAD;
  __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Qs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Gs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float dlt[NT * 16];     // delta[i] = sum_j P[i][j] dP[i][j]
  __shared__ __attribute__((aligned(16))) float lse_s[NT * 16];
  __shared__ unsigned long long dead_q[NT * 16];   // per query: bit j set = key j gets no attention weight
  __shared__ unsigned long long dead_k[NT * 16];   // per key: bit i set = query i gives it no weight
  __shared__ int live_q[NT];                    // query tile has a non-zero dO row
  static_assert(NT * 16 <= 256, "one log-sum-exp value per thread");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
  const float scale = P.scale;
  const bool pad_mask = P.mask != nullptr && P.mask_qs == 0;
  constexpr int V4 = DK / 4, CH = (NT * 16 * V4 + 255) / 256;
  // everything the kernel reads, held in registers from `fetch` to `commit`: K, V, Q, dO (and O) rows, the log-sum-exp row,
  // mask bytes for the dead flags of both orientations -- ONE memory round trip
  float4 rk[CH], rv[CH], rq[CH], rg[CH], ro[SPLIT ? CH : 1];
  uint8_t mq_b[NT * 4], mk_b[NT * 4];
  float lse_r = 0.f;
  auto fetch = [&](int bh, const SeqGeom& G) {
    const int b = bh / P.h, head = bh % P.h, nq = G.nq, nk = G.nk;
    if (nq <= 0 || nk <= 0) return;             // (uniform for the workgroup)
    // mask bytes: NT*4 rows of each orientation per wavefront (lane = key for dead_q, lane = query for dead_k)
#pragma unroll
    for (int u = 0; u < NT * 4; ++u) {
      const int row = 4 * u + wave;
      const int ic = row < nq ? row : nq - 1, jl = lane < nk ? lane : nk - 1;
      const int jc = row < nk ? row : nk - 1, il = lane < nq ? lane : nq - 1;
      mq_b[u] = P.mask != nullptr ? P.mask[b * P.mask_bs + (int64_t)ic * P.mask_qs + jl] : (uint8_t)0;
      mk_b[u] = P.mask != nullptr ? P.mask[b * P.mask_bs + (int64_t)il * P.mask_qs + jc] : (uint8_t)0;
    }
    const int nqt = (nq + 15) / 16;
    lse_r = (int)threadIdx.x < nqt * 16 ? P.lse_in[(int64_t)bh * P.nq + ((int)threadIdx.x < nq ? (int)threadIdx.x : nq - 1)] : 0.f;
    const float* Kg = P.k + G.k0 + head * DK;
    const float* Vg = P.v + G.v0 + head * DK;
    const float* Qg = P.q + G.q0 + head * DK;
    const float* Gg = P.dout + G.o0 + head * DK;
    const float* Og = SPLIT ? P.o_in + G.o0 + head * DK : nullptr;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = threadIdx.x + 256 * u, r = c / V4, q = (c % V4) * 4;
      rk[u] = rv[u] = rq[u] = rg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (SPLIT) ro[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nk) {
        rk[u] = *reinterpret_cast<const float4*>(Kg + (int64_t)r * P.ldk + q);
        rv[u] = *reinterpret_cast<const float4*>(Vg + (int64_t)r * P.ldv + q);
      }
      if (r < nq) {
        rq[u] = *reinterpret_cast<const float4*>(Qg + (int64_t)r * P.ldq + q);
        rg[u] = *reinterpret_cast<const float4*>(Gg + (int64_t)r * P.ldo + q);
        if constexpr (SPLIT) ro[u] = *reinterpret_cast<const float4*>(Og + (int64_t)r * P.ldo + q);
      }
    }
  };
  // registers -> LDS images (rows past the sequence end zero-filled), delta, log-sum-exp, dead-flag bit rows by wave ballots
  auto commit = [&](const SeqGeom& G) {
    const int nq = G.nq, nk = G.nk, nqt = (nq + 15) / 16;
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = threadIdx.x + 256 * u, r = c / V4, q = (c % V4) * 4;
      if (r < NT * 16) {
        *reinterpret_cast<float4*>(Ks + r * LD + q) = rk[u];
        *reinterpret_cast<float4*>(Vs + r * LD + q) = rv[u];
        *reinterpret_cast<float4*>(Qs + r * LD + q) = rq[u];
        *reinterpret_cast<float4*>(Gs + r * LD + q) = rg[u];
        if (rg[u].x != 0.f || rg[u].y != 0.f || rg[u].z != 0.f || rg[u].w != 0.f) live_q[r >> 4] = 1;   // benign race: all store 1
      }
      if constexpr (SPLIT) {                     // delta of row r: the V4 consecutive lanes that hold the row add their pieces
        float part = rg[u].x * ro[u].x + rg[u].y * ro[u].y + rg[u].z * ro[u].z + rg[u].w * ro[u].w;
#pragma unroll
        for (int o = V4 / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (c % V4 == 0 && r < NT * 16) dlt[r] = part;
      }
    }
    if ((int)threadIdx.x < nqt * 16) lse_s[threadIdx.x] = lse_r;
#pragma unroll
    for (int u = 0; u < NT * 4; ++u) {
      const int row = 4 * u + wave;
      const unsigned long long mq = __ballot(row >= nq || lane >= nk || mq_b[u] != 0 || (P.causal && lane > row));
      const unsigned long long mk = __ballot(row >= nk || lane >= nq || mk_b[u] != 0 || (P.causal && row > lane));
      if (lane == 0) {
        dead_q[row] = mq;
        dead_k[row] = mk;
      }
    }
  };

  ATTN_STAMP(0);
  const int bh = blockIdx.x;
  const int b = bh / P.h, head = bh % P.h;
  const SeqGeom G = seq_geom(P, b);
  const int nq = G.nq, nk = G.nk;
  if (b == 0 && P.q_off != nullptr) {
    zero_pad_rows<DK>(P.dq, P.ldq, head, P.q_total, P.q_alloc, threadIdx.x, 256);
    zero_pad_rows<DK>(P.dk, P.ldk, head, P.k_total, P.k_alloc, threadIdx.x, 256);
    zero_pad_rows<DK>(P.dv, P.ldv, head, P.k_total, P.k_alloc, threadIdx.x, 256);
  }
  if (nq <= 0 || nk <= 0) return;               // (uniform for the workgroup)
  float* dQb = P.dq + G.q0 + head * DK;
  float* dKb = P.dk + G.k0 + head * DK;
  float* dVb = P.dv + G.v0 + head * DK;
  const int nqt = (nq + 15) / 16, nkt = (nk + 15) / 16;
  if (threadIdx.x < NT) live_q[threadIdx.x] = 0;
  __syncthreads();                              // (before the loads are requested: a barrier waits for everything in flight)
  fetch(bh, G);
  ATTN_STAMP(1);
  commit(G);
  ATTN_STAMP(2);
  __syncthreads();
  ATTN_STAMP(3);
  // keys past the last one that any query may see never take part (a key-padding mask hides the sentence tail)
  int nk_e = nk;
  if (pad_mask) {
    const unsigned long long seen = ~dead_q[0];
    if (seen != 0ull) nk_e = 64 - __clzll((long long)seen);   // all keys masked: keep the plain path (NaN like the reference)
  }
  const int nkt_e = (nk_e + 15) / 16;

  // ---- phase 1: lane = query.  dQ[i][c] = scale * sum_j dS[i][j] K[j][c] -----------------------------------
  auto phase1_tile = [&](int it) {
    if (!live_q[it]) {                           // dO == 0 for the whole tile: dQ rows are zero
      for (int c = lane; c < 16 * (DK / 4); c += 64) {
        const int io = 16 * it + c / (DK / 4), q4 = (c % (DK / 4)) * 4;
        if (io < nq) *reinterpret_cast<float4*>(dQb + (int64_t)io * P.ldq + q4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    // the tile's work for NKE live key tiles; NKE is a compile-time constant so that every loop unrolls and the
    // score registers stay registers
    auto q_tile = [&](auto nke_c) {
      constexpr int NKE = decltype(nke_c)::value;
      float qf[NS], gf[NS];
      lds_frag<DK>(qf, Qs + (16 * it + r15) * LD, g);
      lds_frag<DK>(gf, Gs + (16 * it + r15) * LD, g);
      const float l = lse_s[16 * it + r15];
      f32x4 st[NKE], dp[NKE];
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt) {
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
      for (int jt = 0; jt < NKE; ++jt) {
        const uint32_t dw = (uint32_t)(dead_q[16 * it + r15] >> (16 * jt + 4 * g)) & 15u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = (dw >> r) & 1u;
          const float p = dead ? 0.f : __expf(st[jt][r] * scale - l);
          st[jt][r] = p;
          dsum = fmaf(p, dp[jt][r], dsum);
        }
      }
      if (SPLIT) {
        dsum = dlt[16 * it + r15];
      } else {
        dsum = xor_sum_g(dsum);
        if (g == 0) dlt[16 * it + r15] = dsum;
      }
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st[jt][r] = st[jt][r] * (dp[jt][r] - dsum) * scale;   // dS * scale
      f32x4 acc[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt)
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
    };
    if (NT >= 4 && nkt_e >= 4) q_tile(std::integral_constant<int, NT >= 4 ? 4 : 1>{});
    else if (NT >= 3 && nkt_e == 3) q_tile(std::integral_constant<int, NT >= 3 ? 3 : 1>{});
    else if (NT >= 2 && nkt_e == 2) q_tile(std::integral_constant<int, NT >= 2 ? 2 : 1>{});
    else q_tile(std::integral_constant<int, 1>{});
  };

  // ---- phase 2: lane = key.  dK[j][c] = scale * sum_i dS[i][j] Q[i][c],  dV[j][c] = sum_i P[i][j] dO[i][c] ----
  auto phase2_tile = [&](int jt) {
    if (jt >= nkt_e) {                            // every key of the tile is masked: dK = dV = 0
      for (int c = lane; c < 16 * (DK / 4); c += 64) {
        const int jo = 16 * jt + c / (DK / 4), q4 = (c % (DK / 4)) * 4;
        if (jo < nk) {
          *reinterpret_cast<float4*>(dKb + (int64_t)jo * P.ldk + q4) = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(dVb + (int64_t)jo * P.ldv + q4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      return;
    }
    float kf[NS], vf[NS];                        // this lane's key: j = 16 jt + r15
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
      if (it < nqt && live_q[it]) {
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
        const uint32_t dw = (uint32_t)(dead_k[16 * jt + r15] >> (16 * it + 4 * g)) & 15u;
        const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[16 * it + 4 * g]);
        const float4 t4 = *reinterpret_cast<const float4*>(&dlt[16 * it + 4 * g]);
        const f32x4 lr = {l4.x, l4.y, l4.z, l4.w}, tr = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = (dw >> r) & 1u;
          const float p = dead ? 0.f : __expf(s4[r] * scale - lr[r]);
          p4[r] = p;
          ds4[r] = p * (d4[r] - tr[r]) * scale;
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
  };
  if (SPLIT) {
    // task t: bit 1 = phase, tile = 2 (t >> 2) + (t & 1); wavefront w takes t = w, w + 4: up to two tiles per side run at once
    for (int t = wave; t < 4 * ((NT + 1) / 2); t += 4) {
      const int tile = 2 * (t >> 2) + (t & 1);
      if ((t >> 1) & 1) {
        if (tile < nkt) phase2_tile(tile);
      } else if (tile < nqt) {
        phase1_tile(tile);
      }
    }
  } else {
    for (int it = wave; it < nqt; it += 4) phase1_tile(it);
    __syncthreads();   // dlt is read by other wavefronts below
    ATTN_STAMP(4);
    for (int jt = wave; jt < nkt; jt += 4) phase2_tile(jt);
  }
  ATTN_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------
// Forward, one WAVEFRONT per (sentence, head), operands straight from global memory (no LDS, no barrier): measured
// faster than the workgroup-cooperative LDS variant above for the forward pass, because a forward tile has too
// little arithmetic to amortise the staging round trip.  The kernel is latency-bound, so it is organised around
// memory round trips: K row fragments AND the V column operands of every live key tile are fetched once, before
// the query loop (they do not depend on the query tile), and the next query tile's Q fragment (and mask bits) is
// requested before the current tile's arithmetic.  Key tiles that a key-padding mask (mask_qs == 0) hides for every
// query contribute exp(-inf) = 0 exactly and are skipped: the body is specialised on the number of live key tiles.
// ---------------------------------------------------------------------------------------------------------
template <int DK, int NKT>
__global__ __launch_bounds__(256) void nast_mha_fwd_wave_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16;
  const int lane = threadIdx.x & 63, r15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bh >= P.BH) return;
  const int b = bh / P.h, head = bh % P.h;
  const SeqGeom G = seq_geom(P, b);
  const int nq = G.nq, nk = G.nk;
  if (b == 0 && P.q_off != nullptr) zero_pad_rows<DK>(P.out, P.ldo, head, P.q_total, P.q_alloc, lane, 64);
  if (nq <= 0 || nk <= 0) return;               // (uniform for the wavefront)
  const float* Kb = P.k + G.k0 + head * DK;
  const float* Vb = P.v + G.v0 + head * DK;
  const float* Qb = P.q + G.q0 + head * DK;
  float* Ob = P.out + G.o0 + head * DK;
  const int nqt = (nq + 15) / 16;
  const bool pad_mask = P.mask != nullptr && P.mask_qs == 0;
  const bool row_mask = P.mask != nullptr && P.mask_qs != 0;
  // bit j set = key j takes no weight (for every query when the mask is a key-padding mask)
  unsigned long long dead_row = nk < 64 ? (~0ull << nk) : 0ull;
  int nk_e = nk;
  if (pad_mask) {
    dead_row |= __ballot(lane < nk && P.mask[b * P.mask_bs + (lane < nk ? lane : 0)] != 0);
    const unsigned long long seen = ~dead_row;
    if (seen != 0ull) nk_e = 64 - __clzll((long long)seen);   // all keys masked: plain path (NaN like the reference)
  }
  const int nkt_e = (nk_e + 15) / 16;           // (packed rows: nk is already the sentence's own length)

  auto run = [&](auto nke_c) {
    constexpr int NKE = decltype(nke_c)::value;
    // 4 dead bits (r = 0..3) per key tile for this lane's query of tile `it`
    auto dead_bits = [&](int it) -> uint32_t {
      uint32_t w = 0;
      if (row_mask) {
        const int i = 16 * it + r15;
        const uint8_t* mrow = P.mask + b * P.mask_bs + (int64_t)(i < nq ? i : nq - 1) * P.mask_qs;
#pragma unroll
        for (int jt = 0; jt < NKE; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * g + r;
            w |= (uint32_t)(j >= nk || mrow[j < nk ? j : 0] != 0) << (4 * jt + r);
          }
      } else {
        unsigned long long dr = dead_row;
        if (P.causal) {                            // keys after this lane's query
          const int i = 16 * it + r15;
          dr |= i < 63 ? (~0ull << (i + 1)) : 0ull;
        }
#pragma unroll
        for (int jt = 0; jt < NKE; ++jt) w |= ((uint32_t)(dr >> (16 * jt + 4 * g)) & 15u) << (4 * jt);
      }
      return w;
    };
    float kf[NKE][NS];
    float vc[NKE][4][NCT];
#pragma unroll
    for (int jt = 0; jt < NKE; ++jt) {
      int j = 16 * jt + r15;
      j = j < nk ? j : nk - 1;
      load_frag<DK>(kf[jt], Kb + (int64_t)j * P.ldk, g);
    }
    float qf[NS];
    load_frag<DK>(qf, Qb + (int64_t)(r15 < nq ? r15 : nq - 1) * P.ldq, g);
    uint32_t dbits = dead_bits(0);
#pragma unroll
    for (int jt = 0; jt < NKE; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = 16 * jt + 4 * g + r;
        j = j < nk ? j : nk - 1;
        const float* vr = Vb + (int64_t)j * P.ldv;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + r15;
          vc[jt][r][ct] = (DK >= 16 || c < DK) ? vr[c < DK ? c : 0] : 0.f;
        }
      }
    for (int it = 0; it < nqt; ++it) {
      const int i = 16 * it + r15;                 // this lane's query
      float qn[NS];
      uint32_t dnext = 0;
      const bool more = it + 1 < nqt;
      if (more) {                                  // next tile's operands travel while this tile computes
        const int in = i + 16;
        load_frag<DK>(qn, Qb + (int64_t)(in < nq ? in : nq - 1) * P.ldq, g);
        dnext = dead_bits(it + 1);
      }
      f32x4 st[NKE];
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt) {
        st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) st[jt] = mfma4(kf[jt][s], qf[s], st[jt]);
      }
      // softmax over keys j = 16 jt + 4 g + r for query i
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sc = st[jt][r] * P.scale;
          if ((dbits >> (4 * jt + r)) & 1u) sc = -INFINITY;
          st[jt][r] = sc;
          mx = fmaxf(mx, sc);
        }
      mx = xor_max_g(mx);
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(st[jt][r] - mx);
          st[jt][r] = e;
          sum += e;
        }
      sum = xor_sum_g(sum);
      const float inv = 1.f / sum;
      if (g == 0 && i < nq) P.lse[(int64_t)bh * P.nq + i] = mx + __logf(sum);
      // O[i][c] = sum_j P[i][j] V[j][c]: A = the score registers, B = V[j = 16 jt + 4 g + r][16 ct + lane&15]
      f32x4 o[NCT];
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < NKE; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) o[ct] = mfma4(st[jt][r], vc[jt][r][ct], o[ct]);
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
      if (more) {
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = qn[s];
        dbits = dnext;
      }
    }
  };
  if (NKT >= 4 && nkt_e >= 4) run(std::integral_constant<int, NKT >= 4 ? 4 : 1>{});
  else if (NKT >= 3 && nkt_e == 3) run(std::integral_constant<int, NKT >= 3 ? 3 : 1>{});
  else if (NKT >= 2 && nkt_e == 2) run(std::integral_constant<int, NKT >= 2 ? 2 : 1>{});
  else run(std::integral_constant<int, 1>{});
}


// ---------------------------------------------------------------------------------------------------------
// Sequences of 65 .. 144 positions (BASELINE configs[4]: max_length 128 -> 129 / 130 rows per sentence).  Same products and
// the same transposed-score orientation as the kernels above, with what does not scale to 9 key tiles replaced:
//   * masks are evaluated per element (a dead byte per key in LDS for key-padding masks and sequence ends, the caller's
//     byte for [n, n] masks, an index compare for the causal flag) instead of 64-bit rows built with ballots;
//   * backward: K and V stay in LDS for the whole workgroup (78 KB at d_k = 64), Q and dO pass through in blocks of 64
//     queries (35 KB); dK / dV accumulators of a wavefront's (up to 3) key tiles live in registers across the blocks.
//     Per block: phase 1 (lane = query) writes the block's dQ rows and delta, phase 2 (lane = key) adds the block to dK / dV;
//   * every loop over key tiles is cut at the sentence's own length (uniform per workgroup), so packed rows pay for their
//     real lengths.
// One workgroup per (sentence, head); forward 2 workgroups per CU, backward 1 (114 KB of LDS).
// ---------------------------------------------------------------------------------------------------------
constexpr int LONG_NT = 9;

template <int DK>
__device__ __forceinline__ void stage_block(float* __restrict__ dst, const float* __restrict__ src, int64_t ld, int row0, int rows,
                                            int rows_pad) {
  constexpr int LD = DK + 4, V4 = DK / 4;
  for (int c = threadIdx.x; c < rows_pad * V4; c += 256) {
    const int r = c / V4, q = (c % V4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < rows) v = *reinterpret_cast<const float4*>(src + (int64_t)(row0 + r) * ld + q);
    *reinterpret_cast<float4*>(dst + r * LD + q) = v;
  }
}

// dead byte per key: past the sentence, or hidden by a key-padding mask (mask_qs == 0)
__device__ __forceinline__ void stage_kdead(uint8_t* __restrict__ kdead, const AttnArgs& P, int b, int nk, int keys_pad) {
  const bool pad_mask = P.mask != nullptr && P.mask_qs == 0;
  for (int j = threadIdx.x; j < keys_pad; j += 256) kdead[j] = (j >= nk || (pad_mask && P.mask[b * P.mask_bs + j] != 0)) ? 1 : 0;
}

template <int DK, int NT>
__global__ __launch_bounds__(256) void nast_mha_fwd_long_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16, LD = DK + 4;
  __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) uint8_t kdead[NT * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x;
  const int b = bh / P.h, head = bh % P.h;
  const SeqGeom G = seq_geom(P, b);
  const int nq = G.nq, nk = G.nk;
  if (b == 0 && P.q_off != nullptr) zero_pad_rows<DK>(P.out, P.ldo, head, P.q_total, P.q_alloc, threadIdx.x, 256);
  if (nq <= 0 || nk <= 0) return;               // (uniform for the workgroup)
  const float* Qb = P.q + G.q0 + head * DK;
  float* Ob = P.out + G.o0 + head * DK;
  const int nqt = (nq + 15) / 16, nkt = (nk + 15) / 16;
  const bool row_mask = P.mask != nullptr && P.mask_qs != 0;
  stage_block<DK>(Ks, P.k + G.k0 + head * DK, P.ldk, 0, nk, nkt * 16);
  stage_block<DK>(Vs, P.v + G.v0 + head * DK, P.ldv, 0, nk, nkt * 16);
  stage_kdead(kdead, P, b, nk, NT * 16);
  __syncthreads();
  for (int it = wave; it < nqt; it += 4) {
    const int i = 16 * it + r15;                 // this lane's query
    const int ic = i < nq ? i : nq - 1;
    float qf[NS];
    load_frag<DK>(qf, Qb + (int64_t)ic * P.ldq, g);
    const uint8_t* mrow = row_mask ? P.mask + b * P.mask_bs + (int64_t)ic * P.mask_qs : nullptr;
    f32x4 st[NT];
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      st[jt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (jt < nkt) {
        float kf[NS];
        lds_frag<DK>(kf, Ks + (16 * jt + r15) * LD, g);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) a = mfma4(kf[s], qf[s], a);
        const uint32_t kd = *reinterpret_cast<const uint32_t*>(&kdead[16 * jt + 4 * g]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * jt + 4 * g + r;
          const bool dead = ((kd >> (8 * r)) & 0xffu) != 0 || (mrow != nullptr && mrow[j < nk ? j : 0] != 0) || (P.causal && j > i);
          const float sc = dead ? -INFINITY : a[r] * P.scale;
          st[jt][r] = sc;
          mx = fmaxf(mx, sc);
        }
      }
    }
    mx = xor_max_g(mx);
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      if (jt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(st[jt][r] - mx);
          st[jt][r] = e;
          sum += e;
        }
      }
    sum = xor_sum_g(sum);
    const float inv = 1.f / sum;
    if (g == 0 && i < nq) P.lse[(int64_t)bh * P.nq + i] = mx + __logf(sum);
    f32x4 o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) o[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      if (jt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* vr = Vs + (16 * jt + 4 * g + r) * LD;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) o[ct] = mfma4(st[jt][r], vr[16 * ct + r15], o[ct]);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = 16 * it + 4 * g + r;
      const float invr = __shfl(inv, 4 * g + r, 64);
      if (io < nq) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Ob[(int64_t)io * P.ldo + 16 * ct + r15] = o[ct][r] * invr;
      }
    }
  }
}

template <int DK, int NT>
__global__ __launch_bounds__(256, 1) void nast_mha_bwd_long_kernel(const AttnArgs P) {
  constexpr int NS = DK / 4, NCT = (DK + 15) / 16, LD = DK + 4, QB = 4, KU = (NT + 3) / 4;
  __shared__ __attribute__((aligned(16))) float Ks[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Vs[NT * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Qs[QB * 16 * LD];
  __shared__ __attribute__((aligned(16))) float Gs[QB * 16 * LD];
  __shared__ __attribute__((aligned(16))) float dlt[QB * 16];
  __shared__ __attribute__((aligned(16))) float lse_s[QB * 16];
  __shared__ __attribute__((aligned(16))) uint8_t kdead[NT * 16];
  __shared__ int live_q[QB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r15 = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x;
  const int b = bh / P.h, head = bh % P.h;
  const SeqGeom G = seq_geom(P, b);
  const int nq = G.nq, nk = G.nk;
  if (b == 0 && P.q_off != nullptr) {
    zero_pad_rows<DK>(P.dq, P.ldq, head, P.q_total, P.q_alloc, threadIdx.x, 256);
    zero_pad_rows<DK>(P.dk, P.ldk, head, P.k_total, P.k_alloc, threadIdx.x, 256);
    zero_pad_rows<DK>(P.dv, P.ldv, head, P.k_total, P.k_alloc, threadIdx.x, 256);
  }
  if (nq <= 0 || nk <= 0) return;               // (uniform for the workgroup)
  float* dQb = P.dq + G.q0 + head * DK;
  float* dKb = P.dk + G.k0 + head * DK;
  float* dVb = P.dv + G.v0 + head * DK;
  const float* Qg = P.q + G.q0 + head * DK;
  const float* Gg = P.dout + G.o0 + head * DK;
  const float scale = P.scale;
  const int nqt = (nq + 15) / 16, nkt = (nk + 15) / 16;
  const bool row_mask = P.mask != nullptr && P.mask_qs != 0;
  stage_block<DK>(Ks, P.k + G.k0 + head * DK, P.ldk, 0, nk, nkt * 16);
  stage_block<DK>(Vs, P.v + G.v0 + head * DK, P.ldv, 0, nk, nkt * 16);
  stage_kdead(kdead, P, b, nk, NT * 16);
  __syncthreads();
  // key tiles past the last key any query may see take no part (a key-padding mask hides the sentence tail)
  int nkt_e = nkt;
  while (nkt_e > 1) {
    bool any = false;
    for (int r = 0; r < 16; ++r) any = any || kdead[16 * (nkt_e - 1) + r] == 0;
    if (any) break;
    --nkt_e;
  }
  f32x4 accK[KU][NCT], accV[KU][NCT];
#pragma unroll
  for (int u = 0; u < KU; ++u)
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      accK[u][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      accV[u][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  for (int qb = 0; qb * QB < nqt; ++qb) {
    const int row0 = qb * QB * 16;
    if (threadIdx.x < QB) live_q[threadIdx.x] = 0;
    __syncthreads();
    {
      constexpr int V4 = DK / 4;
      for (int c = threadIdx.x; c < QB * 16 * V4; c += 256) {
        const int r = c / V4, q = (c % V4) * 4;
        float4 vq = make_float4(0.f, 0.f, 0.f, 0.f), vg = vq;
        if (row0 + r < nq) {
          vq = *reinterpret_cast<const float4*>(Qg + (int64_t)(row0 + r) * P.ldq + q);
          vg = *reinterpret_cast<const float4*>(Gg + (int64_t)(row0 + r) * P.ldo + q);
        }
        *reinterpret_cast<float4*>(Qs + r * LD + q) = vq;
        *reinterpret_cast<float4*>(Gs + r * LD + q) = vg;
        if (vg.x != 0.f || vg.y != 0.f || vg.z != 0.f || vg.w != 0.f) live_q[r >> 4] = 1;   // benign race: all store 1
      }
      for (int r = threadIdx.x; r < QB * 16; r += 256) {
        const int i = row0 + r;
        lse_s[r] = P.lse_in[(int64_t)bh * P.nq + (i < nq ? i : nq - 1)];
        dlt[r] = 0.f;
      }
    }
    __syncthreads();
    // ---- phase 1: lane = query.  dQ[i][c] = scale * sum_j dS[i][j] K[j][c]; one query tile of the block per wavefront ----
    {
      const int itl = wave, it = qb * QB + itl;
      if (it < nqt) {
        if (!live_q[itl]) {                        // dO == 0 for the whole tile: dQ rows are zero
          for (int c = lane; c < 16 * (DK / 4); c += 64) {
            const int io = 16 * it + c / (DK / 4), q4 = (c % (DK / 4)) * 4;
            if (io < nq) *reinterpret_cast<float4*>(dQb + (int64_t)io * P.ldq + q4) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        } else {
          const int i = 16 * it + r15, ic = i < nq ? i : nq - 1;
          const uint8_t* mrow = row_mask ? P.mask + b * P.mask_bs + (int64_t)ic * P.mask_qs : nullptr;
          float qf[NS], gf[NS];
          lds_frag<DK>(qf, Qs + (16 * itl + r15) * LD, g);
          lds_frag<DK>(gf, Gs + (16 * itl + r15) * LD, g);
          const float l = lse_s[16 * itl + r15];
          f32x4 st[NT], dp[NT];
          float dsum = 0.f;
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dp[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (jt < nkt_e) {
              float kf[NS], vf[NS];
              lds_frag<DK>(kf, Ks + (16 * jt + r15) * LD, g);
              lds_frag<DK>(vf, Vs + (16 * jt + r15) * LD, g);
#pragma unroll
              for (int s = 0; s < NS; ++s) {
                st[jt] = mfma4(kf[s], qf[s], st[jt]);
                dp[jt] = mfma4(vf[s], gf[s], dp[jt]);
              }
              const uint32_t kd = *reinterpret_cast<const uint32_t*>(&kdead[16 * jt + 4 * g]);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int j = 16 * jt + 4 * g + r;
                const bool dead = i >= nq || ((kd >> (8 * r)) & 0xffu) != 0 || (mrow != nullptr && mrow[j < nk ? j : 0] != 0) ||
                                  (P.causal && j > i);
                const float p = dead ? 0.f : __expf(st[jt][r] * scale - l);
                st[jt][r] = p;
                dsum = fmaf(p, dp[jt][r], dsum);
              }
            }
          }
          dsum = xor_sum_g(dsum);
          if (g == 0) dlt[16 * itl + r15] = dsum;
          f32x4 acc[NCT];
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
            if (jt < nkt_e) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float ds = st[jt][r] * (dp[jt][r] - dsum) * scale;
                const float* kr = Ks + (16 * jt + 4 * g + r) * LD;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) acc[ct] = mfma4(ds, kr[16 * ct + r15], acc[ct]);
              }
            }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int io = 16 * it + 4 * g + r;
            if (io < nq) {
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) dQb[(int64_t)io * P.ldq + 16 * ct + r15] = acc[ct][r];
            }
          }
        }
      }
    }
    __syncthreads();   // dlt of the block is read by every wavefront below
    // ---- phase 2: lane = key.  dK[j][c] += scale * sum_i dS[i][j] Q[i][c],  dV[j][c] += sum_i P[i][j] dO[i][c] ----
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int jt = wave + 4 * u;
      if (jt < nkt_e) {
        const int j = 16 * jt + r15;
        float kf[NS], vf[NS];
        lds_frag<DK>(kf, Ks + (16 * jt + r15) * LD, g);
        lds_frag<DK>(vf, Vs + (16 * jt + r15) * LD, g);
        const bool jdead = kdead[j] != 0;
#pragma unroll
        for (int itl = 0; itl < QB; ++itl) {
          const int it = qb * QB + itl;
          if (it < nqt && live_q[itl]) {
            float qf[NS], gf[NS];
            lds_frag<DK>(qf, Qs + (16 * itl + r15) * LD, g);
            lds_frag<DK>(gf, Gs + (16 * itl + r15) * LD, g);
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, d4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              s4 = mfma4(qf[s], kf[s], s4);
              d4 = mfma4(gf[s], vf[s], d4);
            }
            const float4 l4 = *reinterpret_cast<const float4*>(&lse_s[16 * itl + 4 * g]);
            const float4 t4 = *reinterpret_cast<const float4*>(&dlt[16 * itl + 4 * g]);
            const f32x4 lr = {l4.x, l4.y, l4.z, l4.w}, tr = {t4.x, t4.y, t4.z, t4.w};
            f32x4 p4, ds4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * it + 4 * g + r;
              const bool dead = jdead || i >= nq || (P.causal && j > i) ||
                                (row_mask && P.mask[b * P.mask_bs + (int64_t)(i < nq ? i : nq - 1) * P.mask_qs + (j < nk ? j : 0)] != 0);
              const float p = dead ? 0.f : __expf(s4[r] * scale - lr[r]);
              p4[r] = p;
              ds4[r] = p * (d4[r] - tr[r]) * scale;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float* qr = Qs + (16 * itl + 4 * g + r) * LD;
              const float* gr = Gs + (16 * itl + 4 * g + r) * LD;
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct) {
                accK[u][ct] = mfma4(ds4[r], qr[16 * ct + r15], accK[u][ct]);
                accV[u][ct] = mfma4(p4[r], gr[16 * ct + r15], accV[u][ct]);
              }
            }
          }
        }
      }
    }
    __syncthreads();   // Qs / Gs / dlt are overwritten by the next block
  }
#pragma unroll
  for (int u = 0; u < KU; ++u) {
    const int jt = wave + 4 * u;
    if (jt < nkt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jo = 16 * jt + 4 * g + r;
        if (jo < nk) {
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const bool livet = jt < nkt_e;
            dKb[(int64_t)jo * P.ldk + 16 * ct + r15] = livet ? accK[u][ct][r] : 0.f;
            dVb[(int64_t)jo * P.ldv + 16 * ct + r15] = livet ? accV[u][ct][r] : 0.f;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Single query per sentence against a long key range (nq == 1, nk > 64): the greedy decode of a long-sentence model
// (configs[4]: 128 positions, 16 heads of 64) asks for this 6 times per step, and the tile kernels above spend 41 us staging
// 144-row images for one query row.  Here the work is a stream of K and V rows: one workgroup per (sentence, head), a row is
// read by DK/4 consecutive lanes as one 16-byte piece each (4 rows per wavefront instruction at DK = 64), key j goes to lane group
// j mod G (G = 1024 / DK groups in the workgroup), and every group keeps its own running (max, sum, weighted V row) -- the groups
// never talk inside the key loop.  All CK rows a group will need of K and of V are requested before any arithmetic (one memory
// round trip for up to G*CK keys); the partial softmaxes meet at the end: across the groups of a wavefront by shuffles, across
// the four wavefronts through LDS.  No MFMA: 2*nk*DK multiply-adds per head is nothing next to 2*nk*DK*4 bytes.
// ---------------------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float lanes_sum(float v) {   // over the LPR (4 / 8 / 16) consecutive lanes that share a row: DPP, no LDS
  v += nast_dpp<0xB1>(v);                     // quad_perm [1,0,3,2]
  v += nast_dpp<0x4E>(v);                     // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) v += nast_dpp<0x141>(v);   // row_half_mirror
  if constexpr (LPR >= 16) v += nast_dpp<0x140>(v);  // row_mirror
  return v;
}

template <int DK, int CK>
__global__ __launch_bounds__(256) void nast_mha_fwd_q1_kernel(const AttnArgs P) {
  constexpr int LPR = DK / 4, GW = 64 / LPR, G = 4 * GW;      // lanes per row, groups per wavefront / per workgroup
  __shared__ float m_s[4], l_s[4];
  __shared__ __attribute__((aligned(16))) float acc_s[4][DK];
  const int bh = blockIdx.x, b = bh / P.h, head = bh % P.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane % LPR, g = wave * GW + lane / LPR;
  const int nk = P.nk;
  const float* __restrict__ Kg = P.k + b * P.k_bs + head * DK + 4 * r;
  const float* __restrict__ Vg = P.v + b * P.v_bs + head * DK + 4 * r;
  const uint8_t* __restrict__ mrow = P.mask != nullptr ? P.mask + b * P.mask_bs : nullptr;
  const float4 q4 = *reinterpret_cast<const float4*>(P.q + b * P.q_bs + head * DK + 4 * r);
  const float scale = P.scale;
  float m = -INFINITY, l = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = 0; j0 < nk; j0 += G * CK) {        // (uniform for the workgroup; one trip up to G*CK keys)
    float4 kr[CK], vr[CK];
    uint8_t mb[CK];
#pragma unroll
    for (int c = 0; c < CK; ++c) {                 // unconditional loads of a clamped row: the waits sink to the first use
      const int j = j0 + c * G + g, jc = j < nk ? j : nk - 1;
      kr[c] = *reinterpret_cast<const float4*>(Kg + (int64_t)jc * P.ldk);
      vr[c] = *reinterpret_cast<const float4*>(Vg + (int64_t)jc * P.ldv);
      mb[c] = mrow != nullptr ? mrow[jc] : (uint8_t)0;
    }
    float sc[CK], mc = -INFINITY;
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const int j = j0 + c * G + g;
      float t = kr[c].x * q4.x + kr[c].y * q4.y + kr[c].z * q4.z + kr[c].w * q4.w;
      t = lanes_sum<LPR>(t) * scale;
      sc[c] = (j < nk && mb[c] == 0) ? t : -INFINITY;
      mc = fmaxf(mc, sc[c]);
    }
    if (mc > -INFINITY) {                          // (else: nothing live in this chunk for this group)
      const float mn = fmaxf(m, mc), corr = __expf(m - mn);     // m = -inf: corr = 0
      l *= corr; acc.x *= corr; acc.y *= corr; acc.z *= corr; acc.w *= corr;
#pragma unroll
      for (int c = 0; c < CK; ++c) {
        const float p = __expf(sc[c] - mn);        // dead key: exp(-inf) = 0
        l += p;
        acc.x = fmaf(p, vr[c].x, acc.x); acc.y = fmaf(p, vr[c].y, acc.y);
        acc.z = fmaf(p, vr[c].z, acc.z); acc.w = fmaf(p, vr[c].w, acc.w);
      }
      m = mn;
    }
  }
  // the groups of this wavefront meet: common maximum, rescaled sums (lanes with the same r hold the same channels)
  float mw = m;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
  const float w = m > -INFINITY ? __expf(m - mw) : 0.f;
  l *= w; acc.x *= w; acc.y *= w; acc.z *= w; acc.w *= w;
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
  }
  if (lane < LPR) {
    *reinterpret_cast<float4*>(&acc_s[wave][4 * r]) = acc;
    if (lane == 0) { m_s[wave] = mw; l_s[wave] = l; }
  }
  __syncthreads();
  if (threadIdx.x < LPR) {                          // the four wavefronts meet
    const float M = fmaxf(fmaxf(m_s[0], m_s[1]), fmaxf(m_s[2], m_s[3]));
    float lt = 0.f;
    float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float wu = m_s[u] > -INFINITY ? __expf(m_s[u] - M) : 0.f;
      const float4 a = *reinterpret_cast<const float4*>(&acc_s[u][4 * r]);
      lt = fmaf(l_s[u], wu, lt);
      o4.x = fmaf(a.x, wu, o4.x); o4.y = fmaf(a.y, wu, o4.y); o4.z = fmaf(a.z, wu, o4.z); o4.w = fmaf(a.w, wu, o4.w);
    }
    const float inv = 1.f / lt;                     // every key masked: 0 / 0 = NaN, like the reference's softmax over -inf
    o4.x *= inv; o4.y *= inv; o4.z *= inv; o4.w *= inv;
    if (!(lt > 0.f)) o4 = make_float4(NAN, NAN, NAN, NAN);
    *reinterpret_cast<float4*>(P.out + b * P.o_bs + head * DK + 4 * r) = o4;
    if (threadIdx.x == 0 && P.lse != nullptr) P.lse[(int64_t)bh * P.nq] = M + __logf(lt);
  }
}

template <int DK>
int launch_q1(const AttnArgs& A, hipStream_t s) {
  constexpr int G = 1024 / DK;
  const dim3 grid(A.BH), block(256);
  if (A.nk <= 3 * G) hipLaunchKernelGGL((nast_mha_fwd_q1_kernel<DK, 3>), grid, block, 0, s, A);
  else if (A.nk <= 6 * G) hipLaunchKernelGGL((nast_mha_fwd_q1_kernel<DK, 6>), grid, block, 0, s, A);
  else hipLaunchKernelGGL((nast_mha_fwd_q1_kernel<DK, 9>), grid, block, 0, s, A);
  return 0;
}
// nq == 1 on a padded batch with more than 64 keys (fewer: the wavefront kernel below is already one round trip)
static int dispatch_q1(const AttnArgs& A, int dk, hipStream_t s) {
  static const bool on = [] { const char* e = getenv("NAST_ATTN_Q1"); return !e || atoi(e) != 0; }();   // A/B aid
  if (!on || A.nq != 1 || A.q_off != nullptr || A.nk <= 64) return 1;
  if ((A.ldk & 3) || (A.ldv & 3) || (A.k_bs & 3) || (A.v_bs & 3) || (A.q_bs & 3) || (A.o_bs & 3)) return 1;
  if ((reinterpret_cast<uintptr_t>(A.q) | reinterpret_cast<uintptr_t>(A.k) | reinterpret_cast<uintptr_t>(A.v) | reinterpret_cast<uintptr_t>(A.out)) & 15) return 1;
  switch (dk) {
    case 16: return launch_q1<16>(A, s);
    case 32: return launch_q1<32>(A, s);
    case 64: return launch_q1<64>(A, s);
  }
  return 1;
}

template <int DK>
int launch_long(const AttnArgs& A, bool bwd, hipStream_t s) {
  const dim3 grid(A.BH), block(256);
  if (bwd) hipLaunchKernelGGL((nast_mha_bwd_long_kernel<DK, LONG_NT>), grid, block, 0, s, A);
  else hipLaunchKernelGGL((nast_mha_fwd_long_kernel<DK, LONG_NT>), grid, block, 0, s, A);
  return 0;
}
static int dispatch_long(const AttnArgs& A, int dk, bool bwd, hipStream_t s) {
  if (A.nq > 16 * LONG_NT || A.nk > 16 * LONG_NT) return 1;
  const char* e = getenv("NAST_ATTN_LONG");            // A/B aid: 0 = the LDS VALU kernels of attention.hip (padded batches only)
  if (e && atoi(e) == 0) return 1;
  switch (dk) {
    case 16: return launch_long<16>(A, bwd, s);
    case 32: return launch_long<32>(A, bwd, s);
    case 64: return launch_long<64>(A, bwd, s);
  }
  return 1;
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
  static const bool split_ok = [] { const char* e = getenv("NAST_ATTN_SPLIT"); return !e || atoi(e) != 0; }();   // A/B aid
  // (measured, 4 heads of 64: split wins 10-16 % from ~1000 items on; on a few hundred items, or at 49-64 positions where a
  // wavefront carries two tiles per side either way, the extra read of O makes it 1-7 % slower)
  if (A.o_in != nullptr && split_ok && nt <= 3 && A.BH >= 1024) {
    switch (nt) {
      case 1: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 1, true>), grid, block, 0, s, A); break;
      case 2: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 2, true>), grid, block, 0, s, A); break;
      default: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 3, true>), grid, block, 0, s, A); break;
    }
    return 0;
  }
  switch (nt) {
    case 1: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 1, false>), grid, block, 0, s, A); break;
    case 2: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 2, false>), grid, block, 0, s, A); break;
    case 3: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 3, false>), grid, block, 0, s, A); break;
    default: hipLaunchKernelGGL((nast_mha_bwd_mfma_kernel<DK, 4, false>), grid, block, 0, s, A); break;
  }
  return 0;
}

}  // namespace

// Entry points used by attention.hip's dispatcher.  Return 1 when the shape is outside this file's limits.
int nast_mha_fwd_mfma(const float* q, const float* k, const float* v, const uint8_t* mask, float* out, float* lse, int B, int h, int nq,
                      int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                      int64_t mask_bs, int64_t mask_qs, float scale, hipStream_t s, const int32_t* q_off, const int32_t* k_off,
                      const int32_t* q_len, const int32_t* k_len, int causal, int q_total, int q_alloc, int k_total, int k_alloc) {
  AttnArgs A = {};
  A.q_off = q_off; A.k_off = k_off; A.q_len = q_len; A.k_len = k_len; A.causal = causal;
  A.q_total = q_total; A.q_alloc = q_alloc; A.k_total = k_total; A.k_alloc = k_alloc;
  A.q = q; A.k = k; A.v = v; A.mask = mask; A.out = out; A.lse = lse;
  A.BH = B * h; A.h = h; A.nq = nq; A.nk = nk; A.ldq = ldq; A.ldk = ldk; A.ldv = ldv; A.ldo = ldo;
  A.q_bs = q_bs; A.k_bs = k_bs; A.v_bs = v_bs; A.o_bs = o_bs; A.mask_bs = mask_bs; A.mask_qs = mask_qs; A.scale = scale;
  if (dispatch_q1(A, dk, s) == 0) return 0;
  if (nq > 64 || nk > 64) return dispatch_long(A, dk, false, s);
  const int nkt = (nk + 15) / 16;
  switch (dk) {
    case 8: return launch_fwd_nt<8>(A, nkt, s);
    case 16: return launch_fwd_nt<16>(A, nkt, s);
    case 32: return launch_fwd_nt<32>(A, nkt, s);
    case 64: return launch_fwd_nt<64>(A, nkt, s);
  }
  return 1;
}

int nast_mha_bwd_mfma(const float* q, const float* k, const float* v, const uint8_t* mask, const float* out, const float* dout,
                      const float* lse, float* dq, float* dk_, float* dv, int B, int h, int nq, int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs,
                      int64_t k_bs, int64_t v_bs, int64_t o_bs, int64_t mask_bs, int64_t mask_qs, float scale, hipStream_t s,
                      const int32_t* q_off, const int32_t* k_off, const int32_t* q_len, const int32_t* k_len, int causal, int q_total, int q_alloc, int k_total, int k_alloc) {
  AttnArgs A = {};
  A.q_off = q_off; A.k_off = k_off; A.q_len = q_len; A.k_len = k_len; A.causal = causal;
  A.q_total = q_total; A.q_alloc = q_alloc; A.k_total = k_total; A.k_alloc = k_alloc;
  A.q = q; A.k = k; A.v = v; A.mask = mask; A.dout = dout; A.o_in = out; A.lse_in = lse; A.dq = dq; A.dk = dk_; A.dv = dv;
  A.BH = B * h; A.h = h; A.nq = nq; A.nk = nk; A.ldq = ldq; A.ldk = ldk; A.ldv = ldv; A.ldo = ldo;
  A.q_bs = q_bs; A.k_bs = k_bs; A.v_bs = v_bs; A.o_bs = o_bs; A.mask_bs = mask_bs; A.mask_qs = mask_qs; A.scale = scale;
  if (nq > 64 || nk > 64) return dispatch_long(A, dk, true, s);
  const int nt = ((nq > nk ? nq : nk) + 15) / 16;
  switch (dk) {
    case 8: return launch_bwd_nt<8>(A, nt, s);
    case 16: return launch_bwd_nt<16>(A, nt, s);
    case 32: return launch_bwd_nt<32>(A, nt, s);
    case 64: return launch_bwd_nt<64>(A, nt, s);
  }
  return 1;
}
