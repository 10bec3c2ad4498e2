// bf16 MFMA tile kernel of the grouped GEMM (included by gemm_f32.hip; north_star / BASELINE configs[4]: "bf16 MFMA").
//
// Operands stay fp32 in HBM (the model's tensors are fp32); they are rounded to bf16 ONCE, on the way into LDS, and the LDS
// images are bf16 and K-CONTIGUOUS for every source layout, so a fragment of v_mfma_f32_32x32x16_bf16 (8 consecutive k of one
// row) is one ds_read_b128.  Against the first bf16 variant (fp32 LDS images, two float4 fragment reads + 8 conversions per
// operand and MFMA, kept for shapes this kernel does not take): half the LDS bytes written, a quarter of the LDS bytes read
// per MFMA, conversions once per element instead of once per use.
//   tile 128 x 128 x 64 per 256-thread workgroup, 4 wavefronts as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles, fp32 accumulators;
//   image = [128 rows][64 bf16 + 8 pad] = 144-B rows: the b128 fragment reads (lane (i, h) -> row i, 16-B chunk 2t + h) and
//   both store patterns below are bank-conflict free (checked with the bank rules of MI355X_MICROARCH.md, LDS);
//   k-contiguous source ([rows][K] fp32): a thread takes (row, chunk) = (c / 8, c % 8), c = tid + 256 j: two float4 loads ->
//   one bf16x8 -> one ds_write_b128;
//   m-contiguous source ([K][rows] fp32: dY^T and X of a weight gradient, W of a dgrad): a thread takes 8 k-rows x 4 columns
//   (kg = tid % 8, column quad tid / 8): eight float4 loads (a wavefront reads eight full 128-B lines per instruction), the
//   transpose happens in registers, four ds_write_b128 (one per column, chunk kg);
//   double-buffered images (73.7 KB: two workgroups per CU), global loads of K-tile t + 1 in flight during the MFMAs of tile t,
//   the stores into the other buffer issued in the middle of the MFMA stream, one barrier per K-tile.
// Host side (plan_and_launch): all dims 16-B aligned, K % 64 == 0 (m-contiguous operands: rows % 4 == 0), else the older path.
#pragma once

namespace bf16tile {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int ROWB = 144;                    // bytes per image row
constexpr int IMG = 128 * ROWB;              // bytes per operand image
constexpr int SMEM = 4 * IMG;                // A | B, two buffers

// m-contiguous source: which 8 k-rows (kg) and which column quad (mq) a thread takes.  16 consecutive lanes cover 4 k-rows x
// 64 B (4 quads): 16 line accesses per wave instruction, what its 1 KB costs the L1 anyway (8 rows per 8 lanes measured 2-4x
// slower launches: 32 line accesses), and the ds_write_b128 of 8 consecutive lanes stay at most 2-way conflicted (16 LDS-array
// cycles against the 13 the instruction takes to hand over its data).
__device__ __forceinline__ int mc_kg(int tid) { return ((tid >> 2) & 3) + 4 * ((tid >> 4) & 1); }
__device__ __forceinline__ int mc_mq(int tid) { return (tid & 3) + 4 * (tid >> 5); }

// global -> registers: 8 float4 per operand and K-tile
template <bool KC>
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int ld, int row0, int rows, int k0, float4 (&r)[8]) {
  const int tid = threadIdx.x;
  if (KC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      int gr = row0 + (c >> 3);
      gr = gr < rows ? gr : rows - 1;                      // clamped rows feed outputs the epilogue never stores
      const float* p = src + (int64_t)gr * ld + k0 + (c & 7) * 8;
      r[2 * j] = *reinterpret_cast<const float4*>(p);
      r[2 * j + 1] = *reinterpret_cast<const float4*>(p + 4);
    }
  } else {
    const int kg = mc_kg(tid), mq = mc_mq(tid);
    int gm = row0 + 4 * mq;
    gm = gm < rows ? gm : rows - 4;
    const float* p = src + (int64_t)(k0 + 8 * kg) * ld + gm;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const float4*>(p + (int64_t)e * ld);
  }
}

__device__ __forceinline__ float comp(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// registers -> bf16 image
template <bool KC>
__device__ __forceinline__ void store_tile(unsigned char* __restrict__ img, const float4 (&r)[8]) {
  const int tid = threadIdx.x;
  if (KC) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tid + 256 * j;
      *reinterpret_cast<bf16x8*>(img + (c >> 3) * ROWB + (c & 7) * 16) = pack_bf16(r[2 * j], r[2 * j + 1]);
    }
  } else {
    const int kg = mc_kg(tid), mq = mc_mq(tid);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)comp(r[e], j);
      *reinterpret_cast<bf16x8*>(img + (4 * mq + j) * ROWB + kg * 16) = v;
    }
  }
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile(const nast_gemm_desc& D, int tm, int tn, int kz, int kt_per_split, unsigned char* smem) {
  constexpr int MT = 2, NT = 2, WTM = 64, WTN = 64;
  const int M = D.M, N = D.N, K = D.K;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = K / BK;
  const int kt_lo = kz * kt_per_split;
  int kt_hi = kt_lo + kt_per_split;
  if (kt_hi > nkt) kt_hi = nkt;
  const int nk = kt_hi - kt_lo;
  const int steps = nk * D.nsrc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // bias gradient (column sums of dY, the m-contiguous A of a weight gradient) from the fp32 registers, before rounding
  const bool do_colsum = !AKC && (D.colsum != nullptr) && (tn == 0);
  float colacc[4] = {0.f, 0.f, 0.f, 0.f};

  float4 ra[8], rb[8];
  auto issue = [&](int step) {
    const int s = step / nk;
    const int kt = kt_lo + step - s * nk;
    load_tile<AKC>(D.A[s], D.lda, m0, M, kt * BK, ra);
    load_tile<BKC>(D.B[s], D.ldb, n0, N, kt * BK, rb);
  };
  auto park = [&](int buf) {
    if (!AKC) {
      if (do_colsum) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { colacc[0] += ra[e].x; colacc[1] += ra[e].y; colacc[2] += ra[e].z; colacc[3] += ra[e].w; }
      }
    }
    store_tile<AKC>(smem + buf * 2 * IMG, ra);
    store_tile<BKC>(smem + buf * 2 * IMG + IMG, rb);
  };

  if (steps > 0) {
    issue(0);
    park(0);
  }
  __syncthreads();
  for (int step = 0; step < steps; ++step) {
    const int cur = step & 1;
    const unsigned char* As = smem + cur * 2 * IMG + (wm * WTM + li) * ROWB + lh * 16;
    const unsigned char* Bs = smem + cur * 2 * IMG + IMG + (wn * WTN + li) * ROWB + lh * 16;
    if (step + 1 < steps) issue(step + 1);
#pragma unroll
    for (int t = 0; t < BK / 16; ++t) {
      bf16x8 a8[MT], b8[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) a8[a] = *reinterpret_cast<const bf16x8*>(As + a * 32 * ROWB + t * 32);
#pragma unroll
      for (int b = 0; b < NT; ++b) b8[b] = *reinterpret_cast<const bf16x8*>(Bs + b * 32 * ROWB + t * 32);
      if (t == 2 && step + 1 < steps) park(cur ^ 1);       // the other buffer was last read before the previous barrier
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[a], b8[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue (as the fp32 kernel's) ----
  if (!AKC) {
    if (do_colsum && steps > 0) {
      // the 8 threads that share a column quad (kg: bits 2..4 of tid) are summed in registers first: one atomic per column
      // and workgroup, not eight on the same address (78k atomics on 256 addresses made these launches 3-4x longer)
      const int mq = mc_mq(threadIdx.x);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = colacc[j];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        if ((threadIdx.x & 28) == 0 && m0 + 4 * mq + j < M) atomicAdd(D.colsum + m0 + 4 * mq + j, v);
      }
    }
  }
  const bool first = (kz == 0);
  const float alpha = D.alpha;
  const int flags = D.flags;
  const float* __restrict__ bias = D.bias;
  const float* __restrict__ resid = D.residual;
  const float* __restrict__ rsrc = D.relu_src;
  float* __restrict__ Cp = D.C;
  const int ldc = D.ldc;
  const bool has_res = resid != nullptr && first;
  const bool has_rsrc = rsrc != nullptr;
  if ((flags & NAST_GEMM_VEC_C) && !(flags & NAST_GEMM_ATOMIC)) {
    // each 32x32 accumulator tile is transposed through a wave-private LDS patch (the operand buffers are free after the
    // loop's last barrier): bias / residual / ReLU-mask loads and the C store are 16-B accesses
    constexpr int PLD = 36;
    float* stg = reinterpret_cast<float*>(smem) + wave * (32 * PLD);
    const bool has_bias = bias != nullptr && first;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * PLD + li] = acc[a][b][r];
        const int c4 = (lane & 7) * 4;
        const int col = n0 + wn * WTN + b * 32 + c4;
        const bool col_ok = col < N;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bias && col_ok) bv = *reinterpret_cast<const float4*>(bias + col);
        float4 tv[4], rv[4], sv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int rl = p * 8 + (lane >> 3);
          tv[p] = *reinterpret_cast<const float4*>(stg + rl * PLD + c4);
          int row = m0 + wm * WTM + a * 32 + rl;
          row = row < M ? row : M - 1;
          const int64_t o = (int64_t)row * ldc + (col_ok ? col : 0);
          rv[p] = has_res ? *reinterpret_cast<const float4*>(resid + o) : make_float4(0.f, 0.f, 0.f, 0.f);
          sv[p] = has_rsrc ? *reinterpret_cast<const float4*>(rsrc + o) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int row = m0 + wm * WTM + a * 32 + p * 8 + (lane >> 3);
          float4 v;
          v.x = alpha * tv[p].x + bv.x + rv[p].x; v.y = alpha * tv[p].y + bv.y + rv[p].y;
          v.z = alpha * tv[p].z + bv.z + rv[p].z; v.w = alpha * tv[p].w + bv.w + rv[p].w;
          if (flags & NAST_GEMM_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          v.x = sv[p].x > 0.f ? v.x : 0.f; v.y = sv[p].y > 0.f ? v.y : 0.f;
          v.z = sv[p].z > 0.f ? v.z : 0.f; v.w = sv[p].w > 0.f ? v.w : 0.f;
          if (col_ok && row < M) {
            float4* cp = reinterpret_cast<float4*>(Cp + (int64_t)row * ldc + col);
            if (flags & NAST_GEMM_ACCUM) { const float4 c0 = *cp; v.x += c0.x; v.y += c0.y; v.z += c0.z; v.w += c0.w; }
            *cp = v;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int col = n0 + wn * WTN + b * 32 + li;
      const bool col_ok = col < N;
      const int colc = col_ok ? col : N - 1;
      const float bv = (bias != nullptr && first) ? bias[colc] : 0.f;
      const int rbase = m0 + wm * WTM + a * 32 + 4 * lh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const int rowc = row < M ? row : M - 1;
        const int64_t oc = (int64_t)rowc * ldc + colc;
        float v = alpha * acc[a][b][r] + bv + (has_res ? resid[oc] : 0.f);
        if (flags & NAST_GEMM_RELU) v = fmaxf(v, 0.f);
        v = (has_rsrc ? rsrc[oc] : 1.f) > 0.f ? v : 0.f;
        if (col_ok && row < M) {
          const int64_t o = (int64_t)row * ldc + col;
          if (flags & NAST_GEMM_ATOMIC) atomicAdd(Cp + o, v);
          else if (flags & NAST_GEMM_ACCUM) Cp[o] += v;
          else Cp[o] = v;
        }
      }
    }
}

}  // namespace bf16tile
