// Grouped fp32 GEMM on gfx950 MFMA (v_mfma_f32_32x32x2_f32), fused epilogue, optional split-K.
//
// One launch runs up to 8 independent problems (QKV projections, the K/V projections of all decoder
// layers, dgrad + wgrads of a block ...) so that the small d_model=256 projections of NAST fill the
// 256 CUs together instead of one ~20-workgroup launch each.
//
// Tile: BM x BN x BK per 256-thread workgroup (4 wavefronts as 2 x 2, each (BM/2) x (BN/2) =
// MT x NT MFMA tiles of 32x32, accumulators in registers).  Operands go global -> registers -> LDS
// (double-buffered, one barrier per K-tile); fragments are read so that lane (i = lane&31, h = lane>>5)
// supplies k = 8g + 4h + s for MFMA step s of k-group g -- A and B use the same k permutation, so the
// K-contiguous layouts are read with one ds_read_b128 per 8 k.
//
// Layout cases (run-time, per problem; uniform per workgroup):
//   A: transA=0 -> k-contiguous [M][K]   | transA=1 -> m-contiguous [K][M]   (wgrad: dY^T)
//   B: transB=1 -> k-contiguous [N][K]   | transB=0 -> n-contiguous [K][N]   (dgrad: W, wgrad: X)
#include <stdlib.h>

#include <string>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define NAST_GEMM_VEC_C (1 << 16)   // internal: C / bias / residual / relu_src allow 16-B accesses
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 pack_bf16(const float4& lo, const float4& hi) {
  bf16x8 r;
  r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
  r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
  return r;
}
#ifndef NAST_BK128
#define NAST_BK128 32
#endif
#ifndef NAST_GEMM_LAB
#define NAST_GEMM_LAB 0      // ablation builds only (tools/gemm_lab.sh): 1 = no global loads after the first K-tile, 2 = no LDS stores
#endif                       // after the first, 4 = fragments read once per K-tile only; results are then wrong by construction

#ifdef GEMM_STAMPS   // debug build only (tools/gemm_stamps.py): per-workgroup s_memtime stamps of the tiled kernel's segments
__device__ unsigned long long g_gemm_stamps[8192 * 8];
#define GEMM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_gemm_stamps[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int nast_gemm_stamps_read(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * n);
}
#else
#define GEMM_STAMP(k) do { } while (0)
#endif

struct GemmLaunch {
  nast_gemm_desc d[NAST_GEMM_MAX_GROUP];
  int tile_start[NAST_GEMM_MAX_GROUP + 1];
  int tiles_m[NAST_GEMM_MAX_GROUP];
  int tiles_n[NAST_GEMM_MAX_GROUP];
  int splitk[NAST_GEMM_MAX_GROUP];
  int kt_per_split[NAST_GEMM_MAX_GROUP];
  int count;
  int xcd_order;
  int swz[NAST_GEMM_MAX_GROUP];   // chunk length of the per-problem XCD swizzle (0 = none)
};

template <int R, int BK>
struct TileGeom {
  static constexpr int KC_LD = BK + 4;   // k-contiguous image: [R][BK+4]
  static constexpr int MC_LD = R + 4;    // m-contiguous image: [BK][R+4]
  static constexpr int SZ = (R * KC_LD > BK * MC_LD) ? R * KC_LD : BK * MC_LD;
  static constexpr int NV = R * BK / 1024;  // float4 chunks per thread (256 threads)
};

// global -> registers.  `rows` bounds the tile's row index, K bounds k.  KC = k-contiguous source.
template <int R, int BK, bool KC, bool FAST>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int ld, bool vec,
                                          int row0, int rows, int k0, int K, float4 (&reg)[TileGeom<R, BK>::NV]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < TileGeom<R, BK>::NV; ++j) {
    const int c = tid + 256 * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FAST) {
      // host-checked: 16-B aligned base, ld % 4 == 0, K % BK == 0 and (m-contiguous operands) rows % 4 == 0.
      // Rows past the edge are clamped to the last valid ones: they only feed output elements that the
      // epilogue never stores, and an unconditional load lets the wait sink down to the LDS store.
      if (KC) {
        const int row = c / (BK / 4), kc = (c % (BK / 4)) * 4;
        int gr = row0 + row;
        gr = gr < rows ? gr : rows - 1;
        v = *reinterpret_cast<const float4*>(src + (int64_t)gr * ld + k0 + kc);
      } else {
        const int krow = c / (R / 4), mc = (c % (R / 4)) * 4;
        int gr = row0 + mc;
        gr = gr < rows ? gr : rows - 4;
        v = *reinterpret_cast<const float4*>(src + (int64_t)(k0 + krow) * ld + gr);
      }
    } else if (KC) {
      const int row = c / (BK / 4), kc = (c % (BK / 4)) * 4;
      const int gr = row0 + row, gk = k0 + kc;
      if (gr < rows && gk < K) {
        const float* p = src + (int64_t)gr * ld + gk;
        if (vec) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
    } else {
      const int krow = c / (R / 4), mc = (c % (R / 4)) * 4;
      const int gk = k0 + krow, gr = row0 + mc;
      if (gk < K && gr < rows) {
        const float* p = src + (int64_t)gk * ld + gr;
        if (vec && gr + 3 < rows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (gr + 1 < rows) v.y = p[1];
          if (gr + 2 < rows) v.z = p[2];
          if (gr + 3 < rows) v.w = p[3];
        }
      }
    }
    reg[j] = v;
  }
}

template <int R, int BK, bool KC>
__device__ __forceinline__ void tile_store(float* __restrict__ lds, const float4 (&reg)[TileGeom<R, BK>::NV]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < TileGeom<R, BK>::NV; ++j) {
    const int c = tid + 256 * j;
    int off;
    if (KC) {
      const int row = c / (BK / 4), kc = (c % (BK / 4)) * 4;
      off = row * TileGeom<R, BK>::KC_LD + kc;
    } else {
      const int krow = c / (R / 4), mc = (c % (R / 4)) * 4;
      off = krow * TileGeom<R, BK>::MC_LD + mc;
    }
    *reinterpret_cast<float4*>(lds + off) = reg[j];
  }
}

// fragment for one 32-row MFMA tile and one 8-wide k-group: 4 values = steps s = 0..3
template <int R, int BK, bool KC>
__device__ __forceinline__ float4 frag_read(const float* __restrict__ lds, int row, int g, int h) {
  if (KC) {
    return *reinterpret_cast<const float4*>(lds + row * TileGeom<R, BK>::KC_LD + 8 * g + 4 * h);
  } else {
    const float* p = lds + (8 * g + 4 * h) * TileGeom<R, BK>::MC_LD + row;
    return make_float4(p[0], p[TileGeom<R, BK>::MC_LD], p[2 * TileGeom<R, BK>::MC_LD], p[3 * TileGeom<R, BK>::MC_LD]);
  }
}

// One output tile (and K slice) of one problem; operand layouts are compile-time so that the K loop is
// branch-free: all fragment reads of a K-tile are issued first, then the MFMAs run back to back.
template <int BM, int BN, int BK, bool AKC, bool BKC, bool FAST, bool BF16>
__device__ __forceinline__ void gemm_tile(const nast_gemm_desc& D, int tm, int tn, int kz, int kt_per_split, float* smem) {
  using GA = TileGeom<BM, BK>;
  using GB = TileGeom<BN, BK>;
  constexpr int WTM = BM / 2, WTN = BN / 2, MT = WTM / 32, NT = WTN / 32, NG = BK / 8;
  const int M = D.M, N = D.N, K = D.K;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = (K + BK - 1) / BK;
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

  float colacc = 0.f;
  const bool do_colsum = !AKC && (D.colsum != nullptr) && (tn == 0);
  const bool va = ((D.lda & 3) == 0) && (!AKC || (K & 3) == 0);
  const bool vb = ((D.ldb & 3) == 0) && (!BKC || (K & 3) == 0);

  // With one 32x32 tile per wavefront (64x64 workgroup tile) the epilogue operands (residual / ReLU mask) are
  // fetched NOW, in the ownership of the wide epilogue (row p*8 + lane/8, columns 4*(lane%8)..+3), and sit in
  // registers through the K loop: the epilogue then has no memory round trip before its stores.
  constexpr bool PREFETCH_EPI = FAST && (MT * NT == 1);
  float4 pre_r[4], pre_s[4];
  const bool epi_wide = FAST && (D.flags & NAST_GEMM_VEC_C) && !(D.flags & NAST_GEMM_ATOMIC);
  if (PREFETCH_EPI && epi_wide) {
    const int c4 = (lane & 7) * 4;
    const int col = n0 + wn * WTN + c4;
    const bool col_ok = col < N;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int row = m0 + wm * WTM + p * 8 + (lane >> 3);
      row = row < M ? row : M - 1;
      const int64_t o = (int64_t)row * D.ldc + (col_ok ? col : 0);
      pre_r[p] = (D.residual != nullptr && kz == 0) ? *reinterpret_cast<const float4*>(D.residual + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      pre_s[p] = D.relu_src != nullptr ? *reinterpret_cast<const float4*>(D.relu_src + o) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
  }

  float4 ra[GA::NV], rb[GB::NV];
  auto issue = [&](int step) {
    const int s = step / nk;
    const int kt = kt_lo + step - s * nk;
    const float* Ap = D.A[s];
    const float* Bp = D.B[s];
    tile_load<BM, BK, AKC, FAST>(Ap, D.lda, va && ((reinterpret_cast<uintptr_t>(Ap) & 15) == 0), m0, M, kt * BK, K, ra);
    tile_load<BN, BK, BKC, FAST>(Bp, D.ldb, vb && ((reinterpret_cast<uintptr_t>(Bp) & 15) == 0), n0, N, kt * BK, K, rb);
  };

  GEMM_STAMP(1);
  if (steps > 0) {
    issue(0);
    tile_store<BM, BK, AKC>(smem, ra);
    tile_store<BN, BK, BKC>(smem + GA::SZ, rb);
  }
  GEMM_STAMP(2);
  __syncthreads();
  GEMM_STAMP(3);

  for (int step = 0; step < steps; ++step) {
    const int cur = step & 1;
    const float* As = smem + cur * (GA::SZ + GB::SZ);
    const float* Bs = As + GA::SZ;
    if (step + 1 < steps && !(NAST_GEMM_LAB & 1)) issue(step + 1);

    // fragments are double-buffered across the 8-wide k-groups: the LDS reads of group g+1 are in flight
    // while the 4 x MT x NT MFMAs of group g issue
    float4 fa[2][MT], fb[2][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) fa[0][a] = frag_read<BM, BK, AKC>(As, wm * WTM + a * 32 + li, 0, lh);
#pragma unroll
    for (int b = 0; b < NT; ++b) fb[0][b] = frag_read<BN, BK, BKC>(Bs, wn * WTN + b * 32 + li, 0, lh);
    if (!AKC) {
      if (do_colsum) {   // bias gradient: all 256 threads share the column sums of this K-tile (m = tid % BM, a slice of k)
        constexpr int PARTS = 256 / BM, KP = BK / PARTS;
        const int cm = threadIdx.x % BM, k0c = (threadIdx.x / BM) * KP;
#pragma unroll
        for (int k = 0; k < KP; ++k) colacc += As[(k0c + k) * GA::MC_LD + cm];
      }
    }
    if constexpr (BF16) {
      // bf16 variant (north_star: "bf16 variant within 2e-2"): same fp32 LDS images and the same fragment reads; two
      // 8-wide k-groups (lane half h: k = 16 t + 4 h + s and 16 t + 8 + 4 h + s) are rounded to bf16 in registers and
      // feed ONE v_mfma_f32_32x32x16_bf16 (fp32 accumulate) -- A and B share the k assignment, so the product is exact
      // in its pairing.  16x the matrix rate of the fp32 instruction: the kernel becomes load-bound.
#pragma unroll
      for (int t = 0; t < NG / 2; ++t) {
        if (t > 0) {
#pragma unroll
          for (int a = 0; a < MT; ++a) fa[0][a] = frag_read<BM, BK, AKC>(As, wm * WTM + a * 32 + li, 2 * t, lh);
#pragma unroll
          for (int b = 0; b < NT; ++b) fb[0][b] = frag_read<BN, BK, BKC>(Bs, wn * WTN + b * 32 + li, 2 * t, lh);
        }
#pragma unroll
        for (int a = 0; a < MT; ++a) fa[1][a] = frag_read<BM, BK, AKC>(As, wm * WTM + a * 32 + li, 2 * t + 1, lh);
#pragma unroll
        for (int b = 0; b < NT; ++b) fb[1][b] = frag_read<BN, BK, BKC>(Bs, wn * WTN + b * 32 + li, 2 * t + 1, lh);
        if (t == 0 && step + 1 < steps) {
          float* An = smem + (cur ^ 1) * (GA::SZ + GB::SZ);
          tile_store<BM, BK, AKC>(An, ra);
          tile_store<BN, BK, BKC>(An + GA::SZ, rb);
        }
        bf16x8 a8[MT], b8[NT];
#pragma unroll
        for (int a = 0; a < MT; ++a) a8[a] = pack_bf16(fa[0][a], fa[1][a]);
#pragma unroll
        for (int b = 0; b < NT; ++b) b8[b] = pack_bf16(fb[0][b], fb[1][b]);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[a], b8[b], acc[a][b], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c = g & 1, n = c ^ 1;
      if (g + 1 < NG) {
        if (NAST_GEMM_LAB & 4) {
#pragma unroll
          for (int a = 0; a < MT; ++a) fa[n][a] = fa[c][a];
#pragma unroll
          for (int b = 0; b < NT; ++b) fb[n][b] = fb[c][b];
        } else {
#pragma unroll
        for (int a = 0; a < MT; ++a) fa[n][a] = frag_read<BM, BK, AKC>(As, wm * WTM + a * 32 + li, g + 1, lh);
#pragma unroll
        for (int b = 0; b < NT; ++b) fb[n][b] = frag_read<BN, BK, BKC>(Bs, wn * WTN + b * 32 + li, g + 1, lh);
        }
      }
      if (g == NG / 2 && step + 1 < steps && !(NAST_GEMM_LAB & 2)) {
        // the next tile's LDS image is written in the middle of this tile's MFMA stream (its buffer was last
        // read before the previous barrier), so the stores issue in the shadow of the matrix pipe
        float* An = smem + (cur ^ 1) * (GA::SZ + GB::SZ);
        tile_store<BM, BK, AKC>(An, ra);
        tile_store<BN, BK, BKC>(An + GA::SZ, rb);
      }
      // k-step outermost: consecutive MFMAs go to different accumulators (dependent distance MT*NT)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][a].x, fb[c][b].x, acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][a].y, fb[c][b].y, acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][a].z, fb[c][b].z, acc[a][b], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][a].w, fb[c][b].w, acc[a][b], 0, 0, 0);
    }
    }
    __syncthreads();
  }
  GEMM_STAMP(4);

  // ---- epilogue ----
  if (!AKC) {
    if (do_colsum && (m0 + (int)(threadIdx.x % BM)) < M && steps > 0)
      atomicAdd(D.colsum + m0 + threadIdx.x % BM, colacc);
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
  if (FAST && (flags & NAST_GEMM_VEC_C) && !(flags & NAST_GEMM_ATOMIC)) {
    // Wide epilogue: each 32x32 accumulator tile is transposed through a wave-private LDS patch (the operand
    // buffers are free after the loop's last barrier) so that a lane owns 4 consecutive columns: bias /
    // residual / ReLU-mask loads and the C store are 16-B accesses, 4 store instructions per tile instead of 16.
    constexpr int PLD = 36;
    float* stg = smem + wave * (32 * PLD);
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
          if (PREFETCH_EPI) {
            rv[p] = pre_r[p];
            sv[p] = pre_s[p];
          } else {
            rv[p] = has_res ? *reinterpret_cast<const float4*>(resid + o) : make_float4(0.f, 0.f, 0.f, 0.f);
            sv[p] = has_rsrc ? *reinterpret_cast<const float4*>(rsrc + o) : make_float4(1.f, 1.f, 1.f, 1.f);
          }
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
            *cp = v;   // (a non-temporal store here: 2-6 % off an isolated launch on rotated operands, nothing off the step)
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
      // gather the epilogue operands of all 16 rows first (clamped, unconditional), then combine and store:
      // one memory round trip per 32x32 tile instead of one per element
      float rv[16], sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = rbase + (r & 3) + 8 * (r >> 2);
        row = row < M ? row : M - 1;
        const int64_t o = (int64_t)row * ldc + colc;
        rv[r] = has_res ? resid[o] : 0.f;
        sv[r] = has_rsrc ? rsrc[o] : 1.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        const int64_t o = (int64_t)row * ldc + col;
        float v = alpha * acc[a][b][r] + bv + rv[r];
        if (flags & NAST_GEMM_RELU) v = fmaxf(v, 0.f);
        v = sv[r] > 0.f ? v : 0.f;
        if (col_ok && row < M) {
          if (flags & NAST_GEMM_ATOMIC) { if (!(NAST_GEMM_LAB & 8)) atomicAdd(Cp + o, v); }
          else if (flags & NAST_GEMM_ACCUM) Cp[o] += v;
          else Cp[o] = v;
        }
      }
    }
}

// which problem / output tile / K slice a workgroup takes (shared by the tile kernels)
struct TileCoord { int gi, tm, tn, kz; };
__device__ __forceinline__ TileCoord decode_tile(const GemmLaunch& P) {
  // Optional XCD-contiguous order (NAST_GEMM_XCD=1): workgroups are dealt round-robin over the 8 XCDs, so making the
  // logical tile index contiguous per XCD lets the N/BN tiles that re-read one A row panel share an L2.  Measured:
  // +1..8 % on single-problem launches in tools/gemm_bench, but -7 % on the whole step (74.5 -> 80.4 ms): in the
  // grouped launches (dgrad + wgrads) a contiguous slice per XCD gives the XCDs unequal work.  Default: the per-problem
  // chunked swizzle below (NAST_GEMM_XCD=3): same step time, L2-miss fetch traffic 204 -> 99 MB per launch.
  int bid = blockIdx.x;
  if (P.xcd_order) {
    const int n = gridDim.x, q = n >> 3, r = n & 7, x = blockIdx.x & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blockIdx.x >> 3);
  }
  int gi = 0;
  while (gi + 1 < P.count && bid >= P.tile_start[gi + 1]) ++gi;
  int local = bid - P.tile_start[gi];
  // Per-problem XCD swizzle (NAST_GEMM_XCD=3): inside chunks of 8 * tiles_n consecutive workgroups, the tiles_n tiles of one
  // A row panel go to the workgroups of ONE XCD (blockIdx % 8) instead of one tile to each XCD, so the panel is fetched
  // into one L2 instead of up to eight.  The launch order is only permuted inside a chunk: the balance of grouped
  // launches (heavy split-K members first) is untouched, unlike the global order above.
  if (const int C = P.swz[gi]) {
    const int s0 = P.tile_start[gi], sa = (s0 + 7) & ~7;
    const int rel = bid - sa;
    if (rel >= 0) {
      const int c = rel / C, r = rel - c * C;
      if ((c + 1) * C <= P.tile_start[gi + 1] - sa) local = (sa - s0) + c * C + (r & 7) * (C >> 3) + (r >> 3);
    }
  }
  TileCoord t;
  t.gi = gi;
  t.tn = local % P.tiles_n[gi];
  const int tq = local / P.tiles_n[gi];
  t.tm = tq % P.tiles_m[gi];
  t.kz = tq / P.tiles_m[gi];
  return t;
}

template <int BM, int BN, int BK, bool FAST, bool BF16>
__global__ __launch_bounds__(256) void nast_gemm_f32_kernel(const GemmLaunch P) {
  using GA = TileGeom<BM, BK>;
  using GB = TileGeom<BN, BK>;
  GEMM_STAMP(0);
  __shared__ __attribute__((aligned(16))) float smem[2 * (GA::SZ + GB::SZ)];
  const TileCoord tc = decode_tile(P);
  const int gi = tc.gi, tm = tc.tm, tn = tc.tn, kz = tc.kz;
  const nast_gemm_desc& D = P.d[gi];
  const int per = P.kt_per_split[gi];
  const bool a_kc = (D.transA == 0), b_kc = (D.transB != 0);
  if (a_kc && b_kc) gemm_tile<BM, BN, BK, true, true, FAST, BF16>(D, tm, tn, kz, per, smem);          // y = x W^T
  else if (a_kc && !b_kc) gemm_tile<BM, BN, BK, true, false, FAST, BF16>(D, tm, tn, kz, per, smem);   // dx = dy W, soft x E
  else if (!a_kc && !b_kc) gemm_tile<BM, BN, BK, false, false, FAST, BF16>(D, tm, tn, kz, per, smem); // dW = dy^T x
  else gemm_tile<BM, BN, BK, false, true, FAST, BF16>(D, tm, tn, kz, per, smem);
  GEMM_STAMP(5);
}

#include "gemm_bf16_tile.h"

__global__ __launch_bounds__(256, 2) void nast_gemm_bf16_kernel(const GemmLaunch P) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[bf16tile::SMEM];
  const TileCoord tc = decode_tile(P);
  const nast_gemm_desc& D = P.d[tc.gi];
  const int per = P.kt_per_split[tc.gi];
  const bool a_kc = (D.transA == 0), b_kc = (D.transB != 0);
  if (a_kc && b_kc) bf16tile::gemm_tile<true, true>(D, tc.tm, tc.tn, tc.kz, per, smem);
  else if (a_kc && !b_kc) bf16tile::gemm_tile<true, false>(D, tc.tm, tc.tn, tc.kz, per, smem);
  else if (!a_kc && !b_kc) bf16tile::gemm_tile<false, false>(D, tc.tm, tc.tn, tc.kz, per, smem);
  else bf16tile::gemm_tile<false, true>(D, tc.tm, tc.tn, tc.kz, per, smem);
}

// -------------------------------------------------------------------------------------------------
// "Skinny" variant for M <= 1024 in the y = x W^T layout (greedy skip decoding: M = batch).  Latency-, not
// MFMA-bound: a 32x32 output tile per workgroup, K split over the 4 wavefronts, and every wavefront loads
// its fragments straight from global memory into registers in bursts of 64 k (no LDS staging, no barrier
// in the K loop: one memory round trip per burst instead of one per K-tile), then the four partial
// accumulators are summed through LDS.  Lane (i, h) holds k = 16*dg + 8*h + s for MFMA step s of the
// 16-wide group dg (the same permutation for A and B), i.e. 32 contiguous bytes per lane.
// -------------------------------------------------------------------------------------------------
struct SkinnyLaunch {
  nast_gemm_desc d[NAST_GEMM_MAX_GROUP];
  int tile_start[NAST_GEMM_MAX_GROUP + 1];
  int tiles_n[NAST_GEMM_MAX_GROUP];
  int count;
};

__global__ __launch_bounds__(256) void nast_gemm_f32_skinny_kernel(const SkinnyLaunch P) {
  __shared__ __attribute__((aligned(16))) float red[4 * 32 * 33];
  __shared__ float ln_gb[2 * 256];      // LayerNorm gamma | beta (K <= 256 when fused)
  // (An XCD-contiguous tile order -- XCD x gets the x-th eighth of the tile list, so that its L2 pulls only its share of A --
  // measured no change: 8.35 vs 8.34 ms per decode.)
  const int bid = blockIdx.x;
  int gi = 0;
  while (gi + 1 < P.count && bid >= P.tile_start[gi + 1]) ++gi;
  const nast_gemm_desc& D = P.d[gi];
  const int local = bid - P.tile_start[gi];
  const int tn = local % P.tiles_n[gi], tm = local / P.tiles_n[gi];
  const int M = D.M, N = D.N, K = D.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int arow = tm * 32 + li, brow = tn * 32 + li;
  const bool aok = arow < M, bok = brow < N;
  const float* __restrict__ Ap = D.A[0] + (int64_t)(aok ? arow : 0) * D.lda;
  const float* __restrict__ Bp = D.B[0] + (int64_t)(bok ? brow : 0) * D.ldb;
  // K slice of this wavefront, in 16-wide groups
  const int ngroups = (K + 15) / 16;
  const int gper = (ngroups + 3) / 4;
  const int g_lo = wave * gper;
  int g_hi = g_lo + gper;
  if (g_hi > ngroups) g_hi = ngroups;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // Everything the kernel will need from memory is requested up front: in-kernel stamps (round 2) put 3.5 us of a 10-us launch
  // on the first operand round trip (activations written by the previous kernel come from the Infinity Cache), and every
  // later load that waits for its own round trip -- the LayerNorm scale / shift, the epilogue's bias and residual -- adds
  // 1.5 - 2 us on top.  gamma / beta go to LDS (one element per thread), bias / residual values to registers.
  const bool fuse_ln = D.ln_gamma != nullptr;       // host guarantees K <= 256: one burst per wavefront
  if (fuse_ln && (int)threadIdx.x < K) {
    ln_gb[threadIdx.x] = D.ln_gamma[threadIdx.x];
    ln_gb[256 + threadIdx.x] = D.ln_beta[threadIdx.x];
  }
  float pre_bias[4], pre_res[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;
    const int gr = tm * 32 + (idx >> 5), gc = tn * 32 + (idx & 31);
    const bool ok = gr < M && gc < N;
    pre_bias[e] = (ok && D.bias) ? D.bias[gc] : 0.f;
    pre_res[e] = (ok && D.residual) ? D.residual[(int64_t)gr * D.ldc + gc] : 0.f;
  }

  constexpr int CH = 4;   // groups per burst: 64 k
  auto load8 = [&](const float* __restrict__ base, bool ok, int g, float (&dst)[8]) {
    const int k0 = g * 16 + lh * 8;
    if (ok && g < g_hi && k0 + 8 <= K) {
      const float4 u = *reinterpret_cast<const float4*>(base + k0);
      const float4 w = *reinterpret_cast<const float4*>(base + k0 + 4);
      dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w; dst[4] = w.x; dst[5] = w.y; dst[6] = w.z; dst[7] = w.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = (ok && g < g_hi && k0 + e < K) ? base[k0 + e] : 0.f;
    }
  };
  float fa[CH][8], fb[CH][8];
  for (int g0 = g_lo; g0 < g_hi || (fuse_ln && g0 == g_lo); g0 += CH) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      load8(Ap, aok, g0 + c, fa[c]);
      load8(Bp, bok, g0 + c, fb[c]);
    }
    if (fuse_ln) {
      // LayerNorm of the A rows inside the GEMM: row statistics are reduced over the two lane halves and the four
      // wavefronts (each holds a K slice of row li) through LDS; two-pass mean / variance like nast_ln_fwd_kernel
      float s1 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) s1 += fa[c][e];
      s1 += __shfl_xor(s1, 32, 64);
      if (lh == 0) red[wave * 32 + li] = s1;
      __syncthreads();
      const float mu = (red[li] + red[32 + li] + red[64 + li] + red[96 + li]) / K;
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = (g0 + c) * 16 + lh * 8 + e;
          const float t = (g0 + c < g_hi && k < K) ? fa[c][e] - mu : 0.f;
          s2 += t * t;
        }
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) red[128 + wave * 32 + li] = s2;
      __syncthreads();
      const float rs = rsqrtf((red[128 + li] + red[160 + li] + red[192 + li] + red[224 + li]) / K + 1e-6f);
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = (g0 + c) * 16 + lh * 8 + e;
          const bool ok = aok && (g0 + c < g_hi) && k < K;
          fa[c][e] = ok ? (fa[c][e] - mu) * rs * ln_gb[k] + ln_gb[256 + k] : 0.f;
        }
      __syncthreads();   // red[] is reused by the accumulator reduction below
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][e], fb[c][e], acc, 0, 0, 0);
  }
  // cross-wavefront reduction: red[wave][row][col] (row stride 33)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    red[(wave * 32 + row) * 33 + li] = acc[r];
  }
  __syncthreads();
  const float alpha = D.alpha;
  const int flags = D.flags;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = threadIdx.x + 256 * e;        // 0..1023
    const int row = idx >> 5, col = idx & 31;
    const int gr = tm * 32 + row, gc = tn * 32 + col;
    if (gr >= M || gc >= N) continue;
    float v = red[row * 33 + col] + red[(32 + row) * 33 + col] + red[(64 + row) * 33 + col] + red[(96 + row) * 33 + col];
    v *= alpha;
    v += pre_bias[e];
    const int64_t o = (int64_t)gr * D.ldc + gc;
    v += pre_res[e];
    if (flags & NAST_GEMM_RELU) v = fmaxf(v, 0.f);
    if (flags & NAST_GEMM_ACCUM) D.C[o] += v;
    else D.C[o] = v;
  }
}

// Small-M GEMM for WIDE models (K >= 512): 16x16 output tiles on v_mfma_f32_16x16x4_f32.  At M = 64 (configs[4]'s decode) the
// 32x32 tiling above gives a projection of the d = 1024 model 64 workgroups -- a quarter of the chip's matrix pipes run 3.4 us
// of MFMAs each while the rest idle, and 4 or 16 wavefronts per tile make no difference to that (measured: 13.5 / 12.4 us per
// launch, weights hot or cold) -- where 16x16 tiles give it 256 (0.85 us of MFMAs per CU).  A wavefront owns a quarter of K and
// fetches ALL of its slice up to K = 1024 in one round trip: lane (r = lane & 15, g = lane >> 4) holds row r's columns
// 16 q + 4 g .. + 3 for MFMA steps 4 q .. 4 q + 3 (the same k permutation for both operands), i.e. the four lanes of a row read
// 64 contiguous bytes.  With K <= 1024 the LayerNorm of the A rows is fused as in the 32x32 kernel (statistics over the lanes of
// a row and the four wavefronts through LDS), which removes the separate LayerNorm launch in front of every projection.
typedef float f32x4g __attribute__((ext_vector_type(4)));
template <bool LN>
__global__ __launch_bounds__(256) void nast_gemm_f32_skinny16_kernel(const SkinnyLaunch P) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 17];
  __shared__ __attribute__((aligned(16))) float ln_gb[LN ? 2 * 1024 : 4];
  __shared__ float ln_red[2 * 4 * 16];
  const int bid = blockIdx.x;
  int gi = 0;
  while (gi + 1 < P.count && bid >= P.tile_start[gi + 1]) ++gi;
  const nast_gemm_desc& D = P.d[gi];
  const int local = bid - P.tile_start[gi];
  const int tn = local % P.tiles_n[gi], tm = local / P.tiles_n[gi];
  const int M = D.M, N = D.N, K = D.K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int arow = tm * 16 + r, brow = tn * 16 + r;
  const float* __restrict__ Ap = D.A[0] + (int64_t)(arow < M ? arow : M - 1) * D.lda + 4 * g;
  const float* __restrict__ Bp = D.B[0] + (int64_t)(brow < N ? brow : N - 1) * D.ldb + 4 * g;
  const int ngroups = K / 16;                       // host: K % 16 == 0
  const int gper = (ngroups + 3) / 4;
  const int q_lo = wave * gper;
  const int q_hi = q_lo + gper < ngroups ? q_lo + gper : ngroups;
  // epilogue operands of this thread's output element, requested with everything else
  const int orow = tm * 16 + ((int)threadIdx.x >> 4), ocol = tn * 16 + ((int)threadIdx.x & 15);
  const bool ook = orow < M && ocol < N;
  const float pre_bias = (ook && D.bias) ? D.bias[ocol] : 0.f;
  const float pre_res = (ook && D.residual) ? D.residual[(int64_t)orow * D.ldc + ocol] : 0.f;
  const bool fuse_ln = LN && D.ln_gamma != nullptr; // (uniform for the workgroup)  host: K <= 1024, one trip of the loop below
  if constexpr (LN) {
    if (fuse_ln) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = (int)threadIdx.x + 256 * e;
        if (k < K) { ln_gb[k] = D.ln_gamma[k]; ln_gb[1024 + k] = D.ln_beta[k]; }
      }
    }
  }
  constexpr int CH = 16;                            // 16-wide k groups per round trip: 256 k per wavefront
  f32x4g acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int q0 = q_lo; q0 < q_hi || (fuse_ln && q0 == q_lo); q0 += CH) {
    float4 fa[CH], fb[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {                  // unconditional loads of a clamped group; the spare ones are zeroed below
      const int q = q0 + c < q_hi ? q0 + c : (q_hi > q_lo ? q_hi - 1 : 0);
      fa[c] = *reinterpret_cast<const float4*>(Ap + 16 * q);
      fb[c] = *reinterpret_cast<const float4*>(Bp + 16 * q);
    }
    __builtin_amdgcn_sched_barrier(0);              // every load is in flight before the first MFMA (else the scheduler trickles them)
    if (fuse_ln) {
      float s1 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (q0 + c < q_hi) s1 += (fa[c].x + fa[c].y) + (fa[c].z + fa[c].w);
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      if (g == 0) ln_red[wave * 16 + r] = s1;
      __syncthreads();
      const float mu = ((ln_red[r] + ln_red[16 + r]) + (ln_red[32 + r] + ln_red[48 + r])) / K;
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (q0 + c < q_hi) {
          const float a = fa[c].x - mu, b = fa[c].y - mu, cc = fa[c].z - mu, d = fa[c].w - mu;
          s2 += (a * a + b * b) + (cc * cc + d * d);
        }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (g == 0) ln_red[64 + wave * 16 + r] = s2;
      __syncthreads();
      const float rs = rsqrtf(((ln_red[64 + r] + ln_red[80 + r]) + (ln_red[96 + r] + ln_red[112 + r])) / K + 1e-6f);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = 16 * (q0 + c < q_hi ? q0 + c : 0) + 4 * g;
        const float4 ga = *reinterpret_cast<const float4*>(&ln_gb[k]);
        const float4 be = *reinterpret_cast<const float4*>(&ln_gb[1024 + k]);
        fa[c].x = (fa[c].x - mu) * rs * ga.x + be.x; fa[c].y = (fa[c].y - mu) * rs * ga.y + be.y;
        fa[c].z = (fa[c].z - mu) * rs * ga.z + be.z; fa[c].w = (fa[c].w - mu) * rs * ga.w + be.w;
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (q0 + c >= q_hi) fa[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c & 1) {
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].x, fb[c].x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].y, fb[c].y, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].z, fb[c].z, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].w, fb[c].w, acc1, 0, 0, 0);
      } else {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].x, fb[c].x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].y, fb[c].y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].z, fb[c].z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[c].w, fb[c].w, acc0, 0, 0, 0);
      }
    }
  }
  // accumulator layout: register e of lane (r, g) is C[row 4 g + e][col r]
#pragma unroll
  for (int e = 0; e < 4; ++e) red[(wave * 16 + 4 * g + e) * 17 + r] = acc0[e] + acc1[e];
  __syncthreads();
  if (ook) {
    const int row = (int)threadIdx.x >> 4, col = (int)threadIdx.x & 15;
    float v = (red[row * 17 + col] + red[(16 + row) * 17 + col]) + (red[(32 + row) * 17 + col] + red[(48 + row) * 17 + col]);
    v = v * D.alpha + pre_bias + pre_res;
    if (D.flags & NAST_GEMM_RELU) v = fmaxf(v, 0.f);
    const int64_t o = (int64_t)orow * D.ldc + ocol;
    if (D.flags & NAST_GEMM_ACCUM) D.C[o] += v;
    else D.C[o] = v;
  }
}

// which small-M kernel: 16x16 tiles when the model is wide (every K >= 512, a multiple of 16) and the 32x32 tiling would leave
// the chip under two tiles per CU, or when a LayerNorm over 256 < K <= 1024 is to be fused (only the 16x16 kernel can)
static bool skinny16_able(const nast_gemm_desc* descs, int count) {
  for (int i = 0; i < count; ++i)
    if (descs[i].K < 512 || (descs[i].K & 15) || (descs[i].ln_gamma && descs[i].K > 1024)) return false;
  return true;
}
static bool skinny_wide_ln(const nast_gemm_desc* descs, int count) {
  for (int i = 0; i < count; ++i)
    if (descs[i].ln_gamma && descs[i].K > 256) return true;
  return false;
}
static bool skinny16_wanted(const nast_gemm_desc* descs, int count) {
  static const bool on = [] { const char* e = getenv("NAST_GEMM_SKINNY16"); return !e || atoi(e) != 0; }();   // A/B aid
  if (!skinny16_able(descs, count)) return false;
  int tiles32 = 0;
  for (int i = 0; i < count; ++i) tiles32 += ((descs[i].M + 31) / 32) * ((descs[i].N + 31) / 32);
  return skinny_wide_ln(descs, count) || (on && tiles32 <= 512);
}

static bool skinny_ok(const nast_gemm_desc* descs, int count) {
  for (int i = 0; i < count; ++i) {
    const nast_gemm_desc& D = descs[i];
    if (D.M > 1024 || D.transA != 0 || D.transB == 0 || D.nsrc != 1 || D.relu_src || D.colsum || (D.flags & NAST_GEMM_ATOMIC)) return false;
    if (D.ln_gamma && !D.ln_beta) return false;
    if ((D.lda & 3) || (D.ldb & 3) || (reinterpret_cast<uintptr_t>(D.A[0]) & 15) || (reinterpret_cast<uintptr_t>(D.B[0]) & 15)) return false;
  }
  // a LayerNorm over more than 256 columns can only be fused by the 16x16 kernel, which then takes the whole group
  if (skinny_wide_ln(descs, count) && !skinny16_able(descs, count)) return false;
  return true;
}

static void launch_skinny(const nast_gemm_desc* descs, int count, hipStream_t stream) {
  SkinnyLaunch P;
  P.count = count;
  const bool t16 = skinny16_wanted(descs, count);
  const int T = t16 ? 16 : 32;
  int start = 0;
  bool any_ln = false;
  for (int i = 0; i < count; ++i) {
    P.d[i] = descs[i];
    P.tiles_n[i] = (descs[i].N + T - 1) / T;
    P.tile_start[i] = start;
    start += P.tiles_n[i] * ((descs[i].M + T - 1) / T);
    any_ln = any_ln || descs[i].ln_gamma != nullptr;
  }
  for (int i = count; i <= NAST_GEMM_MAX_GROUP; ++i) P.tile_start[i] = start;
  if (start == 0) return;
  if (!t16) hipLaunchKernelGGL(nast_gemm_f32_skinny_kernel, dim3(start), dim3(256), 0, stream, P);
  else if (any_ln) hipLaunchKernelGGL(nast_gemm_f32_skinny16_kernel<true>, dim3(start), dim3(256), 0, stream, P);
  else hipLaunchKernelGGL(nast_gemm_f32_skinny16_kernel<false>, dim3(start), dim3(256), 0, stream, P);
}

bool nast_gemm_bstat_eligible(const nast_gemm_desc& D);                                   // gemm_bstat.hip
int nast_gemm_bstat_launch(const nast_gemm_desc* descs, int count, hipStream_t stream);

static bool xcd_split_too(int sk) {   // tuning aid: NAST_GEMM_XCD_SPLIT=0 keeps split-K problems out of the swizzle
  static const bool on = [] { const char* e = getenv("NAST_GEMM_XCD_SPLIT"); return e ? atoi(e) != 0 : true; }();
  return sk == 1 || on;
}

template <int BM, int BN, int BK>
static bool fast_ok(const nast_gemm_desc* descs, int count) {
  for (int i = 0; i < count; ++i) {
    const nast_gemm_desc& D = descs[i];
    if (D.M < 4 || D.N < 4 || D.K % BK != 0 || (D.lda & 3) || (D.ldb & 3)) return false;
    if (D.transA && (D.M & 3)) return false;
    if (!D.transB && (D.N & 3)) return false;
    for (int s = 0; s < D.nsrc; ++s)
      if ((reinterpret_cast<uintptr_t>(D.A[s]) & 15) || (reinterpret_cast<uintptr_t>(D.B[s]) & 15)) return false;
  }
  return true;
}

template <int BM, int BN, int BK, bool BF16TILE = false>
static int plan_and_launch(const nast_gemm_desc* descs, int count, hipStream_t stream, bool dry, int* total_tiles) {
  GemmLaunch P;
  P.count = count;
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    P.d[i] = descs[i];
    {
      const nast_gemm_desc& D = descs[i];
      auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
      if ((D.ldc & 3) == 0 && (D.N & 3) == 0 && al(D.C) && al(D.bias) && al(D.residual) && al(D.relu_src)) P.d[i].flags |= NAST_GEMM_VEC_C;
      else P.d[i].flags &= ~NAST_GEMM_VEC_C;
    }
    P.tiles_m[i] = (descs[i].M + BM - 1) / BM;
    P.tiles_n[i] = (descs[i].N + BN - 1) / BN;
    tiles += P.tiles_m[i] * P.tiles_n[i];
  }
  if (dry) { *total_tiles = tiles; return 0; }
  // split-K plan.  Work is counted in K-tiles per workgroup.  The group should (a) put >= ~3 workgroups on each
  // of the 256 CUs and (b) give every workgroup about the same amount of work, so a K-heavy member (a weight
  // gradient: K = all rows of the batch; the generator dgrad / soft-token embedding: K = vocabulary) is cut
  // into slices of about work / TARGET_WGS K-tiles.  (Until round 3 a slice was also capped at the length of the longest
  // member that cannot be cut -- "every workgroup the same length".  With a K = 256 dgrad beside it that cut an FFN weight
  // gradient into 39 slices, i.e. 41 MB of fp32 atomic adds per launch against the chip's ~1.3 TB/s atomic rate; long slices
  // are dispatched first and the short workgroups fill in behind them, so the cap bought no balance: 86.2 -> 89.4 TFLOP/s
  // over the step's GEMM launches without it.)  Atomically accumulated outputs are cut
  // directly; plain outputs are zeroed here and switched to atomic adds (bias/residual go to slice 0).
  constexpr int MIN_KT = 256 / BK;                       // at least K = 256 per slice
  constexpr int TARGET_WGS = (BM >= 128) ? 768 : 1024;
  int nkt_of[NAST_GEMM_MAX_GROUP];
  bool can_split[NAST_GEMM_MAX_GROUP];
  long long work = 0;
  int max_unsplit = 0;
  for (int i = 0; i < count; ++i) {
    const nast_gemm_desc& D = descs[i];
    nkt_of[i] = ((D.K + BK - 1) / BK) * D.nsrc;
    const bool plain = D.relu_src == nullptr && !(D.flags & NAST_GEMM_RELU) && D.nsrc == 1;
    static const int plain_min = [] { const char* e = getenv("NAST_GEMM_PLAIN_SPLIT_MIN"); return e ? atoi(e) : 8; }();   // tuning aid
    // (a plain output is cut only from K = 2048 up: the cut costs a zero-fill and atomic stores; at K = 1024 and ~620 tiles
    // (packed rows) the uncut launch measured 70 us against 79 us)
    // (a plain output is zero-filled before its slices are added: a residual that aliases C would be erased -- such a problem
    // is never cut)
    const bool alias = D.residual != nullptr && D.residual == D.C;
    can_split[i] = plain && ((D.flags & NAST_GEMM_ATOMIC) || (!alias && !(D.flags & NAST_GEMM_ACCUM) && nkt_of[i] >= plain_min * MIN_KT));
    work += (long long)P.tiles_m[i] * P.tiles_n[i] * nkt_of[i];
    if (!can_split[i] && nkt_of[i] > max_unsplit) max_unsplit = nkt_of[i];
  }
  static const int target_wgs_env = [] { const char* e = getenv("NAST_GEMM_TARGET_WGS"); return e ? atoi(e) : 0; }();   // tuning aid
  static const int split_cap = [] { const char* e = getenv("NAST_GEMM_SPLIT_CAP"); return e ? atoi(e) : 0; }();            // tuning aid: cap on a slice, in % of the longest uncut member (0 = none)
  const int target_wgs = target_wgs_env > 0 ? target_wgs_env : TARGET_WGS;
  long long target = (work + target_wgs - 1) / target_wgs;
  if (max_unsplit > 0 && split_cap > 0 && target > (long long)max_unsplit * split_cap / 100) target = (long long)max_unsplit * split_cap / 100;
  if (target < MIN_KT) target = MIN_KT;
  // longest workgroups first (dispatch follows blockIdx): the tail of the launch is then made of short ones
  int order[NAST_GEMM_MAX_GROUP];
  int per_of[NAST_GEMM_MAX_GROUP], sk_of[NAST_GEMM_MAX_GROUP];
  for (int i = 0; i < count; ++i) {
    const int nkt = (descs[i].K + BK - 1) / BK;
    int sk = 1;
    if (can_split[i]) {
      sk = (int)((nkt + target - 1) / target);
      if (sk < 1) sk = 1;
      if (!(descs[i].flags & NAST_GEMM_ATOMIC) && sk < 2) sk = 1;
    }
    int per = (nkt + sk - 1) / sk;
    if (per < 1) per = 1;
    sk = (nkt + per - 1) / per;
    if (sk < 1) sk = 1;
    if (sk >= 2 && !(descs[i].flags & NAST_GEMM_ATOMIC)) {
      if (descs[i].ldc == descs[i].N) (void)hipMemsetAsync(descs[i].C, 0, (size_t)descs[i].M * descs[i].N * 4, stream);
      else (void)hipMemset2DAsync(descs[i].C, (size_t)descs[i].ldc * 4, 0, (size_t)descs[i].N * 4, descs[i].M, stream);
      P.d[i].flags |= NAST_GEMM_ATOMIC;
    }
    per_of[i] = per; sk_of[i] = sk; order[i] = i;
  }
  // Round quantisation: all workgroup slots of the chip (SLOTS = resident workgroups per CU x 256) fill at once, so a
  // launch of q*SLOTS + r workgroups with a small r pays a whole extra round for the r stragglers (measured: 1056
  // against 1024 tiles costs +18 %).  When the group has atomically accumulated members, their slice count is
  // lowered (slices get a little longer) until the total drops to q*SLOTS.
  {
    constexpr int SLOTS = (BM >= 128) ? 512 : 1024;
    static const bool fit = [] { const char* e = getenv("NAST_GEMM_FIT"); return e ? atoi(e) != 0 : true; }();   // tuning aid
    int W = 0;
    for (int i = 0; i < count; ++i) W += P.tiles_m[i] * P.tiles_n[i] * sk_of[i];
    int r = W % SLOTS;
    if (fit && W > SLOTS && r > 0 && r <= SLOTS / 4) {
      for (int i = 0; i < count && r > 0; ++i) {
        if (!(descs[i].flags & NAST_GEMM_ATOMIC) || sk_of[i] < 2) continue;
        const int t = P.tiles_m[i] * P.tiles_n[i];
        if (t <= 0) continue;
        const int nkt = (descs[i].K + BK - 1) / BK;
        int drop = (r + t - 1) / t;
        if (drop > sk_of[i] / 4) drop = sk_of[i] / 4;          // never lengthen a slice by more than a third
        if (drop <= 0) continue;
        int sk = sk_of[i] - drop;
        const int per = (nkt + sk - 1) / sk;
        sk = (nkt + per - 1) / per;
        r -= (sk_of[i] - sk) * t;
        per_of[i] = per; sk_of[i] = sk;
      }
    }
  }
  for (int a = 1; a < count; ++a)     // insertion sort by per-workgroup work, descending
    for (int b = a; b > 0 && per_of[order[b]] * descs[order[b]].nsrc > per_of[order[b - 1]] * descs[order[b - 1]].nsrc; --b) {
      const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t;
    }
  GemmLaunch Q = P;
  int start = 0;
  for (int j = 0; j < count; ++j) {
    const int i = order[j];
    Q.d[j] = P.d[i];
    Q.tiles_m[j] = P.tiles_m[i];
    Q.tiles_n[j] = P.tiles_n[i];
    Q.splitk[j] = sk_of[i];
    Q.kt_per_split[j] = per_of[i];
    Q.tile_start[j] = start;
    start += P.tiles_m[i] * P.tiles_n[i] * sk_of[i];
  }
  P = Q;
  static const int xcd_order = [] { const char* e = getenv("NAST_GEMM_XCD"); return e ? atoi(e) : 3; }();   // tuning aid (0 off, 1/2 global order, 3 per-problem chunks)
  // 2 = only for launches whose workgroups all do the same amount of work (one problem, or equal unsplit problems)
  bool uniform = true;
  for (int i = 0; i < count; ++i)
    uniform = uniform && P.splitk[i] == 1 && P.d[i].M == P.d[0].M && P.d[i].N == P.d[0].N && P.d[i].K == P.d[0].K && P.d[i].nsrc == P.d[0].nsrc;
  P.xcd_order = xcd_order == 2 ? (uniform ? 1 : 0) : (xcd_order == 1 ? 1 : 0);
  for (int i = 0; i < NAST_GEMM_MAX_GROUP; ++i) P.swz[i] = 0;
  if (xcd_order == 3)
    for (int i = 0; i < count; ++i)
    {
      // unsplit: a "panel" = the tiles_n tiles that share an A row panel; split-K: all tiles of one K slice (they share its
      // A and B slices).  A panel goes to one XCD; a problem needs at least one full chunk of 8 panels to be touched at all.
      const int panel = P.splitk[i] == 1 ? P.tiles_n[i] : P.tiles_m[i] * P.tiles_n[i];
      const bool ok = P.splitk[i] == 1 ? (panel >= 2 && panel <= 32 && P.tiles_m[i] >= 16) : (panel >= 2 && panel <= 64 && P.splitk[i] >= 8);
      if (ok && xcd_split_too(P.splitk[i])) P.swz[i] = 8 * panel;
    }
  for (int i = count; i <= NAST_GEMM_MAX_GROUP; ++i) P.tile_start[i] = start;
  if (start == 0) return 0;
  bool bf16 = true;
  for (int i = 0; i < count; ++i) bf16 = bf16 && (descs[i].flags & NAST_GEMM_BF16);
  if constexpr (BF16TILE) {   // the caller has checked fast_ok<128, 128, 64> and that every member asks for bf16
    hipLaunchKernelGGL(nast_gemm_bf16_kernel, dim3(start), dim3(256), 0, stream, P);
    return 0;
  }
  if (fast_ok<BM, BN, BK>(descs, count)) {
    if (bf16) hipLaunchKernelGGL((nast_gemm_f32_kernel<BM, BN, BK, true, true>), dim3(start), dim3(256), 0, stream, P);
    else hipLaunchKernelGGL((nast_gemm_f32_kernel<BM, BN, BK, true, false>), dim3(start), dim3(256), 0, stream, P);
  } else {   // odd shapes / alignments: the general fp32 kernel (also for bf16 requests: more precise, never less)
    hipLaunchKernelGGL((nast_gemm_f32_kernel<BM, BN, BK, false, false>), dim3(start), dim3(256), 0, stream, P);
  }
  return 0;
}

extern "C" int nast_gemm_f32(const nast_gemm_desc* descs, int count, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  NAST_REQUIRE(descs != nullptr && count >= 1 && count <= NAST_GEMM_MAX_GROUP, "nast_gemm_f32: count %d out of range", count);
  double flops = 0;
  for (int i = 0; i < count; ++i) {
    const nast_gemm_desc& D = descs[i];
    NAST_REQUIRE(D.M >= 0 && D.N >= 0 && D.K >= 0, "nast_gemm_f32[%d]: negative dims", i);
    NAST_REQUIRE(D.nsrc >= 1 && D.nsrc <= NAST_GEMM_MAX_SRC, "nast_gemm_f32[%d]: nsrc %d", i, D.nsrc);
    NAST_REQUIRE(D.C != nullptr || D.M == 0 || D.N == 0, "nast_gemm_f32[%d]: C is NULL", i);
    for (int s = 0; s < D.nsrc; ++s)
      NAST_REQUIRE((D.A[s] != nullptr && D.B[s] != nullptr) || D.M == 0 || D.N == 0 || D.K == 0, "nast_gemm_f32[%d]: A/B[%d] NULL", i, s);
    NAST_REQUIRE(D.lda >= (D.transA ? D.M : D.K), "nast_gemm_f32[%d]: lda %d too small", i, D.lda);
    NAST_REQUIRE(D.ldb >= (D.transB ? D.K : D.N), "nast_gemm_f32[%d]: ldb %d too small", i, D.ldb);
    NAST_REQUIRE(D.ldc >= D.N, "nast_gemm_f32[%d]: ldc %d too small", i, D.ldc);
    NAST_REQUIRE(D.colsum == nullptr || D.transA, "nast_gemm_f32[%d]: colsum needs transA", i);
    NAST_REQUIRE(!((D.flags & NAST_GEMM_ATOMIC) && (D.flags & NAST_GEMM_ACCUM)), "nast_gemm_f32[%d]: ATOMIC and ACCUM are exclusive", i);
    flops += 2.0 * D.M * D.N * (double)D.K * D.nsrc;
  }
  const bool skinny = skinny_ok(descs, count);
  for (int i = 0; i < count; ++i)
    NAST_REQUIRE(descs[i].ln_gamma == nullptr || skinny, "nast_gemm_f32[%d]: fused LayerNorm needs the small-M path (M<=1024, x W^T, K<=256 or K<=1024 with K%%16==0)", i);
  auto signature = [&](const char* head, const nast_gemm_desc* ds, int n) {
    std::string sig = head;
    for (int i = 0; i < n; ++i) {
      const nast_gemm_desc& d = ds[i];
      char b[160];
      snprintf(b, sizeof(b), " [%dx%dx%d ta%d tb%d x%d f%d%s%s%s%s]", d.M, d.N, d.K, d.transA, d.transB, d.nsrc, d.flags,
               d.bias ? " b" : "", d.residual ? " r" : "", d.colsum ? " cs" : "", d.ln_gamma ? " ln" : "");
      sig += b;
    }
    return sig;
  };
  auto flops_of = [](const nast_gemm_desc* ds, int n) {
    double f = 0;
    for (int i = 0; i < n; ++i) f += 2.0 * ds[i].M * ds[i].N * (double)ds[i].K * ds[i].nsrc;
    return f;
  };
  if (skinny) {
    nast_prof_before(4, stream);
    launch_skinny(descs, count, stream);
    if (nast_prof_shapes(4)) nast_prof_after(4, stream, flops, signature("skinny", descs, count).c_str());
    else nast_prof_after(4, stream, flops);
    NAST_CHECK_LAUNCH("nast_gemm_f32");
    return NAST_OK;
  }
  // Opt-in (NAST_GEMM_BSTAT=1): the d_model-deep projections (K = 256, plain row-major A) go to the weight-stationary kernel
  // of gemm_bstat.hip, everything else of the group to the tiled kernel (two launches on the same stream).  Measured in
  // round 2 (tools/gemm_bench, tools/bstat_lab.cpp): its marginal rate is higher (128 against 109 TFLOP/s per added row on
  // x W1^T) but every launch first pays ~7 us for 512 workgroups pulling the same 128-KB weight panel through L2, so at the
  // step's 9.7k rows it ties on QKV / FFN-1 (47.0 vs 46.6 us, 55.3 vs 57.7 us) and loses on the 256-column projection (23.4
  // vs 20.9 us); it wins 10-15 % from ~19k rows.  Off by default for that reason.
  const char* bstat_env = getenv("NAST_GEMM_BSTAT");
  const bool bstat_on = bstat_env != nullptr && atoi(bstat_env) != 0;
  nast_gemm_desc ws[NAST_GEMM_MAX_GROUP], rest[NAST_GEMM_MAX_GROUP];
  int nws = 0, nrest = 0;
  for (int i = 0; i < count; ++i) {
    if (bstat_on && nast_gemm_bstat_eligible(descs[i])) ws[nws++] = descs[i];
    else rest[nrest++] = descs[i];
  }
  if (nws > 0) {
    nast_prof_before(1, stream);
    nast_gemm_bstat_launch(ws, nws, stream);
    if (nast_prof_shapes(1)) nast_prof_after(1, stream, flops_of(ws, nws), signature("ws", ws, nws).c_str());
    else nast_prof_after(1, stream, flops_of(ws, nws));
  }
  if (nrest > 0) {
    int tilesL = 0;
    plan_and_launch<128, 128, NAST_BK128>(rest, nrest, stream, true, &tilesL);
    nast_prof_before(1, stream);
    static const int force = [] { const char* e = getenv("NAST_GEMM_TILE"); return e ? atoi(e) : -1; }();   // tuning aid
    // measured (tools/gemm_bench, NAST_GEMM_TILE=0/2): the 64x64 tile at 4 workgroups per CU beats 128x128 at 2 on
    // every d_model-sized problem of the step (finer quantisation over 256 CUs, prologue/epilogue of one
    // workgroup hidden behind three others); 128x128 only wins (3-8 %) once there are thousands of tiles
    static const int t128_min = [] { const char* e = getenv("NAST_GEMM_T128_MIN"); return e ? atoi(e) : 4096; }();   // tuning aid
    // the bf16 variant's inner loop is ~3x shorter, so its time is operand traffic: the larger tile (half the re-reads)
    // wins from ~128 tiles on (32.8-33.7 against 33.6-34.5 ms per step at 512 sentences; run-to-run spread 0.5 ms)
    bool all_bf16 = true;
    for (int i = 0; i < nrest; ++i) all_bf16 = all_bf16 && (rest[i].flags & NAST_GEMM_BF16);
    int cfg = tilesL >= (all_bf16 ? std::min(t128_min, 128) : t128_min) ? 2 : 0;
    if (force >= 0) cfg = force;
    static const bool bf16_tile_on = [] { const char* e = getenv("NAST_GEMM_BF16_TILE"); return e ? atoi(e) != 0 : true; }();   // tuning aid: 0 = the fp32-image variant
    if (cfg == 2 && all_bf16 && bf16_tile_on && fast_ok<128, 128, 64>(rest, nrest)) plan_and_launch<128, 128, 64, true>(rest, nrest, stream, false, nullptr);
    else if (cfg == 2) plan_and_launch<128, 128, NAST_BK128>(rest, nrest, stream, false, nullptr);
    else plan_and_launch<64, 64, 32>(rest, nrest, stream, false, nullptr);
    if (nast_prof_shapes(1)) nast_prof_after(1, stream, flops_of(rest, nrest), signature(cfg == 2 ? "t128" : "t64", rest, nrest).c_str());
    else nast_prof_after(1, stream, flops_of(rest, nrest));
  }
  NAST_CHECK_LAUNCH("nast_gemm_f32");
  return NAST_OK;
}
