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
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmLaunch {
  nast_gemm_desc d[NAST_GEMM_MAX_GROUP];
  int tile_start[NAST_GEMM_MAX_GROUP + 1];
  int tiles_m[NAST_GEMM_MAX_GROUP];
  int tiles_n[NAST_GEMM_MAX_GROUP];
  int splitk[NAST_GEMM_MAX_GROUP];
  int kt_per_split[NAST_GEMM_MAX_GROUP];
  int count;
};

template <int R, int BK>
struct TileGeom {
  static constexpr int KC_LD = BK + 4;   // k-contiguous image: [R][BK+4]
  static constexpr int MC_LD = R + 4;    // m-contiguous image: [BK][R+4]
  static constexpr int SZ = (R * KC_LD > BK * MC_LD) ? R * KC_LD : BK * MC_LD;
  static constexpr int NV = R * BK / 1024;  // float4 chunks per thread (256 threads)
};

// global -> registers.  `rows` bounds the tile's row index, K bounds k.
template <int R, int BK>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, int ld, bool kcontig, bool vec,
                                          int row0, int rows, int k0, int K, float4 (&reg)[TileGeom<R, BK>::NV]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < TileGeom<R, BK>::NV; ++j) {
    const int c = tid + 256 * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kcontig) {
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

template <int R, int BK>
__device__ __forceinline__ void tile_store(float* __restrict__ lds, bool kcontig,
                                           const float4 (&reg)[TileGeom<R, BK>::NV]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < TileGeom<R, BK>::NV; ++j) {
    const int c = tid + 256 * j;
    int off;
    if (kcontig) {
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
template <int R, int BK>
__device__ __forceinline__ float4 frag_read(const float* __restrict__ lds, bool kcontig, int row, int g, int h) {
  if (kcontig) {
    return *reinterpret_cast<const float4*>(lds + row * TileGeom<R, BK>::KC_LD + 8 * g + 4 * h);
  } else {
    const float* p = lds + (8 * g + 4 * h) * TileGeom<R, BK>::MC_LD + row;
    return make_float4(p[0], p[TileGeom<R, BK>::MC_LD], p[2 * TileGeom<R, BK>::MC_LD], p[3 * TileGeom<R, BK>::MC_LD]);
  }
}

template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void nast_gemm_f32_kernel(const GemmLaunch P) {
  using GA = TileGeom<BM, BK>;
  using GB = TileGeom<BN, BK>;
  constexpr int WTM = BM / 2, WTN = BN / 2, MT = WTM / 32, NT = WTN / 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * (GA::SZ + GB::SZ)];

  // ---- which problem / tile / k-slice ----
  const int bid = blockIdx.x;
  int gi = 0;
  while (gi + 1 < P.count && bid >= P.tile_start[gi + 1]) ++gi;
  const nast_gemm_desc& D = P.d[gi];
  const int local = bid - P.tile_start[gi];
  const int tn = local % P.tiles_n[gi];
  const int tq = local / P.tiles_n[gi];
  const int tm = tq % P.tiles_m[gi];
  const int kz = tq / P.tiles_m[gi];
  const int M = D.M, N = D.N, K = D.K;
  const int m0 = tm * BM, n0 = tn * BN;
  const bool a_kc = (D.transA == 0), b_kc = (D.transB != 0);
  const int nkt = (K + BK - 1) / BK;
  const int kt_lo = kz * P.kt_per_split[gi];
  int kt_hi = kt_lo + P.kt_per_split[gi];
  if (kt_hi > nkt) kt_hi = nkt;
  const int nsrc = D.nsrc;
  const int steps = (kt_hi - kt_lo) * nsrc;

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
  const bool do_colsum = (D.colsum != nullptr) && (tn == 0) && !a_kc;

  float4 ra[GA::NV], rb[GB::NV];
  auto issue = [&](int step) {
    const int s = step / (kt_hi - kt_lo);
    const int kt = kt_lo + step % (kt_hi - kt_lo);
    const float* Ap = D.A[s];
    const float* Bp = D.B[s];
    const bool va = ((D.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(Ap) & 15) == 0) && (a_kc ? (K & 3) == 0 : true);
    const bool vb = ((D.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(Bp) & 15) == 0) && (b_kc ? (K & 3) == 0 : true);
    tile_load<BM, BK>(Ap, D.lda, a_kc, va, m0, M, kt * BK, K, ra);
    tile_load<BN, BK>(Bp, D.ldb, b_kc, vb, n0, N, kt * BK, K, rb);
  };

  if (steps > 0) {
    issue(0);
    tile_store<BM, BK>(smem, a_kc, ra);
    tile_store<BN, BK>(smem + GA::SZ, b_kc, rb);
  }
  __syncthreads();

  for (int step = 0; step < steps; ++step) {
    const int cur = step & 1;
    const float* As = smem + cur * (GA::SZ + GB::SZ);
    const float* Bs = As + GA::SZ;
    if (step + 1 < steps) issue(step + 1);

    if (do_colsum && threadIdx.x < BM) {
#pragma unroll
      for (int k = 0; k < BK; ++k) colacc += As[k * GA::MC_LD + threadIdx.x];
    }
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 fa[MT], fb[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) fa[a] = frag_read<BM, BK>(As, a_kc, wm * WTM + a * 32 + li, g, lh);
#pragma unroll
      for (int b = 0; b < NT; ++b) fb[b] = frag_read<BN, BK>(Bs, b_kc, wn * WTN + b * 32 + li, g, lh);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
    }
    if (step + 1 < steps) {
      float* An = smem + (cur ^ 1) * (GA::SZ + GB::SZ);
      tile_store<BM, BK>(An, a_kc, ra);
      tile_store<BN, BK>(An + GA::SZ, b_kc, rb);
    }
    __syncthreads();
  }

  // ---- epilogue ----
  if (do_colsum && threadIdx.x < BM && (m0 + (int)threadIdx.x) < M && steps > 0)
    atomicAdd(D.colsum + m0 + threadIdx.x, colacc);

  const bool first = (kz == 0);
  const float alpha = D.alpha;
  const int flags = D.flags;
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int col = n0 + wn * WTN + b * 32 + li;
      if (col >= N) continue;
      const float bv = (D.bias != nullptr && first) ? D.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WTM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= M) continue;
        const int64_t o = (int64_t)row * D.ldc + col;
        float v = alpha * acc[a][b][r] + bv;
        if (D.residual != nullptr && first) v += D.residual[o];
        if (flags & NAST_GEMM_RELU) v = fmaxf(v, 0.f);
        if (D.relu_src != nullptr) v = D.relu_src[o] > 0.f ? v : 0.f;
        if (flags & NAST_GEMM_ATOMIC) atomicAdd(D.C + o, v);
        else if (flags & NAST_GEMM_ACCUM) D.C[o] += v;
        else D.C[o] = v;
      }
    }
}

template <int BM, int BN, int BK>
static int plan_and_launch(const nast_gemm_desc* descs, int count, hipStream_t stream, bool dry, int* total_tiles) {
  GemmLaunch P;
  P.count = count;
  int tiles = 0;
  for (int i = 0; i < count; ++i) {
    P.d[i] = descs[i];
    P.tiles_m[i] = (descs[i].M + BM - 1) / BM;
    P.tiles_n[i] = (descs[i].N + BN - 1) / BN;
    tiles += P.tiles_m[i] * P.tiles_n[i];
  }
  if (dry) { *total_tiles = tiles; return 0; }
  // split-K only for atomically accumulated outputs (wgrad): aim for >= ~2 workgroups per CU
  int want = tiles > 0 ? (512 + tiles - 1) / tiles : 1;
  if (want > 32) want = 32;
  int start = 0;
  for (int i = 0; i < count; ++i) {
    const int nkt = (descs[i].K + BK - 1) / BK;
    int sk = 1;
    if ((descs[i].flags & NAST_GEMM_ATOMIC) && descs[i].nsrc == 1 && descs[i].relu_src == nullptr &&
        !(descs[i].flags & NAST_GEMM_RELU)) {
      sk = want;
      const int max_sk = nkt / 4 > 0 ? nkt / 4 : 1;   // at least 4 K-tiles per slice
      if (sk > max_sk) sk = max_sk;
    }
    int per = (nkt + sk - 1) / sk;
    if (per < 1) per = 1;
    sk = (nkt + per - 1) / per;
    if (sk < 1) sk = 1;
    P.splitk[i] = sk;
    P.kt_per_split[i] = per;
    P.tile_start[i] = start;
    start += P.tiles_m[i] * P.tiles_n[i] * sk;
  }
  for (int i = count; i <= NAST_GEMM_MAX_GROUP; ++i) P.tile_start[i] = start;
  if (start == 0) return 0;
  hipLaunchKernelGGL((nast_gemm_f32_kernel<BM, BN, BK>), dim3(start), dim3(256), 0, stream, P);
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
  int tilesL = 0;
  plan_and_launch<128, 128, 16>(descs, count, stream, true, &tilesL);
  nast_prof_before(1, stream);
  if (tilesL >= 384) plan_and_launch<128, 128, 16>(descs, count, stream, false, nullptr);
  else plan_and_launch<64, 64, 32>(descs, count, stream, false, nullptr);
  nast_prof_after(1, stream, flops);
  NAST_CHECK_LAUNCH("nast_gemm_f32");
  return NAST_OK;
}
