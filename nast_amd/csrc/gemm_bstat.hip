// "Weight-stationary" fp32 GEMM for the d_model-sized projections of the step:  C[M,N] = A[M,K] op(B) (+bias)(+residual)(ReLU)
// (mask) with K = 256 (one weight panel) -- QKV / output / FFN-1 projections (transformer.py:280-294, :308-312) and the
// data-gradient products of the same width.
//
// Why a second GEMM kernel: on these problems the tiled kernel of gemm_f32.hip spends a third of its time outside the matrix
// pipe (tools/mfma_lab.hip vs tools/gemm_lab.sh, round 2): a 64x64 output tile with K = 256 lives for 8 K-tiles only, so its
// prologue (first operand round trip), its epilogue and the per-K-tile staging of BOTH operands weigh as much as half of the
// MFMA time.  Here a workgroup is persistent and keeps its weight panel in REGISTERS for the whole launch:
//   * wave w of a workgroup owns 32 output columns (a panel of 128 per workgroup) and holds B[256 k][32 cols] as 128 VGPRs,
//     loaded once;
//   * the workgroup walks over 16-row tiles of A (tiles j, j+J, j+2J, ... of its column group): a tile is 16 rows x 1 KB,
//     fetched with one fully coalesced 1-KB wave-load per row a whole tile ahead (registers), written to an XOR-swizzled LDS
//     image (conflict-free ds_read_b128 fragments) and read by all four waves;
//   * per tile a wave issues 128 v_mfma_f32_16x16x4_f32 (two accumulators, one per 16-column half) with no barrier inside;
//     ONE barrier per tile hands the next A image over;
//   * the epilogue of a tile (bias / residual / ReLU / ReLU-mask, operands fetched before the MFMA loop) is 8 stores per lane.
// A is read once per 128-column panel, B once per workgroup: L2->CU traffic drops from 16 to 4 bytes per clock and CU.
// 16-row tiles keep the launch balanced: (M/16) x (N/128) tile units over 512 resident workgroups.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BStatProblem {
  const float* A; const float* B; float* C;
  const float* bias; const float* residual; const float* relu_src;
  int M, N, lda, ldb, ldc, transB, relu;
  float alpha;
};
struct BStatLaunch {
  BStatProblem p[NAST_GEMM_MAX_GROUP];
  int cg_start[NAST_GEMM_MAX_GROUP + 1];   // prefix sum of 128-column groups per problem
  int count, J;                            // J workgroups share a column group (stride of the tile walk)
#ifdef BSTAT_STAMPS
  unsigned long long* stamps;              // tools/bstat_lab.cpp: 16 s_memtime stamps per workgroup
#endif
};
#ifdef BSTAT_STAMPS
#define STAMP(k) do { if (threadIdx.x == 0 && (k) < 16) P.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// RES / MASK: the launch has a residual / a ReLU-mask operand (compile-time: their 8 registers each exist only when used)
template <bool RES, bool MASK>
__global__ __launch_bounds__(256, 2) void nast_gemm_bstat_kernel(const BStatLaunch P) {
  __shared__ __attribute__((aligned(16))) float lds[2][16 * 256];
  STAMP(0);
  const int cg = blockIdx.x / P.J, j = blockIdx.x - cg * P.J;
  int gi = 0;
  while (gi + 1 < P.count && cg >= P.cg_start[gi + 1]) ++gi;
  const BStatProblem& D = P.p[gi];
  const int M = D.M, lda = D.lda, ldc = D.ldc;
  const int mtiles = (M + 15) >> 4;
  if (j >= mtiles) return;                                    // (uniform per workgroup)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, q = lane >> 4;
  const int n0 = (cg - P.cg_start[gi]) * 128 + wave * 32;     // this wave's 32 columns
  const float* __restrict__ A = D.A;

  // ---- A tiles go global -> LDS directly (global_load_lds, 16 B per lane): no staging registers, no ds_write.  A wave-
  // instruction moves one 1-KB row; the LDS side is lane-linear, so the XOR swizzle of the image (16-B chunk ch of row r
  // sits at position ch ^ r: conflict-free fragment reads) is applied to the SOURCE address.  The first tile is requested
  // before anything else. ----
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[0][0];
  auto dma = [&](int mt, int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 4 * i;
      int gr = mt * 16 + row;
      gr = gr < M ? gr : M - 1;                                       // rows past the edge: clamped, never stored
      const float* src = A + (uint32_t)(gr * lda + ((lane ^ row) << 2));                  // (host-checked: M * lda < 2^31)
      // Issued as inline asm on purpose: behind the builtin hipcc (ROCm 7.2) waits vmcnt(0) in front of the NEXT LDS read --
      // the fragment reads of the tile being computed -- because it cannot tell the two images apart, and the transfer
      // would never overlap anything.  The asm is invisible to that pass; the waits are placed by hand (wait_dma).
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((st * 16 + row) * 1024));
#ifdef BSTAT_DMA_BUILTIN      // debugging aid: the compiler-tracked form (correct, but serialises the transfer with the tile's reads)
      (void)dst;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)&lds[st][row * 256], 16, 0, 0);
#else
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
#endif
    }
  };
  auto wait_dma = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  dma(j, 0);

  // ---- weight panel -> registers.  k permutation shared with the A fragments: MFMA (g, s) of a tile contracts
  // k = 16 g + 4 q + s over the four lane quarters q, so a lane's A fragment is one 16-B LDS read per g. ----
  // All workgroups of a column group read the SAME 128-KB panel at the same moment: taken in the same order, every CU asks
  // the same few L2 channels for the same lines (measured: 18k cycles to ISSUE the 32 loads).  Each workgroup therefore
  // walks the k groups from its own starting point: register slot g holds k group (g + rot) % 16, for B and for A alike.
  const int rot = blockIdx.x & 15;
  float4 breg[16][2];
  if (D.transB) {   // B stored [N][ldb >= K]: the lane's four k of group g are contiguous
    const float* __restrict__ Bp = D.B;
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        breg[g][t] = *reinterpret_cast<const float4*>(Bp + (uint32_t)((n0 + 16 * t + c) * D.ldb + 4 * q + 16 * ((g + rot) & 15)));
  } else {          // B stored [K][ldb >= N]
    const float* __restrict__ Bp = D.B + n0 + c;
    const int ldb = D.ldb;
#pragma unroll
    for (int g = 0; g < 16; ++g)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t o = (uint32_t)((16 * ((g + rot) & 15) + 4 * q) * ldb + 16 * t);
        breg[g][t] = make_float4(Bp[o], Bp[o + ldb], Bp[o + 2 * ldb], Bp[o + 3 * ldb]);
        __builtin_amdgcn_sched_barrier(0);      // addresses die with their loads (else 128 of them are live at once and spill)
      }
  }
  float bv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) bv[t] = D.bias ? D.bias[n0 + 16 * t + c] : 0.f;
  const float alpha = D.alpha;
  const float* __restrict__ resid = D.residual;
  const float* __restrict__ rsrc = D.relu_src;
  float* __restrict__ Cp = D.C;
  const bool relu = D.relu != 0;

  STAMP(1);
  wait_dma();                                                  // this wave's transfers, then everybody's
  __syncthreads();
  STAMP(2);
  int st = 0;
  int tile_no = 0;
  (void)tile_no;
  for (int mt = j; mt < mtiles; mt += P.J) {
    const int nxt = mt + P.J;
    if (nxt < mtiles) dma(nxt, st ^ 1);                        // a whole tile (128 MFMAs per wave) ahead; that image was
                                                               // last read before the previous barrier
    // epilogue operands of THIS tile: in flight during its MFMA loop.  Lane (c, q) owns rows 4q..4q+3, columns 16t + c.
    float rv[2][4], sv[2][4];
    const int row0 = mt * 16 + 4 * q;
    if (RES || MASK) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = row0 + r;
        row = row < M ? row : M - 1;
        const uint32_t o = (uint32_t)(row * ldc + n0 + c);                                 // (host-checked: M * ldc < 2^31)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (RES) rv[t][r] = resid[o + 16 * t];
          if (MASK) sv[t][r] = rsrc[o + 16 * t];
        }
      }
    }
    // The two waves of a SIMD (one per resident workgroup) alternate: while one runs its MFMA loop the other does its
    // epilogue, transfer issue and operand fetches.  Measured with in-kernel stamps (tools/bstat_lab.cpp): at equal priority
    // that short non-matrix section took ~4000 cycles instead of ~700, starved of issue slots by the partner's MFMA stream;
    // it runs at raised priority, the MFMA loop at the base priority.
    __builtin_amdgcn_s_setprio(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* img = &lds[st][c * 256];
    float4 fa[2];
    fa[0] = *reinterpret_cast<const float4*>(img + (((4 * rot + q) ^ c) << 2));
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (g + 1 < 16) fa[(g + 1) & 1] = *reinterpret_cast<const float4*>(img + (((4 * ((g + 1 + rot) & 15) + q) ^ c) << 2));
      const float4 f = fa[g & 1];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, breg[g][0].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.x, breg[g][1].x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, breg[g][0].y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.y, breg[g][1].y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, breg[g][0].z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.z, breg[g][1].z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, breg[g][0].w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w, breg[g][1].w, acc1, 0, 0, 0);
      // keep the fragment prefetch one group deep: without the fence the scheduler hoists all 16 LDS reads (64 registers)
      // and the weight panel spills
      __builtin_amdgcn_sched_barrier(0);
    }
    // The barrier comes BEFORE the epilogue's stores: every memory operation still outstanding here is a tile old (the next
    // image, the previous tile's stores), so the wait in front of the barrier costs nothing, and this tile's stores get a
    // whole tile to complete.
    __builtin_amdgcn_s_setprio(3);
    STAMP(3 + 3 * tile_no);
    wait_dma();
    __syncthreads();
    STAMP(4 + 3 * tile_no);
    st ^= 1;
    // ---- epilogue: C[row0 + r][n0 + 16 t + c] ----
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + r;
        float v = alpha * (t == 0 ? acc0[r] : acc1[r]) + bv[t];
        if (RES) v += rv[t][r];
        if (relu) v = fmaxf(v, 0.f);
        if (MASK) v = sv[t][r] > 0.f ? v : 0.f;
#ifdef BSTAT_NOSTORE      // tools/bstat_lab.cpp: what do the stores cost?
        if (v == 123.456f && row < M) Cp[(uint32_t)(row * ldc + n0 + 16 * t + c)] = v;
#else
        if (row < M) Cp[(uint32_t)(row * ldc + n0 + 16 * t + c)] = v;
#endif
      }
    STAMP(5 + 3 * tile_no);
    ++tile_no;
  }
}

// Which problems of a grouped launch this kernel takes.
bool nast_gemm_bstat_eligible(const nast_gemm_desc& D) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (D.K != 256 || D.transA != 0 || D.nsrc != 1 || D.M <= 1024 || D.N < 128 || (D.N & 127)) return false;
  if (D.flags & (NAST_GEMM_ATOMIC | NAST_GEMM_ACCUM | NAST_GEMM_BF16)) return false;
  if (D.colsum || D.ln_gamma || D.ln_beta) return false;
  if ((D.lda & 3) || !al(D.A[0])) return false;
  if ((int64_t)D.M * D.lda >= (1ll << 31) || (int64_t)D.M * D.ldc >= (1ll << 31) || (int64_t)D.N * D.ldb >= (1ll << 31)) return false;
  if (D.transB && ((D.ldb & 3) || !al(D.B[0]))) return false;
  return true;
}

int nast_gemm_bstat_launch(const nast_gemm_desc* descs, int count, hipStream_t stream) {
  BStatLaunch P;
  P.count = count;
  int cgs = 0, max_mt = 1;
  for (int i = 0; i < count; ++i) {
    const nast_gemm_desc& D = descs[i];
    BStatProblem& p = P.p[i];
    p.A = D.A[0]; p.B = D.B[0]; p.C = D.C; p.bias = D.bias; p.residual = D.residual; p.relu_src = D.relu_src;
    p.M = D.M; p.N = D.N; p.lda = D.lda; p.ldb = D.ldb; p.ldc = D.ldc; p.transB = D.transB;
    p.relu = (D.flags & NAST_GEMM_RELU) ? 1 : 0;
    p.alpha = D.alpha;
    P.cg_start[i] = cgs;
    cgs += D.N / 128;
    const int mt = (D.M + 15) / 16;
    if (mt > max_mt) max_mt = mt;
  }
  for (int i = count; i <= NAST_GEMM_MAX_GROUP; ++i) P.cg_start[i] = cgs;
  if (cgs == 0) return NAST_OK;
  // 2 workgroups per CU are resident (registers): 512 slots shared by the column groups
  int J = 512 / cgs;
  if (J < 1) J = 1;
  if (J > max_mt) J = max_mt;
  P.J = J;
  bool res = false, mask = false;
  for (int i = 0; i < count; ++i) {
    res = res || descs[i].residual != nullptr;
    mask = mask || descs[i].relu_src != nullptr;
  }
  // (a grouped launch whose members differ in these operands would read NULL: the caller groups like with like, checked here)
  for (int i = 0; i < count; ++i)
    if ((res && !descs[i].residual) || (mask && !descs[i].relu_src)) return NAST_EINVAL;
  const dim3 grid(cgs * J), block(256);
  if (res && mask) hipLaunchKernelGGL((nast_gemm_bstat_kernel<true, true>), grid, block, 0, stream, P);
  else if (res) hipLaunchKernelGGL((nast_gemm_bstat_kernel<true, false>), grid, block, 0, stream, P);
  else if (mask) hipLaunchKernelGGL((nast_gemm_bstat_kernel<false, true>), grid, block, 0, stream, P);
  else hipLaunchKernelGGL((nast_gemm_bstat_kernel<false, false>), grid, block, 0, stream, P);
  return NAST_OK;
}
