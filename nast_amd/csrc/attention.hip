// Fused masked multi-head attention for short sequences (n <= ~160), fp32.
// One 256-thread workgroup per (batch, head): the whole K and V of that head live in LDS, queries are
// processed in chunks of R = 256/TPR rows with TPR lanes cooperating on one query row
// (scores: lanes stride over keys; P.V: lanes split the head dimension).  Softmax statistics are
// reduced with wavefront shuffles over the TPR-lane group.  Backward recomputes P from the saved
// log-sum-exp (no n x n tensor ever reaches HBM).
//
// Layout: q/k/v/out are the [B, n, h*dk] outputs/inputs of the projection GEMMs used as they are
// (head = column block), so the view/transpose/contiguous copies of the reference disappear.
#include "common.h"

template <int DK>
struct AttnGeom {
  static constexpr int LD = DK + 4;  // LDS row stride of K/V/Q/dO images (keeps 16-B alignment, staggers banks)
};

template <int TPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int TPR>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = TPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

template <int DK>
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, const float* __restrict__ src, int rows, int64_t ld) {
  // rows x DK floats, 16-B vectors, coalesced along the head dimension
  constexpr int V4 = DK / 4;
  for (int c = threadIdx.x; c < rows * V4; c += 256) {
    const int r = c / V4, q = (c % V4) * 4;
    *reinterpret_cast<float4*>(lds + r * AttnGeom<DK>::LD + q) = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + q);
  }
}

template <int DK, int TPR>
__global__ __launch_bounds__(256) void nast_mha_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const uint8_t* __restrict__ mask,
                                                           float* __restrict__ out, float* __restrict__ lse, int h, int nq,
                                                           int nk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs,
                                                           int64_t v_bs, int64_t o_bs, int64_t mask_bs, int64_t mask_qs, float scale) {
  constexpr int R = 256 / TPR, LD = AttnGeom<DK>::LD, CPT = DK / TPR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x / h, head = blockIdx.x % h;
  const int nkp = nk + 1;
  float* Ks = smem;
  float* Vs = Ks + nk * LD;
  float* Ps = Vs + nk * LD;
  stage_rows<DK>(Ks, k + b * k_bs + head * DK, nk, ldk);
  stage_rows<DK>(Vs, v + b * v_bs + head * DK, nk, ldv);
  __syncthreads();
  const int i = threadIdx.x / TPR, sub = threadIdx.x % TPR;
  for (int q0 = 0; q0 < nq; q0 += R) {
    const int qi = q0 + i;
    const bool valid = qi < nq;
    float qr[DK];
    if (valid) {
      const float* qp = q + b * q_bs + (int64_t)qi * ldq + head * DK;
#pragma unroll
      for (int c = 0; c < DK; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(qp + c);
        qr[c] = t.x * scale; qr[c + 1] = t.y * scale; qr[c + 2] = t.z * scale; qr[c + 3] = t.w * scale;
      }
    } else {
#pragma unroll
      for (int c = 0; c < DK; ++c) qr[c] = 0.f;
    }
    const uint8_t* mrow = mask ? mask + b * mask_bs + (valid ? qi : 0) * mask_qs : nullptr;
    float* prow = Ps + i * nkp;
    float mx = -INFINITY;
    for (int j = sub; j < nk; j += TPR) {
      const float* kr = Ks + j * LD;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DK; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(kr + c);
        s = fmaf(qr[c], t.x, s); s = fmaf(qr[c + 1], t.y, s); s = fmaf(qr[c + 2], t.z, s); s = fmaf(qr[c + 3], t.w, s);
      }
      if (mrow && mrow[j]) s = -INFINITY;
      prow[j] = s;
      mx = fmaxf(mx, s);
    }
    mx = group_max<TPR>(mx);
    float sum = 0.f;
    for (int j = sub; j < nk; j += TPR) {
      const float e = __expf(prow[j] - mx);
      prow[j] = e;
      sum += e;
    }
    sum = group_sum<TPR>(sum);
    const float inv = 1.f / sum;
    if (valid && sub == 0) lse[((int64_t)b * h + head) * nq + qi] = mx + __logf(sum);
    __syncthreads();
    float acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
    for (int j = 0; j < nk; ++j) {
      const float p = prow[j];
      const float* vr = Vs + j * LD + sub * CPT;
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[c] = fmaf(p, vr[c], acc[c]);
    }
    if (valid) {
      float* op = out + b * o_bs + (int64_t)qi * ldo + head * DK + sub * CPT;
#pragma unroll
      for (int c = 0; c < CPT; ++c) op[c] = acc[c] * inv;
    }
    __syncthreads();
  }
}

template <int DK, int TPR>
__global__ __launch_bounds__(256) void nast_mha_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const uint8_t* __restrict__ mask,
                                                           const float* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ dq, float* __restrict__ dk_, float* __restrict__ dv,
                                                           int h, int nq, int nk, int ldq, int ldk, int ldv, int ldo,
                                                           int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                                                           int64_t mask_bs, int64_t mask_qs, float scale) {
  constexpr int R = 256 / TPR, LD = AttnGeom<DK>::LD, CPT = DK / TPR;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x / h, head = blockIdx.x % h;
  const int nkp = nk + 1;
  float* Ks = smem;
  float* Vs = Ks + nk * LD;
  float* Qs = Vs + nk * LD;
  float* Gs = Qs + R * LD;      // dO chunk
  float* Ps = Gs + R * LD;      // P  [R][nkp]
  float* Ss = Ps + R * nkp;     // dS [R][nkp]
  stage_rows<DK>(Ks, k + b * k_bs + head * DK, nk, ldk);
  stage_rows<DK>(Vs, v + b * v_bs + head * DK, nk, ldv);
  const int i = threadIdx.x / TPR, sub = threadIdx.x % TPR;
  for (int q0 = 0; q0 < nq; q0 += R) {
    const int rows = min(R, nq - q0);
    __syncthreads();   // previous chunk done with Qs/Gs/Ps/Ss (and K/V staged on the first pass)
    stage_rows<DK>(Qs, q + b * q_bs + (int64_t)q0 * ldq + head * DK, rows, ldq);
    stage_rows<DK>(Gs, dout + b * o_bs + (int64_t)q0 * ldo + head * DK, rows, ldo);
    __syncthreads();
    const int qi = q0 + i;
    const bool valid = i < rows;
    float qr[DK], gr[DK];
#pragma unroll
    for (int c = 0; c < DK; ++c) {
      qr[c] = valid ? Qs[i * LD + c] * scale : 0.f;
      gr[c] = valid ? Gs[i * LD + c] : 0.f;
    }
    const float l = valid ? lse[((int64_t)b * h + head) * nq + qi] : 0.f;
    const uint8_t* mrow = mask ? mask + b * mask_bs + (valid ? qi : 0) * mask_qs : nullptr;
    float* prow = Ps + i * nkp;
    float* srow = Ss + i * nkp;
    float dsum = 0.f;
    for (int j = sub; j < nk; j += TPR) {
      const float* kr = Ks + j * LD;
      const float* vr = Vs + j * LD;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < DK; c += 4) {
        const float4 t = *reinterpret_cast<const float4*>(kr + c);
        const float4 u = *reinterpret_cast<const float4*>(vr + c);
        s = fmaf(qr[c], t.x, s); s = fmaf(qr[c + 1], t.y, s); s = fmaf(qr[c + 2], t.z, s); s = fmaf(qr[c + 3], t.w, s);
        dp = fmaf(gr[c], u.x, dp); dp = fmaf(gr[c + 1], u.y, dp); dp = fmaf(gr[c + 2], u.z, dp); dp = fmaf(gr[c + 3], u.w, dp);
      }
      const bool masked = (mrow && mrow[j]) || !valid;
      const float p = masked ? 0.f : __expf(s - l);
      prow[j] = p;
      srow[j] = dp;
      dsum = fmaf(p, dp, dsum);
    }
    dsum = group_sum<TPR>(dsum);
    for (int j = sub; j < nk; j += TPR) srow[j] = prow[j] * (srow[j] - dsum) * scale;
    __syncthreads();
    // dQ[i] = sum_j dS[i][j] K[j]
    {
      float acc[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
      for (int j = 0; j < nk; ++j) {
        const float s = srow[j];
        const float* kr = Ks + j * LD + sub * CPT;
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[c] = fmaf(s, kr[c], acc[c]);
      }
      if (valid) {
        float* op = dq + b * q_bs + (int64_t)qi * ldq + head * DK + sub * CPT;
#pragma unroll
        for (int c = 0; c < CPT; ++c) op[c] = acc[c];
      }
    }
    // dK[j] += sum_i dS[i][j] Q[i] ;  dV[j] += sum_i P[i][j] dO[i]   (thread = (key j, column group))
    for (int j = i; j < nk; j += R) {
      float ak[CPT], av[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) { ak[c] = 0.f; av[c] = 0.f; }
      for (int r = 0; r < rows; ++r) {
        const float s = Ss[r * nkp + j], p = Ps[r * nkp + j];
        const float* qv = Qs + r * LD + sub * CPT;
        const float* gv = Gs + r * LD + sub * CPT;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          ak[c] = fmaf(s, qv[c], ak[c]);
          av[c] = fmaf(p, gv[c], av[c]);
        }
      }
      float* kp = dk_ + b * k_bs + (int64_t)j * ldk + head * DK + sub * CPT;
      float* vp = dv + b * v_bs + (int64_t)j * ldv + head * DK + sub * CPT;
      if (q0 == 0) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) { kp[c] = ak[c]; vp[c] = av[c]; }
      } else {
#pragma unroll
        for (int c = 0; c < CPT; ++c) { kp[c] += ak[c]; vp[c] += av[c]; }
      }
    }
  }
}

static int check_common(const char* name, int B, int h, int nq, int nk, int dk, int ldq, int ldk, int ldv, int ldo) {
  NAST_REQUIRE(B >= 0 && h >= 1 && nq >= 1 && nk >= 1, "%s: bad sizes B=%d h=%d nq=%d nk=%d", name, B, h, nq, nk);
  NAST_REQUIRE(dk == 8 || dk == 16 || dk == 32 || dk == 64, "%s: head dim %d unsupported (8/16/32/64)", name, dk);
  NAST_REQUIRE((ldq % 4 | ldk % 4 | ldv % 4 | ldo % 4) == 0, "%s: row strides must be multiples of 4 floats", name);
  NAST_REQUIRE(ldq >= h * dk && ldk >= h * dk && ldv >= h * dk && ldo >= h * dk, "%s: row stride smaller than h*dk", name);
  return NAST_OK;
}

#define LDS_LIMIT (160 * 1024)

// matrix-core kernels for nq, nk <= 64 (attention_mfma.hip); return 1 when the shape is outside their limits
int nast_mha_fwd_mfma(const float* q, const float* k, const float* v, const uint8_t* mask, float* out, float* lse, int B, int h, int nq,
                      int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                      int64_t mask_bs, int64_t mask_qs, float scale, hipStream_t s, const int32_t* q_off = nullptr,
                      const int32_t* k_off = nullptr, const int32_t* q_len = nullptr, const int32_t* k_len = nullptr, int causal = 0,
                      int q_total = 0, int q_alloc = 0, int k_total = 0, int k_alloc = 0);
int nast_mha_bwd_mfma(const float* q, const float* k, const float* v, const uint8_t* mask, const float* out, const float* dout,
                      const float* lse, float* dq, float* dk_, float* dv, int B, int h, int nq, int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs,
                      int64_t k_bs, int64_t v_bs, int64_t o_bs, int64_t mask_bs, int64_t mask_qs, float scale, hipStream_t s,
                      const int32_t* q_off = nullptr, const int32_t* k_off = nullptr, const int32_t* q_len = nullptr,
                      const int32_t* k_len = nullptr, int causal = 0, int q_total = 0, int q_alloc = 0, int k_total = 0, int k_alloc = 0);

template <int DK>
static int launch_fwd(const float* q, const float* k, const float* v, const uint8_t* mask, float* out, float* lse, int B,
                      int h, int nq, int nk, int ldq, int ldk, int ldv, int ldo, int64_t qb, int64_t kb, int64_t vb, int64_t ob,
                      int64_t mbs, int64_t mqs, float scale, hipStream_t s) {
  const int LD = AttnGeom<DK>::LD;
  int tpr = nq <= 32 ? 8 : 4;
  size_t lds = ((size_t)2 * nk * LD + (size_t)(256 / tpr) * (nk + 1)) * 4;
  if (lds > LDS_LIMIT) { tpr = 8; lds = ((size_t)2 * nk * LD + (size_t)32 * (nk + 1)) * 4; }
  NAST_REQUIRE(lds <= LDS_LIMIT, "nast_mha_fwd_f32: nk=%d needs %zu B of LDS (> 160 KiB)", nk, lds);
  if (tpr == 8) {
    auto kern = nast_mha_fwd_kernel<DK, 8>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), lds, s, q, k, v, mask, out, lse, h, nq, nk, ldq, ldk, ldv, ldo, qb, kb, vb, ob, mbs, mqs, scale);
  } else {
    auto kern = nast_mha_fwd_kernel<DK, 4>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), lds, s, q, k, v, mask, out, lse, h, nq, nk, ldq, ldk, ldv, ldo, qb, kb, vb, ob, mbs, mqs, scale);
  }
  NAST_CHECK_LAUNCH("nast_mha_fwd_f32");
  return NAST_OK;
}

template <int DK>
static int launch_bwd(const float* q, const float* k, const float* v, const uint8_t* mask, const float* dout,
                      const float* lse, float* dq, float* dk_, float* dv, int B, int h, int nq, int nk, int ldq, int ldk,
                      int ldv, int ldo, int64_t qb, int64_t kb, int64_t vb, int64_t ob, int64_t mbs, int64_t mqs, float scale,
                      hipStream_t s) {
  const int LD = AttnGeom<DK>::LD;
  auto need = [&](int R) { return ((size_t)2 * nk * LD + (size_t)2 * R * LD + (size_t)2 * R * (nk + 1)) * 4; };
  int tpr = (nq > 32 && need(64) <= 64 * 1024) ? 4 : 8;
  size_t lds = need(256 / tpr);
  NAST_REQUIRE(lds <= LDS_LIMIT, "nast_mha_bwd_f32: nk=%d needs %zu B of LDS (> 160 KiB)", nk, lds);
  if (tpr == 8) {
    auto kern = nast_mha_bwd_kernel<DK, 8>;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), lds, s, q, k, v, mask, dout, lse, dq, dk_, dv, h, nq, nk, ldq, ldk, ldv, ldo, qb, kb, vb, ob, mbs, mqs, scale);
  } else {
    auto kern = nast_mha_bwd_kernel<DK, 4>;
    hipLaunchKernelGGL(kern, dim3(B * h), dim3(256), lds, s, q, k, v, mask, dout, lse, dq, dk_, dv, h, nq, nk, ldq, ldk, ldv, ldo, qb, kb, vb, ob, mbs, mqs, scale);
  }
  NAST_CHECK_LAUNCH("nast_mha_bwd_f32");
  return NAST_OK;
}

extern "C" int nast_mha_fwd_f32(const float* q, const float* k, const float* v, const uint8_t* mask, float* out, float* lse,
                                int B, int h, int nq, int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs,
                                int64_t v_bs, int64_t o_bs, int64_t mask_bs, int64_t mask_qs, float scale, void* stream) {
  int rc = check_common("nast_mha_fwd_f32", B, h, nq, nk, dk, ldq, ldk, ldv, ldo);
  if (rc) return rc;
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(q && k && v && out && lse, "nast_mha_fwd_f32: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (((q_bs | k_bs | v_bs | o_bs) & 3) == 0 &&
      nast_mha_fwd_mfma(q, k, v, mask, out, lse, B, h, nq, nk, dk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s) == 0) {
    NAST_CHECK_LAUNCH("nast_mha_fwd_f32(mfma)");
    return NAST_OK;
  }
  switch (dk) {
    case 8: return launch_fwd<8>(q, k, v, mask, out, lse, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
    case 16: return launch_fwd<16>(q, k, v, mask, out, lse, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
    case 32: return launch_fwd<32>(q, k, v, mask, out, lse, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
    default: return launch_fwd<64>(q, k, v, mask, out, lse, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
  }
}

extern "C" int nast_mha_bwd_f32(const float* q, const float* k, const float* v, const uint8_t* mask, const float* out,
                                const float* dout, const float* lse, float* dq, float* dk_, float* dv, int B, int h, int nq,
                                int nk, int dk, int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs,
                                int64_t o_bs, int64_t mask_bs, int64_t mask_qs, float scale, void* stream) {
  int rc = check_common("nast_mha_bwd_f32", B, h, nq, nk, dk, ldq, ldk, ldv, ldo);
  if (rc) return rc;
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(q && k && v && dout && lse && dq && dk_ && dv, "nast_mha_bwd_f32: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  if (((q_bs | k_bs | v_bs | o_bs) & 3) == 0 &&
      nast_mha_bwd_mfma(q, k, v, mask, out, dout, lse, dq, dk_, dv, B, h, nq, nk, dk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s) == 0) {
    NAST_CHECK_LAUNCH("nast_mha_bwd_f32(mfma)");
    return NAST_OK;
  }
  switch (dk) {
    case 8: return launch_bwd<8>(q, k, v, mask, dout, lse, dq, dk_, dv, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
    case 16: return launch_bwd<16>(q, k, v, mask, dout, lse, dq, dk_, dv, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
    case 32: return launch_bwd<32>(q, k, v, mask, dout, lse, dq, dk_, dv, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
    default: return launch_bwd<64>(q, k, v, mask, dout, lse, dq, dk_, dv, B, h, nq, nk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, mask_bs, mask_qs, scale, s);
  }
}

// ---- packed ("varlen") rows -------------------------------------------------------------------------------------
// Sentence b owns rows q_off[b] .. q_off[b] + q_len[b] - 1 of q / out / dq and k_off[b] .. + k_len[b] - 1 of k / v / dk / dv
// (row pitch ld*, heads side by side in a row).  Every key of a sentence is visible to every query of it: the padding
// that a [B, n, d] batch carries and masks simply does not exist.  max_nq / max_nk bound the lengths (<= 144; beyond 64 positions the head size must be 16, 32 or 64).
extern "C" int nast_mha_fwd_varlen_f32(const float* q, const float* k, const float* v, float* out, float* lse, int B, int h, int max_nq,
                                       int max_nk, int dk, int ldq, int ldk, int ldv, int ldo, const int32_t* q_off,
                                       const int32_t* k_off, const int32_t* q_len, const int32_t* k_len, int causal, int q_total,
                                       int q_alloc, int k_total, int k_alloc, float scale, void* stream) {
  int rc = check_common("nast_mha_fwd_varlen_f32", B, h, max_nq, max_nk, dk, ldq, ldk, ldv, ldo);
  if (rc) return rc;
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(q && k && v && out && lse && q_off && k_off && q_len && k_len, "nast_mha_fwd_varlen_f32: NULL pointer");
  NAST_REQUIRE(max_nq <= 144 && max_nk <= 144, "nast_mha_fwd_varlen_f32: lengths up to 144 (got %d, %d)", max_nq, max_nk);
  NAST_REQUIRE(q_total >= 0 && q_alloc >= q_total && k_total >= 0 && k_alloc >= k_total, "nast_mha_fwd_varlen_f32: row counts");
  NAST_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0, "nast_mha_fwd_varlen_f32: row pitches must be multiples of 4");
  rc = nast_mha_fwd_mfma(q, k, v, nullptr, out, lse, B, h, max_nq, max_nk, dk, ldq, ldk, ldv, ldo, 0, 0, 0, 0, 0, 0, scale,
                         (hipStream_t)stream, q_off, k_off, q_len, k_len, causal, q_total, q_alloc, k_total, k_alloc);
  NAST_REQUIRE(rc == 0, "nast_mha_fwd_varlen_f32: head size %d not supported at these lengths (8 needs <= 64 positions)", dk);
  NAST_CHECK_LAUNCH("nast_mha_fwd_varlen_f32");
  return NAST_OK;
}

extern "C" int nast_mha_bwd_varlen_f32(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                       const float* lse, float* dq, float* dk_, float* dv, int B, int h, int max_nq, int max_nk, int dk, int ldq, int ldk, int ldv,
                                       int ldo, const int32_t* q_off, const int32_t* k_off, const int32_t* q_len,
                                       const int32_t* k_len, int causal, int q_total, int q_alloc, int k_total, int k_alloc,
                                       float scale, void* stream) {
  int rc = check_common("nast_mha_bwd_varlen_f32", B, h, max_nq, max_nk, dk, ldq, ldk, ldv, ldo);
  if (rc) return rc;
  if (B == 0) return NAST_OK;
  NAST_REQUIRE(q && k && v && dout && lse && dq && dk_ && dv && q_off && k_off && q_len && k_len, "nast_mha_bwd_varlen_f32: NULL pointer");
  NAST_REQUIRE(max_nq <= 144 && max_nk <= 144, "nast_mha_bwd_varlen_f32: lengths up to 144 (got %d, %d)", max_nq, max_nk);
  NAST_REQUIRE(q_total >= 0 && q_alloc >= q_total && k_total >= 0 && k_alloc >= k_total, "nast_mha_bwd_varlen_f32: row counts");
  NAST_REQUIRE(((ldq | ldk | ldv | ldo) & 3) == 0, "nast_mha_bwd_varlen_f32: row pitches must be multiples of 4");
  rc = nast_mha_bwd_mfma(q, k, v, nullptr, out, dout, lse, dq, dk_, dv, B, h, max_nq, max_nk, dk, ldq, ldk, ldv, ldo, 0, 0, 0, 0, 0, 0, scale,
                         (hipStream_t)stream, q_off, k_off, q_len, k_len, causal, q_total, q_alloc, k_total, k_alloc);
  NAST_REQUIRE(rc == 0, "nast_mha_bwd_varlen_f32: head size %d not supported at these lengths (8 needs <= 64 positions)", dk);
  NAST_CHECK_LAUNCH("nast_mha_bwd_varlen_f32");
  return NAST_OK;
}
