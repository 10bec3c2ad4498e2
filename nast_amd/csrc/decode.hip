// Greedy skip decoding of the LengthPredictor (transformer.py:525-570) driven from native code: ~26 kernel
// launches per step are enqueued back to back from C++ (no interpreter in the loop), with a real K/V cache
// instead of the reference's per-step re-projection of the whole prefix.
#include <atomic>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void lp_candidates_kernel(const float* __restrict__ oin, const int64_t* __restrict__ olen,
                                                            const float* __restrict__ eos, float* __restrict__ cand,
                                                            uint8_t* __restrict__ mask, int B, int S1, int d) {
  // cand[b,p] = origin_input[b,p] if p == 0 or (p < S1 and p-1 < len[b]) else eos   (transformer.py:471-473)
  // mask[b,p] = 1 where memory position p is padding (p >= 1 and p-1 >= len[b])       (transformer.py:459-461)
  const int lane = threadIdx.x & 63;
  const int NC = S1 + 5;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * NC) return;
  const int b = row / NC, p = row % NC;
  const bool live = p == 0 || (p < S1 && p - 1 < olen[b]);
  const float* src = live ? oin + ((int64_t)b * S1 + p) * d : eos;
  for (int c = lane; c < d; c += 64) cand[row * d + c] = src[c];
  if (lane == 0 && p < S1) mask[(int64_t)b * S1 + p] = live ? 0 : 1;
}

__global__ void lp_init_kernel(const float* __restrict__ oin, const float* __restrict__ pos_w, float* __restrict__ x,
                               int64_t* __restrict__ now_pos, int64_t* __restrict__ allpos, int B, int S1, int d, int ldp) {
  // step-0 input: candidates[b,0] + pos_embed[0]; now_pos = 0; allpos[:,0] = 0
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) x[(int64_t)b * d + c] = oin[(int64_t)b * S1 * d + c] + pos_w[c];
  if (threadIdx.x == 0) {
    now_pos[b] = 0;
    for (int j = 0; j < ldp; ++j) allpos[(int64_t)b * ldp + j] = 0;
  }
}

__global__ void lp_count_done_kernel(const int64_t* __restrict__ now_pos, const int64_t* __restrict__ olen, int* __restrict__ out, int B) {
  int n = 0;
  for (int b = threadIdx.x; b < B; b += 64) n += (now_pos[b] >= olen[b] + 1) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if (threadIdx.x == 0) *out = n;
}

inline size_t al(size_t n) { return (n + 63) / 64 * 64; }   // floats, keeps 256-B alignment

struct Ws {
  float *cand, *KC, *kunk, *mk, *mv, *kc, *vc, *x, *xn, *q, *a, *x1, *x2, *hmid, *qh, *lse, *stat;
  int64_t* now_pos;
  int* done;
  uint8_t* mask;
  unsigned* sync;
  size_t total;
};

Ws carve(float* base, int B, int S1, int d, int dff, int L, int nl, int h) {
  Ws w;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += al(n); return p; };
  const size_t NC = S1 + 5;
  w.cand = take((size_t)B * NC * d);
  w.KC = take((size_t)B * NC * d);
  w.kunk = take(d);
  w.mk = take((size_t)nl * B * S1 * d);
  w.mv = take((size_t)nl * B * S1 * d);
  w.kc = take((size_t)nl * B * L * d);
  w.vc = take((size_t)nl * B * L * d);
  w.x = take((size_t)B * d);
  w.xn = take((size_t)B * d);
  w.q = take((size_t)B * d);
  w.a = take((size_t)B * d);
  w.x1 = take((size_t)B * d);
  w.x2 = take((size_t)B * d);
  w.hmid = take((size_t)B * dff);
  w.qh = take((size_t)B * d);
  w.lse = take((size_t)B * h);
  w.stat = take(64);
  w.now_pos = reinterpret_cast<int64_t*>(take((size_t)B * 2));
  w.done = reinterpret_cast<int*>(take(16));
  w.mask = reinterpret_cast<uint8_t*>(take(((size_t)B * S1 + 3) / 4));
  w.sync = reinterpret_cast<unsigned*>(take(8 * 32));      // arrival counters of the cluster kernel (decode_cluster.hip)
  w.total = o * sizeof(float);
  return w;
}

nast_gemm_desc lin(const float* x, int M, int K, const float* W, const float* b, int N, float* y, int ldc) {
  nast_gemm_desc g = {};
  g.A[0] = x; g.B[0] = W; g.C = y; g.bias = b;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc; g.transA = 0; g.transB = 1; g.nsrc = 1; g.alpha = 1.f;
  return g;
}

}  // namespace

extern "C" int64_t nast_lp_decode_workspace_bytes(int B, int S1, int d, int dff, int L, int nlayers) {
  return (int64_t)carve(nullptr, B, S1, d, dff, L, nlayers, 64).total;
}

bool nast_lp_decode_fused_eligible(const nast_lp_decode_args* A);                                        // decode_fused.hip
int nast_lp_decode_fused_launch(const nast_lp_decode_args* A, const float* cand, const float* KC, const float* kunk, const float* mk,
                                const float* mv, float* kc, float* vc, int* steps_dev, hipStream_t s);

bool nast_lp_decode_cluster_eligible(const nast_lp_decode_args* A);                                      // decode_cluster.hip
int nast_lp_decode_cluster_launch(const nast_lp_decode_args* A, const float* cand, const float* KC, const float* kunk, const float* mk,
                                  const float* mv, float* kc, float* vc, float* x, float* q, float* a, float* x1, float* x2, float* hmid,
                                  float* qh, unsigned* sync, int* steps_dev, hipStream_t s);

#define TRY(expr)            \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

// The decode itself; `workspace` is nast_lp_decode_workspace_bytes() bytes of device memory, 256-B aligned.  Every word of
// it that a kernel reads is written earlier in the same call (candidate tables, K/V caches up to the current step, the
// position / progress counters by lp_init_kernel), so nothing has to start at zero.
static int lp_decode_impl(const nast_lp_decode_args* A, float* workspace, int* steps_out, hipStream_t s) {
  NAST_REQUIRE(A && steps_out, "nast_lp_decode_f32: NULL args");
  // *steps_out < 0 on entry: never read the progress back -- all L steps are enqueued and the call returns without
  // synchronising (positions of finished sentences stay put, so the result's leading columns are the same)
  const bool early_exit = *steps_out >= 0;
  const int B = A->B, S1 = A->S1, d = A->d, h = A->h, dff = A->dff, L = A->L, nl = A->nlayers;
  NAST_REQUIRE(B >= 1 && S1 >= 1 && d >= 1 && L >= 1 && nl >= 1 && nl <= 8 && h >= 1 && d % h == 0, "nast_lp_decode_f32: bad sizes");
  NAST_REQUIRE(h <= 64, "nast_lp_decode_f32: h > 64");
  NAST_REQUIRE(A->layers && workspace && A->memory && A->origin_length && A->allpos, "nast_lp_decode_f32: NULL pointer");
  NAST_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "nast_lp_decode_f32: workspace must be 256-B aligned");
  Ws w = carve(workspace, B, S1, d, dff, L, nl, h);
  const int NC = S1 + 5;
  const float scale = 1.0f / sqrtf((float)(d / h));
  // LayerNorm inside the projection that consumes it: the small-M GEMM's 32x32 kernel up to 256 columns.  Its 16x16 kernel
  // (wide models at a small batch) can fuse up to 1024 columns, but measured no better than the separate launch (19.0 us
  // against 4.8 + 11.4 per projection, configs[4] step 173 against 171.5 ms): every 16-column tile repeats the row statistics
  // and holds both operands in registers across two barriers.  Off unless NAST_DECODE_WIDE_LN=1.
  static const bool wide_ln = [] { const char* e = getenv("NAST_DECODE_WIDE_LN"); return e && atoi(e) != 0; }();
  const bool fuse_ln = (B <= 1024 && d <= 256 && (d & 3) == 0) || (wide_ln && B <= 256 && d >= 512 && d <= 1024 && (d & 15) == 0);

  // one kernel for the whole loop when the model has the shape it is built for (decode_fused.hip); else step by step
  const bool fused = nast_lp_decode_fused_eligible(A) && (reinterpret_cast<uintptr_t>(A->eos) & 15) == 0;
  hipLaunchKernelGGL(lp_candidates_kernel, dim3(((int64_t)B * NC + 3) / 4), dim3(256), 0, s, A->memory, A->origin_length, A->eos,
                     w.cand, w.mask, B, S1, d);
  // one persistent launch of 8 clusters x 32 workgroups with cluster barriers between the operations (decode_cluster.hip)
  const bool cluster = !fused && nast_lp_decode_cluster_eligible(A);
  if (!fused && !cluster)
    hipLaunchKernelGGL(lp_init_kernel, dim3(B), dim3(256), 0, s, A->memory, A->pos_w, w.x, w.now_pos, A->allpos, B, S1, d, L + 1);
  NAST_CHECK_LAUNCH("nast_lp_decode_f32(init)");
  {   // wk(candidates), wk(unk), cross-attention K/V of every layer: grouped launches
    nast_gemm_desc g[NAST_GEMM_MAX_GROUP];     // (fixed array: the call makes no host or device allocation)
    int ng = 0;
    g[ng++] = lin(w.cand, B * NC, d, A->wk_w, A->wk_b, d, w.KC, d);
    g[ng++] = lin(A->unk, 1, d, A->wk_w, A->wk_b, d, w.kunk, d);
    TRY(nast_gemm_f32(g, ng, s));
    ng = 0;
    for (int i = 0; i < nl; ++i) {
      const nast_declayer_weights& Wt = A->layers[i];
      g[ng++] = lin(A->memory, B * S1, d, Wt.ca_wk, Wt.ca_bk, d, w.mk + (size_t)i * B * S1 * d, d);
      g[ng++] = lin(A->memory, B * S1, d, Wt.ca_wv, Wt.ca_bv, d, w.mv + (size_t)i * B * S1 * d, d);
      if (ng == NAST_GEMM_MAX_GROUP || i == nl - 1) {
        TRY(nast_gemm_f32(g, ng, s));
        ng = 0;
      }
    }
  }
  if (fused || cluster) {
    if (fused) TRY(nast_lp_decode_fused_launch(A, w.cand, w.KC, w.kunk, w.mk, w.mv, w.kc, w.vc, early_exit ? w.done : nullptr, s));
    else TRY(nast_lp_decode_cluster_launch(A, w.cand, w.KC, w.kunk, w.mk, w.mv, w.kc, w.vc, w.x, w.q, w.a, w.x1, w.x2, w.hmid, w.qh, w.sync,
                                           early_exit ? w.done : nullptr, s));
    int steps = L;
    if (early_exit) {   // the reference's early exit, read off the finished table (one read-back, as the loop's last one)
      if (hipMemcpyAsync(&steps, w.done, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        nast_set_error("nast_lp_decode_f32: step-count readback failed: %s", hipGetErrorString(hipGetLastError()));
        return NAST_ELAUNCH;
      }
    }
    *steps_out = steps;
    return NAST_OK;
  }
  float* mean = w.stat;          // scratch for the un-fused LayerNorm path
  float* x = w.x;
  float* xn = w.xn;
  int steps = 0;
  for (int k = 0; k < L; ++k) {
    for (int i = 0; i < nl; ++i) {
      const nast_declayer_weights& Wt = A->layers[i];
      float* kc = w.kc + (size_t)i * B * L * d;
      float* vc = w.vc + (size_t)i * B * L * d;
      const float* z = x;
      auto pre_ln = [&](const float* src, const float* gw, const float* gb, nast_gemm_desc* descs, int n) -> int {
        if (fuse_ln) {
          for (int j = 0; j < n; ++j) { descs[j].ln_gamma = gw; descs[j].ln_beta = gb; }
          return 0;
        }
        // large batch / wide model: separate LayerNorm kernel into w.a (free at every call site)
        int rc = nast_layernorm_fwd_f32(src, gw, gb, w.a, w.hmid, w.hmid + B, B, d, 1e-6f, 0, 0, 0, s);
        for (int j = 0; j < n; ++j) descs[j].A[0] = w.a;
        return rc;
      };
      (void)mean; (void)z;
      // self-attention sublayer: q from the new position, K/V appended to the cache (row stride L*d)
      nast_gemm_desc qkv[3] = {lin(x, B, d, Wt.sa_wq, Wt.sa_bq, d, w.q, d),
                               lin(x, B, d, Wt.sa_wk, Wt.sa_bk, d, kc + (size_t)k * d, L * d),
                               lin(x, B, d, Wt.sa_wv, Wt.sa_bv, d, vc + (size_t)k * d, L * d)};
      TRY(pre_ln(x, Wt.ln0_w, Wt.ln0_b, qkv, 3));
      TRY(nast_gemm_f32(qkv, 3, s));
      TRY(nast_mha_fwd_f32(w.q, kc, vc, nullptr, w.a, w.lse, B, h, 1, k + 1, d / h, d, d, d, d, d, (int64_t)L * d, (int64_t)L * d, d, 0,
                           0, scale, s));
      nast_gemm_desc o1 = lin(w.a, B, d, Wt.sa_wo, Wt.sa_bo, d, w.x1, d);
      o1.residual = x;
      TRY(nast_gemm_f32(&o1, 1, s));
      // cross-attention sublayer over the pre-projected memory
      nast_gemm_desc cq = lin(w.x1, B, d, Wt.ca_wq, Wt.ca_bq, d, w.q, d);
      TRY(pre_ln(w.x1, Wt.ln1_w, Wt.ln1_b, &cq, 1));
      TRY(nast_gemm_f32(&cq, 1, s));
      TRY(nast_mha_fwd_f32(w.q, w.mk + (size_t)i * B * S1 * d, w.mv + (size_t)i * B * S1 * d, w.mask, w.a, w.lse, B, h, 1, S1, d / h, d,
                           d, d, d, d, (int64_t)S1 * d, (int64_t)S1 * d, d, S1, 0, scale, s));
      nast_gemm_desc o2 = lin(w.a, B, d, Wt.ca_wo, Wt.ca_bo, d, w.x2, d);
      o2.residual = w.x1;
      TRY(nast_gemm_f32(&o2, 1, s));
      // feed-forward sublayer
      nast_gemm_desc f1 = lin(w.x2, B, d, Wt.ff_w1, Wt.ff_b1, dff, w.hmid, dff);
      f1.flags = NAST_GEMM_RELU;
      if (fuse_ln) { f1.ln_gamma = Wt.ln2_w; f1.ln_beta = Wt.ln2_b; }
      else {
        TRY(nast_layernorm_fwd_f32(w.x2, Wt.ln2_w, Wt.ln2_b, w.a, w.q, w.q + B, B, d, 1e-6f, 0, 0, 0, s));
        f1.A[0] = w.a;
      }
      TRY(nast_gemm_f32(&f1, 1, s));
      nast_gemm_desc f2 = lin(w.hmid, B, dff, Wt.ff_w2, Wt.ff_b2, d, x, d);   // next layer's input overwrites x
      f2.residual = w.x2;
      TRY(nast_gemm_f32(&f2, 1, s));
    }
    // skip decision: qh = wq(LN(x)); the decision kernel applies the same final LN to x for linear(hidden)
    nast_gemm_desc gq = lin(x, B, d, A->wq_w, A->wq_b, d, w.qh, d);
    if (fuse_ln) { gq.ln_gamma = A->norm_w; gq.ln_beta = A->norm_b; }
    else {
      TRY(nast_layernorm_fwd_f32(x, A->norm_w, A->norm_b, w.a, w.q, w.q + B, B, d, 1e-6f, 0, 0, 0, s));
      gq.A[0] = w.a;
    }
    TRY(nast_gemm_f32(&gq, 1, s));
    TRY(nast_lp_step_f32(w.qh, x, w.KC, w.kunk, A->lin_w, A->lin_b, w.cand, A->unk, A->pos_w, A->origin_length, w.now_pos, A->allpos, xn,
                         A->norm_w, A->norm_b, A->sample_u ? A->sample_u + (int64_t)k * B : nullptr, B, S1, d, k, L + 1,
                         k + 1 < L ? 1 : 0, s));
    float* t = x; x = xn; xn = t;
    steps = k + 1;
    if (early_exit && k % 5 == 0) {   // transformer.py:569: stop once every sentence has consumed its source
      hipLaunchKernelGGL(lp_count_done_kernel, dim3(1), dim3(64), 0, s, w.now_pos, A->origin_length, w.done, B);
      int done = 0;
      if (hipMemcpyAsync(&done, w.done, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        nast_set_error("nast_lp_decode_f32: early-exit readback failed: %s", hipGetErrorString(hipGetLastError()));
        return NAST_ELAUNCH;
      }
      if (done == B) break;
    }
  }
  *steps_out = steps;
  NAST_CHECK_LAUNCH("nast_lp_decode_f32");
  return NAST_OK;
}

extern "C" int nast_lp_decode_f32(const nast_lp_decode_args* A, int* steps_out, void* stream_) {
  NAST_REQUIRE(A, "nast_lp_decode_f32: NULL args");
  return lp_decode_impl(A, A->workspace, steps_out, (hipStream_t)stream_);
}

// ---- allocation-free variant ------------------------------------------------------------------------------------------
// For callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the decode's
// workspace (candidate tables, cross-attention K/V of every layer, the self-attention K/V caches, per-step activations) lives
// in a file-scope device array instead of caller memory.  Each GPU the library runs on holds its own copy of the array
// at its own address until the process ends; the address is looked up once per device with hipGetSymbolAddress and kept.
#define NAST_LP_DECODE_STATIC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[NAST_LP_DECODE_STATIC_BYTES];
static std::atomic<void*> g_scratch_ptr[NAST_LP_DECODE_STATIC_MAX_DEVICES];      // per device; nullptr = not looked up yet (two threads may both look it up: same value)

extern "C" int nast_lp_decode_static_f32(const nast_lp_decode_args* A, int* steps_out, void* stream_) {
  NAST_REQUIRE(A && steps_out, "nast_lp_decode_static_f32: NULL args");
  NAST_REQUIRE(A->B >= 1 && A->S1 >= 1 && A->d >= 1 && A->dff >= 1 && A->L >= 1 && A->nlayers >= 1 && A->nlayers <= 8,
               "nast_lp_decode_static_f32: bad sizes");
  const int64_t need = nast_lp_decode_workspace_bytes(A->B, A->S1, A->d, A->dff, A->L, A->nlayers);
  NAST_REQUIRE(need <= (int64_t)NAST_LP_DECODE_STATIC_BYTES,
               "nast_lp_decode_static_f32: this decode needs %lld bytes of workspace, the built-in buffer holds %lld "
               "(B=%d S1=%d d=%d L=%d layers=%d): use nast_lp_decode_f32 with a caller workspace",
               (long long)need, (long long)NAST_LP_DECODE_STATIC_BYTES, A->B, A->S1, A->d, A->L, A->nlayers);
  if (*steps_out >= 0) {     // the early-exit read-backs synchronise the stream: not allowed while it is being captured
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream_, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      nast_set_error("nast_lp_decode_static_f32: the stream is being captured; pass *steps_out < 0 (no read-back of the progress)");
      return NAST_EINVAL;
    }
  }
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NAST_LP_DECODE_STATIC_MAX_DEVICES) {
    nast_set_error("nast_lp_decode_static_f32: hipGetDevice failed or device index %d out of range", dev);
    return NAST_ELAUNCH;
  }
  void* sp = g_scratch_ptr[dev].load(std::memory_order_relaxed);
  if (sp == nullptr) {      // first call on this device (which is current): its copy of the array
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ptr == nullptr) {
      nast_set_error("nast_lp_decode_static_f32: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
      return NAST_ELAUNCH;
    }
    g_scratch_ptr[dev].store(ptr, std::memory_order_relaxed);
    sp = ptr;
  }
  return lp_decode_impl(A, reinterpret_cast<float*>(sp), steps_out, (hipStream_t)stream_);
}
