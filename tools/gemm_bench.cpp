// Kernel-only timing of nast_gemm_f32 on the shapes/groupings the training step issues (run on the GPU box):
//   hipcc -O2 --offload-arch=gfx950 tools/gemm_bench.cpp -Iinclude -Lnast_amd/lib -lnast_hip -Wl,-rpath,'$ORIGIN/../nast_amd/lib' -o tools/gemm_bench
// Launches are enqueued back to back from C++ between two HIP events, so the figure is GPU time per launch
// (host launch cost ~3 us is hidden for anything longer than that).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "nast_hip.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

static float* dalloc(size_t n, bool rnd = true) {
  float* p;
  CK(hipMalloc(&p, n * sizeof(float)));
  if (rnd) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((rand() & 0xffff) / 32768.0 - 1.0);
    CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  } else {
    CK(hipMemset(p, 0, n * sizeof(float)));
  }
  return p;
}

static nast_gemm_desc mk(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int tA, int tB,
                         int flags = 0) {
  nast_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A[0] = A; d.B[0] = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
  d.transA = tA; d.transB = tB; d.nsrc = 1; d.flags = flags; d.alpha = 1.f;
  return d;
}

static const char* g_filter = nullptr;
static const float *g_x = nullptr, *g_y2 = nullptr;
static float* g_y = nullptr;
static size_t g_rot_stride = 0;
static void run(const char* name, std::vector<nast_gemm_desc> g, int iters = 20) {
  if (g_filter && !strstr(name, g_filter)) return;
  if (getenv("GB_BF16")) for (auto& d : g) d.flags |= NAST_GEMM_BF16;
  double fl = 0;
  for (auto& d : g) fl += 2.0 * d.M * d.N * (double)d.K * d.nsrc;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i)
    if (nast_gemm_f32(g.data(), (int)g.size(), nullptr) != 0) { printf("%s: ERROR %s\n", name, nast_last_error()); return; }
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, nullptr));
  // GB_ROT=n: rotate the A operands / outputs through n copies spaced `g_rot_stride` floats apart, so that no launch
  // finds its activations in L2 (what a training step looks like: every GEMM reads tensors written by other kernels)
  static const int rot = getenv("GB_ROT") ? atoi(getenv("GB_ROT")) : 1;
  for (int i = 0; i < iters; ++i) {
    if (rot > 1) {
      std::vector<nast_gemm_desc> h = g;
      const size_t off = (size_t)(i % rot) * g_rot_stride;
      for (auto& d : h) {
        if (!d.transA && d.A[0] == g_x) d.A[0] += off;
        if (d.residual == g_y2) d.residual += off;
        if (d.C == g_y) d.C += off;
      }
      nast_gemm_f32(h.data(), (int)h.size(), nullptr);
    } else {
      nast_gemm_f32(g.data(), (int)g.size(), nullptr);
    }
  }
  CK(hipEventRecord(b, nullptr));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / iters;
  printf("%-52s %9.1f us %7.1f TF\n", name, us, fl / us / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int d = 256, ff = 1024, V = 9600;
  if (argc > 1) g_filter = argv[1];   // substring filter on the case name
  std::vector<int> Ms_list = {2112, 16896};
  if (getenv("GB_M")) {   // GB_M=16384,16896: other row counts (tile-quantisation experiments)
    Ms_list.clear();
    for (char* t = strtok(strdup(getenv("GB_M")), ","); t; t = strtok(nullptr, ",")) Ms_list.push_back(atoi(t));
  }
  for (int M : Ms_list) {
    const int Mv = M == 2112 ? 2048 : 16384;
    const int rotn = getenv("GB_ROT") ? atoi(getenv("GB_ROT")) : 1;
    float* x = dalloc((size_t)M * ff * rotn);
    float* y = dalloc((size_t)M * ff * rotn, false);
    float* y2 = dalloc((size_t)M * ff * rotn, false);
    g_x = x; g_y = y; g_y2 = y2; g_rot_stride = (size_t)M * ff;
    float* y3 = dalloc((size_t)M * ff, false);
    float* W = dalloc((size_t)ff * d);
    float* W2 = dalloc((size_t)ff * d);
    float* W3 = dalloc((size_t)ff * d);
    float* dW = dalloc((size_t)ff * d, false);
    float* dW2 = dalloc((size_t)ff * d, false);
    float* dW3 = dalloc((size_t)ff * d, false);
    float* big = dalloc((size_t)Mv * V);
    float* E = dalloc((size_t)V * d);
    float* dE = dalloc((size_t)V * d, false);
    char nm[128];
    auto N = [&](const char* s) { snprintf(nm, sizeof(nm), "M=%d %s", M, s); return nm; };
    for (int Ms : {64, 512}) {    // greedy-decode shapes (small-M path), with and without the fused LayerNorm
      char nm2[128];
      nast_gemm_desc q = mk(x, W, y, Ms, d, d, d, d, d, 0, 1);
      q.bias = W3;
      snprintf(nm2, sizeof(nm2), "M=%d decode Ms=%d d->d", M, Ms); run(nm2, {q}, 50);
      nast_gemm_desc q2 = q; q2.ln_gamma = W2; q2.ln_beta = W3;
      snprintf(nm2, sizeof(nm2), "M=%d decode Ms=%d LN+d->d", M, Ms); run(nm2, {q2}, 50);
      nast_gemm_desc k2 = q2, v2 = q2; k2.C = y2; v2.C = y3;
      snprintf(nm2, sizeof(nm2), "M=%d decode Ms=%d LN+QKV x3", M, Ms); run(nm2, {q2, k2, v2}, 50);
      nast_gemm_desc f1 = mk(x, W, y, Ms, ff, d, d, d, ff, 0, 1); f1.ln_gamma = W2; f1.ln_beta = W3; f1.flags = NAST_GEMM_RELU;
      snprintf(nm2, sizeof(nm2), "M=%d decode Ms=%d LN+d->4d relu", M, Ms); run(nm2, {f1}, 50);
      nast_gemm_desc f2 = mk(x, W, y, Ms, d, ff, ff, ff, d, 0, 1); f2.residual = y2;
      snprintf(nm2, sizeof(nm2), "M=%d decode Ms=%d 4d->d + res", M, Ms); run(nm2, {f2}, 50);
    }
    // forward: y = x W^T
    run(N("fwd  d->d"), {mk(x, W, y, M, d, d, d, d, d, 0, 1)});
    {
      nast_gemm_desc o = mk(x, W, y, M, d, d, d, d, d, 0, 1);
      o.bias = W2; o.residual = y2;
      run(N("fwd  out-proj d->d + bias + residual"), {o});
      nast_gemm_desc dg = mk(x, W, y, M, d, d, d, d, d, 0, 0);
      nast_gemm_desc wg = mk(x, y2, dW, d, d, M, d, d, d, 1, 0, NAST_GEMM_ATOMIC);
      wg.colsum = dW2;
      run(N("bwd  out-proj: dgrad + wgrad + bias grad"), {dg, wg});
    }
    run(N("fwd  QKV grouped x3"), {mk(x, W, y, M, d, d, d, d, d, 0, 1), mk(x, W2, y2, M, d, d, d, d, d, 0, 1), mk(x, W3, y3, M, d, d, d, d, d, 0, 1)});
    run(N("fwd  d->4d"), {mk(x, W, y, M, ff, d, d, d, ff, 0, 1)});
    run(N("fwd  4d->d"), {mk(x, W, y, M, d, ff, ff, ff, d, 0, 1)});
    // backward pairs as the sublayers issue them
    run(N("bwd  out-proj: dgrad d->d + wgrad"), {mk(x, W, y, M, d, d, d, d, d, 0, 0), mk(x, y2, dW, d, d, M, d, d, d, 1, 0, NAST_GEMM_ATOMIC)});
    {
      nast_gemm_desc dg = mk(x, W, y, M, d, d, d, d, d, 0, 0);
      dg.A[1] = y2; dg.A[2] = y3; dg.B[1] = W2; dg.B[2] = W3; dg.nsrc = 3;
      run(N("bwd  qkv: 3 wgrad + dgrad(nsrc=3)"), {mk(x, y, dW, d, d, M, d, d, d, 1, 0, NAST_GEMM_ATOMIC), mk(y2, y, dW2, d, d, M, d, d, d, 1, 0, NAST_GEMM_ATOMIC),
                                                  mk(y3, y, dW3, d, d, M, d, d, d, 1, 0, NAST_GEMM_ATOMIC), dg});
    }
    run(N("bwd  ffn2: dgrad (M,4d)<-d + wgrad (d,4d)"), {mk(x, W, y, M, ff, d, d, ff, ff, 0, 0), mk(x, y2, dW, d, ff, M, d, ff, ff, 1, 0, NAST_GEMM_ATOMIC)});
    run(N("bwd  ffn1: dgrad (M,d)<-4d + wgrad (4d,d)"), {mk(x, W, y, M, d, ff, ff, d, d, 0, 0), mk(x, y2, dW, ff, d, M, ff, d, d, 1, 0, NAST_GEMM_ATOMIC)});
    run(N("wgrad alone (d,d) K=M"), {mk(x, y2, dW, d, d, M, d, d, d, 1, 0, NAST_GEMM_ATOMIC)});
    run(N("wgrad alone (4d,d) K=M"), {mk(x, y2, dW, ff, d, M, ff, d, d, 1, 0, NAST_GEMM_ATOMIC)});
    // vocabulary-sized problems
    snprintf(nm, sizeof(nm), "Mv=%d generator fwd d->V", Mv);
    run(nm, {mk(x, E, big, Mv, V, d, d, d, V, 0, 1)}, 10);
    snprintf(nm, sizeof(nm), "Mv=%d generator bwd: dgrad K=V + wgrad (V,d)", Mv);
    run(nm, {mk(big, E, y, Mv, d, V, V, d, d, 0, 0), mk(big, x, dE, V, d, Mv, V, d, d, 1, 0, NAST_GEMM_ATOMIC)}, 10);
    snprintf(nm, sizeof(nm), "Mv=%d soft-embed fwd NN K=V", Mv);
    run(nm, {mk(big, E, y, Mv, d, V, V, d, d, 0, 0)}, 10);
    snprintf(nm, sizeof(nm), "Mv=%d soft-embed bwd: dX=dY E^T + dE=X^T dY", Mv);
    run(nm, {mk(x, E, big, Mv, V, d, d, d, V, 0, 1), mk(big, x, dE, V, d, Mv, V, d, d, 1, 0, NAST_GEMM_ATOMIC)}, 10);
    for (float* p : {x, y, y2, y3, W, W2, W3, dW, dW2, dW3, big, E, dE}) CK(hipFree(p));
  }
  return 0;
}
