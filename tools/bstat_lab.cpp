// In-kernel time stamps of the weight-stationary GEMM (debugging aid, round 2):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DBSTAT_STAMPS -Iinclude tools/bstat_lab.cpp nast_amd/csrc/core.o -o tools/bstat_lab
#include "../nast_amd/csrc/gemm_bstat.hip"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 9728, N = argc > 2 ? atoi(argv[2]) : 256, K = 256;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)M * K * 4 * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
  hipMemset(A, 0, (size_t)M * K * 16); hipMemset(B, 0, (size_t)N * K * 4);
  BStatLaunch P;
  memset(&P, 0, sizeof(P));
  P.count = 1;
  P.p[0] = {A, B, C, nullptr, nullptr, nullptr, M, N, K, K, N, 1, 0, 1.f};
  const int cgs = N / 128;
  P.cg_start[0] = 0;
  for (int i = 1; i <= NAST_GEMM_MAX_GROUP; ++i) P.cg_start[i] = cgs;
  P.J = std::min(512 / cgs, (M + 15) / 16);
  if (argc > 3) P.J = atoi(argv[3]);
  const int wgs = cgs * P.J;
  hipMalloc(&P.stamps, (size_t)wgs * 16 * 8);
  for (int it = 0; it < 4; ++it) {
    P.p[0].A = A + (size_t)(it % 4) * M * K;       // rotate: cold activations
    hipMemset(P.stamps, 0, (size_t)wgs * 16 * 8);
    hipLaunchKernelGGL((nast_gemm_bstat_kernel<false, false>), dim3(wgs), dim3(256), 0, 0, P);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h((size_t)wgs * 16);
  hipMemcpy(h.data(), P.stamps, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int w = 0; w < wgs; ++w) { t0 = std::min(t0, h[w * 16]); for (int k = 0; k < 16; ++k) t1 = std::max(t1, h[w * 16 + k]); }
  printf("M=%d N=%d J=%d wgs=%d: first start -> last stamp %llu ticks\n", M, N, P.J, wgs, t1 - t0);
  // medians over workgroups of each interval
  const char* names[] = {"start(after first wg)", "entry->B loaded", "wait+barrier 0", "tile0 loop", "tile0 wait+barrier", "tile0 epilogue", "tile1 loop",
                         "tile1 wait+barrier", "tile1 epilogue", "tile2 loop", "tile2 wait+barrier", "tile2 epilogue", "tile3 loop", "tile3 w+b", "tile3 epi", "tile4 loop"};
  for (int k = 0; k < 16; ++k) {
    std::vector<long long> v;
    for (int w = 0; w < wgs; ++w) {
      if (k == 0) v.push_back((long long)(h[w * 16] - t0));
      else if (h[w * 16 + k] && h[w * 16 + k - 1]) v.push_back((long long)(h[w * 16 + k] - h[w * 16 + k - 1]));
    }
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("  %-24s n=%4zu  min %7lld  median %7lld  p90 %7lld  max %7lld\n", names[k], v.size(), v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  return 0;
}
