// Micro-benchmark: what does a workgroup structure cost on the fp32 matrix pipe, with no memory traffic at all?
// 256-thread workgroups, each wave issues `ktiles` x 16 MFMAs (v_mfma_f32_32x32x2_f32) on NACC accumulators, with or without a
// workgroup barrier after every 16.  Run: hipcc -O3 --offload-arch=gfx950 tools/mfma_lab.hip -o tools/mfma_lab && tools/mfma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BAR, int LDSKB>
__global__ __launch_bounds__(256) void k(float* out, int ktiles, float a0, float b0) {
  __shared__ float pad[LDSKB * 256];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int t = 0; t < ktiles; ++t) {
#pragma unroll
    for (int s = 0; s < 16 / NACC; ++s)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    if (BAR) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.f) out[threadIdx.x] = s + pad[threadIdx.x];
}

template <int NACC, bool BAR, int LDSKB>
void run(const char* name, int wgs, int ktiles) {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<NACC, BAR, LDSKB>), dim3(wgs), dim3(256), 0, 0, out, ktiles, 1.f, 2.f);
  hipEventRecord(e0);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<NACC, BAR, LDSKB>), dim3(wgs), dim3(256), 0, 0, out, ktiles, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)wgs * 4 * ktiles * 16 * 4096.0 * it;
  printf("%-44s wgs=%5d ktiles=%4d  %8.1f us/launch  %6.1f TF\n", name, wgs, ktiles, ms * 1e3 / it, fl / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  // LDS per workgroup selects the residency: 36 KB -> 4 per CU, 72 KB -> 2, 150 KB -> 1
  for (int kt : {8, 32, 256}) {
    run<1, false, 36>("1 acc, no barrier, 4 WG/CU", 1024 * 4, kt);
    run<1, true, 36>("1 acc, barrier/16 MFMA, 4 WG/CU", 1024 * 4, kt);
    run<4, false, 36>("4 acc, no barrier, 4 WG/CU", 1024 * 4, kt);
    run<4, true, 36>("4 acc, barrier/16 MFMA, 4 WG/CU", 1024 * 4, kt);
    run<1, true, 72>("1 acc, barrier, 2 WG/CU", 512 * 8, kt);
    run<4, true, 72>("4 acc, barrier, 2 WG/CU", 512 * 8, kt);
    run<1, true, 150>("1 acc, barrier, 1 WG/CU", 256 * 16, kt);
    run<4, true, 150>("4 acc, barrier, 1 WG/CU", 256 * 16, kt);
    run<4, false, 150>("4 acc, no barrier, 1 WG/CU", 256 * 16, kt);
  }
  // the launch shapes of the step: 2432 workgroups of 8 K-tiles (d->4d), 608 (d->d), 1824 (QKV)
  run<1, true, 36>("d->4d shape: 2432 WGs x 8 ktiles", 2432, 8);
  run<1, true, 36>("QKV   shape: 1824 WGs x 8 ktiles", 1824, 8);
  run<1, true, 36>("d->d  shape:  608 WGs x 8 ktiles", 608, 8);
  run<1, true, 36>("4d->d shape:  608 WGs x 32 ktiles", 608, 32);
  return 0;
}
