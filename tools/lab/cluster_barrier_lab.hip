// Lab: what does a barrier among the 32 workgroups of one "cluster" (blockIdx % 8: one XCD under the observed round-robin
// placement -- speed only, every exchanged byte is stored and loaded sc1) cost on MI355X, with and without a payload?
// Build: hipcc -O3 --offload-arch=gfx950 tools/lab/cluster_barrier_lab.hip -o /tmp/cluster_barrier_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define SC1 16

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

struct Args {
  unsigned* counters;   // [8][32] (128-B line per cluster)
  unsigned* err;
  float* buf;           // [2][8 clusters][64 rows][256]
  float* out;
  int episodes, payload, members;
};

__device__ __forceinline__ bool cluster_barrier(unsigned* cnt, unsigned target, unsigned* err) {
  __syncthreads();
  __shared__ int ok_s;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 4000000u) { ok = 0; atomicExch(err, 1u); break; }
    }
    ok_s = ok;
  }
  __syncthreads();
  return ok_s != 0;
}

__global__ __launch_bounds__(256) void lab_kernel(Args A) {
  const int c = blockIdx.x & 7, m = blockIdx.x >> 3;
  unsigned* cnt = A.counters + c * 32;
  float acc = 0.f;
  const size_t slab = (size_t)8 * 64 * 256;
  for (int e = 0; e < A.episodes; ++e) {
    float* wbuf = A.buf + (size_t)(e & 1) * slab + (size_t)c * 64 * 256;
    if (A.payload) {
      // this member writes a 64 x 8 column block (2 KB): rows r, columns 8m..8m+7: two 16-B sc1 stores per row by 128 lanes
      if (threadIdx.x < 128) {
        const int r = threadIdx.x >> 1, h = threadIdx.x & 1;
        u32x4 v;
        const float f = (float)(e + 1) + 0.001f * m;
        v.x = v.y = v.z = v.w = __float_as_uint(f);
        __builtin_amdgcn_raw_buffer_store_b128(v, make_rsrc(wbuf), (r * 256 + 8 * m + 4 * h) * 4, 0, SC1);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!cluster_barrier(cnt, (unsigned)(e + 1) * A.members, A.err)) return;
    if (A.payload) {
      // read a 32-row x 256 tile (32 KB): 8 x 16 B per thread, sc1
      const int rt = m & 1;
      __amdgpu_buffer_rsrc_t rs = make_rsrc(wbuf);
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = threadIdx.x + 256 * j;          // float4 index in the 32 x 64 float4 tile
        const int r = rt * 32 + (idx >> 6), c4 = idx & 63;
        v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (r * 256 + c4 * 4) * 4, 0, SC1);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = __uint_as_float(v[j].x);
        // check: every value must carry this episode's tag
        if ((((threadIdx.x + 256 * j) & 63) * 4 < 8 * A.members) && (int)f != e + 1) atomicExch(A.err, 2u);
        acc += f + __uint_as_float(v[j].y) + __uint_as_float(v[j].z) + __uint_as_float(v[j].w);
      }
    }
  }
  if (acc == -1.f) A.out[0] = acc;
}

int main(int argc, char** argv) {
  const int episodes = argc > 1 ? atoi(argv[1]) : 1000;
  Args A;
  hipMalloc(&A.counters, 8 * 32 * 4);
  hipMalloc(&A.err, 4);
  hipMalloc(&A.buf, sizeof(float) * 2 * 8 * 64 * 256);
  hipMalloc(&A.out, 4);
  hipEvent_t t0, t1;
  hipEventCreate(&t0); hipEventCreate(&t1);
  for (int members : {32, 16}) {
    for (int payload = 0; payload < 2; ++payload) {
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(A.counters, 0, 8 * 32 * 4);
        hipMemset(A.err, 0, 4);
        A.episodes = episodes; A.payload = payload; A.members = members;
        hipEventRecord(t0);
        hipLaunchKernelGGL(lab_kernel, dim3(8 * members), dim3(256), 0, 0, A);
        hipEventRecord(t1);
        hipEventSynchronize(t1);
        float ms = 0;
        hipEventElapsedTime(&ms, t0, t1);
        unsigned err = 0;
        hipMemcpy(&err, A.err, 4, hipMemcpyDeviceToHost);
        printf("members %d payload %d: %.3f us per episode (%d episodes) err=%u\n", members, payload, ms * 1000.f / episodes, episodes, err);
      }
    }
  }
  return 0;
}
