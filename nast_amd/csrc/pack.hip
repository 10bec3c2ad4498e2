// Index maps of the packed-row ("varlen") layout in ONE launch (nast_amd.ops.PackedSeq): exclusive scan of the sentence
// lengths plus, for every packed row, its sentence, its position and its row in the padded [B * width] layout.
// The torch formulation (cumsum, repeat_interleave, arange, gathers, cat) is ~18 launches per layout and a training step
// builds a dozen layouts.  Integer work on a few thousand elements: one 1024-thread workgroup, scan through LDS, then one
// wavefront per sentence writes that sentence's rows (consecutive addresses).
#include "common.h"

#define PACK_MAX_B 8192

__global__ __launch_bounds__(1024) void nast_pack_index_kernel(const int64_t* __restrict__ len, int B, int width, int total, int rows,
                                                               int64_t* __restrict__ off, int32_t* __restrict__ off32,
                                                               int32_t* __restrict__ len32, int64_t* __restrict__ b_of,
                                                               int64_t* __restrict__ pos, int64_t* __restrict__ idx,
                                                               int64_t* __restrict__ idx_p, int* __restrict__ bad) {
  __shared__ int s_off[PACK_MAX_B + 1];
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  const int per = (B + 1023) / 1024;
  const int lo = tid * per, hi = lo + per < B ? lo + per : B;
  int sum = 0;
  for (int i = lo; i < hi; ++i) {
    const int l = (int)len[i];
    s_off[i] = sum;                              // exclusive within the chunk
    sum += l > 0 ? l : 0;
  }
  s_part[tid] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan of the 1024 chunk sums
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? s_part[tid - o] : 0;
    __syncthreads();
    s_part[tid] += v;
    __syncthreads();
  }
  const int base = tid > 0 ? s_part[tid - 1] : 0;
  for (int i = lo; i < hi; ++i) {
    const int o = s_off[i] + base;
    s_off[i] = o;
    const int l = (int)len[i];
    off[i] = o;
    off32[i] = o;
    len32[i] = l;
  }
  if (tid == 1023) {
    s_off[B] = s_part[1023];
    if (s_part[1023] != total) *bad = 1;         // sticky: the caller's total disagrees with the lengths (rows would be misplaced)
  }
  __syncthreads();
  if (s_off[B] != total) {                       // (uniform) the maps are zero-filled: every index stays inside the caller's buffers
    for (int r = tid; r < total; r += 1024) b_of[r] = pos[r] = idx[r] = 0;
    for (int r = tid; r < rows; r += 1024) idx_p[r] = 0;
    return;
  }
  const int lane = tid & 63, wave = tid >> 6;
  for (int b = wave; b < B; b += 16) {
    const int o = s_off[b], l = s_off[b + 1] - o;
    for (int p = lane; p < l; p += 64) {
      const int r = o + p;
      const int64_t v = (int64_t)b * width + p;
      b_of[r] = b;
      pos[r] = p;
      idx[r] = v;
      idx_p[r] = v;
    }
  }
  for (int r = total + tid; r < rows; r += 1024) idx_p[r] = 0;
}

extern "C" int nast_pack_index_i64(const int64_t* lengths, int B, int width, int total, int rows, int64_t* off, int32_t* off32,
                                   int32_t* len32, int64_t* b_of, int64_t* pos, int64_t* idx, int64_t* idx_p, int* bad, void* stream) {
  NAST_REQUIRE(B >= 1 && B <= PACK_MAX_B, "nast_pack_index_i64: B=%d outside 1..%d", B, PACK_MAX_B);
  NAST_REQUIRE(width >= 0 && total >= 0 && rows >= total, "nast_pack_index_i64: bad sizes width=%d total=%d rows=%d", width, total, rows);
  NAST_REQUIRE(lengths && off && off32 && len32 && bad, "nast_pack_index_i64: NULL pointer");
  NAST_REQUIRE(rows == 0 || (b_of && pos && idx && idx_p), "nast_pack_index_i64: NULL row map");
  hipLaunchKernelGGL(nast_pack_index_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lengths, B, width, total, rows, off, off32,
                     len32, b_of, pos, idx, idx_p, bad);
  NAST_CHECK_LAUNCH("nast_pack_index_i64");
  return NAST_OK;
}
