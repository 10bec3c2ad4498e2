// Index maps of the packed-row ("varlen") layout in ONE launch (nast_amd.ops.PackedSeq): exclusive scan of the sentence
// lengths plus, for every packed row, its sentence, its position and its row in the padded [B * width] layout.
// The torch formulation (cumsum, repeat_interleave, arange, gathers, cat) is ~18 launches per layout and a training step
// builds a dozen layouts.  Integer work on a few thousand elements, but ~32 bytes of maps per packed row: one workgroup
// writing them all is bound by its CU's store path (18 us at 512 sentences / 10k rows), so PACK_WGS workgroups each repeat
// the scan (wave shuffles, one barrier) and write the rows of every PACK_WGS-th group of sentences.
#include "common.h"

#define PACK_MAX_B 8192
#define PACK_WGS 32

__global__ __launch_bounds__(256) void nast_pack_index_kernel(const int64_t* __restrict__ len, int B, int width, int total, int rows,
                                                              int64_t* __restrict__ off, int32_t* __restrict__ off32,
                                                              int32_t* __restrict__ len32, int64_t* __restrict__ b_of,
                                                              int64_t* __restrict__ pos, int64_t* __restrict__ idx,
                                                              int64_t* __restrict__ idx_p, int* __restrict__ bad) {
  __shared__ int s_off[PACK_MAX_B + 1];
  __shared__ int s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (B + 255) / 256;
  const int lo = tid * per < B ? tid * per : B, hi = lo + per < B ? lo + per : B;
  int sum = 0;
  for (int i = lo; i < hi; ++i) {
    const int l = (int)len[i];
    s_off[i] = sum;                              // exclusive within the thread's chunk
    sum += l > 0 ? l : 0;
  }
  int inc = sum;                                 // inclusive scan of the chunk sums over the wavefront ...
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int base = inc - sum;                          // ... and over the 4 wavefronts
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  const int all = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  for (int i = lo; i < hi; ++i) {
    const int o = s_off[i] + base;
    s_off[i] = o;
    if (blockIdx.x == 0) {
      off[i] = o;
      off32[i] = o;
      len32[i] = (int)len[i];
    }
  }
  if (tid == 0) {
    s_off[B] = all;
    if (blockIdx.x == 0 && all != total) *bad = 1;   // sticky: the caller's total disagrees with the lengths (rows would be misplaced)
  }
  __syncthreads();
  const int gtid = blockIdx.x * 256 + tid, gthreads = gridDim.x * 256;
  if (all != total) {                            // (uniform) the maps are zero-filled: every index stays inside the caller's buffers
    for (int r = gtid; r < total; r += gthreads) b_of[r] = pos[r] = idx[r] = 0;
    for (int r = gtid; r < rows; r += gthreads) idx_p[r] = 0;
    return;
  }
  for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
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
  for (int r = total + gtid; r < rows; r += gthreads) idx_p[r] = 0;
}

extern "C" int nast_pack_index_i64(const int64_t* lengths, int B, int width, int total, int rows, int64_t* off, int32_t* off32,
                                   int32_t* len32, int64_t* b_of, int64_t* pos, int64_t* idx, int64_t* idx_p, int* bad, void* stream) {
  NAST_REQUIRE(B >= 1 && B <= PACK_MAX_B, "nast_pack_index_i64: B=%d outside 1..%d", B, PACK_MAX_B);
  NAST_REQUIRE(width >= 0 && total >= 0 && rows >= total, "nast_pack_index_i64: bad sizes width=%d total=%d rows=%d", width, total, rows);
  NAST_REQUIRE(lengths && off && off32 && len32 && bad, "nast_pack_index_i64: NULL pointer");
  NAST_REQUIRE(rows == 0 || (b_of && pos && idx && idx_p), "nast_pack_index_i64: NULL row map");
  hipLaunchKernelGGL(nast_pack_index_kernel, dim3(B < 4 * PACK_WGS ? (B + 3) / 4 : PACK_WGS), dim3(256), 0, (hipStream_t)stream, lengths, B, width, total, rows, off, off32,
                     len32, b_of, pos, idx, idx_p, bad);
  NAST_CHECK_LAUNCH("nast_pack_index_i64");
  return NAST_OK;
}

// ---- moving rows between the padded [B * width, d] layout and the packed one ------------------------------------
// Both directions are gathers (every output row knows its source), so neither needs the zero-fill + scatter pair
// (torch: zeros + index_copy / index_add) that the autograd formulas of index_select / index_copy produce: one launch,
// each byte written once.  One wavefront per output row, 16-byte accesses when the pitches allow.
template <bool VEC>
__global__ __launch_bounds__(256) void nast_rows_gather_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ idx,
                                                               int n_idx, int rows_out, int d, float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows_out) return;
  float* yr = y + (int64_t)r * ldy;
  if (r >= n_idx) {                              // rows past the index list: zeros
    if (VEC) for (int c = 4 * lane; c < d; c += 256) *reinterpret_cast<float4*>(yr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    else for (int c = lane; c < d; c += 64) yr[c] = 0.f;
    return;
  }
  const float* xr = x + idx[r] * ldx;
  if (VEC) for (int c = 4 * lane; c < d; c += 256) *reinterpret_cast<float4*>(yr + c) = *reinterpret_cast<const float4*>(xr + c);
  else for (int c = lane; c < d; c += 64) yr[c] = xr[c];
}

template <bool VEC>
__global__ __launch_bounds__(256) void nast_rows_unpack_kernel(const float* __restrict__ xp, int64_t ldx, const int64_t* __restrict__ off,
                                                               const int32_t* __restrict__ len, int B, int width, int d,
                                                               int rows_in, float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= (int64_t)B * width) return;
  const int b = (int)(r / width), p = (int)(r % width);
  float* yr = y + r * ldy;
  const int64_t src = off[b] + p;
  if (p >= len[b] || src >= rows_in) {           // a padded position of the [B, width] layout (or, after a length / total
                                                 // mismatch that nast_pack_index_i64 has flagged, a row the buffer does not have)
    if (VEC) for (int c = 4 * lane; c < d; c += 256) *reinterpret_cast<float4*>(yr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    else for (int c = lane; c < d; c += 64) yr[c] = 0.f;
    return;
  }
  const float* xr = xp + src * ldx;
  if (VEC) for (int c = 4 * lane; c < d; c += 256) *reinterpret_cast<float4*>(yr + c) = *reinterpret_cast<const float4*>(xr + c);
  else for (int c = lane; c < d; c += 64) yr[c] = xr[c];
}

static bool rows_vec_ok(const void* a, const void* b, int64_t lda, int64_t ldb, int d) {
  return (d & 3) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0;
}

// Key-padding mask of a ragged batch in one launch (the torch formulation is arange, compare, zeros, cat: four):
// mask[b][j] = j < extra ? 0 : (j - extra >= lengths[b]), one byte per element (torch.bool storage).
__global__ __launch_bounds__(256) void nast_pad_mask_kernel(const int64_t* __restrict__ lengths, uint8_t* __restrict__ mask, int B, int cols,
                                                           int extra) {
  const int64_t n = (int64_t)B * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / cols), j = (int)(i % cols);
    mask[i] = (j >= extra && (int64_t)(j - extra) >= lengths[b]) ? 1 : 0;
  }
}

extern "C" int nast_pad_mask_u8(const int64_t* lengths, uint8_t* mask, int B, int width, int extra, void* stream) {
  NAST_REQUIRE(B >= 0 && width >= 0 && extra >= 0, "nast_pad_mask_u8: negative sizes");
  const int64_t n = (int64_t)B * (width + extra);
  if (n == 0) return NAST_OK;
  NAST_REQUIRE(lengths && mask, "nast_pad_mask_u8: NULL pointer");
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(nast_pad_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, lengths, mask, B, width + extra, extra);
  NAST_CHECK_LAUNCH("nast_pad_mask_u8");
  return NAST_OK;
}

extern "C" int nast_rows_gather_f32(const float* x, int64_t ldx, const int64_t* idx, int n_idx, int rows_out, int d, float* y,
                                    int64_t ldy, void* stream) {
  NAST_REQUIRE(n_idx >= 0 && rows_out >= n_idx && d >= 1 && ldx >= d && ldy >= d, "nast_rows_gather_f32: bad sizes n_idx=%d rows_out=%d d=%d", n_idx, rows_out, d);
  if (rows_out == 0) return NAST_OK;
  NAST_REQUIRE(y && (n_idx == 0 || (x && idx)), "nast_rows_gather_f32: NULL pointer");
  const dim3 grid((rows_out + 3) / 4), block(256);
  if (rows_vec_ok(x, y, ldx, ldy, d)) hipLaunchKernelGGL(nast_rows_gather_kernel<true>, grid, block, 0, (hipStream_t)stream, x, ldx, idx, n_idx, rows_out, d, y, ldy);
  else hipLaunchKernelGGL(nast_rows_gather_kernel<false>, grid, block, 0, (hipStream_t)stream, x, ldx, idx, n_idx, rows_out, d, y, ldy);
  NAST_CHECK_LAUNCH("nast_rows_gather_f32");
  return NAST_OK;
}

extern "C" int nast_rows_unpack_f32(const float* xp, int64_t ldx, int rows_in, const int64_t* off, const int32_t* len, int B, int width,
                                    int d, float* y, int64_t ldy, void* stream) {
  NAST_REQUIRE(B >= 0 && width >= 0 && d >= 1 && ldx >= d && ldy >= d && rows_in >= 0, "nast_rows_unpack_f32: bad sizes B=%d width=%d d=%d", B, width, d);
  const int64_t rows = (int64_t)B * width;
  if (rows == 0) return NAST_OK;
  NAST_REQUIRE(rows <= (int64_t)4 * 0x7fffffff, "nast_rows_unpack_f32: too many rows");
  NAST_REQUIRE(xp && off && len && y, "nast_rows_unpack_f32: NULL pointer");
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (rows_vec_ok(xp, y, ldx, ldy, d)) hipLaunchKernelGGL(nast_rows_unpack_kernel<true>, grid, block, 0, (hipStream_t)stream, xp, ldx, off, len, B, width, d, rows_in, y, ldy);
  else hipLaunchKernelGGL(nast_rows_unpack_kernel<false>, grid, block, 0, (hipStream_t)stream, xp, ldx, off, len, B, width, d, rows_in, y, ldy);
  NAST_CHECK_LAUNCH("nast_rows_unpack_f32");
  return NAST_OK;
}
