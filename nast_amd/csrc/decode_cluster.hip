// Skip decoding of the LengthPredictor (transformer.py:525-570) as ONE persistent launch of 8 clusters x 32 workgroups.
//
// Why: the step-by-step path (decode.hip) is ~26 dependent launches per decode step, 830 per decode, each ~9-12 us of which
// most is the cold start of a kernel (first operand round trip 3.5 us); the one-workgroup-per-sentence-
// block kernel (decode_fused.hip) needs no synchronisation at all but streams all 11 MB of decoder weights through ONE CU per
// step (27 GB/s: 13 ms floor).  This kernel sits between the two:
//   * sentences are dealt to 8 CLUSTERS (cluster = blockIdx % 8: the 32 workgroups of a cluster share an XCD under the
//     round-robin placement the hardware shows -- a speed matter only, nothing below depends on it); every operation of the
//     decode is per sentence, so clusters never talk to each other;
//   * inside a cluster each operation (LayerNorm + projection GEMM, single-query attention, skip decision) is spread over the
//     cluster's 32 workgroups as [32 rows x 16 columns] MFMA tiles (v_mfma_f32_16x16x4_f32, K split over the 4 wavefronts), so
//     a CU reads 1/16 .. 1/64 of a weight matrix per operation (weight-stationary across CUs) and the next operation's weight
//     slice is requested BEFORE the workgroup waits at the barrier: its latency hides behind the wait;
//   * operations are separated by a cluster barrier: one agent-scope atomic add per workgroup on the cluster's counter, polled
//     with sc1 loads (tools/lab/cluster_barrier_lab.hip: 0.9 us for 32 workgroups, 2.05 us with a 2-KB write + 32-KB read of
//     handed-over data per workgroup).  Every byte one workgroup writes and another reads inside the launch -- activations,
//     K/V cache rows -- is STORED sc1 (write-through) by 16-byte stores, drained (s_waitcnt vmcnt(0) by every storing wave,
//     then the workgroup barrier) before the arrival add, and LOADED sc1 (buffer loads to registers, bypassing the CU's L1)
//     after the poll and a workgroup barrier: MI355X_MICROARCH.md, visibility table row 1.  Weights and the tables that
//     earlier kernels wrote (candidates, wk(candidates), cross-attention K/V) are read by plain loads.
// All 256 workgroups must be resident for the barriers to complete: 49 KB of LDS and 256 threads per workgroup admit 3 per CU,
// the host checks the CU count, and every poll is bounded (1 s): a timeout raises a sticky device flag that
// nast_lp_decode_cluster_status() reports, the workgroup leaves, and the others time out the same way -- the grid always drains.
//
// Limits (nast_lp_decode_cluster_eligible): d_model 256, 4 heads, d_ff 1024, <= 4 layers, max_length and memory width <= 36,
// <= 2048 sentences.  Anything else takes the step-by-step path.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int CD = 256, CFF = 1024, LDX = CD + 4, MAXKEYS = 36;
constexpr int TG = 2;                                    // column tiles of a workgroup that share one pass over the activation image
constexpr int KB = 12;                                   // keys per attention block
constexpr int NCLUSTER = 8, NMEMBER = 32, MAXSPW = 8;   // sentences per wavefront of a cluster's 128 wavefronts
constexpr int SC1 = 16;                                  // aux bit of the raw buffer builtins: sc1
constexpr int RLD = 20;                                  // row stride of the partial-sum patch [4 waves][32 rows][16 + 4]
constexpr int MAXOPS = 8 * 4 + 2;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef CLUSTER_STAMPS   // debug build only (tools/cluster_stamps.sh): where the time of a decode goes, summed per phase, for 2 workgroups
__device__ long long g_cluster_dbg[2][8];
#define TS(i) do { const long long tn_ = wall_clock64(); tacc[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define TS(i) do { } while (0)
#endif
__device__ unsigned g_cluster_err;       // sticky: a barrier poll timed out (some workgroup of the grid was not resident)

struct ClusterArgs {
  nast_declayer_weights layers[4];
  const float *norm_w, *norm_b, *wq_w, *wq_b, *lin_w, *lin_b, *pos_w, *unk, *memory;
  const float *cand, *KC, *kunk, *mk, *mv;
  float *kc, *vc, *x, *q, *a, *x1, *x2, *hmid, *qh;
  const int64_t* olen;
  int64_t* allpos;
  const float* sample_u;
  unsigned* sync;      // [8 clusters][32 words] arrival counters, zeroed by the launcher
  int B, S1, L, nl, Bc, nrt;
};

enum { OP_GEMM = 0, OP_ATTN = 1, OP_STEP = 2 };

struct Op {
  int type, nchunk, nseg, seg_tiles, relu, self, lda, ldw;
  const float* A;
  const float* W[3];
  const float* bias[3];
  float* out[3];
  int ldc[3], kstride[3];
  const float *ln_g, *ln_b, *resid;
  const float *Kp, *Vp;      // attention: K / V rows of sentence b start at Kp + b * sent_stride (row stride CD)
  float* qin;                // attention: queries [B][CD];  output goes to out[0]
  int sent_stride;
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T* uni(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffffc, 0x00020000);
}
__device__ __forceinline__ float4 ld16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, SC1);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// the same with the wave-uniform part of the address in a scalar register (no address VGPR per load); AUX = SC1 or 0 (plain)
template <int AUX>
__device__ __forceinline__ float4 ld16_so(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t uniform_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, AUX);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4& f) {
  u32x4 v;
  v.x = __float_as_uint(f.x); v.y = __float_as_uint(f.y); v.z = __float_as_uint(f.z); v.w = __float_as_uint(f.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, SC1);
}
// marks a register quad as (re)defined without an instruction: values that a later operation will overwrite anyway must not
// stay live through the operations in between (the attention holds 36 keys per lane in registers)
__device__ __forceinline__ void kill4(float4& v) { asm volatile("" : "=v"(v.x), "=v"(v.y), "=v"(v.z), "=v"(v.w)); }
__device__ __forceinline__ float pick4(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }
// Sums over lane groups without the LDS crossbar (ds_bpermute, what __shfl_xor compiles to): DPP butterflies inside a row of
// 16 lanes (quad_perm xor 1, xor 2; then row_half_mirror and row_mirror, which pair equal-valued quads / halves), v_readlane
// across the four rows.  Every lane of the group gets the sum.  (5.9 -> 5.5 ms per decode at 512 sentences.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group16_sum(float v) {
  v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);     // row_half_mirror
  v = dpp_add<0x140>(v);     // row_mirror
  return v;
}
__device__ __forceinline__ float wave64_sum(float v) {
  v = group16_sum(v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}

__global__ __launch_bounds__(256, 2) void nast_lp_decode_cluster_kernel(const ClusterArgs A) {
  __shared__ __attribute__((aligned(16))) float As[32 * LDX];          // activation tile [32 rows][256 k] (+4 pad)
  __shared__ __attribute__((aligned(16))) float red[4 * 32 * RLD];     // K-split partial sums of one output tile
  __shared__ Op ops[MAXOPS];
  // LayerNorm scale | shift of the current operation: used before the MFMAs, `red` after them -- one piece of LDS, which keeps
  // the workgroup under 49.4 KB, i.e. leaves room for THREE 36.9-KB workgroups of the tiled GEMM on the CU while the decode runs
  float* const ln_s = red;
  __shared__ int now_s[4][MAXSPW];
  __shared__ int abort_s;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int c = blockIdx.x & (NCLUSTER - 1), m = blockIdx.x / NCLUSTER;
  const int B = A.B, S1 = A.S1, L = A.L, nl = A.nl;
  const int NC = S1 + 5;
  const int b0c = c * A.Bc;
  int nb = B - b0c;
  nb = nb < A.Bc ? nb : A.Bc;
  if (nb <= 0) return;                                   // an empty cluster: none of its workgroups ever waits
  const int nrt = A.nrt, cpr = NMEMBER / nrt;
  const int rt = m % nrt, cs = m / nrt;
  const int r0 = b0c + 32 * rt;                          // first sentence of this workgroup's row tile
  int nvalid = b0c + nb - r0;
  nvalid = nvalid < 32 ? nvalid : 32;                    // <= 0: no GEMM work here (the workgroup still joins every barrier)
  const int nops = nl * 8 + 2;
  unsigned* cnt = A.sync + c * 32;

  // ---- operation table of one decode step ----
  if (tid < nops) {
    Op o = {};
    o.nchunk = 1; o.nseg = 1; o.seg_tiles = CD / 16; o.lda = CD; o.ldw = CD;
    if (tid == nops - 1) {
      o.type = OP_STEP;
    } else if (tid == nops - 2) {                        // qh = wq(LN(x))
      o.type = OP_GEMM; o.A = A.x; o.W[0] = A.wq_w; o.bias[0] = A.wq_b; o.out[0] = A.qh; o.ldc[0] = CD;
      o.ln_g = A.norm_w; o.ln_b = A.norm_b;
    } else {
      const int li = tid / 8, k = tid % 8;
      const nast_declayer_weights& Wt = A.layers[li];
      float* kc = A.kc + (size_t)li * B * L * CD;
      float* vc = A.vc + (size_t)li * B * L * CD;
      o.type = (k == 1 || k == 4) ? OP_ATTN : OP_GEMM;
      if (k == 0) {                                      // q | K-cache row | V-cache row = LN0(x) [Wq | Wk | Wv]^T
        o.A = A.x; o.nseg = 3; o.ln_g = Wt.ln0_w; o.ln_b = Wt.ln0_b;
        o.W[0] = Wt.sa_wq; o.bias[0] = Wt.sa_bq; o.out[0] = A.q; o.ldc[0] = CD;
        o.W[1] = Wt.sa_wk; o.bias[1] = Wt.sa_bk; o.out[1] = kc; o.ldc[1] = L * CD; o.kstride[1] = CD;
        o.W[2] = Wt.sa_wv; o.bias[2] = Wt.sa_bv; o.out[2] = vc; o.ldc[2] = L * CD; o.kstride[2] = CD;
      } else if (k == 1) {                               // self-attention over the cache rows 0..step
        o.self = 1; o.qin = A.q; o.Kp = kc; o.Vp = vc; o.sent_stride = L * CD; o.out[0] = A.a;
      } else if (k == 2) {                               // x1 = x + a Wo^T
        o.A = A.a; o.W[0] = Wt.sa_wo; o.bias[0] = Wt.sa_bo; o.out[0] = A.x1; o.ldc[0] = CD; o.resid = A.x;
      } else if (k == 3) {                               // cross-attention query
        o.A = A.x1; o.W[0] = Wt.ca_wq; o.bias[0] = Wt.ca_bq; o.out[0] = A.q; o.ldc[0] = CD; o.ln_g = Wt.ln1_w; o.ln_b = Wt.ln1_b;
      } else if (k == 4) {                               // cross-attention over the pre-projected memory
        o.self = 0; o.qin = A.q; o.Kp = A.mk + (size_t)li * B * S1 * CD; o.Vp = A.mv + (size_t)li * B * S1 * CD;
        o.sent_stride = S1 * CD; o.out[0] = A.a;
      } else if (k == 5) {                               // x2 = x1 + a Wo^T
        o.A = A.a; o.W[0] = Wt.ca_wo; o.bias[0] = Wt.ca_bo; o.out[0] = A.x2; o.ldc[0] = CD; o.resid = A.x1;
      } else if (k == 6) {                               // hidden = relu(LN2(x2) W1^T)
        o.A = A.x2; o.W[0] = Wt.ff_w1; o.bias[0] = Wt.ff_b1; o.out[0] = A.hmid; o.ldc[0] = CFF; o.seg_tiles = CFF / 16;
        o.relu = 1; o.ln_g = Wt.ln2_w; o.ln_b = Wt.ln2_b;
      } else {                                           // x = x2 + hidden W2^T   (the next layer's input)
        o.A = A.hmid; o.lda = CFF; o.ldw = CFF; o.nchunk = CFF / CD; o.W[0] = Wt.ff_w2; o.bias[0] = Wt.ff_b2; o.out[0] = A.x;
        o.ldc[0] = CD; o.resid = A.x2;
      }
    }
    ops[tid] = o;
  }
  if (tid == 0) abort_s = 0;

  // sentences of this wavefront in the per-sentence operations: sl = wave * 32 + m + 128 i  (< nb)
  const int sl0 = wave * NMEMBER + m;
  const __amdgpu_buffer_rsrc_t rs_x = rsrc_of(A.x);

  // ---- step-0 input: candidates[b, 0] + pos_embed[0]  (transformer.py:534-541); allpos[:, 0] = 0 ----
  for (int i = 0; i < MAXSPW; ++i) {
    const int sl = sl0 + 128 * i;
    if (sl >= nb) break;
    const int b = b0c + sl;
    const float4 m4 = *reinterpret_cast<const float4*>(A.memory + (size_t)b * S1 * CD + 4 * lane);
    const float4 p4 = *reinterpret_cast<const float4*>(A.pos_w + 4 * lane);
    st16_sc1(rs_x, (uint32_t)(b * CD + 4 * lane) * 4u, make_float4(m4.x + p4.x, m4.y + p4.y, m4.z + p4.z, m4.w + p4.w));
    if (lane == 0) {
      now_s[wave][i] = 0;
      A.allpos[(int64_t)b * (L + 1)] = 0;
    }
  }

  unsigned epoch = 0;
  float4 wf[TG][4];                // weight ring: slot u holds the 16-deep k slice of this lane for tile u of the current group
  float ln_g1 = 0.f, ln_b1 = 0.f;  // LayerNorm scale / shift of column tid: requested before the barrier wait, parked in LDS after it

  // weights of sequence element (group g, chunk ch, slot u) of operation oi
  auto load_w = [&](int oi, int g, int ch, int u) {
    const Op& o = ops[oi];
    const int seg_tiles = uni(o.seg_tiles), ldw = uni(o.ldw);
    const int total = uni(o.nseg) * seg_tiles;
    const int t = cs + cpr * (TG * g + u);
    if (t >= total) {
#pragma unroll
      for (int i = 0; i < 4; ++i) kill4(wf[u][i]);
      return;
    }
    const int seg = t / seg_tiles, ct = t - seg * seg_tiles;
    const float* W = uni(seg == 0 ? o.W[0] : seg == 1 ? o.W[1] : o.W[2]);
    const float* p = W + (size_t)(16 * ct + lr) * ldw + CD * ch + 64 * wave + 16 * lg;
#pragma unroll
    for (int i = 0; i < 4; ++i) wf[u][i] = *reinterpret_cast<const float4*>(p + 4 * i);
  };
  // what the next operation will need that nobody writes in this launch: requested before the wait at the barrier
  auto prefetch = [&](int oi) {
    const Op& o = ops[oi];
    if (uni(o.type) != OP_GEMM || nvalid <= 0) {
#pragma unroll
      for (int u = 0; u < TG; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) kill4(wf[u][i]);
      return;
    }
#pragma unroll
    for (int u = 0; u < TG; ++u) load_w(oi, 0, 0, u);
    const float* g = uni(o.ln_g);
    if (g) {
      ln_g1 = g[tid];
      ln_b1 = uni(o.ln_b)[tid];
    }
  };
  // cluster barrier: every wave has drained its sc1 stores, one lane adds, polls, the workgroup follows
  auto barrier = [&](int next_op) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prefetch(next_op);
    if (tid == 0) {
      const unsigned target = epoch * NMEMBER;
      unsigned spins = 0;
      long long t0 = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {
          const long long now = wall_clock64();            // 100 MHz
          if (t0 == 0) t0 = now;
          else if (now - t0 > 100000000ll) {                // 1 s: some workgroup of the cluster is not running
            atomicExch(&g_cluster_err, 1u);
            abort_s = 1;
            break;
          }
        }
      }
    }
    __syncthreads();
    return abort_s == 0;
  };

#ifdef CLUSTER_STAMPS
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = wall_clock64();
#endif
  if (!barrier(0)) return;
  TS(0);

  for (int k = 0; k < L; ++k) {
    for (int oi = 0; oi < nops; ++oi) {
      const Op& o = ops[oi];
      const int type = uni(o.type);
      if (type == OP_GEMM) {
        if (nvalid > 0) {
          const int nchunk = uni(o.nchunk), seg_tiles = uni(o.seg_tiles), lda = uni(o.lda), relu = uni(o.relu);
          const int total = uni(o.nseg) * seg_tiles;
          const int nmine = cs < total ? (total - cs + cpr - 1) / cpr : 0;
          const float* Ap = uni(o.A);
          const float* resid = uni(o.resid);
          const bool has_ln = uni(o.ln_g) != nullptr;
          const __amdgpu_buffer_rsrc_t rs_a = rsrc_of(Ap);
          // A chunk -> registers (sc1), rows past the cluster's last sentence clamped to it (finite, never stored)
          float4 ra[8];
          auto load_a = [&](int ch) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              int row = wave + 4 * j;                   // float4 index tid + 256 j: row = index / 64 is the same for a wavefront
              row = row < nvalid ? row : nvalid - 1;
              ra[j] = ld16_so<SC1>(rs_a, (uint32_t)lane * 16u, (uint32_t)((r0 + row) * lda + CD * ch) * 4u);
            }
          };
          auto store_a = [&]() {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int idx = tid + 256 * j;
              *reinterpret_cast<float4*>(As + (idx >> 6) * LDX + 4 * (idx & 63)) = ra[j];
            }
          };
          load_a(0);
          for (int g = 0; TG * g < nmine; ++g) {
            // epilogue operands of this group's tiles (threads 0..127: row tid / 4, columns 4 (tid % 4) ..)
            float4 bias4[TG], res4[TG];
            const int erow = tid >> 2, ec4 = tid & 3;
#pragma unroll
            for (int u = 0; u < TG; ++u) bias4[u] = res4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < 128) {
#pragma unroll
              for (int u = 0; u < TG; ++u) {
                const int t = cs + cpr * (TG * g + u);
                if (TG * g + u < nmine) {
                  const int seg = t / seg_tiles, ct = t - seg * seg_tiles;
                  const float* bp = uni(seg == 0 ? o.bias[0] : seg == 1 ? o.bias[1] : o.bias[2]);
                  bias4[u] = *reinterpret_cast<const float4*>(bp + 16 * ct + 4 * ec4);
                  if (resid) {
                    const int row = erow < nvalid ? erow : nvalid - 1;
                    res4[u] = ld16_sc1(rsrc_of(resid), (uint32_t)((r0 + row) * CD + 16 * ct + 4 * ec4) * 4u);
                  }
                }
              }
            }
            f32x4 acc[TG][2];
#pragma unroll
            for (int u = 0; u < TG; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int ch = 0; ch < nchunk; ++ch) {
              // the activation image: once per operation when K = 256 (every group reads the same one), once per chunk else
              if (nchunk > 1 || g == 0) {
                if (g > 0 || ch > 0) __syncthreads();    // every wave has read the previous image
                store_a();
                TS(1);
                if (has_ln && ch == 0) { ln_s[tid] = ln_g1; ln_s[CD + tid] = ln_b1; }
                if (ch + 1 < nchunk) load_a(ch + 1);     // in flight during this chunk's MFMAs
                else if (nchunk > 1 && TG * (g + 1) < nmine) load_a(0);
                __syncthreads();
                if (has_ln && ch == 0) {
                  // LayerNorm in place (nn.LayerNorm(d, eps=1e-6), two-pass): wavefront w rows 8w..8w+7, 16 lanes per row
#pragma unroll
                  for (int it = 0; it < 2; ++it) {
                    float* rowp = As + (8 * wave + 4 * it + lg) * LDX + 4 * lr;
                    float4 v[4];
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                      v[j] = *reinterpret_cast<const float4*>(rowp + 64 * j);
                      s += v[j].x + v[j].y + v[j].z + v[j].w;
                    }
                    const float mu = group16_sum(s) * (1.f / CD);
                    float s2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                      v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
                      s2 += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
                    }
                    const float rs = rsqrtf(group16_sum(s2) * (1.f / CD) + 1e-6f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                      const float4 g4 = *reinterpret_cast<const float4*>(ln_s + 4 * lr + 64 * j);
                      const float4 b4 = *reinterpret_cast<const float4*>(ln_s + CD + 4 * lr + 64 * j);
                      *reinterpret_cast<float4*>(rowp + 64 * j) =
                          make_float4(v[j].x * rs * g4.x + b4.x, v[j].y * rs * g4.y + b4.y, v[j].z * rs * g4.z + b4.z, v[j].w * rs * g4.w + b4.w);
                    }
                  }
                  __syncthreads();
                  TS(2);
                }
              }
              // MFMA: lane (lr, lg) of wave w feeds k = 64 w + 16 lg + 4 i + s, the same for A and W
              float4 af[2][4];
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  af[rb][i] = *reinterpret_cast<const float4*>(As + (16 * rb + lr) * LDX + 64 * wave + 16 * lg + 4 * i);
#pragma unroll
              for (int u = 0; u < TG; ++u) {
                if (TG * g + u < nmine) {
#pragma unroll
                  for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                      const float bv = pick4(wf[u][i], s);
                      acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick4(af[0][i], s), bv, acc[u][0], 0, 0, 0);
                      acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pick4(af[1][i], s), bv, acc[u][1], 0, 0, 0);
                    }
                  // refill the slot with its successor in the sequence (next chunk of the tile, else next group's tile)
                  if (ch + 1 < nchunk) load_w(oi, g, ch + 1, u);
                  else if (TG * (g + 1) + u < nmine) load_w(oi, g + 1, 0, u);
                }
              }
            }
            TS(3);
            // ---- K-split reduction through LDS + epilogue, tile by tile ----
#pragma unroll
            for (int u = 0; u < TG; ++u) {
              if (TG * g + u < nmine) {
                if (u > 0 || g > 0) __syncthreads();     // the previous tile's sums have been read
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                  for (int r = 0; r < 4; ++r) red[(wave * 32 + 16 * rb + 4 * lg + r) * RLD + lr] = acc[u][rb][r];
                __syncthreads();
                if (tid < 128) {
                  float4 v = bias4[u];
#pragma unroll
                  for (int w = 0; w < 4; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + (w * 32 + erow) * RLD + 4 * ec4);
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                  }
                  v.x += res4[u].x; v.y += res4[u].y; v.z += res4[u].z; v.w += res4[u].w;
                  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                  if (erow < nvalid) {
                    const int t = cs + cpr * (TG * g + u);
                    const int seg = t / seg_tiles, ct = t - seg * seg_tiles;
                    float* outp = uni(seg == 0 ? o.out[0] : seg == 1 ? o.out[1] : o.out[2]);
                    const int ldc = uni(seg == 0 ? o.ldc[0] : seg == 1 ? o.ldc[1] : o.ldc[2]);
                    const int kst = uni(seg == 0 ? o.kstride[0] : seg == 1 ? o.kstride[1] : o.kstride[2]);
                    st16_sc1(rsrc_of(outp), (uint32_t)((r0 + erow) * ldc + kst * k + 16 * ct + 4 * ec4) * 4u, v);
                  }
                }
              }
            }
          }
        }
        TS(4);
      } else if (type == OP_ATTN) {
        // scaled_attention (transformer.py:296-303), one query per sentence and head: lane l owns dims 4l..4l+3 (head l / 16)
        const int self = uni(o.self), sstride = uni(o.sent_stride);
        const float* Kp = uni(o.Kp);
        const float* Vp = uni(o.Vp);
        const __amdgpu_buffer_rsrc_t rs_q = rsrc_of(uni(o.qin)), rs_o = rsrc_of(uni(o.out[0]));
        const __amdgpu_buffer_rsrc_t rs_k = rsrc_of(Kp), rs_v = rsrc_of(Vp);
        for (int i = 0; i < MAXSPW; ++i) {
          const int sl = sl0 + 128 * i;
          if (sl >= nb) break;
          const int b = b0c + sl;
          int nk;
          if (self) nk = k + 1;
          else {
            const int len = (int)A.olen[b];
            nk = (len + 1 < S1) ? len + 1 : S1;            // memory position p is padding when p >= 1 and p - 1 >= len
          }
          nk = uni(nk);
          const float4 q = ld16_sc1(rs_q, (uint32_t)(b * CD + 4 * lane) * 4u);
          const uint32_t sbase = (uint32_t)uni(b * sstride) * 4u, loff = (uint32_t)lane * 16u;
          // keys in blocks of KB with a running maximum (online softmax): the K and V rows of a block are requested together,
          // so a sentence costs ceil(nk / KB) memory round trips and 2 KB float4 registers
          float mrun = -INFINITY, sum = 0.f;
          float4 ov = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
          for (int j0 = 0; j0 < nk; j0 += KB) {
            // (rows past the last key repeat it: every register is defined in every iteration, the score is masked instead)
            float4 kb[KB], vb[KB];
            if (self) {
#pragma unroll
              for (int jj = 0; jj < KB; ++jj) {
                const int j = j0 + jj < nk ? j0 + jj : nk - 1;
                kb[jj] = ld16_so<SC1>(rs_k, loff, sbase + (uint32_t)j * CD * 4u);
                vb[jj] = ld16_so<SC1>(rs_v, loff, sbase + (uint32_t)j * CD * 4u);
              }
            } else {
#pragma unroll
              for (int jj = 0; jj < KB; ++jj) {
                const int j = j0 + jj < nk ? j0 + jj : nk - 1;
                kb[jj] = ld16_so<0>(rs_k, loff, sbase + (uint32_t)j * CD * 4u);
                vb[jj] = ld16_so<0>(rs_v, loff, sbase + (uint32_t)j * CD * 4u);
              }
            }
            float sc[KB];
            float bm = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < KB; ++jj) {
              const float d = group16_sum(q.x * kb[jj].x + q.y * kb[jj].y + q.z * kb[jj].z + q.w * kb[jj].w) * 0.125f;   // / sqrt(d_k = 64)
              sc[jj] = j0 + jj < nk ? d : -INFINITY;
              bm = fmaxf(bm, sc[jj]);
            }
            const float mnew = fmaxf(mrun, bm);
            const float rescale = expf(mrun - mnew);       // 0 for the first block
            sum *= rescale; ov.x *= rescale; ov.y *= rescale; ov.z *= rescale; ov.w *= rescale;
#pragma unroll
            for (int jj = 0; jj < KB; ++jj) {
              const float p = expf(sc[jj] - mnew);         // 0 for a masked key
              sum += p;
              ov.x = fmaf(p, vb[jj].x, ov.x); ov.y = fmaf(p, vb[jj].y, ov.y); ov.z = fmaf(p, vb[jj].z, ov.z); ov.w = fmaf(p, vb[jj].w, ov.w);
            }
            mrun = mnew;
          }
          const float inv = 1.f / sum;
          st16_sc1(rs_o, (uint32_t)(b * CD + 4 * lane) * 4u, make_float4(ov.x * inv, ov.y * inv, ov.z * inv, ov.w * inv));
        }
        TS(5);
      } else {
        // ---- skip decision (:543-567) and the next input (:535-541), one wavefront per sentence ----
        const __amdgpu_buffer_rsrc_t rs_qh = rsrc_of(A.qh);
        for (int i = 0; i < MAXSPW; ++i) {
          const int sl = sl0 + 128 * i;
          if (sl >= nb) break;
          const int b = b0c + sl;
          const int now = now_s[wave][i];
          const float4 x4 = ld16_sc1(rs_x, (uint32_t)(b * CD + 4 * lane) * 4u);
          const float4 q4 = ld16_sc1(rs_qh, (uint32_t)(b * CD + 4 * lane) * 4u);
          const float4 g4 = *reinterpret_cast<const float4*>(A.norm_w + 4 * lane);
          const float4 n4 = *reinterpret_cast<const float4*>(A.norm_b + 4 * lane);
          const float mu = wave64_sum(x4.x + x4.y + x4.z + x4.w) * (1.f / CD);
          const float4 cx = make_float4(x4.x - mu, x4.y - mu, x4.z - mu, x4.w - mu);
          const float rs = rsqrtf(wave64_sum(cx.x * cx.x + cx.y * cx.y + cx.z * cx.z + cx.w * cx.w) * (1.f / CD) + 1e-6f);
          const float4 h4 = make_float4(cx.x * rs * g4.x + n4.x, cx.y * rs * g4.y + n4.y, cx.z * rs * g4.z + n4.z, cx.w * rs * g4.w + n4.w);
          float logit[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float* kr = j == 0 ? A.kunk : A.KC + ((size_t)b * NC + now + j) * CD;
            const float4 k4 = *reinterpret_cast<const float4*>(kr + 4 * lane);
            const float4 l4 = *reinterpret_cast<const float4*>(A.lin_w + j * CD + 4 * lane);
            const float dq = wave64_sum(q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w);
            const float dl = wave64_sum(h4.x * l4.x + h4.y * l4.y + h4.z * l4.z + h4.w * l4.w);
            logit[j] = dq / 16.f + (dl + A.lin_b[j]);                                  // / sqrt(d_model), :551-552
          }
          if (k == 0) {                                                                // :553-555
            const int len = (int)A.olen[b];
            logit[0] = -1e9f;
#pragma unroll
            for (int j = 1; j < 4; ++j) logit[j] = logit[j] - ((j - 1 >= len) ? 1.f : 0.f) * 1e9f;
          }
          int delta = 0;
          float best = logit[0];
#pragma unroll
          for (int j = 1; j < 4; ++j)
            if (logit[j] > best) { best = logit[j]; delta = j; }                       // first maximum wins (:559)
          if (A.sample_u) {                                                            // :560-561, as nast_lp_step_f32
            float p[4], tot = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { p[j] = expf(logit[j] - best); tot += p[j]; }
            const float u = A.sample_u[(size_t)k * B + b] * tot;
            float cdf = 0.f;
            int pickd = -1, last = delta;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              cdf += p[j];
              if (p[j] > 0.f) last = j;
              if (pickd < 0 && p[j] > 0.f && cdf > u) pickd = j;
            }
            delta = pickd < 0 ? last : pickd;
          }
          int nxt = now + delta;
          nxt = nxt > S1 ? S1 : nxt;                                                   // :565
          if (lane == 0) {
            now_s[wave][i] = nxt;
            A.allpos[(int64_t)b * (L + 1) + k + 1] = nxt;
          }
          if (k + 1 < L) {
            const float* src = delta == 0 ? A.unk : A.cand + ((size_t)b * NC + nxt) * CD;
            const float4 s4 = *reinterpret_cast<const float4*>(src + 4 * lane);
            const float4 p4 = *reinterpret_cast<const float4*>(A.pos_w + (size_t)(k + 1) * CD + 4 * lane);
            st16_sc1(rs_x, (uint32_t)(b * CD + 4 * lane) * 4u, make_float4(s4.x + p4.x, s4.y + p4.y, s4.z + p4.z, s4.w + p4.w));
          }
        }
      }
      if (type == OP_STEP) TS(6);
      if (k == L - 1 && oi == nops - 1) break;           // nothing follows: no one waits for this workgroup's last stores
      if (!barrier(oi + 1 == nops ? 0 : oi + 1)) return;
      TS(0);
    }
  }
#ifdef CLUSTER_STAMPS
  if (tid == 0 && c == 0 && (m == 0 || m == NMEMBER - 1))
    for (int i = 0; i < 8; ++i) g_cluster_dbg[m == 0 ? 0 : 1][i] = tacc[i];
#endif
}

}  // namespace

int nast_lp_steps_from_allpos_launch(const int64_t* allpos, const int64_t* olen, int B, int L, int* steps_dev, hipStream_t s);   // decode_fused.hip

bool nast_lp_decode_cluster_eligible(const nast_lp_decode_args* A) {
  const char* e = getenv("NAST_DECODE_CLUSTER");      // read per call: tests switch it in-process
  if (e && atoi(e) == 0) return false;
  if (A->d != CD || A->dff != CFF || A->h != 4 || A->nlayers < 1 || A->nlayers > 4 || A->L > MAXKEYS || A->S1 > MAXKEYS) return false;
  if (A->B < 1 || A->B > 2048) return false;
  static int cus = -1;
  if (cus < 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
    cus = n;
  }
  if (cus < 128) return false;                         // 256 workgroups at <= 3 per CU must all be resident
  auto al = [](const void* p) { return p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  for (int i = 0; i < A->nlayers; ++i) {
    const nast_declayer_weights& w = A->layers[i];
    const float* const ps[] = {w.ln0_w, w.ln0_b, w.sa_wq, w.sa_bq, w.sa_wk, w.sa_bk, w.sa_wv, w.sa_bv, w.sa_wo, w.sa_bo, w.ln1_w, w.ln1_b, w.ca_wq,
                               w.ca_bq, w.ca_wo, w.ca_bo, w.ln2_w, w.ln2_b, w.ff_w1, w.ff_b1, w.ff_w2, w.ff_b2};
    for (const float* p : ps)
      if (!al(p)) return false;
  }
  return al(A->norm_w) && al(A->norm_b) && al(A->wq_w) && al(A->wq_b) && al(A->lin_w) && A->lin_b != nullptr && al(A->pos_w) && al(A->unk) &&
         al(A->memory);
}

// cand / KC / kunk / mk / mv are the tables lp_decode_impl has already enqueued on `s`; the rest is workspace the kernel
// writes before it reads; `sync` (8 x 32 words) is zeroed here, on the stream, before every launch
int nast_lp_decode_cluster_launch(const nast_lp_decode_args* A, const float* cand, const float* KC, const float* kunk, const float* mk,
                                  const float* mv, float* kc, float* vc, float* x, float* q, float* a, float* x1, float* x2, float* hmid,
                                  float* qh, unsigned* sync, int* steps_dev, hipStream_t s) {
  ClusterArgs F;
  for (int i = 0; i < A->nlayers; ++i) F.layers[i] = A->layers[i];
  for (int i = A->nlayers; i < 4; ++i) F.layers[i] = A->layers[0];
  F.norm_w = A->norm_w; F.norm_b = A->norm_b; F.wq_w = A->wq_w; F.wq_b = A->wq_b; F.lin_w = A->lin_w; F.lin_b = A->lin_b;
  F.pos_w = A->pos_w; F.unk = A->unk; F.memory = A->memory;
  F.cand = cand; F.KC = KC; F.kunk = kunk; F.mk = mk; F.mv = mv;
  F.kc = kc; F.vc = vc; F.x = x; F.q = q; F.a = a; F.x1 = x1; F.x2 = x2; F.hmid = hmid; F.qh = qh;
  F.olen = A->origin_length; F.allpos = A->allpos; F.sample_u = A->sample_u; F.sync = sync;
  F.B = A->B; F.S1 = A->S1; F.L = A->L; F.nl = A->nlayers;
  F.Bc = (A->B + NCLUSTER - 1) / NCLUSTER;
  int nrt = 1;
  while (32 * nrt < F.Bc) nrt *= 2;
  F.nrt = nrt;                                         // <= 8 (B <= 2048): at least 4 column slots per row tile
  if (hipMemsetAsync(sync, 0, NCLUSTER * 32 * sizeof(unsigned), s) != hipSuccess) {
    nast_set_error("nast_lp_decode_f32(cluster): memset failed: %s", hipGetErrorString(hipGetLastError()));
    return NAST_ELAUNCH;
  }
  nast_prof_before(6, s);
  hipLaunchKernelGGL(nast_lp_decode_cluster_kernel, dim3(NCLUSTER * NMEMBER), dim3(256), 0, s, F);
  {   // profiling family 6: the whole decode as one launch; work = FLOP of its projections (6 d x d + 2 d x dff per layer, + wq)
    const double per_row = 2.0 * (A->nlayers * (6.0 * CD * CD + 2.0 * CD * CFF) + (double)CD * CD);
    nast_prof_after(6, s, per_row * A->B * A->L);
  }
  if (steps_dev && nast_lp_steps_from_allpos_launch(A->allpos, A->origin_length, A->B, A->L, steps_dev, s) != 0) return NAST_ELAUNCH;
#ifdef CLUSTER_STAMPS
  {
    long long h[2][8];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cluster_dbg), sizeof(h));
    for (int w = 0; w < 2; ++w)
      fprintf(stderr, "[cluster stamps, us, member %d] barrier %.0f  A-load %.0f  LN %.0f  MFMA %.0f  reduce+epilogue %.0f  attention %.0f  step %.0f\n",
              w == 0 ? 0 : NMEMBER - 1, h[w][0] / 100.0, h[w][1] / 100.0, h[w][2] / 100.0, h[w][3] / 100.0, h[w][4] / 100.0, h[w][5] / 100.0, h[w][6] / 100.0);
  }
#endif
  NAST_CHECK_LAUNCH("nast_lp_decode_f32(cluster)");
  return NAST_OK;
}

// 0 = every cluster decode on the current device completed its barriers; 1 = a barrier poll timed out (the results of that
// call are garbage).  Synchronous (reads a device word); call it where the caller synchronises anyway.
extern "C" int nast_lp_decode_cluster_status(int* status_out) {
  NAST_REQUIRE(status_out, "nast_lp_decode_cluster_status: NULL");
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_cluster_err), sizeof(v)) != hipSuccess) {
    nast_set_error("nast_lp_decode_cluster_status: read failed: %s", hipGetErrorString(hipGetLastError()));
    return NAST_ELAUNCH;
  }
  *status_out = (int)v;
  return NAST_OK;
}
