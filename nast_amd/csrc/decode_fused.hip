// Skip decoding of the LengthPredictor (transformer.py:525-570) as ONE kernel: a workgroup owns 16 sentences and takes them
// through every decode step and every decoder layer on its own -- LayerNorm, the Q/K/V projections (K/V appended to a
// cache in global memory), attention over the cache, the cross-attention over the pre-projected memory, the feed-forward
// block, the final LayerNorm, wq, the skip decision and the next input -- with no grid-wide dependency, so no launch per
// operation (the step-by-step path of decode.hip issues ~26 launches per step, 830 per decode, each waiting for the last).
//
// Shape of the work: every GEMM is [16 rows] x [256 x 256 weight block]; the 4 wavefronts split the 256 output columns
// (64 each = 4 MFMA tiles of v_mfma_f32_16x16x4_f32), the activations sit in LDS, and each lane streams its weight rows
// straight from global memory (L2-resident: every workgroup walks the same 11 MB of decoder weights in the same order)
// into a ring of four register buffers, three 32-deep K chunks ahead of the MFMAs.  The ring is handed from one GEMM to
// the next (the last three chunk slots of a GEMM already fetch the first three chunks of the following weight block), so
// the memory latency is paid once per kernel, not once per GEMM.  One CU per 16 sentences: at 512 sentences the decode
// keeps 32 of the 256 CUs busy for its whole duration and the rest of the chip stays free for the pass that runs beside
// it on the other stream (seq2seq.py f_step: the self-reconstruction pass).
//
// Limits (nast_lp_decode_fused_eligible): d_model 256, 4 heads, d_ff 1024, <= 4 layers, max_length and memory width <= 36
// (keys per attention row are held in registers).  Anything else takes the step-by-step path.
//
// MEASURED (round 2, MI355X, tools/decode_bench.py, in-kernel cycle stamps with -DFUSED_STAMPS) and therefore OPT-IN
// (NAST_DECODE_FUSED=1; default is the step-by-step path): one decode takes 18-21 ms against 6.8 / 7.3 / 8.9 ms step by
// step at 16 / 64 / 512 sentences.  Per layer and step: the 14 GEMMs 133 us (9.5 us each = 24k cycles for 256 MFMAs that
// need 8.2k), cross-attention 25-55 us, self-attention 10-20 us, LayerNorm 4 us.  The GEMMs run at the rate one CU can pull
// weights that no cache near it holds -- 256 KB per 9.5 us = 27 GB/s; 11 MB of decoder weights per step against 4 MB of L2
// per XCD -- and a deeper ring (RD = 8, 56 of the 63 loads a wavefront may have pending) changes nothing: the cap is the
// CU's own miss path, not latency x registers.  11 MB x 32 steps at 27 GB/s = 13 ms is the floor of ANY design that keeps a
// sentence block on one CU; the step-by-step path spreads every GEMM over the chip and pays launch latency instead.
// What would beat both is weight-stationary across CUs with a grid barrier per operation (~1500 barriers per decode);
// not built.  The kernel stays as a tested alternative (tests/test_decode_fused_gpu.py: identical positions).
#include "common.h"

namespace {

constexpr int FD = 256, FFF = 1024, FH = 4, FDK = 64, FR = 16, LDX = FD + 4;
constexpr int MAXKEYS = 36;
constexpr int GEMMS_PER_LAYER = 14;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FusedDecodeArgs {
  nast_declayer_weights layers[4];
  const float *norm_w, *norm_b, *wq_w, *wq_b, *lin_w, *lin_b, *pos_w, *unk, *memory;
  const float *cand, *KC, *kunk, *mk, *mv;
  float *kc, *vc;
  const int64_t* olen;
  int64_t* allpos;
  const float* sample_u;
  int B, S1, L, nl;
#ifdef FUSED_STAMPS
  long long* dbg;
#endif
};

struct WRef {
  const float* w;
  int ld;
};

// one K chunk (32 deep) of this lane's weight rows: 4 column tiles x 8 consecutive k
__device__ __forceinline__ void ring_load(float4 (&dst)[4][2], const float* __restrict__ W, int ld, int chunk, int lr, int lg, int wave) {
  const float* p = W + (size_t)(wave * 64 + lr) * ld + 32 * chunk + 8 * lg;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    dst[t][0] = *reinterpret_cast<const float4*>(p + (size_t)t * 16 * ld);
    dst[t][1] = *reinterpret_cast<const float4*>(p + (size_t)t * 16 * ld + 4);
  }
}

__device__ __forceinline__ float pick(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

// acc[t] += A[16 x 256] (LDS, row stride LDX) . W[wave*64 + 16 t + (0..15)][0..255]^T.   Lane (lr = lane % 16, lg = lane / 16)
// feeds row lr of A and column lr of the tile; within a 32-deep chunk c its k values are 32 c + 8 lg + s, s = 0..7 (any
// assignment works as long as A and B agree).  Result: acc[t][i] = C[4 lg + i][wave*64 + 16 t + lr].
// Ring protocol (RD slots, RD - 1 chunks in flight): on entry slots 0..RD-2 hold chunks 0..RD-2 of `cur`; on exit they hold
// chunks 0..RD-2 of `nxt`.  (Measured: RD = 8 -- 56 KB per wavefront outstanding -- is no faster than RD = 4; the stream is
// capped by what one CU's L1 can have outstanding towards L2, see the header.)
constexpr int RD = 4;
__device__ __forceinline__ void gemm256(const float* __restrict__ A, WRef cur, WRef nxt, float4 (&R)[RD][4][2], f32x4 (&acc)[4], int lr,
                                        int lg, int wave) {
  const float* ap = A + lr * LDX + 8 * lg;
#pragma unroll 1
  for (int cc = 0; cc < 8; cc += RD) {
#pragma unroll
    for (int j = 0; j < RD; ++j) {
      const int c = cc + j;
      const int ci = c + RD - 1;
      const bool own = ci < 8;
      ring_load(R[(j + RD - 1) % RD], own ? cur.w : nxt.w, own ? cur.ld : nxt.ld, own ? ci : ci - 8, lr, lg, wave);
      const float4 a0 = *reinterpret_cast<const float4*>(ap + 32 * c);
      const float4 a1 = *reinterpret_cast<const float4*>(ap + 32 * c + 4);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const float av = pick(s < 4 ? a0 : a1, s & 3);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, pick(R[j][t][s >> 2], s & 3), acc[t], 0, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void zero(f32x4 (&acc)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// LayerNorm of the 16 rows (nn.LayerNorm(d, eps=1e-6), transformer.py:332): wavefront w takes rows 4w..4w+3
template <int RPW>
__device__ __forceinline__ void ln_rows(const float* __restrict__ src, const float* __restrict__ gw, const float* __restrict__ gb,
                                        float* __restrict__ dst, int lane, int wave) {
  const float4 g4 = *reinterpret_cast<const float4*>(gw + 4 * lane);
  const float4 b4 = *reinterpret_cast<const float4*>(gb + 4 * lane);
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = RPW * wave + i;
    const float4 x = *reinterpret_cast<const float4*>(src + r * LDX + 4 * lane);
    const float mu = wave_sum(x.x + x.y + x.z + x.w) * (1.f / FD);
    const float4 c = make_float4(x.x - mu, x.y - mu, x.z - mu, x.w - mu);
    const float rs = rsqrtf(wave_sum(c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w) * (1.f / FD) + 1e-6f);
    *reinterpret_cast<float4*>(dst + r * LDX + 4 * lane) =
        make_float4(c.x * rs * g4.x + b4.x, c.y * rs * g4.y + b4.y, c.z * rs * g4.z + b4.z, c.w * rs * g4.w + b4.w);
  }
}

// scaled_attention (transformer.py:296-303) of one query per sentence and head over n keys.  Wavefront w takes rows 4w..4w+3,
// one row at a time with all 4 heads: lane l owns dims 4l..4l+3 of the 256 (head = l / 16), so a key or value row is one
// coalesced 1-KB read.  What bounds this phase is memory latency (one wavefront per SIMD, operands in L2): ALL keys of the
// row are requested before the first is used (n <= 36 float4 registers), then all values -- two latency exposures per row.
// K / V rows of sentence b start at base + b * sent_stride (row stride FD); keys j >= limit[r] are padding (masked to -inf).
template <int RPW>
__device__ __forceinline__ void attend(const float* __restrict__ qs, const float* __restrict__ Kbase, const float* __restrict__ Vbase,
                                       size_t sent_stride, int n, const int* __restrict__ limit, int b0, int B, float* __restrict__ out,
                                       int lane, int wave) {
#pragma unroll 1
  for (int i = 0; i < RPW; ++i) {
    const int r = RPW * wave + i;
    int b = b0 + r;
    b = b < B ? b : B - 1;
    const int lim = limit ? limit[r] : n;           // >= 1: key 0 (the style token / the first position) is never masked
    const int nk = lim < n ? lim : n;
    const float* kp = Kbase + (size_t)b * sent_stride + 4 * lane;
    const float* vp = Vbase + (size_t)b * sent_stride + 4 * lane;
    const float4 q = *reinterpret_cast<const float4*>(qs + r * LDX + 4 * lane);
    float4 buf[MAXKEYS];
#pragma unroll
    for (int j = 0; j < MAXKEYS; ++j)
      if (j < nk) buf[j] = *reinterpret_cast<const float4*>(kp + (size_t)j * FD);
    float sc[MAXKEYS];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXKEYS; ++j) {
      if (j < nk) {
        sc[j] = group16_sum(q.x * buf[j].x + q.y * buf[j].y + q.z * buf[j].z + q.w * buf[j].w) * 0.125f;   // / sqrt(d_k = 64)
        m = fmaxf(m, sc[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < MAXKEYS; ++j)
      if (j < nk) buf[j] = *reinterpret_cast<const float4*>(vp + (size_t)j * FD);
    float sum = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXKEYS; ++j) {
      if (j < nk) {
        const float p = expf(sc[j] - m);
        sum += p;
        o.x = fmaf(p, buf[j].x, o.x); o.y = fmaf(p, buf[j].y, o.y); o.z = fmaf(p, buf[j].z, o.z); o.w = fmaf(p, buf[j].w, o.w);
      }
    }
    const float inv = 1.f / sum;
    *reinterpret_cast<float4*>(out + r * LDX + 4 * lane) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
  }
}

// ROWS = sentences per workgroup: 16 (a full MFMA tile) or 8 (rows 8..15 of the tile idle: the GEMMs are bound by the weight
// stream, not by the matrix pipe, so halving the rows halves the attention / LayerNorm time of a workgroup at no GEMM cost
// and spreads a batch over twice as many CUs).
template <int ROWS>
__global__ __launch_bounds__(256, 1) void nast_lp_decode_fused_kernel(const FusedDecodeArgs A) {
  constexpr int RPW = ROWS / 4;
  __shared__ __attribute__((aligned(16))) float xs[FR * LDX];   // residual stream of the 16 sentences
  __shared__ __attribute__((aligned(16))) float as[FR * LDX];   // GEMM operand: LayerNorm output / attention output
  __shared__ __attribute__((aligned(16))) float qs[FR * LDX];   // queries; FFN hidden quarter; wq(hidden)
  __shared__ WRef sched[4 * GEMMS_PER_LAYER + 1];
  __shared__ int now_s[FR], len_s[FR], lim_s[FR];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int b0 = blockIdx.x * ROWS;
  const int B = A.B, S1 = A.S1, L = A.L, nl = A.nl;
  const int NC = S1 + 5;
  const int ngemm = nl * GEMMS_PER_LAYER + 1;

  if (threadIdx.x < ngemm) {
    const int gi = threadIdx.x;
    WRef w;
    if (gi == ngemm - 1) {
      w = WRef{A.wq_w, FD};
    } else {
      const nast_declayer_weights& Wt = A.layers[gi / GEMMS_PER_LAYER];
      const int o = gi % GEMMS_PER_LAYER;
      if (o == 0) w = WRef{Wt.sa_wq, FD};
      else if (o == 1) w = WRef{Wt.sa_wk, FD};
      else if (o == 2) w = WRef{Wt.sa_wv, FD};
      else if (o == 3) w = WRef{Wt.sa_wo, FD};
      else if (o == 4) w = WRef{Wt.ca_wq, FD};
      else if (o == 5) w = WRef{Wt.ca_wo, FD};
      else {
        const int q = (o - 6) >> 1;
        if (((o - 6) & 1) == 0) w = WRef{Wt.ff_w1 + (size_t)q * 256 * FD, FD};       // rows 256q.. of W1 [1024, 256]
        else w = WRef{Wt.ff_w2 + (size_t)q * 256, FFF};                              // columns 256q.. of W2 [256, 1024]
      }
    }
    sched[gi] = w;
  }
  if (threadIdx.x < ROWS) {
    int b = b0 + threadIdx.x;
    b = b < B ? b : B - 1;
    const int len = (int)A.olen[b];
    now_s[threadIdx.x] = 0;
    len_s[threadIdx.x] = len;
    lim_s[threadIdx.x] = (len + 1 < S1) ? len + 1 : S1;        // memory position p is padding when p >= 1 and p - 1 >= len
    if (b0 + (int)threadIdx.x < B) A.allpos[(int64_t)b * (L + 1)] = 0;
  }
  // step-0 input: candidates[b, 0] + pos_embed[0]   (transformer.py:534-541)
  for (int e = threadIdx.x; e < FR * (FD / 4); e += 256) {
    const int r = e / (FD / 4), c = (e % (FD / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < ROWS) {
      int b = b0 + r;
      b = b < B ? b : B - 1;
      const float4 m4 = *reinterpret_cast<const float4*>(A.memory + (size_t)b * S1 * FD + c);
      const float4 p4 = *reinterpret_cast<const float4*>(A.pos_w + c);
      v = make_float4(m4.x + p4.x, m4.y + p4.y, m4.z + p4.z, m4.w + p4.w);
    }
    *reinterpret_cast<float4*>(xs + r * LDX + c) = v;
    if (r >= ROWS) {                                   // idle rows of the MFMA tile: finite operands, results never read
      *reinterpret_cast<float4*>(as + r * LDX + c) = v;
      *reinterpret_cast<float4*>(qs + r * LDX + c) = v;
    }
  }
  __syncthreads();

#ifdef FUSED_STAMPS
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = clock64();
#define TS(i) do { const long long tn_ = clock64(); tacc[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define TS(i) do { } while (0)
#endif
  float4 R[RD][4][2];
#pragma unroll
  for (int j = 0; j < RD - 1; ++j) ring_load(R[j], sched[0].w, sched[0].ld, j, lr, lg, wave);
  int gi = 0;
  auto next_of = [&](int i) { return sched[i + 1 == ngemm ? 0 : i + 1]; };

  f32x4 acc[4];
  const int colbase = wave * 64 + lr;    // + 16 t
  // epilogue helpers: acc[t][i] belongs to row 4 lg + i, column colbase + 16 t
  auto to_lds = [&](float* dst, const float* __restrict__ bias, bool relu, int bias_off) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float bv = bias[bias_off + colbase + 16 * t];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[t][i] + bv;
        if (relu) v = fmaxf(v, 0.f);
        dst[(4 * lg + i) * LDX + colbase + 16 * t] = v;
      }
    }
  };
  auto add_to_x = [&](const f32x4 (&a)[4], const float* __restrict__ bias) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float bv = bias[colbase + 16 * t];
#pragma unroll
      for (int i = 0; i < 4; ++i) xs[(4 * lg + i) * LDX + colbase + 16 * t] += a[t][i] + bv;
    }
  };
  auto to_cache = [&](float* cache, const float* __restrict__ bias, int k) {      // cache [B, L, FD] of one layer
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float bv = bias[colbase + 16 * t];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int b = b0 + 4 * lg + i;
        if (4 * lg + i < ROWS && b < B) cache[((size_t)b * L + k) * FD + colbase + 16 * t] = acc[t][i] + bv;
      }
    }
  };

  for (int k = 0; k < L; ++k) {
    for (int li = 0; li < nl; ++li) {
      const nast_declayer_weights& Wt = A.layers[li];
      float* kc = A.kc + (size_t)li * B * L * FD;
      float* vc = A.vc + (size_t)li * B * L * FD;
      // ---- self-attention sublayer (transformer.py:262-265) ----
      TS(5);
      ln_rows<RPW>(xs, Wt.ln0_w, Wt.ln0_b, as, lane, wave);
      __syncthreads();
      TS(0);
      zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
      to_lds(qs, Wt.sa_bq, false, 0);
      zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
      to_cache(kc, Wt.sa_bk, k);
      zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
      to_cache(vc, Wt.sa_bv, k);
      __threadfence_block();
      __syncthreads();
      TS(1);
      attend<RPW>(qs, kc, vc, (size_t)L * FD, k + 1, nullptr, b0, B, as, lane, wave);
      __syncthreads();
      TS(2);
      zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
      add_to_x(acc, Wt.sa_bo);
      __syncthreads();
      TS(1);
      // ---- cross-attention sublayer over the pre-projected memory (:266-268) ----
      ln_rows<RPW>(xs, Wt.ln1_w, Wt.ln1_b, as, lane, wave);
      __syncthreads();
      TS(0);
      zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
      to_lds(qs, Wt.ca_bq, false, 0);
      __syncthreads();
      TS(1);
      attend<RPW>(qs, A.mk + (size_t)li * B * S1 * FD, A.mv + (size_t)li * B * S1 * FD, (size_t)S1 * FD, S1, lim_s, b0, B, as, lane, wave);
      __syncthreads();
      TS(3);
      zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
      add_to_x(acc, Wt.ca_bo);
      __syncthreads();
      TS(1);
      // ---- feed-forward sublayer (:269-271), the 1024 hidden units in four quarters ----
      ln_rows<RPW>(xs, Wt.ln2_w, Wt.ln2_b, as, lane, wave);
      __syncthreads();
      TS(0);
      f32x4 acc2[4];
      zero(acc2);
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave); ++gi;
        to_lds(qs, Wt.ff_b1, true, 256 * q);
        __syncthreads();
        gemm256(qs, sched[gi], next_of(gi), R, acc2, lr, lg, wave); ++gi;
        __syncthreads();
      }
      add_to_x(acc2, Wt.ff_b2);
      __syncthreads();
      TS(4);
    }
    // ---- skip decision (:543-567): hidden = LN(x), qh = wq(hidden) ----
    ln_rows<RPW>(xs, A.norm_w, A.norm_b, as, lane, wave);
    __syncthreads();
    zero(acc); gemm256(as, sched[gi], next_of(gi), R, acc, lr, lg, wave);
    gi = 0;
    to_lds(qs, A.wq_b, false, 0);
    __syncthreads();
#pragma unroll 1
    for (int i = 0; i < RPW; ++i) {
      const int r = RPW * wave + i;
      int b = b0 + r;
      const bool live = b < B;
      b = live ? b : B - 1;
      const int now = now_s[r];
      const float4 q4 = *reinterpret_cast<const float4*>(qs + r * LDX + 4 * lane);
      const float4 h4 = *reinterpret_cast<const float4*>(as + r * LDX + 4 * lane);
      float logit[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* kr = j == 0 ? A.kunk : A.KC + ((size_t)b * NC + now + j) * FD;
        const float4 k4 = *reinterpret_cast<const float4*>(kr + 4 * lane);
        const float4 l4 = *reinterpret_cast<const float4*>(A.lin_w + j * FD + 4 * lane);
        const float dq = wave_sum(q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w);
        const float dl = wave_sum(h4.x * l4.x + h4.y * l4.y + h4.z * l4.z + h4.w * l4.w);
        logit[j] = dq / 16.f + (dl + A.lin_b[j]);                                  // / sqrt(d_model), :551-552
      }
      if (k == 0) {                                                                // :553-555
        logit[0] = -1e9f;
#pragma unroll
        for (int j = 1; j < 4; ++j) logit[j] = logit[j] - ((j - 1 >= len_s[r]) ? 1.f : 0.f) * 1e9f;
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
        float c = 0.f;
        int pickd = -1, last = delta;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c += p[j];
          if (p[j] > 0.f) last = j;
          if (pickd < 0 && p[j] > 0.f && c > u) pickd = j;
        }
        delta = pickd < 0 ? last : pickd;
      }
      int nxt = now + delta;
      nxt = nxt > S1 ? S1 : nxt;                                                   // :565
      if (lane == 0) {
        now_s[r] = nxt;
        if (live) A.allpos[(int64_t)b * (L + 1) + k + 1] = nxt;
      }
      if (k + 1 < L) {                                                             // next input (:535-541)
        const float* src = delta == 0 ? A.unk : A.cand + ((size_t)b * NC + nxt) * FD;
        const float4 s4 = *reinterpret_cast<const float4*>(src + 4 * lane);
        const float4 p4 = *reinterpret_cast<const float4*>(A.pos_w + (size_t)(k + 1) * FD + 4 * lane);
        *reinterpret_cast<float4*>(xs + r * LDX + 4 * lane) = make_float4(s4.x + p4.x, s4.y + p4.y, s4.z + p4.z, s4.w + p4.w);
      }
    }
    __syncthreads();
  }
#ifdef FUSED_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0 && A.dbg) {
    TS(5);
    for (int i = 0; i < 6; ++i) A.dbg[i] = tacc[i];
  }
#endif
}

// the reference's early exit (transformer.py:569) from the finished position table: the first step k with k % 5 == 0 after
// which every sentence has consumed its source, + 1; L if there is none
__global__ void lp_steps_from_allpos_kernel(const int64_t* __restrict__ allpos, const int64_t* __restrict__ olen, int B, int L,
                                            int* __restrict__ steps) {
  __shared__ int result;
  if (threadIdx.x == 0) result = L;
  __syncthreads();
  for (int k = 0; k < L; k += 5) {
    int bad = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) bad |= allpos[(int64_t)b * (L + 1) + k + 1] < olen[b] + 1 ? 1 : 0;
    bad = __syncthreads_or(bad);
    if (!bad) {
      if (threadIdx.x == 0) result = k + 1;
      break;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *steps = result;
}

}  // namespace

int nast_lp_steps_from_allpos_launch(const int64_t* allpos, const int64_t* olen, int B, int L, int* steps_dev, hipStream_t s) {
  hipLaunchKernelGGL(lp_steps_from_allpos_kernel, dim3(1), dim3(256), 0, s, allpos, olen, B, L, steps_dev);
  return 0;
}

bool nast_lp_decode_fused_eligible(const nast_lp_decode_args* A) {
  const char* e = getenv("NAST_DECODE_FUSED");      // opt-in (see the header); read per call: tests switch it in-process
  if (!e || atoi(e) == 0) return false;
  if (A->d != FD || A->dff != FFF || A->h != FH || A->nlayers < 1 || A->nlayers > 4 || A->L > MAXKEYS || A->S1 > MAXKEYS) return false;
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

// cand / KC / kunk / mk / mv are the tables lp_decode_impl has already enqueued on `s`; kc / vc the (uninitialised) caches
int nast_lp_decode_fused_launch(const nast_lp_decode_args* A, const float* cand, const float* KC, const float* kunk, const float* mk,
                                const float* mv, float* kc, float* vc, int* steps_dev, hipStream_t s) {
  FusedDecodeArgs F;
  for (int i = 0; i < A->nlayers; ++i) F.layers[i] = A->layers[i];
  for (int i = A->nlayers; i < 4; ++i) F.layers[i] = A->layers[0];
  F.norm_w = A->norm_w; F.norm_b = A->norm_b; F.wq_w = A->wq_w; F.wq_b = A->wq_b; F.lin_w = A->lin_w; F.lin_b = A->lin_b;
  F.pos_w = A->pos_w; F.unk = A->unk; F.memory = A->memory;
  F.cand = cand; F.KC = KC; F.kunk = kunk; F.mk = mk; F.mv = mv; F.kc = kc; F.vc = vc;
  F.olen = A->origin_length; F.allpos = A->allpos; F.sample_u = A->sample_u;
  F.B = A->B; F.S1 = A->S1; F.L = A->L; F.nl = A->nlayers;
#ifdef FUSED_STAMPS
  static long long* dbg = nullptr;
  if (!dbg) { (void)hipMalloc(&dbg, 64); }
  F.dbg = dbg;
#endif
  // 8 sentences per workgroup (less attention / LayerNorm time per workgroup, twice the CUs) up to 512 sentences, else 16
  const char* re = getenv("NAST_DECODE_FUSED_ROWS");
  const int rows = re ? atoi(re) : (A->B <= 512 ? 8 : 16);
  if (rows == 8) hipLaunchKernelGGL(nast_lp_decode_fused_kernel<8>, dim3((A->B + 7) / 8), dim3(256), 0, s, F);
  else hipLaunchKernelGGL(nast_lp_decode_fused_kernel<16>, dim3((A->B + 15) / 16), dim3(256), 0, s, F);
#ifdef FUSED_STAMPS
  {
    long long h[6];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[fused stamps, cycles] ln %lld  gemm(attn proj) %lld  self-attn %lld  cross-attn %lld  ffn %lld  step/other %lld\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  }
#endif
  if (steps_dev) hipLaunchKernelGGL(lp_steps_from_allpos_kernel, dim3(1), dim3(256), 0, s, A->allpos, A->origin_length, A->B, A->L, steps_dev);
  NAST_CHECK_LAUNCH("nast_lp_decode_f32(fused)");
  return NAST_OK;
}
