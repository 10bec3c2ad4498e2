/* nast_hip.h -- C-ABI of libnast_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * NAST-StyTrans generator training step.
 *
 * The reference (thu-coai/NAST) has no FFI: its hot path is a chain of PyTorch ATen calls inside
 * styletransformer/transformer.py, styletransformer/seq2seq.py and utils/gumbel.py.  Each entry point
 * below replaces one such chain (cited as file:line under /root/reference); the Python modules in
 * nast_amd/ keep the reference's nn.Module surface and call these through ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (fp32 unless the name says otherwise), plain row-major;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream;
 *   - no entry point allocates, frees or synchronises (graph-capture safe) unless documented;
 *   - return value: 0 = ok, negative = NAST_E* below; nast_last_error() gives a message.
 */
#ifndef NAST_HIP_H
#define NAST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAST_OK 0
#define NAST_EINVAL (-1)   /* bad shape / pointer / unsupported size */
#define NAST_ELAUNCH (-2)  /* hipLaunch / runtime error */

const char* nast_last_error(void);
int nast_version(void);

/* ------------------------------------------------------------------------------------------------
 * Grouped fp32 GEMM on MFMA (v_mfma_f32_32x32x2_f32), fused epilogue.
 *   C = alpha * sum_{s<nsrc} op(A[s]) * op(B[s])  (+bias[n]) (+residual) (ReLU) (mask by relu_src>0)
 * Replaces: nn.Linear / torch.matmul / addmm chains -- transformer.py:229 (soft-token x embedding),
 * :280-294 (QKV + fc projections), :308-316 (FFN), :211-214 (generator projection), :390-392 and
 * their autograd backward (dgrad: op = NN, wgrad: op = TN).
 * Up to NAST_GEMM_MAX_GROUP independent problems run in ONE launch so that small projections fill
 * the 256 CUs together.
 * ---------------------------------------------------------------------------------------------- */
#define NAST_GEMM_MAX_GROUP 8
#define NAST_GEMM_MAX_SRC 3
#define NAST_GEMM_RELU 1        /* C = max(C, 0) after bias/residual                                 */
#define NAST_GEMM_ATOMIC 2      /* C += result with fp32 atomics (C pre-initialised); enables split-K */
#define NAST_GEMM_ACCUM 4       /* C += result, non-atomic (single writer, no split-K)               */
#define NAST_GEMM_BF16 8        /* bf16 variant: operands rounded to bf16 in registers, v_mfma_f32_32x32x16_bf16, fp32 accumulate; */
                                /* storage stays fp32.  Applies when every problem of the launch sets it (else fp32).            */

typedef struct nast_gemm_desc {
  const float* A[NAST_GEMM_MAX_SRC]; /* op(A) is M x K.  transA=0: A stored [M, lda>=K]; transA=1: [K, lda>=M] */
  const float* B[NAST_GEMM_MAX_SRC]; /* op(B) is K x N.  transB=0: B stored [K, ldb>=N]; transB=1: [N, ldb>=K] */
  float* C;                          /* [M, ldc>=N]                                                            */
  const float* bias;                 /* [N] or NULL                                                            */
  const float* residual;             /* [M, ldc] or NULL, added after bias                                     */
  const float* relu_src;             /* [M, ldc] or NULL: C = relu_src > 0 ? C : 0 (ReLU backward)             */
  float* colsum;                     /* [M] or NULL: with transA=1, colsum[m] (+)= sum_k op(A[0])[m,k]  (bias grad) */
  const float* ln_gamma;             /* [K] or NULL: op(A) rows are LayerNorm-ed on the fly (eps 1e-6) before the   */
  const float* ln_beta;              /* product (pre-LN sublayers, transformer.py:324-330). Small-M path only:      */
                                     /* needs M <= 1024, transA=0, transB=1 and K <= 256, or K <= 1024 with every     */
                                     /* problem of the launch at K >= 512, K % 16 == 0; else NAST_EINVAL.            */
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  int32_t transA, transB;
  int32_t nsrc;                      /* 1..3 products accumulated into the same C (all share M,N,K,ld*,trans*) */
  int32_t flags;
  float alpha;
} nast_gemm_desc;

int nast_gemm_f32(const nast_gemm_desc* descs, int count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (eps as given, reference uses 1e-6), one wavefront per row.
 * Replaces: nn.LayerNorm in SublayerConnection / Encoder.norm / Decoder.norm -- transformer.py:324-330,
 * :171, :188, :348-350.
 * Row map: output row r reads input row  (r / grp) * grp_stride + grp_off + r % grp   (grp = 0: identity).
 * Used to normalise only the rows a consumer needs (generator skips row 0, transformer.py:103;
 * discriminator reads row 0 only, transformer.py:151).
 * ---------------------------------------------------------------------------------------------- */
int nast_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y,
                           float* mean, float* rstd, int rows_out, int d, float eps,
                           int grp, int grp_stride, int grp_off, void* stream);
/* dx is [rows_in, d] and is fully written (rows not selected by the map get 0);
 * dx_add (may be NULL, only with grp == 0): dx = LN-gradient + dx_add  (the residual branch of
 * SublayerConnection, transformer.py:326, fused so that no separate add kernel runs);
 * dgamma/dbeta [d] are ACCUMULATED (+=, fp32 atomics) -- caller zero-initialises or passes .grad; either may be
 * NULL (frozen parameter): that sum is skipped. */
int nast_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean,
                           const float* rstd, float* dx, float* dgamma, float* dbeta, const float* dx_add,
                           int rows_out, int rows_in, int d,
                           int grp, int grp_stride, int grp_off, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused masked multi-head attention for short sequences (LDS-staged, whole K/V of one (batch, head)
 * per workgroup).  Replaces: scaled_attention + the view/transpose/contiguous around it --
 * transformer.py:287-303.
 * q [B, nq, h*dk] (row stride ldq, batch stride q_bs), k/v [B, nk, h*dk] (ldk/ldv, k_bs/v_bs),
 * out [B, nq, h*dk] (ldo, o_bs); strides in elements, so K/V caches and projection outputs are used in place.
 * mask: uint8/bool, element (b, i, j) at mask[b*mask_bs + i*mask_qs + j], nonzero = masked (-inf);
 *       key-padding masks use mask_qs = 0, the causal mask uses mask_bs = 0; NULL = no mask.
 * lse  [B, h, nq] (log-sum-exp of the scaled, masked scores) is saved for backward.
 * Limits: dk in {8,16,32,64}; K/V + score tiles must fit 160 KiB of LDS (nk <= ~160 at dk = 64).
 * ---------------------------------------------------------------------------------------------- */
int nast_mha_fwd_f32(const float* q, const float* k, const float* v, const uint8_t* mask,
                     float* out, float* lse, int B, int h, int nq, int nk, int dk,
                     int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                     int64_t mask_bs, int64_t mask_qs, float scale, void* stream);
int nast_mha_bwd_f32(const float* q, const float* k, const float* v, const uint8_t* mask,
                     const float* out, const float* dout, const float* lse,
                     float* dq, float* dk_, float* dv, int B, int h, int nq, int nk, int dk,
                     int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                     int64_t mask_bs, int64_t mask_qs, float scale, void* stream);

/* Packed ("varlen") rows -- no counterpart in the reference, which pads every sentence to the batch width and masks
 * (transformer.py:57-63, :296-303): sentence b owns rows q_off[b] .. q_off[b]+q_len[b]-1 of q / out / dq and rows
 * k_off[b] .. +k_len[b]-1 of k / v / dk / dv (row pitch ld*, heads side by side); every key of a sentence is visible to
 * every query of it, i.e. exactly what the padding mask leaves.  max_nq, max_nk bound the lengths: <= 64 for any head size, <= 144 for head sizes 16 / 32 / 64.
 * lse: [B, h, max_nq].  Used by Encoder/Decoder when they run on packed rows. */
int nast_mha_fwd_varlen_f32(const float* q, const float* k, const float* v, float* out, float* lse, int B, int h,
                            int max_nq, int max_nk, int dk, int ldq, int ldk, int ldv, int ldo,
                            const int32_t* q_off, const int32_t* k_off, const int32_t* q_len, const int32_t* k_len,
                            int causal /* != 0: key j is visible to queries i >= j only (transformer.py:463-464) */,
                            int q_total, int q_alloc, int k_total, int k_alloc /* rows [total, alloc) of the outputs are
                            buffer padding: written as zeros (alloc == total: none) */,
                            float scale, void* stream);
int nast_mha_bwd_varlen_f32(const float* q, const float* k, const float* v,
                            const float* out /* the forward result (row pitch ldo) or NULL; given it, delta = rowsum(dO * O)
                            is read off the rows and dQ and dK / dV are computed side by side (<= 64 positions) */,
                            const float* dout, const float* lse,
                            float* dq, float* dk, float* dv, int B, int h, int max_nq, int max_nk, int dk_head,
                            int ldq, int ldk, int ldv, int ldo,
                            const int32_t* q_off, const int32_t* k_off, const int32_t* q_len, const int32_t* k_len,
                            int causal, int q_total, int q_alloc, int k_total, int k_alloc, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row kernels over the vocabulary axis (HBM-bound): one workgroup per row.
 * Row addressing for x / y:  row r lives at  base + (r / grp) * grp_stride + (r % grp) * ld   (grp = 0:
 * base + r * ld), so sliced views such as log_probs[:, 1:] need no copy.
 * ---------------------------------------------------------------------------------------------- */
typedef struct nast_rows {
  int64_t ld;          /* elements between consecutive rows inside a group */
  int64_t grp;         /* rows per group (0 = flat)                        */
  int64_t grp_stride;  /* elements between groups                          */
} nast_rows;

/* y = log_softmax(x * inv_temp).  Replaces Generator.forward, transformer.py:213-214. In place allowed. */
int nast_log_softmax_fwd_f32(const float* x, nast_rows xr, float* y, nast_rows yr, int rows, int V,
                             float inv_temp, void* stream);
/* gx = (gy - exp(y) * sum(gy)) * inv_temp */
int nast_log_softmax_bwd_f32(const float* gy, nast_rows gyr, const float* y, nast_rows yr,
                             float* gx, nast_rows gxr, int rows, int V, float inv_temp, void* stream);
/* NLL of the gold token on log-probabilities and its backward fused with the log-softmax backward
 * (nn.NLLLoss(reduction='none') at seq2seq.py:453, :504 applied to Generator's output, transformer.py:214):
 *   nll[r] = -y[r, tokens[r]];   gx[r, v] = g[r] * (exp(y[r, v]) - [v == tokens[r]]) * inv_temp
 * tokens int64 [rows] must lie in [0, V).  The dense gradient w.r.t. the log-probabilities is never materialised. */
int nast_nll_gather_f32(const float* y, nast_rows yr, const int64_t* tokens, float* nll, int rows, int V, void* stream);
int nast_nll_log_softmax_bwd_f32(const float* g, const float* y, nast_rows yr, const int64_t* tokens, float* gx,
                                 nast_rows gxr, int rows, int V, float inv_temp, void* stream);
/* The same pair WITHOUT the [rows, V] log-probabilities (Generator + NLLLoss, transformer.py:213-214 +
 * seq2seq.py:453, :504): the forward reads a row of logits once and keeps two floats of it,
 *   lse[r] = log sum_v exp(x[r, v] * inv_temp);   nll[r] = lse[r] - x[r, tokens[r]] * inv_temp   (nll, tokens may be NULL)
 * the backward rebuilds the probabilities from the logits:
 *   gx[r, v] = g[r] * inv_temp * (exp(x[r, v] * inv_temp - lse[r]) - [v == tokens[r]])            (gx may alias x) */
int nast_lse_nll_f32(const float* x, nast_rows xr, const int64_t* tokens, float* lse, float* nll, int rows, int V,
                     float inv_temp, void* stream);
int nast_nll_logits_bwd_f32(const float* g, const float* x, nast_rows xr, const float* lse, const int64_t* tokens, float* gx,
                            nast_rows gxr, int rows, int V, float inv_temp, void* stream);
/* Gumbel-softmax sample, utils/gumbel.py:9-69:
 *   g = min(-log(E), 1e10), z = (logits + g) * inv_tau, (mask==0 -> z = -1e9), y = softmax(z), idx = argmax(y)
 * E: the Exp(1) draw (injected noise, parity mode) or NULL -> in-kernel Philox4x32-10 keyed by (seed, offset).
 * mask: int64 [rows, V] or NULL (gumbel_max_with_mask).  idx: int64 [rows] or NULL.
 * noise_off==1 disables the noise (straight_max: softmax(logits)).                                   */
int nast_gumbel_softmax_fwd_f32(const float* logits, nast_rows lr, const float* E, nast_rows er,
                                const int64_t* mask, float* y, nast_rows yr, int64_t* idx,
                                int rows, int V, float inv_tau, uint64_t seed, uint64_t offset,
                                int noise_off, void* stream);
/* gx = y * (gy - sum(gy * y)) * inv_tau   (softmax Jacobian) */
int nast_softmax_bwd_f32(const float* gy, nast_rows gyr, const float* y, nast_rows yr,
                         float* gx, nast_rows gxr, int rows, int V, float inv_tau, void* stream);
/* Seq2seq.get_lengths, seq2seq.py:545-549: first eos index + 1, capped at L. tokens int64 [B, L]. */
int nast_first_eos_length_i64(const int64_t* tokens, int64_t* lengths, int B, int L, int64_t eos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding assembly: out[b, e + i, :] = tok(b, i) + pos[i],  out[b, 0..e-1, :] = extra rows.
 * Replaces: torch.cat((style_emb, embed(tokens, pos)), 1) -- transformer.py:63-65, :141-148, :225-231.
 *   tok(b,i) = tok_weight[ids[b,i]]  (ids != NULL)   or   tok_rows[b*L + i]  (soft tokens: GEMM output).
 *   extra row j (j < n_extra <= 2) = extra_w[j][ extra_idx[j] ? extra_idx[j][b] : 0 ].
 * Backward ACCUMULATES with fp32 atomics into d_tok_weight / d_pos / d_extra_w (caller zero-inits or
 * passes .grad) and writes d_tok_rows [B*L, d] densely when soft tokens were used.
 * ---------------------------------------------------------------------------------------------- */
int nast_embed_fwd_f32(float* out, int B, int L, int d, int n_extra,
                       const int64_t* ids, const float* tok_weight, const float* tok_rows,
                       const float* pos_weight,
                       const float* extra_w0, const int64_t* extra_idx0,
                       const float* extra_w1, const int64_t* extra_idx1, void* stream);
int nast_embed_bwd_f32(const float* dout, int B, int L, int d, int n_extra,
                       const int64_t* ids, float* d_tok_weight, float* d_tok_rows,
                       float* d_pos_weight,
                       float* d_extra_w0, const int64_t* extra_idx0,
                       float* d_extra_w1, const int64_t* extra_idx1, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Learnable alignment (LengthPredictor.match, transformer.py:394-449).
 * ---------------------------------------------------------------------------------------------- */
/* cost0[b,i] = |t_i|^2 + 1e-6 ; cost[b,i,s] = (sqrt(sum_c (t_i[c]-o_s[c])^2))^2   (transformer.py:411-412)
 *   t_i = E[tgt_ids[b,i]];
 *   o_s = E[src_ids[b,s]] with unk -> id 0 (hard source, :396-399)            when src_ids != NULL
 *       = src_emb[b,s] + src_unk_w[(b*S+s)*src_unk_stride] * (E[0] - E[unk])   (soft source, :401-403)   */
int nast_align_cost_f32(const float* E, int V, int d, const int64_t* tgt_ids, const int64_t* src_ids,
                        const float* src_emb, const float* src_unk_w, int64_t src_unk_stride,
                        int64_t unk_id, float* cost0, float* cost, int B, int T, int S, void* stream);
/* Min-plus Viterbi DP over monotone alignments with skips {0,1,2,3} + backtrack; one wavefront per
 * sentence, lane = source position, __shfl_up for f[s-1..s-3].  Bit-exact w.r.t. the reference given the
 * same cost/cost0.  allpos int64 [B, T+1] (leading 0 for the style token, positions shifted by +1).
 * Limits: S <= 256, T <= 256.                                                                          */
int nast_align_viterbi_f32(const float* cost0, const float* cost, const int64_t* origin_length,
                           const int64_t* target_length, int64_t* allpos, int B, int T, int S, void* stream);
/* Extension with NO counterpart in the reference ("parity unpinned"): soft (log-sum-exp) version of the
 * same recurrence, value = -tau * log sum_paths exp(-cost(path)/tau), forward + backward (posterior
 * edge marginals = d value / d cost).  tau -> 0 recovers the Viterbi value.                            */
int nast_align_lse_fwd_f32(const float* cost0, const float* cost, const int64_t* origin_length,
                           const int64_t* target_length, float tau, float* value, float* alpha,
                           int B, int T, int S, void* stream);
int nast_align_lse_bwd_f32(const float* cost0, const float* cost, const int64_t* origin_length,
                           const int64_t* target_length, float tau, const float* value, const float* alpha,
                           const float* dvalue, float* dcost0, float* dcost, int B, int T, int S, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Packed-row layout (no counterpart in the reference, which runs every sentence at the padded width): index maps of a
 * ragged batch in one launch.  lengths int64 [B] (negative counts as 0), width = padded positions per sentence, total =
 * sum(lengths) as the HOST knows it, rows >= total = allocated packed rows.  Outputs: off int64 [B] / off32 int32 [B]
 * (exclusive scan), len32 int32 [B], and for packed row r of sentence b at position p: b_of[r] = b, pos[r] = p,
 * idx[r] = b * width + p (all int64 [total]); idx_p int64 [rows] = idx followed by zeros.  *bad (device int, sticky: the
 * caller zeroes it once and may share it between calls) is set to 1 and the row maps are zero-filled (in-bounds,
 * meaningless) when total != sum(lengths).  Limit: B <= 8192.                     */
int nast_pack_index_i64(const int64_t* lengths, int B, int width, int total, int rows, int64_t* off, int32_t* off32,
                        int32_t* len32, int64_t* b_of, int64_t* pos, int64_t* idx, int64_t* idx_p, int* bad, void* stream);

/* Key-padding mask of a ragged batch, one launch: mask[b][j] = j < extra ? 0 : (j - extra >= lengths[b]) for j < extra + width,
 * one byte per element (1 = masked).  Replaces the arange / compare / zeros / cat of transformer.py:56-61, :135-139, :459-461
 * (the reference builds the same mask from the same lengths).  lengths int64 [B].                                     */
int nast_pad_mask_u8(const int64_t* lengths, uint8_t* mask, int B, int width, int extra, void* stream);

/* Rows between the padded [B * width, d] layout and the packed one (ops.pack_rows / ops.unpack_rows and their backward
 * passes) -- no counterpart in the reference (see nast_pack_index_i64).  Both are gathers, one launch, no zero-fill:
 *   gather:  y[r] = x[idx[r]] for r < n_idx, zeros for n_idx <= r < rows_out   (x, y: row pitches ldx, ldy >= d)
 *   unpack:  y[b * width + p] = p < len[b] ? xp[off[b] + p] : zeros            (off int64 [B], len int32 [B]; xp has rows_in
 *            rows, a source row beyond them reads as zeros) */
int nast_rows_gather_f32(const float* x, int64_t ldx, const int64_t* idx, int n_idx, int rows_out, int d, float* y, int64_t ldy,
                         void* stream);
int nast_rows_unpack_f32(const float* xp, int64_t ldx, int rows_in, const int64_t* off, const int32_t* len, int B, int width, int d,
                         float* y, int64_t ldy, void* stream);


/* ------------------------------------------------------------------------------------------------
 * NoiseLayer on the device (noise.py:60-194): blank -> add -> dropout/repeat of
 * NoiseLayer.forward (:60-74) for a whole [S, B] time-major batch of ids, one lane per sentence.  Each stage is
 * skipped when its uniform-draw pointer is NULL.  u_blank [S,B], u_add [M-1,B], new_words int64 [M,B] (the
 * torch.multinomial candidates, :83), u_keep [M,B], u_rep [M-1,B], u_pick [B] (index = floor(u*n) when every word
 * of a sentence was dropped, :150-152): float64 uniforms in [0,1).  out int64 [M+1, B] (rows past a sentence's new
 * length hold pad after the dropout stage), out_len int64 [B].  M = max_length.  Integer work: bit-exact with the
 * reference's per-sentence loops given the same draws.  Limits: M + 1 <= 160, S <= 160.                          */
int nast_noise_i64(const int64_t* x, const int64_t* len, int S, int B, int M,
                   const double* u_blank, double blank_prob,
                   const double* u_add, const int64_t* new_words, double add_prob,
                   const double* u_keep, const double* u_rep, const double* u_pick, double dropout_prob, double repeat_prob,
                   int64_t pad, int64_t unk, int64_t eos, int keep_go, int keep_eos,
                   int64_t* out, int64_t* out_len, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LengthPredictor glue (transformer.py:466-519 teacher-forced, :527-582 greedy decode).
 * ---------------------------------------------------------------------------------------------- */
/* cand(b,p) = origin_input[b,p] if p == 0 or (p < S1 and p-1 < len[b]) else eos            (:471-473)
 * out[b,t,:] = (use_stay && allpos[b,t]==allpos[b,t-1]) ? unk : cand(b, allpos[b,t] + shift)   (:488-491, :504-505)
 *              (+ pos_weight[t] when pos_weight != NULL, :493).  origin_input [B,S1,d], allpos int64 [B,T1].   */
int nast_lp_gather_fwd_f32(const float* origin_input, const int64_t* origin_length, const int64_t* allpos,
                           const float* eos, const float* unk, const float* pos_weight,
                           float* out, int B, int S1, int T1, int d, int shift, int use_stay, void* stream);
/* scatter-add (fp32 atomics) of d(out) into d_origin_input [B,S1,d] (may be NULL), d_eos [d], d_unk [d],
 * d_pos [T1,d] (may be NULL; pos_skip_row0: row 0 gets no gradient, nn.Embedding padding_idx=0, :385). */
int nast_lp_gather_bwd_f32(const float* dout, const int64_t* origin_length, const int64_t* allpos,
                           float* d_origin_input, float* d_eos, float* d_unk, float* d_pos,
                           int B, int S1, int T1, int d, int shift, int use_stay, int pos_skip_row0, void* stream);

/* One greedy skip-decoding step (transformer.py:549-567, z_decode_mode == "max"), one wavefront per sentence:
 *   logit[0] = qh . kunk / sqrt(d) + lin(hid)[0];  logit[j] = qh . KC[b, now+j] / sqrt(d) + lin(hid)[j], j=1..3
 *   step 0: logit[0] = -1e9 and logit[j] -= 1e9 where source position j is padding (:553-555)
 *   delta = first argmax; now = min(now + delta, S1); allpos[b, step+1] = now;
 *   x_next[b] = (delta == 0 ? unk : cand_in[b, now]) + pos_w[step+1]      (next decoder input, :535-540)
 * qh = wq(hidden), hid = hidden [B,d]; KC = wk(cand_in) [B,S1+5,d]; kunk = wk(unk) [d]; cand_in [B,S1+5,d];
 * lin_w [4,d], lin_b [4]; now_pos int64 [B] (in/out); allpos int64 [B, ldp].
 * norm_w/norm_b (may be NULL): when given, `hid` is the un-normalised decoder output and the final
 * LayerNorm of HiddenDecoder.incremental_forward (transformer.py:376) is applied inside the kernel.
 * sample_u (may be NULL): NULL = z_decode_mode "max" (first maximum, :559).  Otherwise [B] uniform numbers in [0,1) for
 * this step and the skip is DRAWN from softmax(logits) (z_decode_mode "sample", :560-561) by inverse CDF:
 * delta = the first j with p_0 + ... + p_j > u (the last j with p_j > 0 if rounding leaves none).            */
int nast_lp_step_f32(const float* qh, const float* hid, const float* KC, const float* kunk,
                     const float* lin_w, const float* lin_b, const float* cand_in, const float* unk,
                     const float* pos_w, const int64_t* origin_length, int64_t* now_pos, int64_t* allpos,
                     float* x_next, const float* norm_w, const float* norm_b, const float* sample_u,
                     int B, int S1, int d, int step, int ldp, int write_next, void* stream);

/* Skip logits of the teacher-forced branch (transformer.py:504-507, the einsum "bld,bltd->blt" over
 * wk(candidates_output) gathered by allpos):  out[b,t,0] = qh[b,t].kunk / sqrt(d),
 * out[b,t,j] = qh[b,t].KC[b, allpos[b,t]+j] / sqrt(d), j = 1..3.   qh [B,T1,d], KC = wk(candidates_input) [B,NC,d]
 * (NC >= max(allpos)+4), kunk = wk(unk) [d], out / g [B,T1,4].  Backward writes dqh and ACCUMULATES (atomics) into
 * dKC [B,NC,d] and dkunk [d] (caller zero-initialises).                                                        */
int nast_lp_logits_fwd_f32(const float* qh, const float* KC, const float* kunk, const int64_t* allpos, float* out,
                           int B, int T1, int NC, int d, void* stream);
int nast_lp_logits_bwd_f32(const float* g, const float* qh, const float* KC, const float* kunk, const int64_t* allpos,
                           float* dqh, float* dKC, float* dkunk, int B, int T1, int NC, int d, void* stream);

/* Whole greedy skip-decoding loop of LengthPredictor.forward (generate branch, transformer.py:525-570) run from
 * native code: candidate table, wk(candidates), cross-attention K/V of all layers, then per step and layer
 * [LN+QKV GEMM] -> attention over the K/V cache -> [out GEMM + residual] -> [LN+q GEMM] -> cross attention ->
 * [out GEMM] -> [LN+FFN1] -> FFN2, the skip decision kernel, and the reference's early exit (every 5th step,
 * :569 -- this is the one place the library synchronises the stream: it copies a "finished" count to the host).
 * All pointers in the structs are device pointers; *steps_out (host) receives the number of executed steps.   */
typedef struct nast_declayer_weights {
  const float *ln0_w, *ln0_b, *sa_wq, *sa_bq, *sa_wk, *sa_bk, *sa_wv, *sa_bv, *sa_wo, *sa_bo;
  const float *ln1_w, *ln1_b, *ca_wq, *ca_bq, *ca_wk, *ca_bk, *ca_wv, *ca_bv, *ca_wo, *ca_bo;
  const float *ln2_w, *ln2_b, *ff_w1, *ff_b1, *ff_w2, *ff_b2;
} nast_declayer_weights;

typedef struct nast_lp_decode_args {
  int32_t B, S1, d, h, dff, L, nlayers;
  const nast_declayer_weights* layers;   /* HOST array of nlayers entries */
  const float *norm_w, *norm_b;          /* HiddenDecoder.norm */
  const float *wq_w, *wq_b, *wk_w, *wk_b, *lin_w, *lin_b;
  const float* pos_w;                    /* length_predictor.pos_embed.weight [L+1,d] */
  const float *eos, *unk;                /* token_embed rows [d] */
  const float* memory;                   /* origin_input [B,S1,d] (encoder memory, style token first) */
  const int64_t* origin_length;          /* [B] */
  int64_t* allpos;                       /* [B, L+1] out; column 0 = 0 */
  float* workspace;                      /* nast_lp_decode_workspace_bytes() bytes, 256-B aligned */
  const float* sample_u;                 /* NULL: z_decode_mode "max".  Else [L,B] uniforms in [0,1): step k draws its
                                            skips from softmax(logits) with row k (z_decode_mode "sample", :560-561) */
} nast_lp_decode_args;

int64_t nast_lp_decode_workspace_bytes(int B, int S1, int d, int dff, int L, int nlayers);
/* steps_out: number of decode steps taken (allpos columns 1..steps are valid).  On entry *steps_out >= 0: the reference's
 * early exit (every 5th step the progress is read back, transformer.py:569 -- the call synchronises the stream);
 * *steps_out < 0: no read-back, all L steps are enqueued and the call returns at once (steps = L). */
int nast_lp_decode_f32(const nast_lp_decode_args* args, int* steps_out, void* stream);

/* Allocation-free variant of nast_lp_decode_f32.  NO counterpart in the reference: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * arguments and results, but args->workspace is IGNORED (may be NULL): the workspace lives in a device array built into the
 * library, NAST_LP_DECODE_STATIC_BYTES per GPU (each GPU the library runs on holds its copy until the process ends).
 * A decode that needs more (nast_lp_decode_workspace_bytes() > NAST_LP_DECODE_STATIC_BYTES) is rejected with NAST_EINVAL and
 * a message; 64 MiB covers 128 sentences at S1 = 33, d = 256, L = 32, 3 layers.  The call makes no hipMalloc / hipFree and no
 * other allocation.  Every call on a GPU shares that GPU's buffer (it keeps the last call's contents; every word a
 * kernel reads is rewritten by the call itself first): calls that overlap on one GPU -- two streams, two threads -- RACE;
 * serialise them, or use nast_lp_decode_f32 with one workspace per concurrent call.
 * Under HIP graph capture pass *steps_out < 0: with *steps_out >= 0 the call reads the progress back (hipMemcpyAsync to
 * host memory + hipStreamSynchronize), which a capturing stream does not allow -- the call then returns NAST_EINVAL. */
#define NAST_LP_DECODE_STATIC_BYTES (64u << 20)
int nast_lp_decode_static_f32(const nast_lp_decode_args* args, int* steps_out, void* stream);

/* nast_lp_decode_f32 runs eligible shapes (d = 256, 4 heads, d_ff = 1024, <= 4 layers, L and S1 <= 36, <= 2048 sentences;
 * NAST_DECODE_CLUSTER=0 turns it off) as ONE persistent launch of 8 clusters x 32 workgroups with cluster barriers between
 * the operations (csrc/decode_cluster.hip; same loop as transformer.py:525-570).  A barrier whose poll runs out (1 s: some
 * workgroup of the grid was not resident, e.g. under a CU mask) raises a sticky flag on the device instead of hanging; the
 * positions of that call are then garbage.  *status_out = 0: no such event on the current device since the library was
 * loaded, 1: there was one.  Synchronous (reads one device word): call it where the caller synchronises anyway. */
int nast_lp_decode_cluster_status(int* status_out);

/* ------------------------------------------------------------------------------------------------
 * Optimiser side (seq2seq.py:536-537, :637-638): multi-tensor L2 norm + clip + Adam in few launches.
 * ptrs are device arrays of device pointers (one entry per parameter tensor), sizes int64 element counts.
 * ---------------------------------------------------------------------------------------------- */
int nast_multi_sqnorm_f32(const float* const* grads, const int64_t* sizes, int n, float* out_sq /*[1], +=*/,
                          const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_elems,
                          void* stream);
int nast_multi_adam_f32(float* const* params, float* const* grads /* scaled in place by the clip */, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* sizes, int n,
                        const float* total_sqnorm /*device [1] or NULL*/, float max_norm,
                        float lr, float beta1, float beta2, float eps, float weight_decay,
                        float bias_c1, float bias_c2,
                        const float* bias_dev /* device [2] = {bias_c1, bias_c2} read by the kernel instead of the two
                                                 scalars, or NULL: for a replayed HIP graph, whose step count is on the device */,
                        const int32_t* chunk_tensor, const int64_t* chunk_off, int n_chunks, int chunk_elems,
                        void* stream);
/* chunk table (device arrays, or NULL for a (32, n) grid): chunk c covers elements chunk_off[c] .. +chunk_elems of tensor
 * chunk_tensor[c]; one workgroup per chunk, so tensors of very different sizes load the chip evenly. */

/* ------------------------------------------------------------------------------------------------
 * In-library kernel timing (HIP events on the caller's stream) for bench.py's roofline block.
 * nast_prof_begin(kind) arms timing of every subsequent launch of that kernel family (kind: 1 = tiled GEMM kernel,
 * work = FLOP; 2 = vocabulary row kernels, work = algorithmic bytes; 3 = alignment DP, work = bytes; 4 = small-M GEMM
 * kernel of the greedy decode, work = FLOP; 5 = the log-sum-exp alignment extension, work = bytes; 6 = the cluster decode
 * kernel (one launch per decode), work = FLOP of its projections; 0 = all of them);
 * nast_prof_end() synchronises the events and reports the GEMM totals, nast_prof_get(kind) any family's.
 * Not graph-capture safe while armed.
 * ---------------------------------------------------------------------------------------------- */
int nast_prof_begin(int kind);
int nast_prof_end(double* total_ms, double* total_flops, int64_t* launches);
int nast_prof_get(int kind, double* total_ms, double* total_work, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* NAST_HIP_H */
