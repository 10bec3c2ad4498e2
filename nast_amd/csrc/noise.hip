// Input noise of the self-reconstruction pass on the device (reference: styletransformer/noise.py:60-194, stages
// blank -> add -> dropout/repeat in that order; shuffle is not handled here).
//
// Integer work on [seq, B] token ids, a few hundred bytes per sentence: one lane per sentence walks its own tokens
// through the three stages in private memory (ids are time-major, so lane b of a wavefront reads x[i*B + b]:
// coalesced).  Every stage is the same pure function of its uniform draws as nast_amd/noise.py's `*_apply`
// (which restate the reference's per-sentence loops), so given the same draws the output is bit-exact.
#include "common.h"

#define NOISE_MAX 160   // max_length + 1 and the input width must fit

struct NoiseArgs {
  const int64_t* x; const int64_t* len; int S, B, M;
  const double* u_blank; double blank_prob;
  const double* u_add; const int64_t* new_words; double add_prob;
  const double* u_keep; const double* u_rep; const double* u_pick; double dropout_prob, repeat_prob;
  int64_t pad, unk, eos; int keep_go, keep_eos;
  int64_t* out; int64_t* out_len;
};

__global__ __launch_bounds__(64) void nast_noise_kernel(const NoiseArgs A) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= A.B) return;
  const int B = A.B, M = A.M, S = A.S;
  int64_t cur[NOISE_MAX], tmp[NOISE_MAX];
  int l = (int)A.len[b];
  l = l < 1 ? 1 : (l > S ? S : l);
  for (int i = 0; i < S; ++i) cur[i] = A.x[(int64_t)i * B + b];

  if (A.u_blank) {   // noise.py:103-135: words before the final token become <unk> where u < blank_prob
    const int protect = A.keep_eos ? l - 1 : l;
    for (int i = 0; i < S; ++i) {
      bool keep = A.u_blank[(int64_t)i * B + b] >= A.blank_prob;
      if (A.keep_go && i == 0) keep = true;
      if (i >= protect) keep = true;
      if (!keep) cur[i] = A.unk;
    }
  }

  if (A.u_add) {     // noise.py:76-101: slot i shows source word (#source slots so far - 1) or an inserted word
    unsigned long long orig_bits[(NOISE_MAX + 63) / 64] = {0};
    int c = 0, lnew = 0;
    for (int i = 0; i < M; ++i) {
      const bool orig = (i == 0) || A.u_add[(int64_t)(i - 1) * B + b] >= A.add_prob;
      if (orig) orig_bits[i >> 6] |= 1ull << (i & 63);
      c += orig ? 1 : 0;
      int src = c - 1;
      const int fit = i - M + l;              // the tail must still fit
      src = src > fit ? src : fit;
      const bool past = src >= l;
      if (!past) ++lnew;
      const int s2 = src < l - 1 ? src : l - 1;
      int64_t v = cur[s2];
      if (!orig) v = A.new_words[(int64_t)i * B + b];
      if (past) v = A.eos;
      tmp[i] = v;
    }
    const int last = lnew - 1 > 0 ? lnew - 1 : 0;
    const bool last_inserted = !((orig_bits[last >> 6] >> (last & 63)) & 1ull);
    if (lnew > 1 && last_inserted) --lnew;    // never end on an inserted word
    for (int i = 0; i < M; ++i) cur[i] = tmp[i];
    l = lnew;
  }

  if (A.u_keep) {    // noise.py:137-194: drop / repeat words, re-append <eos>, re-pad
    const int n = (cur[l - 1] == A.eos) ? l - 1 : l;   // words before the final <eos>
    const int nn = n < M ? n : M;
    int ksz = 0;
    for (int i = 0; i < nn; ++i)
      if (A.u_keep[(int64_t)i * B + b] >= A.dropout_prob) tmp[ksz++] = cur[i];
    if (ksz == 0) {                                    // every word dropped: keep one
      int p = (int)(A.u_pick[b] * (double)n);
      p = p < 0 ? 0 : (p > n - 1 ? n - 1 : p);
      p = p < 0 ? 0 : p;
      tmp[0] = cur[p];
      ksz = 1;
    }
    int adv = 0, cnt = 0;
    for (int i = 0; i < M; ++i) {
      if (i > 0) adv += (A.u_rep[(int64_t)(i - 1) * B + b] >= A.repeat_prob) ? 1 : 0;
      const int fit = i - M + ksz;
      const int idx = adv > fit ? adv : fit;
      if (idx < ksz) ++cnt;
      cur[i] = tmp[idx < ksz - 1 ? idx : ksz - 1];
    }
    for (int i = cnt; i <= M; ++i) cur[i] = A.pad;
    int l2 = cnt;
    if (cnt < M && cur[cnt - 1] != A.eos) {
      cur[cnt] = A.eos;
      l2 = cnt + 1;
    }
    l = l2;
    for (int i = 0; i <= M; ++i) A.out[(int64_t)i * B + b] = cur[i];
  } else {
    // no dropout stage: rows as the earlier stages left them (the caller cuts the width at max length)
    const int W = A.u_add ? M : S;
    for (int i = 0; i <= M; ++i) A.out[(int64_t)i * B + b] = i < W ? cur[i] : A.pad;
  }
  A.out_len[b] = l;
}

extern "C" int nast_noise_i64(const int64_t* x, const int64_t* len, int S, int B, int M,
                              const double* u_blank, double blank_prob,
                              const double* u_add, const int64_t* new_words, double add_prob,
                              const double* u_keep, const double* u_rep, const double* u_pick, double dropout_prob, double repeat_prob,
                              int64_t pad, int64_t unk, int64_t eos, int keep_go, int keep_eos,
                              int64_t* out, int64_t* out_len, void* stream) {
  NAST_REQUIRE(B >= 0 && S >= 1 && M >= 1, "nast_noise_i64: bad sizes B=%d S=%d M=%d", B, S, M);
  NAST_REQUIRE(M + 1 <= NOISE_MAX && S <= NOISE_MAX, "nast_noise_i64: max_length %d / width %d beyond %d", M, S, NOISE_MAX - 1);
  NAST_REQUIRE(x && len && out && out_len, "nast_noise_i64: NULL pointer");
  NAST_REQUIRE(!u_add || new_words, "nast_noise_i64: the add stage needs new_words");
  NAST_REQUIRE(!u_keep || (u_rep && u_pick), "nast_noise_i64: the dropout stage needs u_rep and u_pick");
  NAST_REQUIRE(!u_keep || (!keep_go && keep_eos), "nast_noise_i64: dropout needs keep_go=0, keep_eos=1 (noise.py:143)");
  if (B == 0) return NAST_OK;
  NoiseArgs A{x, len, S, B, M, u_blank, blank_prob, u_add, new_words, add_prob, u_keep, u_rep, u_pick, dropout_prob, repeat_prob,
              pad, unk, eos, keep_go, keep_eos, out, out_len};
  hipLaunchKernelGGL(nast_noise_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, A);
  NAST_CHECK_LAUNCH("nast_noise_i64");
  return NAST_OK;
}
