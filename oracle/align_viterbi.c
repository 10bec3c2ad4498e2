/* Plain-C restatement of the alignment DP of LengthPredictor.match (reference:
 * styletransformer/transformer.py:413-448).  TEST INFRASTRUCTURE ONLY: used by tests/ and bench.py's
 * cpu_baseline as the checker for nast_align_viterbi_f32; never linked into the product.
 * Pinned by tests/golden/g6_match.npz (cost tables + allpos captured from the unmodified reference).
 *
 *   cost0 [B,T], cost [B,T,S] fp32, olen/tlen int64 [B]  ->  allpos int64 [B,T+1]
 */
#include <stdint.h>
#include <stdlib.h>

#define BIG 1e9f

int oracle_align_viterbi(const float* cost0, const float* cost, const int64_t* olen, const int64_t* tlen,
                         int64_t* allpos, int B, int T, int S) {
  float* f = (float*)malloc(sizeof(float) * S);
  float* g = (float*)malloc(sizeof(float) * S);
  unsigned char* back = (unsigned char*)malloc((size_t)(T > 1 ? T - 1 : 1) * S);
  int* pos = (int*)malloc(sizeof(int) * T);
  if (!f || !g || !back || !pos) return -1;
  for (int b = 0; b < B; ++b) {
    const float* c0 = cost0 + (size_t)b * T;
    const float* c = cost + (size_t)b * T * S;
    const int ol = (int)olen[b], tl = (int)tlen[b];
    for (int s = 0; s < S; ++s) {                      /* :413-415 */
      float v = s >= ol ? BIG : c[s];
      f[s] = s < 3 ? v : BIG;
    }
    for (int i = 1; i < T; ++i) {                      /* :423-431 */
      const int live = i < tl;
      for (int s = 0; s < S; ++s) {
        const float cs = s >= ol ? BIG : c[(size_t)i * S + s];
        float best = f[s] + c0[i];
        int bj = 0;
        for (int j = 1; j <= 3; ++j) {
          const float prev = s - j >= 0 ? f[s - j] : BIG;
          const float x = prev + cs;
          if (x < best) { best = x; bj = j; }          /* first minimum wins */
        }
        g[s] = live ? best : f[s];
        back[(size_t)(i - 1) * S + s] = (unsigned char)(live ? bj : 0);
      }
      float* t = f; f = g; g = t;
    }
    int last = 0;                                      /* :433-435 */
    float bestv = 0.f;
    for (int s = 0; s < S; ++s) {
      const float v = s >= ol - 3 ? f[s] : BIG;
      if (s == 0 || v < bestv) { bestv = v; last = s; }
    }
    pos[T - 1] = last;
    for (int i = T - 1; i >= 1; --i) {                 /* :439-442 */
      last -= back[(size_t)(i - 1) * S + last];
      pos[i - 1] = last;
    }
    int64_t* ap = allpos + (size_t)b * (T + 1);        /* :445-448 */
    ap[0] = 0;
    for (int i = 0; i < T; ++i) ap[i + 1] = (i >= tl ? ol : pos[i]) + 1;
  }
  free(f); free(g); free(back); free(pos);
  return 0;
}
