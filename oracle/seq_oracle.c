/* ORACLE (test infrastructure only - never linked into the product).
 * Plain C restatement of the integer algorithms on the hot path, for checking the HIP kernels at full sizes:
 *   oracle_correct_pred : model/hw_with_style.py:18-74  (banded DTW, first-minimum tie break over up/diag/left, backtrace)
 *   oracle_gt_counts    : trainer/hw_with_style_trainer.py:670-697
 * Pinned by tests/golden/seq_kat.npz (vectors recorded from the reference's own functions). Build: make -C oracle
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* pred [T][B][C] float log-probs, label [L][B] int32 -> out [T+2L+1][B] int64 (zero padded), lens[B]; returns max len */
int oracle_correct_pred(const float* pred, const int32_t* label, int T, int B, int C, int L, int64_t* out, int32_t* lens) {
  const int LL = 2 * L + 1;
  const int d = T - LL;
  const int w = (T / 2 > abs(d)) ? T / 2 : abs(d);
  float* cost = (float*)malloc(sizeof(float) * (size_t)(T + 1) * (LL + 1));
  unsigned char* hist = (unsigned char*)malloc((size_t)T * LL);
  int64_t* path = (int64_t*)malloc(sizeof(int64_t) * (size_t)(T + LL + 1));
  int maxlen = 0;
  for (int b = 0; b < B; ++b) {
    for (size_t i = 0; i < (size_t)(T + 1) * (LL + 1); ++i) cost[i] = INFINITY;
    cost[0] = 0.f;
    for (int i = 1; i <= T; ++i) {
      const int lo = (i - w > 1) ? i - w : 1, hi = (i + w < LL) ? i + w : LL;
      for (int j = lo; j <= hi; ++j) {
        const int lab = ((j - 1) & 1) ? label[((j - 1) >> 1) * B + b] : 0;
        const float c = 1.f - pred[((size_t)(i - 1) * B + b) * C + lab];
        const float up = cost[(size_t)(i - 1) * (LL + 1) + j], dg = cost[(size_t)(i - 1) * (LL + 1) + j - 1], lf = cost[(size_t)i * (LL + 1) + j - 1];
        float m = up; unsigned char h = 0;
        if (dg < m) { m = dg; h = 1; }
        if (lf < m) { m = lf; h = 2; }
        hist[(size_t)(i - 1) * LL + j - 1] = h;
        cost[(size_t)i * (LL + 1) + j] = c + m;
      }
    }
    int i = T - 1, j = LL - 1, n = 0;
    path[n++] = (j & 1) ? label[(j >> 1) * B + b] : 0;
    while ((i > 0 || j > 0) && n < T + LL) {
      const unsigned char h = hist[(size_t)i * LL + j];
      if (h == 0) i -= 1; else if (h == 1) { i -= 1; j -= 1; } else j -= 1;
      if (i < 0) i = 0;
      if (j < 0) j = 0;
      path[n++] = (j & 1) ? label[(j >> 1) * B + b] : 0;
    }
    lens[b] = n;
    if (n > maxlen) maxlen = n;
    for (int k = 0; k < T + LL; ++k) out[(size_t)k * B + b] = (k < n) ? path[n - 1 - k] : 0;
  }
  free(cost); free(hist); free(path);
  return maxlen;
}

/* index_spaced [Tp][B] int64, label [L][B] int32 -> gt [L][B][2] float (caller zero-fills); returns min over b of final pos, or -1 on mismatch */
int oracle_gt_counts(const int64_t* idx, const int32_t* label, int Tp, int B, int L, float* gt) {
  int minpos = 1 << 30;
  for (int b = 0; b < B; ++b) {
    int c = 0, d = 0, pos = 0, last = 0;
    for (int i = 0; i < Tp; ++i) {
      const int v = (int)idx[(size_t)i * B + b];
      if (v == 0 && last == 0) c += 1;
      else if (last == 0 || last == v) { d += 1; last = v; }
      else {
        if (pos >= L || label[pos * B + b] != last) return -1;
        gt[((size_t)pos * B + b) * 2] = (float)c; gt[((size_t)pos * B + b) * 2 + 1] = (float)d;
        if (v == 0) { c = 1; d = 0; } else { c = 0; d = 1; }
        pos += 1; last = v;
      }
    }
    if (pos < minpos) minpos = pos;
  }
  return minpos;
}
