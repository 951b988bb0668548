// Sequence-side kernels: log-softmax, CTC loss (alpha/beta in log space), the DTW label alignment
// `correct_pred`, and the run-length "gt counts" scan. Integer outputs are bit-exact restatements of the
// reference's host loops (model/hw_with_style.py:18-74, trainer/hw_with_style_trainer.py:670-697).
#include "hwg_common.h"

namespace {

// ---------------- log-softmax over the channel dim; optional [B][T] -> [T][B] row transpose on output ----------------
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* x, float* y, long long rows, int C, int Bn, int Tn, int transpose) {
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * 256LL + threadIdx.x) >> 6;
  const long long nw = (gridDim.x * 256LL) >> 6;
  for (long long r = wave; r < rows; r += nw) {
    const float* xr = x + r * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, xr[c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(xr[c] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    long long orow = r;
    if (transpose) { const long long b = r / Tn, t = r % Tn; orow = t * Bn + b; }
    float* yr = y + orow * C;
    for (int c = lane; c < C; c += 64) yr[c] = xr[c] - lse;
  }
}
// dx = dy - exp(y) * sum(dy); dy,y are in the (possibly transposed) output layout, dx in the input layout
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* dy, const float* y, float* dx, long long rows, int C, int Bn, int Tn,
                                                              int transpose) {
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * 256LL + threadIdx.x) >> 6;
  const long long nw = (gridDim.x * 256LL) >> 6;
  for (long long r = wave; r < rows; r += nw) {
    long long orow = r;
    if (transpose) { const long long b = r / Tn, t = r % Tn; orow = t * Bn + b; }
    const float* dyr = dy + orow * C;
    const float* yr = y + orow * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += dyr[c];
    s = wave_sum(s);
    float* dxr = dx + r * C;
    for (int c = lane; c < C; c += 64) dxr[c] = dyr[c] - expf(yr[c]) * s;
  }
}

// ---------------- CTC (blank = 0), same recursions as ATen's ctc_loss_cpu ----------------
__device__ __forceinline__ int ctc_ext(const int* tg, int s) { return (s & 1) ? tg[s >> 1] : 0; }

// one block per batch item. lp: [T][B][C]; targets: [B][Lmax] int32; writes log_alpha [B][T][2*Lmax+1] and nll[B]. The recursion's rows
// live in LDS (a chain of T dependent steps: one barrier each, no global-memory round trip); global log_alpha is written for the backward pass
__global__ __launch_bounds__(256) void ctc_alpha_kernel(const float* lp, const int* targets, const int* in_len, const int* tg_len, int T, int B,
                                                        int C, int Lmax, float* log_alpha, float* nll) {
  extern __shared__ float rows[];      // [2][NSmax] alpha rows, then NSmax ints: extended target
  const int b = blockIdx.x;
  const int S = tg_len[b];
  const int Tb = in_len[b];
  const int NS = 2 * S + 1;
  const int NSmax = 2 * Lmax + 1;
  const int* tg = targets + (long long)b * Lmax;
  float* la = log_alpha + (long long)b * T * NSmax;
  int* ext = (int*)(rows + 2 * NSmax);
  const float NEG = -INFINITY;
  for (int s = threadIdx.x; s < NSmax; s += 256) {
    float v = NEG;
    if (Tb > 0) {
      if (s == 0) v = lp[((long long)0 * B + b) * C + 0];
      else if (s == 1 && S > 0) v = lp[((long long)0 * B + b) * C + tg[0]];
    }
    la[s] = v;
    rows[s] = v;
    ext[s] = s < NS ? ctc_ext(tg, s) : 0;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* prev = rows + ((t - 1) & 1) * NSmax;
    float* cur = rows + (t & 1) * NSmax;
    for (int s = threadIdx.x; s < NSmax; s += 256) {
      float v = NEG;
      if (s < NS) {
        const int cs = ext[s];
        const float l = lp[((long long)t * B + b) * C + cs];
        const float la1 = prev[s];
        float lamax = la1;
        float la2 = NEG, la3 = NEG;
        if (s > 0) { la2 = prev[s - 1]; if (la2 > lamax) lamax = la2; }
        if (s > 1 && ext[s - 2] != cs) { la3 = prev[s - 2]; if (la3 > lamax) lamax = la3; }
        if (lamax == NEG) lamax = 0.f;
        v = logf(expf(la1 - lamax) + expf(la2 - lamax) + expf(la3 - lamax)) + lamax + l;
      }
      cur[s] = v;
      la[(long long)t * NSmax + s] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float r;
    if (Tb <= 0) r = (S == 0) ? 0.f : INFINITY;
    else {
      const float* last = rows + ((Tb - 1) & 1) * NSmax;
      const float l1 = last[NS - 1];
      const float l2 = (S > 0) ? last[NS - 2] : NEG;
      float m = fmaxf(l1, l2);
      if (m == NEG) m = 0.f;
      r = -(logf(expf(l1 - m) + expf(l2 - m)) + m);
    }
    nll[b] = r;
  }
}
// loss = mean_b( nll_b / max(tg_len_b, 1) ); a non-finite mean is reported as 0 (model/loss.py:28-30)
__global__ void ctc_mean_kernel(const float* nll, const int* tg_len, int B, float* loss, int* finite_flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) { const int l = tg_len[b] > 1 ? tg_len[b] : 1; s += nll[b] / (float)l; }
    s /= (float)B;
    const int inf = isinf(s) ? 1 : 0;
    *finite_flag = inf ? 0 : 1;
    *loss = inf ? 0.f : s;
  }
}
// backward, part 1: beta recursion (eq. 10-11 of Graves et al., as ATen's ctc_loss_backward_cpu), one block per batch item, rows kept in LDS
// (the recursion is a chain of T dependent steps: a global-memory round trip per step would cost more than the arithmetic); log_beta [B][T][NSmax]
__global__ __launch_bounds__(256) void ctc_beta_kernel(const float* lp, const int* targets, const int* in_len, const int* tg_len, int T, int B,
                                                       int C, int Lmax, float* log_beta) {
  extern __shared__ float rows[];      // [2][NSmax] beta rows, then NSmax ints: extended target
  const int b = blockIdx.x;
  const int S = tg_len[b];
  const int Tb = in_len[b];
  const int NS = 2 * S + 1;
  const int NSmax = 2 * Lmax + 1;
  const int* tg = targets + (long long)b * Lmax;
  float* lb = log_beta + (long long)b * T * NSmax;
  int* ext = (int*)(rows + 2 * NSmax);
  const float NEG = -INFINITY;
  for (int s = threadIdx.x; s < NSmax; s += 256) ext[s] = s < NS ? ctc_ext(tg, s) : 0;
  __syncthreads();
  if (Tb <= 0) return;
  for (int t = Tb - 1; t >= 0; --t) {
    float* cur = rows + (t & 1) * NSmax;
    const float* nxt = rows + ((t + 1) & 1) * NSmax;
    for (int s = threadIdx.x; s < NSmax; s += 256) {
      float v = NEG;
      if (s < NS) {
        const int cs = ext[s];
        const float l = lp[((long long)t * B + b) * C + cs];
        if (t == Tb - 1) {
          if (s == NS - 1 || (S > 0 && s == NS - 2)) v = l;
        } else {
          const float lb1 = nxt[s];
          float lbmax = lb1;
          float lb2 = NEG, lb3 = NEG;
          if (s < NS - 1) { lb2 = nxt[s + 1]; if (lb2 > lbmax) lbmax = lb2; }
          if (s < NS - 2 && ext[s + 2] != cs) { lb3 = nxt[s + 2]; if (lb3 > lbmax) lbmax = lb3; }
          if (lbmax == NEG) lbmax = 0.f;
          v = logf(expf(lb1 - lbmax) + expf(lb2 - lbmax) + expf(lb3 - lbmax)) + lbmax + l;
        }
      }
      cur[s] = v;
      lb[(long long)t * NSmax + s] = v;
    }
    __syncthreads();
  }
}

// backward, part 2: gradient wrt log-probs (eq. 16), one block per (t, b) - every (t, b, class) is independent once alpha and beta exist.
// The log-sum of alpha*beta per class walks the extended target in state order: a fixed summation order (atomics would make the gradient
// differ in its last bits from run to run, which Adam's sign-like first steps amplify).
__global__ __launch_bounds__(128) void ctc_grad_kernel(const float* lp, const int* targets, const int* in_len, const int* tg_len, int T, int B, int C,
                                                       int Lmax, const float* log_alpha, const float* log_beta, const float* nll, const float* grad_out,
                                                       const int* finite_flag, float* grad) {
  extern __shared__ float ab[];        // [NSmax] alpha + beta, then NSmax ints: extended target
  const int t = blockIdx.x, b = blockIdx.y;
  const int S = tg_len[b];
  const int Tb = in_len[b];
  const int NS = 2 * S + 1;
  const int NSmax = 2 * Lmax + 1;
  float* g = grad + ((long long)t * B + b) * C;
  if (t >= Tb) {                       // zero gradient beyond the input length
    for (int c = threadIdx.x; c < C; c += 128) g[c] = 0.f;
    return;
  }
  const int* tg = targets + (long long)b * Lmax;
  int* ext = (int*)(ab + NSmax);
  const long long row = ((long long)b * T + t) * NSmax;
  for (int s = threadIdx.x; s < NS; s += 128) {
    ab[s] = log_alpha[row + s] + log_beta[row + s];
    ext[s] = ctc_ext(tg, s);
  }
  __syncthreads();
  const float NEG = -INFINITY;
  const float nl = nll[b];
  const int tl = S > 1 ? S : 1;
  const float gr = (*finite_flag) ? grad_out[0] / ((float)tl * (float)B) : 0.f;
  const bool bad = !(*finite_flag) || isinf(nl);
  const float* lpr = lp + ((long long)t * B + b) * C;
  for (int c = threadIdx.x; c < C; c += 128) {
    float m = NEG;
    for (int s = (c == 0 ? 0 : 1); s < NS; s += 2)
      if (ext[s] == c && ab[s] > m) m = ab[s];
    float sum = 0.f;
    if (m != NEG)
      for (int s = (c == 0 ? 0 : 1); s < NS; s += 2)
        if (ext[s] == c && ab[s] != NEG) sum += expf(ab[s] - m);
    const float acc = (sum > 0.f) ? logf(sum) + m : NEG;
    const float l = lpr[c];
    float v = (expf(l) - expf(acc + nl - l)) * gr;
    if (bad) v = 0.f;
    g[c] = v;
  }
}

// ---------------- DTW alignment (correct_pred) ----------------
// pred [T][B][C] log-probs, label [L][B] int32. history: [B][T][LL] uint8 scratch, path scratch [B][T+LL] int32.
// out: int64 [T+LL][B] (zero padded), lens[B].
__global__ __launch_bounds__(256) void dtw_kernel(const float* pred, const int* label, int T, int B, int C, int L, unsigned char* history,
                                                  int* path, long long* out, int* lens) {
  extern __shared__ float diag[];  // 3 x (LL+1) rolling anti-diagonals, indexed by j
  const int b = blockIdx.x;
  const int LL = 2 * L + 1;
  const int w = max(T / 2, abs(T - LL));
  float* d0 = diag;                 // diagonal d-2
  float* d1 = diag + (LL + 1);      // diagonal d-1
  float* d2 = diag + 2 * (LL + 1);  // diagonal d
  const float INF = INFINITY;
  unsigned char* hist = history + (long long)b * T * LL;
  // diagonal index d = i + j, i in [0,T], j in [0,LL]; value arrays indexed by j
  for (int j = threadIdx.x; j <= LL; j += 256) { d0[j] = INF; d1[j] = INF; d2[j] = INF; }
  __syncthreads();
  if (threadIdx.x == 0) d1[0] = 0.f;  // dtw[0][0] lives on diagonal 0, which is "d-1" for d = 1
  // diagonal 1 contains (0,1) and (1,0): both inf -> handled by starting with d = 2 and treating d1 as diagonal 1 after a shift
  // Simpler: iterate d from 1; cells with i == 0 or j == 0 are boundary (inf except (0,0)).
  __syncthreads();
  // we keep: d0 = diagonal d-2, d1 = diagonal d-1. Start at d = 1: d1 must be diagonal 0, d0 diagonal -1 (all inf).
  for (int d = 1; d <= T + LL; ++d) {
    for (int j = threadIdx.x; j <= LL; j += 256) {
      const int i = d - j;
      float v = INF;
      if (i >= 1 && i <= T && j >= 1) {
        const int jlo = max(1, i - w), jhi = min(LL, i + w);
        if (j >= jlo && j <= jhi) {
          const int lab = ((j - 1) & 1) ? label[((j - 1) >> 1) * B + b] : 0;
          const float cost = 1.f - pred[((long long)(i - 1) * B + b) * C + lab];
          const float up = d1[j];        // (i-1, j)   on diagonal d-1
          const float dg = d0[j - 1];    // (i-1, j-1) on diagonal d-2
          const float lf = d1[j - 1];    // (i, j-1)   on diagonal d-1
          float m = up; int h = 0;
          if (dg < m) { m = dg; h = 1; }
          if (lf < m) { m = lf; h = 2; }
          v = cost + m;
          hist[(long long)(i - 1) * LL + (j - 1)] = (unsigned char)h;
        }
      }
      d2[j] = v;
    }
    __syncthreads();
    // rotate: d0 <- d1, d1 <- d2
    float* tmp = d0; d0 = d1; d1 = d2; d2 = tmp;
  }
  // backtrace (single thread; at most T+LL steps)
  __shared__ int s_len;
  int* pb = path + (long long)b * (T + LL);
  if (threadIdx.x == 0) {
    int i = T - 1, j = LL - 1, n = 0;
    pb[n++] = (j & 1) ? label[(j >> 1) * B + b] : 0;
    while ((i > 0 || j > 0) && n < T + LL) {
      const int h = hist[(long long)i * LL + j];
      if (h == 0) { if (i > 0) i -= 1; else j -= 1; }
      else if (h == 1) { if (i > 0) i -= 1; if (j > 0) j -= 1; }
      else { if (j > 0) j -= 1; else i -= 1; }
      pb[n++] = (j & 1) ? label[(j >> 1) * B + b] : 0;
    }
    s_len = n;
    lens[b] = n;
  }
  __syncthreads();
  const int n = s_len;
  for (int k = threadIdx.x; k < T + LL; k += 256) out[(long long)k * B + b] = (k < n) ? (long long)pb[n - 1 - k] : 0LL;
}

// ---------------- ground-truth blank/duplicate counts from an aligned label sequence ----------------
// index_spaced [Tp][B] int64, label [L][B] int32 -> gt [L][B][2] float (zero filled by caller), minpos (atomicMin)
__global__ void gt_counts_kernel(const long long* index_spaced, const int* label, int Tp, int B, int L, float* gt, int* minpos, int* mismatch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int c = 0, d = 0, pos = 0, last = 0;
  for (int i = 0; i < Tp; ++i) {
    const int index = (int)index_spaced[(long long)i * B + b];
    if (index == 0 && last == 0) c += 1;
    else if (last == 0 || last == index) { d += 1; last = index; }
    else {
      if (pos < L) {
        if (label[pos * B + b] != last) atomicAdd(mismatch, 1);
        gt[((long long)pos * B + b) * 2 + 0] = (float)c;
        gt[((long long)pos * B + b) * 2 + 1] = (float)d;
      } else atomicAdd(mismatch, 1);
      if (index == 0) { c = 1; d = 0; } else { c = 0; d = 1; }
      pos += 1;
      last = index;
    }
  }
  atomicMin(minpos, pos);
}

}  // namespace

extern "C" int hwg_log_softmax_fwd(const float* x, float* y, long long rows, int C, int B, int T, int transpose_bt, void* stream) {
  HWG_REQUIRE(x && y && rows > 0 && C > 0, "log_softmax_fwd: bad arguments");
  HWG_REQUIRE(!transpose_bt || (long long)B * T == rows, "log_softmax_fwd: B*T != rows");
  long long blocks = (rows + 3) / 4; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, rows, C, B, T, transpose_bt);
  HWG_LAUNCH_CHECK("log_softmax_fwd");
  return HWG_OK;
}
extern "C" int hwg_log_softmax_bwd(const float* dy, const float* y, float* dx, long long rows, int C, int B, int T, int transpose_bt, void* stream) {
  HWG_REQUIRE(dy && y && dx && rows > 0 && C > 0, "log_softmax_bwd: bad arguments");
  long long blocks = (rows + 3) / 4; if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, y, dx, rows, C, B, T, transpose_bt);
  HWG_LAUNCH_CHECK("log_softmax_bwd");
  return HWG_OK;
}

extern "C" size_t hwg_ctc_workspace(int T, int B, int Lmax) {
  const size_t NS = 2 * (size_t)Lmax + 1;
  // log_alpha [B][T][NS] + log_beta [B][T][NS] + nll[B] + flag
  return (2 * (size_t)B * T * NS + B + 4) * sizeof(float);
}
extern "C" int hwg_ctc_fwd(const float* log_probs, const int* targets, const int* input_lengths, const int* target_lengths, int T, int B, int C,
                           int Lmax, float* loss, void* ws, size_t ws_bytes, void* stream) {
  HWG_REQUIRE(log_probs && targets && input_lengths && target_lengths && loss && T > 0 && B > 0 && C > 0 && Lmax > 0, "ctc_fwd: bad arguments");
  if (!ws || ws_bytes < hwg_ctc_workspace(T, B, Lmax)) { hwg_set_error("ctc_fwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const size_t NS = 2 * (size_t)Lmax + 1;
  float* la = (float*)ws;
  float* lb = la + (size_t)B * T * NS;
  float* nll = lb + (size_t)B * T * NS;
  int* flag = (int*)(nll + B);
  hipLaunchKernelGGL(ctc_alpha_kernel, dim3(B), dim3(256), 3 * NS * sizeof(float), st, log_probs, targets, input_lengths, target_lengths, T, B, C, Lmax, la, nll);
  HWG_LAUNCH_CHECK("ctc_alpha");
  hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, st, (const float*)nll, target_lengths, B, loss, flag);
  HWG_LAUNCH_CHECK("ctc_mean");
  return HWG_OK;
}
extern "C" int hwg_ctc_bwd(const float* log_probs, const int* targets, const int* input_lengths, const int* target_lengths, int T, int B, int C,
                           int Lmax, const float* grad_out, float* grad, void* ws, size_t ws_bytes, void* stream) {
  HWG_REQUIRE(log_probs && targets && input_lengths && target_lengths && grad_out && grad, "ctc_bwd: bad arguments");
  if (!ws || ws_bytes < hwg_ctc_workspace(T, B, Lmax)) { hwg_set_error("ctc_bwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const size_t NS = 2 * (size_t)Lmax + 1;
  float* la = (float*)ws;
  float* lb = la + (size_t)B * T * NS;
  float* nll = lb + (size_t)B * T * NS;
  int* flag = (int*)(nll + B);
  const size_t lds = (2 * NS + NS) * sizeof(float);
  hipLaunchKernelGGL(ctc_beta_kernel, dim3(B), dim3(256), lds, st, log_probs, targets, input_lengths, target_lengths, T, B, C, Lmax, lb);
  HWG_LAUNCH_CHECK("ctc_beta");
  hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(128), 2 * NS * sizeof(float), st, log_probs, targets, input_lengths, target_lengths, T, B, C,
                     Lmax, (const float*)la, (const float*)lb, (const float*)nll, grad_out, (const int*)flag, grad);
  HWG_LAUNCH_CHECK("ctc_grad");
  return HWG_OK;
}

extern "C" size_t hwg_dtw_workspace(int T, int B, int L) {
  const size_t LL = 2 * (size_t)L + 1;
  size_t hist = ((size_t)B * T * LL + 15) / 16 * 16;
  return hist + (size_t)B * (T + LL) * sizeof(int);
}
extern "C" int hwg_dtw_align(const float* pred, const int* label, int T, int B, int C, int L, long long* out, int* lens, void* ws, size_t ws_bytes,
                             void* stream) {
  HWG_REQUIRE(pred && label && out && lens && T > 0 && B > 0 && C > 0 && L > 0, "dtw_align: bad arguments");
  if (!ws || ws_bytes < hwg_dtw_workspace(T, B, L)) { hwg_set_error("dtw_align: workspace too small"); return HWG_ERR_WORKSPACE; }
  const size_t LL = 2 * (size_t)L + 1;
  const size_t hist = ((size_t)B * T * LL + 15) / 16 * 16;
  unsigned char* history = (unsigned char*)ws;
  int* path = (int*)((char*)ws + hist);
  const size_t smem = 3 * (LL + 1) * sizeof(float);
  HWG_REQUIRE(smem <= 64 * 1024, "dtw_align: label too long for LDS (%zu B)", smem);
  hipLaunchKernelGGL(dtw_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, pred, label, T, B, C, L, history, path, out, lens);
  HWG_LAUNCH_CHECK("dtw_align");
  return HWG_OK;
}

extern "C" int hwg_gt_counts(const long long* index_spaced, const int* label, int Tp, int B, int L, float* gt, int* minpos, int* mismatch,
                             void* stream) {
  HWG_REQUIRE(index_spaced && label && gt && minpos && mismatch && Tp > 0 && B > 0 && L > 0, "gt_counts: bad arguments");
  hipLaunchKernelGGL(gt_counts_kernel, dim3(hwg_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, index_spaced, label, Tp, B, L, gt, minpos, mismatch);
  HWG_LAUNCH_CHECK("gt_counts");
  return HWG_OK;
}
