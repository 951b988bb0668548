// Spectral-norm power iteration (model/discriminator_ap.py:20-32), loss reductions
// (model/loss.py, trainer/hw_with_style_trainer.py:797-821) and small vector ops.
#include "hwg_common.h"

namespace {

// t[k] = sum_r W[r][k] * u[r]    (W is [R][K] row major, the OIHW weight viewed as [C_out, -1])
__device__ __forceinline__ void sn_wt_u_body(const float* W, const float* u, float* t, int R, int K) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (k < K)
    for (int r = rl; r < R; r += 4) s += W[(long long)r * K + k] * u[r];
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && k < K) t[k] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* W, const float* u, float* t, int R, int K) { sn_wt_u_body(W, u, t, R, K); }
// v = t / (||t|| + eps)   (single block)
__device__ __forceinline__ void sn_normalize_body(const float* t, float* v, int n, float eps, float* norm_out, float* v_copy) {
  __shared__ double sm[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)t[i] * (double)t[i];
  s = block_sum_d(s, sm);
  const float nrm = (float)sqrt(s);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float vi = t[i] / (nrm + eps);
    v[i] = vi;
    if (v_copy) v_copy[i] = vi;
  }
  if (norm_out && threadIdx.x == 0) *norm_out = nrm;
}
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const float* t, float* v, int n, float eps, float* norm_out, float* v_copy = nullptr) {
  sn_normalize_body(t, v, n, eps, norm_out, v_copy);
}
// s[r] = sum_k W[r][k] * v[k]   (one wave per row)
__device__ __forceinline__ void sn_w_v_body(const float* W, const float* v, float* s, int R, int K) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (r >= R) return;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += W[(long long)r * K + k] * v[k];
  acc = wave_sum(acc);
  if (lane == 0) s[r] = acc;
}
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* W, const float* v, float* s, int R, int K) { sn_w_v_body(W, v, s, R, K); }
// u = s/(||s||+eps); sigma = u . s ; inv_sigma = 1/sigma
__device__ __forceinline__ void sn_finish_body(const float* s, float* u, int R, float eps, float* sigma, float* inv_sigma, float* u_copy) {
  __shared__ double sm[16];
  double q = 0.0;
  for (int i = threadIdx.x; i < R; i += blockDim.x) q += (double)s[i] * (double)s[i];
  q = block_sum_d(q, sm);
  const float nrm = (float)sqrt(q);
  double d = 0.0;
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    const float ui = s[i] / (nrm + eps);
    u[i] = ui;
    if (u_copy) u_copy[i] = ui;
    d += (double)ui * (double)s[i];
  }
  d = block_sum_d(d, sm);
  if (threadIdx.x == 0) { *sigma = (float)d; *inv_sigma = (float)(1.0 / d); }
}
__global__ __launch_bounds__(1024) void sn_finish_kernel(const float* s, float* u, int R, float eps, float* sigma, float* inv_sigma, float* u_copy = nullptr) {
  sn_finish_body(s, u, R, eps, sigma, inv_sigma, u_copy);
}
// ---- all spectral-norm layers of a network in four launches (blockIdx.y = layer): the ten layers of the discriminator used to cost 40 ----
struct SnEntry {
  const float* W; float* u; float* v;
  long long copy_off;   // floats into the snapshot buffer: u copy [R], then v copy [K]
  long long ws_off;     // floats into the workspace: t [K], then s [R]
  int R, K;
};
__global__ __launch_bounds__(256) void sn_wt_u_multi_kernel(const SnEntry* tab, float* ws) {
  const SnEntry e = tab[blockIdx.y];
  if ((int)blockIdx.x * 64 >= e.K) return;
  sn_wt_u_body(e.W, e.u, ws + e.ws_off, e.R, e.K);
}
__global__ __launch_bounds__(1024) void sn_normalize_multi_kernel(const SnEntry* tab, const float* ws, float* copies, float eps) {
  const SnEntry e = tab[blockIdx.x];
  sn_normalize_body(ws + e.ws_off, e.v, e.K, eps, nullptr, copies + e.copy_off + e.R);
}
__global__ __launch_bounds__(256) void sn_w_v_multi_kernel(const SnEntry* tab, float* ws) {
  const SnEntry e = tab[blockIdx.y];
  if ((int)blockIdx.x * 4 >= e.R) return;
  sn_w_v_body(e.W, e.v, ws + e.ws_off + e.K, e.R, e.K);
}
__global__ __launch_bounds__(1024) void sn_finish_multi_kernel(const SnEntry* tab, const float* ws, float* copies, float* sig, float eps) {
  const SnEntry e = tab[blockIdx.x];
  sn_finish_body(ws + e.ws_off + e.K, e.u, e.R, eps, sig + 2 * blockIdx.x, sig + 2 * blockIdx.x + 1, copies + e.copy_off);
}
// out = W * (*scale)
__global__ void scale_by_ptr_kernel(const float* W, const float* scale, float* out, long long n) {
  const float s = *scale;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = W[i] * s;
}
// partial dot products: part[b] = sum a*b
__global__ __launch_bounds__(256) void dot_partial_kernel(const float* a, const float* b, long long n, double* part) {
  __shared__ double sm[16];
  double s = 0.0;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)a[i] * (double)b[i];
  s = block_sum_d(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// dW_bar[r][k] (+)= dWsn[r][k]/sigma - (dot/sigma^2) * u[r] * v[k]
__global__ void sn_bwd_kernel(const float* dWsn, const float* u, const float* v, const float* sigma, const double* part, int nparts, float* dWbar,
                              int R, int K, int accumulate) {
  __shared__ float s_coef;
  if (threadIdx.x == 0) {
    double dot = 0.0;
    for (int i = 0; i < nparts; ++i) dot += part[i];
    const double sg = (double)*sigma;
    s_coef = (float)(dot / (sg * sg));
  }
  __syncthreads();
  const float coef = s_coef;
  const float inv = 1.f / *sigma;
  const long long n = (long long)R * K;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / K), k = (int)(i % K);
    const float g = dWsn[i] * inv - coef * u[r] * v[k];
    dWbar[i] = accumulate ? dWbar[i] + g : g;
  }
}

// ---- the backward passes of all spectral-norm layers of a network in two launches (hwg_spectral_bwd_multi) ----------------------------------
// entries by value in the kernel arguments; block -> (entry, block of that entry's own launch): every entry runs exactly the partial
// schedule and the element arithmetic hwg_spectral_bwd would have run for it alone, so the results are bit-identical.
struct SnBwdEntry {
  const float* dWsn; const float* Wbar; const float* u; const float* v; const float* sigma; float* dst;
  int R, K, np, part_off, dot_first, bwd_first, bwd_blocks, accumulate;
};
constexpr int SNB_MAX = 16;
struct SnBwdTable { int n, pad; SnBwdEntry e[SNB_MAX]; };
__global__ __launch_bounds__(256) void sn_dot_multi_kernel(const SnBwdTable t, double* part) {
  __shared__ double sm[16];
  int k = t.n - 1;
  while (k > 0 && t.e[k].dot_first > (int)blockIdx.x) --k;
  const SnBwdEntry e = t.e[k];
  const int b = blockIdx.x - e.dot_first;
  const long long n = (long long)e.R * e.K;
  double s = 0.0;
  for (long long i = b * 256LL + threadIdx.x; i < n; i += (long long)e.np * 256) s += (double)e.dWsn[i] * (double)e.Wbar[i];
  s = block_sum_d(s, sm);
  if (threadIdx.x == 0) part[e.part_off + b] = s;
}
__global__ __launch_bounds__(256) void sn_bwd_multi_kernel(const SnBwdTable t, const double* part) {
  __shared__ float s_coef;
  int k = t.n - 1;
  while (k > 0 && t.e[k].bwd_first > (int)blockIdx.x) --k;
  const SnBwdEntry e = t.e[k];
  const int b = blockIdx.x - e.bwd_first;
  if (threadIdx.x == 0) {
    double dot = 0.0;
    for (int i = 0; i < e.np; ++i) dot += part[e.part_off + i];
    const double sg = (double)*e.sigma;
    s_coef = (float)(dot / (sg * sg));
  }
  __syncthreads();
  const float coef = s_coef;
  const float inv = 1.f / *e.sigma;
  const long long n = (long long)e.R * e.K;
  for (long long i = b * 256LL + threadIdx.x; i < n; i += (long long)e.bwd_blocks * 256) {
    const int r = (int)(i / e.K), c = (int)(i % e.K);
    const float g = e.dWsn[i] * inv - coef * e.u[r] * e.v[c];
    e.dst[i] = e.accumulate ? e.dst[i] + g : g;
  }
}

// ---------------- reductions for losses ----------------
// mode 0: sum |a-b|   1: sum (a-b)^2   2: sum a   3: sum relu(1-a)   4: sum relu(1+a)
__device__ __forceinline__ float loss_term(float a, float b, int mode) {
  switch (mode) {
    case 0: return fabsf(a - b);
    case 1: return (a - b) * (a - b);
    case 2: return a;
    case 3: return fmaxf(1.f - a, 0.f);
    default: return fmaxf(1.f + a, 0.f);
  }
}
__global__ __launch_bounds__(256) void loss_partial_kernel(const float* a, const float* b, long long n, int mode, double* part) {
  __shared__ double sm[16];
  double s = 0.0;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += (double)loss_term(a[i], b ? b[i] : 0.f, mode);
  s = block_sum_d(s, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
// out (+)= scale * sum(part) / n
__global__ void loss_final_kernel(const double* part, int nparts, double inv_n, float scale, float* out, int accumulate) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += part[i];
    const float v = (float)(s * inv_n) * scale;
    *out = accumulate ? *out + v : v;
  }
}
// gradient wrt a (and -that wrt b for the pair losses): da = (*gout) * scale/n * d term/da
__global__ void loss_bwd_kernel(const float* a, const float* b, long long n, int mode, const float* gout, float coef, float* da, float* db,
                                int accumulate) {
  const float g = (*gout) * coef;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float av = a[i], bv = b ? b[i] : 0.f;
    float d;
    switch (mode) {
      case 0: d = (av > bv) ? 1.f : ((av < bv) ? -1.f : 0.f); break;
      case 1: d = 2.f * (av - bv); break;
      case 2: d = 1.f; break;
      case 3: d = (1.f - av > 0.f) ? -1.f : 0.f; break;
      default: d = (1.f + av > 0.f) ? 1.f : 0.f; break;
    }
    d *= g;
    if (da) da[i] = accumulate ? da[i] + d : d;
    if (db) db[i] = accumulate ? db[i] - d : -d;
  }
}

// PixelNorm over the channel dim of [rows][C]: y = x / sqrt(mean(x^2) + 1e-8)  (model/pure_gen.py:306-311)
__global__ __launch_bounds__(256) void pixelnorm_fwd_kernel(const float* x, float* y, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[(long long)r * C + c]; s += v * v; }
  s = wave_sum(s);
  const float d = sqrtf(s / (float)C + eps);
  for (int c = lane; c < C; c += 64) y[(long long)r * C + c] = x[(long long)r * C + c] / d;
}
// dx = dy/d - x * (sum(dy*x) / (C * d^3))
__global__ __launch_bounds__(256) void pixelnorm_bwd_kernel(const float* dy, const float* x, float* dx, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int r = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (r >= rows) return;
  float s = 0.f, t = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[(long long)r * C + c];
    s += v * v;
    t += v * dy[(long long)r * C + c];
  }
  s = wave_sum(s); t = wave_sum(t);
  const float d = sqrtf(s / (float)C + eps);
  const float k = t / ((float)C * d * d * d);
  for (int c = lane; c < C; c += 64) dx[(long long)r * C + c] = dy[(long long)r * C + c] / d - x[(long long)r * C + c] * k;
}

// y = a*x + b*y' style axpby and simple unary maps used as glue
__global__ void axpby_kernel(const float* x, float a, const float* y, float b, float* out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
// weighted sum of up to 16 loss scalars (trainer/hw_with_style_trainer.py:280-298: `loss += losses[name] * lossWeights[name]`): scaled[i] = w_i * x_i
// (x_i itself where w_i == 1, as ops.scale skips that product) and their left-to-right sum, every operation rounded like the chain of
// hwg_axpby launches it replaces; backward: g_i = w_i * gout
struct WSum { int n; const float* x[16]; float w[16]; };
__global__ void weighted_sum_kernel(WSum a, float* scaled, float* sum) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float acc = 0.f;
  for (int i = 0; i < a.n; ++i) {
    const float v = a.w[i] == 1.f ? *a.x[i] : __fmul_rn(*a.x[i], a.w[i]);
    scaled[i] = v;
    acc = i == 0 ? v : __fadd_rn(acc, v);
  }
  *sum = acc;
}
__global__ void scale_scalars_kernel(const float* gout, WSum a, float* out) {
  const int i = threadIdx.x;
  if (i < a.n) out[i] = a.w[i] == 1.f ? *gout : __fmul_rn(*gout, a.w[i]);
}
// out[b][d] = bank[ij[b]][d] * w[b] + bank[ij[B + b]][d] * w[B + b]: the trainer's style interpolation (trainer/hw_with_style_trainer.py:974-988) in one
// launch; both products and the sum rounded to fp32 like the tensor expression it replaces
__global__ void style_mix_kernel(const float* bank, const int* ij, const float* w, float* out, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  out[i] = __fadd_rn(__fmul_rn(bank[(long long)ij[b] * D + d], w[b]), __fmul_rn(bank[(long long)ij[B + b] * D + d], w[B + b]));
}
__global__ void tanh_fwd_kernel(const float* x, float* y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = tanhf(x[i]);
}
__global__ void tanh_bwd_kernel(const float* dy, const float* y, float* dx, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dx[i] = dy[i] * (1.f - y[i] * y[i]);
}
// argmax over channels for each row (first maximum wins)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* x, int* out, long long rows, int C) {
  const int lane = threadIdx.x & 63;
  const long long r = (blockIdx.x * 256LL + threadIdx.x) >> 6;
  if (r >= rows) return;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[r * C + c];
    if (v > best || (v == best && c < bi)) { best = v; bi = c; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) out[r] = bi;
}

// y[r][c] = x[r][c] * scale[c] + shift[c]
__global__ void channel_affine_kernel(const float* x, const float* scale, const float* shift, float* y, long long rows, int C) {
  const long long n = rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    y[i] = x[i] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
  }
}
__global__ void mul_kernel(const float* a, const float* b, float* out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

int dot_parts(long long n) { long long p = (n + 4095) / 4096; if (p > 512) p = 512; if (p < 1) p = 1; return (int)p; }

}  // namespace

extern "C" size_t hwg_spectral_workspace(int R, int K) { return ((size_t)R + K + 8) * sizeof(float) + 512 * sizeof(double) + 16; }

// One power iteration: v <- normalize(W^T u); u <- normalize(W v); sigma = u.(W v). u and v are updated in place,
// sigma/inv_sigma are single floats on the device (consumed by hwg_scale_by_ptr / the weight packer).
extern "C" int hwg_spectral_update_to(const float* W, float* u, float* v, float* u_copy, float* v_copy, int R, int K, float eps, float* sigma,
                                      float* inv_sigma, void* ws, size_t ws_bytes, void* stream) {
  HWG_REQUIRE(W && u && v && sigma && inv_sigma && R > 0 && K > 0, "spectral_update: bad arguments");
  if (!ws || ws_bytes < hwg_spectral_workspace(R, K)) { hwg_set_error("spectral_update: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* t = (float*)ws;       // [K]
  float* s = t + K;            // [R]
  hipLaunchKernelGGL(sn_wt_u_kernel, dim3(hwg_cdiv(K, 64)), dim3(256), 0, st, W, (const float*)u, t, R, K);
  HWG_LAUNCH_CHECK("sn_wt_u");
  hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)t, v, K, eps, (float*)nullptr, v_copy);
  HWG_LAUNCH_CHECK("sn_normalize");
  hipLaunchKernelGGL(sn_w_v_kernel, dim3(hwg_cdiv(R, 4)), dim3(256), 0, st, W, (const float*)v, s, R, K);
  HWG_LAUNCH_CHECK("sn_w_v");
  hipLaunchKernelGGL(sn_finish_kernel, dim3(1), dim3(1024), 0, st, (const float*)s, u, R, eps, sigma, inv_sigma, u_copy);
  HWG_LAUNCH_CHECK("sn_finish");
  return HWG_OK;
}
extern "C" int hwg_spectral_update_multi(const void* table, int n, int max_R, int max_K, float eps, float* ws, float* copies, float* sig, void* stream) {
  HWG_REQUIRE(table && n > 0 && max_R > 0 && max_K > 0 && ws && copies && sig, "spectral_update_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const SnEntry* tab = (const SnEntry*)table;
  hipLaunchKernelGGL(sn_wt_u_multi_kernel, dim3(hwg_cdiv(max_K, 64), n), dim3(256), 0, st, tab, ws);
  hipLaunchKernelGGL(sn_normalize_multi_kernel, dim3(n), dim3(1024), 0, st, tab, (const float*)ws, copies, eps);
  hipLaunchKernelGGL(sn_w_v_multi_kernel, dim3(hwg_cdiv(max_R, 4), n), dim3(256), 0, st, tab, ws);
  hipLaunchKernelGGL(sn_finish_multi_kernel, dim3(n), dim3(1024), 0, st, tab, (const float*)ws, copies, sig, eps);
  HWG_LAUNCH_CHECK("spectral_update_multi");
  return HWG_OK;
}
extern "C" int hwg_spectral_update(const float* W, float* u, float* v, int R, int K, float eps, float* sigma, float* inv_sigma, void* ws,
                                   size_t ws_bytes, void* stream) {
  return hwg_spectral_update_to(W, u, v, nullptr, nullptr, R, K, eps, sigma, inv_sigma, ws, ws_bytes, stream);
}
extern "C" int hwg_scale_by_ptr(const float* x, const float* scale, float* out, long long n, void* stream) {
  HWG_REQUIRE(x && scale && out && n > 0, "scale_by_ptr: bad arguments");
  hipLaunchKernelGGL(scale_by_ptr_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, scale, out, n);
  HWG_LAUNCH_CHECK("scale_by_ptr");
  return HWG_OK;
}
extern "C" int hwg_spectral_bwd(const float* dWsn, const float* Wbar, const float* u, const float* v, const float* sigma, float* dWbar, int R, int K,
                                int accumulate, void* ws, size_t ws_bytes, void* stream) {
  HWG_REQUIRE(dWsn && Wbar && u && v && sigma && dWbar && R > 0 && K > 0, "spectral_bwd: bad arguments");
  if (!ws || ws_bytes < hwg_spectral_workspace(R, K)) { hwg_set_error("spectral_bwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  // the double partials live at the (8-byte aligned) tail of the workspace
  double* part = (double*)((char*)ws + (((size_t)R + K + 8) * sizeof(float) + 7) / 8 * 8);
  const long long n = (long long)R * K;
  const int np = dot_parts(n);
  hipLaunchKernelGGL(dot_partial_kernel, dim3(np), dim3(256), 0, st, dWsn, Wbar, n, part);
  HWG_LAUNCH_CHECK("sn_dot");
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, st, dWsn, u, v, sigma, (const double*)part, np, dWbar, R, K, accumulate);
  HWG_LAUNCH_CHECK("sn_bwd");
  return HWG_OK;
}

// hwg_spectral_bwd for up to 16 layers at once. `table`: n host records {dWsn, Wbar, u, v, sigma, dst (8-byte addresses), R, K, accumulate,
// pad (4-byte ints)} = 64 bytes each; workspace: hwg_spectral_bwd_multi_workspace(n) bytes.
extern "C" size_t hwg_spectral_bwd_multi_workspace(int n) { return (size_t)(n > 0 ? n : 1) * 512 * sizeof(double); }
extern "C" int hwg_spectral_bwd_multi(const void* table, int n, void* ws, size_t ws_bytes, void* stream) {
  HWG_REQUIRE(table && n > 0 && n <= SNB_MAX, "spectral_bwd_multi: 1..16 layers per call");
  if (!ws || ws_bytes < hwg_spectral_bwd_multi_workspace(n)) { hwg_set_error("spectral_bwd_multi: workspace too small"); return HWG_ERR_WORKSPACE; }
  struct Rec { unsigned long long dWsn, Wbar, u, v, sigma, dst; int R, K, accumulate, pad; };
  static_assert(sizeof(Rec) == 64, "record layout");
  const Rec* rec = (const Rec*)table;
  SnBwdTable t;
  t.n = n; t.pad = 0;
  int dot_blocks = 0, bwd_blocks = 0, part_off = 0;
  for (int i = 0; i < n; ++i) {
    const Rec& r = rec[i];
    HWG_REQUIRE(r.dWsn && r.Wbar && r.u && r.v && r.sigma && r.dst && r.R > 0 && r.K > 0, "spectral_bwd_multi: bad record");
    SnBwdEntry& e = t.e[i];
    e.dWsn = (const float*)r.dWsn; e.Wbar = (const float*)r.Wbar; e.u = (const float*)r.u; e.v = (const float*)r.v;
    e.sigma = (const float*)r.sigma; e.dst = (float*)r.dst;
    e.R = r.R; e.K = r.K; e.accumulate = r.accumulate;
    const long long nn = (long long)r.R * r.K;
    e.np = dot_parts(nn);
    e.part_off = part_off; part_off += e.np;
    e.dot_first = dot_blocks; dot_blocks += e.np;
    e.bwd_blocks = hwg_stream_grid(nn, 256);
    e.bwd_first = bwd_blocks; bwd_blocks += e.bwd_blocks;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_dot_multi_kernel, dim3(dot_blocks), dim3(256), 0, st, t, (double*)ws);
  HWG_LAUNCH_CHECK("sn_dot_multi");
  hipLaunchKernelGGL(sn_bwd_multi_kernel, dim3(bwd_blocks), dim3(256), 0, st, t, (const double*)ws);
  HWG_LAUNCH_CHECK("sn_bwd_multi");
  return HWG_OK;
}

extern "C" size_t hwg_loss_workspace(void) { return 512 * sizeof(double); }
// out (+)= scale * mean(term(a,b)); modes: 0 L1, 1 MSE, 2 mean(a), 3 mean(relu(1-a)), 4 mean(relu(1+a))
extern "C" int hwg_loss_fwd(const float* a, const float* b, long long n, int mode, float scale, float* out, int accumulate, void* ws, size_t ws_bytes,
                            void* stream) {
  HWG_REQUIRE(a && out && n > 0 && mode >= 0 && mode <= 4, "loss_fwd: bad arguments");
  HWG_REQUIRE(mode > 1 || b, "loss_fwd: pair loss needs b");
  if (!ws || ws_bytes < hwg_loss_workspace()) { hwg_set_error("loss_fwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  const int np = dot_parts(n);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(np), dim3(256), 0, st, a, b, n, mode, (double*)ws);
  HWG_LAUNCH_CHECK("loss_partial");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, (const double*)ws, np, 1.0 / (double)n, scale, out, accumulate);
  HWG_LAUNCH_CHECK("loss_final");
  return HWG_OK;
}
extern "C" int hwg_loss_bwd(const float* a, const float* b, long long n, int mode, float scale, const float* grad_out, float* da, float* db,
                            int accumulate, void* stream) {
  HWG_REQUIRE(a && grad_out && n > 0 && (da || db), "loss_bwd: bad arguments");
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, mode, grad_out, scale / (float)n, da, db,
                     accumulate);
  HWG_LAUNCH_CHECK("loss_bwd");
  return HWG_OK;
}

extern "C" int hwg_pixelnorm_fwd(const float* x, float* y, int rows, int C, float eps, void* stream) {
  HWG_REQUIRE(x && y && rows > 0 && C > 0, "pixelnorm_fwd: bad arguments");
  hipLaunchKernelGGL(pixelnorm_fwd_kernel, dim3(hwg_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, C, eps);
  HWG_LAUNCH_CHECK("pixelnorm_fwd");
  return HWG_OK;
}
extern "C" int hwg_pixelnorm_bwd(const float* dy, const float* x, float* dx, int rows, int C, float eps, void* stream) {
  HWG_REQUIRE(dy && x && dx && rows > 0 && C > 0, "pixelnorm_bwd: bad arguments");
  hipLaunchKernelGGL(pixelnorm_bwd_kernel, dim3(hwg_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, rows, C, eps);
  HWG_LAUNCH_CHECK("pixelnorm_bwd");
  return HWG_OK;
}
extern "C" int hwg_axpby(const float* x, float a, const float* y, float b, float* out, long long n, void* stream) {
  HWG_REQUIRE(x && out && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, a, y, b, out, n);
  HWG_LAUNCH_CHECK("axpby");
  return HWG_OK;
}
extern "C" int hwg_weighted_sum(const void* x_ptrs, const float* weights, int n, float* scaled, float* sum, void* stream) {
  HWG_REQUIRE(x_ptrs && weights && scaled && sum && n > 0 && n <= 16, "weighted_sum: bad arguments (1..16 terms)");
  WSum a; a.n = n;
  for (int i = 0; i < n; ++i) { a.x[i] = reinterpret_cast<const float*>(((const long long*)x_ptrs)[i]); a.w[i] = weights[i]; HWG_REQUIRE(a.x[i], "weighted_sum: null term"); }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, scaled, sum);
  HWG_LAUNCH_CHECK("weighted_sum");
  return HWG_OK;
}
extern "C" int hwg_weighted_sum_bwd(const float* grad_out, const float* weights, int n, float* grads, void* stream) {
  HWG_REQUIRE(grad_out && weights && grads && n > 0 && n <= 16, "weighted_sum_bwd: bad arguments (1..16 terms)");
  WSum a; a.n = n;
  for (int i = 0; i < n; ++i) { a.x[i] = nullptr; a.w[i] = weights[i]; }
  hipLaunchKernelGGL(scale_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grad_out, a, grads);
  HWG_LAUNCH_CHECK("weighted_sum_bwd");
  return HWG_OK;
}
extern "C" int hwg_style_mix(const float* bank, const int* ij, const float* w, float* out, int K, int B, int D, void* stream) {
  HWG_REQUIRE(bank && ij && w && out && K > 0 && B > 0 && D > 0, "style_mix: bad arguments");
  hipLaunchKernelGGL(style_mix_kernel, dim3(hwg_cdiv((long long)B * D, 256)), dim3(256), 0, (hipStream_t)stream, bank, ij, w, out, B, D);
  HWG_LAUNCH_CHECK("style_mix");
  return HWG_OK;
}
extern "C" int hwg_tanh_fwd(const float* x, float* y, long long n, void* stream) {
  HWG_REQUIRE(x && y && n > 0, "tanh_fwd: bad arguments");
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  HWG_LAUNCH_CHECK("tanh_fwd");
  return HWG_OK;
}
extern "C" int hwg_tanh_bwd(const float* dy, const float* y, float* dx, long long n, void* stream) {
  HWG_REQUIRE(dy && y && dx && n > 0, "tanh_bwd: bad arguments");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
  HWG_LAUNCH_CHECK("tanh_bwd");
  return HWG_OK;
}
extern "C" int hwg_argmax_rows(const float* x, int* out, long long rows, int C, void* stream) {
  HWG_REQUIRE(x && out && rows > 0 && C > 0, "argmax_rows: bad arguments");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, C);
  HWG_LAUNCH_CHECK("argmax_rows");
  return HWG_OK;
}

extern "C" int hwg_channel_affine(const float* x, const float* scale, const float* shift, float* y, long long rows, int C, void* stream) {
  HWG_REQUIRE(x && y && rows > 0 && C > 0, "channel_affine: bad arguments");
  hipLaunchKernelGGL(channel_affine_kernel, dim3(hwg_stream_grid(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, rows, C);
  HWG_LAUNCH_CHECK("channel_affine");
  return HWG_OK;
}
extern "C" int hwg_mul(const float* a, const float* b, float* out, long long n, void* stream) {
  HWG_REQUIRE(a && b && out && n > 0, "mul: bad arguments");
  hipLaunchKernelGGL(mul_kernel, dim3(hwg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  HWG_LAUNCH_CHECK("mul");
  return HWG_OK;
}
