// Multi-tensor optimizer-side kernels (gradient stash / balance / clip / Adam / NaN scan over the ~1.3k
// parameter tensors, trainer/hw_with_style_trainer.py:300-391) and the Philox random number kernels that
// replace torch.randn_like / Dropout2d masks (model/pure_gen.py:206,212; nn.Dropout2d sites).
//
// Tensor lists are described by device tables built once on the host:
//   ptrs*      : int64 array of device addresses, one per tensor (0 = tensor absent -> chunk skipped)
//   chunk_tensor / chunk_off : one entry per fixed-size chunk of a tensor, so one launch covers all tensors.
#include "hwg_common.h"
#include "philox.h"

namespace {

constexpr int MT_U = 4;   // 16-byte pieces per stream in flight per thread in the multi-tensor streaming loops

struct MtArgs {
  const long long* pa;
  const long long* pb;
  const long long* pc;
  const long long* pd;
  const long long* numel;
  const int* chunk_tensor;
  const long long* chunk_off;
  int chunk;
};

// blockIdx.y = set (hwg_mt_abs_sum_sets: pointer tables [nsets][nt], partials [nsets][nchunks]; one set: the plain call)
__global__ __launch_bounds__(256) void mt_abs_sum_kernel(MtArgs a, double* out, int nt) {
  __shared__ double sm[16];
  const int t = a.chunk_tensor[blockIdx.x];
  const float* g = reinterpret_cast<const float*>(a.pa[(size_t)blockIdx.y * nt + t]);
  out += (size_t)blockIdx.y * gridDim.x;
  if (!g) return;
  const long long off = a.chunk_off[blockIdx.x];
  const long long end = min(off + (long long)a.chunk, a.numel[t]);
  double s = 0.0;
  // 16-byte loads over the aligned body (every table entry is 16-byte aligned and chunks are multiples of 4 elements), scalar tail
  const bool al = ((reinterpret_cast<uintptr_t>(g + off)) & 15) == 0;
  const long long n4 = al ? (end - off) >> 2 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g + off);
#pragma unroll 4
  for (long long j = threadIdx.x; j < n4; j += 256) {
    const float4 v = g4[j];
    s += (double)(fabsf(v.x) + fabsf(v.y)) + (double)(fabsf(v.z) + fabsf(v.w));
  }
  for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) s += (double)fabsf(g[i]);
  s = block_sum_d(s, sm);
  if (threadIdx.x == 0) out[blockIdx.x] = s;      // per-chunk partial; mt_abs_sum_final_kernel adds a tensor's chunks in table order
}

// sums[t] = sum of the chunk partials of tensor t, in chunk-table order (a tensor's chunks are consecutive entries): deterministic,
// unlike an atomicAdd per chunk
__global__ __launch_bounds__(256) void mt_abs_sum_final_kernel(const double* part, const int* chunk_tensor, int nchunks, int nt, double* sums) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  part += (size_t)blockIdx.y * nchunks; sums += (size_t)blockIdx.y * nt;
  // first chunk of tensor t by binary search (chunk_tensor is non-decreasing)
  int lo = 0, hi = nchunks;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (chunk_tensor[mid] < t) lo = mid + 1; else hi = mid;
  }
  double s = 0.0;
  for (int c = lo; c < nchunks && chunk_tensor[c] == t; ++c) s += part[c];
  sums[t] = s;
}

// coefficients of the balanced add:  coef[k][t] = x_k * D_t / R_kt   (0 when the stashed tensor is absent or all-zero)
// D_t = mean|grad_t|, zero means are replaced by the mean of the non-zero ones (trainer :341-359)
__global__ __launch_bounds__(256) void mt_balance_coef_kernel(const double* sumD, const double* sumR /*[nsets][nt]*/, const long long* numel,
                                                              const long long* ptr_grad, const long long* ptr_R /*[nsets][nt]*/, const float* xs,
                                                              int nsets, int nt, float* coef /*[nsets][nt]*/) {
  __shared__ double sm[16];
  __shared__ double s_nonzero;
  double zs = 0.0, zc = 0.0;
  for (int t = threadIdx.x; t < nt; t += 256) {
    if (!ptr_grad[t]) continue;
    const float m = (float)(sumD[t] / (double)numel[t]);
    if (m != 0.f) { zs += (double)m; zc += 1.0; }
  }
  zs = block_sum_d(zs, sm);
  zc = block_sum_d(zc, sm);
  if (threadIdx.x == 0) s_nonzero = zc > 0.0 ? (double)((float)zs / (float)zc) : 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < nsets * nt; i += 256) {
    const int t = i % nt, k = i / nt;
    float c = 0.f;
    if (ptr_grad[t] && ptr_R[i]) {
      float d = (float)(sumD[t] / (double)numel[t]);
      if (d == 0.f && zc > 0.0) d = (float)s_nonzero;
      const float r = (float)(sumR[i] / (double)numel[t]);
      if (r != 0.f) c = xs[k] * (d / r);
    }
    coef[i] = c;
  }
}

// dst += coef[t] * src
__global__ __launch_bounds__(256) void mt_axpy_kernel(MtArgs a, const float* coef) {
  const int t = a.chunk_tensor[blockIdx.x];
  float* dst = reinterpret_cast<float*>(a.pa[t]);
  const float* src = reinterpret_cast<const float*>(a.pb[t]);
  if (!dst || !src) return;
  const float c = coef ? coef[t] : 1.f;
  if (c == 0.f) return;
  const long long off = a.chunk_off[blockIdx.x];
  const long long end = min(off + (long long)a.chunk, a.numel[t]);
  const bool al = ((reinterpret_cast<uintptr_t>(dst + off) | reinterpret_cast<uintptr_t>(src + off)) & 15) == 0;
  const long long n4 = al ? (end - off) >> 2 : 0;
  float4* d4 = reinterpret_cast<float4*>(dst + off);
  const float4* s4 = reinterpret_cast<const float4*>(src + off);
  // MT_U 16-byte pieces per stream in flight per thread: the loads of a trip are issued before its first store (the compiler may not move a
  // load across a store through a pointer that could alias it, so the plain loop ran one piece at a time)
  for (long long j = threadIdx.x; j < n4; j += MT_U * 256) {
    float4 d[MT_U], v[MT_U];
#pragma unroll
    for (int u = 0; u < MT_U; ++u) {
      const long long ju = j + u * 256 < n4 ? j + u * 256 : j;
      d[u] = d4[ju];
      v[u] = s4[ju];
    }
#pragma unroll
    for (int u = 0; u < MT_U; ++u) {
      if (j + u * 256 >= n4) break;
      d[u].x += c * v[u].x; d[u].y += c * v[u].y; d[u].z += c * v[u].z; d[u].w += c * v[u].w;
      d4[j + u * 256] = d[u];
    }
  }
  for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) dst[i] += c * src[i];
}

// dst += coef[0][t] * src_0, then += coef[1][t] * src_1, ... in ONE pass over dst (hwg_mt_axpy_sets): per element the same chain of fused
// multiply-adds, in the same order, as nsets launches of mt_axpy_kernel (whose `d += c * v` compiles to v_fma) - dst is read and written once
// instead of nsets times. a.pb = source tables [nsets][nt]; absent sources and zero coefficients are skipped like the single launches skip them.
__global__ __launch_bounds__(256) void mt_axpy_sets_kernel(MtArgs a, const float* coef, int nsets, int nt) {
  const int t = a.chunk_tensor[blockIdx.x];
  float* dst = reinterpret_cast<float*>(a.pa[t]);
  if (!dst) return;
  constexpr int MAXS = 8;
  const float* src[MAXS];     // (indexed by unrolled loops only: registers)
  float c[MAXS];
  bool any = false, al = true;
  const long long off = a.chunk_off[blockIdx.x];
#pragma unroll
  for (int k = 0; k < MAXS; ++k) {
    src[k] = nullptr; c[k] = 0.f;
    if (k < nsets) {
      const float* sk = reinterpret_cast<const float*>(a.pb[(size_t)k * nt + t]);
      const float ck = coef[(size_t)k * nt + t];
      if (sk && ck != 0.f) {
        src[k] = sk; c[k] = ck; any = true;
        al = al && ((reinterpret_cast<uintptr_t>(sk + off)) & 15) == 0;
      }
    }
  }
  if (!any) return;
  const long long end = min(off + (long long)a.chunk, a.numel[t]);
  al = al && ((reinterpret_cast<uintptr_t>(dst + off)) & 15) == 0;
  const long long n4 = al ? (end - off) >> 2 : 0;
  float4* d4 = reinterpret_cast<float4*>(dst + off);
  for (long long j = threadIdx.x; j < n4; j += 256) {
    float4 d = d4[j];
    float4 v[MAXS];
#pragma unroll
    for (int k = 0; k < MAXS; ++k)
      if (src[k]) v[k] = reinterpret_cast<const float4*>(src[k] + off)[j];
#pragma unroll
    for (int k = 0; k < MAXS; ++k)
      if (src[k]) {
        d.x = __fmaf_rn(c[k], v[k].x, d.x); d.y = __fmaf_rn(c[k], v[k].y, d.y);
        d.z = __fmaf_rn(c[k], v[k].z, d.z); d.w = __fmaf_rn(c[k], v[k].w, d.w);
      }
    d4[j] = d;
  }
  for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) {
    float d = dst[i];
#pragma unroll
    for (int k = 0; k < MAXS; ++k)
      if (src[k]) d = __fmaf_rn(c[k], src[k][i], d);
    dst[i] = d;
  }
}

// op 0: a = 0 ; 1: a = clamp(a, -c, c) ; 2: flag |= any(!finite(a)) ; 3: b = a (copy) ; 4: b = a, a = 0 (stash)
__global__ __launch_bounds__(256) void mt_unary_kernel(MtArgs a, int op, float c, int* flag) {
  const int t = a.chunk_tensor[blockIdx.x];
  float* x = reinterpret_cast<float*>(a.pa[t]);
  if (!x) return;
  float* y = a.pb ? reinterpret_cast<float*>(a.pb[t]) : nullptr;
  const long long off = a.chunk_off[blockIdx.x];
  const long long end = min(off + (long long)a.chunk, a.numel[t]);
  bool bad = false;
  const bool al = ((reinterpret_cast<uintptr_t>(x + off) | (y ? reinterpret_cast<uintptr_t>(y + off) : 0)) & 15) == 0;
  const long long n4 = al ? (end - off) >> 2 : 0;
  float4* x4 = reinterpret_cast<float4*>(x + off);
  float4* y4 = y ? reinterpret_cast<float4*>(y + off) : nullptr;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long j0 = threadIdx.x; j0 < n4; j0 += MT_U * 256) {
    float4 vv[MT_U];
    if (op != 0) {
#pragma unroll
      for (int u = 0; u < MT_U; ++u) vv[u] = x4[j0 + u * 256 < n4 ? j0 + u * 256 : j0];
    }
#pragma unroll
    for (int u = 0; u < MT_U; ++u) {
      const long long j = j0 + u * 256;
      if (j >= n4) break;
      const float4 v = vv[u];
      if (op == 0) x4[j] = zero4;
      else if (op == 1) x4[j] = make_float4(fminf(fmaxf(v.x, -c), c), fminf(fmaxf(v.y, -c), c), fminf(fmaxf(v.z, -c), c), fminf(fmaxf(v.w, -c), c));
      else if (op == 2) bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
      else if (op == 3) { if (y4) y4[j] = v; }
      else { if (y4) y4[j] = v; x4[j] = zero4; }
    }
  }
  for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) {
    const float v = x[i];
    if (op == 0) x[i] = 0.f;
    else if (op == 1) x[i] = fminf(fmaxf(v, -c), c);
    else if (op == 2) bad |= !isfinite(v);
    else if (op == 3) { if (y) y[i] = v; }
    else { if (y) y[i] = v; x[i] = 0.f; }
  }
  if (op == 2 && bad) atomicOr(flag, 1);
}

// torch.optim.Adam (no amsgrad / weight decay) with the trainer's clip_grad_value_ fused into the gradient read
__global__ __launch_bounds__(256) void mt_adam_kernel(MtArgs a, const float* step_size, const float* bc2_sqrt, float beta1, float beta2, float eps,
                                                      float clip) {
  const int t = a.chunk_tensor[blockIdx.x];
  float* p = reinterpret_cast<float*>(a.pa[t]);
  float* g = reinterpret_cast<float*>(a.pb[t]);
  float* m = reinterpret_cast<float*>(a.pc[t]);
  float* v = reinterpret_cast<float*>(a.pd[t]);
  if (!p || !g || !m || !v) return;
  const float ss = step_size[t], b2 = bc2_sqrt[t];
  const long long off = a.chunk_off[blockIdx.x];
  const long long end = min(off + (long long)a.chunk, a.numel[t]);
  const bool al = ((reinterpret_cast<uintptr_t>(p + off) | reinterpret_cast<uintptr_t>(g + off) | reinterpret_cast<uintptr_t>(m + off) |
                    reinterpret_cast<uintptr_t>(v + off)) & 15) == 0;
  const long long n4 = al ? (end - off) >> 2 : 0;
  float4* p4 = reinterpret_cast<float4*>(p + off);
  float4* g4 = reinterpret_cast<float4*>(g + off);
  float4* m4 = reinterpret_cast<float4*>(m + off);
  float4* v4 = reinterpret_cast<float4*>(v + off);
  constexpr int AU = 2;     // two pieces of each of the four streams in flight per thread
  for (long long j0 = threadIdx.x; j0 < n4; j0 += AU * 256) {
    float4 gq[AU], mq[AU], vq[AU], pq[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const long long ju = j0 + u * 256 < n4 ? j0 + u * 256 : j0;
      gq[u] = g4[ju]; mq[u] = m4[ju]; vq[u] = v4[ju]; pq[u] = p4[ju];
    }
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const long long j = j0 + u * 256;
      if (j >= n4) break;
      float* gp = reinterpret_cast<float*>(&gq[u]); float* mp = reinterpret_cast<float*>(&mq[u]);
      float* vp = reinterpret_cast<float*>(&vq[u]); float* pp = reinterpret_cast<float*>(&pq[u]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gi = gp[e];
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        gp[e] = gi;
        const float mi = mp[e] + (gi - mp[e]) * (1.f - beta1);
        const float vi = vp[e] * beta2 + (1.f - beta2) * gi * gi;
        mp[e] = mi; vp[e] = vi;
        const float denom = sqrtf(vi) / b2 + eps;
        pp[e] = pp[e] - ss * (mi / denom);
      }
      if (clip > 0.f) g4[j] = gq[u];
      m4[j] = mq[u]; v4[j] = vq[u]; p4[j] = pq[u];
    }
  }
  for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) {
    float gi = g[i];
    if (clip > 0.f) { gi = fminf(fmaxf(gi, -clip), clip); g[i] = gi; }
    // exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / b2 + eps;
    p[i] = p[i] - ss * (mi / denom);
  }
}

// The whole gradient-consumption tail of a stepping lesson in ONE launch (trainer/hw_with_style_trainer.py:379-391: clip_grad_value_ over every
// parameter with a gradient, the per-parameter NaN asserts, optimizer.step()): tensors with a gradient and NO parameter entry (touched, but
// owned by the optimizer that does not step in this lesson) are clipped only; tensors with a parameter entry are clipped and stepped, and
// `flag` is raised when a freshly written parameter value is not finite (a parameter can only become NaN / inf where it is written).
// The clipped gradient is stored back only where clipping changed it (the reference clips in place; almost no element is ever outside +-2).
__global__ __launch_bounds__(256) void mt_clip_adam_kernel(MtArgs a, const float* step_size, const float* bc2_sqrt, float beta1, float beta2, float eps,
                                                           float clip, int* flag) {
  const int t = a.chunk_tensor[blockIdx.x];
  float* g = reinterpret_cast<float*>(a.pb[t]);
  if (!g) return;
  float* p = reinterpret_cast<float*>(a.pa[t]);
  const long long off = a.chunk_off[blockIdx.x];
  const long long end = min(off + (long long)a.chunk, a.numel[t]);
  if (!p) {      // clip only
    const bool al = (reinterpret_cast<uintptr_t>(g + off) & 15) == 0;
    const long long n4 = al ? (end - off) >> 2 : 0;
    float4* g4 = reinterpret_cast<float4*>(g + off);
    for (long long j0 = threadIdx.x; j0 < n4; j0 += MT_U * 256) {
      float4 vv[MT_U];
#pragma unroll
      for (int u = 0; u < MT_U; ++u) vv[u] = g4[j0 + u * 256 < n4 ? j0 + u * 256 : j0];
#pragma unroll
      for (int u = 0; u < MT_U; ++u) {
        const long long j = j0 + u * 256;
        if (j >= n4) break;
        const float4 v = vv[u];
        const float4 c = make_float4(fminf(fmaxf(v.x, -clip), clip), fminf(fmaxf(v.y, -clip), clip), fminf(fmaxf(v.z, -clip), clip), fminf(fmaxf(v.w, -clip), clip));
        if (c.x != v.x || c.y != v.y || c.z != v.z || c.w != v.w) g4[j] = c;      // (NaN: fminf / fmaxf return the bound, as the standalone clip pass did)
      }
    }
    for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) {
      const float v = g[i], c = fminf(fmaxf(v, -clip), clip);
      if (c != v) g[i] = c;
    }
    return;
  }
  float* m = reinterpret_cast<float*>(a.pc[t]);
  float* v = reinterpret_cast<float*>(a.pd[t]);
  if (!m || !v) return;
  const float ss = step_size[t], b2 = bc2_sqrt[t];
  const bool al = ((reinterpret_cast<uintptr_t>(p + off) | reinterpret_cast<uintptr_t>(g + off) | reinterpret_cast<uintptr_t>(m + off) |
                    reinterpret_cast<uintptr_t>(v + off)) & 15) == 0;
  const long long n4 = al ? (end - off) >> 2 : 0;
  float4* p4 = reinterpret_cast<float4*>(p + off);
  float4* g4 = reinterpret_cast<float4*>(g + off);
  float4* m4 = reinterpret_cast<float4*>(m + off);
  float4* v4 = reinterpret_cast<float4*>(v + off);
  bool bad = false;
  constexpr int AU = 2;
  for (long long j0 = threadIdx.x; j0 < n4; j0 += AU * 256) {
    float4 gq[AU], mq[AU], vq[AU], pq[AU];
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const long long ju = j0 + u * 256 < n4 ? j0 + u * 256 : j0;
      gq[u] = g4[ju]; mq[u] = m4[ju]; vq[u] = v4[ju]; pq[u] = p4[ju];
    }
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const long long j = j0 + u * 256;
      if (j >= n4) break;
      float* gp = reinterpret_cast<float*>(&gq[u]); float* mp = reinterpret_cast<float*>(&mq[u]);
      float* vp = reinterpret_cast<float*>(&vq[u]); float* pp = reinterpret_cast<float*>(&pq[u]);
      bool changed = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gi = gp[e];
        const float gc = fminf(fmaxf(gi, -clip), clip);
        changed |= gc != gi;
        gi = gc;
        gp[e] = gi;
        const float mi = mp[e] + (gi - mp[e]) * (1.f - beta1);
        const float vi = vp[e] * beta2 + (1.f - beta2) * gi * gi;
        mp[e] = mi; vp[e] = vi;
        const float denom = sqrtf(vi) / b2 + eps;
        pp[e] = pp[e] - ss * (mi / denom);
        bad |= !isfinite(pp[e]);
      }
      if (changed) g4[j] = gq[u];
      m4[j] = mq[u]; v4[j] = vq[u]; p4[j] = pq[u];
    }
  }
  for (long long i = off + 4 * n4 + threadIdx.x; i < end; i += 256) {
    float gi = g[i];
    const float gc = fminf(fmaxf(gi, -clip), clip);
    if (gc != gi) g[i] = gc;
    gi = gc;
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / b2 + eps;
    const float pn = p[i] - ss * (mi / denom);
    p[i] = pn;
    bad |= !isfinite(pn);
  }
  if (bad && flag) atomicOr(flag, 1);
}

// ---------------- Philox4x32-10 (philox.h) ----------------
__device__ __forceinline__ void philox4(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) { hwg_philox4(seed, ctr, out); }
__device__ __forceinline__ float u01(uint32_t x) { return hwg_u01(x); }

__global__ void randn_kernel(float* out, long long n, uint64_t seed, uint64_t offset) {
  const long long n4 = (n + 3) / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 z = hwg_randn4(seed, offset + (uint64_t)i);
    const float v[4] = {z.x, z.y, z.z, z.w};
    for (int e = 0; e < 4; ++e)
      if (i * 4 + e < n) out[i * 4 + e] = v[e];
  }
}
// out[i] = (u >= p) ? 1/(1-p) : 0     (feature-dropout channel masks)
__global__ void dropmask_kernel(float* out, long long n, float p, uint64_t seed, uint64_t offset) {
  const long long n4 = (n + 3) / 4;
  const float keep = 1.f / (1.f - p);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    uint32_t r[4];
    philox4(seed, offset + (uint64_t)i, r);
    for (int e = 0; e < 4; ++e)
      if (i * 4 + e < n) out[i * 4 + e] = (u01(r[e]) >= p) ? keep : 0.f;
  }
}

// several feature-dropout masks of one network pass in ONE launch: segment j = elements [start_j, start_j+1) of `out` (starts are multiples
// of 4) with its own drop probability; element i takes uniform i % 4 of Philox counter offset + i / 4 - the values consecutive
// hwg_dropmask calls (each advancing the stream by n_j / 4 counters) would have written
constexpr int DROP_SEG_MAX = 16;
struct DropSegs { int n; long long end[DROP_SEG_MAX]; float p[DROP_SEG_MAX]; };
__global__ void dropmask_multi_kernel(float* out, long long n, DropSegs sg, uint64_t seed, uint64_t offset) {
  const long long n4 = n / 4;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    int j = 0;
    while (j + 1 < sg.n && i * 4 >= sg.end[j]) ++j;
    const float p = sg.p[j];
    const float keep = 1.f / (1.f - p);
    uint32_t r[4];
    philox4(seed, offset + (uint64_t)i, r);
    float4 v;
    v.x = (u01(r[0]) >= p) ? keep : 0.f; v.y = (u01(r[1]) >= p) ? keep : 0.f;
    v.z = (u01(r[2]) >= p) ? keep : 0.f; v.w = (u01(r[3]) >= p) ? keep : 0.f;
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

MtArgs make_mt(const void* pa, const void* pb, const void* pc, const void* pd, const void* numel, const void* ct, const void* co, int chunk) {
  MtArgs a;
  a.pa = (const long long*)pa; a.pb = (const long long*)pb; a.pc = (const long long*)pc; a.pd = (const long long*)pd;
  a.numel = (const long long*)numel; a.chunk_tensor = (const int*)ct; a.chunk_off = (const long long*)co; a.chunk = chunk;
  return a;
}

}  // namespace

extern "C" int hwg_mt_abs_sum(const void* ptrs, const void* numel, const void* chunk_tensor, const void* chunk_off, int nchunks, int chunk,
                              int nt, double* chunk_partials, double* out_sums, void* stream) {
  HWG_REQUIRE(ptrs && numel && chunk_tensor && chunk_off && chunk_partials && out_sums && nchunks > 0 && chunk > 0 && nt > 0, "mt_abs_sum: bad arguments");
  MtArgs a = make_mt(ptrs, nullptr, nullptr, nullptr, numel, chunk_tensor, chunk_off, chunk);
  if (hipMemsetAsync(chunk_partials, 0, (size_t)nchunks * sizeof(double), (hipStream_t)stream) != hipSuccess) {   // absent tensors leave theirs untouched
    hwg_set_error("mt_abs_sum: memset failed");
    return HWG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(mt_abs_sum_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, chunk_partials, nt);
  hipLaunchKernelGGL(mt_abs_sum_final_kernel, dim3(hwg_cdiv(nt, 256)), dim3(256), 0, (hipStream_t)stream, (const double*)chunk_partials,
                     (const int*)chunk_tensor, nchunks, nt, out_sums);
  HWG_LAUNCH_CHECK("mt_abs_sum");
  return HWG_OK;
}
extern "C" int hwg_mt_abs_sum_sets(const void* ptrs, int nsets, const void* numel, const void* chunk_tensor, const void* chunk_off, int nchunks,
                                   int chunk, int nt, double* chunk_partials, double* out_sums, void* stream) {
  HWG_REQUIRE(ptrs && numel && chunk_tensor && chunk_off && chunk_partials && out_sums && nchunks > 0 && chunk > 0 && nt > 0 && nsets > 0 &&
              nsets <= 64, "mt_abs_sum_sets: bad arguments");
  MtArgs a = make_mt(ptrs, nullptr, nullptr, nullptr, numel, chunk_tensor, chunk_off, chunk);
  if (hipMemsetAsync(chunk_partials, 0, (size_t)nsets * nchunks * sizeof(double), (hipStream_t)stream) != hipSuccess) {
    hwg_set_error("mt_abs_sum_sets: memset failed");
    return HWG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(mt_abs_sum_kernel, dim3(nchunks, nsets), dim3(256), 0, (hipStream_t)stream, a, chunk_partials, nt);
  hipLaunchKernelGGL(mt_abs_sum_final_kernel, dim3(hwg_cdiv(nt, 256), nsets), dim3(256), 0, (hipStream_t)stream, (const double*)chunk_partials,
                     (const int*)chunk_tensor, nchunks, nt, out_sums);
  HWG_LAUNCH_CHECK("mt_abs_sum_sets");
  return HWG_OK;
}
extern "C" int hwg_mt_balance_coef(const double* sumD, const double* sumR, const void* numel, const void* ptr_grad, const void* ptr_R,
                                   const float* xs, int nsets, int nt, float* coef, void* stream) {
  HWG_REQUIRE(sumD && sumR && numel && ptr_grad && ptr_R && xs && coef && nsets > 0 && nt > 0, "mt_balance_coef: bad arguments");
  hipLaunchKernelGGL(mt_balance_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sumD, sumR, (const long long*)numel,
                     (const long long*)ptr_grad, (const long long*)ptr_R, xs, nsets, nt, coef);
  HWG_LAUNCH_CHECK("mt_balance_coef");
  return HWG_OK;
}
extern "C" int hwg_mt_axpy(const void* ptrs_dst, const void* ptrs_src, const float* coef, const void* numel, const void* chunk_tensor,
                           const void* chunk_off, int nchunks, int chunk, void* stream) {
  HWG_REQUIRE(ptrs_dst && ptrs_src && numel && chunk_tensor && chunk_off && nchunks > 0 && chunk > 0, "mt_axpy: bad arguments");
  MtArgs a = make_mt(ptrs_dst, ptrs_src, nullptr, nullptr, numel, chunk_tensor, chunk_off, chunk);
  hipLaunchKernelGGL(mt_axpy_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, coef);
  HWG_LAUNCH_CHECK("mt_axpy");
  return HWG_OK;
}
extern "C" int hwg_mt_axpy_sets(const void* ptrs_dst, const void* ptrs_src, const float* coef, int nsets, int nt, const void* numel,
                                const void* chunk_tensor, const void* chunk_off, int nchunks, int chunk, void* stream) {
  HWG_REQUIRE(ptrs_dst && ptrs_src && coef && numel && chunk_tensor && chunk_off && nchunks > 0 && chunk > 0 && nt > 0 && nsets > 0 && nsets <= 8,
              "mt_axpy_sets: bad arguments (at most 8 sets)");
  MtArgs a = make_mt(ptrs_dst, ptrs_src, nullptr, nullptr, numel, chunk_tensor, chunk_off, chunk);
  hipLaunchKernelGGL(mt_axpy_sets_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, coef, nsets, nt);
  HWG_LAUNCH_CHECK("mt_axpy_sets");
  return HWG_OK;
}
extern "C" int hwg_mt_unary(const void* ptrs_a, const void* ptrs_b, int op, float c, int* flag, const void* numel, const void* chunk_tensor,
                            const void* chunk_off, int nchunks, int chunk, void* stream) {
  HWG_REQUIRE(ptrs_a && numel && chunk_tensor && chunk_off && nchunks > 0 && chunk > 0 && op >= 0 && op <= 4, "mt_unary: bad arguments");
  HWG_REQUIRE(op != 2 || flag, "mt_unary: nan scan needs a flag");
  MtArgs a = make_mt(ptrs_a, ptrs_b, nullptr, nullptr, numel, chunk_tensor, chunk_off, chunk);
  hipLaunchKernelGGL(mt_unary_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, op, c, flag);
  HWG_LAUNCH_CHECK("mt_unary");
  return HWG_OK;
}
extern "C" int hwg_mt_adam(const void* ptrs_p, const void* ptrs_g, const void* ptrs_m, const void* ptrs_v, const float* step_size,
                           const float* bc2_sqrt, float beta1, float beta2, float eps, float clip, const void* numel, const void* chunk_tensor,
                           const void* chunk_off, int nchunks, int chunk, void* stream) {
  HWG_REQUIRE(ptrs_p && ptrs_g && ptrs_m && ptrs_v && step_size && bc2_sqrt && numel && chunk_tensor && chunk_off && nchunks > 0 && chunk > 0,
              "mt_adam: bad arguments");
  MtArgs a = make_mt(ptrs_p, ptrs_g, ptrs_m, ptrs_v, numel, chunk_tensor, chunk_off, chunk);
  hipLaunchKernelGGL(mt_adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, step_size, bc2_sqrt, beta1, beta2, eps, clip);
  HWG_LAUNCH_CHECK("mt_adam");
  return HWG_OK;
}

extern "C" int hwg_mt_clip_adam(const void* ptrs_p, const void* ptrs_g, const void* ptrs_m, const void* ptrs_v, const float* step_size,
                                const float* bc2_sqrt, float beta1, float beta2, float eps, float clip, int* flag, const void* numel,
                                const void* chunk_tensor, const void* chunk_off, int nchunks, int chunk, void* stream) {
  HWG_REQUIRE(ptrs_p && ptrs_g && ptrs_m && ptrs_v && step_size && bc2_sqrt && numel && chunk_tensor && chunk_off && nchunks > 0 && chunk > 0 && clip > 0.f,
              "mt_clip_adam: bad arguments");
  MtArgs a = make_mt(ptrs_p, ptrs_g, ptrs_m, ptrs_v, numel, chunk_tensor, chunk_off, chunk);
  hipLaunchKernelGGL(mt_clip_adam_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, step_size, bc2_sqrt, beta1, beta2, eps, clip, flag);
  HWG_LAUNCH_CHECK("mt_clip_adam");
  return HWG_OK;
}

// ---- insert_spaces on the device (hw_with_style.py:302-328 with the device generator instead of numpy's): per character of every line
// blanks ~ round(N(count, count_std)), repeats ~ round(N(duplicates, dup_std)) (round half to even, negative -> 0); one Philox4x32 block per
// (line, character). Single workgroup: the whole plan is a few thousand elements.
__global__ __launch_bounds__(256) void insert_spaces_plan_kernel(const float* counts /*[L][B][2]*/, const int* lens_in, int L, int B, float count_std,
                                                                float dup_std, int count_duplicates, uint64_t seed, uint64_t offset,
                                                                int* reps /*[B][2L]*/, int* starts /*[B][L]*/, int* lens_max /*[B+1]*/) {
  __shared__ float smax[256];
  float m = -INFINITY;
  for (int i = threadIdx.x; i < L * B * 2; i += 256) m = fmaxf(m, counts[i]);
  smax[threadIdx.x] = m;
  for (int i = threadIdx.x; i < L * B; i += 256) {
    const int b = i / L, j = i % L;
    int r0 = 0, r1 = 0;
    if (j < lens_in[b]) {
      uint32_t r[4];
      philox4(seed, offset + (uint64_t)i, r);
      const float u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
      const float z0 = sqrtf(-2.f * logf(u0)) * cosf(6.2831853071795864f * u1);
      const float z1 = sqrtf(-2.f * logf(u2)) * cosf(6.2831853071795864f * u3);
      const float* c = counts + ((long long)j * B + b) * 2;
      r0 = max((int)rintf(c[0] + count_std * z0), 0);
      r1 = count_duplicates ? max((int)rintf(c[1] + dup_std * z1), 0) : 1;
    }
    reps[(long long)b * 2 * L + 2 * j] = r0;
    reps[(long long)b * 2 * L + 2 * j + 1] = r1;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += 256) {
    int pos = 0;
    for (int j = 0; j < lens_in[b]; ++j) {
      pos += reps[(long long)b * 2 * L + 2 * j];
      starts[b * L + j] = pos;
      pos += reps[(long long)b * 2 * L + 2 * j + 1];
    }
    lens_max[b] = pos;
  }
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) lens_max[B] = max((int)ceilf(smax[0]), 3);
}
// idx [T][B] (zero = blank, pre-cleared): the run of every character
__global__ __launch_bounds__(256) void insert_spaces_fill_kernel(const int* label /*[L][B]*/, const int* lens_in, const int* reps, const int* starts, int L,
                                                                int B, int T, int* idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L * B) return;
  const int b = i / L, j = i % L;
  if (j >= lens_in[b]) return;
  const int n = reps[(long long)b * 2 * L + 2 * j + 1], s = starts[b * L + j], c = label[j * B + b];
  for (int t = 0; t < n && s + t < T; ++t) idx[(long long)(s + t) * B + b] = c;
}

extern "C" int hwg_insert_spaces_plan(const float* counts, const int* label_lengths, int L, int B, float count_std, float dup_std, int count_duplicates,
                                      unsigned long long seed, unsigned long long offset, int* reps, int* starts, int* lens_max, void* stream) {
  HWG_REQUIRE(counts && label_lengths && reps && starts && lens_max && L > 0 && B > 0, "insert_spaces_plan: bad arguments");
  hipLaunchKernelGGL(insert_spaces_plan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, label_lengths, L, B, count_std, dup_std,
                     count_duplicates, (uint64_t)seed, (uint64_t)offset, reps, starts, lens_max);
  HWG_LAUNCH_CHECK("insert_spaces_plan");
  return HWG_OK;
}
extern "C" int hwg_insert_spaces_fill(const int* label, const int* label_lengths, const int* reps, const int* starts, int L, int B, int T, int* idx,
                                      void* stream) {
  HWG_REQUIRE(label && label_lengths && reps && starts && idx && L > 0 && B > 0 && T > 0, "insert_spaces_fill: bad arguments");
  hipLaunchKernelGGL(insert_spaces_fill_kernel, dim3(hwg_cdiv((long long)L * B, 256)), dim3(256), 0, (hipStream_t)stream, label, label_lengths, reps,
                     starts, L, B, T, idx);
  HWG_LAUNCH_CHECK("insert_spaces_fill");
  return HWG_OK;
}

extern "C" int hwg_randn(float* out, long long n, unsigned long long seed, unsigned long long offset, void* stream) {
  HWG_REQUIRE(out && n > 0, "randn: bad arguments");
  hipLaunchKernelGGL(randn_kernel, dim3(hwg_stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n, (uint64_t)seed, (uint64_t)offset);
  HWG_LAUNCH_CHECK("randn");
  return HWG_OK;
}
extern "C" int hwg_dropmask_multi(float* out, int nseg, const long long* seg_elems, const float* seg_p, unsigned long long seed,
                                  unsigned long long offset, void* stream) {
  HWG_REQUIRE(out && seg_elems && seg_p && nseg > 0 && nseg <= DROP_SEG_MAX, "dropmask_multi: bad arguments (at most %d segments)", DROP_SEG_MAX);
  DropSegs sg;
  sg.n = nseg;
  long long total = 0;
  for (int j = 0; j < nseg; ++j) {
    HWG_REQUIRE(seg_elems[j] > 0 && seg_elems[j] % 4 == 0 && seg_p[j] >= 0.f && seg_p[j] < 1.f, "dropmask_multi: segment %d: element count must be a positive multiple of 4, 0 <= p < 1", j);
    total += seg_elems[j];
    sg.end[j] = total; sg.p[j] = seg_p[j];
  }
  hipLaunchKernelGGL(dropmask_multi_kernel, dim3(hwg_stream_grid(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, total, sg, (uint64_t)seed,
                     (uint64_t)offset);
  HWG_LAUNCH_CHECK("dropmask_multi");
  return HWG_OK;
}
extern "C" int hwg_dropmask(float* out, long long n, float p, unsigned long long seed, unsigned long long offset, void* stream) {
  HWG_REQUIRE(out && n > 0 && p >= 0.f && p < 1.f, "dropmask: bad arguments");
  hipLaunchKernelGGL(dropmask_kernel, dim3(hwg_stream_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, out, n, p, (uint64_t)seed,
                     (uint64_t)offset);
  HWG_LAUNCH_CHECK("dropmask");
  return HWG_OK;
}
