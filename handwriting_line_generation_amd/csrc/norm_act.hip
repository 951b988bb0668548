// Normalisation + activation epilogues on NHWC tensors (HBM-bound streaming kernels).
//   InstanceNorm / GroupNorm / BatchNorm(train) share one three-stage pipeline:
//     partial moments per (n, pixel chunk, c)  ->  finalize per statistic group  ->  apply (+affine, channel mask, activation)
//   and the mirrored backward. The generator's NoiseInjection -> LeakyReLU -> AdaIN chain
//   (model/pure_gen.py:205-214) is fused into the first/last stage of the InstanceNorm pipeline.
// Moments are accumulated in fp64 across threads so var = E[x^2]-E[x]^2 is safe.
#include "hwg_common.h"
#include "philox.h"

namespace {

enum { MODE_IN = 0, MODE_GN = 1, MODE_BN = 2 };
constexpr int NU = 4;   // pixels in flight per thread in the streaming loops
#ifndef NORM_PASSES
#define NORM_PASSES 4   // sweeps of PP pixels per workgroup = one trip of the NU-deep loops (tools/ab_kernels.sh: 4 / 8 / 16 / 32 -> 1.46 / 1.51 / 1.67 / 2.13 ms of norm kernels per step)
#endif

struct Geo {
  int N, HW, C, C4, L, PP, chunks, cs;
};
Geo make_geo(int N, int HW, int C) {
  Geo g;
  g.N = N; g.HW = HW; g.C = C; g.C4 = C / 4;
  g.L = g.C4;                 // float4 lanes per pixel (<= 256)
  g.PP = 256 / g.L;           // pixels per pass
  // a workgroup sweeps PP pixels per pass; aim at NORM_PASSES passes per workgroup so that narrow-and-deep tensors (HWR tail: 126 pixels x 512
  // channels) still spread over the chip instead of 8 workgroups running 63 dependent passes each (37 us -> latency bound)
  int chunks = (HW + g.PP * NORM_PASSES - 1) / (g.PP * NORM_PASSES);
  if (chunks > 64) chunks = 64;
  while (chunks > 1 && (long long)chunks * N > 4096) chunks >>= 1;
  if (chunks < 1) chunks = 1;
  g.cs = (HW + chunks - 1) / chunks;
  g.chunks = (HW + g.cs - 1) / g.cs;
  return g;
}

// Reduce NV float4 accumulators over the pixel lanes of the block and write part[n][chunk][c][NV] (double).
// (Round 6 measured two variants and kept neither, profiles/r06_norm_variants.txt: 1024-thread workgroups on the large few-sample tensors
// - 3 % off the normalisation kernels' time in the step, the tensors stream at 4.6 TB/s in-step already - and fp32 partials in this LDS image
// instead of doubles: NOT the same arithmetic - with the double image the compiler folds the last fp32 accumulate of every lane into the
// fp64 conversion (contraction across the fpext), the fp32 image rounds it first; 1-ulp differences in rstd that flip a ReLU gate of the
// count-lesson recogniser path: 2.95e-6 -> 1.42e-4 from fp64, tests/test_pipeline_gpu.py.)
// COH (here and in the bodies below): the partial sums are read by OTHER workgroups of the same launch (fused kernels): they are written and read
// with agent-scope accesses (hwg_store_agent / hwg_load_agent, hwg_common.h) instead of plain ones - same values, same arithmetic
template <int NV, bool COH = false>
__device__ __forceinline__ void block_reduce_store(const float4 (&acc)[NV], const Geo& g, double* part, double* sm) {
  const int tid = threadIdx.x;
  const int cl = tid % g.L, pl = tid / g.L;
  const bool active = pl < g.PP;
  // sm layout [pl][c][NV]
  if (active) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      double* o = sm + ((size_t)pl * g.C + cl * 4) * NV + v;
      o[0 * NV] = acc[v].x; o[1 * NV] = acc[v].y; o[2 * NV] = acc[v].z; o[3 * NV] = acc[v].w;
    }
  }
  __syncthreads();
  double* po = part + (((size_t)blockIdx.y * g.chunks + blockIdx.x) * g.C) * NV;
  for (int i = tid; i < g.C * NV; i += 256) {
    double s = 0.0;
    for (int p = 0; p < g.PP; ++p) s += sm[(size_t)p * g.C * NV + i];
    if (COH) hwg_store_agent(po + i, s); else po[i] = s;
  }
}

// wait until *c reaches `expected` (fused launches below: all threads of the workgroup call it; thread 0 polls, then barrier + acquire fence);
// false after ~1 s without success
__device__ __forceinline__ bool grid_wait(int* c, int expected) {
  __shared__ int ok_flag;
  if (threadIdx.x == 0) {
    int spins = 0, ok = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1 << 20)) { ok = 0; break; }
    }
    ok_flag = ok;
  }
  __syncthreads();
  return ok_flag != 0;
}

// ---------------- forward stage 1: moments of x (optionally of u = lrelu(x + nw*noise), which is also written) -------------
// NOISE: 0 = moments of x, 1 = of u = lrelu(x + nw*noise) with the noise read from a tensor, 2 = the same with the noise DRAWN here: element
// i of the tensor takes normal i % 4 of Philox counter ctr0 + i / 4 (philox.h) - exactly the value hwg_randn(seed, offset = ctr0) would have
// written to a noise tensor, without the tensor (forward-only calls: nothing reads the noise again)
template <int NOISE, bool COH = false>
__device__ __forceinline__ void moments_fwd_body(const float* x, const Geo& g, double* part,
                                                 const float* noise, const float* nw, float nscale, float slope, float* u,
                                                 unsigned long long seed, unsigned long long ctr0) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  const int cl = tid % g.L, pl = tid / g.L;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * g.cs, p1 = min(p0 + g.cs, g.HW);
  float4 acc[2];
  acc[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  acc[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (NOISE && pl < g.PP) {
    w4 = *reinterpret_cast<const float4*>(nw + cl * 4);
    w4.x *= nscale; w4.y *= nscale; w4.z *= nscale; w4.w *= nscale;
  }
  if (pl < g.PP) {
    // NU pixels per trip, all their loads issued before the first is used: a workgroup makes only ~8 passes over its chunk, so with one
    // pixel per trip the kernel's time is 8 exposed memory latencies (the tensors of an 8-line batch are too small to hide them behind
    // other workgroups); pixels past the chunk re-read pixel p and are skipped. Same accumulation order as the one-pixel loop.
    for (int p = p0 + pl; p < p1; p += NU * g.PP) {
      float4 vv[NU], nn[NU];
      size_t offs[NU];
#pragma unroll
      for (int q = 0; q < NU; ++q) {
        const int pp = p + q * g.PP;
        offs[q] = ((size_t)n * g.HW + (pp < p1 ? pp : p)) * g.C + cl * 4;
        vv[q] = *reinterpret_cast<const float4*>(x + offs[q]);
        if (NOISE == 1) nn[q] = *reinterpret_cast<const float4*>(noise + offs[q]);
        if (NOISE == 2) nn[q] = hwg_randn4(seed, ctr0 + (offs[q] >> 2));
      }
#pragma unroll
      for (int q = 0; q < NU; ++q) {
        if (p + q * g.PP >= p1) break;
        float4 v = vv[q];
        if (NOISE) {
          const float4 nz = nn[q];
          v.x += w4.x * nz.x; v.y += w4.y * nz.y; v.z += w4.z * nz.z; v.w += w4.w * nz.w;
          v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
          v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
          *reinterpret_cast<float4*>(u + offs[q]) = v;
        }
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        acc[1].x += v.x * v.x; acc[1].y += v.y * v.y; acc[1].z += v.z * v.z; acc[1].w += v.w * v.w;
      }
    }
  }
  block_reduce_store<2, COH>(acc, g, part, sm);
}
template <int NOISE>
__global__ __launch_bounds__(256) void moments_fwd_kernel(const float* x, Geo g, double* part,
                                                          const float* noise, const float* nw, float nscale, float slope, float* u,
                                                          unsigned long long seed, unsigned long long ctr0) {
  moments_fwd_body<NOISE>(x, g, part, noise, nw, nscale, slope, u, seed, ctr0);
}

// ---------------- forward stage 2 (BatchNorm only): statistics per channel over all samples -> mean[N][C], rstd[N][C] ------------------
// one wavefront per channel, lanes sweep the N x chunks partial sums, wave reduction in fp64; also updates the running statistics.
// (per-sample normalisations - IN / GN / AdaIN - form their statistics inside the apply kernels, see sample_stats below)
__global__ __launch_bounds__(256) void finalize_fwd_kernel(const double* part, Geo g, float eps, float* mean, float* rstd,
                                                           float* running_mean, float* running_var, float momentum) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (w >= g.C) return;
  double s1 = 0.0, s2 = 0.0;
  const int items = g.N * g.chunks;
  for (int it = lane; it < items; it += 64) {
    const double* p = part + ((size_t)it * g.C + w) * 2;   // it == n*chunks + k
    s1 += p[0]; s2 += p[1];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  const double cnt = (double)g.N * g.HW;
  const double m = s1 / cnt;
  double var = s2 / cnt - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  for (int n = lane; n < g.N; n += 64) { mean[n * g.C + w] = (float)m; rstd[n * g.C + w] = r; }
  if (running_mean && lane == 0) {
    const double unb = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
    running_mean[w] = (1.f - momentum) * running_mean[w] + momentum * (float)m;
    running_var[w] = (1.f - momentum) * running_var[w] + momentum * (float)unb;
  }
}

// ---------------- statistics of one sample from the partial sums, computed inside the apply kernels (IN / GN) --------------------
// For per-sample normalisations every workgroup of the apply pass needs only its own sample's statistics: chunks x C partial pairs.
// Reading them here (L2 resident, just written) removes the separate finalize launches (latency bound, ~5-9 us each).
constexpr int NS_MAXC = 1024;
struct InlineStats {
  const double* part;   // nullptr: statistics come from the mean/rstd (c1/c2) arrays in memory
  int cpg;              // channels per statistic group (1 = instance norm)
  float eps;
  float* out_a;         // forward: mean [N][C], backward: dgamma [N][C] (per-sample affine) - written by chunk 0's workgroup
  float* out_b;         // forward: rstd,        backward: dbeta
  const float* gamma;   // backward only
  int per_sample;
  int accumulate;
};
template <bool COH>
__device__ __forceinline__ double2 load_pair(const double2* p) {
  if (!COH) return *p;
  const double* d = reinterpret_cast<const double*>(p);
  return make_double2(hwg_load_agent(d), hwg_load_agent(d + 1));
}
// forward: s_a = mean, s_b = rstd.  backward: s_a = c1 = mean_grp(g*gamma), s_b = c2 = mean_grp(g*gamma*xhat)
template <bool BWD, bool COH = false>
__device__ __forceinline__ void sample_stats(const InlineStats& is, const Geo& g, int n, float* s_a, float* s_b, double* s_t) {
  const int tid = threadIdx.x;
  for (int c = tid; c < g.C; c += 256) {
    double s1 = 0.0, s2 = 0.0;
    // every workgroup of the apply pass starts here: up to 64 chunk partials per channel, eight loads in flight (one at a time this chain
    // was a third of the kernel on the step's 5-50 MB tensors), added in chunk order
    const double2* pp = reinterpret_cast<const double2*>(is.part) + (size_t)n * g.chunks * g.C + c;
    int k = 0;
    for (; k + 8 <= g.chunks; k += 8) {
      double2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = load_pair<COH>(pp + (size_t)(k + u) * g.C);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += v[u].x; s2 += v[u].y; }
    }
    for (; k < g.chunks; ++k) { const double2 v = load_pair<COH>(pp + (size_t)k * g.C); s1 += v.x; s2 += v.y; }
    if (BWD) {
      if (blockIdx.x == 0 && is.per_sample) {
        const int idx = n * g.C + c;
        if (is.out_a) is.out_a[idx] = (is.accumulate ? is.out_a[idx] : 0.f) + (float)s2;
        if (is.out_b) is.out_b[idx] = (is.accumulate ? is.out_b[idx] : 0.f) + (float)s1;
      }
      const double gm = is.gamma ? (double)is.gamma[is.per_sample ? n * g.C + c : c] : 1.0;
      s1 *= gm; s2 *= gm;
    }
    s_t[2 * c] = s1; s_t[2 * c + 1] = s2;
  }
  __syncthreads();
  const int ngrp = g.C / is.cpg;
  const double cnt = (double)g.HW * is.cpg;
  for (int gi = tid; gi < ngrp; gi += 256) {
    double s1 = 0.0, s2 = 0.0;
    for (int cc = 0; cc < is.cpg; ++cc) { s1 += s_t[2 * (gi * is.cpg + cc)]; s2 += s_t[2 * (gi * is.cpg + cc) + 1]; }
    float a, b;
    if (BWD) {
      a = (float)(s1 / cnt); b = (float)(s2 / cnt);
    } else {
      const double m = s1 / cnt;
      double var = s2 / cnt - m * m;
      if (var < 0.0) var = 0.0;
      a = (float)m; b = (float)(1.0 / sqrt(var + (double)is.eps));
    }
    for (int cc = 0; cc < is.cpg; ++cc) { s_a[gi * is.cpg + cc] = a; s_b[gi * is.cpg + cc] = b; }
  }
  __syncthreads();
  if (!BWD && blockIdx.x == 0) {
    for (int c = tid; c < g.C; c += 256) { is.out_a[n * g.C + c] = s_a[c]; is.out_b[n * g.C + c] = s_b[c]; }
  }
}

// pre-activation value of the forward pass, z = mask * (gamma * xhat + beta), as one fixed sequence of rounded operations: the backward
// kernels recompute it from x (which they read anyway) to get the activation's gate instead of reading y - relu / leaky relu only need
// the sign of z - and for that the two sides must agree to the last bit, whatever the compiler would contract where.
__device__ __forceinline__ float norm_pre(float v, float m, float r, float g, float b, float k) {
  return __fmul_rn(k, __fmaf_rn(g, __fmul_rn(__fsub_rn(v, m), r), b));
}

// ---------------- forward stage 3: y = act(mask * (gamma * xhat + beta)) -----------------------------------------------
template <bool COH = false>
__device__ __forceinline__ void apply_fwd_body(const float* x, float* y, const Geo& g, const float* mean, const float* rstd,
                                               const float* gamma, const float* beta, int per_sample,
                                               const float* mask, int act, float slope, const InlineStats& is) {
  __shared__ __attribute__((aligned(16))) float s_a[NS_MAXC];
  __shared__ __attribute__((aligned(16))) float s_b[NS_MAXC];
  __shared__ double s_t[2 * NS_MAXC];
  const int tid = threadIdx.x;
  const int cl = tid % g.L, pl = tid / g.L;
  const int n = blockIdx.y;
  if (is.part) sample_stats<false, COH>(is, g, n, s_a, s_b, s_t);
  if (pl >= g.PP) return;
  const int p0 = blockIdx.x * g.cs, p1 = min(p0 + g.cs, g.HW);
  const int c = cl * 4;
  const float4 m4 = is.part ? *reinterpret_cast<const float4*>(s_a + c) : *reinterpret_cast<const float4*>(mean + n * g.C + c);
  const float4 r4 = is.part ? *reinterpret_cast<const float4*>(s_b + c) : *reinterpret_cast<const float4*>(rstd + n * g.C + c);
  float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ao = per_sample ? n * g.C + c : c;
  if (gamma) g4 = *reinterpret_cast<const float4*>(gamma + ao);
  if (beta) b4 = *reinterpret_cast<const float4*>(beta + ao);
  float4 k4 = make_float4(1.f, 1.f, 1.f, 1.f);
  if (mask) k4 = *reinterpret_cast<const float4*>(mask + n * g.C + c);
  // y = act(k*(g*(x-m)*r + b)) = act(x*sc + sh)
  float4 sc, sh;
  sc.x = k4.x * g4.x * r4.x; sh.x = k4.x * (b4.x - g4.x * r4.x * m4.x);
  sc.y = k4.y * g4.y * r4.y; sh.y = k4.y * (b4.y - g4.y * r4.y * m4.y);
  sc.z = k4.z * g4.z * r4.z; sh.z = k4.z * (b4.z - g4.z * r4.z * m4.z);
  sc.w = k4.w * g4.w * r4.w; sh.w = k4.w * (b4.w - g4.w * r4.w * m4.w);
  for (int p = p0 + pl; p < p1; p += NU * g.PP) {
    float4 vv[NU];
    size_t offs[NU];
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      const int pp = p + q * g.PP;
      offs[q] = ((size_t)n * g.HW + (pp < p1 ? pp : p)) * g.C + c;
      vv[q] = *reinterpret_cast<const float4*>(x + offs[q]);
    }
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      if (p + q * g.PP >= p1) break;
      const float4 v = vv[q];
      float4 o;
      // keep the reference's operation order (normalise, then affine) for rounding parity
      o.x = act_apply(norm_pre(v.x, m4.x, r4.x, g4.x, b4.x, k4.x), act, slope);
      o.y = act_apply(norm_pre(v.y, m4.y, r4.y, g4.y, b4.y, k4.y), act, slope);
      o.z = act_apply(norm_pre(v.z, m4.z, r4.z, g4.z, b4.z, k4.z), act, slope);
      o.w = act_apply(norm_pre(v.w, m4.w, r4.w, g4.w, b4.w, k4.w), act, slope);
      *reinterpret_cast<float4*>(y + offs[q]) = o;
    }
    (void)sc; (void)sh;
  }
}
__global__ __launch_bounds__(256) void apply_fwd_kernel(const float* x, float* y, Geo g, const float* mean, const float* rstd,
                                                        const float* gamma, const float* beta, int per_sample,
                                                        const float* mask, int act, float slope, InlineStats is) {
  apply_fwd_body(x, y, g, mean, rstd, gamma, beta, per_sample, mask, act, slope, is);
}

// ---------------- backward stage 1: partial sums of g and g*xhat, g = dy * act'(y) * mask ---------------------------------
// `y` == null with act = relu / leaky relu: the gate comes from the recomputed pre-activation (gamma, beta as in the forward call)
template <bool COH = false>
__device__ __forceinline__ void moments_bwd_body(const float* dy, const float* x, const float* y, const Geo& g, double* part,
                                                 const float* mean, const float* rstd, const float* mask, int act, float slope,
                                                 const float* gamma, const float* beta, int per_sample) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int tid = threadIdx.x;
  const int cl = tid % g.L, pl = tid / g.L;
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * g.cs, p1 = min(p0 + g.cs, g.HW);
  float4 acc[2];
  acc[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  acc[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < g.PP) {
    const int c = cl * 4;
    const float4 m4 = *reinterpret_cast<const float4*>(mean + n * g.C + c);
    const float4 r4 = *reinterpret_cast<const float4*>(rstd + n * g.C + c);
    float4 k4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) k4 = *reinterpret_cast<const float4*>(mask + n * g.C + c);
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) g4 = *reinterpret_cast<const float4*>(gamma + (per_sample ? n * g.C + c : c));
    if (beta) b4 = *reinterpret_cast<const float4*>(beta + (per_sample ? n * g.C + c : c));
    for (int p = p0 + pl; p < p1; p += NU * g.PP) {
     float4 dd[NU], vv[NU], yy[NU];
#pragma unroll
     for (int q = 0; q < NU; ++q) {
       const int pp = p + q * g.PP;
       const size_t off = ((size_t)n * g.HW + (pp < p1 ? pp : p)) * g.C + c;
       dd[q] = *reinterpret_cast<const float4*>(dy + off);
       vv[q] = *reinterpret_cast<const float4*>(x + off);
       if (act != 0 && y) yy[q] = *reinterpret_cast<const float4*>(y + off);
     }
#pragma unroll
     for (int q = 0; q < NU; ++q) {
      if (p + q * g.PP >= p1) break;
      float4 d = dd[q];
      const float4 v = vv[q];
      if (act != 0) {
        float4 o;
        if (y) {
          o = yy[q];
        } else {
          o.x = norm_pre(v.x, m4.x, r4.x, g4.x, b4.x, k4.x); o.y = norm_pre(v.y, m4.y, r4.y, g4.y, b4.y, k4.y);
          o.z = norm_pre(v.z, m4.z, r4.z, g4.z, b4.z, k4.z); o.w = norm_pre(v.w, m4.w, r4.w, g4.w, b4.w, k4.w);
        }
        d.x *= act_grad_from_out(o.x, act, slope); d.y *= act_grad_from_out(o.y, act, slope);
        d.z *= act_grad_from_out(o.z, act, slope); d.w *= act_grad_from_out(o.w, act, slope);
      }
      d.x *= k4.x; d.y *= k4.y; d.z *= k4.z; d.w *= k4.w;
      acc[0].x += d.x; acc[0].y += d.y; acc[0].z += d.z; acc[0].w += d.w;
      acc[1].x += d.x * ((v.x - m4.x) * r4.x); acc[1].y += d.y * ((v.y - m4.y) * r4.y);
      acc[1].z += d.z * ((v.z - m4.z) * r4.z); acc[1].w += d.w * ((v.w - m4.w) * r4.w);
     }
    }
  }
  block_reduce_store<2, COH>(acc, g, part, sm);
}
__global__ __launch_bounds__(256) void moments_bwd_kernel(const float* dy, const float* x, const float* y, Geo g, double* part,
                                                          const float* mean, const float* rstd, const float* mask, int act, float slope,
                                                          const float* gamma, const float* beta, int per_sample) {
  moments_bwd_body(dy, x, y, g, part, mean, rstd, mask, act, slope, gamma, beta, per_sample);
}

// ---------------- backward stage 2 (BatchNorm only): coefficients c1,c2 [N][C] and parameter gradients --------------------------------
// dx = rstd * (g*gamma - c1 - xhat*c2),  c1 = mean(g*gamma), c2 = mean(g*gamma*xhat) over (N,H,W); one wavefront per channel
__global__ __launch_bounds__(256) void finalize_bwd_kernel(const double* part, Geo g, const float* gamma, float* c1, float* c2, float* dgamma,
                                                           float* dbeta, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (w >= g.C) return;
  double s1 = 0.0, s2 = 0.0;
  const int items = g.N * g.chunks;
  for (int it = lane; it < items; it += 64) {
    const double* p = part + ((size_t)it * g.C + w) * 2;
    s1 += p[0]; s2 += p[1];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  const double cnt = (double)g.N * g.HW;
  const double gm = gamma ? (double)gamma[w] : 1.0;
  for (int n = lane; n < g.N; n += 64) { c1[n * g.C + w] = (float)(gm * s1 / cnt); c2[n * g.C + w] = (float)(gm * s2 / cnt); }
  if (lane == 0) {
    if (dgamma) dgamma[w] = (accumulate ? dgamma[w] : 0.f) + (float)s2;
    if (dbeta) dbeta[w] = (accumulate ? dbeta[w] : 0.f) + (float)s1;
  }
}

// per-channel parameter gradients for IN/GN with shared (per-channel) affine: sum over n and chunks, one wavefront per channel
__global__ __launch_bounds__(256) void param_grad_kernel(const double* part, Geo g, float* dgamma, float* dbeta, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (c >= g.C) return;
  double s1 = 0.0, s2 = 0.0;
  const int items = g.N * g.chunks;
  for (int it = lane; it < items; it += 64) {
    const double* p = part + ((size_t)it * g.C + c) * 2;
    s1 += p[0]; s2 += p[1];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (lane == 0) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
  }
}

// ---------------- backward stage 3 -----------------------------------------------------------------------------------------
// PRE != 0 additionally back-propagates through u = lrelu(x + nw*noise): dt = du * lrelu'(u); writes dx = dt and
// accumulates per-channel partials of dt (conv bias grad) and dt*noise (noise weight grad) into part2[n][chunk][c][2].
// `all_arrived` (fused launch only): counter that reaches gridDim.x * gridDim.y once EVERY workgroup's moment partials are in memory - the first
// sample's workgroups wait for it before they sum the partials of all samples
template <bool PRE, bool COH = false>
__device__ __forceinline__ void apply_bwd_body(const float* dy, const float* x, const float* y, float* dx, const Geo& g,
                                               const float* mean, const float* rstd, const float* gamma, int per_sample,
                                               const float* c1, const float* c2, const float* mask, int act, float slope,
                                               const float* noise, float pre_slope, double* part2, const InlineStats& is, const float* beta,
                                               float* pg_gamma, float* pg_beta, int pg_accumulate, int* all_arrived) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ __attribute__((aligned(16))) float s_a[NS_MAXC];
  __shared__ __attribute__((aligned(16))) float s_b[NS_MAXC];
  __shared__ double s_t[2 * NS_MAXC];
  const int tid = threadIdx.x;
  const int cl = tid % g.L, pl = tid / g.L;
  const int n = blockIdx.y;
  if (is.part) sample_stats<true, COH>(is, g, n, s_a, s_b, s_t);
  const int p0 = blockIdx.x * g.cs, p1 = min(p0 + g.cs, g.HW);
  float4 acc[2];
  acc[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  acc[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pl < g.PP) {
    const int c = cl * 4;
    const float4 m4 = *reinterpret_cast<const float4*>(mean + n * g.C + c);
    const float4 r4 = *reinterpret_cast<const float4*>(rstd + n * g.C + c);
    const float4 a4 = is.part ? *reinterpret_cast<const float4*>(s_a + c) : *reinterpret_cast<const float4*>(c1 + n * g.C + c);
    const float4 b4 = is.part ? *reinterpret_cast<const float4*>(s_b + c) : *reinterpret_cast<const float4*>(c2 + n * g.C + c);
    float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (gamma) g4 = *reinterpret_cast<const float4*>(gamma + (per_sample ? n * g.C + c : c));
    float4 k4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask) k4 = *reinterpret_cast<const float4*>(mask + n * g.C + c);
    float4 be4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (beta) be4 = *reinterpret_cast<const float4*>(beta + (per_sample ? n * g.C + c : c));
    for (int p = p0 + pl; p < p1; p += NU * g.PP) {
     float4 dd[NU], vv[NU], yy[NU], nn[NU];
     size_t offs[NU];
#pragma unroll
     for (int q = 0; q < NU; ++q) {
       const int pp = p + q * g.PP;
       offs[q] = ((size_t)n * g.HW + (pp < p1 ? pp : p)) * g.C + c;
       dd[q] = *reinterpret_cast<const float4*>(dy + offs[q]);
       vv[q] = *reinterpret_cast<const float4*>(x + offs[q]);
       if (act != 0 && y) yy[q] = *reinterpret_cast<const float4*>(y + offs[q]);
       if (PRE) nn[q] = *reinterpret_cast<const float4*>(noise + offs[q]);
     }
#pragma unroll
     for (int q = 0; q < NU; ++q) {
      if (p + q * g.PP >= p1) break;
      const size_t off = offs[q];
      float4 d = dd[q];
      const float4 v = vv[q];
      if (act != 0) {
        float4 z;
        if (y) {
          z = yy[q];
        } else {     // gate from the recomputed pre-activation (see norm_pre)
          z.x = norm_pre(v.x, m4.x, r4.x, g4.x, be4.x, k4.x); z.y = norm_pre(v.y, m4.y, r4.y, g4.y, be4.y, k4.y);
          z.z = norm_pre(v.z, m4.z, r4.z, g4.z, be4.z, k4.z); z.w = norm_pre(v.w, m4.w, r4.w, g4.w, be4.w, k4.w);
        }
        d.x *= act_grad_from_out(z.x, act, slope); d.y *= act_grad_from_out(z.y, act, slope);
        d.z *= act_grad_from_out(z.z, act, slope); d.w *= act_grad_from_out(z.w, act, slope);
      }
      float4 o;
      o.x = r4.x * (d.x * k4.x * g4.x - a4.x - ((v.x - m4.x) * r4.x) * b4.x);
      o.y = r4.y * (d.y * k4.y * g4.y - a4.y - ((v.y - m4.y) * r4.y) * b4.y);
      o.z = r4.z * (d.z * k4.z * g4.z - a4.z - ((v.z - m4.z) * r4.z) * b4.z);
      o.w = r4.w * (d.w * k4.w * g4.w - a4.w - ((v.w - m4.w) * r4.w) * b4.w);
      if (PRE) {
        // x here is u (the lrelu output that was normalised)
        o.x *= v.x > 0.f ? 1.f : pre_slope; o.y *= v.y > 0.f ? 1.f : pre_slope;
        o.z *= v.z > 0.f ? 1.f : pre_slope; o.w *= v.w > 0.f ? 1.f : pre_slope;
        const float4 nz = nn[q];
        acc[0].x += o.x; acc[0].y += o.y; acc[0].z += o.z; acc[0].w += o.w;
        acc[1].x += o.x * nz.x; acc[1].y += o.y * nz.y; acc[1].z += o.z * nz.z; acc[1].w += o.w * nz.w;
      }
      *reinterpret_cast<float4*>(dx + off) = o;
     }
    }
  }
  if (PRE) block_reduce_store<2>(acc, g, part2, sm);
  // parameter gradients of a shared (per-channel) affine, IN / GN: sum of the moment partials over samples and chunks. It was a launch of its
  // own between the two passes (param_grad_kernel, 11 launches per training step); now the first sample's workgroups do it behind their
  // streaming work, one wavefront per channel, in the same order (lane-strided items, butterfly) - bit-identical.
  if (!PRE && (pg_gamma || pg_beta) && blockIdx.y == 0 && is.part) {
    if (all_arrived) grid_wait(all_arrived, gridDim.x * gridDim.y);
    const int lane = tid & 63, wv = tid >> 6;
    const int items = g.N * g.chunks;
    for (int c = blockIdx.x * 4 + wv; c < g.C; c += gridDim.x * 4) {
      double s1 = 0.0, s2 = 0.0;
      for (int it = lane; it < items; it += 64) {
        const double2 v = load_pair<COH>(reinterpret_cast<const double2*>(is.part) + (size_t)it * g.C + c);
        s1 += v.x; s2 += v.y;
      }
      s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
      if (lane == 0) {
        if (pg_gamma) pg_gamma[c] = (pg_accumulate ? pg_gamma[c] : 0.f) + (float)s2;
        if (pg_beta) pg_beta[c] = (pg_accumulate ? pg_beta[c] : 0.f) + (float)s1;
      }
    }
  }
}
template <bool PRE>
__global__ __launch_bounds__(256) void apply_bwd_kernel(const float* dy, const float* x, const float* y, float* dx, Geo g,
                                                        const float* mean, const float* rstd, const float* gamma, int per_sample,
                                                        const float* c1, const float* c2, const float* mask, int act, float slope,
                                                        const float* noise, float pre_slope, double* part2, InlineStats is, const float* beta,
                                                        float* pg_gamma, float* pg_beta, int pg_accumulate) {
  apply_bwd_body<PRE>(dy, x, y, dx, g, mean, rstd, gamma, per_sample, c1, c2, mask, act, slope, noise, pre_slope, part2, is, beta, pg_gamma, pg_beta,
                      pg_accumulate, nullptr);
}

// ---------------- per-sample normalisations in ONE launch per direction ------------------------------------------------------------------------
// The moments pass and the apply pass of IN / GN / AdaIN have the same grid (chunk, sample), and a workgroup of the apply pass needs exactly the
// partial sums of its own sample's <= 64 workgroups. The fused kernels run the two bodies above back to back in one launch - same code, same
// partial-sum layout and order: bit-identical - with a barrier over the SAMPLE's workgroups in between (arrival counter per sample in the
// stream's counter buffer, hwg_split_counters): a launch and a round trip through the command processor less per normalisation and direction,
// and the second pass reads what this very workgroup streamed a few microseconds earlier.
// Waiting inside a kernel needs the workgroups waited for to get onto the chip: workgroups are dispatched in linear order (chunk fastest), so when
// a workgroup of sample n waits, everything ahead of it in the queue belongs to samples <= n - at most 63 workgroups of its own sample, the
// earlier samples' finish without waiting for anybody - and one sample's workgroups always fit (<= 64 of >= 1024 resident). The first sample's
// workgroups of the backward kernel also wait - at their very end - for ALL workgroups' partials (shared-affine parameter gradients): the same
// argument with 64 waiting workgroups. BatchNorm needs every workgroup to wait for every other (grids of 512 on several streams at once): not
// fused. A wait gives up after ~1 s (the sample's statistics are then poisoned with NaN - the trainer's non-finite check fires) rather than hang.
// Counters: [2n] arrivals, [2n + 1] departures of sample n; [2N], [2N + 1] the same over the whole grid. The last workgroup to depart sets both
// back to zero (launches of one stream are ordered, the buffer is per stream).
__device__ __forceinline__ void grid_arrive(int* a, int* b) {
  hwg_stores_done();
  __syncthreads();
  if (threadIdx.x == 0) { atomicAdd(a, 1); if (b) atomicAdd(b, 1); }
}
__device__ __forceinline__ void grid_depart(int* arrive, int* depart, int expected) {
  if (threadIdx.x == 0 && atomicAdd(depart, 1) == expected - 1) { atomicExch(arrive, 0); atomicExch(depart, 0); }
}

template <int NOISE>
__global__ __launch_bounds__(256) void norm_fwd_fused_kernel(const float* x, float* y, Geo g, double* part, const float* noise, const float* nw,
                                                             float nscale, float nslope, float* u, unsigned long long seed, unsigned long long ctr0,
                                                             const float* gamma, const float* beta, int per_sample, const float* mask, int act,
                                                             float slope, InlineStats is, int* sync) {
  moments_fwd_body<NOISE, true>(x, g, part, noise, nw, nscale, nslope, u, seed, ctr0);
  int* mine = sync + 2 * blockIdx.y;
  grid_arrive(mine, nullptr);
  const bool ok = grid_wait(mine, gridDim.x);
  apply_fwd_body<true>(NOISE ? (const float*)u : x, y, g, is.out_a, is.out_b, gamma, beta, per_sample, mask, act, slope, is);
  if (!ok && blockIdx.x == 0) { __syncthreads(); for (int c = threadIdx.x; c < g.C; c += 256) is.out_b[blockIdx.y * g.C + c] = __builtin_nanf(""); }
  grid_depart(mine, mine + 1, gridDim.x);
}

template <bool PRE>
__global__ __launch_bounds__(256) void norm_bwd_fused_kernel(const float* dy, const float* x, const float* y, float* dx, Geo g, double* part,
                                                             const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                             int per_sample, const float* c1, const float* c2, const float* mask, int act,
                                                             float slope, const float* noise, float pre_slope, double* part2, InlineStats is,
                                                             float* pg_gamma, float* pg_beta, int pg_accumulate, int* sync) {
  // (the moments pass of the generator epilogue's backward takes no gate, mask or affine: hwg_adain_bwd)
  if (PRE) moments_bwd_body<true>(dy, x, nullptr, g, part, mean, rstd, nullptr, 0, 0.f, nullptr, nullptr, 0);
  else moments_bwd_body<true>(dy, x, y, g, part, mean, rstd, mask, act, slope, gamma, beta, per_sample);
  int* mine = sync + 2 * blockIdx.y;
  int* all = sync + 2 * gridDim.y;
  const bool fold = !PRE && (pg_gamma || pg_beta);
  grid_arrive(mine, fold ? all : nullptr);
  const bool ok = grid_wait(mine, gridDim.x);
  apply_bwd_body<PRE, true>(dy, x, y, dx, g, mean, rstd, gamma, per_sample, c1, c2, mask, act, slope, noise, pre_slope, part2, is, beta, pg_gamma,
                            pg_beta, pg_accumulate, fold ? all : nullptr);
  if (!ok) { __syncthreads(); for (long long i = threadIdx.x; i < 4; i += 256) dx[i] = __builtin_nanf(""); }
  grid_depart(mine, mine + 1, gridDim.x);
  if (fold) grid_depart(all, all + 1, gridDim.x * gridDim.y);
}

// dbias[c] = sum part2[...][c][0];  dnoise_w[c] = nscale * sum part2[...][c][1]; one wavefront per channel
__global__ __launch_bounds__(256) void adain_param_grad_kernel(const double* part2, Geo g, float nscale, float* dbias, float* dnw, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (c >= g.C) return;
  double s1 = 0.0, s2 = 0.0;
  const int items = g.N * g.chunks;
  for (int it = lane; it < items; it += 64) {
    const double* p = part2 + ((size_t)it * g.C + c) * 2;
    s1 += p[0]; s2 += p[1];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (lane == 0) {
    if (dbias) dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)s1;
    if (dnw) dnw[c] = (accumulate ? dnw[c] : 0.f) + (float)(s2 * (double)nscale);
  }
}

// ---------------- plain elementwise: y = act(mask * (x + bias)) and its backward ----------------------------------------------
// IDX: the element needs its channel (bias) / sample (mask) coordinates; without either (the ReLU-only passes) no index arithmetic at all.
// 32-bit indices (the entry points check the size), two 16-byte pieces per stream in flight per thread.
template <bool IDX>
__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ mask,
                                                           float* __restrict__ y, unsigned total4, int HW, int C, int act, float slope) {
  const unsigned C4 = C / 4, stride = gridDim.x * blockDim.x;
  for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += 2 * stride) {
    const unsigned i1 = i0 + stride < total4 ? i0 + stride : i0;
    float4 v[2] = {reinterpret_cast<const float4*>(x)[i0], reinterpret_cast<const float4*>(x)[i1]};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned i = u ? i1 : i0;
      if (u && i1 == i0) break;
      float4 w = v[u];
      if (IDX) {
        const unsigned row = i / C4, c = (i - row * C4) * 4;
        if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + c); w.x += b.x; w.y += b.y; w.z += b.z; w.w += b.w; }
        if (mask) {
          const float4 k = *reinterpret_cast<const float4*>(mask + (size_t)(row / HW) * C + c);
          w.x *= k.x; w.y *= k.y; w.z *= k.z; w.w *= k.w;
        }
      }
      w.x = act_apply(w.x, act, slope); w.y = act_apply(w.y, act, slope);
      w.z = act_apply(w.z, act, slope); w.w = act_apply(w.w, act, slope);
      reinterpret_cast<float4*>(y)[i] = w;
    }
  }
}
template <bool IDX>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ mask,
                                                           float* __restrict__ dx, unsigned total4, int HW, int C, int act, float slope) {
  const unsigned C4 = C / 4, stride = gridDim.x * blockDim.x;
  for (unsigned i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total4; i0 += 2 * stride) {
    const unsigned i1 = i0 + stride < total4 ? i0 + stride : i0;
    float4 d[2] = {reinterpret_cast<const float4*>(dy)[i0], reinterpret_cast<const float4*>(dy)[i1]};
    float4 o[2];
    if (act != 0) { o[0] = reinterpret_cast<const float4*>(y)[i0]; o[1] = reinterpret_cast<const float4*>(y)[i1]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned i = u ? i1 : i0;
      if (u && i1 == i0) break;
      float4 w = d[u];
      if (act != 0) {
        w.x *= act_grad_from_out(o[u].x, act, slope); w.y *= act_grad_from_out(o[u].y, act, slope);
        w.z *= act_grad_from_out(o[u].z, act, slope); w.w *= act_grad_from_out(o[u].w, act, slope);
      }
      if (IDX) {
        const unsigned row = i / C4, c = (i - row * C4) * 4;
        const float4 k = *reinterpret_cast<const float4*>(mask + (size_t)(row / HW) * C + c);
        w.x *= k.x; w.y *= k.y; w.z *= k.z; w.w *= k.w;
      }
      reinterpret_cast<float4*>(dx)[i] = w;
    }
  }
}
// scalar-tail versions for C % 4 != 0 (tiny tensors only)
__global__ void bias_act_fwd_scalar_kernel(const float* x, const float* bias, const float* mask, float* y, long long rows, int HW, int C,
                                           int act, float slope) {
  const long long total = rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    float v = x[i];
    if (bias) v += bias[c];
    if (mask) v *= mask[(row / HW) * C + c];
    y[i] = act_apply(v, act, slope);
  }
}
__global__ void bias_act_bwd_scalar_kernel(const float* dy, const float* y, const float* mask, float* dx, long long rows, int HW, int C,
                                           int act, float slope) {
  const long long total = rows * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    float d = dy[i];
    if (act != 0) d *= act_grad_from_out(y[i], act, slope);
    if (mask) d *= mask[(row / HW) * C + c];
    dx[i] = d;
  }
}

// frozen (eval-mode BatchNorm) statistics expanded to [N][C]
__global__ void frozen_stats_kernel(const float* rm, const float* rv, float eps, int N, int C, float* mean, float* rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int c = i % C;
  mean[i] = rm[c];
  rstd[i] = 1.f / sqrtf(rv[c] + eps);
}

int check_geo(int N, int HW, int C, const char* who) {
  HWG_REQUIRE(N > 0 && HW > 0 && C > 0, "%s: non-positive size", who);
  HWG_REQUIRE(C % 4 == 0 && C <= 1024, "%s: C must be a multiple of 4 and <= 1024 (C=%d)", who, C);
  return HWG_OK;
}
size_t part_bytes(const Geo& g) { return (size_t)g.N * g.chunks * g.C * 2 * sizeof(double); }
size_t red_smem(const Geo& g) { return (size_t)g.PP * g.C * 2 * sizeof(double); }

}  // namespace

// counter buffer of a fused launch (norm_*_fused_kernel), nullptr = run the two-launch path
static int* fused_sync(int N, hipStream_t st) {
  if (!hwg_tune().norm_fused || 2 * N + 2 > HWG_SPLIT_COUNTERS) return nullptr;
  return hwg_split_counters(st);
}

extern "C" size_t hwg_norm_workspace(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0 || C % 4) return 0;
  Geo g = make_geo(N, HW, C);
  // two partial buffers (moments + adain param partials) and c1/c2 coefficient arrays
  return 2 * part_bytes(g) + 2 * (size_t)N * C * sizeof(float) + 256;
}

extern "C" int hwg_norm_fwd(const float* x, float* y, int N, int HW, int C, int mode, int groups, float eps,
                            const float* gamma, const float* beta, int affine_per_sample, const float* chan_mask,
                            int act, float slope, float* mean, float* rstd, float* running_mean, float* running_var,
                            float momentum, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_geo(N, HW, C, "norm_fwd");
  if (rc) return rc;
  HWG_REQUIRE(x && y && mean && rstd, "norm_fwd: null pointer");
  HWG_REQUIRE(mode >= 0 && mode <= 2, "norm_fwd: bad mode %d", mode);
  if (mode == MODE_GN) HWG_REQUIRE(groups > 0 && C % groups == 0, "norm_fwd: C=%d not divisible by groups=%d", C, groups);
  Geo g = make_geo(N, HW, C);
  if (!ws || ws_bytes < hwg_norm_workspace(N, HW, C)) { hwg_set_error("norm_fwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  dim3 grid(g.chunks, N);
  if (int* sync = mode != MODE_BN ? fused_sync(N, st) : nullptr) {
    InlineStats is = {};
    is.part = part; is.cpg = (mode == MODE_GN) ? C / groups : 1; is.eps = eps; is.out_a = mean; is.out_b = rstd;
    hipLaunchKernelGGL(norm_fwd_fused_kernel<0>, grid, dim3(256), red_smem(g), st, x, y, g, part, (const float*)nullptr, (const float*)nullptr, 0.f,
                       0.f, (float*)nullptr, 0ull, 0ull, gamma, beta, affine_per_sample, chan_mask, act, slope, is, sync);
    HWG_LAUNCH_CHECK("norm_fwd.fused");
    return HWG_OK;
  }
  hipLaunchKernelGGL(moments_fwd_kernel<0>, grid, dim3(256), red_smem(g), st, x, g, part, (const float*)nullptr, (const float*)nullptr, 0.f, 0.f,
                     (float*)nullptr, 0ull, 0ull);
  HWG_LAUNCH_CHECK("norm_fwd.moments");
  InlineStats is = {};
  if (mode == MODE_BN) {   // batch statistics span the samples: separate finalize pass (also updates the running statistics)
    hipLaunchKernelGGL(finalize_fwd_kernel, dim3(hwg_cdiv(C, 4)), dim3(256), 0, st, (const double*)part, g, eps, mean, rstd, running_mean,
                       running_var, momentum);
    HWG_LAUNCH_CHECK("norm_fwd.finalize");
  } else {                 // per-sample statistics are formed inside the apply pass
    is.part = part; is.cpg = (mode == MODE_GN) ? C / groups : 1; is.eps = eps; is.out_a = mean; is.out_b = rstd;
  }
  hipLaunchKernelGGL(apply_fwd_kernel, grid, dim3(256), 0, st, x, y, g, (const float*)mean, (const float*)rstd, gamma, beta,
                     affine_per_sample, chan_mask, act, slope, is);
  HWG_LAUNCH_CHECK("norm_fwd.apply");
  return HWG_OK;
}

extern "C" int hwg_norm_bwd(const float* dy, const float* x, const float* y, float* dx, int N, int HW, int C, int mode, int groups,
                            const float* gamma, const float* beta, int affine_per_sample, const float* chan_mask, int act, float slope,
                            const float* mean, const float* rstd, float* dgamma, float* dbeta, int accumulate,
                            void* ws, size_t ws_bytes, void* stream) {
  int rc = check_geo(N, HW, C, "norm_bwd");
  if (rc) return rc;
  HWG_REQUIRE(dy && x && dx && mean && rstd, "norm_bwd: null pointer");
  HWG_REQUIRE(act == 0 || act == HWG_ACT_RELU || act == HWG_ACT_LRELU || y, "norm_bwd: y required when tanh is fused");
  // relu / leaky relu: the gate is the sign of the forward pre-activation, recomputed from x, the statistics and the affine of the forward
  // call - two of the seven passes over the tensor (y in both sweeps) are not made
  if (act == HWG_ACT_RELU || act == HWG_ACT_LRELU) y = nullptr;
  Geo g = make_geo(N, HW, C);
  if (!ws || ws_bytes < hwg_norm_workspace(N, HW, C)) { hwg_set_error("norm_bwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  float* c1 = (float*)((char*)ws + 2 * part_bytes(g));
  float* c2 = c1 + (size_t)N * C;
  dim3 grid(g.chunks, N);
  if (int* sync = mode != MODE_BN ? fused_sync(N, st) : nullptr) {
    InlineStats is = {};
    is.part = part; is.cpg = (mode == MODE_GN) ? C / groups : 1; is.gamma = gamma; is.per_sample = affine_per_sample;
    is.out_a = affine_per_sample ? dgamma : nullptr; is.out_b = affine_per_sample ? dbeta : nullptr; is.accumulate = accumulate;
    const bool fold = !affine_per_sample && (dgamma || dbeta);
    hipLaunchKernelGGL(norm_bwd_fused_kernel<false>, grid, dim3(256), red_smem(g), st, dy, x, y, dx, g, part, mean, rstd, gamma, beta,
                       affine_per_sample, (const float*)c1, (const float*)c2, chan_mask, act, slope, (const float*)nullptr, 0.f, (double*)nullptr, is,
                       fold ? dgamma : (float*)nullptr, fold ? dbeta : (float*)nullptr, accumulate, sync);
    HWG_LAUNCH_CHECK("norm_bwd.fused");
    return HWG_OK;
  }
  hipLaunchKernelGGL(moments_bwd_kernel, grid, dim3(256), red_smem(g), st, dy, x, y, g, part, mean, rstd, chan_mask, act, slope, gamma, beta,
                     affine_per_sample);
  HWG_LAUNCH_CHECK("norm_bwd.moments");
  InlineStats is = {};
  if (mode == MODE_BN) {
    hipLaunchKernelGGL(finalize_bwd_kernel, dim3(hwg_cdiv(C, 4)), dim3(256), 0, st, (const double*)part, g, gamma, c1, c2, dgamma, dbeta,
                       accumulate);
    HWG_LAUNCH_CHECK("norm_bwd.finalize");
  } else {
    is.part = part; is.cpg = (mode == MODE_GN) ? C / groups : 1; is.gamma = gamma; is.per_sample = affine_per_sample;
    is.out_a = affine_per_sample ? dgamma : nullptr; is.out_b = affine_per_sample ? dbeta : nullptr; is.accumulate = accumulate;
  }
  const bool fold_pg = mode != MODE_BN && !affine_per_sample && (dgamma || dbeta);    // summed by the apply pass's first-sample workgroups
  hipLaunchKernelGGL(apply_bwd_kernel<false>, grid, dim3(256), 0, st, dy, x, y, dx, g, mean, rstd, gamma, affine_per_sample,
                     (const float*)c1, (const float*)c2, chan_mask, act, slope, nullptr, 0.f, nullptr, is, beta,
                     fold_pg ? dgamma : (float*)nullptr, fold_pg ? dbeta : (float*)nullptr, accumulate);
  HWG_LAUNCH_CHECK("norm_bwd.apply");
  return HWG_OK;
}

extern "C" int hwg_adain_fwd(const float* x, const float* noise, const float* noise_w, float noise_scale, float slope,
                             const float* gamma, const float* beta, float eps, float* u, float* y, float* mean, float* rstd,
                             int N, int HW, int C, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_geo(N, HW, C, "adain_fwd");
  if (rc) return rc;
  HWG_REQUIRE(x && noise && noise_w && gamma && beta && u && y && mean && rstd, "adain_fwd: null pointer");
  Geo g = make_geo(N, HW, C);
  if (!ws || ws_bytes < hwg_norm_workspace(N, HW, C)) { hwg_set_error("adain_fwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  dim3 grid(g.chunks, N);
  InlineStats is = {};
  is.part = part; is.cpg = 1; is.eps = eps; is.out_a = mean; is.out_b = rstd;
  if (int* sync = fused_sync(N, st)) {
    hipLaunchKernelGGL(norm_fwd_fused_kernel<1>, grid, dim3(256), red_smem(g), st, x, y, g, part, noise, noise_w, noise_scale, slope, u, 0ull, 0ull,
                       gamma, beta, 1, (const float*)nullptr, 0, 0.f, is, sync);
    HWG_LAUNCH_CHECK("adain_fwd.fused");
    return HWG_OK;
  }
  hipLaunchKernelGGL(moments_fwd_kernel<1>, grid, dim3(256), red_smem(g), st, x, g, part, noise, noise_w, noise_scale, slope, u, 0ull, 0ull);
  HWG_LAUNCH_CHECK("adain_fwd.moments");
  hipLaunchKernelGGL(apply_fwd_kernel, grid, dim3(256), 0, st, (const float*)u, y, g, (const float*)mean, (const float*)rstd, gamma, beta, 1,
                     (const float*)nullptr, 0, 0.f, is);
  HWG_LAUNCH_CHECK("adain_fwd.apply");
  return HWG_OK;
}

// hwg_adain_fwd with the noise drawn inside the kernel (forward-only calls): the values are those hwg_randn(seed, offset) writes for a
// tensor of N*HW*C elements, which is never materialised - one write and one read pass of the largest tensors of a generation call less.
// `u` may alias `x` (every element is read once, by the thread that overwrites it).
extern "C" int hwg_adain_fwd_rng(const float* x, unsigned long long seed, unsigned long long offset, const float* noise_w, float noise_scale,
                                 float slope, const float* gamma, const float* beta, float eps, float* u, float* y, float* mean, float* rstd,
                                 int N, int HW, int C, void* ws, size_t ws_bytes, void* stream) {
  int rc = check_geo(N, HW, C, "adain_fwd_rng");
  if (rc) return rc;
  HWG_REQUIRE(x && noise_w && gamma && beta && u && y && mean && rstd, "adain_fwd_rng: null pointer");
  Geo g = make_geo(N, HW, C);
  if (!ws || ws_bytes < hwg_norm_workspace(N, HW, C)) { hwg_set_error("adain_fwd_rng: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  dim3 grid(g.chunks, N);
  InlineStats is = {};
  is.part = part; is.cpg = 1; is.eps = eps; is.out_a = mean; is.out_b = rstd;
  if (int* sync = fused_sync(N, st)) {
    hipLaunchKernelGGL(norm_fwd_fused_kernel<2>, grid, dim3(256), red_smem(g), st, x, y, g, part, (const float*)nullptr, noise_w, noise_scale, slope, u,
                       (unsigned long long)seed, (unsigned long long)offset, gamma, beta, 1, (const float*)nullptr, 0, 0.f, is, sync);
    HWG_LAUNCH_CHECK("adain_fwd_rng.fused");
    return HWG_OK;
  }
  hipLaunchKernelGGL(moments_fwd_kernel<2>, grid, dim3(256), red_smem(g), st, x, g, part, (const float*)nullptr, noise_w, noise_scale, slope, u,
                     (unsigned long long)seed, (unsigned long long)offset);
  HWG_LAUNCH_CHECK("adain_fwd_rng.moments");
  hipLaunchKernelGGL(apply_fwd_kernel, grid, dim3(256), 0, st, (const float*)u, y, g, (const float*)mean, (const float*)rstd, gamma, beta, 1,
                     (const float*)nullptr, 0, 0.f, is);
  HWG_LAUNCH_CHECK("adain_fwd_rng.apply");
  return HWG_OK;
}

// ---- deferred parameter-gradient sums of the generator epilogue -------------------------------------------------------------------
// adain_param_grad_kernel sums the [n][chunk][c][2] partials the apply pass just wrote: a 5 us launch behind each of the generator's ten
// epilogues, per backward pass. With the caller's deferral mark set (hwg_wgrad_defer_next: the workspace then lives in the caller's arena
// until the flush) the sum is queued and hwg_wgrad_defer_flush makes all of them with one table-driven launch - same arithmetic per channel.
#include <mutex>
#include <vector>
bool hwg_wgrad_defer_take();
namespace {
struct PgEntry { const double* part2; float* dbias; float* dnw; float nscale; int items, C, accumulate, first_block; };
constexpr int PG_MAX = 64;
struct PgTable { int n, pad; PgEntry e[PG_MAX]; };
std::mutex g_pg_mu;
std::vector<PgEntry> g_pg_queue;

__global__ __launch_bounds__(256) void adain_param_grad_multi_kernel(const PgTable t) {
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.e[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PgEntry& e = t.e[lo];
  const int lane = threadIdx.x & 63;
  const int c = (((int)blockIdx.x - e.first_block) * 256 + threadIdx.x) >> 6;
  if (c >= e.C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int it = lane; it < e.items; it += 64) {
    const double* p = e.part2 + ((size_t)it * e.C + c) * 2;
    s1 += p[0]; s2 += p[1];
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (lane == 0) {
    if (e.dbias) e.dbias[c] = (e.accumulate ? e.dbias[c] : 0.f) + (float)s1;
    if (e.dnw) e.dnw[c] = (e.accumulate ? e.dnw[c] : 0.f) + (float)(s2 * (double)e.nscale);
  }
}
}  // namespace

// called by hwg_wgrad_defer_flush (conv_mfma.hip): the queued epilogue parameter-gradient sums, several sums into one buffer in queue order
int hwg_pg_defer_flush(hipStream_t st, int* launches) {
  std::vector<PgEntry> q;
  {
    std::lock_guard<std::mutex> lock(g_pg_mu);
    q.swap(g_pg_queue);
  }
  std::vector<int> pass(q.size(), 0);
  int npass = 0;
  for (size_t i = 0; i < q.size(); ++i) {
    for (size_t j = 0; j < i; ++j)
      if ((q[i].dnw && q[j].dnw == q[i].dnw) || (q[i].dbias && q[j].dbias == q[i].dbias)) pass[i] = pass[j] + 1 > pass[i] ? pass[j] + 1 : pass[i];
    if (pass[i] + 1 > npass) npass = pass[i] + 1;
  }
  for (int ps = 0; ps < npass; ++ps) {
    size_t i = 0;
    while (i < q.size()) {
      PgTable t; t.n = 0; t.pad = 0;
      int blocks = 0;
      for (; i < q.size() && t.n < PG_MAX; ++i) {
        if (pass[i] != ps) continue;
        PgEntry e = q[i];
        e.first_block = blocks;
        blocks += hwg_cdiv(e.C, 4);
        t.e[t.n++] = e;
      }
      if (t.n == 0) break;
      hipLaunchKernelGGL(adain_param_grad_multi_kernel, dim3(blocks), dim3(256), 0, st, t);
      HWG_LAUNCH_CHECK("adain_param_grad_multi");
      if (launches) ++*launches;
    }
  }
  return HWG_OK;
}

extern "C" int hwg_adain_bwd(const float* dy, const float* u, const float* noise, float noise_scale, float slope,
                             const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                             float* dnoise_w, float* dbias, int accumulate_params, int N, int HW, int C,
                             void* ws, size_t ws_bytes, void* stream) {
  const bool defer = hwg_wgrad_defer_take();
  int rc = check_geo(N, HW, C, "adain_bwd");
  if (rc) return rc;
  HWG_REQUIRE(dy && u && noise && gamma && mean && rstd && dx && dgamma && dbeta, "adain_bwd: null pointer");
  Geo g = make_geo(N, HW, C);
  if (!ws || ws_bytes < hwg_norm_workspace(N, HW, C)) { hwg_set_error("adain_bwd: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  double* part = (double*)ws;
  double* part2 = (double*)((char*)ws + part_bytes(g));
  float* c1 = (float*)((char*)ws + 2 * part_bytes(g));
  float* c2 = c1 + (size_t)N * C;
  dim3 grid(g.chunks, N);
  // dgamma/dbeta are per (n,c) and are NOT accumulated (they feed the style Linear's backward); c1/c2 are formed inside the apply pass
  InlineStats is = {};
  is.part = part; is.cpg = 1; is.gamma = gamma; is.per_sample = 1; is.out_a = dgamma; is.out_b = dbeta; is.accumulate = 0;
  if (int* sync = fused_sync(N, st)) {
    hipLaunchKernelGGL(norm_bwd_fused_kernel<true>, grid, dim3(256), red_smem(g), st, dy, u, (const float*)nullptr, dx, g, part, mean, rstd, gamma,
                       (const float*)nullptr, 1, (const float*)c1, (const float*)c2, (const float*)nullptr, 0, 0.f, noise, slope, part2, is,
                       (float*)nullptr, (float*)nullptr, 0, sync);
    HWG_LAUNCH_CHECK("adain_bwd.fused");
  } else {
    hipLaunchKernelGGL(moments_bwd_kernel, grid, dim3(256), red_smem(g), st, dy, u, (const float*)nullptr, g, part, mean, rstd,
                       (const float*)nullptr, 0, 0.f, (const float*)nullptr, (const float*)nullptr, 0);
    HWG_LAUNCH_CHECK("adain_bwd.moments");
    hipLaunchKernelGGL(apply_bwd_kernel<true>, grid, dim3(256), red_smem(g), st, dy, u, (const float*)nullptr, dx, g, mean, rstd, gamma, 1,
                       (const float*)c1, (const float*)c2, (const float*)nullptr, 0, 0.f, noise, slope, part2, is, (const float*)nullptr,
                       (float*)nullptr, (float*)nullptr, 0);
    HWG_LAUNCH_CHECK("adain_bwd.apply");
  }
  if ((dnoise_w || dbias) && defer) {
    PgEntry e;
    e.part2 = part2; e.dbias = dbias; e.dnw = dnoise_w; e.nscale = noise_scale; e.items = g.N * g.chunks; e.C = g.C; e.accumulate = accumulate_params;
    e.first_block = 0;
    std::lock_guard<std::mutex> lock(g_pg_mu);
    g_pg_queue.push_back(e);
  } else if (dnoise_w || dbias) {
    hipLaunchKernelGGL(adain_param_grad_kernel, dim3(hwg_cdiv(C, 4)), dim3(256), 0, st, (const double*)part2, g, noise_scale, dbias, dnoise_w,
                       accumulate_params);
    HWG_LAUNCH_CHECK("adain_bwd.param_grad");
  }
  return HWG_OK;
}

extern "C" int hwg_bias_act_fwd(const float* x, const float* bias, const float* chan_mask, float* y, long long rows, int HW, int C,
                                int act, float slope, void* stream) {
  HWG_REQUIRE(x && y && rows > 0 && C > 0 && HW > 0, "bias_act_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0) {
    HWG_REQUIRE(rows * C / 4 < (1ll << 31), "bias_act_fwd: tensor too large for 32-bit indexing");
    const unsigned total4 = (unsigned)(rows * C / 4);
    const dim3 grid(hwg_stream_grid((total4 + 1) / 2, 256));
    if (bias || chan_mask) hipLaunchKernelGGL(bias_act_fwd_kernel<true>, grid, dim3(256), 0, st, x, bias, chan_mask, y, total4, HW, C, act, slope);
    else hipLaunchKernelGGL(bias_act_fwd_kernel<false>, grid, dim3(256), 0, st, x, bias, chan_mask, y, total4, HW, C, act, slope);
  } else {
    hipLaunchKernelGGL(bias_act_fwd_scalar_kernel, dim3(hwg_stream_grid(rows * C, 256)), dim3(256), 0, st, x, bias, chan_mask, y, rows, HW, C, act, slope);
  }
  HWG_LAUNCH_CHECK("bias_act_fwd");
  return HWG_OK;
}

extern "C" int hwg_bias_act_bwd(const float* dy, const float* y, const float* chan_mask, float* dx, long long rows, int HW, int C,
                                int act, float slope, void* stream) {
  HWG_REQUIRE(dy && dx && rows > 0 && C > 0 && HW > 0, "bias_act_bwd: bad arguments");
  HWG_REQUIRE(act == 0 || y, "bias_act_bwd: y required when an activation is fused");
  hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0) {
    HWG_REQUIRE(rows * C / 4 < (1ll << 31), "bias_act_bwd: tensor too large for 32-bit indexing");
    const unsigned total4 = (unsigned)(rows * C / 4);
    const dim3 grid(hwg_stream_grid((total4 + 1) / 2, 256));
    if (chan_mask) hipLaunchKernelGGL(bias_act_bwd_kernel<true>, grid, dim3(256), 0, st, dy, y, chan_mask, dx, total4, HW, C, act, slope);
    else hipLaunchKernelGGL(bias_act_bwd_kernel<false>, grid, dim3(256), 0, st, dy, y, chan_mask, dx, total4, HW, C, act, slope);
  } else {
    hipLaunchKernelGGL(bias_act_bwd_scalar_kernel, dim3(hwg_stream_grid(rows * C, 256)), dim3(256), 0, st, dy, y, chan_mask, dx, rows, HW, C, act, slope);
  }
  HWG_LAUNCH_CHECK("bias_act_bwd");
  return HWG_OK;
}

// y = act(gamma * (x - running_mean) / sqrt(running_var + eps) + beta)   (BatchNorm in eval mode; no statistics pass)
extern "C" int hwg_norm_frozen_fwd(const float* x, float* y, int N, int HW, int C, const float* running_mean, const float* running_var, float eps,
                                   const float* gamma, const float* beta, int act, float slope, float* mean, float* rstd, void* stream) {
  int rc = check_geo(N, HW, C, "norm_frozen_fwd");
  if (rc) return rc;
  HWG_REQUIRE(x && y && running_mean && running_var && mean && rstd, "norm_frozen_fwd: null pointer");
  Geo g = make_geo(N, HW, C);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(frozen_stats_kernel, dim3(hwg_cdiv(N * C, 128)), dim3(128), 0, st, running_mean, running_var, eps, N, C, mean, rstd);
  HWG_LAUNCH_CHECK("norm_frozen.stats");
  hipLaunchKernelGGL(apply_fwd_kernel, dim3(g.chunks, N), dim3(256), 0, st, x, y, g, (const float*)mean, (const float*)rstd, gamma, beta, 0,
                     (const float*)nullptr, act, slope, InlineStats{});
  HWG_LAUNCH_CHECK("norm_frozen.apply");
  return HWG_OK;
}
