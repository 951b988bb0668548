// Pooling / resampling / padding / concatenation kernels on NHWC tensors. All HBM-bound; every kernel is
// a grid-stride loop over output vectors of V channels (V = 4 when C % 4 == 0, else 1).
#include "hwg_common.h"

namespace {

template <int V> struct VecT { float v[V]; };
template <int V> __device__ __forceinline__ VecT<V> vload(const float* p) {
  VecT<V> r;
  if (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); r.v[0] = t.x; r.v[1 % V] = t.y; r.v[2 % V] = t.z; r.v[3 % V] = t.w; }
  else r.v[0] = p[0];
  return r;
}
template <int V> __device__ __forceinline__ void vstore(float* p, const VecT<V>& r) {
  if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(r.v[0], r.v[1 % V], r.v[2 % V], r.v[3 % V]);
  else p[0] = r.v[0];
}
template <int V> __device__ __forceinline__ VecT<V> vzero() { VecT<V> r; for (int i = 0; i < V; ++i) r.v[i] = 0.f; return r; }

// 32-bit element indices: a 64-bit division costs ~40 VALU instructions on this ISA and every kernel here splits its index three or four
// times per 16 bytes moved - with 64-bit indices these "HBM-bound" kernels were VALU-bound at ~3 TB/s. LAUNCH_V refuses tensors of 2^31
// vectors or more.
#define GRID_STRIDE(i, total) \
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)(total); i += gridDim.x * blockDim.x)

// ---------------- average pooling (kernel == stride, no padding, floor) ----------------
template <int V>
__global__ void avgpool_fwd_kernel(const float* x, float* y, int N, int H, int W, int C, int kh, int kw, int P, int Q) {
  const int CV = C / V;
  const long long total = (long long)N * P * Q * CV;
  const float inv = 1.f / (kh * kw);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P); const int n = (int)(t / P);
    VecT<V> acc = vzero<V>();
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) {
        const VecT<V> v = vload<V>(x + (((long long)n * H + p * kh + a) * W + q * kw + b) * C + c);
        for (int e = 0; e < V; ++e) acc.v[e] += v.v[e];
      }
    for (int e = 0; e < V; ++e) acc.v[e] *= inv;
    vstore<V>(y + (size_t)i * V, acc);
  }
}
template <int V>
__global__ void avgpool_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C, int kh, int kw, int P, int Q) {
  const int CV = C / V;
  const long long total = (long long)N * H * W * CV;
  const float inv = 1.f / (kh * kw);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int n = (int)(t / H);
    const int p = h / kh, q = w / kw;
    VecT<V> r = vzero<V>();
    if (p < P && q < Q) {
      r = vload<V>(dy + (((long long)n * P + p) * Q + q) * C + c);
      for (int e = 0; e < V; ++e) r.v[e] *= inv;
    }
    vstore<V>(dx + (size_t)i * V, r);
  }
}

// ---------------- activation + average pooling in one pass (model/discriminator_ap.py:84-131: SN conv -> Dropout2d -> LeakyReLU -> AvgPool2d) ------------
// y = avgpool(act(mask[n][c] * x)): the full-resolution activation is never written (one read of x and a quarter-size write instead of
// read + write + read + quarter write); the backward pass recomputes the gate from x: dx = mask * act'(mask * x) * dy[pooled] / (kh kw).
// Every element goes through the same rounded operations, in the same order, as hwg_bias_act_fwd followed by hwg_avgpool_fwd (and their
// backward kernels): bit-identical (explicitly rounded multiplies, so that no product is contracted into the window sum).
__device__ __forceinline__ float act_rn(float v, int act, float slope) {
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v > 0.f ? v : __fmul_rn(v, slope);
  return v;
}
template <int V>
__global__ void act_avgpool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ y, int N, int H, int W, int C,
                                       int kh, int kw, int P, int Q, int act, float slope) {
  const int CV = C / V;
  const long long total = (long long)N * P * Q * CV;
  const float inv = 1.f / (kh * kw);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P); const int n = (int)(t / P);
    VecT<V> k;
    for (int e = 0; e < V; ++e) k.v[e] = 1.f;
    if (mask) k = vload<V>(mask + (size_t)n * C + c);
    VecT<V> acc = vzero<V>();
    for (int a = 0; a < kh; ++a)
      for (int b = 0; b < kw; ++b) {
        const VecT<V> v = vload<V>(x + (((long long)n * H + p * kh + a) * W + q * kw + b) * C + c);
        for (int e = 0; e < V; ++e) acc.v[e] = __fadd_rn(acc.v[e], act_rn(mask ? __fmul_rn(v.v[e], k.v[e]) : v.v[e], act, slope));
      }
    for (int e = 0; e < V; ++e) acc.v[e] = __fmul_rn(acc.v[e], inv);
    vstore<V>(y + (size_t)i * V, acc);
  }
}
template <int V>
__global__ void act_avgpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ dx,
                                       int N, int H, int W, int C, int kh, int kw, int P, int Q, int act, float slope) {
  const int CV = C / V;
  const long long total = (long long)N * H * W * CV;
  const float inv = 1.f / (kh * kw);
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int n = (int)(t / H);
    const int p = h / kh, q = w / kw;
    VecT<V> r = vzero<V>();
    if (p < P && q < Q) {
      r = vload<V>(dy + (((long long)n * P + p) * Q + q) * C + c);
      const VecT<V> v = vload<V>(x + (size_t)i * V);
      VecT<V> k;
      for (int e = 0; e < V; ++e) k.v[e] = 1.f;
      if (mask) k = vload<V>(mask + (size_t)n * C + c);
      for (int e = 0; e < V; ++e) {
        const float z = mask ? __fmul_rn(v.v[e], k.v[e]) : v.v[e];       // the forward pass's pre-activation: same sign as its output
        float g = __fmul_rn(r.v[e], inv);
        if (act == 1) g = z > 0.f ? g : __fmul_rn(g, 0.f);
        if (act == 2) g = z > 0.f ? g : __fmul_rn(g, slope);
        r.v[e] = mask ? __fmul_rn(g, k.v[e]) : g;
      }
    }
    vstore<V>(dx + (size_t)i * V, r);
  }
}

// ---------------- max pooling (general kernel/stride/padding), argmax saved as h*W+w ----------------
// RELU: y = relu(max) and, backward, dy gated by y > 0 (model/cnn_only_hwr.py:31-43: conv -> ReLU -> MaxPool2d; relu(max(w)) == max(relu(w))
// exactly, and a window whose maximum is <= 0 passes no gradient in either order) - the ReLU passes over the pooled tensor ride along
template <int V, bool RELU>
__global__ void maxpool_fwd_kernel(const float* x, float* y, int* idx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                                   int ph, int pw, int P, int Q) {
  const int CV = C / V;
  const long long total = (long long)N * P * Q * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P); const int n = (int)(t / P);
    VecT<V> best; int bi[V];
    for (int e = 0; e < V; ++e) { best.v[e] = -INFINITY; bi[e] = -1; }
    for (int a = 0; a < kh; ++a) {
      const int h = p * sh - ph + a;
      if (h < 0 || h >= H) continue;
      for (int b = 0; b < kw; ++b) {
        const int w = q * sw - pw + b;
        if (w < 0 || w >= W) continue;
        const VecT<V> v = vload<V>(x + (((long long)n * H + h) * W + w) * C + c);
        // same update rule as ATen's max_pool2d: first maximum wins, NaN propagates
        for (int e = 0; e < V; ++e)
          if (v.v[e] > best.v[e] || v.v[e] != v.v[e] || bi[e] < 0) { best.v[e] = v.v[e]; bi[e] = h * W + w; }
      }
    }
    if (RELU)
      for (int e = 0; e < V; ++e) best.v[e] = best.v[e] > 0.f ? best.v[e] : 0.f;     // (act_apply's ReLU, as hwg_bias_act_fwd applies it)
    vstore<V>(y + (size_t)i * V, best);
    for (int e = 0; e < V; ++e) idx[(size_t)i * V + e] = bi[e];
  }
}
template <int V, bool RELU>
__global__ void maxpool_bwd_kernel(const float* dy, const float* y, const int* idx, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                                   int ph, int pw, int P, int Q) {
  const int CV = C / V;
  const long long total = (long long)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int n = (int)(t / H);
    const int me = h * W + w;
    VecT<V> acc = vzero<V>();
    // output rows p with p*sh-ph <= h <= p*sh-ph+kh-1
    int pmin = (h + ph - kh + 1 + sh - 1); pmin = pmin > 0 ? pmin / sh : 0;
    int pmax = (h + ph) / sh; if (pmax > P - 1) pmax = P - 1;
    int qmin = (w + pw - kw + 1 + sw - 1); qmin = qmin > 0 ? qmin / sw : 0;
    int qmax = (w + pw) / sw; if (qmax > Q - 1) qmax = Q - 1;
    for (int p = pmin; p <= pmax; ++p)
      for (int q = qmin; q <= qmax; ++q) {
        const long long o = (((long long)n * P + p) * Q + q) * C + c;
        for (int e = 0; e < V; ++e)
          if (idx[o + e] == me) acc.v[e] += RELU ? __fmul_rn(dy[o + e], y[o + e] > 0.f ? 1.f : 0.f) : dy[o + e];
      }
    vstore<V>(dx + (size_t)i * V, acc);
  }
}

// ---------------- nearest upsample by integer factors ----------------
template <int V>
__global__ void upsample_fwd_kernel(const float* x, float* y, int N, int H, int W, int C, int fh, int fw) {
  const int CV = C / V;
  const int P = H * fh, Q = W * fw;
  const long long total = (long long)N * P * Q * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P); const int n = (int)(t / P);
    vstore<V>(y + (size_t)i * V, vload<V>(x + (((long long)n * H + p / fh) * W + q / fw) * C + c));
  }
}
template <int V>
__global__ void upsample_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C, int fh, int fw) {
  const int CV = C / V;
  const int P = H * fh, Q = W * fw;
  const long long total = (long long)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int n = (int)(t / H);
    VecT<V> acc = vzero<V>();
    for (int a = 0; a < fh; ++a)
      for (int b = 0; b < fw; ++b) {
        const VecT<V> v = vload<V>(dy + (((long long)n * P + h * fh + a) * Q + w * fw + b) * C + c);
        for (int e = 0; e < V; ++e) acc.v[e] += v.v[e];
      }
    vstore<V>(dx + (size_t)i * V, acc);
  }
}

// ---------------- depthwise 3x3 binomial blur, zero padding 1 (symmetric: forward == backward) ----------------
template <int V>
__global__ void blur3_kernel(const float* x, float* y, int N, int H, int W, int C) {
  const int CV = C / V;
  const long long total = (long long)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int n = (int)(t / H);
    VecT<V> acc = vzero<V>();
    // accumulate in the same tap order as the 3x3 correlation (row major)
    for (int a = -1; a <= 1; ++a) {
      const int hh = h + a;
      if (hh < 0 || hh >= H) continue;
      for (int b = -1; b <= 1; ++b) {
        const int ww = w + b;
        if (ww < 0 || ww >= W) continue;
        const float k = ((a == 0 ? 2.f : 1.f) * (b == 0 ? 2.f : 1.f)) * (1.f / 16.f);
        const VecT<V> v = vload<V>(x + (((long long)n * H + hh) * W + ww) * C + c);
        for (int e = 0; e < V; ++e) acc.v[e] += k * v.v[e];
      }
    }
    vstore<V>(y + (size_t)i * V, acc);
  }
}

// ---------------- 2-D padding: mode 0 constant(value), mode 1 replicate ----------------
template <int V>
__global__ void pad2d_fwd_kernel(const float* x, float* y, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, float value) {
  const int CV = C / V;
  const int P = H + pt + pb, Q = W + pl + pr;
  const long long total = (long long)N * P * Q * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int q = (int)(t % Q); t /= Q;
    const int p = (int)(t % P); const int n = (int)(t / P);
    int h = p - pt, w = q - pl;
    VecT<V> r;
    if (mode == 1) {
      h = h < 0 ? 0 : (h >= H ? H - 1 : h);
      w = w < 0 ? 0 : (w >= W ? W - 1 : w);
      r = vload<V>(x + (((long long)n * H + h) * W + w) * C + c);
    } else if (h >= 0 && h < H && w >= 0 && w < W) {
      r = vload<V>(x + (((long long)n * H + h) * W + w) * C + c);
    } else {
      for (int e = 0; e < V; ++e) r.v[e] = value;
    }
    vstore<V>(y + (size_t)i * V, r);
  }
}
template <int V>
__global__ void pad2d_bwd_kernel(const float* dy, float* dx, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode) {
  const int CV = C / V;
  const int P = H + pt + pb, Q = W + pl + pr;
  const long long total = (long long)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * V; unsigned t = i / CV;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); const int n = (int)(t / H);
    int p0 = h + pt, p1 = h + pt, q0 = w + pl, q1 = w + pl;
    if (mode == 1) {
      if (h == 0) p0 = 0;
      if (h == H - 1) p1 = P - 1;
      if (w == 0) q0 = 0;
      if (w == W - 1) q1 = Q - 1;
    }
    VecT<V> acc = vzero<V>();
    for (int p = p0; p <= p1; ++p)
      for (int q = q0; q <= q1; ++q) {
        if (p < 0 || p >= P || q < 0 || q >= Q) continue;  // negative pads (crop)
        const VecT<V> v = vload<V>(dy + (((long long)n * P + p) * Q + q) * C + c);
        for (int e = 0; e < V; ++e) acc.v[e] += v.v[e];
      }
    vstore<V>(dx + (size_t)i * V, acc);
  }
}

// ---------------- channel slice copies (concat / split), with optional per-sample broadcast over HW ----------------
// dst[row][doff + c] = src[(bcast ? row / HW : row)][soff + c],  c < Cn
__global__ void copy_channels_kernel(const float* src, int Cs, int soff, float* dst, int Cd, int doff, int Cn, long long rows, int HW, int bcast,
                                     int accumulate) {
  const long long total = rows * Cn;
  GRID_STRIDE(i, total) {
    const unsigned row = i / Cn;
    const int c = (int)(i - row * Cn);
    const unsigned srow = bcast ? row / HW : row;
    const float v = src[(size_t)srow * Cs + soff + c];
    float* d = dst + (size_t)row * Cd + doff + c;
    *d = accumulate ? *d + v : v;
  }
}
// dst[row][0..Cpad) = src[row][0..C) followed by zeros; one 16-byte store per thread
__global__ void pad_channels_kernel(const float* __restrict__ src, int C, float* __restrict__ dst, int Cpad, unsigned total4) {
  const unsigned q = Cpad / 4;
  GRID_STRIDE(i, total4) {
    const unsigned row = (unsigned)i / q;
    const int c = (int)((unsigned)i - row * q) * 4;
    const float* s = src + (size_t)row * C + c;
    float4 v;
    v.x = c + 0 < C ? s[0] : 0.f; v.y = c + 1 < C ? s[1] : 0.f;
    v.z = c + 2 < C ? s[2] : 0.f; v.w = c + 3 < C ? s[3] : 0.f;
    reinterpret_cast<float4*>(dst)[i] = v;
  }
}
// out[n][c] (+)= sum_hw src[n*HW + hw][soff + c]   (backward of the broadcast)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* src, int Cs, int soff, float* out, int Cn, int HW, int accumulate) {
  __shared__ float red[256];
  const int n = blockIdx.x;
  const int c = blockIdx.y;
  float s = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) s += src[((long long)n * HW + p) * Cs + soff + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n * Cn + c] = accumulate ? out[n * Cn + c] + red[0] : red[0];
}

// one-hot rows: label is [L][B] (time major, as the reference passes it); out is [B][L][ncls]
__global__ void onehot_kernel(const int* label, float* out, int L, int B, int ncls, int Cd, int doff) {
  const long long total = (long long)B * L * ncls;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % ncls);
    const unsigned t = i / ncls;
    const int l = (int)(t % L), b = (int)(t / L);
    out[((long long)b * L + l) * Cd + doff + c] = (label[l * B + b] == c) ? 1.f : 0.f;
  }
}

// both layouts of the one-hot in one launch: blc[b][l][c] (what the networks read, NHWC [B,1,L,C]) and lbc[l][b][c] (the reference's
// time-major [L,B,C], what HWWithStyle.onehot returns) - the model API hands the time-major tensor around and the generator / spacer
// permuted it straight back: three launches per one-hot
__global__ void onehot_both_kernel(const int* label, float* blc, float* lbc, int L, int B, int ncls) {
  const long long total = (long long)B * L * ncls;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % ncls);
    const unsigned t = i / ncls;
    const int l = (int)(t % L), b = (int)(t / L);
    const float v = (label[l * B + b] == c) ? 1.f : 0.f;
    blc[i] = v;
    lbc[((long long)l * B + b) * ncls + c] = v;
  }
}

// generic strided 4-d permute copy: out[i0][i1][i2][i3] (contiguous) = in[i0*s0 + i1*s1 + i2*s2 + i3*s3]
__global__ void permute4_kernel(const float* in, float* out, int d0, int d1, int d2, int d3, long long s0, long long s1, long long s2, long long s3) {
  const long long total = (long long)d0 * d1 * d2 * d3;
  GRID_STRIDE(i, total) {
    const int i3 = (int)(i % d3); unsigned t = i / d3;
    const int i2 = (int)(t % d2); t /= d2;
    const int i1 = (int)(t % d1); const int i0 = (int)(t / d1);
    out[i] = in[i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3];
  }
}

// FusedUpsample weight (model/pure_gen.py:268-276): w4 = avg of the four 1-shifted copies of pad(w3*mult, 1); [A][B][3][3] -> [A][B][4][4]
__global__ void fused_up_weight_fwd_kernel(const float* w3, float* w4, long long AB, float mult) {
  const long long total = AB * 16;
  GRID_STRIDE(i, total) {
    const int s = (int)(i % 4), r = (int)((i / 4) % 4);
    const long long ab = i / 16;
    float acc = 0.f;
    // padded index (r+dr, s+ds) for dr,ds in {0,1}; padded[p][q] = w3[p-1][q-1]
    for (int dr = 0; dr < 2; ++dr)
      for (int ds = 0; ds < 2; ++ds) {
        const int p = r + dr - 1, q = s + ds - 1;
        if (p >= 0 && p < 3 && q >= 0 && q < 3) acc += w3[ab * 9 + p * 3 + q] * mult;
      }
    w4[i] = acc / 4.f;
  }
}
__global__ void fused_up_weight_bwd_kernel(const float* dw4, float* dw3, long long AB, float mult, int accumulate) {
  const long long total = AB * 9;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % 3), p = (int)((i / 3) % 3);
    const long long ab = i / 9;
    float acc = 0.f;
    // w3[p][q] contributes to w4[r][s] with r = p+1-dr, s = q+1-ds
    for (int dr = 0; dr < 2; ++dr)
      for (int ds = 0; ds < 2; ++ds) acc += dw4[ab * 16 + (p + 1 - dr) * 4 + (q + 1 - ds)];
    const float v = acc * mult / 4.f;
    dw3[i] = accumulate ? __fadd_rn(dw3[i], v) : v;
  }
}

#define LAUNCH_V(kern, total_of_v, C, ...)                                                                         \
  do {                                                                                                               \
    HWG_REQUIRE((long long)(total_of_v) < (1ll << 31), "tensor too large for the 32-bit indices of the resampling kernels"); \
    if ((C) % 4 == 0) hipLaunchKernelGGL(kern<4>, dim3(hwg_stream_grid((total_of_v) / 4, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern<1>, dim3(hwg_stream_grid((total_of_v), 256)), dim3(256), 0, st, __VA_ARGS__);           \
  } while (0)

#define LAUNCH_V2(kern, flag, total_of_v, C, ...)                                                                  \
  do {                                                                                                               \
    HWG_REQUIRE((long long)(total_of_v) < (1ll << 31), "tensor too large for the 32-bit indices of the resampling kernels"); \
    if ((C) % 4 == 0) hipLaunchKernelGGL((kern<4, flag>), dim3(hwg_stream_grid((total_of_v) / 4, 256)), dim3(256), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL((kern<1, flag>), dim3(hwg_stream_grid((total_of_v), 256)), dim3(256), 0, st, __VA_ARGS__);           \
  } while (0)

// Data gradient of a single-input-channel convolution (first layers), second half: dx[n,ih,iw] = sum_{r,s} t[n, ih+ph-r*dh, iw+pw-s*dw, r*S+s]
// where t = dy x W^T is the per-pixel tap matrix produced by a 1x1 convolution on the matrix cores (stride 1). Taps are summed in
// (r,s) order, one thread per input pixel.
__global__ __launch_bounds__(256) void col2im_taps_kernel(const float* __restrict__ t, float* __restrict__ dx, int N, int H, int W, int P, int Q, int R,
                                                          int S, int ph, int pw, int dh, int dw) {
  const int iw = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ih = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  if (iw >= W || ih >= H) return;
  const int RS = R * S;
  float acc = 0.f;
  for (int r = 0; r < R; ++r) {
    const int p = ih + ph - r * dh;
    if (p < 0 || p >= P) continue;
    const float* row = t + ((long long)n * P + p) * Q * RS + r * S;
    for (int s2 = 0; s2 < S; ++s2) {
      const int q = iw + pw - s2 * dw;
      if (q >= 0 && q < Q) acc += row[(long long)q * RS + s2];
    }
  }
  dx[((long long)n * H + ih) * W + iw] = acc;
}

// The same fold through LDS (dilation 1, square 3x3 / 5x5 / 7x7): in the kernel above neighbouring lanes read t[.., q, tap] R*S floats apart - 64
// cache lines per load instruction, every tap row of the tile re-fetched R x S times through L1 (52 us for 46 MB of t on the
// discriminator's 7x7 layer). Here a workgroup copies the (TH + R - 1) x (TW + S - 1) pixel block of t it needs into LDS with contiguous
// loads (pixels of one row are (TW + S - 1) * R*S consecutive floats) and every thread folds its output pixel from LDS: lanes are R*S floats
// apart there too, an odd stride - conflict free. Taps are added in the same (r, s) order (absent ones as zeros): bit-identical results.
template <int R, int S, int TH, int TW>
__global__ __launch_bounds__(256) void col2im_taps_lds_kernel(const float* __restrict__ t, float* __restrict__ dx, int N, int H, int W, int P, int Q, int ph,
                                                              int pw) {
  constexpr int RS = R * S, PH = TH + R - 1, PW = TW + S - 1, ROW = PW * RS;
  __shared__ float tile[PH * ROW];
  const int iw0 = blockIdx.x * TW, ih0 = blockIdx.y * TH, n = blockIdx.z;
  const int p_lo = ih0 + ph - (R - 1), q_lo = iw0 + pw - (S - 1);
  // (row pl of the tile is ROW consecutive floats of t starting at pixel q_lo; eight loads per thread in flight: a workgroup copies 12-60 KB and
  // one load at a time made the copy a chain of 47-59 memory latencies)
  constexpr int UL = 8;
  for (int i0 = threadIdx.x; i0 < PH * ROW; i0 += UL * 256) {
    float v[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
      const int i = i0 + u * 256;
      const int ic = i < PH * ROW ? i : i0;
      const int pl = ic / ROW, rem = ic - pl * ROW;
      const int p = p_lo + pl, q = q_lo + rem / RS;
      const bool ok = p >= 0 && p < P && q >= 0 && q < Q;
      const long long off = ((long long)n * P + (ok ? p : 0)) * Q * RS + (long long)(ok ? q_lo : 0) * RS + (ok ? rem : 0);
      const float x = t[off];
      v[u] = ok ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UL; ++u)
      if (i0 + u * 256 < PH * ROW) tile[i0 + u * 256] = v[u];
  }
  __syncthreads();
  if (threadIdx.x >= TH * TW) return;
  const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
  const int iw = iw0 + tx, ih = ih0 + ty;
  if (iw >= W || ih >= H) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int s2 = 0; s2 < S; ++s2) acc += tile[((ty + (R - 1) - r) * PW + (tx + (S - 1) - s2)) * RS + r * S + s2];
  dx[((long long)n * H + ih) * W + iw] = acc;
}

}  // namespace

extern "C" int hwg_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int kh, int kw, void* stream) {
  HWG_REQUIRE(x && y && N > 0 && H >= kh && W >= kw && C > 0 && kh > 0 && kw > 0, "avgpool_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int P = H / kh, Q = W / kw;
  LAUNCH_V(avgpool_fwd_kernel, (long long)N * P * Q * C, C, x, y, N, H, W, C, kh, kw, P, Q);
  HWG_LAUNCH_CHECK("avgpool_fwd");
  return HWG_OK;
}
extern "C" int hwg_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int kh, int kw, void* stream) {
  HWG_REQUIRE(dy && dx && N > 0 && H >= kh && W >= kw && C > 0 && kh > 0 && kw > 0, "avgpool_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int P = H / kh, Q = W / kw;
  LAUNCH_V(avgpool_bwd_kernel, (long long)N * H * W * C, C, dy, dx, N, H, W, C, kh, kw, P, Q);
  HWG_LAUNCH_CHECK("avgpool_bwd");
  return HWG_OK;
}

extern "C" int hwg_act_avgpool_fwd(const float* x, const float* chan_mask, float* y, int N, int H, int W, int C, int kh, int kw, int act, float slope,
                                   void* stream) {
  HWG_REQUIRE(x && y && N > 0 && H >= kh && W >= kw && C > 0 && kh > 0 && kw > 0, "act_avgpool_fwd: bad arguments");
  HWG_REQUIRE(act == 0 || act == HWG_ACT_RELU || act == HWG_ACT_LRELU, "act_avgpool_fwd: activation must be none / relu / leaky relu");
  hipStream_t st = (hipStream_t)stream;
  const int P = H / kh, Q = W / kw;
  LAUNCH_V(act_avgpool_fwd_kernel, (long long)N * P * Q * C, C, x, chan_mask, y, N, H, W, C, kh, kw, P, Q, act, slope);
  HWG_LAUNCH_CHECK("act_avgpool_fwd");
  return HWG_OK;
}
extern "C" int hwg_act_avgpool_bwd(const float* dy, const float* x, const float* chan_mask, float* dx, int N, int H, int W, int C, int kh, int kw,
                                   int act, float slope, void* stream) {
  HWG_REQUIRE(dy && x && dx && N > 0 && H >= kh && W >= kw && C > 0 && kh > 0 && kw > 0, "act_avgpool_bwd: bad arguments");
  HWG_REQUIRE(act == 0 || act == HWG_ACT_RELU || act == HWG_ACT_LRELU, "act_avgpool_bwd: activation must be none / relu / leaky relu");
  hipStream_t st = (hipStream_t)stream;
  const int P = H / kh, Q = W / kw;
  LAUNCH_V(act_avgpool_bwd_kernel, (long long)N * H * W * C, C, dy, x, chan_mask, dx, N, H, W, C, kh, kw, P, Q, act, slope);
  HWG_LAUNCH_CHECK("act_avgpool_bwd");
  return HWG_OK;
}

extern "C" int hwg_maxpool_fwd(const float* x, float* y, int* idx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                               int P, int Q, void* stream) {
  HWG_REQUIRE(x && y && idx && N > 0 && C > 0 && P > 0 && Q > 0, "maxpool_fwd: bad arguments");
  HWG_REQUIRE(P == (H + 2 * ph - kh) / sh + 1 && Q == (W + 2 * pw - kw) / sw + 1, "maxpool_fwd: inconsistent output size");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V2(maxpool_fwd_kernel, false, (long long)N * P * Q * C, C, x, y, idx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q);
  HWG_LAUNCH_CHECK("maxpool_fwd");
  return HWG_OK;
}
extern "C" int hwg_maxpool_relu_fwd(const float* x, float* y, int* idx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                                    int P, int Q, void* stream) {
  HWG_REQUIRE(x && y && idx && N > 0 && C > 0 && P > 0 && Q > 0, "maxpool_relu_fwd: bad arguments");
  HWG_REQUIRE(P == (H + 2 * ph - kh) / sh + 1 && Q == (W + 2 * pw - kw) / sw + 1, "maxpool_relu_fwd: inconsistent output size");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V2(maxpool_fwd_kernel, true, (long long)N * P * Q * C, C, x, y, idx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q);
  HWG_LAUNCH_CHECK("maxpool_relu_fwd");
  return HWG_OK;
}
extern "C" int hwg_maxpool_bwd(const float* dy, const int* idx, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                               int pw, int P, int Q, void* stream) {
  HWG_REQUIRE(dy && dx && idx && N > 0 && C > 0 && P > 0 && Q > 0, "maxpool_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V2(maxpool_bwd_kernel, false, (long long)N * H * W * C, C, dy, (const float*)nullptr, idx, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q);
  HWG_LAUNCH_CHECK("maxpool_bwd");
  return HWG_OK;
}
extern "C" int hwg_maxpool_relu_bwd(const float* dy, const float* y, const int* idx, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                                    int ph, int pw, int P, int Q, void* stream) {
  HWG_REQUIRE(dy && y && dx && idx && N > 0 && C > 0 && P > 0 && Q > 0, "maxpool_relu_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V2(maxpool_bwd_kernel, true, (long long)N * H * W * C, C, dy, y, idx, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q);
  HWG_LAUNCH_CHECK("maxpool_relu_bwd");
  return HWG_OK;
}

extern "C" int hwg_upsample_nearest_fwd(const float* x, float* y, int N, int H, int W, int C, int fh, int fw, void* stream) {
  HWG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && fh > 0 && fw > 0, "upsample_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V(upsample_fwd_kernel, (long long)N * H * fh * W * fw * C, C, x, y, N, H, W, C, fh, fw);
  HWG_LAUNCH_CHECK("upsample_fwd");
  return HWG_OK;
}
extern "C" int hwg_upsample_nearest_bwd(const float* dy, float* dx, int N, int H, int W, int C, int fh, int fw, void* stream) {
  HWG_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && fh > 0 && fw > 0, "upsample_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V(upsample_bwd_kernel, (long long)N * H * W * C, C, dy, dx, N, H, W, C, fh, fw);
  HWG_LAUNCH_CHECK("upsample_bwd");
  return HWG_OK;
}

extern "C" int hwg_blur3(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  HWG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "blur3: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V(blur3_kernel, (long long)N * H * W * C, C, x, y, N, H, W, C);
  HWG_LAUNCH_CHECK("blur3");
  return HWG_OK;
}

extern "C" int hwg_pad2d_fwd(const float* x, float* y, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, float value,
                             void* stream) {
  HWG_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "pad2d_fwd: bad arguments");
  HWG_REQUIRE(H + pt + pb > 0 && W + pl + pr > 0, "pad2d_fwd: empty output");
  HWG_REQUIRE(mode == 0 || (pt >= 0 && pb >= 0 && pl >= 0 && pr >= 0), "pad2d_fwd: replicate needs non-negative pads");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V(pad2d_fwd_kernel, (long long)N * (H + pt + pb) * (W + pl + pr) * C, C, x, y, N, H, W, C, pt, pb, pl, pr, mode, value);
  HWG_LAUNCH_CHECK("pad2d_fwd");
  return HWG_OK;
}
extern "C" int hwg_pad2d_bwd(const float* dy, float* dx, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, void* stream) {
  HWG_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0, "pad2d_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LAUNCH_V(pad2d_bwd_kernel, (long long)N * H * W * C, C, dy, dx, N, H, W, C, pt, pb, pl, pr, mode);
  HWG_LAUNCH_CHECK("pad2d_bwd");
  return HWG_OK;
}

extern "C" int hwg_copy_channels(const float* src, int Cs, int soff, float* dst, int Cd, int doff, int Cn, long long rows, int HW, int bcast,
                                 int accumulate, void* stream) {
  HWG_REQUIRE(src && dst && rows > 0 && Cn > 0 && soff >= 0 && doff >= 0 && soff + Cn <= Cs && doff + Cn <= Cd && HW > 0,
              "copy_channels: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  HWG_REQUIRE((long long)(rows * Cn) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(hwg_stream_grid(rows * Cn, 256)), dim3(256), 0, st, src, Cs, soff, dst, Cd, doff, Cn, rows, HW,
                     bcast, accumulate);
  HWG_LAUNCH_CHECK("copy_channels");
  return HWG_OK;
}
extern "C" int hwg_pad_channels(const float* src, int C, float* dst, int Cpad, long long rows, void* stream) {
  HWG_REQUIRE(src && dst && rows > 0 && C > 0 && Cpad >= C && Cpad % 4 == 0, "pad_channels: bad arguments");
  HWG_REQUIRE(rows * (Cpad / 4) < (1ll << 31), "pad_channels: tensor too large for 32-bit element indices");
  hipStream_t st = (hipStream_t)stream;
  const unsigned total4 = (unsigned)(rows * (Cpad / 4));
  hipLaunchKernelGGL(pad_channels_kernel, dim3(hwg_stream_grid(total4, 256)), dim3(256), 0, st, src, C, dst, Cpad, total4);
  HWG_LAUNCH_CHECK("pad_channels");
  return HWG_OK;
}
extern "C" int hwg_reduce_rows(const float* src, int Cs, int soff, float* out, int Cn, int N, int HW, int accumulate, void* stream) {
  HWG_REQUIRE(src && out && N > 0 && HW > 0 && Cn > 0 && soff >= 0 && soff + Cn <= Cs, "reduce_rows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(N, Cn), dim3(256), 0, st, src, Cs, soff, out, Cn, HW, accumulate);
  HWG_LAUNCH_CHECK("reduce_rows");
  return HWG_OK;
}
extern "C" int hwg_onehot(const int* label, float* out, int L, int B, int ncls, int Cd, int doff, void* stream) {
  HWG_REQUIRE(label && out && L > 0 && B > 0 && ncls > 0 && doff >= 0 && doff + ncls <= Cd, "onehot: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  HWG_REQUIRE((long long)((long long)B * L * ncls) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipLaunchKernelGGL(onehot_kernel, dim3(hwg_stream_grid((long long)B * L * ncls, 256)), dim3(256), 0, st, label, out, L, B, ncls, Cd, doff);
  HWG_LAUNCH_CHECK("onehot");
  return HWG_OK;
}
extern "C" int hwg_onehot_both(const int* label, float* out_blc, float* out_lbc, int L, int B, int ncls, void* stream) {
  HWG_REQUIRE(label && out_blc && out_lbc && L > 0 && B > 0 && ncls > 0, "onehot_both: bad arguments");
  HWG_REQUIRE((long long)((long long)B * L * ncls) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(onehot_both_kernel, dim3(hwg_stream_grid((long long)B * L * ncls, 256)), dim3(256), 0, st, label, out_blc, out_lbc, L, B, ncls);
  HWG_LAUNCH_CHECK("onehot_both");
  return HWG_OK;
}
extern "C" int hwg_permute4(const float* in, float* out, int d0, int d1, int d2, int d3, long long s0, long long s1, long long s2, long long s3,
                            void* stream) {
  HWG_REQUIRE(in && out && d0 > 0 && d1 > 0 && d2 > 0 && d3 > 0, "permute4: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  HWG_REQUIRE((long long)((long long)d0 * d1 * d2 * d3) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipLaunchKernelGGL(permute4_kernel, dim3(hwg_stream_grid((long long)d0 * d1 * d2 * d3, 256)), dim3(256), 0, st, in, out, d0, d1, d2, d3, s0, s1,
                     s2, s3);
  HWG_LAUNCH_CHECK("permute4");
  return HWG_OK;
}

extern "C" int hwg_fused_upsample_weight_fwd(const float* w3, float* w4, long long AB, float mult, void* stream) {
  HWG_REQUIRE(w3 && w4 && AB > 0, "fused_upsample_weight_fwd: bad arguments");
  HWG_REQUIRE((long long)(AB * 16) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipLaunchKernelGGL(fused_up_weight_fwd_kernel, dim3(hwg_stream_grid(AB * 16, 256)), dim3(256), 0, (hipStream_t)stream, w3, w4, AB, mult);
  HWG_LAUNCH_CHECK("fused_upsample_weight_fwd");
  return HWG_OK;
}
extern "C" int hwg_fused_upsample_weight_bwd(const float* dw4, float* dw3, long long AB, float mult, void* stream) {
  HWG_REQUIRE(dw4 && dw3 && AB > 0, "fused_upsample_weight_bwd: bad arguments");
  HWG_REQUIRE((long long)(AB * 9) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipLaunchKernelGGL(fused_up_weight_bwd_kernel, dim3(hwg_stream_grid(AB * 9, 256)), dim3(256), 0, (hipStream_t)stream, dw4, dw3, AB, mult, 0);
  HWG_LAUNCH_CHECK("fused_upsample_weight_bwd");
  return HWG_OK;
}
extern "C" int hwg_fused_upsample_weight_bwd_acc(const float* dw4, float* dw3, long long AB, float mult, void* stream) {
  HWG_REQUIRE(dw4 && dw3 && AB > 0, "fused_upsample_weight_bwd_acc: bad arguments");
  HWG_REQUIRE((long long)(AB * 9) < (1ll << 31), "tensor too large for 32-bit element indices");
  hipLaunchKernelGGL(fused_up_weight_bwd_kernel, dim3(hwg_stream_grid(AB * 9, 256)), dim3(256), 0, (hipStream_t)stream, dw4, dw3, AB, mult, 1);
  HWG_LAUNCH_CHECK("fused_upsample_weight_bwd_acc");
  return HWG_OK;
}

extern "C" int hwg_col2im_taps(const float* t, float* dx, int N, int H, int W, int P, int Q, int R, int S, int pad_h, int pad_w, int dil_h, int dil_w,
                               void* stream) {
  HWG_REQUIRE(t && dx && N > 0 && H > 0 && W > 0 && P > 0 && Q > 0 && R > 0 && S > 0 && dil_h > 0 && dil_w > 0, "col2im_taps: bad arguments");
  static const int lds_on = [] { const char* e = getenv("HWG_COL2IM_LDS"); return e && *e ? atoi(e) : 1; }();      // 0: the direct gather (A/B timing)
  if (lds_on && dil_h == 1 && dil_w == 1 && R == S && (R == 3 || R == 5 || R == 7) && (long long)N * P * Q * R * S < (1ll << 31)) {
    hipStream_t st = (hipStream_t)stream;
    if (R == 3) hipLaunchKernelGGL((col2im_taps_lds_kernel<3, 3, 8, 32>), dim3(hwg_cdiv(W, 32), hwg_cdiv(H, 8), N), dim3(256), 0, st, t, dx, N, H, W, P, Q, pad_h, pad_w);
    else if (R == 5) hipLaunchKernelGGL((col2im_taps_lds_kernel<5, 5, 8, 32>), dim3(hwg_cdiv(W, 32), hwg_cdiv(H, 8), N), dim3(256), 0, st, t, dx, N, H, W, P, Q, pad_h, pad_w);
    else hipLaunchKernelGGL((col2im_taps_lds_kernel<7, 7, 8, 16>), dim3(hwg_cdiv(W, 16), hwg_cdiv(H, 8), N), dim3(256), 0, st, t, dx, N, H, W, P, Q, pad_h, pad_w);
    HWG_LAUNCH_CHECK("col2im_taps_lds");
    return HWG_OK;
  }
  hipLaunchKernelGGL(col2im_taps_kernel, dim3(hwg_cdiv(W, 64), hwg_cdiv(H, 4), N), dim3(256), 0, (hipStream_t)stream, t, dx, N, H, W, P, Q, R, S, pad_h,
                     pad_w, dil_h, dil_w);
  HWG_LAUNCH_CHECK("col2im_taps");
  return HWG_OK;
}
