// Direct (non-MFMA) convolution kernels for the single-channel ends of the networks:
//   c1  : C == 1 input (first layers: D 7x7, encoders 5x5, HWR 3x3) and, with flipped taps, the data gradient
//         of K == 1 heads
//   to1 : K <= 2 outputs (D heads, spacer head, decoder/generator output convs) and the data gradient of first layers
//   wgrad_direct : weight gradients of both
// These are HBM/L2-bound reductions; a GEMM formulation would waste 15/16 of the MFMA work.
#include "hwg_common.h"

namespace {

struct DirK {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  int N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, P, Q;
  int accumulate;
};

// ---- C == 1 on the matrix cores: the taps are the contraction (im2col tile built on the fly, like the TAPN weight gradient) ----
// D[m = pixel][n = k] = sum_t A[m][t] * w[t][k]: a workgroup (4 wavefronts, one 32-pixel row block each) stages the filter once and walks
// 128-pixel tiles; per tile every thread gathers TP / 2 input values (neighbouring lanes = neighbouring pixels: coalesced, masked after an
// unconditional clamped load), then TP / 2 v_mfma_f32_32x32x2_f32 per 32 output channels. The VALU kernel below spends 16 FMAs per
// 4-byte weight read and runs at a third of the vector rate (23 TFLOP/s on the discriminator's 7x7 first layer, 34 here). K = 32 or 64,
// 36..64 taps (smaller filters stay on the VALU kernel: the gather phase dominates there).
template <int BN>
__global__ __launch_bounds__(256) void conv_c1_mfma_kernel(DirK a, int TP, int ntiles) {
  constexpr int BM = 128, NI = BN / 32;
  __shared__ __attribute__((aligned(16))) float As[64 * BM];
  __shared__ __attribute__((aligned(16))) float Bs[64 * BN];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int RS = a.R * a.S;
  for (int i = tid; i < TP * BN; i += 256) {
    const int t = i / BN, k = i - t * BN;
    Bs[i] = (t < RS && k < a.K) ? a.w[t * a.K + k] : 0.f;
  }
  const int px = tid & (BM - 1), thalf = tid >> 7;      // this thread gathers pixel px for the taps thalf, thalf + 2, ...
  const long long Mtot = (long long)a.N * a.P * a.Q;
  float bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bv[ni] = (a.bias && ni * 32 + l31 < a.K) ? a.bias[ni * 32 + l31] : 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long m = (long long)tile * BM + px;
    const bool mok = m < Mtot;
    const long long mm = mok ? m : 0;
    const int q = (int)(mm % a.Q);
    const long long t2 = mm / a.Q;
    const int p = (int)(t2 % a.P), n = (int)(t2 / a.P);
    const int ih0 = p * a.sh - a.ph, iw0 = q * a.sw - a.pw;
    const float* xn = a.x + (long long)n * a.H * a.W;
    __syncthreads();                                    // the previous tile's fragment reads are done (and Bs is staged)
    for (int t = thalf; t < TP; t += 2) {
      const int r = t / a.S, sx = t - r * a.S;
      const int ih = ih0 + r * a.dh, iw = iw0 + sx * a.dw;
      const bool ok = mok && t < RS && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      const float v = xn[(long long)min(max(ih, 0), a.H - 1) * a.W + min(max(iw, 0), a.W - 1)];
      As[t * BM + px] = ok ? v : 0.f;
    }
    __syncthreads();
    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
    for (int kk = 0; kk < TP; kk += 2) {
      const float af = As[(kk + lhi) * BM + wid * 32 + l31];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, Bs[(kk + lhi) * BN + ni * 32 + l31], acc[ni], 0, 0, 0);
    }
    // C/D layout of the 32x32 MFMA: col (k) = lane & 31, row (pixel) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const long long mo = (long long)tile * BM + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      if (mo >= Mtot) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int k = ni * 32 + l31;
        if (k >= a.K) continue;
        float* yo = a.y + mo * a.K + k;
        const float v = acc[ni][e] + bv[ni];
        *yo = a.accumulate ? *yo + v : v;
      }
    }
  }
}

// ---- C == 1, stride 1, dilation 1 (every first layer: discriminator 7x7, style extractor / perceptual encoder 5x5, recogniser 3x3) ------------
// The kernel above gathers its im2col tile tap by tap (49 clamped 4-byte loads per pixel, each with its own coordinate arithmetic and bounds
// test) and is bound by that gather: 83 us for 16x64x512 (1.5 TB/s of output) where the matrix cores need 19 us and the output 30. Here a
// workgroup owns 128 consecutive output pixels of ONE output row: the R input rows it reads are staged in LDS once (coalesced, zero-filled
// outside the image: R x (128 + S - 1) floats), a tap (r, s) of pixel m is then rows[r][m + s] - one conflict-free ds_read_b32 per MFMA
// operand with compile-time offsets - and the filter's B fragments live in registers for the whole launch (TP x BN / 64 values per lane).
// Wavefront w multiplies pixels 32 w .. 32 w + 31 by all BN output channels: TP / 2 steps of v_mfma_f32_32x32x2_f32 per 32 channels.
template <int R, int S, int BN>
__global__ __launch_bounds__(256) void conv_c1_rows_kernel(DirK a, int qtiles, int nseg) {
  constexpr int T = R * S, TP = (T + 1) & ~1, STEPS = TP / 2, NI = BN / 32, QT = 128, RW = QT + S - 1;
  __shared__ float rows[R * RW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  float bf[STEPS][NI];
#pragma unroll
  for (int st = 0; st < STEPS; ++st)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int t = 2 * st + lhi, k = ni * 32 + l31;
      bf[st][ni] = (t < T && k < a.K) ? a.w[t * a.K + k] : 0.f;
    }
  float bv[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bv[ni] = (a.bias && ni * 32 + l31 < a.K) ? a.bias[ni * 32 + l31] : 0.f;
  for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const int qt = seg % qtiles, t2 = seg / qtiles;
    const int p = t2 % a.P, n = t2 / a.P;
    const int q0 = qt * QT;
    __syncthreads();                                    // the previous segment's operand reads are done
    for (int i = tid; i < R * RW; i += 256) {
      const int r = i / RW, c = i - r * RW;
      const int ih = p - a.ph + r, iw = q0 - a.pw + c;
      const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
      rows[i] = ok ? a.x[((long long)n * a.H + ih) * a.W + iw] : 0.f;
    }
    __syncthreads();
    f32x16 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
    const float* base = rows + wid * 32 + l31;
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int t0 = 2 * st, t1 = 2 * st + 1 < T ? 2 * st + 1 : 2 * st;      // (the padding tap of an odd filter: weight 0, any valid address)
      const int o0 = (t0 / S) * RW + t0 % S, o1 = (t1 / S) * RW + t1 % S;
      const float af = base[lhi ? o1 : o0];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf[st][ni], acc[ni], 0, 0, 0);
    }
    // C/D layout of the 32x32 MFMA: col (k) = lane & 31, row (pixel) = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
    const long long rowbase = ((long long)n * a.P + p) * a.Q;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = q0 + wid * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
      if (q >= a.Q) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int k = ni * 32 + l31;
        if (k >= a.K) continue;
        float* yo = a.y + (rowbase + q) * a.K + k;
        const float v = acc[ni][e] + bv[ni];
        *yo = a.accumulate ? *yo + v : v;
      }
    }
  }
}

// ---- C == 1 : each thread computes 4 consecutive q pixels x 4 channels; weights [R*S][K] staged in LDS ----
__global__ __launch_bounds__(256) void conv_c1_kernel(DirK a, int KG, int PG) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [R*S][KG*4]
  const int K4 = KG * 4;
  for (int i = threadIdx.x; i < a.R * a.S * K4; i += 256) {
    const int tap = i / K4, k = i % K4;
    wsm[i] = (k < a.K) ? a.w[tap * a.K + k] : 0.f;
  }
  __syncthreads();
  const int kg = threadIdx.x % KG;
  const int pg = threadIdx.x / KG;
  if (pg >= PG) return;
  const int qtiles = (a.Q + PG * 4 - 1) / (PG * 4);
  const int bid = blockIdx.x;
  const int qt = bid % qtiles;
  const int t = bid / qtiles;
  const int p = t % a.P;
  const int n = t / a.P;
  const int q0 = (qt * PG + pg) * 4;
  if (q0 >= a.Q) return;
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* xn = a.x + (long long)n * a.H * a.W;
  for (int r = 0; r < a.R; ++r) {
    const int ih = p * a.sh - a.ph + r * a.dh;
    if (ih < 0 || ih >= a.H) continue;
    const float* xr = xn + (long long)ih * a.W;
    for (int s = 0; s < a.S; ++s) {
      const float4 wv = *reinterpret_cast<const float4*>(wsm + (r * a.S + s) * K4 + kg * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iw = (q0 + j) * a.sw - a.pw + s * a.dw;
        const float xv = (iw >= 0 && iw < a.W) ? xr[iw] : 0.f;
        acc[j].x += xv * wv.x; acc[j].y += xv * wv.y; acc[j].z += xv * wv.z; acc[j].w += xv * wv.w;
      }
    }
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int k0 = kg * 4;
  if (a.bias) {
    if (k0 + 0 < a.K) bv.x = a.bias[k0 + 0];
    if (k0 + 1 < a.K) bv.y = a.bias[k0 + 1];
    if (k0 + 2 < a.K) bv.z = a.bias[k0 + 2];
    if (k0 + 3 < a.K) bv.w = a.bias[k0 + 3];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int q = q0 + j;
    if (q >= a.Q) break;
    float* yo = a.y + (((long long)n * a.P + p) * a.Q + q) * a.K + k0;
    float v[4] = {acc[j].x + bv.x, acc[j].y + bv.y, acc[j].z + bv.z, acc[j].w + bv.w};
    if (k0 + 3 < a.K && (a.K & 3) == 0) {
      float4 o = make_float4(v[0], v[1], v[2], v[3]);
      if (a.accumulate) { const float4 old = *reinterpret_cast<float4*>(yo); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
      *reinterpret_cast<float4*>(yo) = o;
    } else {
      for (int e = 0; e < 4; ++e)
        if (k0 + e < a.K) yo[e] = a.accumulate ? yo[e] + v[e] : v[e];
    }
  }
}

// ---- K <= 2 : LPP lanes per output pixel (64 / LPP pixels per wave) sweep (tap, channel/4), reduction inside the lane group.
// LPP is the power of two covering the taps x channel-groups of a pixel: the generator's 16 -> 1 1x1 head keeps 16 pixels per wave busy
// (one wave per pixel left 60 of 64 lanes idle there), the discriminator's 3x3 256 -> 1 heads use the whole wave per pixel.
template <int LPP>
__global__ __launch_bounds__(256) void conv_to1_kernel(DirK a) {
  constexpr int PPW = 64 / LPP;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPP, li = lane % LPP;
  const long long wave = (blockIdx.x * 256LL + threadIdx.x) >> 6;
  const long long nwaves = (gridDim.x * 256LL) >> 6;
  const int C4 = a.C >> 2;
  const int E = a.R * a.S * C4;
  const long long M = (long long)a.N * a.P * a.Q;
  for (long long m0 = wave * PPW; m0 < M; m0 += nwaves * PPW) {
    const long long m = m0 + sub;
    const bool mv = m < M;
    const long long mm = mv ? m : 0;
    const int q = (int)(mm % a.Q);
    const long long t = mm / a.Q;
    const int p = (int)(t % a.P);
    const int n = (int)(t / a.P);
    float acc0 = 0.f, acc1 = 0.f;
    for (int e = li; e < E && mv; e += LPP) {
      const int tap = e / C4, c4 = e % C4;
      const int r = tap / a.S, s = tap % a.S;
      const int ih = p * a.sh - a.ph + r * a.dh;
      const int iw = q * a.sw - a.pw + s * a.dw;
      if (ih < 0 || ih >= a.H || iw < 0 || iw >= a.W) continue;
      const float4 xv = *reinterpret_cast<const float4*>(a.x + (((long long)n * a.H + ih) * a.W + iw) * a.C + c4 * 4);
      const float4 w0 = *reinterpret_cast<const float4*>(a.w + ((long long)tap * a.K + 0) * a.C + c4 * 4);
      acc0 += xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
      if (a.K > 1) {
        const float4 w1 = *reinterpret_cast<const float4*>(a.w + ((long long)tap * a.K + 1) * a.C + c4 * 4);
        acc1 += xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
      }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) {
      acc0 += __shfl_xor(acc0, o, 64);
      acc1 += __shfl_xor(acc1, o, 64);
    }
    if (li == 0 && mv) {
      float v0 = acc0 + (a.bias ? a.bias[0] : 0.f);
      float* yo = a.y + m * a.K;
      yo[0] = a.accumulate ? yo[0] + v0 : v0;
      if (a.K > 1) {
        float v1 = acc1 + (a.bias ? a.bias[1] : 0.f);
        yo[1] = a.accumulate ? yo[1] + v1 : v1;
      }
    }
  }
}

// 3x3 / stride 1 / one output channel on <= 64 input channels (the autoencoder decoder's image head): LPP lanes hold one tap's channel groups,
// the nine filter taps of a lane stay in registers and the nine input loads of a pixel are in flight together (the generic loop above
// issues them one by one behind a bounds test: 266 us at 28 x 64 x 512 x 64, latency bound at 0.9 TB/s).
template <int LPP>
__global__ __launch_bounds__(256) void conv_to1_3x3_kernel(DirK a) {
  constexpr int PPW = 64 / LPP;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPP, li = lane % LPP;
  const long long wave = (blockIdx.x * 256LL + threadIdx.x) >> 6;
  const long long nwaves = (gridDim.x * 256LL) >> 6;
  const bool active = li < (a.C >> 2);
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
    w[t] = active ? *reinterpret_cast<const float4*>(a.w + (long long)t * a.C + li * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float bias = a.bias ? a.bias[0] : 0.f;
  const long long M = (long long)a.N * a.P * a.Q;
  for (long long m0 = wave * PPW; m0 < M; m0 += nwaves * PPW) {
    const long long m = m0 + sub;
    const bool mv = m < M;
    const long long mm = mv ? m : 0;
    const int q = (int)(mm % a.Q);
    const long long t2 = mm / a.Q;
    const int p = (int)(t2 % a.P);
    const int n = (int)(t2 / a.P);
    float4 xv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ih = p - a.ph + r, iw = q - a.pw + s;
        const bool ok = mv && active && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        xv[r * 3 + s] = ok ? *reinterpret_cast<const float4*>(a.x + (((long long)n * a.H + ih) * a.W + iw) * a.C + li * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc += xv[t].x * w[t].x + xv[t].y * w[t].y + xv[t].z * w[t].z + xv[t].w * w[t].w;
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (li == 0 && mv) {
      const float v = acc + bias;
      a.y[m] = a.accumulate ? a.y[m] + v : v;
    }
  }
}

// ---- direct weight gradient: one side of the contraction has <= 2 channels ----
// big_on_anchor != 0 : u is [M][K] with K large, v is single channel (C == 1):   part[blk][tap][k]      = sum_m u[m][k] * v[g(m,tap)]
// big_on_anchor == 0 : u is [M][K<=2],          v is [.., C] with C large:        part[blk][tap][ko][c]  = sum_m u[m][ko] * v[g(m,tap)][c]
struct WdK {
  const float* u;
  const float* v;
  float* part;
  int N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, P, Q;
  int Mtot, chunk;
  int G;     // channel groups of 4 on the big side
  int TSL;   // tap slots
  int PL;    // pixel lanes
  int MAXT;  // taps per thread
};
constexpr int WD_MAXT = 8;

template <int BIG_ON_ANCHOR>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WdK a) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [PL][TSL*MAXT][NACC][G*4]
  const int tid = threadIdx.x;
  const int g = tid % a.G;
  const int slot = tid / a.G;
  const int tsl = slot % a.TSL;
  const int pl = slot / a.TSL;
  const bool active = pl < a.PL;
  const int RS = a.R * a.S;
  constexpr int NACC = BIG_ON_ANCHOR ? 1 : 2;
  float4 acc[WD_MAXT][NACC];
#pragma unroll
  for (int t = 0; t < WD_MAXT; ++t)
#pragma unroll
    for (int e = 0; e < NACC; ++e) acc[t][e] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int pb = blockIdx.x * a.chunk;
  const int pe = min(pb + a.chunk, a.Mtot);
  if (active) {
    for (int m = pb + pl; m < pe; m += a.PL) {
      const int q = m % a.Q;
      const int tt = m / a.Q;
      const int p = tt % a.P;
      const int n = tt / a.P;
      float4 ub = make_float4(0.f, 0.f, 0.f, 0.f);
      float us0 = 0.f, us1 = 0.f;
      if (BIG_ON_ANCHOR) {
        ub = *reinterpret_cast<const float4*>(a.u + (long long)m * a.K + g * 4);
      } else {
        us0 = a.u[(long long)m * a.K];
        if (a.K > 1) us1 = a.u[(long long)m * a.K + 1];
      }
#pragma unroll
      for (int t = 0; t < WD_MAXT; ++t) {
        const int tap = tsl + t * a.TSL;
        if (t >= a.MAXT || tap >= RS) break;
        const int r = tap / a.S, s = tap % a.S;
        const int ih = p * a.sh - a.ph + r * a.dh;
        const int iw = q * a.sw - a.pw + s * a.dw;
        if (ih < 0 || ih >= a.H || iw < 0 || iw >= a.W) continue;
        const long long vo = (((long long)n * a.H + ih) * a.W + iw) * a.C;
        if (BIG_ON_ANCHOR) {
          const float xv = a.v[vo];
          acc[t][0].x += ub.x * xv; acc[t][0].y += ub.y * xv; acc[t][0].z += ub.z * xv; acc[t][0].w += ub.w * xv;
        } else {
          const float4 vv = *reinterpret_cast<const float4*>(a.v + vo + g * 4);
          acc[t][0].x += us0 * vv.x; acc[t][0].y += us0 * vv.y; acc[t][0].z += us0 * vv.z; acc[t][0].w += us0 * vv.w;
          if (NACC > 1) {
            acc[t][NACC - 1].x += us1 * vv.x; acc[t][NACC - 1].y += us1 * vv.y;
            acc[t][NACC - 1].z += us1 * vv.z; acc[t][NACC - 1].w += us1 * vv.w;
          }
        }
      }
    }
  }
  // reduce pixel lanes through LDS, then write part[blk][tap][ko][ch]
  const int G4 = a.G * 4;
  const int per_pl = a.TSL * a.MAXT * NACC * G4;
  if (active) {
#pragma unroll
    for (int t = 0; t < WD_MAXT; ++t) {
      if (t >= a.MAXT) break;
#pragma unroll
      for (int e = 0; e < NACC; ++e)
        *reinterpret_cast<float4*>(red + pl * per_pl + (((tsl * a.MAXT + t) * NACC + e) * a.G + g) * 4) = acc[t][e];
    }
  }
  __syncthreads();
  const int nko = BIG_ON_ANCHOR ? 1 : a.K;
  const int nch = BIG_ON_ANCHOR ? a.K : a.C;  // big-side channel count (== G*4)
  float* pout = a.part + (long long)blockIdx.x * RS * nko * nch;
  for (int i = tid; i < RS * nko * nch; i += 256) {
    const int ch = i % nch;
    const int t2 = i / nch;
    const int ko = t2 % nko;
    const int tap = t2 / nko;
    const int ts = tap % a.TSL, tt = tap / a.TSL;
    float sum = 0.f;
    for (int l = 0; l < a.PL; ++l) sum += red[l * per_pl + ((ts * a.MAXT + tt) * NACC + ko) * G4 + ch];
    pout[i] = sum;
  }
}

// same final reduction as the MFMA path (duplicated signature; defined in conv_mfma.hip's TU would need export)
__global__ void wgrad_direct_reduce_kernel(const float* part, float* dw, int nblk, int RS, int S, int K, int C,
                                           long long sa, long long sb, long long sr, long long ss, int accumulate) {
  const long long total = (long long)RS * K * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int k = (int)(t % K);
    const int tap = (int)(t / K);
    float sum = 0.f;
    for (int b = 0; b < nblk; ++b) sum += part[(long long)b * total + i];
    const long long o = k * sa + c * sb + (tap / S) * sr + (tap % S) * ss;
    dw[o] = accumulate ? dw[o] + sum : sum;
  }
}

struct WdPlan { int big_on_anchor, G, TSL, PL, MAXT, chunk, nblk; size_t smem; };
bool plan_wd(const hwg_conv_desc* d, WdPlan* p) {
  const int RS = d->R * d->S;
  p->big_on_anchor = (d->K > 2) ? 1 : 0;
  const int bigc = p->big_on_anchor ? d->K : d->C;
  if (bigc % 4 != 0 || bigc > 1024) return false;
  if (p->big_on_anchor && d->C != 1) return false;
  p->G = bigc / 4;
  const int slots = 256 / p->G;
  if (slots < 1) return false;
  p->TSL = slots < RS ? slots : RS;
  p->PL = slots / p->TSL;
  p->MAXT = (RS + p->TSL - 1) / p->TSL;
  if (p->MAXT > WD_MAXT) return false;
  const long long Mtot = (long long)d->N * d->P * d->Q;
  long long nblk = (Mtot + 511) / 512;
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  long long chunk = (Mtot + nblk - 1) / nblk;
  nblk = (Mtot + chunk - 1) / chunk;
  p->chunk = (int)chunk;
  p->nblk = (int)nblk;
  const int NACC = p->big_on_anchor ? 1 : 2;
  p->smem = (size_t)p->PL * p->TSL * p->MAXT * NACC * p->G * 4 * sizeof(float);
  return p->smem <= 64 * 1024;
}

}  // namespace

int hwg_conv_c1_fwd_impl(const hwg_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int accumulate, hipStream_t st) {
  HWG_REQUIRE(d->C == 1, "conv_c1: C must be 1");
  // stride-1 square 3x3 / 5x5 / 7x7 filters with 32 or 64 output channels: input rows staged in LDS, filter in registers (conv_c1_rows_kernel)
  if (hwg_tune().c1_rows && (d->K == 32 || d->K == 64) && d->R == d->S && (d->R == 3 || d->R == 5 || d->R == 7) && d->stride_h == 1 && d->stride_w == 1 &&
      d->dil_h == 1 && d->dil_w == 1 && (long long)d->N * d->P * hwg_cdiv(d->Q, 128) < (1ll << 31)) {
    DirK k;
    k.x = x; k.w = w; k.bias = bias; k.y = y;
    k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
    k.sh = 1; k.sw = 1; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = 1; k.dw = 1;
    k.P = d->P; k.Q = d->Q; k.accumulate = accumulate;
    const int qtiles = hwg_cdiv(d->Q, 128);
    const int nseg = d->N * d->P * qtiles;
    const int grid = nseg < 2048 ? nseg : 2048;
    const int prof = hwg_prof_open(HWG_PROF_CONV_DIRECT, 2.0 * d->N * d->P * d->Q * d->K * d->C * d->R * d->S, st);
#define HWG_C1R(R_, BN_) hipLaunchKernelGGL((conv_c1_rows_kernel<R_, R_, BN_>), dim3(grid), dim3(256), 0, st, k, qtiles, nseg)
    if (d->R == 3 && d->K == 64) HWG_C1R(3, 64);
    else if (d->R == 3) HWG_C1R(3, 32);
    else if (d->R == 5 && d->K == 64) HWG_C1R(5, 64);
    else if (d->R == 5) HWG_C1R(5, 32);
    else if (d->K == 64) HWG_C1R(7, 64);
    else HWG_C1R(7, 32);
#undef HWG_C1R
    hwg_prof_close(prof, st);
    HWG_LAUNCH_CHECK("conv_c1_rows");
    return HWG_OK;
  }
  // the matrix-core kernel pays from ~36 taps (7x7: 128 -> 88 us on 16x64x512; 5x5 and 3x3 layers are no faster or slower: tools/probes/probe_r3_c1.txt)
  if (hwg_tune().c1_mfma && (d->K == 32 || d->K == 64) && d->R * d->S <= 64 && d->R * d->S >= 36) {
    DirK k;
    k.x = x; k.w = w; k.bias = bias; k.y = y;
    k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
    k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = d->dil_h; k.dw = d->dil_w;
    k.P = d->P; k.Q = d->Q; k.accumulate = accumulate;
    const int TP = (d->R * d->S + 1) & ~1;
    const long long M = (long long)d->N * d->P * d->Q;
    const int ntiles = (int)((M + 127) / 128);
    const int grid = ntiles < 768 ? ntiles : 768;       // three workgroups per CU (48 KB of LDS each) walk the tiles
    const int prof = hwg_prof_open(HWG_PROF_CONV_DIRECT, 2.0 * d->N * d->P * d->Q * d->K * d->C * d->R * d->S, st);
    if (d->K == 64) hipLaunchKernelGGL(conv_c1_mfma_kernel<64>, dim3(grid), dim3(256), 0, st, k, TP, ntiles);
    else hipLaunchKernelGGL(conv_c1_mfma_kernel<32>, dim3(grid), dim3(256), 0, st, k, TP, ntiles);
    hwg_prof_close(prof, st);
    HWG_LAUNCH_CHECK("conv_c1_mfma");
    return HWG_OK;
  }
  const int KG = (d->K + 3) / 4;
  HWG_REQUIRE(KG <= 256, "conv_c1: K=%d too large", d->K);
  const int PG = 256 / KG;
  const size_t smem = (size_t)d->R * d->S * KG * 4 * sizeof(float);
  HWG_REQUIRE(smem <= 64 * 1024, "conv_c1: weights do not fit LDS (%zu B)", smem);
  DirK k;
  k.x = x; k.w = w; k.bias = bias; k.y = y;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
  k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = d->dil_h; k.dw = d->dil_w;
  k.P = d->P; k.Q = d->Q; k.accumulate = accumulate;
  const int qtiles = hwg_cdiv(d->Q, PG * 4);
  const long long blocks = (long long)d->N * d->P * qtiles;
  const int prof = hwg_prof_open(HWG_PROF_CONV_DIRECT, 2.0 * d->N * d->P * d->Q * d->K * d->C * d->R * d->S, st);
  hipLaunchKernelGGL(conv_c1_kernel, dim3((unsigned)blocks), dim3(256), smem, st, k, KG, PG);
  hwg_prof_close(prof, st);
  HWG_LAUNCH_CHECK("conv_c1");
  return HWG_OK;
}

int hwg_conv_to1_fwd_impl(const hwg_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int accumulate, hipStream_t st) {
  HWG_REQUIRE(d->K <= 2, "conv_to1: K must be <= 2");
  HWG_REQUIRE(d->C % 4 == 0, "conv_to1: C %% 4 != 0 (C=%d)", d->C);
  DirK k;
  k.x = x; k.w = w; k.bias = bias; k.y = y;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
  k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = d->dil_h; k.dw = d->dil_w;
  k.P = d->P; k.Q = d->Q; k.accumulate = accumulate;
  const long long M = (long long)d->N * d->P * d->Q;
  const int E = d->R * d->S * (d->C / 4);
  // lanes per pixel: enough to cover the taps x channel groups of a small pixel, otherwise the channel groups of ONE tap (the lanes then walk
  // the taps: 4 or 16 pixels per wave instead of one - the autoencoder decoder's 3x3 64 -> 1 head ran one wave per pixel, 321 us at 28 lines)
  const int C4 = d->C / 4;
  int lpp = E <= 4 ? 4 : E <= 16 ? 16 : 64;
  if (hwg_tune().to1_lanes && E > 16 && C4 <= 16) lpp = C4 <= 4 ? 4 : 16;
  long long blocks = (M * lpp / 64 + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const int prof = hwg_prof_open(HWG_PROF_CONV_DIRECT, 2.0 * d->N * d->P * d->Q * d->K * d->C * d->R * d->S, st);
  const bool r3 = hwg_tune().to1_lanes && d->K == 1 && d->R == 3 && d->S == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 &&
                  C4 > 4 && C4 <= 16;
  if (r3) hipLaunchKernelGGL(conv_to1_3x3_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, k);
  else if (lpp == 4) hipLaunchKernelGGL(conv_to1_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, k);
  else if (lpp == 16) hipLaunchKernelGGL(conv_to1_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, k);
  else hipLaunchKernelGGL(conv_to1_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, st, k);
  hwg_prof_close(prof, st);
  HWG_LAUNCH_CHECK("conv_to1");
  return HWG_OK;
}

size_t hwg_conv_wgrad_direct_workspace(const hwg_conv_desc* d) {
  WdPlan p;
  if (!plan_wd(d, &p)) return 0;
  return (size_t)p.nblk * d->R * d->S * d->K * d->C * sizeof(float);
}

int hwg_conv_wgrad_direct_impl(const hwg_conv_desc* d, const float* u, const float* v, float* dw, long long sa, long long sb,
                               long long sr, long long ss, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  WdPlan p;
  HWG_REQUIRE(plan_wd(d, &p), "conv_wgrad(direct): unsupported shape K=%d C=%d R=%d S=%d", d->K, d->C, d->R, d->S);
  const size_t need = hwg_conv_wgrad_direct_workspace(d);
  if (!ws || ws_bytes < need) {
    hwg_set_error("conv_wgrad(direct): workspace too small (%zu < %zu)", ws_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  WdK k;
  k.u = u; k.v = v; k.part = (float*)ws;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.R = d->R; k.S = d->S;
  k.sh = d->stride_h; k.sw = d->stride_w; k.ph = d->pad_h; k.pw = d->pad_w; k.dh = d->dil_h; k.dw = d->dil_w;
  k.P = d->P; k.Q = d->Q;
  k.Mtot = d->N * d->P * d->Q;
  k.chunk = p.chunk; k.G = p.G; k.TSL = p.TSL; k.PL = p.PL; k.MAXT = p.MAXT;
  const int prof = hwg_prof_open(HWG_PROF_WGRAD_DIRECT, 2.0 * k.Mtot * d->K * d->C * d->R * d->S, st);
  if (p.big_on_anchor) hipLaunchKernelGGL(wgrad_direct_kernel<1>, dim3(p.nblk), dim3(256), p.smem, st, k);
  else hipLaunchKernelGGL(wgrad_direct_kernel<0>, dim3(p.nblk), dim3(256), p.smem, st, k);
  HWG_LAUNCH_CHECK("conv_wgrad_direct");
  const long long total = (long long)d->R * d->S * d->K * d->C;
  hipLaunchKernelGGL(wgrad_direct_reduce_kernel, dim3(hwg_stream_grid(total, 256)), dim3(256), 0, st, (const float*)ws, dw,
                     p.nblk, d->R * d->S, d->S, d->K, d->C, sa, sb, sr, ss, accumulate);
  hwg_prof_close(prof, st);
  HWG_LAUNCH_CHECK("conv_wgrad_direct_reduce");
  return HWG_OK;
}
