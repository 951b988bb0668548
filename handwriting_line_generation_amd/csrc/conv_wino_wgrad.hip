// Weight gradient of 3x3 stride-1 convolutions in the Winograd domain, F(3x3, 2x2) (the transpose of conv_wino.hip's F(2x2, 3x3)):
//   forward   Y = A^T [ (G g G^T) . (B^T d B) ] A        per 2x2 output tile, summed over input channels
//   backward  dU[pos][k][c] = sum over tiles  Z[pos][k] * V[pos][c],   Z = A dY A^T (4x4 from the 2x2 tile of dy),  V = B^T d B
//             dg[k][c]      = G^T dU[.][k][c] G                       (3x3 from 4x4, in the epilogue)
// 16 multiplies per tile and channel pair instead of 36: 2.25x fewer matrix-core FLOPs than the direct tap-by-tap product, and every
// input patch / dy tile is read ONCE for all nine taps (the direct kernel streams them once per tap).
//
// One workgroup (8 wavefronts) owns a 64 (k) x 64 (c) block of dU for all 16 positions - 64k fp32 accumulators, 128 registers per
// lane: wavefront w keeps positions 2w and 2w+1 as 4x4 blocks of 16x16 (v_mfma_f32_16x16x4_f32, contraction over tiles). A round
// covers 8 tiles (4 horizontally adjacent pairs): wavefronts 0-3 fetch the 4x6 input patch of one pair for 64 channels (one channel
// per lane: 256-byte coalesced rows), transform it in registers and store V, wavefronts 4-7 do the same for the 2x4 dy pixels (Z).
// LDS image of a round: [Z | V][16 positions][4 channel blocks][4 tile pairs][16 channels][2 tiles], 64 KB, double buffered (wg_elem: pair
// slots rotated by the block index, so the ds_write_b64 stores (lane = channel) and the ds_read_b64 fragment reads are conflict free). The matrix pipe sees 64 MFMAs per wavefront per round against ~200 VALU
// instructions of transform work, which the second wavefront of each SIMD overlaps.
// The epilogue applies G^T . G (two 32-row halves through an LDS exchange image), partial images go to [split][9][K][C] (+ K bias sums) and are
// summed by the direct kernel's fixed-order reduce (hwg_wgrad_reduce_launch).
#include "hwg_common.h"
#include <stdlib.h>
#ifndef HWG_WWG_SX
#define HWG_WWG_SX 8
#define HWG_WWG_SY 4
#endif

namespace {

struct WinoWgK {
  const float* x;    // [N,H,W,C]
  const float* dy;   // [N,P,Q,K]
  float* part;       // [nsplit][9 taps][K][C] (+ K bias sums): pstride floats per split
  int N, H, W, C, K, P, Q, ph, pw, TP, TQ2;
  int MP;            // tile pairs: N * TP * TQ2
  int kt, ct, nsplit;
  int sets;          // gradient sets: dy holds `sets` tensors [N,P,Q,K] back to back, x is shared; nsplit pixel ranges (and partial images) per set
  int bias_on;       // also emit the column sums of dy (the bias gradient) from the dy tiles that pass through anyway
  long long pstride; // taps*K*C + K
  // MT = 3 (4x4 stride-2 pad-0 layers as two-tap convolutions on the space-to-depth image, F(2x2 taps, 3x3 gradient tiles)): H, W, C describe the
  // VIRTUAL input [N, P+1, Q+1, 4*Creal]; its element (n, h, w, channel cv) lies at x + n x_img + h x_row + w x_pix + (cv / x_kc) x_run + cv % x_kc
  int x_img, x_row, x_pix, x_kc, x_run, Creal;
};

// LDS image of one position of one operand: [16-channel block b][slot (pair + b) & 3][channel & 15][2 tiles]. A fragment read (lane = row fr,
// pair fg) of block b covers its 128 floats exactly once - linear in the lane up to the rotation of the four 32-float segments - and the
// transform's ds_write_b64 (lane = channel, fixed pair) lands neighbouring 16-lane groups on opposite halves of the 64 banks.
__device__ __forceinline__ int wg_elem(int ch, int pair) {
  const int b = ch >> 4;
  return b * 128 + ((((pair + b) & 3) << 4) + (ch & 15)) * 2;
}

constexpr int WG_PLANE = 64 * 8;           // floats of one position of one operand
constexpr int WG_OPER = 16 * WG_PLANE;     // one operand, all positions
constexpr int WG_BUF = 2 * WG_OPER;        // Z then V

// One role's rounds. Stage r of the pipeline, all in one basic block so that the scheduler can interleave them (the matrix pipe needs 32
// cycles per MFMA and is shared by the two wavefronts of a SIMD; the ~8 other instructions per MFMA fit in its shadow):
//   A  fragment reads + 64 MFMAs on the LDS image of round r
//   B  transform of the elements fetched for round r+1 (in registers since the previous stage) -> other LDS image
//   C  global loads for round r+2 (in flight during the next stage's MFMAs)
// Rounds past the end of the pixel range run with everything masked (B writes zeros, C reads clamped addresses) - no control flow.
template <bool XROLE, int DBG, int MT>
__device__ __forceinline__ void wg_rounds(const WinoWgK& a, float* smem, f32x4 (&acc)[2][4][4], int widu, int lane, int ch0, int p_lo, int p_hi, int rounds, float& bias_sum) {
  // MT = 2: 3x3 taps, 2x2 gradient tiles (a pair = 4x6 patch, 2x4 dy pixels); MT = 3: 2x2 taps, 3x3 gradient tiles (4x7 patch, 3x6 dy pixels)
  constexpr int NQ = XROLE ? 2 * MT + 2 : 2 * MT;       // columns of a fetched pair
  constexpr int NR = (XROLE ? 4 : MT) * NQ;
  const int pairw = widu & 3;
  // Everything about a fetch except the channel is the same for the 64 lanes of a wavefront (they share the tile pair). The (row, column)
  // byte offsets and validity of the NR elements are computed across the lanes (lane i < NR = element i: ~15 VALU instructions for all of
  // them), then handed to the loads as scalar offsets (v_readlane); the loads are buffer loads with the channel in the vector offset.
  // Out-of-image elements are fetched from the clamped coordinate and multiplied by 0 in the transform, so the loads carry no control
  // flow and stay in flight behind the next stage's MFMAs. Lanes past the last channel repeat it (their rows of dU are never stored).
  const int chl = min(ch0 + lane, (XROLE ? a.C : a.K) - 1);
  const int voff = (XROLE && MT == 3) ? 4 * ((chl / a.x_kc) * a.x_run + chl % a.x_kc) : 4 * chl;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(XROLE ? a.x : a.dy), 0, 0x7fffffff, 0x00020000);
  const int rows = XROLE ? a.H : a.P, cols = XROLE ? a.W : a.Q, chans = XROLE ? a.C : a.K;
  const int el = lane < NR ? lane : 0;
  const int e_r = el / NQ, e_q = el - e_r * NQ;
  float raw[NR];
  float e_mask = 0.f;                     // lane i: 1.0 when element i of the fetched pair lies inside the image
  float bsum = 0.f;                       // dy role: running column sum of this lane's channel

  auto fetch = [&](int rnd) {
    const int p = p_lo + rnd * 4 + pairw;
    const bool pv = p < p_hi;
    const int pp = pv ? p : p_lo;
    const int t2 = pp / a.TQ2, tjp = pp - t2 * a.TQ2;
    const int n = t2 / a.TP, ti = t2 - n * a.TP;
    const int h0 = XROLE ? MT * ti - a.ph : MT * ti, w0 = XROLE ? 2 * MT * tjp - a.pw : 2 * MT * tjp;
    const int hr = h0 + e_r, wq = w0 + e_q;                         // per lane
    const bool ok = pv && (unsigned)hr < (unsigned)rows && (unsigned)wq < (unsigned)cols;
    int off;
    if constexpr (XROLE && MT == 3) off = 4 * (n * a.x_img + min(max(hr, 0), rows - 1) * a.x_row + min(max(wq, 0), cols - 1) * a.x_pix);
    else off = ((n * rows + min(max(hr, 0), rows - 1)) * cols + min(max(wq, 0), cols - 1)) * (4 * chans);
    e_mask = ok ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i)
      raw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, __builtin_amdgcn_readlane(off, i), 0));
  };
  // transform the fetched pair and store its two tiles (one ds_write_b64 per position)
  auto transform_store = [&](int buf) {
    float* dst = smem + buf * WG_BUF + (XROLE ? WG_OPER : 0) + wg_elem(lane, pairw);
    float d[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) d[i] = raw[i] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_mask), i));
    if (XROLE) {
      float t[4][NQ];                     // column transform (B^T over the rows) of the patch columns
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float d0 = d[q], d1 = d[NQ + q], d2 = d[2 * NQ + q], d3 = d[3 * NQ + q];
        t[0][q] = d0 - d2; t[1][q] = d1 + d2; t[2][q] = d2 - d1; t[3][q] = d1 - d3;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // tile 0 uses columns 0..3, tile 1 columns MT..MT+3 (the patches of neighbouring tiles overlap by 4 - MT columns)
        *reinterpret_cast<float2*>(dst + (i * 4 + 0) * WG_PLANE) = make_float2(t[i][0] - t[i][2], t[i][MT] - t[i][MT + 2]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 1) * WG_PLANE) = make_float2(t[i][1] + t[i][2], t[i][MT + 1] + t[i][MT + 2]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 2) * WG_PLANE) = make_float2(t[i][2] - t[i][1], t[i][MT + 2] - t[i][MT + 1]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 3) * WG_PLANE) = make_float2(t[i][1] - t[i][3], t[i][MT + 1] - t[i][MT + 3]);
      }
    } else if constexpr (MT == 3) {
      // Z = A dY A^T, A = [[1,0,0],[1,1,1],[1,-1,1],[0,0,1]] (the transpose of F(3x3,2x2)'s output transform); d = rows (y0, y1, y2) x
      // columns (tile 0: 0..2, tile 1: 3..5)
      float rr[4][6];
      float colsum = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float y0 = d[q], y1 = d[6 + q], y2 = d[12 + q];
        colsum += (y0 + y1) + y2;
        rr[0][q] = y0; rr[1][q] = (y0 + y1) + y2; rr[2][q] = (y0 - y1) + y2; rr[3][q] = y2;
      }
      bsum += colsum;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float2*>(dst + (i * 4 + 0) * WG_PLANE) = make_float2(rr[i][0], rr[i][3]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 1) * WG_PLANE) = make_float2((rr[i][0] + rr[i][1]) + rr[i][2], (rr[i][3] + rr[i][4]) + rr[i][5]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 2) * WG_PLANE) = make_float2((rr[i][0] - rr[i][1]) + rr[i][2], (rr[i][3] - rr[i][4]) + rr[i][5]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 3) * WG_PLANE) = make_float2(rr[i][2], rr[i][5]);
      }
    } else {
      // Z = A dY A^T, A = [[1,0],[1,1],[1,-1],[0,-1]]; d = rows (y0, y1) x columns (tile 0: 0,1; tile 1: 2,3)
      float rr[4][4];
      bsum += ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float y0 = d[q], y1 = d[4 + q];
        rr[0][q] = y0; rr[1][q] = y0 + y1; rr[2][q] = y0 - y1; rr[3][q] = -y1;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<float2*>(dst + (i * 4 + 0) * WG_PLANE) = make_float2(rr[i][0], rr[i][2]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 1) * WG_PLANE) = make_float2(rr[i][0] + rr[i][1], rr[i][2] + rr[i][3]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 2) * WG_PLANE) = make_float2(rr[i][0] - rr[i][1], rr[i][2] - rr[i][3]);
        *reinterpret_cast<float2*>(dst + (i * 4 + 3) * WG_PLANE) = make_float2(-rr[i][1], -rr[i][3]);
      }
    }
  };

  // fragment addressing: lane -> (row r of the 16-channel block, 2-tile slot g): MFMA e of a round contracts tiles {e, 2+e, 4+e, 6+e}
  const int fr = lane & 15, fg = lane >> 4;
  int f_off[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) f_off[b] = wg_elem(b * 16 + fr, fg);

  fetch(0);
  transform_store(0);
  fetch(1);
  __syncthreads();
  for (int rnd = 0; rnd < rounds; ++rnd) {
    const int cur = rnd & 1;
    const float* Zb = smem + cur * WG_BUF + (2 * widu) * WG_PLANE;
    const float* Vb = Zb + WG_OPER;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float2 af[4], bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        af[b] = *reinterpret_cast<const float2*>(Zb + p * WG_PLANE + f_off[b]);
        bf[b] = *reinterpret_cast<const float2*>(Vb + p * WG_PLANE + f_off[b]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (DBG & 1) acc[p][i][j][0] += af[i].x * bf[j].x;
          else acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[p][i][j], 0, 0, 0);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (DBG & 1) acc[p][i][j][1] += af[i].y * bf[j].y;
          else acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[p][i][j], 0, 0, 0);
        }
    }
    if (!(DBG & 2)) {
      transform_store(cur ^ 1);
      fetch(rnd + 2);
    }
    // interleave: one MFMA, then up to 8 of anything else (VALU / SALU / VMEM / DS) that is ready
#pragma unroll
    for (int g = 0; g < 64; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x096, XROLE ? HWG_WWG_SX : HWG_WWG_SY, 0);
    }
    __syncthreads();
  }
  bias_sum = bsum;
}

template <int DBG, int MT = 2>
__global__ __launch_bounds__(512) void wino_wgrad_kernel(WinoWgK a_in) {
  WinoWgK a = a_in;
  constexpr int LDC = 68;                 // exchange row stride: the four row groups of a C/D block land 16 banks apart
  constexpr int XCH = 16 * 32 * LDC;      // epilogue exchange image [16 positions][32 k][64 c], aliases the operand buffers
  __shared__ __attribute__((aligned(16))) float smem[(2 * WG_BUF > XCH + 256) ? 2 * WG_BUF : XCH + 256];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // work item: every XCD takes a contiguous run of the (split, k-tile, c-tile) sequence - the workgroups of one pixel range share
  // their x / dy reads through that XCD's L2
  const int total = a.kt * a.ct * a.nsplit * a.sets;
  int w = blockIdx.x;
  {
    const int xcd = w & 7, slot = w >> 3;
    const int lo = (int)((long long)xcd * total >> 3), hi = (int)((long long)(xcd + 1) * total >> 3);
    w = lo + slot;
    if (w >= hi) return;
  }
  const int c0 = (w % a.ct) * 64;
  const int k0 = ((w / a.ct) % a.kt) * 64;
  const int split_all = w / (a.ct * a.kt);        // (set, pixel range of that set)
  const int set = split_all / a.nsplit;
  const int split = split_all - set * a.nsplit;
  a.dy += (long long)set * a.N * a.P * a.Q * a.K;
  const int p_lo = (int)((long long)a.MP * split / a.nsplit);
  const int p_hi = (int)((long long)a.MP * (split + 1) / a.nsplit);
  const int rounds = (p_hi - p_lo + 3) >> 2;
  const int widu = __builtin_amdgcn_readfirstlane(wid);

  f32x4 acc[2][4][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // wavefronts 0-3 fetch and transform input patches (V), 4-7 the dy tiles (Z); all eight run the MFMAs of their two positions
  float bias_sum = 0.f;
  if (widu < 4) wg_rounds<true, DBG, MT>(a, smem, acc, widu, lane, c0, p_lo, p_hi, rounds, bias_sum);
  else wg_rounds<false, DBG, MT>(a, smem, acc, widu, lane, k0, p_lo, p_hi, rounds, bias_sum);

  // ---- epilogue: dg = G^T dU G (16 positions -> 9 taps) before anything leaves the CU: the positions of a (k, c) pair live in eight
  // wavefronts, so the blocks cross through LDS (two halves of 32 k rows); partial image [split][tap][K][C] like the direct kernel's.
  // C/D layout of a 16x16 block: column (c) = lane & 15, row (k) = (lane >> 4) * 4 + e
  float* X = smem;
  float* out = a.part + (long long)split_all * a.pstride;
  if constexpr ((DBG & 4) != 0) {          // timing ablation (HWG_WWG_DEBUG=4, garbage results): no output transform, no partial-image stores
    float keep = 0.f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) keep += acc[p][i][j][0] + acc[p][i][j][1] + acc[p][i][j][2] + acc[p][i][j][3];
    if (keep == 123456.789f) out[tid] = keep;
    return;
  }
  if (a.bias_on && c0 == 0) {              // the four dy wavefronts saw every dy element of this k-tile and pixel range exactly once
    float* bs = smem + XCH;
    __syncthreads();
    if (widu >= 4) bs[(widu - 4) * 64 + lane] = bias_sum;
    __syncthreads();
    if (tid < 64 && k0 + tid < a.K) out[a.pstride - a.K + k0 + tid] = (bs[tid] + bs[64 + tid]) + (bs[128 + tid] + bs[192 + tid]);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            X[((2 * wid + p) * 32 + i2 * 16 + (lane >> 4) * 4 + e) * LDC + j * 16 + (lane & 15)] = acc[p][2 * h + i2][j][e];
    __syncthreads();
    const int c = tid & 63;
#pragma unroll 1
    for (int kl = tid >> 6; kl < 32; kl += 8) {
      const int kk = k0 + h * 32 + kl;
      if (kk >= a.K || c0 + c >= a.C) continue;
      if constexpr (MT == 3) {
        // dg = G^T (S dU' S) G, G = [[1,0],[.5,.5],[.5,-.5],[0,1]], S = diag(1,1,1,-1) (the kernel's input transform is F(2x2,3x3)'s, whose fourth
        // row is the negative of F(3x3,2x2)'s); tap (u, v) of virtual channel (a, b, c) is tap (2u + a, 2v + b) of the real 4x4 filter
        float t2[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u0 = X[((0 * 4 + j) * 32 + kl) * LDC + c], u1 = X[((1 * 4 + j) * 32 + kl) * LDC + c];
          const float u2 = X[((2 * 4 + j) * 32 + kl) * LDC + c], u3 = X[((3 * 4 + j) * 32 + kl) * LDC + c];
          t2[0][j] = u0 + 0.5f * (u1 + u2);
          t2[1][j] = 0.5f * (u1 - u2) - u3;
        }
        const int cv = c0 + c, ab = cv / a.Creal, cr = cv - ab * a.Creal, pa = ab >> 1, pb = ab & 1;
        const long long plane = (long long)a.K * a.Creal;
        float* o = out + (long long)kk * a.Creal + cr;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          o[((2 * u + pa) * 4 + pb) * plane] = t2[u][0] + 0.5f * (t2[u][1] + t2[u][2]);
          o[((2 * u + pa) * 4 + 2 + pb) * plane] = 0.5f * (t2[u][1] - t2[u][2]) - t2[u][3];
        }
        continue;
      }
      float t[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u0 = X[((0 * 4 + j) * 32 + kl) * LDC + c], u1 = X[((1 * 4 + j) * 32 + kl) * LDC + c];
        const float u2 = X[((2 * 4 + j) * 32 + kl) * LDC + c], u3 = X[((3 * 4 + j) * 32 + kl) * LDC + c];
        t[0][j] = u0 + 0.5f * (u1 + u2);
        t[1][j] = 0.5f * (u1 - u2);
        t[2][j] = 0.5f * (u1 + u2) + u3;
      }
      float* o = out + (long long)kk * a.C + c0 + c;
      const long long plane = (long long)a.K * a.C;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        o[(r * 3 + 0) * plane] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
        o[(r * 3 + 1) * plane] = 0.5f * (t[r][1] - t[r][2]);
        o[(r * 3 + 2) * plane] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
      }
    }
  }
}

}  // namespace

int hwg_wgrad_reduce_launch(const float* part, float* dw, int nsplit, int RS, int S, int K, int C, long long sa, long long sb, long long sr,
                            long long ss, int accumulate, long long pstride, float* dbias, int bias_accumulate, hipStream_t st, bool defer);
bool hwg_wgrad_defer_take();

namespace {
struct WgPlan { int kt, ct, nsplit, MP; };

// 4x4 stride-2 pad-0 layers: the weight gradient of the stride-1 two-tap convolution on the space-to-depth image (conv_wino.hip, hwg_wino_s2_*),
// F(2x2 taps, 3x3 gradient tiles) - the same 16 multiplies per tile and channel pair for 36 products
bool wgrad_is_s2(const hwg_conv_desc* d) {
  return d && d->R == 4 && d->S == 4 && d->stride_h == 2 && d->stride_w == 2 && d->pad_h == 0 && d->pad_w == 0 && d->dil_h == 1 && d->dil_w == 1 &&
         !d->transposed && d->H >= 4 && d->W >= 4 && d->P == (d->H - 4) / 2 + 1 && d->Q == (d->W - 4) / 2 + 1;
}

WgPlan plan_wino_wgrad(const hwg_conv_desc* d, int sets = 1) {
  WgPlan p;
  const bool s2 = wgrad_is_s2(d);
  const int mt = s2 ? 3 : 2;
  p.kt = hwg_cdiv(d->K, 64);
  p.ct = hwg_cdiv(s2 ? 4 * d->C : d->C, 64);
  const int TP = hwg_cdiv(d->P, mt), TQ2 = hwg_cdiv(hwg_cdiv(d->Q, mt), 2);
  p.MP = d->N * TP * TQ2;
  const int rounds_all = hwg_cdiv(p.MP, 4);
  // one workgroup per CU (128 KB of LDS): as many pixel ranges as fill the 256 CUs once, at least 4 rounds each
  int ns = 256 / (p.kt * p.ct * sets);     // (several gradient sets in one launch share the chip: fewer, longer pixel ranges per set)
  if (ns < 1) ns = 1;
  const int cap = rounds_all / 4 > 1 ? rounds_all / 4 : 1;
  if (ns > cap) ns = cap;
  const int force = hwg_tune().wino_wgrad_split;
  if (force > 0) ns = force < rounds_all ? force : rounds_all;
  p.nsplit = ns;
  return p;
}

}  // namespace

extern "C" int hwg_wino_wgrad_supported(const hwg_conv_desc* d) {
  if (!d) return 0;
  if (wgrad_is_s2(d))
    return hwg_tune().wino_s2 != 0 && d->K >= 16 && d->C >= 16 && d->C % 4 == 0 && (long long)d->N * d->H * d->W * d->C < (1ll << 29) &&
           (long long)d->N * d->P * d->Q * d->K < (1ll << 29);
  return d->R == 3 && d->S == 3 && d->stride_h == 1 && d->stride_w == 1 && d->dil_h == 1 && d->dil_w == 1 && !d->transposed &&
         d->K >= 16 && d->C >= 16 && d->P == d->H + 2 * d->pad_h - 2 && d->Q == d->W + 2 * d->pad_w - 2 && d->P >= 1 && d->Q >= 1 &&
         (long long)d->N * d->H * d->W * d->C < (1ll << 29) && (long long)d->N * d->P * d->Q * d->K < (1ll << 29);   // 32-bit byte offsets
}

extern "C" int hwg_wino_wgrad_preferred(const hwg_conv_desc* d) {
  const int mode = hwg_tune().wino_wgrad;      // 0 never, 2 always (tests), default: the rule below
  if (mode == 0 || !hwg_wino_wgrad_supported(d)) return 0;
  if (wgrad_is_s2(d)) {
    if (hwg_tune().wino_s2 == 2 || mode == 2) return 1;
    const long long tiles3 = (long long)d->N * hwg_cdiv(d->P, 3) * hwg_cdiv(d->Q, 3);
    return d->K >= 48 && 4 * d->C >= 48 && tiles3 * hwg_cdiv(d->K, 64) * hwg_cdiv(4 * d->C, 64) >= 256 * 28;
  }
  if (mode == 2) return 1;
  // the 64 x 64 block wastes matrix-core work on narrower layers, and short pixel ranges cannot amortise the 16-position epilogue
  // (tools/conv_probe.py WGRAD=1, round 3: 8x8x122x128->128 39.5 vs 45.1 us and 8x16x122x128->64 37.2 vs 43.4 us in favour of this kernel
  //  at 7808 tile-blocks; 16x5x64x128->256, three output rows, 46.8 vs 36.0 us against it)
  const long long tiles = (long long)d->N * hwg_cdiv(d->P, 2) * hwg_cdiv(d->Q, 2);
  const int blocks = hwg_cdiv(d->K, 64) * hwg_cdiv(d->C, 64);
  if (d->P <= 3 && blocks <= 8) return 0;
  return d->K >= 48 && d->C >= 48 && tiles * blocks >= 256 * 28;
}

extern "C" size_t hwg_wino_wgrad_workspace(const hwg_conv_desc* d) {
  if (!hwg_wino_wgrad_supported(d)) return 0;
  const WgPlan p = plan_wino_wgrad(d);
  return (size_t)p.nsplit * ((size_t)d->R * d->S * d->K * d->C + d->K) * sizeof(float);
}

extern "C" size_t hwg_wino_wgrad_sets_workspace(const hwg_conv_desc* d, int sets) {
  if (!hwg_wino_wgrad_supported(d) || sets < 1) return 0;
  const WgPlan p = plan_wino_wgrad(d, sets);
  return (size_t)sets * p.nsplit * ((size_t)d->R * d->S * d->K * d->C + d->K) * sizeof(float);
}

static int wino_wgrad_run(const hwg_conv_desc* d, const float* dy, const float* x, int sets, float* const* dws, long long sa, long long sb,
                          long long sr, long long ss, int accumulate, float* const* dbiases, int bias_accumulate, void* workspace,
                          size_t workspace_bytes, hipStream_t st, bool defer) {
  HWG_REQUIRE(d && dy && x && dws && dws[0] && sets >= 1, "wino_wgrad: null pointer");
  HWG_REQUIRE(hwg_wino_wgrad_supported(d), "wino_wgrad: needs a 3x3 stride-1 dilation-1 (or 4x4 stride-2 pad-0) convolution with K, C >= 16");
  const size_t need = hwg_wino_wgrad_sets_workspace(d, sets);
  if (!workspace || workspace_bytes < need) {
    hwg_set_error("wino_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  float* const dbias = dbiases ? dbiases[0] : nullptr;
  const WgPlan p = plan_wino_wgrad(d, sets);
  WinoWgK k;
  k.x = x; k.dy = dy; k.part = (float*)workspace;
  const bool s2 = wgrad_is_s2(d);
  const int mt = s2 ? 3 : 2, taps = d->R * d->S;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.P = d->P; k.Q = d->Q; k.ph = d->pad_h; k.pw = d->pad_w;
  k.x_img = k.x_row = k.x_pix = k.x_kc = k.x_run = 0; k.Creal = d->C;
  if (s2) {      // the virtual stride-1 two-tap problem: input [N, P+1, Q+1, 4C] read through strides (a row of it is a row pair of x)
    k.H = d->P + 1; k.W = d->Q + 1; k.C = 4 * d->C;
    k.x_img = d->H * d->W * d->C; k.x_row = 2 * d->W * d->C; k.x_pix = 2 * d->C; k.x_kc = 2 * d->C; k.x_run = d->W * d->C;
  }
  k.TP = hwg_cdiv(d->P, mt); k.TQ2 = hwg_cdiv(hwg_cdiv(d->Q, mt), 2);
  k.MP = p.MP; k.kt = p.kt; k.ct = p.ct; k.nsplit = p.nsplit; k.sets = sets;
  k.bias_on = dbias ? 1 : 0;
  k.pstride = (long long)taps * d->K * d->C + d->K;
  const int total = p.kt * p.ct * p.nsplit * sets;
  int prof = hwg_prof_open(HWG_PROF_WGRAD_WINO, 2.0 * sets * d->N * d->P * d->Q * (double)d->K * d->C * taps, st);
  const int dbg = hwg_tune().wwg_debug;
  if (s2) hipLaunchKernelGGL((wino_wgrad_kernel<0, 3>), dim3((total + 7) / 8 * 8), dim3(512), 0, st, k);
  else if (dbg == 1) hipLaunchKernelGGL(wino_wgrad_kernel<1>, dim3((total + 7) / 8 * 8), dim3(512), 0, st, k);
  else if (dbg == 2) hipLaunchKernelGGL(wino_wgrad_kernel<2>, dim3((total + 7) / 8 * 8), dim3(512), 0, st, k);
  else if (dbg == 4) hipLaunchKernelGGL(wino_wgrad_kernel<4>, dim3((total + 7) / 8 * 8), dim3(512), 0, st, k);
  else hipLaunchKernelGGL(wino_wgrad_kernel<0>, dim3((total + 7) / 8 * 8), dim3(512), 0, st, k);
  hwg_prof_close(prof, st);
  hwg_note_plan(HWG_PROF_WGRAD_WINO, s2 ? 36 : 0, p.nsplit);
  HWG_LAUNCH_CHECK("wino_wgrad");
  // the partial images have the direct kernel's layout ([split][tap][K][C] + K bias sums): same fixed-order reduce into the weight's layout
  for (int s_ = 0; s_ < sets; ++s_) {
    prof = hwg_prof_open(HWG_PROF_WGRAD_REDUCE, (double)need / sets, st);
    const int rc = hwg_wgrad_reduce_launch((const float*)workspace + (long long)s_ * p.nsplit * k.pstride, dws[s_], p.nsplit, taps, d->S, d->K, d->C, sa, sb, sr,
                                           ss, accumulate, k.pstride, dbiases ? dbiases[s_] : nullptr, bias_accumulate, st, defer);
    hwg_prof_close(prof, st);
    if (rc) return rc;
  }
  return HWG_OK;
}

extern "C" int hwg_wino_wgrad(const hwg_conv_desc* d, const float* dy, const float* x, float* dw, long long sa, long long sb, long long sr,
                              long long ss, int accumulate, float* dbias, int bias_accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  const bool defer = hwg_wgrad_defer_take();
  float* dws[1] = {dw};
  float* dbs[1] = {dbias};
  return wino_wgrad_run(d, dy, x, 1, dws, sa, sb, sr, ss, accumulate, dbias ? dbs : nullptr, bias_accumulate, workspace, workspace_bytes,
                        (hipStream_t)stream, defer);
}

extern "C" int hwg_wino_wgrad_sets(const hwg_conv_desc* d, const float* dy, const float* x, int sets, const long long* dw_ptrs, long long sa,
                                   long long sb, long long sr, long long ss, int accumulate, const long long* dbias_ptrs, int bias_accumulate,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  const bool defer = hwg_wgrad_defer_take();
  HWG_REQUIRE(sets >= 1 && sets <= 8 && dw_ptrs, "wino_wgrad_sets: 1..8 sets");
  float* dws[8]; float* dbs[8];
  for (int i = 0; i < sets; ++i) { dws[i] = (float*)(uintptr_t)dw_ptrs[i]; dbs[i] = dbias_ptrs ? (float*)(uintptr_t)dbias_ptrs[i] : nullptr; }
  return wino_wgrad_run(d, dy, x, sets, dws, sa, sb, sr, ss, accumulate, dbias_ptrs ? dbs : nullptr, bias_accumulate, workspace, workspace_bytes,
                        (hipStream_t)stream, defer);
}
