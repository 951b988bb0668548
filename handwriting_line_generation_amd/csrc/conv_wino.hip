// Winograd F(2x2, 3x3) convolution for gfx950, fused into one kernel per direction:
//
//   y = A^T [ sum_c (G g G^T)[k][c] (.) (B^T d B)[c] ] A        (3x3, stride 1, dilation 1)
//
// 16 independent GEMMs (one per position of the 4x4 transform domain) with M = 2x2 output tiles, N = output channels,
// contraction = input channels: 2.25x fewer matrix-core MACs than the direct form, exact fp32 products, transform
// coefficients 0, +-1, +-1/2 (error growth ~2x of the direct sum; parity tests hold it to the same 1e-4).
//
//   wino_conv_kernel   forward and data gradient (the data gradient of a 3x3 stride-1 conv is the same conv with mirrored taps and
//                      swapped channel roles, which the weight transform absorbs)
//   wino_wgrad_kernel  weight gradient: dU[pos][k][c] = sum_tiles (A dy A^T)[pos][tile][k] * (B^T x B)[pos][tile][c], dw = G^T dU G
//
// Nothing of the transform domain ever touches HBM: the input patches are transformed in registers on their way into LDS
// (column transform per lane, row transform across the 4 lanes of a quad with DPP), the pre-transformed filters U stream
// through LDS, the 16 accumulator sets of one (tile, k) live in ONE lane (v_mfma_f32_16x16x4_f32: 4 accumulator registers
// per position), so the output transform is register-local as well.
#include "hwg_common.h"
#include "wino_pack.h"
#include <stdlib.h>

namespace {

struct WinoK {
  const float* x;      // [N,H,W,C]
  const float* u;      // [C/16][16 pos][Kpad][16]   (hwg_wino_pack_weight)
  const float* bias;   // [K] or null
  float* y;            // [N,P,Q,K]
  int N, H, W, C, K, Kpad, P, Q, ph, pw, TP, TQ;
  int M;               // tiles = N*TP*TQ
  int accumulate;
  int nsplit;          // the C/16 chunk loop is cut into nsplit ranges (blockIdx.z); partial outputs go to `part`
  float* part;         // [nsplit][N*P*Q*K]
  int* cnt;            // uniform split: arrival counters (hwg_split_counters), nullptr = the partial images are summed by conv_split_reduce_kernel
  int mt, nt;          // workgroup tiles along M and K
  int xcd_order;       // 1: every XCD owns a contiguous run of the (split, n-tile, m-tile) sequence, 0: plain launch order
  int bal;             // > 0 (64 x 64 DMA kernel only): the tiles from `bal_tile0` on are scheduled BALANCED on `bal` workgroups - their (tile, chunk)
  int bal_tile0;       //   unit sequence in equal runs, launched behind the whole-tile workgroups of the tiles before (a multiple of 256 of them)
  // F(3x3,2x2) launches (wino_conv64d_kernel<.., 3>) and wino_bal_reduce_kernel address the images through strides. N, H, W, C, K, P, Q then describe
  // the VIRTUAL stride-1 two-tap convolution; element (n, h, w, channel chunk t) of its input lies at
  //   x + n x_img + h x_row + w x_pix + (t / x_tc) x_run + (t % x_tc) 16,   output (n, p, q, k) at  y + n y_img + p y_row + q y_pix + (k / y_kc) y_run + k % y_kc
  int mt_edge;         // 2 or 3
  int dbg;             // timing ablations of the ABL build of wino_conv64d_kernel (HWG_CONV_DBG, results are garbage): 1024 no global stores in the
                       // epilogue, 2048 no output transform at all (one dependent store per lane keeps the accumulators alive), 4096 patch loads of chunk 0 only
  int x_img, x_row, x_pix, x_tc, x_run;
  int y_img, y_row, y_pix, y_kc, y_run;
};

// Work item of this workgroup. The dispatcher places block b on XCD b % 8 and every XCD has its own 4 MiB L2: with the plain order all
// eight L2s stream the complete filter image and input (the 512-channel layers of the recogniser measured 10 x their operand bytes
// in L2 misses); giving each XCD a contiguous run of the (split, n-tile, m-tile) sequence leaves it one slice of the filters and,
// when the channel loop is split, only its channel range of the input.
__device__ __forceinline__ bool wino_work_item(const WinoK& a, int& mtile, int& ntile, int& split) {
  const int total = a.mt * a.nt * a.nsplit;
  int w = blockIdx.x;
  if (a.xcd_order) {
    const int xcd = w & 7, slot = w >> 3;
    const int lo = (int)((long long)xcd * total >> 3), hi = (int)((long long)(xcd + 1) * total >> 3);
    w = lo + slot;
    if (w >= hi) return false;
  } else if (w >= total) {
    return false;
  }
  mtile = w % a.mt;
  const int r = w / a.mt;
  ntile = r % a.nt;
  split = r / a.nt;
  return true;
}

// block -> position in the work order of a launch of `total` work items (same XCD-contiguous order as above)
__device__ __forceinline__ bool wino_block(const WinoK& a, int block, int total, int& w) {
  w = block;
  if (a.xcd_order) {
    const int xcd = w & 7, slot = w >> 3;
    const int lo = (int)((long long)xcd * total >> 3), hi = (int)((long long)(xcd + 1) * total >> 3);
    w = lo + slot;
    return w < hi;
  }
  return w < total;
}
// balanced schedule: workgroup g of G owns the units [U g / G, U (g + 1) / G); the owner of unit u is the largest g with U g / G <= u
// (32-bit arithmetic: wino_balance keeps U * G below 2^31)
__host__ __device__ __forceinline__ int wino_bal_owner(int G, int U, int u) { return (int)(((unsigned)(u + 1) * (unsigned)G - 1u) / (unsigned)U); }

__device__ __forceinline__ float quad_partner(float v) {
  // lane b of every quad receives the value of lane {2,2,1,1}[b]  (DPP quad_perm)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x5A, 0xf, 0xf, false));
}

// LDS image of one round (4 positions of the transform domain): planes [4][rows][16 floats]; the four 16-byte chunks of a row are
// XOR-swizzled with (row >> 1) & 3, which makes the ds_read_b128 fragment reads (lane -> row l&15, chunk l>>4) conflict free
// without padding; the plane stride is == 8 (mod 32) dwords so that the 8-lane groups of the ds_write_b128 stores spread over
// all banks.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 16 + 4 * (chunk ^ ((row >> 1) & 3)); }

template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void wino_conv_kernel(WinoK a) {
  constexpr int NT = 64 * WGM * WGN;      // 8 wavefronts (64 x 32 / 32 x 64 / 128 x 16 tiles) or 4 (32 x 32: twice the workgroups for small layers)
  constexpr int TM = 16 * WGM, TN = 16 * WGN;
  constexpr int PSV = TM * 16 + 8, PSU = TN * 16 + 8;
  constexpr int XI = (TM * 16 + NT - 1) / NT;        // x work items per thread: (tile, 4-channel group, patch column)
  constexpr int UI = (TN * 16 + NT - 1) / NT;        // u work items per thread per round: (position, k, 4-channel group)
  constexpr int BUF = 4 * (PSV + PSU);
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  int mtile, ntile, split;
  if (!wino_work_item(a, mtile, ntile, split)) return;
  const int m0 = mtile * TM;
  const int n0 = ntile * TN;
  const int T_all = a.C >> 4;
  const int t0 = (int)((long long)T_all * split / a.nsplit);
  const int t1 = (int)((long long)T_all * (split + 1) / a.nsplit);

  // ---- per-thread gather coordinates of the input patches -------------------------------------------------------------
  int x_off[XI];        // element offset of (n, first patch row, patch column, 4-channel group); rows advance by W*C
  int x_ok[XI];         // bit r: patch row r is inside the image (and the column / tile are valid)
  int x_row[XI], x_c4[XI], x_b[XI];
  bool x_item[XI];
  const int row_stride = a.W * a.C;
#pragma unroll
  for (int it = 0; it < XI; ++it) {
    const int id = tid + it * NT;
    x_item[it] = id < TM * 16;
    const int b = id & 3, c4 = (id >> 2) & 3, row = (id >> 4) % TM;
    x_b[it] = b; x_c4[it] = c4; x_row[it] = row;
    const int m = m0 + row;
    const bool mok = x_item[it] && m < a.M;
    const int mm = mok ? m : 0;
    const int tj = mm % a.TQ;
    const int t2 = mm / a.TQ;
    const int ti = t2 % a.TP;
    const int n = t2 / a.TP;
    const int w = 2 * tj - a.pw + b;
    const bool wok = mok && w >= 0 && w < a.W;
    const int h0 = 2 * ti - a.ph;
    x_off[it] = ((n * a.H + h0) * a.W + (wok ? w : 0)) * a.C + c4 * 4;
    int okm = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (wok && h0 + r >= 0 && h0 + r < a.H) okm |= 1 << r;
    x_ok[it] = okm;
  }
  // filters: item -> (position of the round, k row, 4-channel group); the image of one (chunk, position) is [Kpad][16] contiguous
  int u_off[UI], u_lds[UI];
  bool u_item[UI];
#pragma unroll
  for (int it = 0; it < UI; ++it) {
    const int id = tid + it * NT;
    u_item[it] = id < TN * 16;
    const int c4 = id & 3, k = (id >> 2) % TN, pb = (id >> 2) / TN;
    u_off[it] = (pb * a.Kpad + n0 + k) * 16 + c4 * 4;
    u_lds[it] = pb * PSU + swz(k, c4);
    if (n0 + k >= a.Kpad) u_item[it] = false;
  }

  f32x4 acc[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 raw[XI][4];     // patch column of the chunk being fetched (4 rows)
  float4 tr[XI][4];      // its column transform (B^T along the rows), one float4 per transform row a'
  float4 ur[UI];

  auto load_x = [&](int t) {
#pragma unroll
    for (int it = 0; it < XI; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // masked rows read a valid dummy address (no exec-mask branches between the loads); they are zeroed in col_transform
        const int off = (x_ok[it] >> r) & 1 ? x_off[it] + r * row_stride + t * 16 : 0;
        raw[it][r] = *reinterpret_cast<const float4*>(a.x + off);
      }
  };
  auto col_transform = [&]() {
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      float4 d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r] = raw[it][r];
        if (!((x_ok[it] >> r) & 1)) d[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      tr[it][0] = make_float4(d[0].x - d[2].x, d[0].y - d[2].y, d[0].z - d[2].z, d[0].w - d[2].w);
      tr[it][1] = make_float4(d[1].x + d[2].x, d[1].y + d[2].y, d[1].z + d[2].z, d[1].w + d[2].w);
      tr[it][2] = make_float4(d[2].x - d[1].x, d[2].y - d[1].y, d[2].z - d[1].z, d[2].w - d[1].w);
      tr[it][3] = make_float4(d[1].x - d[3].x, d[1].y - d[3].y, d[1].z - d[3].z, d[1].w - d[3].w);
    }
  };
  auto load_u = [&](int t, int r) {
    const float* base = a.u + ((long long)t * 16 + r * 4) * a.Kpad * 16;
#pragma unroll
    for (int it = 0; it < UI; ++it) ur[it] = *reinterpret_cast<const float4*>(base + (u_item[it] ? u_off[it] : 0));
  };
  // row transform of transform row r across the quad (lane b holds patch column b, produces transform column b) + LDS stores
  auto store_round = [&](int buf, int r) {
    float* Vb = smem + buf * BUF;
    float* Ub = Vb + 4 * PSV;
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      const float4 v = tr[it][r];
      const float sa = (x_b[it] == 3) ? -1.f : 1.f;
      const float sb = (x_b[it] & 1) ? 1.f : -1.f;
      float4 o;
      o.x = sa * v.x + sb * quad_partner(v.x);
      o.y = sa * v.y + sb * quad_partner(v.y);
      o.z = sa * v.z + sb * quad_partner(v.z);
      o.w = sa * v.w + sb * quad_partner(v.w);
      if (x_item[it]) *reinterpret_cast<float4*>(Vb + x_b[it] * PSV + swz(x_row[it], x_c4[it])) = o;
    }
#pragma unroll
    for (int it = 0; it < UI; ++it) {
      float4 v = ur[it];
      if (!u_item[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tid + it * NT < TN * 16) *reinterpret_cast<float4*>(Ub + u_lds[it]) = v;
    }
  };

  const int frow = lane & 15, fchunk = lane >> 4;
  const int a_off = swz(wm * 16 + frow, fchunk);
  const int b_off = swz(wn * 16 + frow, fchunk);

  if (t1 > t0) {
    load_x(t0);
    load_u(t0, 0);
    col_transform();
    store_round(0, 0);
  }
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cur = r & 1;
      const bool more = (r < 3) || (t + 1 < t1);
      if (r == 1 && t + 1 < t1) load_x(t + 1);
      if (more) load_u(r < 3 ? t : t + 1, (r + 1) & 3);
      const float* Vb = smem + cur * BUF;
      const float* Ub = Vb + 4 * PSV;
      float4 af[4], bf[4];
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        af[pb] = *reinterpret_cast<const float4*>(Vb + pb * PSV + a_off);
        bf[pb] = *reinterpret_cast<const float4*>(Ub + pb * PSU + b_off);
      }
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        // element j of both fragments belongs to channel 4*(lane>>4)+j: MFMA j contracts channels {j, 4+j, 8+j, 12+j} of the chunk
        acc[r * 4 + pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pb].x, bf[pb].x, acc[r * 4 + pb], 0, 0, 0);
        acc[r * 4 + pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pb].y, bf[pb].y, acc[r * 4 + pb], 0, 0, 0);
        acc[r * 4 + pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pb].z, bf[pb].z, acc[r * 4 + pb], 0, 0, 0);
        acc[r * 4 + pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[pb].w, bf[pb].w, acc[r * 4 + pb], 0, 0, 0);
      }
      if (more) {
        if (r == 3) col_transform();
        store_round(cur ^ 1, (r + 1) & 3);
      }
      __syncthreads();
    }
  }

  // ---- output transform (register local) and store ----------------------------------------------------------------------
  // C/D layout of the 16x16 MFMA: column (k) = lane & 15, row (tile) = (lane >> 4) * 4 + e
  const int k = n0 + wn * 16 + (lane & 15);
  const bool direct = a.nsplit == 1;
  float* yg = direct ? a.y : a.part + (long long)split * ((long long)a.N * a.P * a.Q * a.K);
  const float bv = (direct && a.bias && k < a.K) ? a.bias[k] : 0.f;
  const bool accum = direct && a.accumulate, coh = !direct && a.cnt;      // coh: the partial image is read by another workgroup of this launch
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + wm * 16 + (lane >> 4) * 4 + e;
    if (m >= a.M || k >= a.K) continue;
    const int tj = m % a.TQ;
    const int t2 = m / a.TQ;
    const int ti = t2 % a.TP;
    const int n = t2 / a.TP;
    float mm[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) mm[p] = acc[p][e];
    float y2[2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {      // columns of the transform domain first: s0 = m0j+m1j+m2j, s1 = m1j-m2j-m3j
      const float s0 = mm[0 * 4 + j] + mm[1 * 4 + j] + mm[2 * 4 + j];
      const float s1 = mm[1 * 4 + j] - mm[2 * 4 + j] - mm[3 * 4 + j];
      mm[0 * 4 + j] = s0;
      mm[1 * 4 + j] = s1;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      y2[i][0] = mm[i * 4 + 0] + mm[i * 4 + 1] + mm[i * 4 + 2];
      y2[i][1] = mm[i * 4 + 1] - mm[i * 4 + 2] - mm[i * 4 + 3];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = 2 * ti + i;
      if (p >= a.P) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = 2 * tj + j;
        if (q >= a.Q) continue;
        const long long o = (((long long)n * a.P + p) * a.Q + q) * a.K + k;
        float v = y2[i][j] + bv;
        if (accum) v += yg[o];
        if (coh) hwg_store_agent(yg + o, v); else yg[o] = v;
      }
    }
  }
  if (direct || !a.cnt) return;
  // channel split without a reduce launch: the wavefront that delivers its 16 x 16 block's last partial image sums them (hwg_split_arrive_wave)
  if (!hwg_split_arrive_wave(a.cnt + ((size_t)ntile * a.mt + mtile) * (WGM * WGN) + wid, a.nsplit)) return;
  const long long total = (long long)a.N * a.P * a.Q * a.K;
  const float bl = (a.bias && k < a.K) ? a.bias[k] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + wm * 16 + (lane >> 4) * 4 + e;
    if (m >= a.M || k >= a.K) continue;
    const int tj = m % a.TQ;
    const int t2 = m / a.TQ;
    const int ti = t2 % a.TP;
    const int n = t2 / a.TP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = 2 * ti + i;
      if (p >= a.P) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = 2 * tj + j;
        if (q >= a.Q) continue;
        const long long o = (((long long)n * a.P + p) * a.Q + q) * a.K + k;
        float v = hwg_load_agent(a.part + o);
        for (int sp = 1; sp < a.nsplit; ++sp) v += hwg_load_agent(a.part + sp * total + o);
        if (a.bias) v += bl;
        if (a.accumulate) v += a.y[o];
        a.y[o] = v;
      }
    }
  }
}

// 64 tile x 64 channel workgroup tile with register-level operand reuse. Same producer side as wino_conv_kernel (round = one transform
// row a of a 16-channel chunk: positions (a, b = 0..3)), but wavefront (b, nh) keeps ONE position column b for all four rows a as a
// 64 tile x 32 channel block (4 x 2 MFMA blocks x 4 rows = 128 accumulator registers): 6 ds_read_b128 feed 32 MFMAs (the 16 x 16 wave
// tiles above read 8 fragments per 16 MFMAs and spend ~70 % of the LDS cycles), and a round carries twice the MFMA work per barrier.
// The output transform Y = A^T M A is register-local over a; the sum over b crosses wavefronts through LDS once per workgroup
// ([b][i][tile][channel] exchange image, 136 KB, aliases the operand buffers).
__global__ __launch_bounds__(512) void wino_conv64_kernel(WinoK a) {
  constexpr int NT = 512, TM = 64, TN = 64;
  constexpr int PSV = TM * 16 + 8, PSU = TN * 16 + 8;
  constexpr int XI = TM * 16 / NT;                   // 2 x work items per thread: (tile, 4-channel group, patch column)
  constexpr int UI = TN * 16 / NT;                   // 2 u work items per thread per round: (position, k, 4-channel group)
  constexpr int BUF = 4 * (PSV + PSU);
  constexpr int LDK = 68;                            // exchange row stride (floats): the 4 row groups of a C/D block land 16 banks apart
  constexpr int XCH = 8 * 64 * LDK;
  __shared__ __attribute__((aligned(16))) float smem[(2 * BUF > XCH) ? 2 * BUF : XCH];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int bq = wid & 3, nh = wid >> 2;
  int mtile, ntile, split;
  if (!wino_work_item(a, mtile, ntile, split)) return;
  const int m0 = mtile * TM;
  const int n0 = ntile * TN;
  const int T_all = a.C >> 4;
  const int t0 = (int)((long long)T_all * split / a.nsplit);
  const int t1 = (int)((long long)T_all * (split + 1) / a.nsplit);

  int x_off[XI], x_ok[XI], x_row[XI], x_c4[XI], x_b[XI];
  const int row_stride = a.W * a.C;
#pragma unroll
  for (int it = 0; it < XI; ++it) {
    const int id = tid + it * NT;
    const int b = id & 3, c4 = (id >> 2) & 3, row = id >> 4;
    x_b[it] = b; x_c4[it] = c4; x_row[it] = row;
    const int m = m0 + row;
    const bool mok = m < a.M;
    const int mm = mok ? m : 0;
    const int tj = mm % a.TQ;
    const int t2 = mm / a.TQ;
    const int ti = t2 % a.TP;
    const int n = t2 / a.TP;
    const int w = 2 * tj - a.pw + b;
    const bool wok = mok && w >= 0 && w < a.W;
    const int h0 = 2 * ti - a.ph;
    x_off[it] = ((n * a.H + h0) * a.W + (wok ? w : 0)) * a.C + c4 * 4;
    int okm = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (wok && h0 + r >= 0 && h0 + r < a.H) okm |= 1 << r;
    x_ok[it] = okm;
  }
  int u_off[UI], u_lds[UI];
  bool u_item[UI];
#pragma unroll
  for (int it = 0; it < UI; ++it) {
    const int id = tid + it * NT;
    const int c4 = id & 3, k = (id >> 2) % TN, pb = (id >> 2) / TN;
    u_off[it] = (pb * a.Kpad + n0 + k) * 16 + c4 * 4;
    u_lds[it] = pb * PSU + swz(k, c4);
    u_item[it] = n0 + k < a.Kpad;
  }

  f32x4 acc[4][4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[r][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 raw[XI][4], tr[XI][4], ur[UI];
  auto load_x = [&](int t) {
#pragma unroll
    for (int it = 0; it < XI; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = (x_ok[it] >> r) & 1 ? x_off[it] + r * row_stride + t * 16 : 0;
        raw[it][r] = *reinterpret_cast<const float4*>(a.x + off);
      }
  };
  auto col_transform = [&]() {
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      float4 d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r] = raw[it][r];
        if (!((x_ok[it] >> r) & 1)) d[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      tr[it][0] = make_float4(d[0].x - d[2].x, d[0].y - d[2].y, d[0].z - d[2].z, d[0].w - d[2].w);
      tr[it][1] = make_float4(d[1].x + d[2].x, d[1].y + d[2].y, d[1].z + d[2].z, d[1].w + d[2].w);
      tr[it][2] = make_float4(d[2].x - d[1].x, d[2].y - d[1].y, d[2].z - d[1].z, d[2].w - d[1].w);
      tr[it][3] = make_float4(d[1].x - d[3].x, d[1].y - d[3].y, d[1].z - d[3].z, d[1].w - d[3].w);
    }
  };
  auto load_u = [&](int t, int r) {
    const float* base = a.u + ((long long)t * 16 + r * 4) * a.Kpad * 16;
#pragma unroll
    for (int it = 0; it < UI; ++it) ur[it] = *reinterpret_cast<const float4*>(base + (u_item[it] ? u_off[it] : 0));
  };
  auto store_round = [&](int buf, int r) {
    float* Vb = smem + buf * BUF;
    float* Ub = Vb + 4 * PSV;
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      const float4 v = tr[it][r];
      const float sa = (x_b[it] == 3) ? -1.f : 1.f;
      const float sb = (x_b[it] & 1) ? 1.f : -1.f;
      float4 o;
      o.x = sa * v.x + sb * quad_partner(v.x);
      o.y = sa * v.y + sb * quad_partner(v.y);
      o.z = sa * v.z + sb * quad_partner(v.z);
      o.w = sa * v.w + sb * quad_partner(v.w);
      *reinterpret_cast<float4*>(Vb + x_b[it] * PSV + swz(x_row[it], x_c4[it])) = o;
    }
#pragma unroll
    for (int it = 0; it < UI; ++it) {
      float4 v = ur[it];
      if (!u_item[it]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(Ub + u_lds[it]) = v;
    }
  };

  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], b_off[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) a_off[m] = bq * PSV + swz(m * 16 + frow, fchunk);
#pragma unroll
  for (int n = 0; n < 2; ++n) b_off[n] = 4 * PSV + bq * PSU + swz((nh * 2 + n) * 16 + frow, fchunk);

  if (t1 > t0) {
    load_x(t0);
    load_u(t0, 0);
    col_transform();
    store_round(0, 0);
  }
  __syncthreads();

  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cur = r & 1;
      const bool more = (r < 3) || (t + 1 < t1);
      if (r == 1 && t + 1 < t1) load_x(t + 1);
      if (more) load_u(r < 3 ? t : t + 1, (r + 1) & 3);
      const float* Sb = smem + cur * BUF;
      float4 af[4], bf[2];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const float4*>(Sb + a_off[m]);
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const float4*>(Sb + b_off[n]);
      // element j of both fragments belongs to channel 4*(lane>>4)+j: MFMA j contracts channels {j, 4+j, 8+j, 12+j} of the chunk
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, bf[n].x, acc[r][m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, bf[n].y, acc[r][m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, bf[n].z, acc[r][m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, bf[n].w, acc[r][m][n], 0, 0, 0);
      if (more) {
        if (r == 3) col_transform();
        store_round(cur ^ 1, (r + 1) & 3);
      }
      __syncthreads();
    }
  }

  // ---- output transform: over the rows a in registers (s0 = m0+m1+m2, s1 = m1-m2-m3), over the columns b through LDS ----------------
  // exchange image X[b][i][tile][channel]; C/D layout of a block: channel = lane & 15, tile = (lane >> 4) * 4 + e
  float* X = smem;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = acc[0][m][n][e] + acc[1][m][n][e] + acc[2][m][n][e];
        const float s1 = acc[1][m][n][e] - acc[2][m][n][e] - acc[3][m][n][e];
        const int tile = m * 16 + (lane >> 4) * 4 + e, kk = (nh * 2 + n) * 16 + (lane & 15);
        X[((bq * 2 + 0) * 64 + tile) * LDK + kk] = s0;
        X[((bq * 2 + 1) * 64 + tile) * LDK + kk] = s1;
      }
  __syncthreads();
  const int kk = tid & 63;
  const int k = n0 + kk;
  const bool direct = a.nsplit == 1;
  float* yg = direct ? a.y : a.part + (long long)split * ((long long)a.N * a.P * a.Q * a.K);
  const float bv = (direct && a.bias && k < a.K) ? a.bias[k] : 0.f;
  const bool accum = direct && a.accumulate;
#pragma unroll 2
  for (int tile = tid >> 6; tile < 64; tile += 8) {
    const int m = m0 + tile;
    if (m >= a.M || k >= a.K) continue;
    const int tj = m % a.TQ;
    const int t2 = m / a.TQ;
    const int ti = t2 % a.TP;
    const int n = t2 / a.TP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float r0 = X[((0 * 2 + i) * 64 + tile) * LDK + kk], r1 = X[((1 * 2 + i) * 64 + tile) * LDK + kk];
      const float r2 = X[((2 * 2 + i) * 64 + tile) * LDK + kk], r3 = X[((3 * 2 + i) * 64 + tile) * LDK + kk];
      const int p = 2 * ti + i;
      if (p >= a.P) continue;
      const float y0 = r0 + r1 + r2, y1 = r1 - r2 - r3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int q = 2 * tj + j;
        if (q >= a.Q) continue;
        const long long o = (((long long)n * a.P + p) * a.Q + q) * a.K + k;
        float v = (j == 0 ? y0 : y1) + bv;
        if (accum) v += yg[o];
        yg[o] = v;
      }
    }
  }
}

// Wave-specialised variant for layers with >= 32 output channels: wavefronts 0-3 ("consumers", one per SIMD) issue nothing but fragment
// reads and MFMAs on a 32 tile x 16 channel x 16 position block each (128 accumulator registers), wavefronts 4-7 ("producers", the
// second wave of every SIMD) feed LDS ahead of them:
//   * input patches: global -> registers -> column transform -> (DPP row transform) -> LDS, two rounds ahead (double buffer);
//   * filters: already in the transform domain, so they go global -> LDS directly (global_load_lds_dwordx4, no registers, no VALU,
//     no ds_write), three rounds ahead into a ring of three buffers; the XOR swizzle of the LDS image is applied on the global
//     side (each lane picks the 16-byte piece that belongs into its linear LDS slot).
// The consumers read the fragments of round s+1 into registers while the matrix cores work on round s, so no LDS latency is exposed
// behind a barrier; the VALU / LDS-store / global-load work of the producers runs beside the MFMA pipe of the same SIMD.
__device__ __forceinline__ void wait_vmcnt(int n) {
  // s_waitcnt vmcnt(n) only (expcnt / lgkmcnt fields left at "no wait"); n is wave-uniform, the immediate needs a constant
  switch (n) {
#define HWG_VM(N) case N: __builtin_amdgcn_s_waitcnt(((N) & 15) | (7 << 4) | (15 << 8) | (((N) >> 4) << 14)); break;
    HWG_VM(1) HWG_VM(2) HWG_VM(3) HWG_VM(4) HWG_VM(6) HWG_VM(8) HWG_VM(10) HWG_VM(12) HWG_VM(16) HWG_VM(20) HWG_VM(24) HWG_VM(28)
    HWG_VM(32) HWG_VM(40) HWG_VM(48)
#undef HWG_VM
    default: __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8)); break;   // vmcnt(0)
  }
}

constexpr int WINO_DA = 6;      // filter rounds in flight / resident ahead of the consumers (ring depth)

template <int CM, int CN>
__global__ __launch_bounds__(512) void wino_conv_ws_kernel(WinoK a) {
  static_assert(CM * CN == 4, "4 consumer wavefronts");
  constexpr int NX = 128;                             // threads of the two patch-producer waves
  constexpr int TM = 32 * CM, TN = 16 * CN;
  constexpr int PSV = TM * 16 + 8, PSU = TN * 16 + 8;
  constexpr int VBUF = 4 * PSV, UBUF = 4 * PSU;
  constexpr int DA = WINO_DA;
  constexpr int XI = TM * 16 / NX;                    // (tile, 4-channel group, patch column) items per patch-producer thread and chunk
  constexpr int UD = TN / 16;                         // LDS-DMA instructions per filter plane (TN rows of 64 B, 1 KiB per instruction)
  __shared__ __attribute__((aligned(16))) float smem[2 * VBUF + DA * UBUF];
  float* const Vs = smem;
  float* const Us = smem + 2 * VBUF;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int mtile, ntile, split;
  if (!wino_work_item(a, mtile, ntile, split)) return;
  const int m0 = mtile * TM;
  const int n0 = ntile * TN;
  const int T_all = a.C >> 4;
  const int t0 = (int)((long long)T_all * split / a.nsplit);
  const int t1 = (int)((long long)T_all * (split + 1) / a.nsplit);
  const int S = 4 * (t1 - t0);                        // rounds: (channel chunk, row of the transform domain)
  const int R0 = 4 * t0;                              // absolute index of this workgroup's first round

  if (wid >= 6) {
    // =========================== filter movers (wavefronts 6, 7) ===========================
    // Every round's filters (4 positions x TN channels x 16 input channels) go global -> LDS by DMA, DA rounds ahead, into a ring of DA
    // buffers. Wave 6 moves planes 0, 1, wave 7 planes 2, 3. The only VMEM traffic of these waves are the DMAs, so "round q has
    // landed" is exactly "at most (groups issued after q) * 2 UD operations outstanding".
    const int pw = (wid - 6) * 2;
    int u_src[2][UD];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i < UD; ++i) {
        // lane -> (row, LDS slot); the slot holds channel group slot ^ ((row >> 1) & 3) (the swizzle, applied on the global side);
        // rows past the padded channel count re-read the last row (their outputs are never stored)
        const int row = 16 * i + (lane >> 2), slot = lane & 3;
        int k = n0 + row;
        if (k >= a.Kpad) k = a.Kpad - 1;
        u_src[pl][i] = ((pw + pl) * a.Kpad + k) * 16 + 4 * (slot ^ ((row >> 1) & 3));
      }
    const long long u_round = 4LL * a.Kpad * 16;      // floats per round in the filter image
    auto dma_u = [&](int ring, int round) {           // round absolute
      const float* base = a.u + (long long)round * u_round;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        float* dst = Us + ring * UBUF + (pw + pl) * PSU;
#pragma unroll
        for (int i = 0; i < UD; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + u_src[pl][i]),
                                           (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
      }
    };
    int issued = 0;                                   // rounds whose DMA has been issued
    for (; issued < DA && issued < S; ++issued) dma_u(issued, R0 + issued);
    wait_vmcnt((issued - 1) * 2 * UD);                // round 0
    __syncthreads();
    wait_vmcnt(issued > 1 ? (issued - 2) * 2 * UD : 0);   // round 1
    __syncthreads();
    int ring = 0;                                     // slot of round s + DA == slot of round s (left by the consumers at step s - 1)
    for (int s = 0; s < S; ++s) {
      if (issued < S) { dma_u(ring, R0 + issued); ++issued; }
      ring = ring == DA - 1 ? 0 : ring + 1;
      if (s + 2 < S) wait_vmcnt((issued - (s + 3)) * 2 * UD);   // round s + 2 has landed before the consumers are let at it
      __syncthreads();
    }
    return;
  }
  if (wid >= 4) {
    // =========================== patch producers (wavefronts 4, 5) ===========================
    const int pt = tid - 256;
    int x_off[XI], x_ok[XI], x_lds[XI];
    float x_sa[XI], x_sb[XI];
    const int row_stride = a.W * a.C;
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      const int id = pt + it * NX;
      const int b = id & 3, c4 = (id >> 2) & 3, row = id >> 4;
      x_lds[it] = b * PSV + swz(row, c4);
      x_sa[it] = (b == 3) ? -1.f : 1.f;
      x_sb[it] = (b & 1) ? 1.f : -1.f;
      const int m = m0 + row;
      const bool mok = m < a.M;
      const int mm = mok ? m : 0;
      const int tj = mm % a.TQ;
      const int t2 = mm / a.TQ;
      const int ti = t2 % a.TP;
      const int n = t2 / a.TP;
      const int w = 2 * tj - a.pw + b;
      const bool wok = mok && w >= 0 && w < a.W;
      const int h0 = 2 * ti - a.ph;
      x_off[it] = ((n * a.H + h0) * a.W + (wok ? w : 0)) * a.C + c4 * 4;
      int okm = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (wok && h0 + r >= 0 && h0 + r < a.H) okm |= 1 << r;
      x_ok[it] = okm;
    }
    float4 raw[XI][4], tr[XI][4];
    auto load_x = [&](int t) {
#pragma unroll
      for (int it = 0; it < XI; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int off = (x_ok[it] >> r) & 1 ? x_off[it] + r * row_stride + t * 16 : 0;
          raw[it][r] = *reinterpret_cast<const float4*>(a.x + off);
        }
    };
    auto col_transform = [&]() {
#pragma unroll
      for (int it = 0; it < XI; ++it) {
        float4 d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[r] = raw[it][r];
          if (!((x_ok[it] >> r) & 1)) d[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        tr[it][0] = make_float4(d[0].x - d[2].x, d[0].y - d[2].y, d[0].z - d[2].z, d[0].w - d[2].w);
        tr[it][1] = make_float4(d[1].x + d[2].x, d[1].y + d[2].y, d[1].z + d[2].z, d[1].w + d[2].w);
        tr[it][2] = make_float4(d[2].x - d[1].x, d[2].y - d[1].y, d[2].z - d[1].z, d[2].w - d[1].w);
        tr[it][3] = make_float4(d[1].x - d[3].x, d[1].y - d[3].y, d[1].z - d[3].z, d[1].w - d[3].w);
      }
    };
    auto write_v = [&](int buf, int r) {
      float* B = Vs + buf * VBUF;
#pragma unroll
      for (int it = 0; it < XI; ++it) {
        const float4 v = tr[it][r];
        float4 o;
        o.x = x_sa[it] * v.x + x_sb[it] * quad_partner(v.x);
        o.y = x_sa[it] * v.y + x_sb[it] * quad_partner(v.y);
        o.z = x_sa[it] * v.z + x_sb[it] * quad_partner(v.z);
        o.w = x_sa[it] * v.w + x_sb[it] * quad_partner(v.w);
        *reinterpret_cast<float4*>(B + x_lds[it]) = o;
      }
    };
    // patches of chunk c are fetched a whole chunk (4 rounds) before their column transform
    if (S > 0) {
      load_x(t0);
      col_transform();
      if (t0 + 1 < t1) load_x(t0 + 1);
      write_v(0, 0);
    }
    __syncthreads();
    if (S > 0) write_v(1, 1);
    __syncthreads();
    for (int t = t0; t < t1; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // step s = 4 (t - t0) + r: the consumers multiply round s and read the fragments of round s + 1; this side fills round s + 2
        const int s = 4 * (t - t0) + r;
        if (s + 2 < S) {
          if (r == 2) {
            col_transform();                          // chunk t + 1 (fetched during chunk t - 1 / the prologue)
            if (t + 2 < t1) load_x(t + 2);
          }
          write_v(r & 1, (r + 2) & 3);
        }
        __syncthreads();
      }
    }
    return;
  }

  // =========================== consumers ===========================
  __builtin_amdgcn_s_setprio(1);
  const int cm = wid / CN, cn = wid % CN;
  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[2];
  a_off[0] = swz(cm * 32 + frow, fchunk);
  a_off[1] = swz(cm * 32 + 16 + frow, fchunk);
  const int b_off = swz(cn * 16 + frow, fchunk);

  f32x4 acc[16][2];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    acc[p][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[p][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float4 fa[2][4][2], fb[2][4];
  auto read_frags = [&](int set, int vbuf, int uring) {
    const float* A = Vs + vbuf * VBUF;
    const float* B = Us + uring * UBUF + b_off;
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      fa[set][pb][0] = *reinterpret_cast<const float4*>(A + pb * PSV + a_off[0]);
      fa[set][pb][1] = *reinterpret_cast<const float4*>(A + pb * PSV + a_off[1]);
      fb[set][pb] = *reinterpret_cast<const float4*>(B + pb * PSU);
    }
  };
  __syncthreads();
  if (S > 0) read_frags(0, 0, 0);
  __syncthreads();
  int ring = 1;                                       // ring slot of round s + 1
  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = 4 * (t - t0) + r;
      const int cur = r & 1;
      if (s + 1 < S) read_frags(cur ^ 1, cur ^ 1, ring);
      ring = ring == WINO_DA - 1 ? 0 : ring + 1;
      // element j of the fragments is channel 4*(lane>>4)+j of the chunk; j outermost: 8 independent accumulators between two
      // MFMAs on the same one (dependent latency of v_mfma_f32_16x16x4_f32 is 40 cycles, issue interval 32)
#define HWG_WINO_MFMA(J)                                                                                                     \
      _Pragma("unroll") for (int pb = 0; pb < 4; ++pb) {                                                                     \
        acc[r * 4 + pb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][pb][0].J, fb[cur][pb].J, acc[r * 4 + pb][0], 0, 0, 0); \
        acc[r * 4 + pb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][pb][1].J, fb[cur][pb].J, acc[r * 4 + pb][1], 0, 0, 0); \
      }
      HWG_WINO_MFMA(x) HWG_WINO_MFMA(y) HWG_WINO_MFMA(z) HWG_WINO_MFMA(w)
#undef HWG_WINO_MFMA
      // keep the barrier BEHIND the MFMAs: hoisted in front of them (legal, they touch no LDS) it would stall the matrix pipe on the
      // fragment reads of the next round and on the producers every round
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  }
  __builtin_amdgcn_s_setprio(0);

  // output transform (register local) and store: column (k) = lane & 15, row (tile) = (lane >> 4) * 4 + e of each 16x16 block
  const int k = n0 + cn * 16 + (lane & 15);
  const bool direct = a.nsplit == 1;
  float* yg = direct ? a.y : a.part + (long long)split * ((long long)a.N * a.P * a.Q * a.K);
  const float bv = (direct && a.bias && k < a.K) ? a.bias[k] : 0.f;
  const bool accum = direct && a.accumulate;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + cm * 32 + blk * 16 + (lane >> 4) * 4 + e;
      if (m >= a.M || k >= a.K) continue;
      const int tj = m % a.TQ;
      const int t2 = m / a.TQ;
      const int ti = t2 % a.TP;
      const int n = t2 / a.TP;
      float mm[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) mm[p] = acc[p][blk][e];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s0 = mm[0 * 4 + j] + mm[1 * 4 + j] + mm[2 * 4 + j];
        const float s1 = mm[1 * 4 + j] - mm[2 * 4 + j] - mm[3 * 4 + j];
        mm[0 * 4 + j] = s0;
        mm[1 * 4 + j] = s1;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = 2 * ti + i;
        if (p >= a.P) continue;
        const float y0 = mm[i * 4 + 0] + mm[i * 4 + 1] + mm[i * 4 + 2];
        const float y1 = mm[i * 4 + 1] - mm[i * 4 + 2] - mm[i * 4 + 3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = 2 * tj + j;
          if (q >= a.Q) continue;
          const long long o = (((long long)n * a.P + p) * a.Q + q) * a.K + k;
          float v = (j == 0 ? y0 : y1) + bv;
          if (accum) v += yg[o];
          yg[o] = v;
        }
      }
    }
}

// 64 tile x 64 channel workgroup tile on 12 wavefronts: 8 consumers (two per SIMD, 32 tiles x 16 channels x 16 positions = 128
// accumulator registers each) and 4 producers (patch transform + filter DMA, one filter plane each). Twice the MFMA work per byte
// that enters the CU compared with the 32 x 64 kernel above - the transform-domain operands are 16/9 larger than the direct ones and
// a Winograd tile is only 4 pixels, so operand traffic (L2 -> LDS), not the matrix pipe, bounds the smaller tiles. The register file is
// the limit: 12 waves x 168 registers is all of it, hence the fragments are read one position ahead (24 registers) instead of one
// round ahead; the second consumer of each SIMD covers the LDS latency behind the barrier.
#ifndef HWG_W64D_SCHED
#define HWG_W64D_SCHED 6
#endif
template <int ABL, bool BAL, int MT = 2>
__global__ __launch_bounds__(512) void wino_conv64d_kernel(WinoK a) {
  constexpr int NT = 512, TM = 64, TN = 64;
  constexpr int PSV = TM * 16 + 8, PSU = TN * 16 + 8;
  constexpr int XI = TM * 16 / NT;                   // 2 x work items per thread: (tile, 4-channel group, patch column)
  constexpr int VBUF = 4 * PSV, UBUF = 4 * PSU;      // two patch buffers; ring of four filter rounds = the four rounds of a chunk, each its own array:
  // the compiler tracks pending LDS DMA per underlying object, so reading the array of round r does not force a wait for the DMA that is
  // filling the array of round r + 2 (one shared array costs a vmcnt(0) before every fragment read)
  constexpr int LDK = 68;                            // exchange row stride (floats): the 4 row groups of a C/D block land 16 banks apart
  constexpr int XP = 64 * LDK;                       // one (b, i) plane of the exchange image
  static_assert(XP >= UBUF && 4 * XP >= 2 * VBUF, "exchange planes alias the operand buffers");
  __shared__ __attribute__((aligned(16))) float smem[4 * XP];
  __shared__ __attribute__((aligned(16))) float U0[XP], U1[XP], U2[XP], U3[XP];
  // output coordinates of the workgroup's 64 tiles, decoded ONCE (by 64 lanes) instead of by every lane for each of its 8 tiles: the three
  // integer divisions per tile (~120 VALU instructions) made the output loop 5.5 us of a 31 us workgroup on 4-chunk layers
  // (tools/probes/probe_r5_wino_fixed.txt: HWG_CONV_DBG 512 vs 2560)
  __shared__ int tile_out[64], tile_ext[64], tile_w0[64];
  float* const Vs = smem;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int bq = wid & 3, nh = wid >> 2;
  const int T_all = a.C >> 4;
  // Work of this workgroup: one (tile, chunk range) item of the uniform split - or, in a balanced launch (a.bal > 0), either one whole tile
  // of the leading region or its G-th of the (tile, chunk) unit sequence of the tail region, which may end one tile and begin the next:
  // every tile it touches is one SEGMENT of the loop below. A tile cut by workgroup boundaries is written as numbered pieces into `part`
  // (piece = workgroup - first workgroup of the tile) and summed by wino_bal_reduce_kernel; a tile inside one workgroup is written directly.
  int mtile, ntile, split = 0, wg = 0;
  int unit = 0, unit_end = 0;               // units of the balanced region, relative to its first tile (32-bit: the planner keeps units * G < 2^31)
  const int U_bal = (a.mt * a.nt - a.bal_tile0) * T_all;
  bool whole = false;                        // balanced launch, whole-tile workgroup of the leading region
  if constexpr (BAL) {
    if ((int)blockIdx.x < a.bal_tile0) {
      if (!wino_block(a, blockIdx.x, a.bal_tile0, wg)) return;
      whole = true;
      mtile = wg % a.mt;
      ntile = wg / a.mt;
    } else {
      if (!wino_block(a, blockIdx.x - a.bal_tile0, a.bal, wg)) return;
      unit = (int)((long long)U_bal * wg / a.bal);
      unit_end = (int)((long long)U_bal * (wg + 1) / a.bal);
      if (unit >= unit_end) return;
    }
  } else if (!wino_work_item(a, mtile, ntile, split)) {
    return;
  }
  for (;;) {
  int t0, t1;
  if (BAL && !whole) {
    const int rel = unit / T_all, tile = a.bal_tile0 + rel;
    t0 = unit - rel * T_all;
    t1 = unit_end - unit < T_all - t0 ? t0 + (unit_end - unit) : T_all;
    mtile = tile % a.mt;
    ntile = tile / a.mt;
  } else {
    t0 = (int)((long long)T_all * split / a.nsplit);
    t1 = (int)((long long)T_all * (split + 1) / a.nsplit);
  }
  const int m0 = mtile * TM;
  const int n0 = ntile * TN;

  // MT = output tile edge: 2 = F(2x2,3x3) on a plain NHWC image; 3 = F(3x3,2x2) (two-tap layers: the 4x4 stride-2 convolutions and their data
  // gradients as stride-1 2x2 convolutions on the space-to-depth image). Same 4x4 patches, same input transform (the sign of the fourth
  // position is folded into the packed filters), hence the same main loop; the image is addressed through the strides of WinoK (rows of the
  // virtual image may be row PAIRS of the real one, its channels split in runs that lie a real row apart).
  int x_off[XI], x_ok[XI], x_row[XI], x_c4[XI], x_b[XI];
  const int row_stride = MT == 2 ? a.W * a.C : a.x_row;
  // patch origins of the 64 tiles, decoded once per workgroup (three integer divisions per tile) and shared through LDS
  if (tid < 64) {
    const int m = m0 + tid;
    const bool mok = m < a.M;
    const int mm = mok ? m : 0;
    const int tj = mm % a.TQ;
    const int t2 = mm / a.TQ;
    const int ti = t2 % a.TP;
    tile_out[tid] = mok ? t2 / a.TP : -1;       // sample, -1: no such tile
    tile_ext[tid] = MT * ti - a.ph;
    tile_w0[tid] = MT * tj - a.pw;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < XI; ++it) {
    const int id = tid + it * NT;
    const int b = id & 3, c4 = (id >> 2) & 3, row = id >> 4;
    x_b[it] = b; x_c4[it] = c4; x_row[it] = row;
    const int n = max(tile_out[row], 0);
    const bool mok = tile_out[row] >= 0;
    const int w = tile_w0[row] + b;
    const bool wok = mok && w >= 0 && w < a.W;
    const int h0 = tile_ext[row];
    if constexpr (MT == 2) x_off[it] = ((n * a.H + h0) * a.W + (wok ? w : 0)) * a.C + c4 * 4;
    else x_off[it] = n * a.x_img + h0 * a.x_row + (wok ? w : 0) * a.x_pix + c4 * 4;
    int okm = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (wok && h0 + r >= 0 && h0 + r < a.H) okm |= 1 << r;
    x_ok[it] = okm;
  }
  // Filters: every round's 4 positions x 64 channels x 16 input channels (16 KB, already in the transform domain) go global -> LDS by DMA
  // (global_load_lds_dwordx4: no registers, no VALU, no ds_write), TWO rounds ahead into a ring of three buffers - the L2 latency of
  // the filter stream used to sit on every round's critical path (loads issued at the top of a round, stored at its end). A wave
  // instruction fills 16 rows x 64 B linearly; the XOR swizzle of the image is applied on the global side (each lane fetches the 16-byte
  // piece that belongs into its linear slot). Wavefront w moves instructions 2w, 2w+1 of the round's 16.
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  int u_src[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int inst = 2 * widu + j, plane = inst >> 2, i = inst & 3;
    const int row = 16 * i + (lane >> 2), slot = lane & 3;
    int kk = n0 + row;
    if (kk >= a.Kpad) kk = a.Kpad - 1;              // rows past the padded channel count re-read the last row (never stored)
    u_src[j] = (plane * a.Kpad + kk) * 16 + 4 * (slot ^ ((row >> 1) & 3));
  }
  const long long u_round = 4LL * a.Kpad * 16;        // floats per round in the filter image
  const int R_all = 4 * (t1 - t0);                    // rounds of this workgroup
  auto dma_u = [&](int R, float* dst) {              // round index relative to t0; rounds past the end re-fetch the last one (unused)
    const int Rc = R < R_all ? R : R_all - 1;
    const float* base = a.u + ((long long)t0 * 4 + Rc) * u_round;
    // issued as inline assembly: for a DMA it can see, the compiler makes EVERY later LDS access (and the barrier's fence) wait for all pending
    // DMA (vmcnt(0)) - here two rounds are meant to stay in flight; the waits that matter are placed by hand before the round barriers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int inst = 2 * widu + j;
      const float* src = base + u_src[j];
      const unsigned ldst = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(dst + (inst >> 2) * PSU + (inst & 3) * 256);
      asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(ldst) : "memory");   // (m0 is written; clang reserves it and rejects it as a clobber - nothing else in this kernel uses it)
    }
  };

  f32x4 acc[4][4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[r][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 raw[XI][4], tr[XI][4];
  auto load_x = [&](int t) {
#pragma unroll
    for (int it = 0; it < XI; ++it)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int off = (x_ok[it] >> r) & 1 ? x_off[it] + r * row_stride + (MT == 2 ? t * 16 : (t / a.x_tc) * a.x_run + (t % a.x_tc) * 16) : 0;
        raw[it][r] = *reinterpret_cast<const float4*>(a.x + off);
      }
  };
  auto col_transform = [&]() {
    if constexpr (ABL & 1) {          // timing ablation (HWG_CONV_DBG=512): no patch transform - what a kernel fed with pre-transformed patches would save
#pragma unroll
      for (int it = 0; it < XI; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[it][r] = raw[it][r];
      return;
    }
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      float4 d[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r] = raw[it][r];
        if (!((x_ok[it] >> r) & 1)) d[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      tr[it][0] = make_float4(d[0].x - d[2].x, d[0].y - d[2].y, d[0].z - d[2].z, d[0].w - d[2].w);
      tr[it][1] = make_float4(d[1].x + d[2].x, d[1].y + d[2].y, d[1].z + d[2].z, d[1].w + d[2].w);
      tr[it][2] = make_float4(d[2].x - d[1].x, d[2].y - d[1].y, d[2].z - d[1].z, d[2].w - d[1].w);
      tr[it][3] = make_float4(d[1].x - d[3].x, d[1].y - d[3].y, d[1].z - d[3].z, d[1].w - d[3].w);
    }
  };
  auto store_round = [&](int buf, int r) {
    float* Vb = Vs + buf * VBUF;
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      const float4 v = tr[it][r];
      const float sa = (x_b[it] == 3) ? -1.f : 1.f;
      const float sb = (x_b[it] & 1) ? 1.f : -1.f;
      float4 o;
      if constexpr (ABL & 1) {
        o = v;
      } else {
        o.x = sa * v.x + sb * quad_partner(v.x);
        o.y = sa * v.y + sb * quad_partner(v.y);
        o.z = sa * v.z + sb * quad_partner(v.z);
        o.w = sa * v.w + sb * quad_partner(v.w);
      }
      *reinterpret_cast<float4*>(Vb + x_b[it] * PSV + swz(x_row[it], x_c4[it])) = o;
    }
  };

  const int frow = lane & 15, fchunk = lane >> 4;
  int a_off[4], b_off[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) a_off[m] = bq * PSV + swz(m * 16 + frow, fchunk);
#pragma unroll
  for (int n = 0; n < 2; ++n) b_off[n] = bq * PSU + swz((nh * 2 + n) * 16 + frow, fchunk);

  if (t1 > t0) {
    dma_u(0, U0);
    dma_u(1, U1);
    load_x(t0);
    col_transform();
    store_round(0, 0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt((2 & 15) | (7 << 4) | (0 << 8));       // vmcnt(2): round 0's filters have landed, round 1's may be in flight
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int R = 4 * (t - t0) + r;
      const int cur = r & 1;
      // VMEM order of a round: filter DMA of round R + 2 (2 operations), then - in rounds r == 1 - the 8 patch loads of the next chunk
      // (the last chunk re-fetches itself, so the counts below are the same in every round)
      if (r == 3) { col_transform(); __builtin_amdgcn_sched_barrier(0); }   // needs the patch loads of round 1: before this round's DMA joins the queue behind them
      dma_u(R + 2, r == 0 ? U2 : r == 1 ? U3 : r == 2 ? U0 : U1);
      if (r == 1) load_x((ABL & 1) && (a.dbg & 4096) ? t0 : (t + 1 < t1 ? t + 1 : t));      // (4096: timing ablation, every chunk re-reads the first one's patches)
      const float* Vb = Vs + cur * VBUF;
      const float* Ub = r == 0 ? U0 : r == 1 ? U1 : r == 2 ? U2 : U3;
      float4 af[4], bf[2];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const float4*>(Vb + a_off[m]);
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const float4*>(Ub + b_off[n]);
      // element j of both fragments belongs to channel 4*(lane>>4)+j: MFMA j contracts channels {j, 4+j, 8+j, 12+j} of the chunk
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, bf[n].x, acc[r][m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, bf[n].y, acc[r][m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, bf[n].z, acc[r][m][n], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[r][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, bf[n].w, acc[r][m][n], 0, 0, 0);
      store_round(cur ^ 1, (r + 1) & 3);
#if HWG_W64D_SCHED
      // the next round's patch transform (DPP + FMA) and its LDS stores in the shadow of this round's MFMAs
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, HWG_W64D_SCHED, 0);
        if ((g & 7) == 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
#endif
      // filters of round R + 1 (issued one round ago) must have landed before the barrier lets the consumers at them; younger in the
      // in-order VMEM queue: this round's DMA (2) and the patch loads of rounds r == 1 (this round) / r == 2 (issued last round)
      // Raw barrier instead of __syncthreads(): its workgroup fence would wait for ALL pending LDS DMA (vmcnt(0)), i.e. also for the rounds
      // that are meant to stay in flight. The LDS stores of this round (patches) are drained with lgkmcnt(0).
      __builtin_amdgcn_sched_barrier(0);      // nothing of a later round (e.g. the masking of the patch loads) is to be pulled up into this one
      asm volatile("" ::: "memory");
      if (r == 1 || r == 2) __builtin_amdgcn_s_waitcnt((10 & 15) | (7 << 4) | (0 << 8));
      else __builtin_amdgcn_s_waitcnt((2 & 15) | (7 << 4) | (0 << 8));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // the last two rounds issued filter DMAs for rounds that do not exist (kept so that the VMEM counts are the same in every round): they must
  // have landed before the exchange image below reuses their LDS arrays
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool direct = a.nsplit == 1;
  int piece = split;
  int pieces = a.nsplit, slot_id = ntile * a.mt + mtile;      // partial images this tile is written in; its arrival counter
  if (BAL && !whole) {
    const int rel = unit / T_all;
    const int g_first = wino_bal_owner(a.bal, U_bal, rel * T_all), g_last = wino_bal_owner(a.bal, U_bal, rel * T_all + T_all - 1);
    direct = g_first == g_last;
    piece = wg - g_first;
    pieces = g_last - g_first + 1; slot_id = rel;
  }
  if constexpr ((ABL & 1) != 0) {
    if (a.dbg & 2048) {      // timing ablation: no output transform (the accumulators stay alive through one store that never happens)
      float keep = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) keep += acc[r][m][n][0] + acc[r][m][n][1] + acc[r][m][n][2] + acc[r][m][n][3];
      if (keep == 123456.789f) a.y[tid] = keep;
      if (!BAL || whole) break;
      unit += t1 - t0;
      if (unit >= unit_end) break;
      __syncthreads();
      continue;
    }
  }
  if constexpr (MT == 2) {
  // ---- output transform: over the rows a in registers (s0 = m0+m1+m2, s1 = m1-m2-m3), over the columns b through LDS ----------------
  // exchange image X[b][i][tile][channel]; C/D layout of a block: channel = lane & 15, tile = (lane >> 4) * 4 + e
  auto xplane = [&](int pl) -> float* { return pl < 4 ? smem + pl * XP : pl == 4 ? U0 : pl == 5 ? U1 : pl == 6 ? U2 : U3; };
  if (tid < 64) {
    const int m = m0 + tid;
    int base = -1, ext = 0;
    if (m < a.M) {
      const int tj = m % a.TQ;
      const int t2 = m / a.TQ;
      const int ti = t2 % a.TP;
      const int n = t2 / a.TP;
      base = (n * a.P + 2 * ti) * a.Q + 2 * tj;                         // pixel index of the tile's first output (the planner keeps N P Q K < 2^31)
      ext = (2 * ti + 1 < a.P ? 1 : 0) | (2 * tj + 1 < a.Q ? 2 : 0);    // second output row / column inside the image
    }
    tile_out[tid] = base; tile_ext[tid] = ext;
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float s0 = acc[0][m][n][e] + acc[1][m][n][e] + acc[2][m][n][e];
        const float s1 = acc[1][m][n][e] - acc[2][m][n][e] - acc[3][m][n][e];
        const int tile = m * 16 + (lane >> 4) * 4 + e, kk = (nh * 2 + n) * 16 + (lane & 15);
        xplane(bq * 2 + 0)[tile * LDK + kk] = s0;
        xplane(bq * 2 + 1)[tile * LDK + kk] = s1;
      }
  __syncthreads();
  const int kk = tid & 63;
  const int k = n0 + kk;
  float* yg = direct ? a.y : a.part + (long long)piece * ((long long)a.N * a.P * a.Q * a.K);
  const float bv = (direct && a.bias && k < a.K) ? a.bias[k] : 0.f;
  const bool accum = direct && a.accumulate, coh = !direct && a.cnt;      // coh: the partial image is read by another workgroup of this launch
#pragma unroll 2
  for (int tile = tid >> 6; tile < 64; tile += 8) {
    const int base = tile_out[tile], ext = tile_ext[tile];
    if (base < 0 || k >= a.K) continue;
    const long long ob = (long long)base * a.K + k;      // one 64-bit product per tile; the four outputs are 32-bit steps from it
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float r0 = xplane(0 * 2 + i)[tile * LDK + kk], r1 = xplane(1 * 2 + i)[tile * LDK + kk];
      const float r2 = xplane(2 * 2 + i)[tile * LDK + kk], r3 = xplane(3 * 2 + i)[tile * LDK + kk];
      if (i && !(ext & 1)) continue;
      const float y0 = r0 + r1 + r2, y1 = r1 - r2 - r3;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (j && !(ext & 2)) continue;
        const long long o = ob + (i * a.Q + j) * a.K;
        float v = (j == 0 ? y0 : y1) + bv;
        if (accum) v += yg[o];
        if constexpr ((ABL & 1) != 0) { if ((a.dbg & 1024) && v != 123456.789f) continue; }      // timing ablation: no global stores
        if (coh) hwg_store_agent(yg + o, v); else yg[o] = v;
      }
    }
  }
  } else {
    // F(3x3,2x2): A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]] - three output rows per tile, so twelve (b, i) planes: the exchange runs in two
    // passes of 32 tiles (planes of 32 x LDK floats: eight in smem, two in each filter array)
    constexpr int XH = 32 * LDK;
    auto xplane3 = [&](int pl) -> float* {
      return pl < 8 ? smem + pl * XH : pl < 10 ? U0 + (pl - 8) * XH : pl < 12 ? U1 + (pl - 10) * XH : pl < 14 ? U2 + (pl - 12) * XH : U3 + (pl - 14) * XH;
    };
    const int kk = tid & 63;
    const int k = n0 + kk;
    float* yg = direct ? a.y : a.part + (long long)piece * ((long long)a.N * a.P * a.Q * a.K);
    const float bv = (direct && a.bias && k < a.K) ? a.bias[k] : 0.f;
    const bool accum = direct && a.accumulate, coh = !direct && a.cnt;      // coh: the partial image is read by another workgroup of this launch
    const long long kch = (long long)(k / a.y_kc) * a.y_run + (k % a.y_kc);
    if (tid < 64) {
      const int m = m0 + tid;
      int base = -1, ext = 0;
      if (m < a.M) {
        const int tj = m % a.TQ;
        const int t2 = m / a.TQ;
        const int ti = t2 % a.TP;
        const int n = t2 / a.TP;
        base = n * a.y_img + 3 * ti * a.y_row + 3 * tj * a.y_pix;                 // element offset of the tile's first output (images < 2^31 elements)
        ext = min(3, a.P - 3 * ti) | (min(3, a.Q - 3 * tj) << 2);               // output rows / columns of the tile inside the image
      }
      tile_out[tid] = base; tile_ext[tid] = ext;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half) __syncthreads();
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = half * 2 + mh;
            const float a0 = acc[0][m][n][e], a1 = acc[1][m][n][e], a2 = acc[2][m][n][e], a3 = acc[3][m][n][e];
            const int tile = mh * 16 + (lane >> 4) * 4 + e, kc = (nh * 2 + n) * 16 + (lane & 15);
            xplane3(bq * 3 + 0)[tile * LDK + kc] = a0 + a1 + a2;
            xplane3(bq * 3 + 1)[tile * LDK + kc] = a1 - a2;
            xplane3(bq * 3 + 2)[tile * LDK + kc] = a1 + a2 + a3;
          }
      __syncthreads();
#pragma unroll 2
      for (int tl = tid >> 6; tl < 32; tl += 8) {
        const int base = tile_out[half * 32 + tl], ext = tile_ext[half * 32 + tl];
        if (base < 0 || k >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (i >= (ext & 3)) continue;
          const float r0 = xplane3(0 * 3 + i)[tl * LDK + kk], r1 = xplane3(1 * 3 + i)[tl * LDK + kk];
          const float r2 = xplane3(2 * 3 + i)[tl * LDK + kk], r3 = xplane3(3 * 3 + i)[tl * LDK + kk];
          const float yv[3] = {r0 + r1 + r2, r1 - r2, r1 + r2 + r3};
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            if (j >= (ext >> 2)) continue;
            const long long o = (long long)base + (long long)i * a.y_row + (long long)j * a.y_pix + kch;
            float v = yv[j] + bv;
            if (accum) v += yg[o];
            if (coh) hwg_store_agent(yg + o, v); else yg[o] = v;
          }
        }
      }
    }
  }
  // Partial images without a reduce launch: the workgroup that delivers its tile's last piece sums them in piece order (hwg_split_arrive_block) -
  // the arithmetic of conv_split_reduce_kernel (uniform split) / wino_bal_reduce_kernel (a tile cut by the balanced schedule's workgroup
  // boundaries), bit for bit. Every thread revisits the outputs it stored above (tile_out / tile_ext are still in LDS).
  if (!direct && a.cnt) {
    __shared__ int last_flag;
    if (hwg_split_arrive_block(a.cnt + slot_id, pieces, &last_flag)) {
      const long long total = (long long)a.N * a.P * a.Q * a.K;
      const int k = n0 + (tid & 63);
      const float bl = (a.bias && k < a.K) ? a.bias[k] : 0.f;
      auto finish = [&](long long o) {
        float v = BAL ? 0.f : hwg_load_agent(a.part + o);
        for (int sp = BAL ? 0 : 1; sp < pieces; ++sp) v += hwg_load_agent(a.part + sp * total + o);
        if (a.bias) v += bl;
        if (a.accumulate) v += a.y[o];
        a.y[o] = v;
      };
      if constexpr (MT == 2) {
        for (int tile = tid >> 6; tile < 64; tile += 8) {
          const int base = tile_out[tile], ext = tile_ext[tile];
          if (base < 0 || k >= a.K) continue;
          const long long ob = (long long)base * a.K + k;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            if (i && !(ext & 1)) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (j && !(ext & 2)) continue;
              finish(ob + (i * a.Q + j) * a.K);
            }
          }
        }
      } else {
        const long long kch = (long long)(k / a.y_kc) * a.y_run + (k % a.y_kc);
        for (int tile = tid >> 6; tile < 64; tile += 8) {
          const int base = tile_out[tile], ext = tile_ext[tile];
          if (base < 0 || k >= a.K) continue;
          for (int i = 0; i < (ext & 3); ++i)
            for (int j = 0; j < (ext >> 2); ++j) finish((long long)base + (long long)i * a.y_row + (long long)j * a.y_pix + kch);
        }
      }
    }
  }
  if (!BAL || whole) break;
  unit += t1 - t0;
  if (unit >= unit_end) break;
  __syncthreads();      // the exchange image has been read: the next segment's prologue may overwrite the operand buffers it aliases
  }
}


constexpr int WINO_DB = 4;      // filter rounds in the ring of the big kernel

__global__ __launch_bounds__(768) void wino_conv_big_kernel(WinoK a) {
  constexpr int TM = 64, TN = 64, NX = 256;
  constexpr int PSV = TM * 16 + 8, PSU = TN * 16 + 8;
  constexpr int VBUF = 4 * PSV, UBUF = 4 * PSU;
  constexpr int DA = WINO_DB;
  constexpr int XI = TM * 16 / NX;                    // 4
  constexpr int UD = TN / 16;                         // 4 DMA instructions per plane
  __shared__ __attribute__((aligned(16))) float smem[2 * VBUF + DA * UBUF];
  float* const Vs = smem;
  float* const Us = smem + 2 * VBUF;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int mtile, ntile, split;
  if (!wino_work_item(a, mtile, ntile, split)) return;
  const int m0 = mtile * TM;
  const int n0 = ntile * TN;
  const int T_all = a.C >> 4;
  const int t0 = (int)((long long)T_all * split / a.nsplit);
  const int t1 = (int)((long long)T_all * (split + 1) / a.nsplit);
  const int S = 4 * (t1 - t0);
  // Every workgroup walks the channel chunks in the same cyclic order but starts at a different one: otherwise all 256 CUs stream the
  // same 16 KB of filters at the same moment and queue up behind the few L2 channels that hold it (measured: the DMA issue alone took
  // 1700 cycles per round). The sum over the chunks is order independent up to fp32 rounding and fixed per workgroup.
  const int TC = t1 - t0;
  const int rot = TC > 0 ? (int)(((unsigned)mtile * 7u + (unsigned)ntile * 3u) % (unsigned)TC) : 0;
  auto chunk_of = [&](int tau) { int c = tau + rot; if (c >= TC) c -= TC; return t0 + c; };   // tau in [0, TC)

  if (wid >= 8) {
    // =========================== producers (wavefronts 8..11) ===========================
    const int pt = tid - 512;
    const int pw = wid - 8;                           // filter plane moved by this wave
    // item it of a thread = (tile row (pt >> 4) + 16 it, 4-channel group (pt >> 2) & 3, patch column pt & 3): column, channel group, signs
    // of the row transform and the swizzle term are the same for all four items
    int x_off[XI], x_ok[XI];
    const int xb = pt & 3, xc4 = (pt >> 2) & 3;
    const int x_lds0 = xb * PSV + swz(pt >> 4, xc4);  // + 256 it
    const float x_sa = (xb == 3) ? -1.f : 1.f, x_sb = (xb & 1) ? 1.f : -1.f;
    const int row_stride = a.W * a.C;
#pragma unroll
    for (int it = 0; it < XI; ++it) {
      const int b = xb, c4 = xc4, row = (pt >> 4) + 16 * it;
      const int m = m0 + row;
      const bool mok = m < a.M;
      const int mm = mok ? m : 0;
      const int tj = mm % a.TQ;
      const int t2 = mm / a.TQ;
      const int ti = t2 % a.TP;
      const int n = t2 / a.TP;
      const int w = 2 * tj - a.pw + b;
      const bool wok = mok && w >= 0 && w < a.W;
      const int h0 = 2 * ti - a.ph;
      x_off[it] = ((n * a.H + h0) * a.W + (wok ? w : 0)) * a.C + c4 * 4;
      int okm = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (wok && h0 + r >= 0 && h0 + r < a.H) okm |= 1 << r;
      x_ok[it] = okm;
    }
    int u_src[UD];
#pragma unroll
    for (int i = 0; i < UD; ++i) {
      const int row = 16 * i + (lane >> 2), slot = lane & 3;
      int k = n0 + row;
      if (k >= a.Kpad) k = a.Kpad - 1;
      u_src[i] = (pw * a.Kpad + k) * 16 + 4 * (slot ^ ((row >> 1) & 3));
    }
    const long long u_round = 4LL * a.Kpad * 16;
    float4 raw[XI][4], tr[XI][4];
    auto load_x = [&](int t) {
#pragma unroll
      for (int it = 0; it < XI; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int off = (x_ok[it] >> r) & 1 ? x_off[it] + r * row_stride + t * 16 : 0;
          raw[it][r] = *reinterpret_cast<const float4*>(a.x + off);
        }
    };
    auto col_transform = [&]() {
#pragma unroll
      for (int it = 0; it < XI; ++it) {
        float4 d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d[r] = raw[it][r];
          if (!((x_ok[it] >> r) & 1)) d[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        tr[it][0] = make_float4(d[0].x - d[2].x, d[0].y - d[2].y, d[0].z - d[2].z, d[0].w - d[2].w);
        tr[it][1] = make_float4(d[1].x + d[2].x, d[1].y + d[2].y, d[1].z + d[2].z, d[1].w + d[2].w);
        tr[it][2] = make_float4(d[2].x - d[1].x, d[2].y - d[1].y, d[2].z - d[1].z, d[2].w - d[1].w);
        tr[it][3] = make_float4(d[1].x - d[3].x, d[1].y - d[3].y, d[1].z - d[3].z, d[1].w - d[3].w);
      }
    };
    auto dma_u = [&](int ring, int round) {
      const float* base = a.u + (long long)round * u_round;
      float* dst = Us + ring * UBUF + pw * PSU;
#pragma unroll
      for (int i = 0; i < UD; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + u_src[i]),
                                         (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
    };
    auto write_v = [&](int buf, int r) {
      float* B = Vs + buf * VBUF;
#pragma unroll
      for (int it = 0; it < XI; ++it) {
        const float4 v = tr[it][r];
        float4 o;
        o.x = x_sa * v.x + x_sb * quad_partner(v.x);
        o.y = x_sa * v.y + x_sb * quad_partner(v.y);
        o.z = x_sa * v.z + x_sb * quad_partner(v.z);
        o.w = x_sa * v.w + x_sb * quad_partner(v.w);
        *reinterpret_cast<float4*>(B + x_lds0 + it * 256) = o;
      }
    };
    // VMEM bookkeeping for the filter DMAs (in-order counter shared with the patch loads): hist[j] = operations issued in each of the
    // last DA steps (oldest first); every step's DMA is the FIRST operation of its step.
    int hist[DA], hdma[DA];
#pragma unroll
    for (int j = 0; j < DA; ++j) { hist[j] = 0; hdma[j] = 0; }
    int issued = 0;
    // prologue: patches of chunk t0 (transformed, round 0 stored), filter rounds 0 .. DA-1 in flight
    if (S > 0) {
      load_x(chunk_of(0));
      col_transform();
      write_v(0, 0);
    }
    for (; issued < DA && issued < S; ++issued) dma_u(issued, 4 * chunk_of(issued >> 2) + (issued & 3));
    wait_vmcnt((issued - 1) * UD);                    // round 0 landed: (issued - 1) younger DMA groups may stay in flight
    // history as if those had been issued in the DA steps before step 0 (round j by "step j - DA")
#pragma unroll
    for (int j = 0; j < DA; ++j) { hdma[j] = j < issued ? UD : 0; hist[j] = hdma[j]; }
    __syncthreads();
    int ring = 0;
    for (int t = t0; t < t1; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // step s: the consumers multiply round s. This side stores the patches of round s + 1, issues the filter DMA of round s + DA
        // (slot of round s ... free only after this step's barrier, so it is issued for the slot the consumers left LAST step: round
        // s - 1 + DA), and makes sure the filters of round s + 1 have landed before the barrier.
        const int s = 4 * (t - t0) + r;
        int ops = 0, dops = 0;
        if (s >= 1 && issued < S) { dma_u(ring, 4 * chunk_of(issued >> 2) + (issued & 3)); ++issued; ring = ring == DA - 1 ? 0 : ring + 1; ops = dops = UD; }
        if (s + 1 < S) {
          if (r == 3) col_transform();                // chunk t + 1 (fetched two steps ago, when half of this chunk's registers were free)
          write_v((r + 1) & 1, (r + 1) & 3);
          if (r == 1 && t + 1 < t1) { load_x(chunk_of(t + 1 - t0)); ops += XI * 4; }
        }
        // shift the history, then: round s + 1 was issued DA - 1 steps ago at the earliest ... find it: it is the DMA of history entry
        // DA - 1 - (issued - 1 - (s + 1)) = the (issued - s - 2)-th youngest DMA group
#pragma unroll
        for (int j = 0; j + 1 < DA; ++j) { hist[j] = hist[j + 1]; hdma[j] = hdma[j + 1]; }
        hist[DA - 1] = ops; hdma[DA - 1] = dops;
        if (s + 1 < S) {
          // newer than round s + 1's DMA: the rest of its own step + all later steps. Groups are issued one per step in round order,
          // so round s + 1 sits (issued - 1 - (s + 1)) DMA-carrying steps back from the youngest DMA-carrying step.
          int back = issued - 2 - s;                  // DMA groups younger than round s + 1
          int n = 0, seen = 0;
#pragma unroll
          for (int j = DA - 1; j >= 0; --j) {
            if (seen < back || (seen == back && hdma[j] == 0)) { n += hist[j]; seen += hdma[j] ? 1 : 0; }
            else if (seen == back && hdma[j] != 0) { n += hist[j] - hdma[j]; seen = back + 1; }
          }
          wait_vmcnt(n);
        }
        __syncthreads();
      }
    }
    return;
  }

  // =========================== consumers (wavefronts 0..7) ===========================
  const int cm = wid >> 2, cn = wid & 3;
  const int frow = lane & 15, fchunk = lane >> 4;
  const int a_off0 = swz(cm * 32 + frow, fchunk);
  const int a_off1 = swz(cm * 32 + 16 + frow, fchunk);
  const int b_off = swz(cn * 16 + frow, fchunk);

  f32x4 acc[16][2];
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    acc[p][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[p][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  int ring = 0;
  for (int t = t0; t < t1; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* A = Vs + (r & 1) * VBUF;
      const float* B = Us + ring * UBUF + b_off;
      ring = ring == DA - 1 ? 0 : ring + 1;
      float4 fa0[2], fa1[2], fb[2];
      fa0[0] = *reinterpret_cast<const float4*>(A + a_off0);
      fa1[0] = *reinterpret_cast<const float4*>(A + a_off1);
      fb[0] = *reinterpret_cast<const float4*>(B);
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        const int c = pb & 1;
        if (pb < 3) {
          fa0[c ^ 1] = *reinterpret_cast<const float4*>(A + (pb + 1) * PSV + a_off0);
          fa1[c ^ 1] = *reinterpret_cast<const float4*>(A + (pb + 1) * PSV + a_off1);
          fb[c ^ 1] = *reinterpret_cast<const float4*>(B + (pb + 1) * PSU);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4& c0 = acc[r * 4 + pb][0];
        f32x4& c1 = acc[r * 4 + pb][1];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[c].x, fb[c].x, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[c].x, fb[c].x, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[c].y, fb[c].y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[c].y, fb[c].y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[c].z, fb[c].z, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[c].z, fb[c].z, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[c].w, fb[c].w, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[c].w, fb[c].w, c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }

  const int k = n0 + cn * 16 + (lane & 15);
  const bool direct = a.nsplit == 1;
  float* yg = direct ? a.y : a.part + (long long)split * ((long long)a.N * a.P * a.Q * a.K);
  const float bv = (direct && a.bias && k < a.K) ? a.bias[k] : 0.f;
  const bool accum = direct && a.accumulate;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + cm * 32 + blk * 16 + (lane >> 4) * 4 + e;
      if (m >= a.M || k >= a.K) continue;
      const int tj = m % a.TQ;
      const int t2 = m / a.TQ;
      const int ti = t2 % a.TP;
      const int n = t2 / a.TP;
      float mm[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) mm[p] = acc[p][blk][e];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float s0 = mm[0 * 4 + j] + mm[1 * 4 + j] + mm[2 * 4 + j];
        const float s1 = mm[1 * 4 + j] - mm[2 * 4 + j] - mm[3 * 4 + j];
        mm[0 * 4 + j] = s0;
        mm[1 * 4 + j] = s1;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = 2 * ti + i;
        if (p >= a.P) continue;
        const float y0 = mm[i * 4 + 0] + mm[i * 4 + 1] + mm[i * 4 + 2];
        const float y1 = mm[i * 4 + 1] - mm[i * 4 + 2] - mm[i * 4 + 3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int q = 2 * tj + j;
          if (q >= a.Q) continue;
          const long long o = (((long long)n * a.P + p) * a.Q + q) * a.K + k;
          float v = (j == 0 ? y0 : y1) + bv;
          if (accum) v += yg[o];
          yg[o] = v;
        }
      }
    }
}

__global__ __launch_bounds__(256) void wino_pack_weight_kernel(const float* src, float* dst, int A, int Apad, int B, int Bpad,
                                                               long long sa, long long sb, long long sr, long long ss, int flip) {
  const long long total = (long long)Apad * Bpad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    wino_pack_one(src, dst, i, A, Apad, B, Bpad, sa, sb, sr, ss, flip);
}

// Balanced schedule (wino_conv64d_kernel, a.bal > 0): sums the numbered pieces of every tile the workgroup boundaries cut (+ bias, + the
// accumulate operand) into y; tiles written directly (whole-tile region, or inside one workgroup's run) are left alone.
template <int VEC>
__global__ __launch_bounds__(256) void wino_bal_reduce_kernel(WinoK a, int T_all) {
  const int KV = a.K / VEC;
  const int total = a.N * a.P * a.Q * KV;
  const long long plane = (long long)a.N * a.P * a.Q * a.K;
  const int U_bal = (a.mt * a.nt - a.bal_tile0) * T_all;
  const int e = a.mt_edge;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = (i % KV) * VEC, pix = i / KV;
    const int q = pix % a.Q, t = pix / a.Q, p = t % a.P, n = t / a.P;
    const int m = (n * a.TP + p / e) * a.TQ + q / e;
    const int rel = (m >> 6) + a.mt * (k >> 6) - a.bal_tile0;
    if (rel < 0) continue;
    const int g_first = wino_bal_owner(a.bal, U_bal, rel * T_all), g_last = wino_bal_owner(a.bal, U_bal, rel * T_all + T_all - 1);
    if (g_first == g_last) continue;
    const long long o = (long long)n * a.y_img + (long long)p * a.y_row + (long long)q * a.y_pix + (long long)(k / a.y_kc) * a.y_run + k % a.y_kc;
    float s[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = 0.f;
    for (int j = 0; j <= g_last - g_first; ++j) {
      if constexpr (VEC == 4) {
        const float4 v = *reinterpret_cast<const float4*>(a.part + j * plane + o);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      } else {
        s[0] += a.part[j * plane + o];
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if (a.bias) s[v] += a.bias[k + v];
      if (a.accumulate) s[v] += a.y[o + v];
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(a.y + o) = make_float4(s[0], s[1], s[2], s[3]);
    else a.y[o] = s[0];
  }
}

struct WinoPlan {
  int cfg;      // 0: 64 tiles x 32 k, 1: 32 x 64, 2: 128 x 16 (all-purpose kernel); 4: 32 x 64 (wave-specialised kernel); 5: 64 x 64 (12 waves);
                // 6: 64 x 64 with register-level operand reuse (wino_conv64_kernel); 3 unused
  int tm, tn, nsplit;
  int bal, bal_tile0, bal_pieces;   // cfg 6 only, bal > 0: balanced schedule (WinoK.bal / bal_tile0), at most bal_pieces partial images
  int bal_segs, bal_cut, bal_cut_pieces;
  double model_s;   // modelled duration of the chosen schedule (seconds)
};
static void wino_cfg(WinoPlan& p, int cfg) {
  static const int tms[8] = {64, 32, 128, 64, 32, 64, 64, 32}, tns[8] = {32, 64, 16, 32, 64, 64, 64, 32};
  p.cfg = cfg; p.tm = tms[cfg]; p.tn = tns[cfg];
  p.bal = p.bal_tile0 = p.bal_pieces = p.bal_segs = p.bal_cut = p.bal_cut_pieces = 0;
}
// Balanced schedule of the 64 x 64 DMA kernel for `tiles` workgroup tiles of `chunks` channel chunks: the first floor(tiles / 256) * 256 tiles
// run as whole-tile workgroups (full rounds of the chip), the rest - the round that would be partly filled - is cut into G equal runs of
// (tile, chunk) units. Returns false where that cannot help (nothing left over, or the chip is filled anyway). bal_segs = most tiles one
// workgroup touches, bal_cut = tiles that are cut, bal_cut_pieces = their pieces in total (the reduce pass reads those and writes bal_cut tiles).
static bool wino_balance(WinoPlan& p, long long tiles, int chunks, int g_force = 0, int a_force = -1) {
  const long long lead = a_force >= 0 ? (long long)a_force : tiles / 256 * 256;
  const long long rem = tiles - lead;
  if (lead % 256 || rem <= 0 || (g_force <= 0 && rem >= 240)) return false;
  const long long units = rem * chunks;
  long long G = g_force > 0 ? g_force : (units < 256 ? units : 256);
  if (G > units) G = units;
  if (G < 1 || (g_force <= 0 && G <= rem) || units * G >= (1ll << 31) || units >= (1ll << 24)) return false;
  p.bal = (int)G; p.bal_tile0 = (int)lead; p.nsplit = 1;
  p.bal_pieces = 1; p.bal_segs = 1; p.bal_cut = 0; p.bal_cut_pieces = 0;
  for (long long t = 0; t < rem; ++t) {
    const int n = wino_bal_owner((int)G, (int)units, (int)(t * chunks + chunks - 1)) - wino_bal_owner((int)G, (int)units, (int)(t * chunks)) + 1;
    if (n > p.bal_pieces) p.bal_pieces = n;
    if (n > 1) { ++p.bal_cut; p.bal_cut_pieces += n; }
  }
  for (long long g = 0; g < G; ++g) {
    const long long u0 = units * g / G, u1 = units * (g + 1) / G;
    if (u1 > u0) p.bal_segs = std::max(p.bal_segs, (int)((u1 - 1) / chunks - u0 / chunks + 1));
  }
  return true;
}
// Schedule = (kernel variant, channel split) with the smallest modelled time. Model (fitted to tools/wino_check.py sweeps on MI355X,
// profiles/r02_wino_shapes.txt): one workgroup per CU at a time, so the launch takes ceil(workgroups / 256) rounds of
// (fixed + rounds-of-the-channel-loop x step) microseconds; a channel split adds the pass that sums the partial outputs.
struct WinoCost { int cfg; double fixed_us, step_us; };
static const WinoCost kWinoCost[6] = {{1, 2.5, 1.17}, {5, 11.0, 2.09}, {0, 8.0, 1.2}, {2, 6.5, 1.5}, {6, 15.7, 1.04}, {7, 2.0, 1.0}};
static WinoPlan plan_wino_model(const hwg_conv_desc* d);
static WinoPlan plan_wino(const hwg_conv_desc* d, double* model_s = nullptr) {
  static thread_local HwgPlanCache<WinoPlan> cache;
  const WinoPlan p = cache.get(d, plan_wino_model);
  if (model_s) *model_s = p.model_s;
  return p;
}
static WinoPlan plan_wino_model(const hwg_conv_desc* d) {
  WinoPlan best;
  wino_cfg(best, d->K <= 16 ? 2 : d->K <= 48 ? 0 : 1);
  best.nsplit = 1;
  const long long M = (long long)d->N * hwg_cdiv(d->P, 2) * hwg_cdiv(d->Q, 2);
  const int chunks = d->C / 16;
  const double out_bytes = 4.0 * d->N * d->P * d->Q * d->K;
  double best_t = 1e30;
  WinoCost cost6 = kWinoCost[4];              // tuning aid: HWG_WINO_COST6="fixed_us,step_us" overrides the 64x64 DMA kernel's model constants
  if (const char* e = hwg_tune().wino_cost6; *e) { double f = 0, s2 = 0; if (sscanf(e, "%lf,%lf", &f, &s2) == 2) { cost6.fixed_us = f; cost6.step_us = s2; } }
  for (int ci = 0; ci < 6; ++ci) {
    const WinoCost& wc = ci == 4 ? cost6 : kWinoCost[ci];
    if (wc.cfg == 7 && (d->K <= 16 || d->K > 64)) continue;     // 32 x 32 tiles (4 waves): pays on 32..64-channel layers with few tiles
    if (wc.cfg == 2 && d->K > 16) continue;
    if (wc.cfg == 0 && d->K > 48) continue;
    if ((wc.cfg == 1 || wc.cfg == 5 || wc.cfg == 6) && d->K <= 48) continue;
    if (wc.cfg == 5 && d->C < 64) continue;
    WinoPlan p;
    wino_cfg(p, wc.cfg);
    const long long blocks = (long long)hwg_cdiv(M, p.tm) * hwg_cdiv(d->K, p.tn);
    for (int ns = 1; ns <= 8 && ns * 2 <= (chunks > 1 ? chunks : 2); ns *= 2) {
      if (ns > 1 && (chunks / ns < 2 || out_bytes * ns > 1.5e9)) break;
      // workgroups are handed out dynamically: beyond three full rounds the last, partly filled one costs about half a round, not a whole
      // one (round 4, tools/plan_sweep.py: 8x6x127x512->512 at split 4 = 4.1 rounds measured 258 us, modelled 272 with ceil -> 253)
      const double q = (double)(blocks * ns) / 256.0;
      const double rounds = q <= 3.0 ? (double)hwg_cdiv(blocks * ns, 256) : q + 0.5;
      double t = rounds * (wc.fixed_us + 4.0 * hwg_cdiv(chunks, ns) * wc.step_us) * 1e-6;
      if (ns > 1) t += (ns + 1) * out_bytes / 3.0e12 + 6e-6;
      if (t < best_t) { best_t = t; best = p; best.nsplit = ns; }
    }
  }
  // balanced schedule of the 64 x 64 DMA kernel (wino_balance): whole-tile rounds + the leftover tiles cut into equal unit runs, the cut tiles
  // summed by one pass over their pieces
  const int bal_mode = hwg_tune().wino_bal[0] ? atoi(hwg_tune().wino_bal) : 0;      // HWG_WINO_BAL: -1 never, 0 by model, "G[,lead tiles]" forced
  if (d->K > 48 && bal_mode >= 0 && !hwg_tune().w64_nodma && !(hwg_tune().conv_dbg & 512) && (long long)d->N * d->P * d->Q * d->K < (1ll << 31)) {
    WinoPlan p;
    wino_cfg(p, 6);
    const long long tiles = (long long)hwg_cdiv(M, 64) * hwg_cdiv(d->K, 64);
    int gf = 0, af = -1;
    if (bal_mode > 0) sscanf(hwg_tune().wino_bal, "%d,%d", &gf, &af);
    if (wino_balance(p, tiles, chunks, gf, af) && (p.bal_pieces <= 1 || (double)p.bal_pieces * out_bytes <= 1.5e9)) {
      // fitted on tools/probes/probe_r5_bal.txt (profiles/r05_probe_bal.txt): a tail workgroup starts while the last whole-tile ones drain
      // (4 us of its fixed cost hidden), every further tile it touches costs 5 us (output transform + prologue)
      const long long units = (tiles - p.bal_tile0) * chunks;
      const long long c = hwg_cdiv(units, (long long)p.bal);
      double t = (p.bal_tile0 / 256) * (cost6.fixed_us + 4.0 * chunks * cost6.step_us) * 1e-6;
      t += (double)hwg_cdiv((long long)p.bal, 256ll) * (cost6.fixed_us - 4.0 + (p.bal_segs - 1) * 5.0 + 4.0 * c * cost6.step_us) * 1e-6;
      if (p.bal_cut) t += (p.bal_cut_pieces + p.bal_cut) * (out_bytes / tiles) / 3.0e12 + out_bytes / 2.0e13 + 6e-6;
      if (bal_mode > 0 || t < 0.97 * best_t) { best = p; best_t = t; }
    }
  }
  if (const char* f = hwg_tune().wino_force; *f) {   // tuning aid: "cfg[,nsplit]"
    int fc = -1, fs = 0;
    const int n = sscanf(f, "%d,%d", &fc, &fs);
    if (n >= 1 && fc >= 0 && fc <= 7 && fc != 3 && fc != 4 && (fc == 2 || d->K > 16)) { const int ns = best.nsplit; wino_cfg(best, fc); best.nsplit = ns; }
    if (n >= 2 && fs >= 1) best.nsplit = fs > chunks ? chunks : fs;
  }
  best.model_s = best_t;
  return best;
}

}  // namespace

extern "C" int hwg_wino_supported(const hwg_conv_desc* d) {
  if (!d || d->transposed) return 0;
  if (d->R != 3 || d->S != 3 || d->stride_h != 1 || d->stride_w != 1 || d->dil_h != 1 || d->dil_w != 1) return 0;
  if (d->C % 16 != 0 || d->K < 16) return 0;
  if (d->P != d->H + 2 * d->pad_h - 2 || d->Q != d->W + 2 * d->pad_w - 2) return 0;
  return hwg_tune().wino == 0 ? 0 : 1;
}

double hwg_conv_direct_model_seconds(const hwg_conv_desc* d);

/* 1 when the Winograd schedule is modelled faster than the direct implicit-GEMM one (few tiles x many channels - the 512-channel layers
 * of the recogniser at 8 x 129 - stream 16 MB of transform-domain filters per 2000 tiles and stay on the direct kernels) */
extern "C" int hwg_wino_preferred(const hwg_conv_desc* d) {
  if (!hwg_wino_supported(d)) return 0;
  if (hwg_tune().wino == 2) return 1;     // 2: always, 0: never (hwg_wino_supported), default: by model
  double tw = 0.0;
  (void)plan_wino(d, &tw);
  return tw < 0.97 * hwg_conv_direct_model_seconds(d) ? 1 : 0;
}

extern "C" size_t hwg_wino_weight_floats(int A, int B) {
  const size_t Apad = (size_t)(A + 15) / 16 * 16, Bpad = (size_t)(B + 15) / 16 * 16;
  return 16 * Apad * Bpad;
}

extern "C" int hwg_wino_pack_weight(const float* src, float* dst, int A, int B, long long sa, long long sb, long long sr, long long ss,
                                    int flip, void* stream) {
  HWG_REQUIRE(src && dst && A > 0 && B > 0, "wino_pack_weight: bad arguments");
  const int Apad = (A + 15) / 16 * 16, Bpad = (B + 15) / 16 * 16;
  hipLaunchKernelGGL(wino_pack_weight_kernel, dim3(hwg_stream_grid((long long)Apad * Bpad, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, A, Apad,
                     B, Bpad, sa, sb, sr, ss, flip);
  HWG_LAUNCH_CHECK("wino_pack_weight");
  return HWG_OK;
}

extern "C" size_t hwg_wino_conv_workspace(const hwg_conv_desc* d) {
  if (!hwg_wino_supported(d)) return 0;
  const WinoPlan p = plan_wino(d);
  if (p.bal > 0) return p.bal_pieces > 1 ? (size_t)p.bal_pieces * d->N * d->P * d->Q * d->K * sizeof(float) : 0;
  if (p.nsplit <= 1) return 0;
  return (size_t)p.nsplit * d->N * d->P * d->Q * d->K * sizeof(float);
}

int hwg_conv_split_reduce_launch(const float* part, const float* bias, float* y, long long total, int K, int nsplit, int accumulate, hipStream_t st);

extern "C" int hwg_wino_conv_fwd(const hwg_conv_desc* d, const float* x, const float* u, const float* bias, float* y, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  HWG_REQUIRE(d && x && u && y, "wino_conv_fwd: null pointer");
  HWG_REQUIRE(hwg_wino_supported(d), "wino_conv_fwd: needs a 3x3 stride-1 dilation-1 convolution with C %% 16 == 0 and K >= 16");
  hipStream_t st = (hipStream_t)stream;
  const WinoPlan p = plan_wino(d);
  const size_t need = hwg_wino_conv_workspace(d);
  if (need && (!workspace || workspace_bytes < need)) {
    hwg_set_error("wino_conv_fwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  WinoK k;
  k.x = x; k.u = u; k.bias = bias; k.y = y;
  k.N = d->N; k.H = d->H; k.W = d->W; k.C = d->C; k.K = d->K; k.Kpad = (d->K + 15) / 16 * 16;
  k.P = d->P; k.Q = d->Q; k.ph = d->pad_h; k.pw = d->pad_w;
  k.TP = hwg_cdiv(d->P, 2); k.TQ = hwg_cdiv(d->Q, 2);
  k.M = d->N * k.TP * k.TQ;
  k.accumulate = accumulate;
  k.nsplit = p.nsplit;
  k.part = (float*)workspace;
  k.mt = hwg_cdiv(k.M, p.tm); k.nt = hwg_cdiv(d->K, p.tn);
  k.xcd_order = hwg_tune().wino_order;
  k.bal = p.bal; k.bal_tile0 = p.bal_tile0;
  k.mt_edge = 2; k.dbg = hwg_tune().conv_dbg;
  k.x_img = k.x_row = k.x_pix = k.x_tc = k.x_run = 0;      // (the F(2x2,3x3) kernels address the plain NHWC image themselves)
  k.y_img = d->P * d->Q * d->K; k.y_row = d->Q * d->K; k.y_pix = d->K; k.y_kc = d->K; k.y_run = 0;
  dim3 grid(p.bal > 0 ? p.bal_tile0 + (p.bal + 7) / 8 * 8 : (k.mt * k.nt * p.nsplit + 7) / 8 * 8);
  k.cnt = nullptr;
  const bool counted = p.cfg == 0 || p.cfg == 1 || p.cfg == 2 || p.cfg == 7 || (p.cfg == 6 && !hwg_tune().w64_nodma && !(hwg_tune().conv_dbg & 512));
  if ((p.bal > 0 ? p.bal_pieces > 1 : p.nsplit > 1) && counted && hwg_tune().split_inkernel && (long long)k.mt * k.nt * 8 <= HWG_SPLIT_COUNTERS) {
    k.cnt = hwg_split_counters(st);        // (8: the most wavefront blocks a workgroup tile of these kernels has)
    if (!k.cnt) return HWG_ERR_LAUNCH;
  }
  const int prof = hwg_prof_open(HWG_PROF_CONV_WINO, 2.0 * d->N * d->P * d->Q * (double)d->K * d->C * 9.0, st);
  if (p.cfg == 0) hipLaunchKernelGGL((wino_conv_kernel<4, 2>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 1) hipLaunchKernelGGL((wino_conv_kernel<2, 4>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 2) hipLaunchKernelGGL((wino_conv_kernel<8, 1>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 7) hipLaunchKernelGGL((wino_conv_kernel<2, 2>), grid, dim3(256), 0, st, k);
  else if (p.cfg == 5) hipLaunchKernelGGL(wino_conv_big_kernel, grid, dim3(768), 0, st, k);
  else if (p.cfg == 6 && !hwg_tune().w64_nodma && (hwg_tune().conv_dbg & 512)) hipLaunchKernelGGL((wino_conv64d_kernel<1, false>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 6 && !hwg_tune().w64_nodma && p.bal > 0) hipLaunchKernelGGL((wino_conv64d_kernel<0, true>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 6 && !hwg_tune().w64_nodma) hipLaunchKernelGGL((wino_conv64d_kernel<0, false>), grid, dim3(512), 0, st, k);
  else if (p.cfg == 6) hipLaunchKernelGGL(wino_conv64_kernel, grid, dim3(512), 0, st, k);
  else hipLaunchKernelGGL((wino_conv_ws_kernel<1, 4>), grid, dim3(512), 0, st, k);
  hwg_prof_close(prof, st);
  hwg_note_plan(HWG_PROF_CONV_WINO, p.cfg, p.bal > 0 ? -p.bal : p.nsplit);
  HWG_LAUNCH_CHECK("wino_conv_fwd");
  if (k.cnt) {
    // (partial images summed inside the kernel)
  } else if (p.bal > 0 && p.bal_pieces > 1) {
    const long long total = (long long)d->N * d->P * d->Q * d->K;
    const double cut = (double)(k.mt * k.nt - p.bal_tile0) / (k.mt * k.nt);
    const int prof2 = hwg_prof_open(HWG_PROF_CONV_REDUCE, 4.0 * total * cut * (p.bal_pieces + 1), st);
    const int vec = d->K % 4 == 0 ? 4 : 1;
    const dim3 rgrid(hwg_stream_grid(total / vec, 256));
    if (vec == 4) hipLaunchKernelGGL(wino_bal_reduce_kernel<4>, rgrid, dim3(256), 0, st, k, d->C / 16);
    else hipLaunchKernelGGL(wino_bal_reduce_kernel<1>, rgrid, dim3(256), 0, st, k, d->C / 16);
    hwg_prof_close(prof2, st);
    HWG_LAUNCH_CHECK("wino_bal_reduce");
  } else if (p.nsplit > 1) {
    const long long total = (long long)d->N * d->P * d->Q * d->K;
    const int prof2 = hwg_prof_open(HWG_PROF_CONV_REDUCE, 4.0 * total * (p.nsplit + 1), st);
    const int rc = hwg_conv_split_reduce_launch((const float*)workspace, bias, y, total, d->K, p.nsplit, accumulate, st);
    hwg_prof_close(prof2, st);
    if (rc) return rc;
  }
  return HWG_OK;
}


// ---- F(3x3,2x2): 4x4 stride-2 pad-0 convolutions and their data gradients ---------------------------------------------------------------------
// y[p,q] = sum_{r,s<4} x[2p+r, 2q+s] w[r,s]; with r = 2u+a, s = 2v+b this is a stride-1 2x2-tap convolution (u, v) over the space-to-depth image
// x'[i, j, (a,b,c)] = x[2i+a, 2j+b, c] - no copy: a row of x' is a row PAIR of x, its channels two runs of 2C floats one row apart. The data
// gradient is the 2x2-tap full correlation of dy to the 4C channels (a,b,c) of dx's 2x2 blocks, written depth-to-space. Two taps per axis
// take F(3x3,2x2): 16 multiplies per 9 outputs x 4 taps = the same 2.25 x as F(2x2,3x3), on the same 4x4 patches with the same input
// transform, so it is the 64 x 64 DMA kernel with MT = 3 (output transform A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]]; filters G g G^T with
// G = [[1,0],[.5,.5],[.5,-.5],[0,1]], the sign of the fourth position folded in because the kernel's input transform is F(2x2,3x3)'s).
namespace {

struct S2Geom {
  int N, Hv, Wv, Cv, Kv, Pv, Qv, pad;
  int x_img, x_row, x_pix, x_tc, x_run, y_img, y_row, y_pix, y_kc, y_run;
  double flops;
};

bool s2_geom(const hwg_conv_desc* d, S2Geom& g) {
  if (!d || d->R != 4 || d->S != 4 || d->stride_h != 2 || d->stride_w != 2 || d->pad_h || d->pad_w || d->dil_h != 1 || d->dil_w != 1) return false;
  if (d->C % 16 || d->K % 4 || d->N < 1 || d->H < 1 || d->W < 1) return false;
  const long long in_e = (long long)d->N * d->H * d->W * d->C, out_e = (long long)d->N * d->P * d->Q * d->K;
  if (in_e >= (1ll << 31) || out_e >= (1ll << 31)) return false;
  g.N = d->N;
  if (!d->transposed) {
    if (d->H < 4 || d->W < 4 || d->P != (d->H - 4) / 2 + 1 || d->Q != (d->W - 4) / 2 + 1) return false;
    g.Hv = d->P + 1; g.Wv = d->Q + 1; g.Cv = 4 * d->C; g.Kv = d->K; g.Pv = d->P; g.Qv = d->Q; g.pad = 0;
    g.x_img = d->H * d->W * d->C; g.x_row = 2 * d->W * d->C; g.x_pix = 2 * d->C; g.x_tc = 2 * d->C / 16; g.x_run = d->W * d->C;
    g.y_img = d->P * d->Q * d->K; g.y_row = d->Q * d->K; g.y_pix = d->K; g.y_kc = d->K; g.y_run = 0;
    g.flops = 2.0 * d->N * d->P * d->Q * (double)d->K * d->C * 16.0;
  } else {
    if (d->P != 2 * d->H + 2 || d->Q != 2 * d->W + 2) return false;      // (an odd image would keep a last row / column the blocks never write)
    g.Hv = d->H; g.Wv = d->W; g.Cv = d->C; g.Kv = 4 * d->K; g.Pv = d->H + 1; g.Qv = d->W + 1; g.pad = 1;
    g.x_img = d->H * d->W * d->C; g.x_row = d->W * d->C; g.x_pix = d->C; g.x_tc = d->C / 16; g.x_run = 0;
    g.y_img = d->P * d->Q * d->K; g.y_row = 2 * d->Q * d->K; g.y_pix = 2 * d->K; g.y_kc = 2 * d->K; g.y_run = d->Q * d->K;
    g.flops = 2.0 * d->N * d->H * d->W * (double)d->K * d->C * 16.0;
  }
  return g.Kv > 48;
}

WinoPlan plan_wino_s2_model(const hwg_conv_desc* d) {
  WinoPlan best;
  wino_cfg(best, 6);
  best.nsplit = 1;
  best.model_s = 1e30;
  S2Geom g;
  if (!s2_geom(d, g)) return best;
  const long long M = (long long)g.N * hwg_cdiv(g.Pv, 3) * hwg_cdiv(g.Qv, 3);
  const int chunks = g.Cv / 16;
  const double out_bytes = 4.0 * g.N * g.Pv * g.Qv * g.Kv;
  const WinoCost wc = kWinoCost[4];
  const long long tiles = (long long)hwg_cdiv(M, 64) * hwg_cdiv(g.Kv, 64);
  double best_t = 1e30;
  for (int ns = 1; ns <= 8 && ns * 2 <= (chunks > 1 ? chunks : 2); ns *= 2) {
    if (ns > 1 && (chunks / ns < 2 || out_bytes * ns > 1.5e9)) break;
    const double q = (double)(tiles * ns) / 256.0;
    const double rounds = q <= 3.0 ? (double)hwg_cdiv(tiles * ns, 256) : q + 0.5;
    double t = rounds * (wc.fixed_us + 4.0 * hwg_cdiv(chunks, ns) * wc.step_us) * 1e-6;
    if (ns > 1) t += (ns + 1) * out_bytes / 3.0e12 + 6e-6;
    if (t < best_t) { best_t = t; best.nsplit = ns; }
  }
  const int bal_mode = hwg_tune().wino_bal[0] ? atoi(hwg_tune().wino_bal) : 0;
  if (bal_mode >= 0) {
    WinoPlan p;
    wino_cfg(p, 6);
    int gf = 0, af = -1;
    if (bal_mode > 0) sscanf(hwg_tune().wino_bal, "%d,%d", &gf, &af);
    if (wino_balance(p, tiles, chunks, gf, af) && (p.bal_pieces <= 1 || (double)p.bal_pieces * out_bytes <= 1.5e9)) {
      const long long units = (tiles - p.bal_tile0) * chunks;
      const long long c = hwg_cdiv(units, (long long)p.bal);
      double t = (p.bal_tile0 / 256) * (wc.fixed_us + 4.0 * chunks * wc.step_us) * 1e-6;
      t += (double)hwg_cdiv((long long)p.bal, 256ll) * (wc.fixed_us - 4.0 + (p.bal_segs - 1) * 5.0 + 4.0 * c * wc.step_us) * 1e-6;
      if (p.bal_cut) t += (p.bal_cut_pieces + p.bal_cut) * (out_bytes / tiles) / 3.0e12 + out_bytes / 2.0e13 + 6e-6;
      if (bal_mode > 0 || t < 0.97 * best_t) { best = p; best_t = t; }
    }
  }
  if (const char* f = hwg_tune().wino_force; *f) {   // "cfg[,nsplit]": only the split applies here
    int fc = -1, fs = 0;
    if (sscanf(f, "%d,%d", &fc, &fs) >= 2 && fs >= 1) { const int keep = fs > chunks ? chunks : fs; wino_cfg(best, 6); best.nsplit = keep; }
  }
  best.model_s = best_t;
  return best;
}
WinoPlan plan_wino_s2(const hwg_conv_desc* d) {
  static thread_local HwgPlanCache<WinoPlan> cache;
  return cache.get(d, plan_wino_s2_model);
}

__global__ __launch_bounds__(256) void wino_s2_pack_kernel(const float* src, float* dst, int Kc, int Cc, long long sk, long long sc, int dgrad, long long total) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    wino_s2_pack_one(src, dst, i, Kc, Cc, sk, sc, dgrad);
}

}  // namespace

extern "C" int hwg_wino_s2_supported(const hwg_conv_desc* d) {
  S2Geom g;
  return hwg_tune().wino_s2 != 0 && !hwg_tune().w64_nodma && s2_geom(d, g) ? 1 : 0;
}

extern "C" int hwg_wino_s2_preferred(const hwg_conv_desc* d) {
  if (!hwg_wino_s2_supported(d)) return 0;
  if (hwg_tune().wino_s2 == 2) return 1;
  const WinoPlan p = plan_wino_s2(d);
  return p.model_s < 0.9 * hwg_conv_direct_model_seconds(d) ? 1 : 0;
}

extern "C" size_t hwg_wino_s2_weight_floats(int Kc, int Cc, int dgrad) {
  const size_t A = dgrad ? 4 * (size_t)Cc : (size_t)Kc, B = dgrad ? (size_t)Kc : 4 * (size_t)Cc;
  return 16 * ((A + 15) / 16 * 16) * ((B + 15) / 16 * 16);
}

extern "C" int hwg_wino_s2_pack_weight(const float* src, float* dst, int Kc, int Cc, long long sk, long long sc, int dgrad, void* stream) {
  HWG_REQUIRE(src && dst && Kc > 0 && Cc > 0, "wino_s2_pack_weight: bad arguments");
  const long long total = (long long)hwg_wino_s2_weight_floats(Kc, Cc, dgrad) / 16;
  hipLaunchKernelGGL(wino_s2_pack_kernel, dim3(hwg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, Kc, Cc, sk, sc, dgrad, total);
  HWG_LAUNCH_CHECK("wino_s2_pack_weight");
  return HWG_OK;
}

extern "C" size_t hwg_wino_s2_workspace(const hwg_conv_desc* d) {
  if (!hwg_wino_s2_supported(d)) return 0;
  const WinoPlan p = plan_wino_s2(d);
  const size_t out = (size_t)d->N * d->P * d->Q * d->K * sizeof(float);
  if (p.bal > 0) return p.bal_pieces > 1 ? p.bal_pieces * out : 0;
  return p.nsplit > 1 ? p.nsplit * out : 0;
}

extern "C" int hwg_wino_s2_conv(const hwg_conv_desc* d, const float* x, const float* u, const float* bias, float* y, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream) {
  HWG_REQUIRE(d && x && u && y, "wino_s2_conv: null pointer");
  S2Geom g;
  HWG_REQUIRE(hwg_wino_s2_supported(d) && s2_geom(d, g), "wino_s2_conv: needs a 4x4 stride-2 pad-0 convolution (or its data gradient) with C %% 16 == 0");
  HWG_REQUIRE(!(d->transposed && bias), "wino_s2_conv: the data-gradient form takes no bias");
  hipStream_t st = (hipStream_t)stream;
  const WinoPlan p = plan_wino_s2(d);
  const size_t need = hwg_wino_s2_workspace(d);
  if (need && (!workspace || workspace_bytes < need)) {
    hwg_set_error("wino_s2_conv: workspace too small (%zu < %zu)", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  WinoK k;
  k.x = x; k.u = u; k.bias = bias; k.y = y;
  k.N = g.N; k.H = g.Hv; k.W = g.Wv; k.C = g.Cv; k.K = g.Kv; k.Kpad = (g.Kv + 15) / 16 * 16;
  k.P = g.Pv; k.Q = g.Qv; k.ph = g.pad; k.pw = g.pad;
  k.TP = hwg_cdiv(g.Pv, 3); k.TQ = hwg_cdiv(g.Qv, 3);
  k.M = g.N * k.TP * k.TQ;
  k.accumulate = accumulate;
  k.nsplit = p.nsplit;
  k.part = (float*)workspace;
  k.mt = hwg_cdiv(k.M, 64); k.nt = hwg_cdiv(g.Kv, 64);
  k.xcd_order = hwg_tune().wino_order;
  k.bal = p.bal; k.bal_tile0 = p.bal_tile0;
  k.mt_edge = 3; k.dbg = 0;
  k.x_img = g.x_img; k.x_row = g.x_row; k.x_pix = g.x_pix; k.x_tc = g.x_tc; k.x_run = g.x_run;
  k.y_img = g.y_img; k.y_row = g.y_row; k.y_pix = g.y_pix; k.y_kc = g.y_kc; k.y_run = g.y_run;
  dim3 grid(p.bal > 0 ? p.bal_tile0 + (p.bal + 7) / 8 * 8 : (k.mt * k.nt * p.nsplit + 7) / 8 * 8);
  k.cnt = nullptr;
  if ((p.bal > 0 ? p.bal_pieces > 1 : p.nsplit > 1) && hwg_tune().split_inkernel && (long long)k.mt * k.nt <= HWG_SPLIT_COUNTERS) {
    k.cnt = hwg_split_counters(st);
    if (!k.cnt) return HWG_ERR_LAUNCH;
  }
  const int prof = hwg_prof_open(HWG_PROF_CONV_WINO, g.flops, st);
  if (p.bal > 0) hipLaunchKernelGGL((wino_conv64d_kernel<0, true, 3>), grid, dim3(512), 0, st, k);
  else hipLaunchKernelGGL((wino_conv64d_kernel<0, false, 3>), grid, dim3(512), 0, st, k);
  hwg_prof_close(prof, st);
  hwg_note_plan(HWG_PROF_CONV_WINO, 36, p.bal > 0 ? -p.bal : p.nsplit);      // (schedule id 36: F(3x3,2x2) on the 64 x 64 DMA kernel)
  HWG_LAUNCH_CHECK("wino_s2_conv");
  const long long total = (long long)d->N * d->P * d->Q * d->K;
  if (k.cnt) {
    // (partial images summed inside the kernel)
  } else if (p.bal > 0 && p.bal_pieces > 1) {
    const double cut = (double)(k.mt * k.nt - p.bal_tile0) / (k.mt * k.nt);
    const int prof2 = hwg_prof_open(HWG_PROF_CONV_REDUCE, 4.0 * total * cut * (p.bal_pieces + 1), st);
    hipLaunchKernelGGL(wino_bal_reduce_kernel<4>, dim3(hwg_stream_grid(total / 4, 256)), dim3(256), 0, st, k, g.Cv / 16);
    hwg_prof_close(prof2, st);
    HWG_LAUNCH_CHECK("wino_bal_reduce");
  } else if (p.bal <= 0 && p.nsplit > 1) {
    const int prof2 = hwg_prof_open(HWG_PROF_CONV_REDUCE, 4.0 * total * (p.nsplit + 1), st);
    const int rc = hwg_conv_split_reduce_launch((const float*)workspace, bias, y, total, d->K, p.nsplit, accumulate, st);
    hwg_prof_close(prof2, st);
    if (rc) return rc;
  }
  return HWG_OK;
}

/* the schedule hwg_wino_conv_fwd (3x3 stride 1) / hwg_wino_s2_conv (4x4 stride 2 pad 0, either direction) would run for this product, without launching
 * anything: out[8] = {tile config, uniform channel split, balanced tail workgroups (0 = uniform schedule), whole-tile lead workgroups, most
 * pieces a cut tile is written in, workgroup tiles, channel chunks, most tiles one tail workgroup touches}; -1s when the product is not supported */
extern "C" int hwg_wino_conv_describe(const hwg_conv_desc* d, int* out) {
  HWG_REQUIRE(d && out, "wino_conv_describe: null pointer");
  for (int i = 0; i < 8; ++i) out[i] = -1;
  WinoPlan p;
  long long tiles;
  int chunks;
  S2Geom g;
  if (d->R == 4 && d->S == 4) {
    if (!hwg_wino_s2_supported(d) || !s2_geom(d, g)) return HWG_OK;
    p = plan_wino_s2(d);
    tiles = (long long)hwg_cdiv((long long)g.N * hwg_cdiv(g.Pv, 3) * hwg_cdiv(g.Qv, 3), 64ll) * hwg_cdiv(g.Kv, 64);
    chunks = g.Cv / 16;
  } else {
    if (!hwg_wino_supported(d)) return HWG_OK;
    p = plan_wino(d);
    tiles = (long long)hwg_cdiv((long long)d->N * hwg_cdiv(d->P, 2) * hwg_cdiv(d->Q, 2), (long long)p.tm) * hwg_cdiv(d->K, p.tn);
    chunks = d->C / 16;
  }
  out[0] = p.cfg; out[1] = p.nsplit; out[2] = p.bal; out[3] = p.bal_tile0; out[4] = p.bal > 0 ? p.bal_pieces : (p.nsplit > 1 ? p.nsplit : 1);
  out[5] = (int)tiles; out[6] = chunks; out[7] = p.bal > 0 ? p.bal_segs : 1;
  return HWG_OK;
}

