// Grouped ("one expert per window") 1-D convolutions / linears for the 79 character-style experts
// (model/char_style.py:84-124, 210-235). The reference runs one tiny network per recognised character in a Python loop;
// here all windows of all characters go through each layer in ONE launch. Windows are sorted by expert, so every run of windows
// is a small GEMM against that expert's weights (read through a device pointer table):
//   forward        Y[rows, Cout]   = Xunf[rows, Cin*S] * W^T      rows = windows of the run x R positions
//   data gradient  dX[rows, Cin]   = dYunf[rows, Cout*S] * W
//   weight grad    dW[Cout, Cin*S] += dY^T[Cout, rows] * Xunf
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32). The layers are bound by streaming the present experts' weights once
// (1.45 MB per expert), not by FLOPs; the work list is (run, GT_ROWS-row tile) so that one frequent character does not serialise a launch.
//
// Layouts: x [n][R][Cin], y [n][R][Cout] (R positions, channels fastest); weights in the PyTorch Conv1d layout [Cout][Cin][S]
// (nn.Linear [Cout][Cin] is S = 1, R = 1). seg_start[G+1]/seg_eid[G] describe the runs, tile_seg/tile_row0 the row tiles.
#include "hwg_common.h"

namespace {

constexpr int MAXR = 8;      // positions per window (5 for window=2)
constexpr int GT_ROWS = 32;  // rows per work tile (MT 32-row MFMA tiles): the experts see ~90 rows each, small tiles = more workgroups streaming weights
constexpr int MT = GT_ROWS / 32;
constexpr int GT_CK = 256;   // channels of the LDS-staged operand per step (the experts' 128 / 256-channel layers: one step, 68 KB of LDS)

// ---- forward: D[m = row][n = co]; A = shifted x rows from LDS, B = W rows straight from global (16-byte loads along (ci,s)) ----
// block = 8 waves: wave (wk, wq) owns output channels [co_blk + 32 wq, +32) of the work tile and half wk of every staged K range (the two
// halves' sums meet in LDS at the end): two wavefronts per SIMD, so one's weight / LDS latencies hide behind the other's MFMAs. Pays on long
// contractions (Cin*S >= 768: 31 -> 28 us); shorter ones are launched with 4 waves and no split (28 -> 33 us with it). tools/expert_probe.py
template <int S>
__global__ __launch_bounds__(512) void gmm_fwd_kernel(const float* __restrict__ x, const int* seg_start, const int* seg_eid, const int* tile_seg,
                                                      const int* tile_row0, const long long* wptr, const long long* bptr, float* __restrict__ y,
                                                      int R, int Cin, int Cout, int pad) {
  constexpr int LD = GT_CK + 1;
  __shared__ float xs[(GT_ROWS + 2 * (S / 2)) * LD];
  const int g = tile_seg[blockIdx.x], rc0 = tile_row0[blockIdx.x];
  const int i0 = seg_start[g];
  const int nrows = (seg_start[g + 1] - i0) * R;
  const int e = seg_eid[g];
  const float* __restrict__ W = reinterpret_cast<const float*>(wptr[e]);
  const float* bias = bptr ? reinterpret_cast<const float*>(bptr[e]) : nullptr;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int wk = wid >> 2, wq = wid & 3;
  const int co = blockIdx.y * 128 + wq * 32 + l31;
  const bool co_ok = co < Cout;
  const long long xbase = (long long)i0 * R * Cin;
  const int KS = Cin * S;
  // k entries of a weight row are (ci, s) pairs, ci major. A group = 8 channels = S pieces of 8 entries; lane (col, half) takes entries
  // 8u + 4 half + j (u < S, j < 4) of every group: their channel offset / tap never change, so the LDS offset and the window mask of the A
  // operand are per-lane constants. The A reads are unconditional (the staged tile is zero-padded; the mask only keeps a tap from reaching
  // into the neighbouring window) - a predicated read per MFMA cannot be moved off the MFMA's critical path by the compiler.
  int xoff[MT][S][4];
  float xmask[MT][S][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int pm = (rc0 + mt * 32 + l31) % R;   // position inside its window of the row this lane feeds as A operand
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int ke = 8 * u + 4 * lhi + jj;
        const int cl = ke / S, sft = ke - cl * S;
        const int q = pm + sft - pad;
        xmask[mt][u][jj] = (q >= 0 && q < R) ? 1.f : 0.f;
        xoff[mt][u][jj] = (mt * 32 + l31 + sft) * LD + cl;
      }
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[mt][t] = 0.f;
  // The weights come straight from global memory (every lane streams its own output channel's row): NP 16-byte pieces per lane are in
  // flight at a time, and the next batch's loads are issued before this batch's MFMAs.
  constexpr int GP = S == 1 ? 8 : 2;   // groups per batch
  constexpr int NP = GP * S;

  for (int ci0 = 0; ci0 < Cin; ci0 += GT_CK) {
    const int ngroups = min(GT_CK, Cin - ci0) / 8;
    const int ghalf = blockDim.x == 512 ? (ngroups + 1) / 2 : ngroups;     // launched with 256 threads: one wavefront per 32 channels, no K split
    const int gbeg = wk ? ghalf : 0, gend = wk ? ngroups : ghalf;     // this wavefront's groups
    const float* wr = W + (long long)(co_ok ? co : 0) * KS + ci0 * S + 4 * lhi;
    auto load_w = [&](int g0, float4 (&w)[NP]) {
#pragma unroll
      for (int gi = 0; gi < GP; ++gi)
#pragma unroll
        for (int u = 0; u < S; ++u)
          w[gi * S + u] = (co_ok && g0 + gi < gend) ? *reinterpret_cast<const float4*>(wr + 8 * (S * (g0 + gi) + u)) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 wa[NP], wb[NP];
    load_w(gbeg, wa);              // in flight while the x tile is staged
    __syncthreads();
    // stage rows [rc0 - pad, rc0 + GT_ROWS + pad) x channels [ci0, ci0 + GT_CK)
    const int cw4 = (min(GT_CK, Cin - ci0) + 3) / 4;      // 16-byte column groups of this step
    for (int t = threadIdx.x; t < (GT_ROWS + 2 * pad) * cw4; t += blockDim.x) {
      const int rl = t / cw4, c4 = t - rl * cw4;
      const int row = rc0 - pad + rl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row >= 0 && row < nrows && ci0 + c4 * 4 < Cin) v = *reinterpret_cast<const float4*>(x + xbase + (long long)row * Cin + ci0 + c4 * 4);
      float* d = xs + rl * LD + c4 * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // A operands of a batch (one LDS read per MFMA) are read a batch ahead as well: left to the compiler, every second MFMA waits for
    // its own ds_read (108 VGPRs used, read - s_waitcnt lgkmcnt(0) - multiply - MFMA in a row) and the matrix pipe idles half the time.
    auto load_a = [&](int g0, float (&a)[GP][S][4][MT]) {
#pragma unroll
      for (int gi = 0; gi < GP; ++gi) {
        const float* xg = xs + 8 * min(g0 + gi, ngroups - 1);
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[gi][u][jj][mt] = xg[xoff[mt][u][jj]] * xmask[mt][u][jj];
      }
    };
    auto run_w = [&](int g0, const float4 (&w)[NP], const float (&a)[GP][S][4][MT]) {
#pragma unroll
      for (int gi = 0; gi < GP; ++gi) {
        if (g0 + gi >= gend) break;
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const float4 wv = w[gi * S + u];
            const float b = jj == 0 ? wv.x : jj == 1 ? wv.y : jj == 2 ? wv.z : wv.w;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[gi][u][jj][mt], b, acc[mt], 0, 0, 0);
          }
      }
    };
    float aa[GP][S][4][MT], ab[GP][S][4][MT];
    load_a(gbeg, aa);
    for (int g = gbeg; g < gend; g += 2 * GP) {
      load_w(g + GP, wb);
      load_a(g + GP, ab);
      __builtin_amdgcn_sched_barrier(0);
      run_w(g, wa, aa);
      __builtin_amdgcn_sched_barrier(0);
      load_w(g + 2 * GP, wa);
      load_a(g + 2 * GP, aa);
      __builtin_amdgcn_sched_barrier(0);
      run_w(g + GP, wb, ab);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the second K half hands its sums to the first through LDS (the tile buffer is free now)
  if (blockDim.x == 512) {
  __syncthreads();
  if (wk) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 16; ++t) xs[((wq * MT + mt) * 16 + t) * 64 + lane] = acc[mt][t];
  }
  __syncthreads();
  if (wk || !co_ok) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[mt][t] += xs[((wq * MT + mt) * 16 + t) * 64 + lane];
  }
  if (!co_ok) return;
  const float bv = bias ? bias[co] : 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = rc0 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
      if (row < nrows) y[((long long)i0 * R + row) * Cout + co] = acc[mt][t] + bv;
    }
}

// ---- data gradient: D[m = row][n = ci]; A = shifted dy rows from LDS, B = W[co][ci][0..S) (one S-float load per lane) ----
template <int S>
__global__ __launch_bounds__(512) void gmm_dgrad_kernel(const float* __restrict__ dy, const int* seg_start, const int* seg_eid, const int* tile_seg,
                                                        const int* tile_row0, const long long* wptr, float* __restrict__ dx, int R, int Cin,
                                                        int Cout, int pad) {
  constexpr int LD = GT_CK + 1;
  __shared__ float dys[(GT_ROWS + 2 * (S / 2)) * LD];
  const int g = tile_seg[blockIdx.x], rc0 = tile_row0[blockIdx.x];
  const int i0 = seg_start[g];
  const int nrows = (seg_start[g + 1] - i0) * R;
  const float* __restrict__ W = reinterpret_cast<const float*>(wptr[seg_eid[g]]);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int wk = wid >> 2, wq = wid & 3;     // K-split wavefront pairs as in gmm_fwd_kernel
  const int ci = blockIdx.y * 128 + wq * 32 + l31;
  const bool ci_ok = ci < Cin;
  const long long ybase = (long long)i0 * R * Cout;
  // offsets / window masks of the A operand per (row tile, tap): unconditional LDS reads, see gmm_fwd_kernel
  int doff[MT][S];
  float dmask[MT][S];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int pm = (rc0 + mt * 32 + l31) % R;
#pragma unroll
    for (int sft = 0; sft < S; ++sft) {
      const int q = pm - sft + pad;   // position of the output row this tap came from
      dmask[mt][sft] = (q >= 0 && q < R) ? 1.f : 0.f;
      doff[mt][sft] = (mt * 32 + l31 - sft + 2 * pad) * LD;
    }
  }
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[mt][t] = 0.f;

  for (int co0 = 0; co0 < Cout; co0 += GT_CK) {
    const int cn = min(GT_CK, Cout - co0);
    const int chalf = ((cn + 1) / 2 + 1) & ~1;
    const int cbeg = wk ? chalf : 0, cend = wk ? cn : chalf;     // this wavefront's output channels of the contraction
    // DU k pairs (2 output channels each) of weights in flight, the next group's loads issued before this group's MFMAs (see gmm_fwd_kernel)
    constexpr int DU = 8;
    auto load_w = [&](int cp0, float (&w)[DU][S]) {
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int col = cp0 + 2 * u + lhi;  // this half's output channel of the k pair
        const bool ok = ci_ok && col < cend;
        const float* wr = W + ((long long)(co0 + (ok ? col : 0)) * Cin + (ci_ok ? ci : 0)) * S;
#pragma unroll
        for (int sft = 0; sft < S; ++sft) w[u][sft] = ok ? wr[sft] : 0.f;
      }
    };
    float wa[DU][S], wb[DU][S];
    load_w(cbeg, wa);              // in flight while the dy tile is staged
    __syncthreads();
    const int cw4 = (cn + 3) / 4;                          // 16-byte column groups of this step
    for (int t = threadIdx.x; t < (GT_ROWS + 2 * pad) * cw4; t += 512) {
      const int rl = t / cw4, c4 = t - rl * cw4;
      const int row = rc0 - pad + rl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row >= 0 && row < nrows && co0 + c4 * 4 < Cout) v = *reinterpret_cast<const float4*>(dy + ybase + (long long)row * Cout + co0 + c4 * 4);
      float* d = dys + rl * LD + c4 * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    auto load_a = [&](int cp0, float (&a)[DU][S][MT]) {      // the A operands of a group, read a group ahead (see gmm_fwd_kernel)
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int col = min(cp0 + 2 * u, cn - 2) + lhi;
#pragma unroll
        for (int sft = 0; sft < S; ++sft)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) a[u][sft][mt] = dys[doff[mt][sft] + col] * dmask[mt][sft];
      }
    };
    auto run_w = [&](int cp0, const float (&w)[DU][S], const float (&a)[DU][S][MT]) {
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        if (cp0 + 2 * u >= cend) break;
#pragma unroll
        for (int sft = 0; sft < S; ++sft)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][sft][mt], w[u][sft], acc[mt], 0, 0, 0);
      }
    };
    float aa[DU][S][MT], ab[DU][S][MT];
    load_a(cbeg, aa);
    for (int cp = cbeg; cp < cend; cp += 4 * DU) {
      load_w(cp + 2 * DU, wb);
      load_a(cp + 2 * DU, ab);
      __builtin_amdgcn_sched_barrier(0);
      run_w(cp, wa, aa);
      __builtin_amdgcn_sched_barrier(0);
      load_w(cp + 4 * DU, wa);
      load_a(cp + 4 * DU, aa);
      __builtin_amdgcn_sched_barrier(0);
      run_w(cp + 2 * DU, wb, ab);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  if (wk) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < 16; ++t) dys[((wq * MT + mt) * 16 + t) * 64 + lane] = acc[mt][t];
  }
  __syncthreads();
  if (wk || !ci_ok) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[mt][t] += dys[((wq * MT + mt) * 16 + t) * 64 + lane];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = rc0 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
      if (row < nrows) dx[((long long)i0 * R + row) * Cin + ci] = acc[mt][t];
    }
}

// ---- weight gradient: D[m = co][n = ci] per tap, contraction over rows; both operands coalesced from global ----
// Work item = (row tile of at most tile_rows rows of one run, 32 output channels, 256 input channels); wave w owns input channels
// [ci_blk + 32 w, +32) for all S taps. A run of several tiles (a very frequent character, spread over many workgroups) writes one
// partial dW (and db) image per tile to the workspace and gmm_wgrad_reduce_kernel adds them, in tile order, to the expert's gradient buffer
// (deterministic); the only tile of a run adds its block to the gradient buffer directly.
template <int S>
__global__ __launch_bounds__(512) void gmm_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, const int* seg_start,
                                                        const int* tile_seg, const int* tile_row0, int tile_rows, float* __restrict__ part,
                                                        const int* run_tile0, const int* seg_eid, const long long* gwptr, const long long* gbptr,
                                                        int R, int Cin, int Cout, int pad) {
  const int tile = blockIdx.x;
  const int g = tile_seg[tile], rc0 = tile_row0[tile];
  const int i0 = seg_start[g];
  const int nrows = (seg_start[g + 1] - i0) * R;
  const int rend = min(rc0 + tile_rows, nrows);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int co = blockIdx.y * 32 + l31;         // A operand column of this lane
  const int ci = blockIdx.z * 256 + wid * 32 + l31;
  const bool ci_ok = ci < Cin, co_ok = co < Cout;   // ragged channel blocks: operands outside are fed as zeros, results outside are not stored
  if (blockIdx.z * 256 + wid * 32 >= Cin) return;   // whole wave outside (no barriers in this kernel)
  const float* dyb = dy + (long long)i0 * R * Cout + (co_ok ? co : 0);
  const float* xb = x + (long long)i0 * R * Cin + (ci_ok ? ci : 0);
  f32x16 acc[S];
#pragma unroll
  for (int sft = 0; sft < S; ++sft)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[sft][t] = 0.f;
  float bsum = 0.f;
  // A tile is ~90 rows (one expert's windows): the whole row loop is a handful of load batches, so its time is (batches x memory latency)
  // unless the next batch is already in flight: U row pairs per batch, two batches in registers.
  constexpr int U = 8;
  auto load_rows = [&](int r0, float (&av)[U], float (&bv)[U][S]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rw = r0 + lhi + 2 * u;          // this half's row of k pair u
      const int pu = rw % R;                    // its position inside the window
      // unconditional loads from clamped rows, masked afterwards: predicated loads would each sit in their own exec-mask region
      const bool rok = rw < rend;
      const int rwc = min(rw, rend - 1);
      av[u] = dyb[(long long)rwc * Cout] * ((rok && co_ok) ? 1.f : 0.f);
#pragma unroll
      for (int sft = 0; sft < S; ++sft) {
        const int q = pu + sft - pad;
        const int rx = min(max(rwc + sft - pad, 0), nrows - 1);
        bv[u][sft] = xb[(long long)rx * Cin] * ((rok && ci_ok && q >= 0 && q < R) ? 1.f : 0.f);
      }
    }
  };
  auto run_rows = [&](const float (&av)[U], const float (&bv)[U][S]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bsum += av[u];
#pragma unroll
      for (int sft = 0; sft < S; ++sft) acc[sft] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][sft], acc[sft], 0, 0, 0);
    }
  };
  float a0[U], b0[U][S], a1[U], b1[U][S];
  load_rows(rc0, a0, b0);
  for (int r2 = rc0; r2 < rend; r2 += 4 * U) {
    load_rows(r2 + 2 * U, a1, b1);             // rows past the tile are masked to zero
    run_rows(a0, b0);
    if (r2 + 2 * U >= rend) break;
    load_rows(r2 + 4 * U, a0, b0);
    run_rows(a1, b1);
  }
  if (run_tile0[g + 1] - run_tile0[g] == 1) {
    // the run's only tile (most experts see fewer rows than one tile): no other workgroup adds to this (expert, co block, ci block), so the
    // block goes straight into the expert's gradient - no partial image (Cout*Cin*S floats written and read again per tile) and nothing
    // for the reduce pass to do. Same value as the reduce pass would add: 0 + this tile's sums.
    const int e = seg_eid[g];
    if (ci_ok) {
      float* dW = reinterpret_cast<float*>(gwptr[e]);
#pragma unroll
      for (int sft = 0; sft < S; ++sft)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int cr = blockIdx.y * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
          if (cr < Cout) dW[((long long)cr * Cin + ci) * S + sft] += acc[sft][t];
        }
    }
    if (blockIdx.z == 0 && wid == 0 && gbptr) {
      const float tot = bsum + __shfl_xor(bsum, 32, 64);
      if (lhi == 0 && co_ok) reinterpret_cast<float*>(gbptr[e])[co] += tot;
    }
    return;
  }
  const long long img = (long long)Cout * Cin * S + Cout;   // partial image: dW then db
  float* pw = part + (long long)tile * img;
  if (ci_ok) {
#pragma unroll
    for (int sft = 0; sft < S; ++sft)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int cr = blockIdx.y * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
        if (cr < Cout) pw[((long long)sft * Cout + cr) * Cin + ci] = acc[sft][t];   // tap-major partial image: 128-byte runs per store
      }
  }
  if (blockIdx.z == 0 && wid == 0) {
    const float tot = bsum + __shfl_xor(bsum, 32, 64);
    if (lhi == 0 && co_ok) pw[(long long)Cout * Cin * S + co] = tot;
  }
}
// dW[e(g)] += sum_{tiles of run g} part[tile]   (and db). The partial images are tap-major ([S][Cout][Cin], so that the weight-gradient
// kernel stores whole 128-byte runs instead of one float every S); 256 (co, ci) pairs per workgroup are summed tile by tile (fixed order)
// and leave through LDS as 256*S consecutive floats of the expert's [Cout][Cin][S] gradient.
__global__ __launch_bounds__(256) void gmm_wgrad_reduce_kernel(const float* __restrict__ part, const int* run_tile0, const int* seg_eid,
                                                               const long long* gwptr, const long long* gbptr, int Cout, int Cin, int S,
                                                               int main_blocks) {
  extern __shared__ float red[];          // [256][S | 1]
  const int ldp = S | 1;
  const int g = blockIdx.y;
  const int t0 = run_tile0[g], t1 = run_tile0[g + 1];
  if (t1 - t0 == 1) return;               // single-tile runs were added by the weight-gradient kernel itself
  const int e = seg_eid[g];
  const long long pairs = (long long)Cout * Cin;
  const long long img = pairs * S + Cout;
  if ((int)blockIdx.x >= main_blocks) {
    float* db = gbptr ? reinterpret_cast<float*>(gbptr[e]) : nullptr;
    const int k = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (!db || k >= Cout) return;
    float sacc = 0.f;
    for (int t = t0; t < t1; ++t) sacc += part[(long long)t * img + pairs * S + k];
    db[k] += sacc;
    return;
  }
  float* dW = reinterpret_cast<float*>(gwptr[e]);
  const long long base = (long long)blockIdx.x * 256;
  const long long pr = base + threadIdx.x;
  if (pr < pairs)
    for (int sft = 0; sft < S; ++sft) {
      float sacc = 0.f;
      for (int t = t0; t < t1; ++t) sacc += part[(long long)t * img + sft * pairs + pr];
      red[threadIdx.x * ldp + sft] = sacc;
    }
  __syncthreads();
  const int n_out = (int)min(256LL, pairs - base) * S;
  for (int q = threadIdx.x; q < n_out; q += 256) {
    const int pl = q / S;
    dW[base * S + q] += red[pl * ldp + (q - pl * S)];
  }
}

// out[i][c] = (*ptrs[eid[i]])[c]
__global__ void gather_rows_ptr_kernel(const long long* ptrs, const int* eid, float* out, int n, int C) {
  const long long total = (long long)n * C;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), c = (int)(t % C);
    out[t] = reinterpret_cast<const float*>(ptrs[eid[i]])[c];
  }
}
// (*gptrs[e])[c] += sum_{i in run} rows[i][c]
__global__ void segment_accumulate_ptr_kernel(const float* rows, const int* seg_start, const int* seg_eid, const long long* gptrs, int C) {
  const int g = blockIdx.x;
  float* dst = reinterpret_cast<float*>(gptrs[seg_eid[g]]);
  const int i0 = seg_start[g], i1 = seg_start[g + 1];
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    int i = i0;
    for (; i + 8 <= i1; i += 8) {         // eight independent loads in flight, added in row order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = rows[(long long)(i + u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; i < i1; ++i) s += rows[(long long)i * C + c];
    dst[c] += s;
  }
}

}  // namespace

extern "C" int hwg_grouped_conv1d_fwd(const float* x, const int* seg_start, const int* seg_eid, const int* tile_seg, const int* tile_row0, int ntiles,
                                      const void* wptr, const void* bptr, float* y, int R, int Cin, int Cout, int S, int pad, void* stream) {
  HWG_REQUIRE(x && seg_start && seg_eid && tile_seg && tile_row0 && wptr && y && ntiles > 0 && R > 0 && R <= MAXR && Cin > 0 && Cout > 0,
              "grouped_conv1d_fwd: bad arguments");
  HWG_REQUIRE((S == 1 && pad == 0) || (S == 3 && pad == 1), "grouped_conv1d_fwd: only S=1/pad=0 and S=3/pad=1 are built (got S=%d pad=%d)", S, pad);
  HWG_REQUIRE(Cin % 8 == 0, "grouped_conv1d_fwd: Cin must be a multiple of 8 (got %d)", Cin);
  dim3 grid(ntiles, hwg_cdiv(Cout, 128));
  hipStream_t st = (hipStream_t)stream;
  const dim3 blk(Cin * S >= 768 ? 512 : 256);      // K-split wavefront pairs on long contractions only
  if (S == 1)
    hipLaunchKernelGGL(gmm_fwd_kernel<1>, grid, blk, 0, st, x, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, (const long long*)bptr, y,
                       R, Cin, Cout, pad);
  else
    hipLaunchKernelGGL(gmm_fwd_kernel<3>, grid, blk, 0, st, x, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, (const long long*)bptr, y,
                       R, Cin, Cout, pad);
  HWG_LAUNCH_CHECK("grouped_conv1d_fwd");
  return HWG_OK;
}
extern "C" int hwg_grouped_conv1d_dgrad(const float* dy, const int* seg_start, const int* seg_eid, const int* tile_seg, const int* tile_row0, int ntiles,
                                        const void* wptr, float* dx, int R, int Cin, int Cout, int S, int pad, void* stream) {
  HWG_REQUIRE(dy && seg_start && seg_eid && tile_seg && tile_row0 && wptr && dx && ntiles > 0 && R > 0 && R <= MAXR && Cin > 0 && Cout > 0,
              "grouped_conv1d_dgrad: bad arguments");
  HWG_REQUIRE((S == 1 && pad == 0) || (S == 3 && pad == 1), "grouped_conv1d_dgrad: only S=1/pad=0 and S=3/pad=1 are built (got S=%d pad=%d)", S, pad);
  HWG_REQUIRE(Cout % 4 == 0, "grouped_conv1d_dgrad: Cout must be a multiple of 4 (got %d)", Cout);
  dim3 grid(ntiles, hwg_cdiv(Cin, 128));
  hipStream_t st = (hipStream_t)stream;
  if (S == 1)
    hipLaunchKernelGGL(gmm_dgrad_kernel<1>, grid, dim3(512), 0, st, dy, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, dx, R, Cin, Cout, pad);
  else
    hipLaunchKernelGGL(gmm_dgrad_kernel<3>, grid, dim3(512), 0, st, dy, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, dx, R, Cin, Cout, pad);
  HWG_LAUNCH_CHECK("grouped_conv1d_dgrad");
  return HWG_OK;
}
extern "C" size_t hwg_grouped_conv1d_wgrad_workspace(int ntiles, int Cin, int Cout, int S) {
  return (size_t)(ntiles > 0 ? ntiles : 0) * ((size_t)Cout * Cin * S + Cout) * sizeof(float);
}
extern "C" int hwg_grouped_conv1d_wgrad(const float* dy, const float* x, const int* seg_start, const int* seg_eid, int G, const int* tile_seg,
                                        const int* tile_row0, const int* run_tile0, int ntiles, int tile_rows, const void* gwptr, const void* gbptr, int R,
                                        int Cin, int Cout, int S, int pad, void* workspace, size_t workspace_bytes, void* stream) {
  HWG_REQUIRE(dy && x && seg_start && seg_eid && tile_seg && tile_row0 && run_tile0 && gwptr && G > 0 && ntiles >= G && tile_rows > 0 && R > 0 && R <= MAXR && Cin > 0 &&
                  Cout > 0, "grouped_conv1d_wgrad: bad arguments");
  HWG_REQUIRE((S == 1 && pad == 0) || (S == 3 && pad == 1), "grouped_conv1d_wgrad: only S=1/pad=0 and S=3/pad=1 are built (got S=%d pad=%d)", S, pad);
  const size_t need = hwg_grouped_conv1d_wgrad_workspace(ntiles, Cin, Cout, S);
  if (!workspace || workspace_bytes < need) {
    hwg_set_error("grouped_conv1d_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return HWG_ERR_WORKSPACE;
  }
  dim3 grid(ntiles, hwg_cdiv(Cout, 32), hwg_cdiv(Cin, 256));
  hipStream_t st = (hipStream_t)stream;
  if (S == 1)
    hipLaunchKernelGGL(gmm_wgrad_kernel<1>, grid, dim3(512), 0, st, dy, x, seg_start, tile_seg, tile_row0, tile_rows, (float*)workspace, run_tile0, seg_eid,
                       (const long long*)gwptr, (const long long*)gbptr, R, Cin, Cout, pad);
  else
    hipLaunchKernelGGL(gmm_wgrad_kernel<3>, grid, dim3(512), 0, st, dy, x, seg_start, tile_seg, tile_row0, tile_rows, (float*)workspace, run_tile0, seg_eid,
                       (const long long*)gwptr, (const long long*)gbptr, R, Cin, Cout, pad);
  HWG_LAUNCH_CHECK("grouped_conv1d_wgrad");
  const int main_blocks = hwg_cdiv((long long)Cout * Cin, 256);
  hipLaunchKernelGGL(gmm_wgrad_reduce_kernel, dim3(main_blocks + hwg_cdiv(Cout, 256), G), dim3(256), (size_t)256 * (S | 1) * sizeof(float), st,
                     (const float*)workspace, run_tile0, seg_eid, (const long long*)gwptr, (const long long*)gbptr, Cout, Cin, S, main_blocks);
  HWG_LAUNCH_CHECK("grouped_conv1d_wgrad_reduce");
  return HWG_OK;
}
extern "C" int hwg_gather_rows_ptr(const void* ptrs, const int* eid, float* out, int n, int C, void* stream) {
  HWG_REQUIRE(ptrs && eid && out && n > 0 && C > 0, "gather_rows_ptr: bad arguments");
  hipLaunchKernelGGL(gather_rows_ptr_kernel, dim3(hwg_stream_grid((long long)n * C, 256)), dim3(256), 0, (hipStream_t)stream, (const long long*)ptrs, eid,
                     out, n, C);
  HWG_LAUNCH_CHECK("gather_rows_ptr");
  return HWG_OK;
}
extern "C" int hwg_segment_accumulate_ptr(const float* rows, const int* seg_start, const int* seg_eid, int G, const void* gptrs, int C, void* stream) {
  HWG_REQUIRE(rows && seg_start && seg_eid && gptrs && G > 0 && C > 0, "segment_accumulate_ptr: bad arguments");
  hipLaunchKernelGGL(segment_accumulate_ptr_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, rows, seg_start, seg_eid, (const long long*)gptrs, C);
  HWG_LAUNCH_CHECK("segment_accumulate_ptr");
  return HWG_OK;
}
