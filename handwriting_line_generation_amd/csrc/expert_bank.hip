// Grouped ("one expert per window") 1-D convolutions / linears for the 79 character-style experts
// (model/char_style.py:84-124, 210-235). The reference runs one tiny network per recognised character in a Python loop;
// here all windows of all characters go through each layer in ONE launch. Windows are sorted by expert, so every run of windows
// is a small GEMM against that expert's weights (read through a device pointer table):
//   forward        Y[rows, Cout]   = Xunf[rows, Cin*S] * W^T      rows = windows of the run x R positions
//   data gradient  dX[rows, Cin]   = dYunf[rows, Cout*S] * W
//   weight grad    dW[Cout, Cin*S] += dY^T[Cout, rows] * Xunf
// on the fp32 matrix cores (v_mfma_f32_32x32x2_f32). The layers are bound by streaming the present experts' weights once
// (1.45 MB per expert), not by FLOPs; the work list is (run, 64-row tile) so that one frequent character does not serialise a launch.
//
// Layouts: x [n][R][Cin], y [n][R][Cout] (R positions, channels fastest); weights in the PyTorch Conv1d layout [Cout][Cin][S]
// (nn.Linear [Cout][Cin] is S = 1, R = 1). seg_start[G+1]/seg_eid[G] describe the runs, tile_seg/tile_row0 the row tiles.
#include "hwg_common.h"

namespace {

constexpr int MAXR = 8;      // positions per window (5 for window=2)
constexpr int GT_ROWS = 64;  // rows per work tile (two 32-row MFMA tiles)
constexpr int GT_CK = 64;    // channels of the LDS-staged operand per step

// ---- forward: D[m = row][n = co]; A = shifted x rows from LDS, B = W rows straight from global (16-byte loads along (ci,s)) ----
// block = 4 waves, wave w owns output channels [co_blk + 32 w, +32) and both 32-row tiles of the work tile.
template <int S>
__global__ __launch_bounds__(256) void gmm_fwd_kernel(const float* __restrict__ x, const int* seg_start, const int* seg_eid, const int* tile_seg,
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
  const int co = blockIdx.y * 128 + wid * 32 + l31;
  const bool co_ok = co < Cout;
  const long long xbase = (long long)i0 * R * Cin;
  const int KS = Cin * S;
  int pm[2];  // position inside its window of the row this lane feeds as A operand (per row tile)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) pm[mt] = (rc0 + mt * 32 + l31) % R;
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[mt][t] = 0.f;

  for (int ci0 = 0; ci0 < Cin; ci0 += GT_CK) {
    __syncthreads();
    // stage rows [rc0 - pad, rc0 + 64 + pad) x channels [ci0, ci0 + 64)
    for (int t = threadIdx.x; t < (GT_ROWS + 2 * pad) * (GT_CK / 4); t += 256) {
      const int rl = t / (GT_CK / 4), c4 = t % (GT_CK / 4);
      const int row = rc0 - pad + rl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row >= 0 && row < nrows && ci0 + c4 * 4 < Cin) v = *reinterpret_cast<const float4*>(x + xbase + (long long)row * Cin + ci0 + c4 * 4);
      float* d = xs + rl * LD + c4 * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int kchunk = min(GT_CK, Cin - ci0) * S;   // contiguous (ci,s) entries of this chunk in a weight row
    const float* wr = W + (long long)(co_ok ? co : 0) * KS + ci0 * S;
    for (int kg = 0; kg < kchunk; kg += 8) {
      // lane (col, half) takes 4 consecutive (ci,s) entries starting at kg + 4*half; entry j feeds MFMA j (same K permutation for A)
      float4 wv = *reinterpret_cast<const float4*>(wr + kg + 4 * lhi);
      if (!co_ok) wv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = kg + 4 * lhi + j;
        const int cl = kk / S, sft = kk - cl * S;
        const float b = j == 0 ? wv.x : j == 1 ? wv.y : j == 2 ? wv.z : wv.w;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int q = pm[mt] + sft - pad;
          const float av = (q >= 0 && q < R) ? xs[(mt * 32 + l31 + sft) * LD + cl] : 0.f;
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b, acc[mt], 0, 0, 0);
        }
      }
    }
  }
  if (!co_ok) return;
  const float bv = bias ? bias[co] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = rc0 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
      if (row < nrows) y[((long long)i0 * R + row) * Cout + co] = acc[mt][t] + bv;
    }
}

// ---- data gradient: D[m = row][n = ci]; A = shifted dy rows from LDS, B = W[co][ci][0..S) (one S-float load per lane) ----
template <int S>
__global__ __launch_bounds__(256) void gmm_dgrad_kernel(const float* __restrict__ dy, const int* seg_start, const int* seg_eid, const int* tile_seg,
                                                        const int* tile_row0, const long long* wptr, float* __restrict__ dx, int R, int Cin,
                                                        int Cout, int pad) {
  constexpr int LD = GT_CK + 1;
  __shared__ float dys[(GT_ROWS + 2 * (S / 2)) * LD];
  const int g = tile_seg[blockIdx.x], rc0 = tile_row0[blockIdx.x];
  const int i0 = seg_start[g];
  const int nrows = (seg_start[g + 1] - i0) * R;
  const float* __restrict__ W = reinterpret_cast<const float*>(wptr[seg_eid[g]]);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int ci = blockIdx.y * 128 + wid * 32 + l31;
  const bool ci_ok = ci < Cin;
  const long long ybase = (long long)i0 * R * Cout;
  int pm[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) pm[mt] = (rc0 + mt * 32 + l31) % R;
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[mt][t] = 0.f;

  for (int co0 = 0; co0 < Cout; co0 += GT_CK) {
    __syncthreads();
    for (int t = threadIdx.x; t < (GT_ROWS + 2 * pad) * (GT_CK / 4); t += 256) {
      const int rl = t / (GT_CK / 4), c4 = t % (GT_CK / 4);
      const int row = rc0 - pad + rl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row >= 0 && row < nrows && co0 + c4 * 4 < Cout) v = *reinterpret_cast<const float4*>(dy + ybase + (long long)row * Cout + co0 + c4 * 4);
      float* d = dys + rl * LD + c4 * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const int cn = min(GT_CK, Cout - co0);
    for (int cp = 0; cp < cn; cp += 2) {
      const int col = cp + lhi;  // this half's output channel of the k pair
      float wv[S];
      const float* wr = W + ((long long)(co0 + col) * Cin + (ci_ok ? ci : 0)) * S;
#pragma unroll
      for (int sft = 0; sft < S; ++sft) wv[sft] = (ci_ok && col < cn) ? wr[sft] : 0.f;
#pragma unroll
      for (int sft = 0; sft < S; ++sft) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int q = pm[mt] - sft + pad;   // position of the output row this tap came from
          const float av = (q >= 0 && q < R) ? dys[(mt * 32 + l31 - sft + 2 * pad) * LD + col] : 0.f;
          acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv[sft], acc[mt], 0, 0, 0);
        }
      }
    }
  }
  if (!ci_ok) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = rc0 + mt * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
      if (row < nrows) dx[((long long)i0 * R + row) * Cin + ci] = acc[mt][t];
    }
}

// ---- weight gradient: D[m = co][n = ci] per tap, contraction over rows; both operands coalesced from global ----
// Work item = (row tile of at most tile_rows rows of one run, 32 output channels, 256 input channels); wave w owns input channels
// [ci_blk + 32 w, +32) for all S taps. Every tile writes its partial dW (and db) image to the workspace; gmm_wgrad_reduce_kernel
// then adds the tiles of each run, in tile order, to the expert's gradient buffer, so a very frequent character is spread over
// many workgroups and the result is still deterministic.
template <int S>
__global__ __launch_bounds__(512) void gmm_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, const int* seg_start,
                                                        const int* tile_seg, const int* tile_row0, int tile_rows, float* __restrict__ part,
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
  const float* dyb = dy + (long long)i0 * R * Cout + co;
  const float* xb = x + (long long)i0 * R * Cin + (ci_ok ? ci : 0);
  f32x16 acc[S];
#pragma unroll
  for (int sft = 0; sft < S; ++sft)
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[sft][t] = 0.f;
  float bsum = 0.f;
  constexpr int U = 4;   // k pairs in flight: all loads of U pairs are issued before their MFMAs
  int row = rc0 + lhi, p = (rc0 + lhi) % R;   // this half's row of the current k pair and its position inside the window
  for (int r2 = rc0; r2 < rend; r2 += 2 * U) {
    float av[U], bv[U][S];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rw = row + 2 * u;
      int pu = p + 2 * u;
      pu %= R;
      const bool rok = rw < rend;
      av[u] = (rok && co_ok) ? dyb[(long long)rw * Cout] : 0.f;
#pragma unroll
      for (int sft = 0; sft < S; ++sft) {
        const int q = pu + sft - pad;
        bv[u][sft] = (rok && ci_ok && q >= 0 && q < R) ? xb[(long long)(rw + sft - pad) * Cin] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bsum += av[u];
#pragma unroll
      for (int sft = 0; sft < S; ++sft) acc[sft] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][sft], acc[sft], 0, 0, 0);
    }
    row += 2 * U;
    p = (p + 2 * U) % R;
  }
  const long long img = (long long)Cout * Cin * S + Cout;   // partial image: dW then db
  float* pw = part + (long long)tile * img;
  if (ci_ok) {
#pragma unroll
    for (int sft = 0; sft < S; ++sft)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int cr = blockIdx.y * 32 + (t & 3) + 8 * (t >> 2) + 4 * lhi;
        if (cr < Cout) pw[((long long)cr * Cin + ci) * S + sft] = acc[sft][t];
      }
  }
  if (blockIdx.z == 0 && wid == 0) {
    const float tot = bsum + __shfl_xor(bsum, 32, 64);
    if (lhi == 0 && co_ok) pw[(long long)Cout * Cin * S + co] = tot;
  }
}
// dW[e(g)] += sum_{tiles of run g} part[tile]   (and db)
__global__ __launch_bounds__(256) void gmm_wgrad_reduce_kernel(const float* __restrict__ part, const int* run_tile0, const int* seg_eid,
                                                               const long long* gwptr, const long long* gbptr, long long wsize, int Cout) {
  const int g = blockIdx.y;
  const int t0 = run_tile0[g], t1 = run_tile0[g + 1];
  const int e = seg_eid[g];
  float* dW = reinterpret_cast<float*>(gwptr[e]);
  float* db = gbptr ? reinterpret_cast<float*>(gbptr[e]) : nullptr;
  const long long img = wsize + Cout;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < img; i += (long long)gridDim.x * 256) {
    float sacc = 0.f;
    for (int t = t0; t < t1; ++t) sacc += part[(long long)t * img + i];
    if (i < wsize) dW[i] += sacc;
    else if (db) db[i - wsize] += sacc;
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
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int i = seg_start[g]; i < seg_start[g + 1]; ++i) s += rows[(long long)i * C + c];
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
  if (S == 1)
    hipLaunchKernelGGL(gmm_fwd_kernel<1>, grid, dim3(256), 0, st, x, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, (const long long*)bptr, y,
                       R, Cin, Cout, pad);
  else
    hipLaunchKernelGGL(gmm_fwd_kernel<3>, grid, dim3(256), 0, st, x, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, (const long long*)bptr, y,
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
    hipLaunchKernelGGL(gmm_dgrad_kernel<1>, grid, dim3(256), 0, st, dy, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, dx, R, Cin, Cout, pad);
  else
    hipLaunchKernelGGL(gmm_dgrad_kernel<3>, grid, dim3(256), 0, st, dy, seg_start, seg_eid, tile_seg, tile_row0, (const long long*)wptr, dx, R, Cin, Cout, pad);
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
    hipLaunchKernelGGL(gmm_wgrad_kernel<1>, grid, dim3(512), 0, st, dy, x, seg_start, tile_seg, tile_row0, tile_rows, (float*)workspace, R, Cin, Cout,
                       pad);
  else
    hipLaunchKernelGGL(gmm_wgrad_kernel<3>, grid, dim3(512), 0, st, dy, x, seg_start, tile_seg, tile_row0, tile_rows, (float*)workspace, R, Cin, Cout,
                       pad);
  HWG_LAUNCH_CHECK("grouped_conv1d_wgrad");
  const long long wsize = (long long)Cout * Cin * S;
  hipLaunchKernelGGL(gmm_wgrad_reduce_kernel, dim3(hwg_cdiv(wsize + Cout, 256 * 4), G), dim3(256), 0, st, (const float*)workspace, run_tile0, seg_eid,
                     (const long long*)gwptr, (const long long*)gbptr, wsize, Cout);
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
