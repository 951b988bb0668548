// Grouped ("one expert per sample") 1-D convolutions / linears for the 79 character-style experts
// (model/char_style.py:84-124, 210-235). The reference runs one tiny network per recognised character in a Python loop;
// here all windows of all characters go through each layer in ONE launch, every window reading the weights of its own
// expert through a device pointer table. These layers are bound by streaming the experts' weights (1.45 MB per expert) from
// L2/HBM, not by FLOPs (1.4 MMAC per window), so they are written as coalesced weight-streaming kernels, not GEMMs.
//
// Layouts: x [n][R][Cin], y [n][R][Cout] (R positions, channels fastest); weights in the PyTorch Conv1d layout [Cout][Cin][S]
// (nn.Linear [Cout][Cin] is S = 1, R = 1). Windows are sorted by expert; seg_start/seg_eid describe the runs.
#include "hwg_common.h"

namespace {

constexpr int MAXR = 8;  // positions per window (5 for window=2)

// y[i][p][co] = b[co] + sum_{ci,s} x[i][p+s-pad][ci] * W[co][ci][s]; one wave per (window, output channel), lanes sweep (ci,s)
__global__ __launch_bounds__(256) void gconv_fwd_kernel(const float* x, const int* eid, const long long* wptr, const long long* bptr, float* y,
                                                        int n, int R, int Cin, int Cout, int S, int pad, int co_per_block) {
  extern __shared__ float xs[];  // [R][Cin]
  const int i = blockIdx.x;
  const int e = eid[i];
  const float* W = reinterpret_cast<const float*>(wptr[e]);
  const float* bias = bptr ? reinterpret_cast<const float*>(bptr[e]) : nullptr;
  for (int t = threadIdx.x; t < R * Cin; t += 256) xs[t] = x[(long long)i * R * Cin + t];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int KS = Cin * S;
  const int co0 = blockIdx.y * co_per_block;
  for (int co = co0 + wid; co < min(co0 + co_per_block, Cout); co += 4) {
    const float* wr = W + (long long)co * KS;
    float acc[MAXR];
#pragma unroll
    for (int p = 0; p < MAXR; ++p) acc[p] = 0.f;
    for (int j = lane; j < KS; j += 64) {
      const float w = wr[j];
      const int ci = j / S, s = j - ci * S;
#pragma unroll
      for (int p = 0; p < MAXR; ++p) {
        if (p >= R) break;
        const int q = p + s - pad;
        if (q >= 0 && q < R) acc[p] += w * xs[q * Cin + ci];
      }
    }
#pragma unroll
    for (int p = 0; p < MAXR; ++p) {
      if (p >= R) break;
      const float v = wave_sum(acc[p]);
      if (lane == 0) y[((long long)i * R + p) * Cout + co] = v + (bias ? bias[co] : 0.f);
    }
  }
}

// dx[i][q][ci] = sum_{co,s} dy[i][q-s+pad][co] * W[co][ci][s]; one thread per input channel, weights read as contiguous [ci][s] runs
__global__ __launch_bounds__(256) void gconv_dgrad_kernel(const float* dy, const int* eid, const long long* wptr, float* dx, int n, int R, int Cin,
                                                          int Cout, int S, int pad) {
  extern __shared__ float dys[];  // [R][Cout]
  const int i = blockIdx.x;
  const float* W = reinterpret_cast<const float*>(wptr[eid[i]]);
  for (int t = threadIdx.x; t < R * Cout; t += 256) dys[t] = dy[(long long)i * R * Cout + t];
  __syncthreads();
  for (int ci = blockIdx.y * 256 + threadIdx.x; ci < Cin; ci += gridDim.y * 256) {
    float acc[MAXR];
#pragma unroll
    for (int q = 0; q < MAXR; ++q) acc[q] = 0.f;
    for (int co = 0; co < Cout; ++co) {
      const float* wr = W + ((long long)co * Cin + ci) * S;
      for (int s = 0; s < S; ++s) {
        const float w = wr[s];
#pragma unroll
        for (int q = 0; q < MAXR; ++q) {
          if (q >= R) break;
          const int p = q - s + pad;
          if (p >= 0 && p < R) acc[q] += w * dys[p * Cout + co];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < MAXR; ++q) {
      if (q >= R) break;
      dx[((long long)i * R + q) * Cin + ci] = acc[q];
    }
  }
}

// dW[e][co][ci][s] += sum_{i in run} sum_p dy[i][p][co] * x[i][p+s-pad][ci];  db[e][co] += sum dy
// block = (run of one expert, tile of CPB output channels). A lane owns up to WG_MAXJ (ci,s) entries; for every window it
// first pulls the R shifted x values of its entries into registers, then reuses them for all channels of the tile, so the
// inner loop is one LDS broadcast read (dy) per R*WG_MAXJ FMAs.
constexpr int WG_MAXJ = 12;  // (Cin*S)/64 <= 12  (256*3/64)
constexpr int WG_CPB = 16;   // output channels per block
template <int R>
__global__ __launch_bounds__(256) void gconv_wgrad_kernel(const float* dy, const float* x, const int* seg_start, const int* seg_eid,
                                                          const long long* gwptr, const long long* gbptr, int Cin, int Cout, int S, int pad) {
  extern __shared__ float sm[];  // xs [R][Cin] | dys [R][WG_CPB]
  float* xs = sm;
  float* dys = sm + R * Cin;
  const int g = blockIdx.x;
  const int e = seg_eid[g];
  const int i0 = seg_start[g], i1 = seg_start[g + 1];
  float* dW = reinterpret_cast<float*>(gwptr[e]);
  float* db = gbptr ? reinterpret_cast<float*>(gbptr[e]) : nullptr;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int KS = Cin * S;
  const int co0 = blockIdx.y * WG_CPB;
  const int nco = min(WG_CPB, Cout - co0);
  constexpr int MAXC = WG_CPB / 4;
  float acc[MAXC][WG_MAXJ];
  float bacc[MAXC];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    bacc[c] = 0.f;
#pragma unroll
    for (int j = 0; j < WG_MAXJ; ++j) acc[c][j] = 0.f;
  }
  // the (ci, s) entries this lane owns
  int ent_ci[WG_MAXJ], ent_s[WG_MAXJ];
#pragma unroll
  for (int jj = 0; jj < WG_MAXJ; ++jj) {
    const int j = lane + 64 * jj;
    ent_ci[jj] = (j < KS) ? j / S : 0;
    ent_s[jj] = (j < KS) ? j % S : 0;
  }
  for (int i = i0; i < i1; ++i) {
    __syncthreads();
    for (int t = threadIdx.x; t < R * Cin; t += 256) xs[t] = x[(long long)i * R * Cin + t];
    for (int t = threadIdx.x; t < R * nco; t += 256) {
      const int p = t / nco, c = t - p * nco;
      dys[p * WG_CPB + c] = dy[((long long)i * R + p) * Cout + co0 + c];
    }
    __syncthreads();
    float xv[WG_MAXJ][R];
#pragma unroll
    for (int jj = 0; jj < WG_MAXJ; ++jj)
#pragma unroll
      for (int p = 0; p < R; ++p) {
        const int q = p + ent_s[jj] - pad;
        xv[jj][p] = (lane + 64 * jj < KS && q >= 0 && q < R) ? xs[q * Cin + ent_ci[jj]] : 0.f;
      }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int cl = wid + 4 * c;
      if (cl < nco) {
        float d[R];
        float b = 0.f;
#pragma unroll
        for (int p = 0; p < R; ++p) { d[p] = dys[p * WG_CPB + cl]; b += d[p]; }
        bacc[c] += b;
#pragma unroll
        for (int jj = 0; jj < WG_MAXJ; ++jj)
#pragma unroll
          for (int p = 0; p < R; ++p) acc[c][jj] += d[p] * xv[jj][p];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int cl = wid + 4 * c;
    if (cl >= nco) continue;
    float* wr = dW + (long long)(co0 + cl) * KS;
#pragma unroll
    for (int jj = 0; jj < WG_MAXJ; ++jj) {
      const int j = lane + 64 * jj;
      if (j < KS) wr[j] += acc[c][jj];
    }
    if (lane == 0 && db) db[co0 + cl] += bacc[c];
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

extern "C" int hwg_grouped_conv1d_fwd(const float* x, const int* eid, const void* wptr, const void* bptr, float* y, int n, int R, int Cin, int Cout,
                                      int S, int pad, void* stream) {
  HWG_REQUIRE(x && eid && wptr && y && n > 0 && R > 0 && R <= MAXR && Cin > 0 && Cout > 0 && S > 0, "grouped_conv1d_fwd: bad arguments");
  const int cpb = 32;
  const size_t smem = (size_t)R * Cin * sizeof(float);
  HWG_REQUIRE(smem <= 64 * 1024, "grouped_conv1d_fwd: window does not fit LDS");
  hipLaunchKernelGGL(gconv_fwd_kernel, dim3(n, hwg_cdiv(Cout, cpb)), dim3(256), smem, (hipStream_t)stream, x, eid, (const long long*)wptr,
                     (const long long*)bptr, y, n, R, Cin, Cout, S, pad, cpb);
  HWG_LAUNCH_CHECK("grouped_conv1d_fwd");
  return HWG_OK;
}
extern "C" int hwg_grouped_conv1d_dgrad(const float* dy, const int* eid, const void* wptr, float* dx, int n, int R, int Cin, int Cout, int S, int pad,
                                        void* stream) {
  HWG_REQUIRE(dy && eid && wptr && dx && n > 0 && R > 0 && R <= MAXR && Cin > 0 && Cout > 0 && S > 0, "grouped_conv1d_dgrad: bad arguments");
  const size_t smem = (size_t)R * Cout * sizeof(float);
  HWG_REQUIRE(smem <= 64 * 1024, "grouped_conv1d_dgrad: window does not fit LDS");
  hipLaunchKernelGGL(gconv_dgrad_kernel, dim3(n, hwg_cdiv(Cin, 256)), dim3(256), smem, (hipStream_t)stream, dy, eid, (const long long*)wptr, dx, n, R,
                     Cin, Cout, S, pad);
  HWG_LAUNCH_CHECK("grouped_conv1d_dgrad");
  return HWG_OK;
}
extern "C" int hwg_grouped_conv1d_wgrad(const float* dy, const float* x, const int* seg_start, const int* seg_eid, int G, const void* gwptr,
                                        const void* gbptr, int R, int Cin, int Cout, int S, int pad, void* stream) {
  HWG_REQUIRE(dy && x && seg_start && seg_eid && gwptr && G > 0 && R > 0 && R <= MAXR && Cin > 0 && Cout > 0 && S > 0, "grouped_conv1d_wgrad: bad arguments");
  HWG_REQUIRE(Cin * S <= 64 * WG_MAXJ, "grouped_conv1d_wgrad: Cin*S=%d too large", Cin * S);
  const size_t smem = ((size_t)R * Cin + (size_t)R * WG_CPB) * sizeof(float);
  dim3 grid(G, hwg_cdiv(Cout, WG_CPB));
  hipStream_t st = (hipStream_t)stream;
#define HWG_WG_CASE(RR)                                                                                                              \
  case RR:                                                                                                                           \
    hipLaunchKernelGGL(gconv_wgrad_kernel<RR>, grid, dim3(256), smem, st, dy, x, seg_start, seg_eid, (const long long*)gwptr,        \
                       (const long long*)gbptr, Cin, Cout, S, pad);                                                                  \
    break;
  switch (R) {
    HWG_WG_CASE(1) HWG_WG_CASE(2) HWG_WG_CASE(3) HWG_WG_CASE(4) HWG_WG_CASE(5) HWG_WG_CASE(6) HWG_WG_CASE(7) HWG_WG_CASE(8)
    default: break;
  }
#undef HWG_WG_CASE
  HWG_LAUNCH_CHECK("grouped_conv1d_wgrad");
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
