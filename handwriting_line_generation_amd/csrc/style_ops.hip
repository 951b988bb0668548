// Character-specific style extraction helpers (model/char_style.py:204-235, 286): window gather around
// recognised characters, its scatter-add backward, and the confidence-weighted per-author mean.
#include "hwg_common.h"

namespace {

// patches[i][j][c] = x[b_i][pos_i - w + j][c] (zero outside [0,Wx)), j in [0, 2w+1)
__global__ void gather_windows_kernel(const float* x, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int w, float* patches) {
  const int WW = 2 * w + 1;
  const long long total = (long long)n * WW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int j = (int)(t % WW);
    const int k = (int)(t / WW);
    const int pos = idx_pos[k] - w + j;
    patches[i] = (pos >= 0 && pos < Wx) ? x[((long long)idx_b[k] * Wx + pos) * C + c] : 0.f;
  }
}
// dx[b_i][pos][c] += dpatches[i][j][c]   (windows overlap -> atomics)
__global__ void scatter_windows_kernel(const float* dpatches, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int w, float* dx) {
  const int WW = 2 * w + 1;
  const long long total = (long long)n * WW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int j = (int)(t % WW);
    const int k = (int)(t / WW);
    const int pos = idx_pos[k] - w + j;
    if (pos >= 0 && pos < Wx) atomicAdd(dx + ((long long)idx_b[k] * Wx + pos) * C + c, dpatches[i]);
  }
}
// total[b][c] = sum_{i: seg_i == b} wgt_i * v[i][c];  wsum[b] = sum wgt_i;  out = wsum != 0 ? total / wsum : total
// (sequential over i in list order, like the reference's python accumulation loop)
__global__ void segment_weighted_mean_kernel(const float* v, const float* wgt, const int* seg, int n, int C, int B, float* out, float* wsum_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  float tot = 0.f, ws = 0.f;
  for (int k = 0; k < n; ++k)
    if (seg[k] == b) { tot += wgt[k] * v[(long long)k * C + c]; ws += wgt[k]; }
  out[i] = ws != 0.f ? tot / ws : tot;
  if (c == 0) wsum_out[b] = ws;
}
// dv[i][c] = wgt_i / wsum[seg_i] * dout[seg_i][c]
__global__ void segment_weighted_mean_bwd_kernel(const float* dout, const float* wgt, const int* seg, const float* wsum, int n, int C, float* dv) {
  const long long total = (long long)n * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int k = (int)(i / C);
    const int b = seg[k];
    const float ws = wsum[b];
    dv[i] = (ws != 0.f ? wgt[k] / ws : wgt[k]) * dout[(long long)b * C + c];
  }
}
// out[i] = exp(x[b_i][pos_i][cls_i])   (confidence of the recognised character)
__global__ void gather_scores_kernel(const float* x, int Wx, int C, const int* idx_b, const int* idx_pos, const int* idx_cls, int n, float* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  out[k] = expf(x[((long long)idx_b[k] * Wx + idx_pos[k]) * C + idx_cls[k]]);
}

}  // namespace

extern "C" int hwg_gather_windows(const float* x, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int window, float* patches,
                                  void* stream) {
  HWG_REQUIRE(x && idx_b && idx_pos && patches && B > 0 && Wx > 0 && C > 0 && n > 0 && window >= 0, "gather_windows: bad arguments");
  const long long total = (long long)n * (2 * window + 1) * C;
  hipLaunchKernelGGL(gather_windows_kernel, dim3(hwg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, Wx, C, idx_b, idx_pos, n, window,
                     patches);
  HWG_LAUNCH_CHECK("gather_windows");
  return HWG_OK;
}
extern "C" int hwg_scatter_windows(const float* dpatches, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int window, float* dx,
                                   void* stream) {
  HWG_REQUIRE(dpatches && idx_b && idx_pos && dx && B > 0 && Wx > 0 && C > 0 && n > 0 && window >= 0, "scatter_windows: bad arguments");
  const long long total = (long long)n * (2 * window + 1) * C;
  hipLaunchKernelGGL(scatter_windows_kernel, dim3(hwg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dpatches, Wx, C, idx_b, idx_pos, n,
                     window, dx);
  HWG_LAUNCH_CHECK("scatter_windows");
  return HWG_OK;
}
extern "C" int hwg_segment_weighted_mean(const float* v, const float* wgt, const int* seg, int n, int C, int B, float* out, float* wsum, void* stream) {
  HWG_REQUIRE(v && wgt && seg && out && wsum && n > 0 && C > 0 && B > 0, "segment_weighted_mean: bad arguments");
  hipLaunchKernelGGL(segment_weighted_mean_kernel, dim3(hwg_cdiv(B * C, 128)), dim3(128), 0, (hipStream_t)stream, v, wgt, seg, n, C, B, out, wsum);
  HWG_LAUNCH_CHECK("segment_weighted_mean");
  return HWG_OK;
}
extern "C" int hwg_segment_weighted_mean_bwd(const float* dout, const float* wgt, const int* seg, const float* wsum, int n, int C, float* dv,
                                             void* stream) {
  HWG_REQUIRE(dout && wgt && seg && wsum && dv && n > 0 && C > 0, "segment_weighted_mean_bwd: bad arguments");
  hipLaunchKernelGGL(segment_weighted_mean_bwd_kernel, dim3(hwg_stream_grid((long long)n * C, 256)), dim3(256), 0, (hipStream_t)stream, dout, wgt, seg,
                     wsum, n, C, dv);
  HWG_LAUNCH_CHECK("segment_weighted_mean_bwd");
  return HWG_OK;
}
extern "C" int hwg_gather_scores(const float* x, int B, int Wx, int C, const int* idx_b, const int* idx_pos, const int* idx_cls, int n, float* out,
                                 void* stream) {
  HWG_REQUIRE(x && idx_b && idx_pos && idx_cls && out && n > 0, "gather_scores: bad arguments");
  hipLaunchKernelGGL(gather_scores_kernel, dim3(hwg_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, x, Wx, C, idx_b, idx_pos, idx_cls, n, out);
  HWG_LAUNCH_CHECK("gather_scores");
  return HWG_OK;
}
