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


// ---- bank of linear layers sharing one input (the generator's 10 AdaIN style->(gamma,beta) affines, pure_gen.py:52-69) ----------
// y_l = x W_l^T + b_l for l < L, x [B][I]; layer l has O_l outputs split in `halves` column blocks of O_l/halves, each written as
// a contiguous [B][O_l/halves] matrix at y + off_l + h*B*(O_l/halves)  (so gamma and beta come out as separate contiguous tensors).
// One wavefront per output neuron: lanes sweep the input features, the weight row is read once and reused for all B rows.
constexpr int LB_MAXB = 16;
__global__ __launch_bounds__(256) void linear_bank_fwd_kernel(const float* __restrict__ x, const long long* wptr, const long long* bptr, const int* O,
                                                              const int* first_wave, const long long* off, int L, int B, int I, int halves,
                                                              float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;   // global output neuron index over all layers
  if (w >= first_wave[L]) return;
  int l = 0;
  while (l + 1 < L && first_wave[l + 1] <= w) ++l;
  const int o = w - first_wave[l];
  const float* W = reinterpret_cast<const float*>(wptr[l]) + (long long)o * I;
  float acc[LB_MAXB];
#pragma unroll
  for (int b = 0; b < LB_MAXB; ++b) acc[b] = 0.f;
  for (int i = lane; i < I; i += 64) {
    const float wv = W[i];
#pragma unroll
    for (int b = 0; b < LB_MAXB; ++b)
      if (b < B) acc[b] += wv * x[(long long)b * I + i];
  }
  const int C = O[l] / halves, h = o / C, c = o - h * C;
  const float bias = bptr ? reinterpret_cast<const float*>(bptr[l])[o] : 0.f;
#pragma unroll
  for (int b = 0; b < LB_MAXB; ++b) {
    if (b < B) {
      const float v = wave_sum(acc[b]);
      if (lane == 0) y[off[l] + ((long long)h * B + b) * C + c] = v + bias;
    }
  }
}
// parameter gradients, ADDED into the buffers of the tables: dW_l[o][i] += sum_b dy_l[b][o] x[b][i];  db_l[o] += sum_b dy_l[b][o]
// dyptr[l*halves + h] points to the [B][O_l/halves] gradient of block h of layer l (0: that output was not used)
__global__ __launch_bounds__(256) void linear_bank_wgrad_kernel(const float* __restrict__ x, const long long* dyptr, const long long* gwptr,
                                                                const long long* gbptr, const int* O, const int* first_wave, int L, int B, int I,
                                                                int halves) {
  const int lane = threadIdx.x & 63;
  const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (w >= first_wave[L]) return;
  int l = 0;
  while (l + 1 < L && first_wave[l + 1] <= w) ++l;
  const int o = w - first_wave[l];
  const int C = O[l] / halves, h = o / C, c = o - h * C;
  const float* dy = reinterpret_cast<const float*>(dyptr[l * halves + h]);
  if (!dy) return;
  float d[LB_MAXB];
  float bs = 0.f;
#pragma unroll
  for (int b = 0; b < LB_MAXB; ++b) {
    d[b] = (b < B) ? dy[(long long)b * C + c] : 0.f;
    bs += d[b];
  }
  if (!gwptr[l]) return;   // layer frozen
  float* gW = reinterpret_cast<float*>(gwptr[l]) + (long long)o * I;
  for (int i = lane; i < I; i += 64) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < LB_MAXB; ++b)
      if (b < B) s += d[b] * x[(long long)b * I + i];
    gW[i] += s;
  }
  if (lane == 0 && gbptr && gbptr[l]) reinterpret_cast<float*>(gbptr[l])[o] += bs;
}
// input gradient: dx[b][i] = sum_l sum_o dy_l[b][o] W_l[o][i]; one workgroup per row b, 8 thread groups split the neurons and are
// combined through LDS in a fixed order
__global__ __launch_bounds__(1024) void linear_bank_dgrad_kernel(const long long* dyptr, const long long* wptr, const int* O, const int* first_wave, int L,
                                                                 int B, int I, int halves, float* __restrict__ dx) {
  __shared__ float red[8][128];
  const int b = blockIdx.x;
  const int part = threadIdx.x >> 7, il = threadIdx.x & 127;
  const int total = first_wave[L];
  for (int i0 = 0; i0 < I; i0 += 128) {
    const int i = i0 + il;
    float s = 0.f;
    if (i < I) {
      int l = 0;
      for (int w = part; w < total; w += 8) {
        while (l + 1 < L && first_wave[l + 1] <= w) ++l;
        const int o = w - first_wave[l];
        const int C = O[l] / halves, h = o / C, c = o - h * C;
        const float* dy = reinterpret_cast<const float*>(dyptr[l * halves + h]);
        if (!dy) continue;
        s += dy[(long long)b * C + c] * reinterpret_cast<const float*>(wptr[l])[(long long)o * I + i];
      }
    }
    red[part][il] = s;
    __syncthreads();
    if (part == 0 && i < I) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 8; ++p) t += red[p][il];
      dx[(long long)b * I + i] = t;
    }
    __syncthreads();
  }
}


// ---- chain of L square linear layers with LeakyReLU (the generator's style embedding MLP, 6 x Linear(128,128), pure_gen.py:29-38) -------
// in ONE workgroup per pass: h_{l+1} = lrelu(W_l h_l + b_l); the B x D activations live in LDS, weights stream from L2 (64 KB / layer).
// acts [L+1][B][D] keeps every h_l for the backward pass (acts[0] = x).
constexpr int MC_MAXB = 16, MC_MAXD = 256;
__global__ __launch_bounds__(1024) void mlp_chain_fwd_kernel(const float* __restrict__ x, const long long* wptr, const long long* bptr, int L, int B, int D,
                                                             float slope, float* __restrict__ acts) {
  __shared__ float h[MC_MAXB * MC_MAXD];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int t = tid; t < B * D; t += nt) { h[t] = x[t]; acts[t] = x[t]; }
  __syncthreads();
  for (int l = 0; l < L; ++l) {
    const float* W = reinterpret_cast<const float*>(wptr[l]);
    const float* bias = reinterpret_cast<const float*>(bptr[l]);
    float outv[(MC_MAXB * MC_MAXD + 1023) / 1024];
    int cnt = 0;
    for (int t = tid; t < B * D; t += nt, ++cnt) {
      const int o = t % D, b = t / D;
      const float* wr = W + (long long)o * D;
      const float* hr = h + b * D;
      float a = bias[o];
      for (int i = 0; i < D; ++i) a += wr[i] * hr[i];
      outv[cnt] = a > 0.f ? a : a * slope;
    }
    __syncthreads();
    cnt = 0;
    for (int t = tid; t < B * D; t += nt, ++cnt) {
      h[t] = outv[cnt];
      acts[(long long)(l + 1) * B * D + t] = outv[cnt];
    }
    __syncthreads();
  }
}
// backward of the chain; parameter gradients are ADDED into the tables' buffers
__global__ __launch_bounds__(1024) void mlp_chain_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ acts, const long long* wptr,
                                                             const long long* gwptr, const long long* gbptr, int L, int B, int D, float slope,
                                                             float* __restrict__ dx) {
  __shared__ float dpre[MC_MAXB * MC_MAXD];   // d(pre-activation) of the current layer
  __shared__ float hin[MC_MAXB * MC_MAXD];    // its input h_l
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int t = tid; t < B * D; t += nt) dpre[t] = dout[t];
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    // dpre currently holds dL/dh_{l+1}; apply the LeakyReLU derivative (sign of h_{l+1} == sign of the pre-activation)
    for (int t = tid; t < B * D; t += nt) {
      const float hv = acts[(long long)(l + 1) * B * D + t];
      dpre[t] *= hv > 0.f ? 1.f : slope;
      hin[t] = acts[(long long)l * B * D + t];
    }
    __syncthreads();
    const float* W = reinterpret_cast<const float*>(wptr[l]);
    float* gW = gwptr[l] ? reinterpret_cast<float*>(gwptr[l]) : nullptr;
    float* gb = (gbptr && gbptr[l]) ? reinterpret_cast<float*>(gbptr[l]) : nullptr;
    if (gW) {
      for (int t = tid; t < D * D; t += nt) {       // dW[o][i] += sum_b dpre[b][o] * h_l[b][i]
        const int ii = t % D, o = t / D;
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dpre[b * D + o] * hin[b * D + ii];
        gW[t] += a;
      }
    }
    if (gb) {
      for (int o = tid; o < D; o += nt) {
        float a = 0.f;
        for (int b = 0; b < B; ++b) a += dpre[b * D + o];
        gb[o] += a;
      }
    }
    // dh_l[b][i] = sum_o dpre[b][o] * W[o][i]
    float outv[(MC_MAXB * MC_MAXD + 1023) / 1024];
    int cnt = 0;
    for (int t = tid; t < B * D; t += nt, ++cnt) {
      const int ii = t % D, b = t / D;
      float a = 0.f;
      for (int o = 0; o < D; ++o) a += dpre[b * D + o] * W[(long long)o * D + ii];
      outv[cnt] = a;
    }
    __syncthreads();
    cnt = 0;
    for (int t = tid; t < B * D; t += nt, ++cnt) dpre[t] = outv[cnt];
    __syncthreads();
  }
  if (dx)
    for (int t = tid; t < B * D; t += nt) dx[t] = dpre[t];
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

extern "C" int hwg_linear_bank_fwd(const float* x, const void* wptr, const void* bptr, const int* O, const int* first_wave, const void* off, int L, int B,
                                   int I, int halves, int total_outputs, float* y, void* stream) {
  HWG_REQUIRE(x && wptr && O && first_wave && off && y && L > 0 && B > 0 && B <= LB_MAXB && I > 0 && halves > 0 && total_outputs > 0,
              "linear_bank_fwd: bad arguments (B <= %d)", LB_MAXB);
  hipLaunchKernelGGL(linear_bank_fwd_kernel, dim3(hwg_cdiv(total_outputs, 4)), dim3(256), 0, (hipStream_t)stream, x, (const long long*)wptr,
                     (const long long*)bptr, O, first_wave, (const long long*)off, L, B, I, halves, y);
  HWG_LAUNCH_CHECK("linear_bank_fwd");
  return HWG_OK;
}
extern "C" int hwg_linear_bank_bwd(const float* x, const void* dyptr, const void* wptr, const void* gwptr, const void* gbptr, const int* O,
                                   const int* first_wave, int L, int B, int I, int halves, int total_outputs, float* dx, void* stream) {
  HWG_REQUIRE(x && dyptr && wptr && gwptr && O && first_wave && L > 0 && B > 0 && B <= LB_MAXB && I > 0 && halves > 0 && total_outputs > 0,
              "linear_bank_bwd: bad arguments (B <= %d)", LB_MAXB);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_bank_wgrad_kernel, dim3(hwg_cdiv(total_outputs, 4)), dim3(256), 0, st, x, (const long long*)dyptr, (const long long*)gwptr,
                     (const long long*)gbptr, O, first_wave, L, B, I, halves);
  HWG_LAUNCH_CHECK("linear_bank_wgrad");
  if (dx) {
    hipLaunchKernelGGL(linear_bank_dgrad_kernel, dim3(B), dim3(1024), 0, st, (const long long*)dyptr, (const long long*)wptr, O, first_wave, L, B, I, halves,
                       dx);
    HWG_LAUNCH_CHECK("linear_bank_dgrad");
  }
  return HWG_OK;
}

extern "C" int hwg_mlp_chain_fwd(const float* x, const void* wptr, const void* bptr, int L, int B, int D, float slope, float* acts, void* stream) {
  HWG_REQUIRE(x && wptr && bptr && acts && L > 0 && B > 0 && B <= MC_MAXB && D > 0 && D <= MC_MAXD, "mlp_chain_fwd: bad arguments (B <= %d, D <= %d)",
              MC_MAXB, MC_MAXD);
  hipLaunchKernelGGL(mlp_chain_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (const long long*)wptr, (const long long*)bptr, L, B, D, slope,
                     acts);
  HWG_LAUNCH_CHECK("mlp_chain_fwd");
  return HWG_OK;
}
extern "C" int hwg_mlp_chain_bwd(const float* dout, const float* acts, const void* wptr, const void* gwptr, const void* gbptr, int L, int B, int D,
                                 float slope, float* dx, void* stream) {
  HWG_REQUIRE(dout && acts && wptr && gwptr && L > 0 && B > 0 && B <= MC_MAXB && D > 0 && D <= MC_MAXD, "mlp_chain_bwd: bad arguments (B <= %d, D <= %d)",
              MC_MAXB, MC_MAXD);
  hipLaunchKernelGGL(mlp_chain_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, dout, acts, (const long long*)wptr, (const long long*)gwptr,
                     (const long long*)gbptr, L, B, D, slope, dx);
  HWG_LAUNCH_CHECK("mlp_chain_bwd");
  return HWG_OK;
}
