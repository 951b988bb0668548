// Character-specific style extraction helpers (model/char_style.py:204-235, 286): window gather around
// recognised characters, its scatter-add backward, and the confidence-weighted per-author mean.
#include "hwg_common.h"

namespace {

// patches[i][j][c] = x[b_i][pos_i - w + j][c] (zero outside [0,Wx)), j in [0, 2w+1)
__global__ void gather_windows_kernel(const float* x, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int w, float* patches) {
  const int WW = 2 * w + 1;
  const long long total = (long long)n * WW * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int j = (int)(t % WW);
    const int k = (int)(t / WW);
    const int pos = idx_pos[k] - w + j;
    const int b = idx_b[k];
    patches[i] = (pos >= 0 && pos < Wx && b >= 0 && b < B) ? x[((long long)b * Wx + pos) * C + c] : 0.f;
  }
}
// dx[b][pos][c] = sum over the windows that cover (b, pos) of dpatches[k][j][c]. The windows overlap, so a scatter would need floating
// point atomics (non-deterministic summation order); instead the window list is inverted once (at most ONE window is centred on a given
// (sample, column): the column's arg-max class) and every output element gathers its <= 2w+1 contributions in a fixed order.
// PRECONDITION (ops.gather_windows states and, under HWG_DEBUG=1, asserts it): window centres (idx_b, idx_pos) are unique. Centres outside
// the tensor are ignored (their windows lie outside it entirely or were zero-filled by the gather); with duplicate centres the window with
// the largest list index would be the only one to contribute - atomicMax keeps even that case deterministic.
__global__ void window_index_kernel(const int* idx_b, const int* idx_pos, int n, int B, int Wx, int* win_of) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int b = idx_b[k], p = idx_pos[k];
  if (b >= 0 && b < B && p >= 0 && p < Wx) atomicMax(&win_of[b * Wx + p], k);
}
__global__ void scatter_windows_kernel(const float* dpatches, int B, int Wx, int C, const int* win_of, int w, float* dx) {
  const int WW = 2 * w + 1;
  const long long total = (long long)B * Wx * C;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long t = i / C;
    const int pos = (int)(t % Wx);
    const int b = (int)(t / Wx);
    float acc = 0.f;
    for (int j = 0; j < WW; ++j) {
      const int centre = pos + w - j;
      if (centre < 0 || centre >= Wx) continue;
      const int k = win_of[b * Wx + centre];
      if (k >= 0) acc += dpatches[((long long)k * WW + j) * C + c];
    }
    dx[i] = acc;
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
    if (seg[k] == b) { tot = fmaf(wgt[k], v[(long long)k * C + c], tot); ws += wgt[k]; }
  out[i] = ws != 0.f ? tot / ws : tot;
  if (c == 0) wsum_out[b] = ws;
}
// Same sums, one workgroup per segment: the first wavefront compacts the segment's members (ascending k: the order of the loop above) into
// LDS, then every thread adds only those, eight independent loads at a time. The kernel above walks all n items per output with a
// dependent load chain (120 us for 970 windows x 8 lines); this one takes the time of n / B loads.
__global__ __launch_bounds__(256) void segment_weighted_mean_list_kernel(const float* __restrict__ v, const float* __restrict__ wgt,
                                                                         const int* __restrict__ seg, int n, int C, float* out, float* wsum_out) {
  extern __shared__ int seg_list[];                 // [n] member indices, then [n] their weights, then the member count
  float* lw = reinterpret_cast<float*>(seg_list + n);
  int* count = seg_list + 2 * n;
  const int b = blockIdx.x, lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    int base = 0;
    for (int k0 = 0; k0 < n; k0 += 256) {
      int sg[4];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {               // four groups of 64 items in flight
        const int k = k0 + 64 * u + lane;
        sg[u] = k < n ? seg[k] : -1;
        w[u] = k < n ? wgt[k] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool m = sg[u] == b;
        const unsigned long long mask = __ballot(m);
        if (m) {
          const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
          seg_list[pos] = k0 + 64 * u + lane;
          lw[pos] = w[u];
        }
        base += __popcll(mask);
      }
    }
    if (lane == 0) *count = base;
  }
  __syncthreads();
  const int cnt = *count;
  float ws = 0.f;
  for (int i = 0; i < cnt; ++i) ws += lw[i];
  for (int c = threadIdx.x; c < C; c += 256) {
    float tot = 0.f;
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = v[(long long)seg_list[i + u] * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) tot = fmaf(lw[i + u], x[u], tot);
    }
    for (; i < cnt; ++i) tot = fmaf(lw[i], v[(long long)seg_list[i] * C + c], tot);
    out[(long long)b * C + c] = ws != 0.f ? tot / ws : tot;
  }
  if (threadIdx.x == 0) wsum_out[b] = ws;
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
  // rows b0 .. b0 + nb of the batch (blockIdx.y: 16 rows each; training batches are one chunk, a generation service's 64-256 lines several)
  const int b0 = blockIdx.y * LB_MAXB;
  const int nb = min(B - b0, LB_MAXB);
  const float* __restrict__ xb = x + (long long)b0 * I;
  float acc[LB_MAXB];
#pragma unroll
  for (int b = 0; b < LB_MAXB; ++b) acc[b] = 0.f;
  for (int i = lane; i < I; i += 64) {
    const float wv = W[i];
#pragma unroll
    for (int b = 0; b < LB_MAXB; ++b)
      if (b < nb) acc[b] += wv * xb[(long long)b * I + i];
  }
  const int C = O[l] / halves, h = o / C, c = o - h * C;
  const float bias = bptr ? reinterpret_cast<const float*>(bptr[l])[o] : 0.f;
#pragma unroll
  for (int b = 0; b < LB_MAXB; ++b) {
    if (b < nb) {
      const float v = wave_sum(acc[b]);
      if (lane == 0) y[off[l] + ((long long)h * B + b0 + b) * C + c] = v + bias;
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
// input gradient: dx[b][i] = sum_l sum_o dy_l[b][o] W_l[o][i]. Stage 1: workgroup c sums its chunk of LB_OCHUNK neurons for all rows
// (thread = (b, i), weight rows read coalesced along i and shared by the B rows) into part[c][b][i]; stage 2 adds the chunks in order.
constexpr int LB_OCHUNK = 32;
__global__ __launch_bounds__(1024) void linear_bank_dgrad_kernel(const long long* dyptr, const long long* wptr, const int* O, const int* first_wave, int L,
                                                                 int B, int I, int halves, float* __restrict__ part) {
  const int total = first_wave[L];
  const int w0 = blockIdx.x * LB_OCHUNK, w1 = min(w0 + LB_OCHUNK, total);
  int l0 = 0;
  while (l0 + 1 < L && first_wave[l0 + 1] <= w0) ++l0;
  for (int t = threadIdx.x; t < B * I; t += blockDim.x) {
    const int b = t / I, i = t - b * I;
    float s = 0.f;
    int l = l0;
    for (int w = w0; w < w1; ++w) {
      while (l + 1 < L && first_wave[l + 1] <= w) ++l;
      const int o = w - first_wave[l];
      const int C = O[l] / halves, h = o / C, c = o - h * C;
      const float* dy = reinterpret_cast<const float*>(dyptr[l * halves + h]);
      if (!dy) continue;
      s += dy[(long long)b * C + c] * reinterpret_cast<const float*>(wptr[l])[(long long)o * I + i];
    }
    part[((long long)blockIdx.x * B + b) * I + i] = s;
  }
}
__global__ __launch_bounds__(256) void linear_bank_dgrad_reduce_kernel(const float* __restrict__ part, int nchunks, int n, float* __restrict__ dx) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  float s = 0.f;
  for (int c = 0; c < nchunks; ++c) s += part[(long long)c * n + t];
  dx[t] = s;
}

// ---- chain of L square linear layers with LeakyReLU (the generator's style embedding MLP, 6 x Linear(128,128), pure_gen.py:29-38) -------
// in ONE workgroup per pass: h_{l+1} = lrelu(W_l h_l + b_l); the B x D activations live in LDS, weights stream from L2 (64 KB / layer).
// acts [L+1][B][D] keeps every h_l for the backward pass (acts[0] = x).
constexpr int MC_MAXB = 16, MC_MAXD = 128;
// A single workgroup is latency bound: every dependent global access costs ~2 us. So: the whole weight matrix of a layer sits in LDS
// (row stride DD+1: threads holding different neurons hit different banks), thread (b, o) runs its own dot product (no cross-lane
// reductions), and the NEXT layer's weights (and, backward, the old gradient values) are already in flight in registers while the
// current layer is computed. DD = feature count (64-multiple), BM = row capacity (rows >= B are kept at zero).
template <int DD, int BM>
__global__ __launch_bounds__(1024) void mlp_chain_fwd_kernel(const float* __restrict__ x, const long long* wptr, const long long* bptr, int L, int B,
                                                             float slope, float* __restrict__ acts) {
  constexpr int LDW = DD + 1, PER = DD * DD / 1024;
  __shared__ float h[BM * DD];
  __shared__ float hn[BM * DD];
  __shared__ float ws[DD * LDW];
  const int tid = threadIdx.x;
  float nxt[PER];
  {
    const float* W0 = reinterpret_cast<const float*>(wptr[0]);
#pragma unroll
    for (int u = 0; u < PER; ++u) nxt[u] = W0[tid + u * 1024];
  }
  // rows b0 .. b0 + nb of the batch (one workgroup per BM rows: the rows of an MLP are independent)
  const int b0 = blockIdx.x * BM;
  const int nb = min(B - b0, BM);
  x += (long long)b0 * DD;
  acts += (long long)b0 * DD;
  for (int t = tid; t < BM * DD; t += 1024) {
    const float v = t < nb * DD ? x[t] : 0.f;
    h[t] = v;
    if (t < nb * DD) acts[t] = v;
  }
  for (int l = 0; l < L; ++l) {
    const float* __restrict__ bias = reinterpret_cast<const float*>(bptr[l]);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int t = tid + u * 1024;
      ws[(t / DD) * LDW + (t % DD)] = nxt[u];
    }
    if (l + 1 < L) {
      const float* Wn = reinterpret_cast<const float*>(wptr[l + 1]);
#pragma unroll
      for (int u = 0; u < PER; ++u) nxt[u] = Wn[tid + u * 1024];
    }
    __syncthreads();
    for (int p = tid; p < BM * DD; p += 1024) {
      const int b = p / DD, o = p % DD;
      const float* wr = ws + o * LDW;
      const float* hr = h + b * DD;
      float a = bias[o];
#pragma unroll 8
      for (int i = 0; i < DD; ++i) a += wr[i] * hr[i];
      hn[p] = a > 0.f ? a : a * slope;
    }
    __syncthreads();
    for (int t = tid; t < BM * DD; t += 1024) {
      const float v = t < nb * DD ? hn[t] : 0.f;
      h[t] = v;
      if (t < nb * DD) acts[(long long)(l + 1) * B * DD + t] = v;
    }
  }
}
// backward of the chain; parameter gradients are ADDED into the tables' buffers (a null table entry = frozen layer).
template <int DD, int BM, int LMAX>
__global__ __launch_bounds__(1024) void mlp_chain_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ acts, const long long* wptr,
                                                             const long long* gwptr, const long long* gbptr, int L, int B, float slope,
                                                             float* __restrict__ dx) {
  constexpr int PER = DD * DD / 1024;
  __shared__ float dpre[BM * DD];               // d(pre-activation) of the current layer
  __shared__ float dnew[BM * DD];
  __shared__ float ha[(LMAX + 1) * BM * DD];    // every h_l, loaded once
  __shared__ float ws[DD * DD];                 // W of the current layer, [o][i]
  const int tid = threadIdx.x;
  float wn[PER], gold[PER];
  auto prefetch = [&](int l) {
    const float* W = reinterpret_cast<const float*>(wptr[l]);
    const float* gW = gwptr[l] ? reinterpret_cast<const float*>(gwptr[l]) : nullptr;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      wn[u] = W[tid + u * 1024];
      gold[u] = gW ? gW[tid + u * 1024] : 0.f;
    }
  };
  prefetch(L - 1);
  for (int t = tid; t < (L + 1) * BM * DD; t += 1024) {
    const int l = t / (BM * DD), r = t % (BM * DD);
    ha[t] = r < B * DD ? acts[(long long)l * B * DD + r] : 0.f;
  }
  for (int t = tid; t < BM * DD; t += 1024) dpre[t] = t < B * DD ? dout[t] : 0.f;
  for (int l = L - 1; l >= 0; --l) {
    __syncthreads();
    // dpre holds dL/dh_{l+1}: apply the LeakyReLU derivative (sign of h_{l+1} == sign of the pre-activation); stage W_l
    for (int t = tid; t < BM * DD; t += 1024) dpre[t] *= ha[(l + 1) * BM * DD + t] > 0.f ? 1.f : slope;
#pragma unroll
    for (int u = 0; u < PER; ++u) ws[tid + u * 1024] = wn[u];
    float gcur[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) gcur[u] = gold[u];
    float* gW = gwptr[l] ? reinterpret_cast<float*>(gwptr[l]) : nullptr;
    float* gb = (gbptr && gbptr[l]) ? reinterpret_cast<float*>(gbptr[l]) : nullptr;
    if (l > 0) prefetch(l - 1);
    __syncthreads();
    const float* hin = ha + l * BM * DD;
    if (gW) {
#pragma unroll
      for (int u = 0; u < PER; ++u) {            // dW[o][i] += sum_b dpre[b][o] * h_l[b][i]   (coalesced along i)
        const int t = tid + u * 1024;
        const int ii = t % DD, o = t / DD;
        float a = 0.f;
#pragma unroll
        for (int b = 0; b < BM; ++b) a = __fmaf_rn(dpre[b * DD + o], hin[b * DD + ii], a);      // (explicit FMA chain: the two-launch variant must produce the same bits)
        gW[t] = gcur[u] + a;
      }
    }
    if (gb && tid < DD) {
      float a = 0.f;
#pragma unroll
      for (int b = 0; b < BM; ++b) a += dpre[b * DD + tid];
      gb[tid] += a;
    }
    for (int p = tid; p < BM * DD; p += 1024) {  // dh_l[b][i] = sum_o dpre[b][o] * W[o][i]
      const int b = p / DD, ii = p % DD;
      float a = 0.f;
#pragma unroll 8
      for (int o = 0; o < DD; ++o) a = __fmaf_rn(dpre[b * DD + o], ws[o * DD + ii], a);
      dnew[p] = a;
    }
    __syncthreads();
    for (int t = tid; t < BM * DD; t += 1024) dpre[t] = dnew[t];
  }
  __syncthreads();
  if (dx)
    for (int t = tid; t < B * DD; t += 1024) dx[t] = dpre[t];
}

// The same backward pass as two launches (hwg_mlp_chain_bwd_split). The single-workgroup kernel above spends 69 us per call on ONE compute unit:
// per layer 512 LDS reads per thread (half of them for the parameter gradients) and 192 KB of weight / gradient traffic through one CU's
// memory port. Only the chain of data gradients is sequential: (1) one workgroup walks it (delta_l = (delta_{l+1} W_{l+1}) * lrelu') and leaves
// every delta_l in `deltas` [L][BM][D]; (2) L x D*D/1024 workgroups add dW_l = delta_l^T h_l and db_l into the tables' buffers in parallel.
// Every sum runs over the same terms in the same order as in the kernel above: bit-identical results.
template <int DD, int BM, int LMAX>
__global__ __launch_bounds__(1024) void mlp_chain_delta_kernel(const float* __restrict__ dout, const float* __restrict__ acts, const long long* wptr,
                                                               int L, int B, float slope, float* __restrict__ deltas, float* __restrict__ dx) {
  constexpr int PER = DD * DD / 1024;
  __shared__ float dpre[BM * DD];
  __shared__ float dnew[BM * DD];
  __shared__ float ws[DD * DD];
  const int tid = threadIdx.x;
  float wn[PER];
  auto prefetch = [&](int l) {
    const float* W = reinterpret_cast<const float*>(wptr[l]);
#pragma unroll
    for (int u = 0; u < PER; ++u) wn[u] = W[tid + u * 1024];
  };
  prefetch(L - 1);
  for (int t = tid; t < BM * DD; t += 1024) dpre[t] = t < B * DD ? dout[t] : 0.f;
  for (int l = L - 1; l >= 0; --l) {
    __syncthreads();
    for (int t = tid; t < BM * DD; t += 1024) {
      const float h = t < B * DD ? acts[(long long)(l + 1) * B * DD + t] : 0.f;
      const float d = dpre[t] * (h > 0.f ? 1.f : slope);
      dpre[t] = d;
      deltas[(long long)l * BM * DD + t] = d;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) ws[tid + u * 1024] = wn[u];
    if (l > 0) prefetch(l - 1);
    __syncthreads();
    for (int p = tid; p < BM * DD; p += 1024) {
      const int b = p / DD, ii = p % DD;
      float a = 0.f;
#pragma unroll 8
      for (int o = 0; o < DD; ++o) a = __fmaf_rn(dpre[b * DD + o], ws[o * DD + ii], a);
      dnew[p] = a;
    }
    __syncthreads();
    for (int t = tid; t < BM * DD; t += 1024) dpre[t] = dnew[t];
  }
  __syncthreads();
  if (dx)
    for (int t = tid; t < B * DD; t += 1024) dx[t] = dpre[t];
}
template <int DD, int BM>
__global__ __launch_bounds__(1024) void mlp_chain_pgrad_kernel(const float* __restrict__ deltas, const float* __restrict__ acts, const long long* gwptr,
                                                               const long long* gbptr, int B) {
  __shared__ float dl[BM * DD];
  __shared__ float hl[BM * DD];
  const int l = blockIdx.y, tid = threadIdx.x;
  float* gW = gwptr[l] ? reinterpret_cast<float*>(gwptr[l]) : nullptr;
  float* gb = (gbptr && gbptr[l]) ? reinterpret_cast<float*>(gbptr[l]) : nullptr;
  if (!gW && !gb) return;
  for (int t = tid; t < BM * DD; t += 1024) {
    dl[t] = deltas[(long long)l * BM * DD + t];
    hl[t] = t < B * DD ? acts[(long long)l * B * DD + t] : 0.f;
  }
  __syncthreads();
  const int t = blockIdx.x * 1024 + tid;            // element of dW_l [o][i]
  if (gW) {
    const int ii = t % DD, o = t / DD;
    float a = 0.f;
#pragma unroll
    for (int b = 0; b < BM; ++b) a = __fmaf_rn(dl[b * DD + o], hl[b * DD + ii], a);
    gW[t] = gW[t] + a;
  }
  if (gb && blockIdx.x == 0 && tid < DD) {
    float a = 0.f;
#pragma unroll
    for (int b = 0; b < BM; ++b) a += dl[b * DD + tid];
    gb[tid] += a;
  }
}

}  // namespace

extern "C" int hwg_gather_windows(const float* x, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int window, float* patches,
                                  void* stream) {
  HWG_REQUIRE(x && idx_b && idx_pos && patches && B > 0 && Wx > 0 && C > 0 && n > 0 && window >= 0, "gather_windows: bad arguments");
  const long long total = (long long)n * (2 * window + 1) * C;
  hipLaunchKernelGGL(gather_windows_kernel, dim3(hwg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, B, Wx, C, idx_b, idx_pos, n, window,
                     patches);
  HWG_LAUNCH_CHECK("gather_windows");
  return HWG_OK;
}
extern "C" int hwg_scatter_windows(const float* dpatches, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int window, int* win_of,
                                   float* dx, void* stream) {
  HWG_REQUIRE(dpatches && idx_b && idx_pos && win_of && dx && B > 0 && Wx > 0 && C > 0 && n > 0 && window >= 0, "scatter_windows: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(win_of, 0xFF, (size_t)B * Wx * sizeof(int), st) != hipSuccess) {
    hwg_set_error("scatter_windows: memset failed");
    return HWG_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(window_index_kernel, dim3(hwg_cdiv(n, 256)), dim3(256), 0, st, idx_b, idx_pos, n, B, Wx, win_of);
  hipLaunchKernelGGL(scatter_windows_kernel, dim3(hwg_stream_grid((long long)B * Wx * C, 256)), dim3(256), 0, st, dpatches, B, Wx, C, (const int*)win_of,
                     window, dx);
  HWG_LAUNCH_CHECK("scatter_windows");
  return HWG_OK;
}
extern "C" int hwg_segment_weighted_mean(const float* v, const float* wgt, const int* seg, int n, int C, int B, float* out, float* wsum, void* stream) {
  HWG_REQUIRE(v && wgt && seg && out && wsum && n > 0 && C > 0 && B > 0, "segment_weighted_mean: bad arguments");
  const size_t lds = ((size_t)2 * n + 1) * sizeof(int);
  if (lds <= 60 * 1024)
    hipLaunchKernelGGL(segment_weighted_mean_list_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, v, wgt, seg, n, C, out, wsum);
  else
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
  HWG_REQUIRE(x && wptr && O && first_wave && off && y && L > 0 && B > 0 && B <= 65535 * LB_MAXB && I > 0 && halves > 0 && total_outputs > 0,
              "linear_bank_fwd: bad arguments");
  hipLaunchKernelGGL(linear_bank_fwd_kernel, dim3(hwg_cdiv(total_outputs, 4), hwg_cdiv(B, LB_MAXB)), dim3(256), 0, (hipStream_t)stream, x, (const long long*)wptr,
                     (const long long*)bptr, O, first_wave, (const long long*)off, L, B, I, halves, y);
  HWG_LAUNCH_CHECK("linear_bank_fwd");
  return HWG_OK;
}
extern "C" size_t hwg_linear_bank_bwd_workspace(int total_outputs, int B, int I) {
  return (size_t)hwg_cdiv(total_outputs > 0 ? total_outputs : 1, LB_OCHUNK) * (B > 0 ? B : 0) * (I > 0 ? I : 0) * sizeof(float);
}
extern "C" int hwg_linear_bank_bwd(const float* x, const void* dyptr, const void* wptr, const void* gwptr, const void* gbptr, const int* O,
                                   const int* first_wave, int L, int B, int I, int halves, int total_outputs, float* dx, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  HWG_REQUIRE(x && dyptr && wptr && gwptr && O && first_wave && L > 0 && B > 0 && B <= LB_MAXB && I > 0 && halves > 0 && total_outputs > 0,
              "linear_bank_bwd: bad arguments (B <= %d)", LB_MAXB);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(linear_bank_wgrad_kernel, dim3(hwg_cdiv(total_outputs, 4)), dim3(256), 0, st, x, (const long long*)dyptr, (const long long*)gwptr,
                     (const long long*)gbptr, O, first_wave, L, B, I, halves);
  HWG_LAUNCH_CHECK("linear_bank_wgrad");
  if (dx) {
    const size_t need = hwg_linear_bank_bwd_workspace(total_outputs, B, I);
    if (!workspace || workspace_bytes < need) {
      hwg_set_error("linear_bank_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
      return HWG_ERR_WORKSPACE;
    }
    const int nch = hwg_cdiv(total_outputs, LB_OCHUNK);
    hipLaunchKernelGGL(linear_bank_dgrad_kernel, dim3(nch), dim3(1024), 0, st, (const long long*)dyptr, (const long long*)wptr, O, first_wave, L, B, I, halves,
                       (float*)workspace);
    HWG_LAUNCH_CHECK("linear_bank_dgrad");
    hipLaunchKernelGGL(linear_bank_dgrad_reduce_kernel, dim3(hwg_cdiv(B * I, 256)), dim3(256), 0, st, (const float*)workspace, nch, B * I, dx);
    HWG_LAUNCH_CHECK("linear_bank_dgrad_reduce");
  }
  return HWG_OK;
}

extern "C" int hwg_mlp_chain_fwd(const float* x, const void* wptr, const void* bptr, int L, int B, int D, float slope, float* acts, void* stream) {
  HWG_REQUIRE(x && wptr && bptr && acts && L > 0 && B > 0 && D > 0 && D <= MC_MAXD, "mlp_chain_fwd: bad arguments (D <= %d)", MC_MAXD);
  hipStream_t st = (hipStream_t)stream;
  HWG_REQUIRE(D == 64 || D == 128, "mlp_chain_fwd: D must be 64 or 128 (got %d)", D);
  // forward: any number of rows, one workgroup per 8 / 16 of them (the backward kernel keeps the whole batch in one workgroup: B <= 16)
#define HWG_MC_FWD(DD_, BM_) hipLaunchKernelGGL((mlp_chain_fwd_kernel<DD_, BM_>), dim3(hwg_cdiv(B, BM_)), dim3(1024), 0, st, x, (const long long*)wptr, (const long long*)bptr, L, B, slope, acts)
  if (D == 128 && B <= 8) HWG_MC_FWD(128, 8);
  else if (D == 128) HWG_MC_FWD(128, 16);
  else HWG_MC_FWD(64, 16);
#undef HWG_MC_FWD
  HWG_LAUNCH_CHECK("mlp_chain_fwd");
  return HWG_OK;
}
extern "C" int hwg_mlp_chain_bwd(const float* dout, const float* acts, const void* wptr, const void* gwptr, const void* gbptr, int L, int B, int D,
                                 float slope, float* dx, void* stream) {
  HWG_REQUIRE(dout && acts && wptr && gwptr && L > 0 && L <= 8 && B > 0 && B <= MC_MAXB && (D == 64 || D == 128),
              "mlp_chain_bwd: bad arguments (L <= 8, B <= %d, D 64 or 128)", MC_MAXB);
  hipStream_t st = (hipStream_t)stream;
#define HWG_MC_BWD(DD_, BM_) hipLaunchKernelGGL((mlp_chain_bwd_kernel<DD_, BM_, 8>), dim3(1), dim3(1024), 0, st, dout, acts, (const long long*)wptr, \
                                                (const long long*)gwptr, (const long long*)gbptr, L, B, slope, dx)
  if (D == 128 && B <= 8) HWG_MC_BWD(128, 8);
  else if (D == 128) HWG_MC_BWD(128, 16);
  else if (B <= 8) HWG_MC_BWD(64, 8);
  else HWG_MC_BWD(64, 16);
#undef HWG_MC_BWD
  HWG_LAUNCH_CHECK("mlp_chain_bwd");
  return HWG_OK;
}
extern "C" size_t hwg_mlp_chain_bwd_workspace(int L, int B, int D) { return (size_t)(L > 0 ? L : 0) * 16 * (D > 0 ? D : 0) * sizeof(float); }
extern "C" int hwg_mlp_chain_bwd_split(const float* dout, const float* acts, const void* wptr, const void* gwptr, const void* gbptr, int L, int B,
                                       int D, float slope, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
  HWG_REQUIRE(dout && acts && wptr && gwptr && L > 0 && L <= 8 && B > 0 && B <= MC_MAXB && (D == 64 || D == 128),
              "mlp_chain_bwd_split: bad arguments (L <= 8, B <= %d, D 64 or 128)", MC_MAXB);
  if (!workspace || workspace_bytes < hwg_mlp_chain_bwd_workspace(L, B, D)) { hwg_set_error("mlp_chain_bwd_split: workspace too small"); return HWG_ERR_WORKSPACE; }
  hipStream_t st = (hipStream_t)stream;
  float* deltas = (float*)workspace;
#define HWG_MC_SPLIT(DD_, BM_)                                                                                                              \
  do {                                                                                                                                      \
    hipLaunchKernelGGL((mlp_chain_delta_kernel<DD_, BM_, 8>), dim3(1), dim3(1024), 0, st, dout, acts, (const long long*)wptr, L, B, slope, deltas, dx); \
    hipLaunchKernelGGL((mlp_chain_pgrad_kernel<DD_, BM_>), dim3(DD_ * DD_ / 1024, L), dim3(1024), 0, st, (const float*)deltas, acts,         \
                       (const long long*)gwptr, (const long long*)gbptr, B);                                                                 \
  } while (0)
  if (D == 128 && B <= 8) HWG_MC_SPLIT(128, 8);
  else if (D == 128) HWG_MC_SPLIT(128, 16);
  else if (B <= 8) HWG_MC_SPLIT(64, 8);
  else HWG_MC_SPLIT(64, 16);
#undef HWG_MC_SPLIT
  HWG_LAUNCH_CHECK("mlp_chain_bwd_split");
  return HWG_OK;
}
