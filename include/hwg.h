/*
 * hwg.h — C-ABI of libhwg_hip.so, the MI355X (gfx950) kernel library behind the
 * GAN training hot path of herobd/handwriting_line_generation.
 *
 * The reference has no FFI layer: every op below replaces an ATen
 * call made by the reference's Python modules; the reference call site each entry
 * point stands in for is cited next to it (paths relative to the reference root).
 *
 * Conventions
 *  - all tensors are device pointers to contiguous fp32, activations are NHWC
 *    ([N,H,W,C], C fastest) unless stated otherwise; integer tensors are int32/int64
 *    as stated;
 *  - the caller owns every buffer (including workspaces); the library never
 *    allocates device memory and never synchronises, all work is enqueued on
 *    `stream` (a hipStream_t passed as void*);
 *  - every function returns HWG_OK (0) or a negative hwg_status; the message is
 *    available from hwg_last_error() (thread local). Nothing throws across the ABI.
 */
#ifndef HWG_H_
#define HWG_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hwg_status {
  HWG_OK = 0,
  HWG_ERR_ARG = -1,      /* unsupported / inconsistent arguments */
  HWG_ERR_LAUNCH = -2,   /* hipLaunch / runtime failure */
  HWG_ERR_WORKSPACE = -3 /* workspace too small */
} hwg_status;

const char* hwg_last_error(void);
/* library/ABI version, bumped when a signature changes */
int hwg_abi_version(void);
/* 1 when a HIP device is present and usable */
int hwg_device_ok(void);
/* Schedules (tile, split factor, kernel variant) are planned once per geometry and cached per thread; the tuning knobs of the
 * environment (HWG_WINO, HWG_WINO_FORCE, HWG_CONV_FORCE, HWG_WGRAD_FORCE, HWG_WINO_WGRAD, HWG_WGRAD_NARROW, ...) are read at plan time.
 * A process that changes them afterwards calls hwg_tuning_reload() (drops every cached plan). hwg_last_plan() reports what the process's
 * last convolution-family launch ran (any thread: backward passes launch from the autograd engine's thread): out[0] engine (the hwg_prof_stop kind codes), out[1] schedule id, out[2] split factor -
 * how the parity tests assert that a forced schedule was the one launched. */
int hwg_tuning_reload(void);
int hwg_last_plan(int* engine_cfg_nsplit);
/* Stream dependencies for kernels launched off the caller's main stream (weight gradients on a side stream: nothing in a backward pass
 * waits for them, only the optimizer does). fork: `side_stream` waits for everything enqueued on `main_stream` so far; join: `main_stream`
 * waits for everything enqueued on `side_stream` so far. Host cost: one event record + one stream wait on a reused event. */
int hwg_stream_fork(void* main_stream, void* side_stream);
int hwg_stream_join(void* side_stream, void* main_stream);

/* Launch profiler for the matrix-core kernels (bench.py's roofline measurement): between hwg_prof_start() and hwg_prof_stop() every
 * MFMA convolution / weight-gradient launch (and their reduce passes) is bracketed by a HIP event pair on its stream. hwg_prof_tag()
 * labels the launches of the calling thread's next calls. hwg_prof_stop() waits for the recorded events and returns, per launch,
 * kind (0 conv MFMA, 1 wgrad MFMA, 2 conv split reduce, 3 wgrad reduce, 4 direct conv, 5 direct wgrad incl. its reduce, 6 Winograd conv, 7 Winograd wgrad), tag, algorithmic work (flops; bytes for the reduce passes) and ms.
 * Returns the number of records written. */
int hwg_prof_start(int max_records);
int hwg_prof_enable(int on);   /* pause / resume recording inside an open profile (sampled profiling: the event pairs cost ~7 % of a step) */
int hwg_prof_tag(int tag);
int hwg_prof_stop(int* kinds, int* tags, double* work, float* ms, int capacity);

/* activation codes used by fused epilogues */
enum { HWG_ACT_NONE = 0, HWG_ACT_RELU = 1, HWG_ACT_LRELU = 2, HWG_ACT_TANH = 3 };

/* ------------------------------------------------------------------------------------------
 * Convolution engine (implicit GEMM on fp32 MFMA, direct kernels for 1-channel ends).
 * Replaces nn.Conv2d / nn.Conv1d / nn.ConvTranspose2d / F.conv_transpose2d / nn.Linear at
 * model/pure_gen.py:161-197,268-278,286; model/discriminator_ap.py:77-131;
 * model/cnn_only_hwr.py:31-32,78-92; model/char_style.py:65-71,90-93,162-168;
 * model/count_cnn.py:12-23; model/autoencoder.py:307-330,346-395,601-617.
 * ------------------------------------------------------------------------------------------ */
typedef struct hwg_conv_desc {
  int N, H, W, C;             /* gathered tensor  [N,H,W,C]            */
  int K, R, S;                /* K channels on the anchor grid, RxS taps */
  int stride_h, stride_w;
  int pad_h, pad_w;
  int dil_h, dil_w;
  int P, Q;                   /* anchor grid     [N,P,Q,K]             */
  int transposed;             /* 0: anchor(p,q) <-> gathered(p*stride-pad+r*dil)   (convolution)
                                 1: fractionally strided gather (conv-transpose with stride>1), dil must be 1:
                                    out(p,q) += x(ih,iw) * w(r,s)  where  ih*stride-pad+r == p            */
} hwg_conv_desc;

/* Re-layout a weight tensor into the engine's [R*S][A][Bpad] form (b fastest, zero padded to Bpad).
 * src element (a,b,r,s) lives at src[a*sa + b*sb + r*sr + s*ss]; flip!=0 mirrors the taps. */
int hwg_conv_pack_weight(const float* src, float* dst, int A, int B, int Bpad, int R, int S,
                         long long sa, long long sb, long long sr, long long ss, int flip, void* stream);

/* The same re-layout for many weights in one launch (after an optimizer step). table: device array of n_entries records
 * { const float* src; float* dst; int A, B, Bpad, R, S, flip; long long sa, sb, sr, ss, total, first_block; int mode, Apad; }
 * (96 bytes) where total = R*S*A*Bpad and first_block = running sum of ceil(total/1024); total_blocks = that sum over all entries.
 * mode 1 = the Winograd filter transform of hwg_wino_pack_weight (then Apad = ceil16(A), Bpad = ceil16(B), total = Apad*Bpad);
 * mode 2 = hwg_wino_s2_pack_weight (A = Kc, B = Cc, sa = sk, sb = sc, flip = dgrad; Apad / Bpad = the image's padded extents, total = Apad*Bpad). */
int hwg_conv_pack_weight_multi(const void* table, int n_entries, long long total_blocks, void* stream);
/* ... and for weights that are a device-side scalar multiple of a stored tensor: the spectral-norm layers' W_bar / sigma
 * (model/discriminator_ap.py:31-32, recomputed on every forward pass). Records of 112 bytes: the 96-byte record above followed by
 * { long long dst_off; int scale_idx; int pad; }: the image goes to dst_base + dst_off floats (the record's own dst is ignored), every source
 * element is multiplied (rounded product) by scale_base[scale_idx] - all images a pass through the network needs, one launch per forward. */
int hwg_conv_pack_weight_multi_scaled(const void* table, int n_entries, long long total_blocks, float* dst_base, const float* scale_base, void* stream);

/* y[N,P,Q,K] = gather-conv(x[N,H,W,C], w[R*S][K][C]) (+ bias[K] if bias != NULL).
 * transposed==0: standard convolution. transposed==1: conv-transpose with stride>1.
 * accumulate!=0 adds into y instead of overwriting it.
 * Layers with few output pixels but a long contraction (taps x channels) are scheduled split-K: partial outputs go to the
 * caller's workspace (hwg_conv_fwd_workspace() bytes, 0 when the schedule does not split) and are summed in a fixed order. */
size_t hwg_conv_fwd_workspace(const hwg_conv_desc* d);
int hwg_conv_fwd(const hwg_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                 int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Data gradient of a convolution whose input has ONE channel (first layers; stride 1), second half: the caller first forms the
 * per-pixel tap matrix t[N,P,Q,R*S] = dy[N,P,Q,K] x W[K][R*S] with hwg_conv_fwd (1x1, on the matrix cores), then
 * dx[n,ih,iw] = sum_{r,s} t[n, ih+pad_h-r*dil_h, iw+pad_w-s*dil_w, r*S+s]. */
int hwg_col2im_taps(const float* t, float* dx, int N, int H, int W, int P, int Q, int R, int S, int pad_h, int pad_w, int dil_h, int dil_w,
                    void* stream);

/* weight gradient: dw(k,c,r,s) = sum_{n,p,q} u[n,p,q,k] * v[n, p*stride-pad+r*dil, q*stride-pad+s*dil, c]
 * written to dw[k*sa + c*sb + r*sr + s*ss] (so it lands directly in the PyTorch parameter layout).
 * u is the tensor living on the anchor grid [N,P,Q,K], v the gathered one [N,H,W,C]; d->transposed is ignored.
 * accumulate!=0 adds into dw. dbias (may be NULL) additionally receives the column sums of u, dbias[k] = sum_{n,p,q} u[n,p,q,k]
 * (the bias gradient when u is the output gradient), taken from the u tiles the kernel stages anyway; bias_accumulate!=0 adds.
 * Not available on the direct path (K <= 2 or C <= 2 without the taps-as-N kernel): use hwg_colsum there. */
size_t hwg_conv_wgrad_workspace(const hwg_conv_desc* d);
int hwg_conv_wgrad(const hwg_conv_desc* d, const float* u, const float* v, float* dw,
                   long long sa, long long sb, long long sr, long long ss, int accumulate,
                   float* dbias, int bias_accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Gradient sets: `sets` weight gradients of one layer whose anchors differ (u / dy = the sets' [N,P,Q,K] tensors back to back) and whose
 * gathered tensor (v / x, [N,H,W,C]) is shared - the two or three upstream gradients a balanced lesson sends through the generator
 * (reference trainer/hw_with_style_trainer.py:300-338 runs one backward() per loss group on the same activations). One launch; set s is
 * summed into the buffers dw_ptrs[s] / dbias_ptrs[s] (host arrays of device addresses). d describes ONE set; workspace =
 * hwg_*_wgrad_sets_workspace(d, sets) bytes (the pixel ranges are planned for the whole launch and shared out over the sets). set_on_v != 0 (hwg_conv_wgrad_sets only): the sets differ in the GATHERED tensor v instead and share the anchor u
 * (transposed layers, whose anchor is the layer input); no fused bias gradient then. hwg_conv_wgrad_sets_supported: the layer runs on the MFMA path (not the K <= 2 / C <= 2 direct kernels). */
int hwg_conv_wgrad_sets_supported(const hwg_conv_desc* d);
size_t hwg_conv_wgrad_sets_workspace(const hwg_conv_desc* d, int sets);
size_t hwg_wino_wgrad_sets_workspace(const hwg_conv_desc* d, int sets);
int hwg_conv_wgrad_sets(const hwg_conv_desc* d, const float* u, const float* v, int sets, int set_on_v, const long long* dw_ptrs,
                        long long sa, long long sb, long long sr, long long ss, int accumulate,
                        const long long* dbias_ptrs, int bias_accumulate, void* workspace, size_t workspace_bytes, void* stream);
int hwg_wino_wgrad_sets(const hwg_conv_desc* d, const float* dy, const float* x, int sets, const long long* dw_ptrs, long long sa,
                        long long sb, long long sr, long long ss, int accumulate, const long long* dbias_ptrs, int bias_accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Deferred sum of the weight-gradient partial images (reference: the per-parameter gradient accumulation autograd does after every backward
 * node, trainer/hw_with_style_trainer.py:300-338 reads the results only after the pass). hwg_conv_wgrad / hwg_wino_wgrad write one partial
 * image per pixel range into the workspace and sum them into dw (+dbias) with a launch of their own - 65 such launches per training step, most
 * too small to stream at memory rate. hwg_wgrad_defer_next() marks the calling thread's NEXT weight-gradient call: its sum is queued instead of
 * launched, and the caller must keep that call's workspace alive and untouched (its own arena, not a shared scratch buffer) until
 * hwg_wgrad_defer_flush(stream, &launches) has been enqueued, which sums everything queued so far (any thread) with one table-driven launch
 * per 32 gradients: every gradient runs the schedule it would have run alone (bit-identical results), several gradients of the same tensor
 * are summed in queue order by consecutive launches. dw / dbias are undefined between the call and the flush. The mark is consumed by the next
 * hwg_conv_wgrad / hwg_wino_wgrad call whatever path it takes (paths without partial images simply ignore it). */
int hwg_wgrad_defer_next(void);
long long hwg_wgrad_defer_pending(void);
int hwg_wgrad_defer_flush(void* stream, int* launches);

/* Winograd F(2x2,3x3) path for 3x3 / stride 1 / dilation 1 convolutions with C % 16 == 0 and K >= 16 (same call sites as
 * hwg_conv_fwd: the 3x3 layers of model/discriminator_ap.py:84-131, model/cnn_only_hwr.py:31-32, model/pure_gen.py:161-197,
 * model/char_style.py:65-71, model/autoencoder.py:346-395 and their data gradients). Input, filter and output transforms are
 * fused into the kernel; the 16 transform-domain GEMMs run on v_mfma_f32_16x16x4_f32 (2.25x fewer MACs than the direct form).
 * hwg_wino_pack_weight turns a weight (element (a,b,r,s) at src[a*sa+b*sb+r*sr+s*ss], a = output channel of THIS product,
 * b = contracted channel, flip mirrors the taps) into U = G g G^T laid out [ceil16(B)/16][16][ceil16(A)][16]
 * (hwg_wino_weight_floats(A, B) floats). d describes the product as for hwg_conv_fwd (transposed must be 0). */
int hwg_wino_supported(const hwg_conv_desc* d);
/* 1 when the library's cost models put the Winograd schedule ahead of the direct one for this product (callers pick the weight image accordingly) */
int hwg_wino_preferred(const hwg_conv_desc* d);
size_t hwg_wino_weight_floats(int A, int B);
int hwg_wino_pack_weight(const float* src, float* dst, int A, int B, long long sa, long long sb, long long sr, long long ss,
                         int flip, void* stream);
size_t hwg_wino_conv_workspace(const hwg_conv_desc* d);
int hwg_wino_conv_fwd(const hwg_conv_desc* d, const float* x, const float* u, const float* bias, float* y,
                      int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* 4x4 stride-2 pad-0 dilation-1 convolutions (d->transposed = 0; reference: the style extractor's down-sampling convolutions,
 * model/char_style.py:154 Conv2dBlock(dim, 2*dim, 4, 2, 1): a pad layer followed by nn.Conv2d(.., 4, 2), :69-71) and their data gradients (d->transposed = 1, described like the
 * fractionally strided product of hwg_conv_fwd: input = dy [N,H,W,C], output = dx [N,P,Q,K] with P = 2H + 2, Q = 2W + 2) as stride-1 two-tap
 * convolutions on the space-to-depth image in the Winograd domain F(3x3, 2x2) (conv_wino.hip). The filter image (hwg_wino_s2_weight_floats
 * floats) is packed from the CONVOLUTION's weight: element (output channel kc, input channel cc, r, s) at src[kc*sk + cc*sc + r*4 + s];
 * dgrad = 1 packs the image of the data-gradient product. hwg_wino_s2_preferred: 1 when the cost models put it ahead of the direct kernels. */
int hwg_wino_s2_supported(const hwg_conv_desc* d);
int hwg_wino_s2_preferred(const hwg_conv_desc* d);
size_t hwg_wino_s2_weight_floats(int Kc, int Cc, int dgrad);
int hwg_wino_s2_pack_weight(const float* src, float* dst, int Kc, int Cc, long long sk, long long sc, int dgrad, void* stream);
size_t hwg_wino_s2_workspace(const hwg_conv_desc* d);
int hwg_wino_s2_conv(const hwg_conv_desc* d, const float* x, const float* u, const float* bias, float* y,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* (Their weight gradient goes through hwg_wino_wgrad* below, which accept these descriptors as well.) */

/* Diagnostic, like hwg_last_plan: the schedule hwg_wino_conv_fwd (3x3 stride 1) / hwg_wino_s2_conv (4x4 stride 2 pad 0, either direction) would
 * run for this product, without launching anything. out[8] = {tile config, uniform channel split, balanced tail workgroups (0 = uniform
 * schedule), whole-tile lead workgroups, most pieces a cut tile is written in (= workspace images), workgroup tiles, channel chunks, most tiles
 * one tail workgroup touches}; all -1 when the product is not supported. */
int hwg_wino_conv_describe(const hwg_conv_desc* d, int* out);

/* Weight gradient of a 3x3 stride-1 dilation-1 convolution in the Winograd domain F(3x3, 2x2) - or of a 4x4 stride-2 pad-0 one as the two-tap
 * problem on the space-to-depth image, F(2x2 taps, 3x3 gradient tiles) - (conv_wino_wgrad.hip): same operands and weight
 * strides as hwg_conv_wgrad (dy = anchor [N,P,Q,K], x = gathered [N,H,W,C]; reference: the weight gradients autograd produces for
 * model/pure_gen.py, model/discriminator_ap.py and model/cnn_only_hwr.py's 3x3 layers). dbias (or null): the bias gradient, column sums of dy,
 * taken from the dy tiles on their way through the kernel. */
int hwg_wino_wgrad_supported(const hwg_conv_desc* d);
int hwg_wino_wgrad_preferred(const hwg_conv_desc* d);
size_t hwg_wino_wgrad_workspace(const hwg_conv_desc* d);
int hwg_wino_wgrad(const hwg_conv_desc* d, const float* dy, const float* x, float* dw, long long sa, long long sb, long long sr, long long ss,
                   int accumulate, float* dbias, int bias_accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* out[C] (+)= sum over rows of x[rows][C]  (bias gradients, channel sums) */
size_t hwg_colsum_workspace(long long rows, int C);
int hwg_colsum(const float* x, long long rows, int C, float* out, int accumulate,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation + activation epilogues (NHWC, x is [N][HW][C]).
 * mode: 0 InstanceNorm, 1 GroupNorm(groups), 2 BatchNorm in training mode (batch statistics,
 * running stats updated with `momentum` when running_mean != NULL).
 * y = act(mask[n,c] * (gamma * xhat + beta)); gamma/beta are [C] or, with affine_per_sample, [N][C].
 * Replaces nn.GroupNorm (+Dropout2d +ReLU/LeakyReLU) at model/discriminator_ap.py:78-79,103-104,
 * model/autoencoder.py:346-395,307-330, model/char_style.py:41,91,100,166, model/count_cnn.py:13-21;
 * nn.BatchNorm2d/1d at model/cnn_only_hwr.py:36,80-89; nn.InstanceNorm2d at model/pure_gen.py:56.
 * mean/rstd are [N][C] outputs kept for the backward pass.
 * hwg_norm_bwd takes the forward call's gamma / beta: with relu / leaky relu fused it recomputes the activation gate (the sign of the
 * pre-activation, bit-identical to the forward pass) from x instead of reading y - `y` may be NULL then; tanh needs y.
 * CONSTRAINT: gamma / beta must still hold the forward call's values (no optimizer step or in-place write between forward and backward);
 * ops._Norm stamps the parameters' version / optimizer epoch at forward time and refuses a backward pass after a change.
 * ------------------------------------------------------------------------------------------ */
size_t hwg_norm_workspace(int N, int HW, int C);
int hwg_norm_fwd(const float* x, float* y, int N, int HW, int C, int mode, int groups, float eps,
                 const float* gamma, const float* beta, int affine_per_sample, const float* chan_mask,
                 int act, float slope, float* mean, float* rstd, float* running_mean, float* running_var,
                 float momentum, void* ws, size_t ws_bytes, void* stream);
int hwg_norm_bwd(const float* dy, const float* x, const float* y, float* dx, int N, int HW, int C, int mode, int groups,
                 const float* gamma, const float* beta, int affine_per_sample, const float* chan_mask, int act, float slope,
                 const float* mean, const float* rstd, float* dgamma, float* dbeta, int accumulate,
                 void* ws, size_t ws_bytes, void* stream);

/* BatchNorm in eval mode (running statistics); mean/rstd are [N][C] scratch outputs */
int hwg_norm_frozen_fwd(const float* x, float* y, int N, int HW, int C, const float* running_mean, const float* running_var, float eps,
                        const float* gamma, const float* beta, int act, float slope, float* mean, float* rstd, void* stream);

/* Generator epilogue, model/pure_gen.py:205-214 (NoiseInjection -> LeakyReLU -> AdaptiveInstanceNorm):
 *   u = lrelu(x + noise_w[c]*noise_scale*noise, slope);  y = gamma[n,c] * IN(u) + beta[n,c]
 * backward also returns the conv-bias gradient (sum of d/dx) and the raw noise-weight gradient. */
int hwg_adain_fwd(const float* x, const float* noise, const float* noise_w, float noise_scale, float slope,
                  const float* gamma, const float* beta, float eps, float* u, float* y, float* mean, float* rstd,
                  int N, int HW, int C, void* ws, size_t ws_bytes, void* stream);
/* the same with the noise drawn in the kernel (forward-only calls): element i of the [N][HW][C] tensor takes the value hwg_randn(seed, offset)
 * writes to element i of a tensor of that size; no noise tensor exists. u may alias x. */
int hwg_adain_fwd_rng(const float* x, unsigned long long seed, unsigned long long offset, const float* noise_w, float noise_scale, float slope,
                      const float* gamma, const float* beta, float eps, float* u, float* y, float* mean, float* rstd, int N, int HW, int C,
                      void* ws, size_t ws_bytes, void* stream);
int hwg_adain_bwd(const float* dy, const float* u, const float* noise, float noise_scale, float slope,
                  const float* gamma, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                  float* dnoise_w, float* dbias, int accumulate_params, int N, int HW, int C,
                  void* ws, size_t ws_bytes, void* stream);

/* y = act(mask[row/HW][c] * (x + bias[c])) on x[rows][C]; backward takes y (nn.ReLU / nn.LeakyReLU / nn.Dropout2d sites) */
int hwg_bias_act_fwd(const float* x, const float* bias, const float* chan_mask, float* y, long long rows, int HW, int C,
                     int act, float slope, void* stream);
int hwg_bias_act_bwd(const float* dy, const float* y, const float* chan_mask, float* dx, long long rows, int HW, int C,
                     int act, float slope, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling / resampling / padding / concatenation (NHWC).
 * nn.AvgPool2d (discriminator_ap.py:88-127, autoencoder.py:350-389), nn.MaxPool2d/1d (cnn_only_hwr.py:45-55,
 * char_style.py:164), nn.Upsample nearest (pure_gen.py:176-178), Blur (pure_gen.py:80-137),
 * F.pad / ReplicationPad2d (char_style.py:19-21,198-202; trainer :590-595,727-737,771-795), torch.cat.
 * ------------------------------------------------------------------------------------------ */
int hwg_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int kh, int kw, void* stream);
int hwg_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int kh, int kw, void* stream);
/* y = avgpool(act(chan_mask[n][c] * x)) in one pass, and its backward (gate recomputed from x): model/discriminator_ap.py:84-131, the
 * SN conv -> Dropout2d -> LeakyReLU -> AvgPool2d runs of the discriminator; bit-identical to hwg_bias_act_fwd + hwg_avgpool_fwd (resp. their
 * backward kernels) without the full-resolution activation / gradient round trips. act: none / relu / leaky relu; chan_mask may be NULL */
int hwg_act_avgpool_fwd(const float* x, const float* chan_mask, float* y, int N, int H, int W, int C, int kh, int kw, int act, float slope,
                        void* stream);
int hwg_act_avgpool_bwd(const float* dy, const float* x, const float* chan_mask, float* dx, int N, int H, int W, int C, int kh, int kw,
                        int act, float slope, void* stream);
int hwg_maxpool_fwd(const float* x, float* y, int* idx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                    int P, int Q, void* stream);
int hwg_maxpool_bwd(const float* dy, const int* idx, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                    int pw, int P, int Q, void* stream);
/* max-pool with the ReLU that precedes it in the reference riding along (model/cnn_only_hwr.py:31-43: conv -> ReLU -> MaxPool2d):
 * y = relu(maxpool(x)) == maxpool(relu(x)) exactly; backward: dx = scatter(dy * [y > 0]) - same bits as the separate ReLU passes */
int hwg_maxpool_relu_fwd(const float* x, float* y, int* idx, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                         int P, int Q, void* stream);
int hwg_maxpool_relu_bwd(const float* dy, const float* y, const int* idx, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                         int ph, int pw, int P, int Q, void* stream);
int hwg_upsample_nearest_fwd(const float* x, float* y, int N, int H, int W, int C, int fh, int fw, void* stream);
int hwg_upsample_nearest_bwd(const float* dy, float* dx, int N, int H, int W, int C, int fh, int fw, void* stream);
int hwg_blur3(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* mode 0 constant(value) (negative pads crop), 1 replicate */
int hwg_pad2d_fwd(const float* x, float* y, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, float value, void* stream);
int hwg_pad2d_bwd(const float* dy, float* dx, int N, int H, int W, int C, int pt, int pb, int pl, int pr, int mode, void* stream);
/* dst[row][doff+c] (+)= src[bcast ? row/HW : row][soff+c], c < Cn */
int hwg_copy_channels(const float* src, int Cs, int soff, float* dst, int Cd, int doff, int Cn, long long rows, int HW, int bcast,
                      int accumulate, void* stream);
/* dst[row][c] = c < C ? src[row][c] : 0 for c < Cpad (Cpad % 4 == 0): the zero-padded channel copy the MFMA paths take when a contraction
 * side is not a multiple of their channel step (model/cnn_only_hwr.py:92: the RIMES recogniser's 78 classes) - copy and fill in ONE launch
 * (it was a torch-side zero fill plus hwg_copy_channels; the fill was invisible to a recorded call list) */
int hwg_pad_channels(const float* src, int C, float* dst, int Cpad, long long rows, void* stream);
/* out[n][c] (+)= sum_hw src[n*HW+hw][soff+c] */
int hwg_reduce_rows(const float* src, int Cs, int soff, float* out, int Cn, int N, int HW, int accumulate, void* stream);
/* label [L][B] int32 -> out[b][l][doff + cls] one-hot rows of width ncls inside rows of width Cd (HWWithStyle.onehot, hw_with_style.py:333-337) */
int hwg_onehot(const int* label, float* out, int L, int B, int ncls, int Cd, int doff, void* stream);
/* the same one-hot in both layouts at once: out_blc [B][L][ncls] (NHWC rows, what the networks read) and out_lbc [L][B][ncls] (the reference's
 * time-major tensor, hw_with_style.py:333-337) - one launch instead of a one-hot and two permutes */
int hwg_onehot_both(const int* label, float* out_blc, float* out_lbc, int L, int B, int ncls, void* stream);
/* FusedUpsample weight transform, model/pure_gen.py:268-276: [A][B][3][3] -> [A][B][4][4] (AB = A*B) and its adjoint */
int hwg_fused_upsample_weight_fwd(const float* w3, float* w4, long long AB, float mult, void* stream);
int hwg_fused_upsample_weight_bwd(const float* dw4, float* dw3, long long AB, float mult, void* stream);
/* the adjoint ADDED into dw3 (the parameter's gradient buffer): saves the add launch behind it */
int hwg_fused_upsample_weight_bwd_acc(const float* dw4, float* dw3, long long AB, float mult, void* stream);
int hwg_permute4(const float* in, float* out, int d0, int d1, int d2, int d3, long long s0, long long s1, long long s2, long long s3, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sequence ops: LogSoftmax (cnn_only_hwr.py:92, autoencoder.py:617), F.ctc_loss (model/loss.py:28-30),
 * DTW alignment correct_pred (model/hw_with_style.py:18-74), gt-count scan (trainer :670-697).
 * ------------------------------------------------------------------------------------------ */
/* rows of x are (b,t) ordered; with transpose_bt the output rows are (t,b) ordered ([T][B][C], what F.ctc_loss takes) */
int hwg_log_softmax_fwd(const float* x, float* y, long long rows, int C, int B, int T, int transpose_bt, void* stream);
int hwg_log_softmax_bwd(const float* dy, const float* y, float* dx, long long rows, int C, int B, int T, int transpose_bt, void* stream);
/* log_probs [T][B][C]; targets [B][Lmax] int32 (blank = 0); reduction 'mean'; an infinite mean is reported as 0.
 * hwg_ctc_bwd must be called with the same workspace contents hwg_ctc_fwd left behind. */
size_t hwg_ctc_workspace(int T, int B, int Lmax);
int hwg_ctc_fwd(const float* log_probs, const int* targets, const int* input_lengths, const int* target_lengths, int T, int B, int C,
                int Lmax, float* loss, void* ws, size_t ws_bytes, void* stream);
int hwg_ctc_bwd(const float* log_probs, const int* targets, const int* input_lengths, const int* target_lengths, int T, int B, int C,
                int Lmax, const float* grad_out, float* grad, void* ws, size_t ws_bytes, void* stream);
/* pred [T][B][C] log-probs, label [L][B] int32 -> out int64 [T+2L+1][B] zero padded, lens[B] path lengths (bit exact) */
size_t hwg_dtw_workspace(int T, int B, int L);
int hwg_dtw_align(const float* pred, const int* label, int T, int B, int C, int L, long long* out, int* lens, void* ws, size_t ws_bytes,
                  void* stream);
/* index_spaced int64 [Tp][B], label int32 [L][B] -> gt [L][B][2] (caller zero-fills), minpos (caller sets to INT_MAX), mismatch counter */
int hwg_gt_counts(const long long* index_spaced, const int* label, int Tp, int B, int L, float* gt, int* minpos, int* mismatch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spectral norm (model/discriminator_ap.py:11-65), losses (model/loss.py:16-27, trainer :797-821),
 * PixelNorm (pure_gen.py:306-311) and small vector helpers.
 * ------------------------------------------------------------------------------------------ */
size_t hwg_spectral_workspace(int R, int K);
int hwg_spectral_update(const float* W, float* u, float* v, int R, int K, float eps, float* sigma, float* inv_sigma, void* ws,
                        size_t ws_bytes, void* stream);
/* the same iteration (u, v updated in place) that also writes the new vectors to u_copy / v_copy (either may be null): the snapshot a forward
 * pass keeps for its backward pass, without a separate copy (discriminator_ap.py:31-45 clones them) */
int hwg_spectral_update_to(const float* W, float* u, float* v, float* u_copy, float* v_copy, int R, int K, float eps, float* sigma,
                           float* inv_sigma, void* workspace, size_t workspace_bytes, void* stream);
/* the same iteration for n layers at once (four launches in total): `table` = n device records {const float* W; float* u; float* v;
 * long long copy_off; long long ws_off; int R; int K;} (48 bytes); layer i writes its snapshot to copies + copy_off (u [R], then v [K]), uses
 * ws + ws_off (K + R floats) as scratch and leaves (sigma, 1/sigma) in sig[2i], sig[2i+1] */
int hwg_spectral_update_multi(const void* table, int n, int max_R, int max_K, float eps, float* workspace, float* copies, float* sig, void* stream);
int hwg_scale_by_ptr(const float* x, const float* scale, float* out, long long n, void* stream);
int hwg_spectral_bwd(const float* dWsn, const float* Wbar, const float* u, const float* v, const float* sigma, float* dWbar, int R, int K,
                     int accumulate, void* ws, size_t ws_bytes, void* stream);
/* the same for up to 16 layers in two launches (block -> layer, each layer on its own partial schedule: bit-identical to n single calls).
 * table: n host records of 64 bytes {dWsn, Wbar, u, v, sigma, dWbar: 8-byte addresses; R, K, accumulate, pad: 4-byte ints}
 * (reference: the ten SpectralNorm layers of DiscriminatorAP walk backward one by one, model/discriminator_ap.py:11-65) */
size_t hwg_spectral_bwd_multi_workspace(int n);
int hwg_spectral_bwd_multi(const void* table, int n, void* ws, size_t ws_bytes, void* stream);
size_t hwg_loss_workspace(void);
/* out (+)= scale * mean(term); mode 0 |a-b|, 1 (a-b)^2, 2 a, 3 relu(1-a), 4 relu(1+a) */
int hwg_loss_fwd(const float* a, const float* b, long long n, int mode, float scale, float* out, int accumulate, void* ws, size_t ws_bytes,
                 void* stream);
int hwg_loss_bwd(const float* a, const float* b, long long n, int mode, float scale, const float* grad_out, float* da, float* db,
                 int accumulate, void* stream);
int hwg_pixelnorm_fwd(const float* x, float* y, int rows, int C, float eps, void* stream);
int hwg_pixelnorm_bwd(const float* dy, const float* x, float* dx, int rows, int C, float eps, void* stream);
/* scaled[i] = weights[i] * *x_i and *sum = their left-to-right sum (x_ptrs: HOST array of n <= 16 device addresses of scalars, weights: HOST
 * array; both read during the call) - the trainer's weighted loss accumulation (trainer/hw_with_style_trainer.py:280-298) as one launch,
 * rounded like the chain of hwg_axpby launches it replaces; _bwd: grads[i] = weights[i] * *grad_out */
int hwg_weighted_sum(const void* x_ptrs, const float* weights, int n, float* scaled, float* sum, void* stream);
int hwg_weighted_sum_bwd(const float* grad_out, const float* weights, int n, float* grads, void* stream);
/* out[b] = bank[ij[b]] * w[b] + bank[ij[B + b]] * w[B + b] over rows of D floats (bank [K][D], ij int32 [2][B], w [2][B], all on the device): the
 * interpolation of two stored styles per generated line (trainer/hw_with_style_trainer.py:974-988), products and sum rounded to fp32 */
int hwg_style_mix(const float* bank, const int* ij, const float* w, float* out, int K, int B, int D, void* stream);
int hwg_axpby(const float* x, float a, const float* y, float b, float* out, long long n, void* stream);
/* y = x*scale[c] + shift[c] on x[rows][C] (CountCNN output scaling, count_cnn.py:44); out = a*b elementwise */
int hwg_channel_affine(const float* x, const float* scale, const float* shift, float* y, long long rows, int C, void* stream);
int hwg_mul(const float* a, const float* b, float* out, long long n, void* stream);
int hwg_tanh_fwd(const float* x, float* y, long long n, void* stream);
int hwg_tanh_bwd(const float* dy, const float* y, float* dx, long long n, void* stream);
int hwg_argmax_rows(const float* x, int* out, long long rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Character-specific style extraction helpers (model/char_style.py:204-235,286).
 * ------------------------------------------------------------------------------------------ */
int hwg_gather_windows(const float* x, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int window, float* patches, void* stream);
/* gradient of hwg_gather_windows: dx[b][pos][c] = sum of the window entries that cover (b, pos), gathered in a fixed order (no atomics).
 * At most one window may be centred on a given (sample, column) - true for the arg-max map the windows come from. win_of: B*Wx ints of scratch. */
int hwg_scatter_windows(const float* dpatches, int B, int Wx, int C, const int* idx_b, const int* idx_pos, int n, int window, int* win_of,
                        float* dx, void* stream);
int hwg_segment_weighted_mean(const float* v, const float* wgt, const int* seg, int n, int C, int B, float* out, float* wsum, void* stream);
int hwg_segment_weighted_mean_bwd(const float* dout, const float* wgt, const int* seg, const float* wsum, int n, int C, float* dv, void* stream);
int hwg_gather_scores(const float* x, int B, int Wx, int C, const int* idx_b, const int* idx_pos, const int* idx_cls, int n, float* out, void* stream);

/* Bank of L linear layers that share one input (the generator's ten AdaIN style -> (gamma, beta) affines, pure_gen.py:52-69, in one
 * launch instead of ten): y_l = x W_l^T + b_l, x [B][I] (forward: any B, 16 rows per block; backward: B <= 16), W_l [O_l][I] and b_l [O_l] through device pointer tables.
 * Layer l's outputs are written as `halves` contiguous [B][O_l/halves] blocks starting at y + off[l]; first_wave[L+1] is the
 * running sum of O_l (neuron -> layer map), total_outputs = first_wave[L]. The backward ADDS dW_l / db_l into the tables'
 * buffers and writes dx [B][I] (optional); dyptr[l*halves + h] is the gradient of block h of layer l (0 = unused output). */
int hwg_linear_bank_fwd(const float* x, const void* wptr, const void* bptr, const int* O, const int* first_wave, const void* off, int L, int B,
                        int I, int halves, int total_outputs, float* y, void* stream);
size_t hwg_linear_bank_bwd_workspace(int total_outputs, int B, int I);
int hwg_linear_bank_bwd(const float* x, const void* dyptr, const void* wptr, const void* gwptr, const void* gbptr, const int* O,
                        const int* first_wave, int L, int B, int I, int halves, int total_outputs, float* dx, void* workspace,
                        size_t workspace_bytes, void* stream);

/* Chain of L square Linear(D,D)+LeakyReLU layers (the generator's style-embedding MLP, pure_gen.py:29-38) in one single-workgroup
 * launch per direction: h_{l+1} = lrelu(W_l h_l + b_l, slope). acts [L+1][B][D] receives every h_l (acts[0] = x, acts[L] = output) and is
 * what the backward consumes; the backward ADDS dW_l / db_l into the tables' buffers (null entry = frozen) and writes dx (optional).
 * D = 64 or 128, L <= 8; forward: any B (one workgroup per 8 / 16 rows), backward: B <= 16. */
int hwg_mlp_chain_fwd(const float* x, const void* wptr, const void* bptr, int L, int B, int D, float slope, float* acts, void* stream);
int hwg_mlp_chain_bwd(const float* dout, const float* acts, const void* wptr, const void* gwptr, const void* gbptr, int L, int B, int D,
                      float slope, float* dx, void* stream);
/* the same backward pass as two launches: the sequential chain of data gradients on one workgroup (every delta_l kept in the workspace), then
 * all layers' dW_l / db_l in parallel (L x D*D/1024 workgroups) - 69 -> ~25 us per call; bit-identical to hwg_mlp_chain_bwd */
size_t hwg_mlp_chain_bwd_workspace(int L, int B, int D);
int hwg_mlp_chain_bwd_split(const float* dout, const float* acts, const void* wptr, const void* gwptr, const void* gbptr, int L, int B, int D,
                            float slope, float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* Grouped "one expert per window" layers for the 79 character-style experts (model/char_style.py:84-124, 210-235).
 * x [n][R][Cin] -> y [n][R][Cout]; wptr/bptr are device tables (int64 addresses, one per expert) of weights in the Conv1d layout
 * [Cout][Cin][S] and biases; R <= 8; (S, pad) is (1, 0) or (3, 1). Windows are sorted by expert: seg_start[G+1] / seg_eid[G] describe
 * the runs; the rows of a run (windows x R positions) are cut into work tiles of at most 64 rows, tile_seg[t] = run and
 * tile_row0[t] = first row of tile t. Every run is one small GEMM on the matrix cores against its expert's weights.
 * wgrad and segment_accumulate ADD into the buffers addressed by the grad-pointer tables. */
int hwg_grouped_conv1d_fwd(const float* x, const int* seg_start, const int* seg_eid, const int* tile_seg, const int* tile_row0, int ntiles,
                           const void* wptr, const void* bptr, float* y, int R, int Cin, int Cout, int S, int pad, void* stream);
int hwg_grouped_conv1d_dgrad(const float* dy, const int* seg_start, const int* seg_eid, const int* tile_seg, const int* tile_row0, int ntiles,
                             const void* wptr, float* dx, int R, int Cin, int Cout, int S, int pad, void* stream);
size_t hwg_grouped_conv1d_wgrad_workspace(int ntiles, int Cin, int Cout, int S);
/* the weight gradient uses its own tiling of at most tile_rows rows per tile; run_tile0[G+1] = first tile of every run. The tiles'
 * partial images go to the workspace and are added to the experts' buffers in tile order (deterministic). */
int hwg_grouped_conv1d_wgrad(const float* dy, const float* x, const int* seg_start, const int* seg_eid, int G, const int* tile_seg,
                             const int* tile_row0, const int* run_tile0, int ntiles, int tile_rows, const void* gwptr, const void* gbptr, int R,
                             int Cin, int Cout, int S, int pad, void* workspace, size_t workspace_bytes, void* stream);
int hwg_gather_rows_ptr(const void* ptrs, const int* eid, float* out, int n, int C, void* stream);
int hwg_segment_accumulate_ptr(const float* rows, const int* seg_start, const int* seg_eid, int G, const void* gptrs, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-tensor optimizer-side ops (trainer/hw_with_style_trainer.py:300-391) and RNG.
 * Tensor lists are device tables: ptrs (int64 addresses, 0 = absent), numel (int64), and a chunk table
 * (chunk_tensor int32, chunk_off int64) with `chunk` elements per entry.
 * ------------------------------------------------------------------------------------------ */
/* out_sums[t] = sum |x| over tensor t (all nt tensors; absent ones give 0). Two stages, no atomics: per-chunk partials (chunk_partials,
 * nchunks doubles of scratch), then the chunks of a tensor - consecutive entries of the chunk table - are added in table order. */
int hwg_mt_abs_sum(const void* ptrs, const void* numel, const void* chunk_tensor, const void* chunk_off, int nchunks, int chunk,
                   int nt, double* chunk_partials, double* out_sums, void* stream);
/* the same for nsets tensor lists at once (ptrs [nsets][nt], chunk_partials [nsets][nchunks], out_sums [nsets][nt]): the current gradients and
 * every stashed set of a balanced lesson (trainer :340-359) in two launches instead of two per set; per set the arithmetic of hwg_mt_abs_sum */
int hwg_mt_abs_sum_sets(const void* ptrs, int nsets, const void* numel, const void* chunk_tensor, const void* chunk_off, int nchunks,
                        int chunk, int nt, double* chunk_partials, double* out_sums, void* stream);
int hwg_mt_balance_coef(const double* sumD, const double* sumR, const void* numel, const void* ptr_grad, const void* ptr_R,
                        const float* xs, int nsets, int nt, float* coef, void* stream);
int hwg_mt_axpy(const void* ptrs_dst, const void* ptrs_src, const float* coef, const void* numel, const void* chunk_tensor,
                const void* chunk_off, int nchunks, int chunk, void* stream);
/* dst_t += coef[0][t] * src_0t, then += coef[1][t] * src_1t, ...: the balanced adds of all stashed sets (trainer :360-377) in ONE pass over the
 * gradients (ptrs_src, coef: [nsets][nt], nsets <= 8) - per element the same chain of fused multiply-adds in the same order as nsets calls of
 * hwg_mt_axpy, bit for bit */
int hwg_mt_axpy_sets(const void* ptrs_dst, const void* ptrs_src, const float* coef, int nsets, int nt, const void* numel,
                     const void* chunk_tensor, const void* chunk_off, int nchunks, int chunk, void* stream);
/* op 0: a=0, 1: clamp(a,-c,c), 2: flag |= any non-finite, 3: b=a, 4: b=a then a=0 */
int hwg_mt_unary(const void* ptrs_a, const void* ptrs_b, int op, float c, int* flag, const void* numel, const void* chunk_tensor,
                 const void* chunk_off, int nchunks, int chunk, void* stream);
int hwg_mt_adam(const void* ptrs_p, const void* ptrs_g, const void* ptrs_m, const void* ptrs_v, const float* step_size,
                const float* bc2_sqrt, float beta1, float beta2, float eps, float clip, const void* numel, const void* chunk_tensor,
                const void* chunk_off, int nchunks, int chunk, void* stream);
/* clip_grad_value_(clip) + the NaN asserts + Adam of one stepping lesson in ONE launch (trainer/hw_with_style_trainer.py:379-391): a tensor with a
 * gradient entry and a NULL parameter entry is clipped only (touched, but not stepped in this lesson); a tensor with both is clipped and
 * stepped, *flag |= 1 when a freshly written parameter is not finite (flag is sticky: never cleared here). Gradients are stored back only
 * where clipping changed them. Same arithmetic per element as hwg_mt_unary(op 1) followed by hwg_mt_adam. */
int hwg_mt_clip_adam(const void* ptrs_p, const void* ptrs_g, const void* ptrs_m, const void* ptrs_v, const float* step_size,
                     const float* bc2_sqrt, float beta1, float beta2, float eps, float clip, int* flag, const void* numel,
                     const void* chunk_tensor, const void* chunk_off, int nchunks, int chunk, void* stream);
int hwg_randn(float* out, long long n, unsigned long long seed, unsigned long long offset, void* stream);
int hwg_dropmask(float* out, long long n, float p, unsigned long long seed, unsigned long long offset, void* stream);
/* the Dropout2d masks of one network pass in ONE launch (model/discriminator_ap.py:84-131 has up to six, model/autoencoder.py:341-410 four):
 * nseg (<= 16) segments of seg_elems[j] floats (multiples of 4; HOST arrays, read during the call) laid out back to back in `out`, segment
 * j with drop probability seg_p[j]. Same values as nseg consecutive hwg_dropmask calls at offsets offset + sum_{i<j} seg_elems[i] / 4. */
int hwg_dropmask_multi(float* out, int nseg, const long long* seg_elems, const float* seg_p, unsigned long long seed,
                       unsigned long long offset, void* stream);

/* `insert_spaces` (model/hw_with_style.py:302-328) with the device generator: plan = per character of every line the number of blank columns
 * before it and of repeats, drawn as round(N(count, count_std)) / round(N(duplicates, dup_std)) (half to even, negative -> 0) from Philox block
 * offset + (line * L + character); lens_max [B+1] = expanded length per line, then max(ceil(max counts), 3) (the reference's tail padding);
 * fill writes the character runs into the zero-initialised index map idx [T][B] (0 = blank). counts [L][B][2], label [L][B]. */
int hwg_insert_spaces_plan(const float* counts, const int* label_lengths, int L, int B, float count_std, float dup_std, int count_duplicates,
                           unsigned long long seed, unsigned long long offset, int* reps, int* starts, int* lens_max, void* stream);
int hwg_insert_spaces_fill(const int* label, const int* label_lengths, const int* reps, const int* starts, int L, int B, int T, int* idx,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HWG_H_ */
