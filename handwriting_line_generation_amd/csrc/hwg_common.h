// Shared device/host helpers for the hwg HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/hwg.h"

#define HWG_WAVE 64

// thread-local last error text, exposed through hwg_last_error()
void hwg_set_error(const char* fmt, ...);

#define HWG_REQUIRE(cond, ...)                                  \
  do {                                                          \
    if (!(cond)) {                                              \
      hwg_set_error(__VA_ARGS__);                               \
      return HWG_ERR_ARG;                                       \
    }                                                           \
  } while (0)

#define HWG_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      hwg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return HWG_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

// launch profiler (hwg_core.hip): hwg_prof_open returns a record index (-1 when profiling is off) and makes it the calling thread's current
// record until hwg_prof_close. Every kernel of this library is launched through hipExtLaunchKernelGGL (the hipLaunchKernelGGL spelling is
// redirected below); with a current record the launch carries the record's events as the dispatch's own start / stop events, i.e. the
// record holds the KERNEL's begin and end timestamps - no extra packets between kernels, no launch gaps inside the measured interval
// (event pairs recorded around a launch measured 5.4 us more per launch than rocprofv3's kernel trace: +20..50 % on the 20 us kernels of
// the generator). A second launch inside the same bracket (a kernel and its reduce pass) moves the stop event only.
int hwg_prof_open(int kind, double work, hipStream_t st);
void hwg_prof_close(int rec, hipStream_t st);
struct HwgProfEv { hipEvent_t e0, e1; };
HwgProfEv hwg_prof_launch_events();
#include <hip/hip_ext.h>
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shm, st, ...)                                             \
  do {                                                                                                    \
    const HwgProfEv pe__ = hwg_prof_launch_events();                                                      \
    hipExtLaunchKernelGGL(kernel, grid, block, shm, st, pe__.e0, pe__.e1, 0, __VA_ARGS__);                \
  } while (0)
enum { HWG_PROF_CONV = 0, HWG_PROF_WGRAD = 1, HWG_PROF_CONV_REDUCE = 2, HWG_PROF_WGRAD_REDUCE = 3, HWG_PROF_CONV_DIRECT = 4, HWG_PROF_WGRAD_DIRECT = 5, HWG_PROF_CONV_WINO = 6, HWG_PROF_WGRAD_WINO = 7 };

static inline int hwg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- schedules are planned once per geometry ------------------------------------------------------------------------------
// The cost models (and the environment knobs that override them: HWG_WINO, HWG_WINO_FORCE, HWG_CONV_FORCE, HWG_WGRAD_FORCE, ...) are
// evaluated when a geometry is first seen; every later launch of that geometry is a lookup in a per-thread table. The knobs are
// read at plan time only: a process that changes them afterwards (tests, sweeps) calls hwg_tuning_reload(), which starts a new epoch
// and thereby drops every cached plan.
unsigned hwg_tuning_epoch();
// snapshot of the tuning knobs, taken on first use and after every hwg_tuning_reload() (strings: "" when unset)
struct HwgTune {
  unsigned epoch;
  int wino;              // HWG_WINO: 0 never, 2 always, 1 (default) by cost model
  int wino_wgrad;        // HWG_WINO_WGRAD: 0 never, 2 always, 1 (default) by rule
  int wgrad_narrow;      // HWG_WGRAD_NARROW: 0 never, 2 also 2 / 4 channel blocks, 1 default
  int w64_nodma;         // HWG_W64_NODMA: register-staged filter stream in the 64x64 Winograd kernel (A/B timing)
  int wino_order;        // HWG_WINO_ORDER: 1 (default) XCD-contiguous work order
  int wino_s2;           // HWG_WINO_S2: F(3x3,2x2) for 4x4 stride-2 pad-0 layers: 0 never, 2 always, 1 (default) by cost model
  int conv_merge;        // HWG_CONV_MERGE: transposed convolutions with R % sh == 0, S % sw == 0 with merged parity classes: 0 never, 1 (default) K < 32, 2 always
  int wino_wgrad_split;  // HWG_WINO_WGRAD_SPLIT: forced pixel-range count of the Winograd weight gradient (0 = model)
  int wwg_debug;         // HWG_WWG_DEBUG
  int conv_pf;           // HWG_CONV_PF: register prefetch depth of the implicit-GEMM conv kernel (1 or 2)
  int conv_wk;           // HWG_CONV_WK: 2 (default) = K-split wavefront pairs on the 64 x 64 direct conv tile (8 wavefronts), 1 = four wavefronts
  int conv_dbg;          // HWG_CONV_DBG: timing ablations of the 128 x 128 direct conv kernel (garbage results), 0 default
  int c1_mfma;           // HWG_C1_MFMA: 0 = single-input-channel forward convs on the VALU kernel (A/B timing), 1 default = taps-as-K on the matrix cores
  int to1_lanes;         // HWG_TO1_LANES: 0 = one wave per output pixel whenever taps x channel groups > 16 (A/B timing), 1 default = lanes cover one tap's channel groups
  int conv_lds;          // HWG_CONV_LDS: 0 keeps strided layers on the per-tap gather kernel (A/B timing), 1 default
  int wgrad_c1;          // HWG_WGRAD_C1: 0 default = single-channel first-layer weight gradients on the taps-as-N MFMA kernel, 1 = VALU kernel (wgrad_c1_kernel: measured slower, kept for A/B runs)
  int wgrad_c1_rows;     // HWG_WGRAD_C1_ROWS: 0 = single-channel first-layer weight gradients on the taps-as-N gather kernel (A/B timing), 1 default = input rows staged in LDS
  int c1_rows;           // HWG_C1_ROWS: 0 = single-input-channel forward convs on the gather kernels (A/B timing), 1 default = input rows staged in LDS, filter in registers
  int split_inkernel;    // HWG_SPLIT_INKERNEL: 0 default = split-K / channel-split partial images summed by a reduce launch; 1 = by the wavefront / workgroup that delivers a tile's LAST partial (bit-identical; measured SLOWER in the step - what crosses XCDs has to bypass the L2s -, kept for A/B runs: profiles/r06_inkernel_sums.txt)
  int norm_fused;        // HWG_NORM_FUSED: 0 default = moments pass and apply pass of the per-sample normalisations as two launches; 1 = one launch per direction with a barrier over the sample's workgroups (bit-identical; measured SLOWER in the step, kept for A/B runs: profiles/r06_inkernel_sums.txt)
  int wgrad_reduce_rows; // HWG_WGRAD_REDUCE_ROWS: 0 = tap-at-a-time partial-image reduce (A/B timing), 1 default = row-contiguous stores
  char wino_force[32];   // HWG_WINO_FORCE  "cfg[,nsplit]"
  char wino_bal[32];     // HWG_WINO_BAL    balanced schedule of the 64 x 64 Winograd kernel: -1 never, unset / 0 by model, "G[,lead tiles]" forced
  char conv_force[48];   // HWG_CONV_FORCE  "bm,bn,bk[,nsplit]"
  char wgrad_force[32];  // HWG_WGRAD_FORCE "cfg,target_blocks"
  char wino_cost6[48];   // HWG_WINO_COST6  "fixed_us,step_us"
};
const HwgTune& hwg_tune();
// Arrival counters of the split-K kernels (one int per (output tile, wavefront sub-tile)): a per-stream device buffer of HWG_SPLIT_COUNTERS ints,
// zeroed when it is created; the wavefront that finds its counter at nsplit - 1 sets it back to 0, so every launch finds zeros (launches of one
// stream run in order; two streams never share a buffer). nullptr (with the error set) when the allocation fails.
constexpr int HWG_SPLIT_COUNTERS = 1 << 18;
int* hwg_split_counters(hipStream_t st);
#if defined(__HIPCC__)
// Arrival protocol of the split kernels (the wavefront / workgroup that delivers a tile's LAST partial image sums all of them itself, in split
// order - same bits as a separate reduce launch, no second launch). What crosses workgroups - the partial images and the counter - is written
// and read with AGENT-SCOPE relaxed atomics (hwg_store_agent / hwg_load_agent: `sc1` stores write through to the memory side, `sc1` loads do not
// hit this XCD's possibly stale L2 / L1 lines), ordered by `s_waitcnt vmcnt(0)` between the stores and the counter increment (a store is
// acknowledged once it is visible at its scope). NOT with __threadfence(): the XCDs' L2s are not coherent with each other, so an agent-scope
// fence is `buffer_wbl2` + `buffer_inv sc1` - a write-back and an invalidation of the WHOLE L2, per wavefront; measured on the training step:
// 83 -> 62 steps/s with fences in the split convolutions alone. With the agent-scope accesses it is 90 -> 77: the store acknowledgements from
// the memory side, the counter round trip and the dependent `sc1` loads sit at the end of every 25 us kernel where a 5 us reduce launch runs
// at full width. OFF by default (HWG_SPLIT_INKERNEL), kept as a measured alternative: profiles/r06_inkernel_sums.txt.
__device__ __forceinline__ void hwg_store_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float hwg_load_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hwg_store_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double hwg_load_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hwg_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Both return the same answer to all calling lanes; the counter is set back to zero by whoever read `expected - 1`.
// Wavefront form: the calling wavefront owns its output elements alone (same elements in every split); all 64 lanes call it.
__device__ __forceinline__ bool hwg_split_arrive_wave(int* slot, int expected) {
  hwg_stores_done();
  int arrived = 0;
  if ((threadIdx.x & 63) == 0) arrived = atomicAdd(slot, 1);
  arrived = __builtin_amdgcn_readfirstlane(arrived);
  if (arrived != expected - 1) return false;
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
  return true;
}
// Workgroup form: all threads of the workgroup call it (two barriers inside); `flag` is one int of LDS.
__device__ __forceinline__ bool hwg_split_arrive_block(int* slot, int expected, int* flag) {
  hwg_stores_done();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int last = atomicAdd(slot, 1) == expected - 1;
    if (last) __hip_atomic_store(slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}
#endif
// what the calling thread's last convolution-family launch ran: engine (HWG_PROF_* kind), schedule id, split factor
void hwg_note_plan(int engine, int cfg, int nsplit);

template <class Plan, int SLOTS = 256>
struct HwgPlanCache {
  struct Entry { hwg_conv_desc d; Plan p; unsigned epoch; };
  Entry e[SLOTS];
  HwgPlanCache() { for (int i = 0; i < SLOTS; ++i) e[i].epoch = 0; }
  static unsigned slot(const hwg_conv_desc* d) {
    const unsigned* w = (const unsigned*)d;
    unsigned h = 2166136261u;
    for (unsigned i = 0; i < sizeof(hwg_conv_desc) / sizeof(unsigned); ++i) h = (h ^ w[i]) * 16777619u;
    return (h ^ (h >> 15)) % SLOTS;
  }
  template <class Make>
  const Plan& get(const hwg_conv_desc* d, Make make) {
    Entry& x = e[slot(d)];
    const unsigned ep = hwg_tuning_epoch();
    if (x.epoch != ep || memcmp(&x.d, d, sizeof(hwg_conv_desc)) != 0) {
      x.p = make(d);
      x.d = *d;
      x.epoch = ep;
    }
    return x.p;
  }
};

// cap grid for grid-stride streaming kernels: 256 CUs x 8 blocks
static inline int hwg_stream_grid(long long work_items, int block) {
  long long g = (work_items + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); result valid in all threads
__device__ __forceinline__ double block_sum_d(double v, double* smem /* >= 16 doubles */) {
  v = wave_sum_d(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double r = 0.0;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}
__device__ __forceinline__ float block_sum_f(float v, float* smem /* >= 16 floats */) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  // act: 0 none, 1 relu, 2 leaky relu(slope), 3 tanh
  if (act == 1) return v > 0.f ? v : 0.f;
  if (act == 2) return v > 0.f ? v : v * slope;
  if (act == 3) return tanhf(v);
  return v;
}
// derivative given the OUTPUT value y (valid for relu / lrelu with slope>0 / tanh)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
  if (act == 1) return y > 0.f ? 1.f : 0.f;
  if (act == 2) return y > 0.f ? 1.f : slope;
  if (act == 3) return 1.f - y * y;
  return 1.f;
}
#endif
