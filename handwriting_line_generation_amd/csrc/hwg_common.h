// Shared device/host helpers for the hwg HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

// launch profiler (hwg_core.hip): returns a record index or -1 when profiling is off
int hwg_prof_open(int kind, double work, hipStream_t st);
void hwg_prof_close(int rec, hipStream_t st);
enum { HWG_PROF_CONV = 0, HWG_PROF_WGRAD = 1, HWG_PROF_CONV_REDUCE = 2, HWG_PROF_WGRAD_REDUCE = 3, HWG_PROF_CONV_DIRECT = 4, HWG_PROF_WGRAD_DIRECT = 5, HWG_PROF_CONV_WINO = 6, HWG_PROF_WGRAD_WINO = 7 };

static inline int hwg_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

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
