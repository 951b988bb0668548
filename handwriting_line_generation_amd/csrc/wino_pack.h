// Winograd F(2x2,3x3) filter transform, shared by conv_wino.hip (single-weight pack) and conv_mfma.hip (multi-weight re-pack).
#pragma once
#include <hip/hip_runtime.h>

// U = G g G^T with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], written as [C/16][16 pos][Apad][16]; element (a, b, r, s) of the source
// at src[a*sa + b*sb + r*sr + s*ss]; flip mirrors the taps (data gradient / stride-1 transposed convolution).
static __device__ __forceinline__ void wino_filter_transform(const float g[3][3], float U[4][4]) {
  float t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
    t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
    t[3][s] = g[2][s];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    U[r][0] = t[r][0];
    U[r][1] = 0.5f * (t[r][0] + t[r][1] + t[r][2]);
    U[r][2] = 0.5f * (t[r][0] - t[r][1] + t[r][2]);
    U[r][3] = t[r][2];
  }
}

static __device__ __forceinline__ void wino_pack_one(const float* src, float* dst, long long i, int A, int Apad, int B, int Bpad, long long sa,
                                              long long sb, long long sr, long long ss, int flip, float scale = 1.f) {
  const int b = (int)(i % Bpad);
  const int aa = (int)(i / Bpad);
  float g[3][3];
  const bool ok = aa < A && b < B;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int rr = flip ? 2 - r : r, s2 = flip ? 2 - s : s;
      g[r][s] = ok ? __fmul_rn(src[aa * sa + b * sb + rr * sr + s2 * ss], scale) : 0.f;   // (rounded product: the transform must see w * scale, not an fma)
    }
  float U[4][4];
  wino_filter_transform(g, U);
  const long long base = ((long long)(b >> 4) * 16 * Apad + aa) * 16 + (b & 15);
#pragma unroll
  for (int p = 0; p < 16; ++p) dst[base + (long long)p * Apad * 16] = U[p >> 2][p & 3];
}

