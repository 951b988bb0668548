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


// F(3x3,2x2) image of a 4x4 stride-2 convolution's weight (conv_wino.hip, hwg_wino_s2_*): pair i = (A index aa, B index b) of the product -
// forward: A = output channels, B = the 4 Cc space-to-depth channels (a, b, c); data gradient: A = the 4 Cc block channels, B = Kc, taps
// mirrored. g[u][v] = w[kc, cc, 2u+a, 2v+b] at src[kc*sk + cc*sc + r*4 + s]; U = G g G^T with G = [[1,0],[.5,.5],[.5,-.5],[0,1]], the entries
// with exactly one index equal to 3 negated (the kernel's input transform is F(2x2,3x3)'s, whose fourth row is the negative of F(3x3,2x2)'s).
static __device__ __forceinline__ void wino_s2_pack_one(const float* src, float* dst, long long i, int Kc, int Cc, long long sk, long long sc, int dgrad,
                                                 float scale = 1.f) {
  const int A = dgrad ? 4 * Cc : Kc, B = dgrad ? Kc : 4 * Cc;
  const int Apad = (A + 15) / 16 * 16, Bpad = (B + 15) / 16 * 16;
  const int b = (int)(i % Bpad), aa = (int)(i / Bpad);
  const bool ok = aa < A && b < B;
  const int comp = dgrad ? aa : b, ab = ok ? comp / Cc : 0, cc = ok ? comp % Cc : 0, kc = ok ? (dgrad ? b : aa) : 0;
  const int pa = ab >> 1, pb = ab & 1;
  float g[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int uu = dgrad ? 1 - u : u, vv = dgrad ? 1 - v : v;
      g[u][v] = ok ? __fmul_rn(src[kc * sk + cc * sc + (2 * uu + pa) * 4 + (2 * vv + pb)], scale) : 0.f;
    }
  float t[4][2], U[4][4];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    t[0][v] = g[0][v]; t[1][v] = 0.5f * (g[0][v] + g[1][v]); t[2][v] = 0.5f * (g[0][v] - g[1][v]); t[3][v] = g[1][v];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    U[r][0] = t[r][0]; U[r][1] = 0.5f * (t[r][0] + t[r][1]); U[r][2] = 0.5f * (t[r][0] - t[r][1]); U[r][3] = t[r][1];
  }
  const long long base = ((long long)(b >> 4) * 16 * Apad + aa) * 16 + (b & 15);
#pragma unroll
  for (int ps = 0; ps < 16; ++ps) {
    const bool neg = ((ps >> 2) == 3) != ((ps & 3) == 3);
    dst[base + (long long)ps * Apad * 16] = neg ? -U[ps >> 2][ps & 3] : U[ps >> 2][ps & 3];
  }
}
