// Philox4x32-10 counter-based generator and the Box-Muller pair step shared by optim_rng.hip (hwg_randn) and norm_act.hip (the generator
// epilogue that draws its noise in the kernel instead of reading a noise tensor): both must produce the SAME normals for the same
// (seed, counter), so the arithmetic lives here once.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
__device__ __forceinline__ void hwg_philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  c[0] = hi1 ^ c[1] ^ k[0]; c[1] = lo1; c[2] = hi0 ^ c[3] ^ k[1]; c[3] = lo0;
  k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}
__device__ __forceinline__ void hwg_philox4(uint64_t seed, uint64_t ctr, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
  for (int r = 0; r < 10; ++r) hwg_philox_round(c, k);
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
__device__ __forceinline__ float hwg_u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.f / 16777216.f); }  // (0,1)
// the four standard normals of counter `ctr` (elements 4*ctr .. 4*ctr+3 of the stream)
__device__ __forceinline__ float4 hwg_randn4(uint64_t seed, uint64_t ctr) {
  uint32_t r[4];
  hwg_philox4(seed, ctr, r);
  const float u0 = hwg_u01(r[0]), u1 = hwg_u01(r[1]), u2 = hwg_u01(r[2]), u3 = hwg_u01(r[3]);
  const float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
  float s0, c0, s1, c1;
  sincosf(6.2831853071795864f * u1, &s0, &c0);
  sincosf(6.2831853071795864f * u3, &s1, &c1);
  return make_float4(ra * c0, ra * s0, rb * c1, rb * s1);
}
#endif
