// Counter-based RNG of the samplers: jax.random's threefry2x32 key derivation and normal() (jax 0.5.3, the reference's
// pin, setup.py:18-19), so that the per-step draws of ddpm.py:204-206 / continuous_ddpm.py:161,174 happen inside the
// kernel instead of arriving as a (T, B, A) tensor.  Integer part bit-exact; the float part follows XLA's erf_inv
// (single-precision polynomial of M. Giles on w = -log1p(-x*x)) and agrees with a CPU evaluation to a few ulp (log1p,
// fma contraction).  Oracle: oracle/threefry_oracle.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace frl {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, 32 - r); }

// Threefry-2x32, 20 rounds (Salmon et al., SC'11)
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0, uint32_t& o1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
#define FRL_TF_ROUND(r) { x0 += x1; x1 = rotl32(x1, r) ^ x0; }
#define FRL_TF_A FRL_TF_ROUND(13) FRL_TF_ROUND(15) FRL_TF_ROUND(26) FRL_TF_ROUND(6)
#define FRL_TF_B FRL_TF_ROUND(17) FRL_TF_ROUND(29) FRL_TF_ROUND(16) FRL_TF_ROUND(24)
  FRL_TF_A x0 += ks[1]; x1 += ks[2] + 1u;
  FRL_TF_B x0 += ks[2]; x1 += ks[0] + 2u;
  FRL_TF_A x0 += ks[0]; x1 += ks[1] + 3u;
  FRL_TF_B x0 += ks[1]; x1 += ks[2] + 4u;
  FRL_TF_A x0 += ks[2]; x1 += ks[0] + 5u;
#undef FRL_TF_A
#undef FRL_TF_B
#undef FRL_TF_ROUND
  o0 = x0; o1 = x1;
}

// 32 random bits of element `idx` of an n-element draw under `key` (random_bits(key, 32, shape), flat row-major idx)
__device__ __forceinline__ uint32_t rng_bits(uint32_t k0, uint32_t k1, int impl, unsigned long long n, unsigned long long idx) {
  uint32_t o0, o1;
  if (impl == FRL_RNG_THREEFRY_PARTITIONABLE) {
    threefry2x32(k0, k1, (uint32_t)(idx >> 32), (uint32_t)idx, o0, o1);
    return o0 ^ o1;
  }
  // legacy layout: counts 0..n-1 (zero-padded to even) cut into halves (x0, x1); words = concat(out0, out1)
  const unsigned long long half = (n + 1) >> 1;
  if (idx < half) {
    const unsigned long long c1 = idx + half;
    threefry2x32(k0, k1, (uint32_t)idx, c1 < n ? (uint32_t)c1 : 0u, o0, o1);
    return o0;
  }
  threefry2x32(k0, k1, (uint32_t)(idx - half), (uint32_t)idx, o0, o1);
  return o1;
}

// jax.random.normal's map: u = max(lo, f * 2 + lo), f = mantissa-trick uniform in [0, 1), lo = nextafter(-1, 0); sqrt(2) erf_inv(u)
__device__ __forceinline__ float rng_bits_to_normal(uint32_t bits) {
  const float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
  const float lo = -0.99999994f;
  const float x = fmaxf(lo, f * 2.0f + lo);
  const float w = -log1pf(-(x * x));
  float p;
  if (w < 5.0f) {
    const float v = w - 2.5f;
    p = 2.81022636e-08f;
    p = 3.43273939e-07f + p * v;
    p = -3.5233877e-06f + p * v;
    p = -4.39150654e-06f + p * v;
    p = 0.00021858087f + p * v;
    p = -0.00125372503f + p * v;
    p = -0.00417768164f + p * v;
    p = 0.246640727f + p * v;
    p = 1.50140941f + p * v;
  } else {
    const float v = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = 0.000100950558f + p * v;
    p = 0.00134934322f + p * v;
    p = -0.00367342844f + p * v;
    p = 0.00573950773f + p * v;
    p = -0.0076224613f + p * v;
    p = 0.00943887047f + p * v;
    p = 1.00167406f + p * v;
    p = 2.83297682f + p * v;
  }
  return 1.41421356f * (p * x);
}

// The N(0, 1) draw of loop iteration j for element idx of the (B, A) state; keys = the iteration keys (2 words each)
__device__ __forceinline__ float rng_step_normal(const uint32_t* __restrict__ keys, int impl, int j, unsigned long long n,
                                                 unsigned long long idx) {
  return rng_bits_to_normal(rng_bits(keys[2 * j], keys[2 * j + 1], impl, n, idx));
}

}  // namespace frl
