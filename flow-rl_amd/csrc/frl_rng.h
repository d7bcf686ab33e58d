// Counter-based RNG of the samplers: jax.random's threefry2x32 key derivation and normal() (jax 0.5.3, the reference's
// pin, setup.py:18-19), so that the per-step draws of ddpm.py:204-206 / continuous_ddpm.py:161,174 happen inside the
// kernel instead of arriving as a (T, B, A) tensor.  Integer part bit-exact; the float part follows XLA's erf_inv
// (single-precision polynomial of M. Giles on w = -log1p(-x*x)) and agrees with a CPU evaluation to a few ulp (the log1p
// form below, fma contraction).  Oracle: oracle/threefry_oracle.py.
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

// Branch-free form of rng_bits (selects only): the samplers run it between the MFMAs of a layer, where a branch would end
// the scheduling region.
__device__ __forceinline__ uint32_t rng_bits_nb(uint32_t k0, uint32_t k1, int impl, unsigned long long n, unsigned long long idx) {
  const bool part = impl == FRL_RNG_THREEFRY_PARTITIONABLE;
  const unsigned long long half = (n + 1) >> 1;
  const bool first = idx < half;
  const unsigned long long c1l = idx + half;
  const uint32_t c0 = part ? (uint32_t)(idx >> 32) : (first ? (uint32_t)idx : (uint32_t)(idx - half));
  const uint32_t c1 = part ? (uint32_t)idx : (first ? (c1l < n ? (uint32_t)c1l : 0u) : (uint32_t)idx);
  uint32_t o0, o1;
  threefry2x32(k0, k1, c0, c1, o0, o1);
  return part ? (o0 ^ o1) : (first ? o0 : o1);
}

__device__ __forceinline__ float rng_normal_x(uint32_t bits);
__device__ __forceinline__ float rng_normal_w(float x);

// One draw as TWELVE micro-steps of <= ~14 VALU instructions each (selects only, no branch): the samplers issue one
// micro-step behind each MFMA of a k-step, so that the matrix pipe covers them (an in-order wave hides VALU work only when
// it sits between its MFMAs in program order).  m: 0 counters, 1..5 four threefry rounds + key injection each, 6 output
// word -> x, 7 w, 8 central polynomial, 9 sqrt, 10 tail polynomial, 11 select -> z.
struct RngDraw {
  uint32_t x0, x1;      // threefry state
  uint32_t m0, m1;      // output word = (o0 & m0) ^ (o1 & m1): partitionable both all-ones; legacy one of them (no select, no branch)
  float x, w, p, t;
};

template <int G>
__device__ __forceinline__ void threefry_group(uint32_t& x0, uint32_t& x1, uint32_t ks0, uint32_t ks1, uint32_t ks2) {
#define FRL_TF_ROUND(r) { x0 += x1; x1 = rotl32(x1, r) ^ x0; }
  if constexpr ((G & 1) == 1) { FRL_TF_ROUND(13) FRL_TF_ROUND(15) FRL_TF_ROUND(26) FRL_TF_ROUND(6) }
  else { FRL_TF_ROUND(17) FRL_TF_ROUND(29) FRL_TF_ROUND(16) FRL_TF_ROUND(24) }
#undef FRL_TF_ROUND
  const uint32_t ks[3] = {ks0, ks1, ks2};
  x0 += ks[G % 3];
  x1 += ks[(G + 1) % 3] + (uint32_t)G;
}

template <int M>
__device__ __forceinline__ void rng_micro(RngDraw& d, uint32_t k0, uint32_t k1, int impl, unsigned long long n, unsigned long long idx) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  if constexpr (M == 0) {
    const bool part = impl == FRL_RNG_THREEFRY_PARTITIONABLE;
    const unsigned long long half = (n + 1) >> 1;
    const bool first = idx < half;
    const unsigned long long c1l = idx + half;
    const uint32_t c0 = part ? (uint32_t)(idx >> 32) : (first ? (uint32_t)idx : (uint32_t)(idx - half));
    const uint32_t c1 = part ? (uint32_t)idx : (first ? (c1l < n ? (uint32_t)c1l : 0u) : (uint32_t)idx);
    d.m0 = (part || first) ? 0xFFFFFFFFu : 0u;
    d.m1 = (part || !first) ? 0xFFFFFFFFu : 0u;
    d.x0 = c0 + k0;
    d.x1 = c1 + k1;
  } else if constexpr (M <= 5) {
    threefry_group<M>(d.x0, d.x1, k0, k1, k2);
  } else if constexpr (M == 6) {
    const uint32_t bits = (d.x0 & d.m0) ^ (d.x1 & d.m1);
    d.x = rng_normal_x(bits);
  } else if constexpr (M == 7) {
    d.w = rng_normal_w(d.x);
  } else if constexpr (M == 8) {
    const float v = d.w - 2.5f;
    float p = 2.81022636e-08f;
    p = 3.43273939e-07f + p * v;
    p = -3.5233877e-06f + p * v;
    p = -4.39150654e-06f + p * v;
    p = 0.00021858087f + p * v;
    p = -0.00125372503f + p * v;
    p = -0.00417768164f + p * v;
    p = 0.246640727f + p * v;
    d.p = 1.50140941f + p * v;
  } else if constexpr (M == 9) {
    d.t = __builtin_sqrtf(fmaxf(d.w, 0.0f)) - 3.0f;
  } else if constexpr (M == 10) {
    const float t = d.t;
    float q = -0.000200214257f;
    q = 0.000100950558f + q * t;
    q = 0.00134934322f + q * t;
    q = -0.00367342844f + q * t;
    q = 0.00573950773f + q * t;
    q = -0.0076224613f + q * t;
    q = 0.00943887047f + q * t;
    q = 1.00167406f + q * t;
    d.t = 2.83297682f + q * t;
  } else {
    d.p = 1.41421356f * ((d.w < 5.0f ? d.p : d.t) * d.x);      // the draw
  }
}

// jax.random.normal's map: u = max(lo, f * 2 + lo), f = mantissa-trick uniform in [0, 1), lo = nextafter(-1, 0);
// sqrt(2) erf_inv(u) with XLA's erf_inv: w = -log1p(-x x), two polynomials split at w = 5.  In three branch-free pieces
// (the samplers interleave them with MFMAs):
//   rng_normal_x: bits -> x
//   rng_normal_w: x -> w.  log1p(y), y = -x^2 in (-1, 0], as log(u) - c / u with u = fl(1 + y) and c = (u - 1) - y the
//                 exact rounding error of that sum (first-order correction; |c / u| <= 2^-24 / u, its square is far below
//                 one ulp of the result) — a tenth of the instructions of ocml's double-float log1pf at <= 4 ulp of w, which
//                 moves the normal by < 0.6 x the 4e-7 bar of tests/test_gpu_rng.py (scripts/ubench/rng_fastlog.hip, all 2^23 inputs)
//   rng_normal_z: (x, w) -> z, both polynomials evaluated, one selected
__device__ __forceinline__ float rng_normal_x(uint32_t bits) {
  const float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
  const float lo = -0.99999994f;
  return fmaxf(lo, f * 2.0f + lo);
}

__device__ __forceinline__ float rng_normal_w(float x) {
  // No contraction here: XLA evaluates log1p of the ROUNDED product -(x x), whose rounding error 1 / u amplifies (1e-5 of w
  // in the tail) — u must be fl(1 + fl(-x x)), not the fused 1 - x x, and c the exact error of that sum.
#pragma clang fp contract(off)
  const float y = -(x * x);
  const float u = 1.0f + y;                     // >= 2^-23: |x| <= 1 - 2^-24
  const float c = (u - 1.0f) - y;
  const float l2 = __builtin_amdgcn_logf(u);    // v_log_f32: log2
  const float l = __builtin_fmaf(l2, 0.693147182f, l2 * -1.90465498e-09f);    // times ln 2, in two pieces
  return -__builtin_fmaf(-c, __builtin_amdgcn_rcpf(u), l);
}

__device__ __forceinline__ float rng_normal_z(float x, float w) {
  const float v = w - 2.5f;
  float p = 2.81022636e-08f;
  p = 3.43273939e-07f + p * v;
  p = -3.5233877e-06f + p * v;
  p = -4.39150654e-06f + p * v;
  p = 0.00021858087f + p * v;
  p = -0.00125372503f + p * v;
  p = -0.00417768164f + p * v;
  p = 0.246640727f + p * v;
  p = 1.50140941f + p * v;
  const float t = __builtin_sqrtf(fmaxf(w, 0.0f)) - 3.0f;
  float q = -0.000200214257f;
  q = 0.000100950558f + q * t;
  q = 0.00134934322f + q * t;
  q = -0.00367342844f + q * t;
  q = 0.00573950773f + q * t;
  q = -0.0076224613f + q * t;
  q = 0.00943887047f + q * t;
  q = 1.00167406f + q * t;
  q = 2.83297682f + q * t;
  return 1.41421356f * ((w < 5.0f ? p : q) * x);
}

__device__ __forceinline__ float rng_bits_to_normal(uint32_t bits) {
  const float x = rng_normal_x(bits);
  return rng_normal_z(x, rng_normal_w(x));
}

// The N(0, 1) draw of loop iteration j for element idx of the (B, A) state; keys = the iteration keys (2 words each)
__device__ __forceinline__ float rng_step_normal(const uint32_t* __restrict__ keys, int impl, int j, unsigned long long n,
                                                 unsigned long long idx) {
  return rng_bits_to_normal(rng_bits(keys[2 * j], keys[2 * j + 1], impl, n, idx));
}

}  // namespace frl
