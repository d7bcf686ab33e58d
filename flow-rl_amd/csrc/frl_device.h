// Device-side helpers shared by the sampler kernels: vector types, compile-time loops, MFMA geometry traits,
// activations, buffer loads.  Not part of the public ABI.
#pragma once

#include <utility>

#include "frl_internal.h"

namespace frl {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int... I, typename F>
__device__ __forceinline__ void sfor_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
// compile-time loop: register-array indices stay constants whatever the unroller decides
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  sfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int TR>
struct Geo;
template <>
struct Geo<32> {
  static constexpr int KS = 2, NR = 16, RING = 4;
  typedef f32x16 acc_t;
  static constexpr int rowr(int r) { return (r & 3) + 8 * (r >> 2); }
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};
template <>
struct Geo<16> {
  static constexpr int KS = 4, NR = 4, RING = 2;
  typedef f32x4 acc_t;
  static constexpr int rowr(int r) { return (r & 3) + 16 * (r >> 2); }
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// ---- activations (flowrl/functional/activation.py:8-18; flax gelu tanh form) -------------------
// exp via v_exp_f32 and reciprocal via v_rcp_f32 (1 ulp each); every form below is free of
// cancellation so the relative error stays at a few ulp over the whole range.
// Internal activation id of the post-activation-LayerNorm builds (FQL's MLP, fql/network.py:26-30: Dense -> act -> LayerNorm):
// the LayerNorm pass applies the real activation in place before it normalises, every consumer then takes the tile as it is.
#define FRL_ACT_IDENTITY 6

template <int ACT>
__device__ __forceinline__ float act_fn(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (ACT == FRL_ACT_IDENTITY) {
    return x;
  } else if constexpr (ACT == FRL_ACT_MISH) {
    // x * tanh(softplus(x)) == x * n / (n + 2),  n = e^x (e^x + 2)
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -88.0f, 20.0f) * LOG2E);
    const float n = e * (e + 2.0f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
  } else if constexpr (ACT == FRL_ACT_SILU) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-x, -88.0f, 80.0f) * LOG2E);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
  } else if constexpr (ACT == FRL_ACT_RELU) {
    return fmaxf(x, 0.0f);
  } else if constexpr (ACT == FRL_ACT_GELU_TANH) {
    // 0.5 x (1 + tanh(u)) == x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)
    const float u = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-2.0f * u, -88.0f, 80.0f) * LOG2E);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
  } else if constexpr (ACT == FRL_ACT_TANH) {
    const float t = __builtin_amdgcn_exp2f(-2.0f * LOG2E * fabsf(x));
    const float y = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    return copysignf(y, x);
  } else {  // ELU
    const float e = __builtin_amdgcn_exp2f(fminf(x, 0.0f) * LOG2E);
    return x > 0.0f ? x : e - 1.0f;
  }
}

// Two activations at once: the plain multiply/add/fma steps become v_pk_*_f32 (one issue slot for both).
// The f32 MFMA shares the vector ALU, so VALU work is NOT hidden behind it (measured: +4.3..6.5 cycles per
// plain VALU, +9..12 per transcendental beside v_mfma_f32_32x32x2_f32; v_pk_fma_f32 costs what v_fma_f32 does).
template <int ACT>
__device__ __forceinline__ f32x2 act_fn2(f32x2 x) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (ACT == FRL_ACT_IDENTITY) {
    return x;
  } else if constexpr (ACT == FRL_ACT_MISH) {
    f32x2 c = {__builtin_amdgcn_fmed3f(x[0], -88.0f, 20.0f), __builtin_amdgcn_fmed3f(x[1], -88.0f, 20.0f)};
    c = c * LOG2E;
    const f32x2 e = {__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])};
    const f32x2 u = e + 2.0f;
    const f32x2 n = e * u;
    const f32x2 d = e * u + 2.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * (n * r);
  } else if constexpr (ACT == FRL_ACT_SILU) {
    f32x2 c = {__builtin_amdgcn_fmed3f(-x[0], -88.0f, 80.0f), __builtin_amdgcn_fmed3f(-x[1], -88.0f, 80.0f)};
    c = c * LOG2E;
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])} + 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
  } else if constexpr (ACT == FRL_ACT_GELU_TANH) {
    const f32x2 x2 = x * x;
    const f32x2 u = (x2 * (0.044715f * -2.0f * 0.7978845608028654f) + (-2.0f * 0.7978845608028654f)) * x;   // -2u
    f32x2 c = {__builtin_amdgcn_fmed3f(u[0], -88.0f, 80.0f), __builtin_amdgcn_fmed3f(u[1], -88.0f, 80.0f)};
    c = c * LOG2E;
    const f32x2 d = f32x2{__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])} + 1.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    return x * r;
  } else {
    return f32x2{act_fn<ACT>(x[0]), act_fn<ACT>(x[1])};
  }
}

// activation chosen at run time (uniform): the in-place activation of a post-activation LayerNorm pass
__device__ __forceinline__ float act_rt(int act, float x) {
  switch (act) {
    case FRL_ACT_MISH: return act_fn<FRL_ACT_MISH>(x);
    case FRL_ACT_SILU: return act_fn<FRL_ACT_SILU>(x);
    case FRL_ACT_RELU: return act_fn<FRL_ACT_RELU>(x);
    case FRL_ACT_GELU_TANH: return act_fn<FRL_ACT_GELU_TANH>(x);
    case FRL_ACT_TANH: return act_fn<FRL_ACT_TANH>(x);
    default: return act_fn<FRL_ACT_ELU>(x);
  }
}

__device__ __forceinline__ f32x4 ldw(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

__device__ __forceinline__ const float* pick_ptr_rs(const float* const (&arr)[FRL_MAX_HIDDEN - 1], int idx) {
  const float* p = arr[0];
#pragma unroll
  for (int k = 1; k < FRL_MAX_HIDDEN - 1; ++k) p = (idx == k) ? arr[k] : p;
  return p;
}

}  // namespace frl
