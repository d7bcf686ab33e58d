// Shared by the chain-VJP kernels (sampler_bwd.hip) and the stash-writing latency forward (sampler_lat.hip): the 8-byte
// stash store and every activation together with its derivative.
#pragma once
#include "frl_device.h"

namespace frl {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ void st2(f32x2 v, __amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, soff, 0);
}

// activation and its derivative for a register pair (shared exp / rcp)
template <int ACT>
__device__ __forceinline__ void act_grad2(f32x2 z, f32x2& h, f32x2& d) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (ACT == FRL_ACT_MISH) {
    // h = z n / (n + 2), n = e (e + 2);  h' = n r + 4 z e (e + 1) r^2,  r = 1 / (n + 2)
    f32x2 c = {__builtin_amdgcn_fmed3f(z[0], -88.0f, 20.0f), __builtin_amdgcn_fmed3f(z[1], -88.0f, 20.0f)};
    c = c * LOG2E;
    const f32x2 e = {__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])};
    const f32x2 u = e + 2.0f;
    const f32x2 n = e * u;
    const f32x2 dd = e * u + 2.0f;
    const f32x2 r = {__builtin_amdgcn_rcpf(dd[0]), __builtin_amdgcn_rcpf(dd[1])};
    const f32x2 nr = n * r;
    const f32x2 q = e * (e + 1.0f);
    const f32x2 zr = z * r;
    h = z * nr;
    d = (q * 4.0f) * (zr * r) + nr;
  } else if constexpr (ACT == FRL_ACT_SILU) {
    f32x2 c = {__builtin_amdgcn_fmed3f(-z[0], -88.0f, 80.0f), __builtin_amdgcn_fmed3f(-z[1], -88.0f, 80.0f)};
    c = c * LOG2E;
    const f32x2 dd = f32x2{__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])} + 1.0f;
    const f32x2 s = {__builtin_amdgcn_rcpf(dd[0]), __builtin_amdgcn_rcpf(dd[1])};
    h = z * s;
    d = h * (1.0f - s) + s;
  } else if constexpr (ACT == FRL_ACT_RELU) {
    h = f32x2{fmaxf(z[0], 0.0f), fmaxf(z[1], 0.0f)};
    d = f32x2{z[0] > 0.0f ? 1.0f : 0.0f, z[1] > 0.0f ? 1.0f : 0.0f};
  } else if constexpr (ACT == FRL_ACT_GELU_TANH) {
    // h = z s, s = sigmoid(2u), u = c (z + k z^3) (the forward's form, frl_device.h);  h' = s + h (1 - s) 2 c (1 + 3 k z^2)
    constexpr float C2 = 2.0f * 0.7978845608028654f, K = 0.044715f;
    const f32x2 z2 = z * z;
    const f32x2 m2u = (z2 * (K * -C2) + (-C2)) * z;                                  // -2u
    f32x2 c = {__builtin_amdgcn_fmed3f(m2u[0], -88.0f, 80.0f), __builtin_amdgcn_fmed3f(m2u[1], -88.0f, 80.0f)};
    c = c * LOG2E;
    const f32x2 dd = f32x2{__builtin_amdgcn_exp2f(c[0]), __builtin_amdgcn_exp2f(c[1])} + 1.0f;
    const f32x2 s = {__builtin_amdgcn_rcpf(dd[0]), __builtin_amdgcn_rcpf(dd[1])};
    h = z * s;
    d = (h * (1.0f - s)) * (z2 * (3.0f * K * C2) + C2) + s;
  } else if constexpr (ACT == FRL_ACT_TANH) {
    h = f32x2{act_fn<FRL_ACT_TANH>(z[0]), act_fn<FRL_ACT_TANH>(z[1])};
    d = 1.0f - h * h;
  } else {  // ELU: h = z (z > 0) | e^z - 1;  h' = 1 | e^z
    const f32x2 e = {__builtin_amdgcn_exp2f(fminf(z[0], 0.0f) * LOG2E), __builtin_amdgcn_exp2f(fminf(z[1], 0.0f) * LOG2E)};
    h = f32x2{z[0] > 0.0f ? z[0] : e[0] - 1.0f, z[1] > 0.0f ? z[1] : e[1] - 1.0f};
    d = f32x2{z[0] > 0.0f ? 1.0f : e[0], z[1] > 0.0f ? 1.0f : e[1]};
  }
}

}  // namespace frl
