// Device helpers shared by the bf16-MFMA translation units (sampler_bf16.hip, sampler_bwd_bf16.hip): bf16 rounding /
// packing, v_mfma_f32_32x32x16_bf16, the LDS-DMA piece copy.  Not part of the public ABI.
#pragma once

#include "frl_device.h"

namespace frl {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// accumulator register r of lane half s of a 32x32 MFMA tile holds output row BF_ROWR(r) + 4 s
#define BF_ROWR(r) (((r) & 3) + 8 * ((r) >> 2))

__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even (NaN not expected in weights)
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

__device__ __forceinline__ f32x16 mfma_bf(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int OFF>
__device__ __forceinline__ void dma16_bf(const void* gsrc, void* lds_dst) {
  // 64 lanes x 16 B: per-lane global source, LDS destination = wave-uniform base + lane * 16; the instruction's
  // immediate offset OFF applies to both sides (pieces are 1 KB apart in the image and in the ring alike)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, OFF, 0);
}

// One solver update on a chain-state element, f32: shared by the bf16 sampler and the bf16 training forward, which must agree
// bit for bit (tests/test_gpu_vjp_bf16.py).  Contraction is OFF: which of two products the compiler fuses into an fma is
// otherwise its choice per call site, and the two kernels did get different choices once their bodies diverged.
__device__ __forceinline__ float bf_solver_update(int mode, int clip, float x_min, float x_max, float c0, float c1, float c2,
                                                  float c3, float c4, float c5, float xt, float eps, float z) {
#pragma clang fp contract(off)
  float xn;
  if (mode == MODE_DDPM_DDPM) {
    float x0h = c0 * (xt - c1 * eps);                       // ddpm.py:199
    if (clip) x0h = fminf(fmaxf(x0h, x_min), x_max);        // ddpm.py:200-201
    xn = c2 * (c3 * x0h + c4 * xt);                         // ddpm.py:202-203
    xn += c5 * z;                                           // ddpm.py:206
  } else if (mode == MODE_DDPM_DDIM) {
    xn = c0 * (xt - c1 * eps) + c2 * eps;                   // ddpm.py:211
    if (clip) xn = fminf(fmaxf(xn, x_min), x_max);
  } else if (mode == MODE_CDDPM_DDPM) {
    float x0h = (xt - c1 * eps) / c0;                       // continuous_ddpm.py:165
    if (clip) x0h = fminf(fmaxf(x0h, x_min), x_max);
    xn = c2 * x0h + c3 * xt;                                // :173
    xn += c5 * z;                                           // :174
  } else if (mode == MODE_CDDPM_DDIM) {
    float x0h = (xt - c1 * eps) / c0;
    if (clip) x0h = fminf(fmaxf(x0h, x_min), x_max);
    xn = c2 * x0h + c3 * eps;                               // :177
  } else {
    xn = xt + eps / c0;                                     // cnf.py:108
    if (clip) xn = fminf(fmaxf(xn, x_min), x_max);
  }
  return xn;
}

}  // namespace frl
