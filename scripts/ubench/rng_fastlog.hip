// Experiment (round 3): can the normal() transform of frl_rng.h drop ocml's log1pf (~120 VALU of double-float arithmetic)
// for   log1p(y) = ln2 * v_log_f32(u) - c / u,   u = fl(1 + y), c = (u - 1) - y  (the exact rounding error of the sum)?
// Sweeps all 2^23 mantissa patterns random_bits can produce and prints the largest deviation of w and of the final
// normal z from the ocml form, in units of 4e-7 max(1, |z|) (the parity bar of tests/test_gpu_rng.py).
//   hipcc --offload-arch=gfx950 -O3 -I include -I flow-rl_amd/csrc scripts/ubench/rng_fastlog.hip -o /tmp/rng_fastlog && /tmp/rng_fastlog
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include "flowrl_hip.h"
#include "frl_rng.h"

using namespace frl;

__device__ __forceinline__ float poly_normal(float x, float w) {
  float p;
  if (w < 5.0f) {
    const float v = w - 2.5f;
    p = 2.81022636e-08f; p = 3.43273939e-07f + p * v; p = -3.5233877e-06f + p * v; p = -4.39150654e-06f + p * v;
    p = 0.00021858087f + p * v; p = -0.00125372503f + p * v; p = -0.00417768164f + p * v; p = 0.246640727f + p * v; p = 1.50140941f + p * v;
  } else {
    const float v = sqrtf(w) - 3.0f;
    p = -0.000200214257f; p = 0.000100950558f + p * v; p = 0.00134934322f + p * v; p = -0.00367342844f + p * v;
    p = 0.00573950773f + p * v; p = -0.0076224613f + p * v; p = 0.00943887047f + p * v; p = 1.00167406f + p * v; p = 2.83297682f + p * v;
  }
  return 1.41421356f * (p * x);
}

__global__ void sweep(float* out /* [4] max dw_rel, max dz_bar, worst bits (as float), count over bar */) {
  __shared__ float s_w[256], s_z[256];
  float mw = 0.0f, mz = 0.0f;
  for (unsigned m = blockIdx.x * blockDim.x + threadIdx.x; m < (1u << 23); m += gridDim.x * blockDim.x) {
    const unsigned bits = m << 9;
    const float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
    const float lo = -0.99999994f;
    const float x = fmaxf(lo, f * 2.0f + lo);
    const float y = -(x * x);
    const float w_ref = -log1pf(y);
    const float u = 1.0f + y;
    const float c = (u - 1.0f) - y;
    const float l2 = __builtin_amdgcn_logf(u);                                  // v_log_f32: log2
    float l = __builtin_fmaf(l2, 0.693147182f, l2 * -1.90465498e-09f);          // ln2 in two pieces
    l = __builtin_fmaf(-c, __builtin_amdgcn_rcpf(u), l);
    const float w_fast = -l;
    const float zr = poly_normal(x, w_ref), zf = poly_normal(x, w_fast);
    const float dw = fabsf(w_fast - w_ref) / fmaxf(fabsf(w_ref), 1e-30f);
    const float dz = fabsf(zf - zr) / (4e-7f * fmaxf(1.0f, fabsf(zr)));
    if (w_ref > 1e-12f) mw = fmaxf(mw, dw);
    mz = fmaxf(mz, dz);
  }
  s_w[threadIdx.x] = mw; s_z[threadIdx.x] = mz;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 256; ++i) { mw = fmaxf(mw, s_w[i]); mz = fmaxf(mz, s_z[i]); }
    out[2 * blockIdx.x] = mw; out[2 * blockIdx.x + 1] = mz;
  }
}

int main() {
  float* d; hipMalloc(&d, 2 * 1024 * sizeof(float));
  hipLaunchKernelGGL(sweep, dim3(1024), dim3(256), 0, 0, d);
  float h[2048]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  float mw = 0, mz = 0;
  for (int i = 0; i < 1024; ++i) { mw = fmaxf(mw, h[2 * i]); mz = fmaxf(mz, h[2 * i + 1]); }
  printf("fast log1p vs ocml over 2^23 inputs: max rel dw = %.3e (%.2f ulp), max dz = %.3f x the 4e-7 bar\n", mw, mw / 5.96e-8f, mz);
  return 0;
}
