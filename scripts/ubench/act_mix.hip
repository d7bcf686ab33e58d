// Microbenchmark: one hidden-layer k-step of the bf16 sampler = 8 v_mfma_f32_32x32x16_bf16 + the activation of NEL
// accumulator values per lane (+ cvt_pk), one wave per SIMD.  Prints cycles per k-step for several activation forms,
// to price transcendental vs plain VALU work beside the MFMA stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int FORM>
__device__ __forceinline__ float act(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (FORM == 0) { return x; }                                   // nothing
  else if constexpr (FORM == 1) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; }   // relu
  else if constexpr (FORM == 2) {                                            // mish: exp + rcp + 5 plain
    const float e = __builtin_amdgcn_exp2f(x * LOG2E);
    const float d = __builtin_fmaf(e, e + 2.0f, 2.0f);
    return __builtin_fmaf(-2.0f, x * __builtin_amdgcn_rcpf(d), x);
  } else if constexpr (FORM == 3) {                                          // exp only + 5 plain (no rcp)
    const float e = __builtin_amdgcn_exp2f(x * LOG2E);
    const float d = __builtin_fmaf(e, e + 2.0f, 2.0f);
    return __builtin_fmaf(-2.0f, x * d, x);
  } else if constexpr (FORM == 4) {                                          // 7 plain, no transcendental
    const float e = x * LOG2E;
    const float d = __builtin_fmaf(e, e + 2.0f, 2.0f);
    const float g = __builtin_fmaf(d, e, 1.0f);
    const float h = __builtin_fmaf(g, d, e);
    return __builtin_fmaf(-2.0f, x * h, x);
  } else {                                                                   // silu: exp + rcp + 3 plain
    const float e = __builtin_amdgcn_exp2f(x * -LOG2E);
    return x * __builtin_amdgcn_rcpf(e + 1.0f);
  }
}

template <int FORM, int NEL, int SPREAD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
kern(float* out, long long* cyc, int iters) {
  f32x16 Q[8], P[2];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) Q[i][r] = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) P[i][r] = threadIdx.x * 0.001f + r * 0.1f - 0.7f + i;
  u32x4 aw[8];
  for (int i = 0; i < 8; ++i) for (int w = 0; w < 4; ++w) aw[i][w] = 0x3f803f80u + threadIdx.x + i;
  u32x4 bop = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    u32x4 nb = bop;
#pragma unroll
    for (int w = 0; w < NEL / 2; ++w) {
      const f32x2 v = {act<FORM>(P[it & 1][2 * w]), act<FORM>(P[it & 1][2 * w + 1])};
      nb[w & 3] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
      Q[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw[m]), __builtin_bit_cast(bf16x8, bop), Q[m], 0, 0, 0);
    if (SPREAD) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (NEL * 9 + 7) / 8, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    bop = nb;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += Q[i][0] + Q[i][5];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + __builtin_bit_cast(float, bop[0]);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int FORM, int NEL, int SPREAD>
void run(float* out, long long* cyc, const char* name) {
  const int iters = 4000;
  hipLaunchKernelGGL((kern<FORM, NEL, SPREAD>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-22s elements/k-step %2d spread %d : %.1f cycles per k-step (8 MFMA = 256)\n", name, NEL, SPREAD, (double)c / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, 8, 1>(out, cyc, "none(cvt only)");
  run<1, 8, 1>(out, cyc, "relu"); run<1, 16, 1>(out, cyc, "relu");
  run<2, 4, 1>(out, cyc, "mish exp+rcp"); run<2, 8, 1>(out, cyc, "mish exp+rcp"); run<2, 8, 0>(out, cyc, "mish exp+rcp"); run<2, 16, 1>(out, cyc, "mish exp+rcp");
  run<3, 8, 1>(out, cyc, "exp only +5"); run<3, 16, 1>(out, cyc, "exp only +5");
  run<4, 8, 1>(out, cyc, "7 plain"); run<4, 16, 1>(out, cyc, "7 plain");
  run<5, 8, 1>(out, cyc, "silu exp+rcp"); run<5, 16, 1>(out, cyc, "silu exp+rcp");
  return 0;
}
