// Microbenchmark: how many VALU fillers hide in the shadow of v_mfma_f32_32x32x2_f32 / 16x16x4_f32
// with ONE wave per SIMD (the sampler kernel's regime).  Prints cycles per MFMA for each filler count.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NF, int KIND, int SHAPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
kern(float* out, long long* cyc, int iters, const float* wbuf) {
  f32x16 acc[8];
  f32x4 acc4[8];
  for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(threadIdx.x * 1e-3f + i); hb[i] = (__bf16)1.0f; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 0.01f + i;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk[4];
  for (int i = 0; i < 4; ++i) { pk[i][0] = f[i]; pk[i][1] = f[i + 4]; }
  f32x4 ld[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 4; ++r) ld[i][r] = 0.f;
  unsigned voff = threadIdx.x * 16;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wbuf, 0, 1 << 20, 0x00020000);
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (SHAPE == 3216) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[m], 0, 0, 0);
      else if (SHAPE == 32) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      else acc4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k % 8]) : "v"(b));          // independent-ish plain VALU
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k % 8]));                        // transcendental
        else if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[0]) : "v"(b));                          // dependent chain
        else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pk[k % 4]) : "v"(pk[(k + 1) % 4]));  // packed f32
        else if (KIND == 4) asm volatile("s_nop 0");
        else if (KIND == 5) asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(f[k % 8]));
        else if (KIND == 6) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(ld[k % 4]) : "v"(voff), "s"(rs) : "memory");
        else if (KIND == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[k % 4]) : "v"(voff) : "memory");
        else if (KIND == 8) asm volatile("s_mov_b32 s90, 0x1234" ::: "s90");
        else if (KIND == 9) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[k % 8]) : "v"(f[(k + 1) % 8]), "v"(f[(k + 2) % 8]));
        else if (KIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[k % 4]) : "v"(pk[(k + 1) % 4]));
        else if (KIND == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k % 8]));
        else if (KIND == 12) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[k % 8]) : "v"(b));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc4[i][0] + f[i];
  for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1] + ld[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NF, int KIND, int SHAPE>
void run(float* out, long long* cyc, const char* name) {
  static float* wb = nullptr; if (!wb) { hipMalloc(&wb, 1 << 20); hipMemset(wb, 0, 1 << 20); }
  const int iters = 2000;
  hipLaunchKernelGGL((kern<NF, KIND, SHAPE>), dim3(256), dim3(256), 0, 0, out, cyc, iters, (const float*)wb);
  hipDeviceSynchronize();
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("shape %4d %-10s fillers/gap %2d : %.1f cycles per MFMA\n", SHAPE, name, NF, (double)c / (iters * 8.0));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  run<0, 0, 32>(out, cyc, "plain"); run<2, 0, 32>(out, cyc, "plain"); run<4, 0, 32>(out, cyc, "plain");
  run<6, 0, 32>(out, cyc, "plain"); run<8, 0, 32>(out, cyc, "plain"); run<12, 0, 32>(out, cyc, "plain");
  run<1, 1, 32>(out, cyc, "trans"); run<2, 1, 32>(out, cyc, "trans"); run<4, 1, 32>(out, cyc, "trans");
  run<2, 2, 32>(out, cyc, "dep"); run<4, 2, 32>(out, cyc, "dep"); run<8, 2, 32>(out, cyc, "dep");
  run<2, 3, 32>(out, cyc, "pk_fma"); run<4, 3, 32>(out, cyc, "pk_fma");
  run<2, 4, 32>(out, cyc, "s_nop"); run<4, 4, 32>(out, cyc, "s_nop");
  run<2, 5, 32>(out, cyc, "accread"); run<4, 5, 32>(out, cyc, "accread");
  run<1, 6, 32>(out, cyc, "bufload"); run<2, 6, 32>(out, cyc, "bufload");
  run<1, 7, 32>(out, cyc, "ds_read"); run<2, 7, 32>(out, cyc, "ds_read");
  run<2, 8, 32>(out, cyc, "s_mov"); run<4, 8, 32>(out, cyc, "s_mov");
  run<0, 0, 16>(out, cyc, "plain"); run<1, 0, 16>(out, cyc, "plain"); run<2, 0, 16>(out, cyc, "plain");
  run<4, 0, 16>(out, cyc, "plain"); run<1, 1, 16>(out, cyc, "trans"); run<2, 2, 16>(out, cyc, "dep");
  // 32x32x16 bf16 (the bf16 sampler's regime)
  run<0, 0, 3216>(out, cyc, "plain"); run<2, 0, 3216>(out, cyc, "plain"); run<4, 0, 3216>(out, cyc, "plain");
  run<6, 0, 3216>(out, cyc, "plain"); run<7, 0, 3216>(out, cyc, "plain"); run<8, 0, 3216>(out, cyc, "plain"); run<10, 0, 3216>(out, cyc, "plain");
  run<1, 1, 3216>(out, cyc, "trans"); run<2, 1, 3216>(out, cyc, "trans"); run<4, 1, 3216>(out, cyc, "trans");
  run<2, 11, 3216>(out, cyc, "rcp"); run<2, 12, 3216>(out, cyc, "med3"); run<4, 12, 3216>(out, cyc, "med3");
  run<2, 3, 3216>(out, cyc, "pk_fma"); run<4, 3, 3216>(out, cyc, "pk_fma"); run<6, 3, 3216>(out, cyc, "pk_fma");
  run<2, 10, 3216>(out, cyc, "pk_mul"); run<4, 10, 3216>(out, cyc, "pk_mul");
  run<1, 9, 3216>(out, cyc, "cvt_pk"); run<2, 9, 3216>(out, cyc, "cvt_pk");
  run<1, 7, 3216>(out, cyc, "ds_read"); run<2, 7, 3216>(out, cyc, "ds_read");
  run<1, 6, 3216>(out, cyc, "bufload");
  run<4, 5, 3216>(out, cyc, "accread");
  return 0;
}
