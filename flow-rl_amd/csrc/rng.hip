// jax.random key chain of the samplers' scans and the standalone split / normal / bits entry points (frl_rng.h has
// the device functions; oracle/threefry_oracle.py the restatement they are checked against).
#include "frl_internal.h"
#include "frl_rng.h"

namespace frl {

__device__ __forceinline__ void split_word(uint32_t k0, uint32_t k1, int impl, int num, int i, uint32_t& o0, uint32_t& o1) {
  // split(key, num)[i]: partitionable = the block of counter i; legacy = words 2i, 2i+1 of the 2 num-word draw
  if (impl == FRL_RNG_THREEFRY_PARTITIONABLE) { threefry2x32(k0, k1, 0u, (uint32_t)i, o0, o1); return; }
  o0 = rng_bits(k0, k1, impl, 2ull * num, 2ull * i);
  o1 = rng_bits(k0, k1, impl, 2ull * num, 2ull * i + 1);
}

// The split sequence inside the reference's scans (one thread: 2..3 blocks per iteration):
//   DDPM   (ddpm.py:188,204):            rng, drop = split(rng);  ddpm solver: rng, key = split(rng)
//   CDDPM  (continuous_ddpm.py:161,190): rng, drop, key = split(rng, 3);  dpm2: rng, drop2 = split(rng)
//   CNF    (cnf.py:126):                 rng, drop = split(rng, 2)
// keys[e] = key of network evaluation row e that draws noise (ddpm solvers: e = iteration), rng_out = final rng.
__global__ void frl_rng_chain_kernel(const uint32_t* __restrict__ key_in, int impl, int mode, int T, uint32_t* __restrict__ keys,
                                     uint32_t* __restrict__ rng_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t r0 = key_in[0], r1 = key_in[1];
  for (int j = 0; j < T; ++j) {
    uint32_t n0, n1;
    if (mode == MODE_DDPM_DDPM || mode == MODE_DDPM_DDIM) {
      split_word(r0, r1, impl, 2, 0, n0, n1); r0 = n0; r1 = n1;
      if (mode == MODE_DDPM_DDPM) {
        uint32_t k0, k1;
        split_word(r0, r1, impl, 2, 1, k0, k1);
        split_word(r0, r1, impl, 2, 0, n0, n1); r0 = n0; r1 = n1;
        keys[2 * j] = k0; keys[2 * j + 1] = k1;
      }
    } else if (mode == MODE_CNF_EULER) {
      split_word(r0, r1, impl, 2, 0, n0, n1); r0 = n0; r1 = n1;
    } else {
      if (mode == MODE_CDDPM_DDPM) {
        uint32_t k0, k1;
        split_word(r0, r1, impl, 3, 2, k0, k1);
        keys[2 * j] = k0; keys[2 * j + 1] = k1;
      }
      split_word(r0, r1, impl, 3, 0, n0, n1); r0 = n0; r1 = n1;
      if (mode == MODE_CDDPM_DPM2) { split_word(r0, r1, impl, 2, 0, n0, n1); r0 = n0; r1 = n1; }
    }
  }
  if (rng_out != nullptr) { rng_out[0] = r0; rng_out[1] = r1; }
}

int launch_rng_chain(const uint32_t* key_in, int impl, int mode, int T, uint32_t* keys, uint32_t* rng_out, hipStream_t stream) {
  hipLaunchKernelGGL(frl_rng_chain_kernel, dim3(1), dim3(64), 0, stream, key_in, impl, mode, T, keys, rng_out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("rng chain launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

__global__ void frl_random_split_kernel(const uint32_t* __restrict__ key, int impl, int num, uint32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num) return;
  uint32_t o0, o1;
  split_word(key[0], key[1], impl, num, i, o0, o1);
  out[2 * i] = o0; out[2 * i + 1] = o1;
}

// KIND 0: raw bits; 1: normal; 2: gumbel = -log(-log(u)), u = uniform(minval = FLT_MIN, maxval = 1): the mantissa-trick
// float f in [0, 1) times (1 - tiny) (= 1 in f32) plus tiny, floored at tiny, i.e. f, or tiny where f = 0
template <int KIND>
__global__ void frl_random_fill_kernel(const uint32_t* __restrict__ key, int impl, unsigned long long n, void* __restrict__ out) {
  const uint32_t k0 = key[0], k1 = key[1];
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t b = rng_bits(k0, k1, impl, n, i);
    if (KIND == 1) ((float*)out)[i] = rng_bits_to_normal(b);
    else if (KIND == 2) {
      const float tiny = 1.17549435e-38f;
      const float f = __uint_as_float((b >> 9) | 0x3F800000u) - 1.0f;
      const float u = fmaxf(tiny, f * 1.0f + tiny);
      ((float*)out)[i] = -logf(-logf(u));
    } else ((uint32_t*)out)[i] = b;
  }
}

// jax.random.uniform(key, shape, minval, maxval): mantissa-trick float in [0, 1) * (maxval - minval) + minval, floored at
// minval (jax/_src/random.py _uniform) — the t draw of ContinuousDDPM.add_noise / CNF.linear_interpolation
__global__ void frl_random_uniform_kernel(const uint32_t* __restrict__ key, int impl, unsigned long long n, float lo, float hi, float* __restrict__ out) {
#pragma clang fp contract(off)      // separate multiply and add, as XLA:CPU evaluates floats * (maxval - minval) + minval
  const uint32_t k0 = key[0], k1 = key[1];
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t b = rng_bits(k0, k1, impl, n, i);
    const float f = __uint_as_float((b >> 9) | 0x3F800000u) - 1.0f;
    const float scaled = f * (hi - lo);
    out[i] = fmaxf(lo, scaled + lo);
  }
}

// jax.random.randint(key, shape, minval, maxval) for int32 (jax/_src/random.py _randint): k1, k2 = split(key); two uint32
// draws (higher, lower); span = maxval - minval; multiplier = (2^16 mod span)^2 mod span;
// offset = ((higher mod span) * multiplier + lower mod span) mod span in wrapping uint32 arithmetic — the t draw of DDPM.add_noise
__global__ void frl_random_randint_kernel(const uint32_t* __restrict__ key, int impl, unsigned long long n, int32_t minval, int32_t maxval,
                                          int32_t* __restrict__ out) {
  uint32_t a0, a1, b0, b1;
  split_word(key[0], key[1], impl, 2, 0, a0, a1);
  split_word(key[0], key[1], impl, 2, 1, b0, b1);
  uint32_t span = (uint32_t)(maxval - minval);
  if (maxval <= minval) span = 1u;
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    const uint32_t hi = rng_bits(a0, a1, impl, n, i), lo = rng_bits(b0, b1, impl, n, i);
    const uint32_t off = ((hi % span) * mult + (lo % span)) % span;
    out[i] = minval + (int32_t)off;
  }
}

static int check_impl(int impl) {
  if (impl == FRL_RNG_THREEFRY_PARTITIONABLE || impl == FRL_RNG_THREEFRY_LEGACY) return FRL_OK;
  set_error("rng impl %d: FRL_RNG_THREEFRY_PARTITIONABLE or FRL_RNG_THREEFRY_LEGACY expected", impl);
  return FRL_EINVAL;
}

int launch_random_split(const uint32_t* key, int impl, int num, uint32_t* out, hipStream_t stream) {
  int rc = check_impl(impl);
  if (rc != FRL_OK) return rc;
  if (key == nullptr || out == nullptr || num < 1) { set_error("random_split: bad arguments"); return FRL_EINVAL; }
  hipLaunchKernelGGL(frl_random_split_kernel, dim3((num + 255) / 256), dim3(256), 0, stream, key, impl, num, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("random_split launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_random_fill(const uint32_t* key, int impl, long long n, void* out, int kind, hipStream_t stream) {
  int rc = check_impl(impl);
  if (rc != FRL_OK) return rc;
  if (key == nullptr || (out == nullptr && n > 0) || n < 0) { set_error("random fill: bad arguments"); return FRL_EINVAL; }
  if (impl == FRL_RNG_THREEFRY_LEGACY && n > 0xFFFFFFFFll) { set_error("legacy threefry layout: at most 2^32 - 1 elements per draw"); return FRL_EUNSUPPORTED; }
  if (n == 0) return FRL_OK;
  const long long want = (n + 255) / 256, cap = 16ll * device_cus();
  const int grid = (int)(want < cap ? want : cap);
  if (kind == 1) hipLaunchKernelGGL(frl_random_fill_kernel<1>, dim3(grid), dim3(256), 0, stream, key, impl, (unsigned long long)n, out);
  else if (kind == 2) hipLaunchKernelGGL(frl_random_fill_kernel<2>, dim3(grid), dim3(256), 0, stream, key, impl, (unsigned long long)n, out);
  else hipLaunchKernelGGL(frl_random_fill_kernel<0>, dim3(grid), dim3(256), 0, stream, key, impl, (unsigned long long)n, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("random fill launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_random_uniform(const uint32_t* key, int impl, long long n, float lo, float hi, float* out, hipStream_t stream) {
  int rc = check_impl(impl);
  if (rc != FRL_OK) return rc;
  if (key == nullptr || (out == nullptr && n > 0) || n < 0) { set_error("random uniform: bad arguments"); return FRL_EINVAL; }
  if (impl == FRL_RNG_THREEFRY_LEGACY && n > 0xFFFFFFFFll) { set_error("legacy threefry layout: at most 2^32 - 1 elements per draw"); return FRL_EUNSUPPORTED; }
  if (n == 0) return FRL_OK;
  const long long want = (n + 255) / 256, cap = 16ll * device_cus();
  hipLaunchKernelGGL(frl_random_uniform_kernel, dim3((int)(want < cap ? want : cap)), dim3(256), 0, stream, key, impl, (unsigned long long)n, lo, hi, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("random uniform launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_random_randint(const uint32_t* key, int impl, long long n, int32_t minval, int32_t maxval, int32_t* out, hipStream_t stream) {
  int rc = check_impl(impl);
  if (rc != FRL_OK) return rc;
  if (key == nullptr || (out == nullptr && n > 0) || n < 0) { set_error("random randint: bad arguments"); return FRL_EINVAL; }
  if (impl == FRL_RNG_THREEFRY_LEGACY && n > 0xFFFFFFFFll) { set_error("legacy threefry layout: at most 2^32 - 1 elements per draw"); return FRL_EUNSUPPORTED; }
  if (n == 0) return FRL_OK;
  const long long want = (n + 255) / 256, cap = 16ll * device_cus();
  hipLaunchKernelGGL(frl_random_randint_kernel, dim3((int)(want < cap ? want : cap)), dim3(256), 0, stream, key, impl, (unsigned long long)n, minval, maxval, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("random randint launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
