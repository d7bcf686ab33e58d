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

}  // namespace frl
