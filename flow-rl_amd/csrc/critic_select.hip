// Best-of-N selection after critic scoring: min over the ensemble, argmax (or Gumbel-max
// categorical) over a state's candidates, gather of the chosen action.
//   flowrl/agent/offline/dql.py:155-161   qs.min(axis=0).reshape(B, N); argmax / categorical; actions[arange(B), idx]
//   flowrl/agent/online/qvpo.py:40-43     same with argmax
// HBM-bound and tiny next to the scoring MLP: 4 (E + 1) bytes per candidate read, A + 1 words per state written.
// One wave per state; lanes stride over the candidates, ties resolve to the LOWEST index (jnp.argmax).
#include "frl_internal.h"

namespace frl {

__global__ void __launch_bounds__(256) frl_critic_select_kernel(const float* __restrict__ qm, int E, long long n_states, int N, int A,
                                                                const float* __restrict__ actions, const float* __restrict__ gumbel,
                                                                float inv_temp, float* __restrict__ q, int32_t* __restrict__ best_idx,
                                                                float* __restrict__ best_action) {
  const int lane = threadIdx.x & 63;
  const long long B = n_states * N;
  for (long long s = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < n_states;
       s += (long long)gridDim.x * (blockDim.x >> 6)) {
    float best = 0.0f;
    int bi = 0x7fffffff;                                  // "no candidate yet" sorts after every real index
    for (int p = lane; p < N; p += 64) {
      const long long c = s * N + p;
      float v = qm[c];
      for (int e = 1; e < E; ++e) v = fminf(v, qm[(size_t)e * B + c]);     // qs.min(axis=0)
      if (q != nullptr) q[c] = v;
      const float key = (gumbel != nullptr) ? fmaf(v, inv_temp, gumbel[c]) : v; // categorical == argmax(logits + Gumbel)
      if (bi == 0x7fffffff || key > best) { best = key; bi = p; }           // strict >: the first maximum wins
    }
    // wave reduction: larger key wins, equal keys -> smaller index
    for (int off = 32; off >= 1; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bi, off);
      const bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi));
      if (take) { best = ob; bi = oi; }
    }
    if (best_idx != nullptr && lane == 0) best_idx[s] = bi;
    if (best_action != nullptr) {
      const float* src = actions + (s * N + bi) * (long long)A;
      for (int a = lane; a < A; a += 64) best_action[s * A + a] = src[a];
    }
  }
}

// the fused sampler's per-member pointer table (frl_internal.h CriticTab.ptrs), from kernel arguments to device memory
__global__ void __launch_bounds__(64) frl_critic_tab_kernel(const CriticPtrs v, const float** __restrict__ dst) {
  const int i = threadIdx.x;
  if (i < FUSE_MAX_E * CT_STRIDE) dst[i] = v.p[i];
}

int launch_critic_tab(const CriticPtrs& v, const float** dst, hipStream_t stream) {
  hipLaunchKernelGGL(frl_critic_tab_kernel, dim3(1), dim3(64), 0, stream, v, dst);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("critic table launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_critic_select(const float* q_members, int E, long long n_states, int particles, int A, const float* actions,
                         const float* gumbel, float inv_temperature, float* q, int32_t* best_idx, float* best_action,
                         hipStream_t stream) {
  if (n_states <= 0) return FRL_OK;
  const long long blocks = (n_states + 3) / 4;
  const int grid = (int)(blocks < 4096 ? blocks : 4096);
  hipLaunchKernelGGL(frl_critic_select_kernel, dim3(grid), dim3(256), 0, stream, q_members, E, n_states, particles, A, actions,
                     gumbel, inv_temperature, q, best_idx, best_action);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("critic select launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl

// ---- MFMA lane-map self test (tests only): the operand / accumulator maps every f32 kernel relies on --------------
namespace frl {

__global__ void __launch_bounds__(64) frl_selftest_mfma32_kernel(const float* A, const float* B, float* D, int K) {
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k + lh], B[(k + lh) * 32 + li], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[r];   // row(r, h) of the 32x32 accumulator
}

int launch_selftest_mfma32(const float* A, const float* B, float* D, int K, hipStream_t stream) {
  hipLaunchKernelGGL(frl_selftest_mfma32_kernel, dim3(1), dim3(64), 0, stream, A, B, D, K);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("selftest launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
