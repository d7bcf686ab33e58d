// cond_embedding MLP of DDPMBackbone / FlowBackbone (flowrl/flow/ddpm.py:89-90): every online agent uses
// MLP(hidden_dims=(128, 128), activation=actor_activation) with no output layer (dacer.py:263-266,
// qvpo.py:204-207, sdac.py:162-165, qsm.py:152-155, dpmd.py:206-209, dppo.py:246-248), i.e.
//     cfeat = act(act(obs @ W0 + b0) @ W1 + b1)            obs (n_states, S) -> cfeat (n_states, 128)
// It depends on the STATE only, so it runs once per state (not per particle, not per step): < 0.1 % of a
// sampler call's FLOPs.  Same transposed f32-MFMA formulation as the sampler: a wave owns 32 states, the
// 128 x 32 layer-1 tile stays in accumulators and feeds layer 2 as the B operand directly.
#include <utility>

#include "frl_internal.h"

namespace frl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int... I, typename F>
__device__ __forceinline__ void cfor_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void cfor(F&& f) { cfor_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }

__device__ __forceinline__ float cact(int act, float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  switch (act) {
    case FRL_ACT_MISH: {
      const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(x, -88.0f, 20.0f) * LOG2E);
      const float n = e * (e + 2.0f);
      return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
    }
    case FRL_ACT_SILU: {
      const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-x, -88.0f, 80.0f) * LOG2E);
      return x * __builtin_amdgcn_rcpf(1.0f + e);
    }
    case FRL_ACT_RELU: return fmaxf(x, 0.0f);
    case FRL_ACT_GELU_TANH: {
      const float u = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
      const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-2.0f * u, -88.0f, 80.0f) * LOG2E);
      return x * __builtin_amdgcn_rcpf(1.0f + e);
    }
    case FRL_ACT_TANH: {
      const float t = __builtin_amdgcn_exp2f(-2.0f * LOG2E * fabsf(x));
      return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
    }
    default: {
      const float e = __builtin_amdgcn_exp2f(fminf(x, 0.0f) * LOG2E);
      return x > 0.0f ? x : e - 1.0f;
    }
  }
}

#define CE_ROWR(r) (((r) & 3) + 8 * ((r) >> 2))

__global__ void __launch_bounds__(256)
frl_cond_embed_kernel(int act, const float* __restrict__ obs, long long n_states, int S, const float* w0, const float* b0,
                      const float* w1, const float* b1, float* __restrict__ cfeat) {
  constexpr int E = COND_EMBED_DIM, ROWB = E * 4;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const long long wt = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long state = wt * 32 + li;
  if (wt * 32 >= n_states) return;
  const bool valid = state < n_states;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)w0, 0, S * ROWB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)w1, 0, E * ROWB, 0x00020000);

  // ---- layer 1: acc[j][r] <- b0[f] + sum_k obs[state][k] W0[k][f],  f = 4 row(r,h) + j ----------------
  f32x16 acc[4];
  cfor<16>([&](auto RI) {
    constexpr int r = decltype(RI)::value;
    const f32x4 bv = *(const f32x4*)(b0 + 4 * (CE_ROWR(r) + 4 * lh));
    cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; acc[j][r] = bv[j]; });
  });
  const int nk = (S + 1) / 2;
  const float* orow = obs + (valid ? state : 0) * S;
#pragma unroll 4
  for (int s = 0; s < nk; ++s) {
    const int k = 2 * s + lh;
    const float bop = (valid && k < S) ? orow[k] : 0.0f;
    const f32x4 aw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, lh * ROWB + li * 16, s * 2 * ROWB, 0));
    cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j], bop, acc[j], 0, 0, 0); });
  }
  // ---- layer 2: out[j][r] <- b1[f] + sum_k act(h[k]) W1[k][f],  k-step (block jb, register r): k = 4 row(r,h) + jb ---
  f32x16 out[4];
  cfor<16>([&](auto RI) {
    constexpr int r = decltype(RI)::value;
    const f32x4 bv = *(const f32x4*)(b1 + 4 * (CE_ROWR(r) + 4 * lh));
    cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; out[j][r] = bv[j]; });
  });
  cfor<4>([&](auto BI) {
    constexpr int jb = decltype(BI)::value;
    cfor<16>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const float bop = cact(act, acc[jb][r]);
      const f32x4 aw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rs1, lh * (16 * ROWB) + li * 16, (jb + 4 * CE_ROWR(r)) * ROWB, 0));
      cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; out[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j], bop, out[j], 0, 0, 0); });
    });
  });
  if (valid) {
    float* crow = cfeat + state * E;
    cfor<16>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      f32x4 v;
      cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; v[j] = cact(act, out[j][r]); });
      *(f32x4*)(crow + 4 * (CE_ROWR(r) + 4 * lh)) = v;
    });
  }
}

// Per-state slice of a Dense layer: zs[state][f] = (bias[f] or 0) + sum_k obs[state][k] W[k][f], f < H (H a multiple of 128).
// Used to hoist the observation rows of a critic's Dense_0 out of the per-candidate work (critic.py:185-188 concatenates
// [obs, action]; the obs part is the same for every candidate of a state): S x H MACs once per STATE.  Same tiling as
// layer 1 above, one 128-feature group at a time.
__global__ void __launch_bounds__(256)
frl_state_bias_kernel(const float* __restrict__ obs, long long n_states, int S, int H, const float* w, const float* b,
                      float* __restrict__ zs) {
  const int ROWB = H * 4;
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const long long wt = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long state = wt * 32 + li;
  if (wt * 32 >= n_states) return;
  const bool valid = state < n_states;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, S * ROWB, 0x00020000);
  const int nk = (S + 1) / 2;
  const float* orow = obs + (valid ? state : 0) * S;
  for (int g = 0; g < H / 128; ++g) {
    f32x16 acc[4];
    cfor<16>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const f32x4 bv = (b != nullptr) ? *(const f32x4*)(b + 128 * g + 4 * (CE_ROWR(r) + 4 * lh)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; acc[j][r] = bv[j]; });
    });
#pragma unroll 4
    for (int s = 0; s < nk; ++s) {
      const int k = 2 * s + lh;
      const float bop = (valid && k < S) ? orow[k] : 0.0f;
      // lane (li, lh): W[k = 2 s + lh][128 g + 4 li .. + 3]; rows beyond S read as zero (descriptor bounds)
      const f32x4 aw = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lh * ROWB + 512 * g + li * 16, s * 2 * ROWB, 0));
      cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[j], bop, acc[j], 0, 0, 0); });
    }
    if (valid) {
      float* zrow = zs + state * H + 128 * g;
      cfor<16>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        f32x4 v;
        cfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; v[j] = acc[j][r]; });
        *(f32x4*)(zrow + 4 * (CE_ROWR(r) + 4 * lh)) = v;
      });
    }
  }
}

int launch_state_bias(const float* obs, long long n_states, int in_dim, int H, const float* w, const float* b, float* zs,
                      hipStream_t stream) {
  if (n_states <= 0) return FRL_OK;
  if (H % 128 != 0) { set_error("state bias: width %d is not a multiple of 128", H); return FRL_EUNSUPPORTED; }
  const long long n_wt = (n_states + 31) / 32;
  const int grid = (int)((n_wt + 3) / 4);
  hipLaunchKernelGGL(frl_state_bias_kernel, dim3(grid), dim3(256), 0, stream, obs, n_states, in_dim, H, w, b, zs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("state bias launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_cond_embed(int act, const float* obs, long long n_states, int in_dim, const float* w0, const float* b0,
                      const float* w1, const float* b1, float* cfeat, hipStream_t stream) {
  if (n_states <= 0) return FRL_OK;
  const long long n_wt = (n_states + 31) / 32;
  const int grid = (int)((n_wt + 3) / 4);
  hipLaunchKernelGGL(frl_cond_embed_kernel, dim3(grid), dim3(256), 0, stream, act, obs, n_states, in_dim, w0, b0, w1, b1, cfeat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("cond_embed launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
