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

// ---- the same two kernels with the FEATURES split over waves -------------------------------------------------------------
// Above, one wave carries all 128 features of its 32 states: four MFMAs per k-step behind one 16-byte load, S / 2 dependent
// k-steps — 65 .. 150 us however few states there are (an acting call has ONE), because a call of n states runs n / 32 waves.
// Here a wave owns ONE of the four 32x32 tiles (features 4 i + j of its group): one MFMA per k-step, operands prefetched 32
// k-steps ahead in registers, four (cond_embed) or H / 32 (state bias) times as many waves.  Per output element the k order and
// the instruction are those of the kernels above, so the results are the same bits; the launchers take these up to 32768 states.
constexpr int CE_CHUNK = 32;    // k-steps per prefetch chunk (two register buffers)
constexpr long long CE_WIDE_MAX_STATES = 32768;

// acc <- acc + sum over k-steps [0, nk) of A[k][col] * bop(k), one MFMA per k-step, in k order.
//   wload(s): this lane's A operand of k-step s; bload(s): its B operand.  Both may be called for s >= nk (must return 0 / be safe).
template <typename WLoad, typename BLoad>
__device__ __forceinline__ f32x16 ce_contract(f32x16 acc, int nk, WLoad&& wload, BLoad&& bload) {
  float wa[2][CE_CHUNK], ba[2][CE_CHUNK];
  cfor<CE_CHUNK>([&](auto I) { constexpr int i = decltype(I)::value; wa[0][i] = wload(i); ba[0][i] = bload(i); });
  for (int c = 0; c * CE_CHUNK < nk; c += 2) {
    cfor<CE_CHUNK>([&](auto I) { constexpr int i = decltype(I)::value; wa[1][i] = wload((c + 1) * CE_CHUNK + i); ba[1][i] = bload((c + 1) * CE_CHUNK + i); });
    cfor<CE_CHUNK>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if (c * CE_CHUNK + i < nk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[0][i], ba[0][i], acc, 0, 0, 0);
    });
    cfor<CE_CHUNK>([&](auto I) { constexpr int i = decltype(I)::value; wa[0][i] = wload((c + 2) * CE_CHUNK + i); ba[0][i] = bload((c + 2) * CE_CHUNK + i); });
    cfor<CE_CHUNK>([&](auto I) {
      constexpr int i = decltype(I)::value;
      if ((c + 1) * CE_CHUNK + i < nk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[1][i], ba[1][i], acc, 0, 0, 0);
    });
  }
  return acc;
}

// grid (H / 32, state tiles of 32), one wave per block: feature block fb = 4 g + j <-> features 128 g + 4 i + j
__global__ void __launch_bounds__(64)
frl_state_bias_wide_kernel(const float* __restrict__ obs, long long n_states, int S, int H, const float* w, const float* b,
                           float* __restrict__ zs) {
  const int ROWB = H * 4;
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const int g = blockIdx.x >> 2, j = blockIdx.x & 3;
  const long long state = (long long)blockIdx.y * 32 + li;
  const bool valid = state < n_states;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, S * ROWB, 0x00020000);
  const int nk = (S + 1) / 2;
  const float* orow = obs + (valid ? state : 0) * S;
  f32x16 acc;
  cfor<16>([&](auto RI) {
    constexpr int r = decltype(RI)::value;
    acc[r] = (b != nullptr) ? b[128 * g + 4 * (CE_ROWR(r) + 4 * lh) + j] : 0.0f;
  });
  const unsigned voff = (unsigned)(lh * ROWB + 512 * g + li * 16 + j * 4);
  acc = ce_contract(acc, nk,
      [&](int s) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, s * 2 * ROWB, 0)); },   // rows >= S: zero (descriptor bounds)
      [&](int s) { const int k = 2 * s + lh; return (valid && k < S) ? orow[k] : 0.0f; });
  if (valid) {
    float* zrow = zs + state * H + 128 * g;
    cfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; zrow[4 * (CE_ROWR(r) + 4 * lh) + j] = acc[r]; });
  }
}

// grid = state tiles of 32, four waves per block: wave j owns features 4 i + j of both layers; layer 1 reaches layer 2 through LDS
__global__ void __launch_bounds__(256)
frl_cond_embed_wide_kernel(int act, const float* __restrict__ obs, long long n_states, int S, const float* w0, const float* b0,
                           const float* w1, const float* b1, float* __restrict__ cfeat) {
  constexpr int E = COND_EMBED_DIM, ROWB = E * 4;
  __shared__ float s_h[E * 32];          // [feature][state of the tile]
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long state = (long long)blockIdx.x * 32 + li;
  const bool valid = state < n_states;
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)w0, 0, S * ROWB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)w1, 0, E * ROWB, 0x00020000);
  const int nk = (S + 1) / 2;
  const float* orow = obs + (valid ? state : 0) * S;
  // ---- layer 1 (k order and operands of frl_cond_embed_kernel's tile j) ----
  f32x16 acc;
  cfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; acc[r] = b0[4 * (CE_ROWR(r) + 4 * lh) + j]; });
  const unsigned voff0 = (unsigned)(lh * ROWB + li * 16 + j * 4);
  acc = ce_contract(acc, nk,
      [&](int s) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs0, voff0, s * 2 * ROWB, 0)); },
      [&](int s) { const int k = 2 * s + lh; return (valid && k < S) ? orow[k] : 0.0f; });
  cfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; s_h[(4 * (CE_ROWR(r) + 4 * lh) + j) * 32 + li] = cact(act, acc[r]); });
  __syncthreads();
  // ---- layer 2: k-step q = 16 jb + r contracts k = 4 (ROWR(r) + 4 lh) + jb, in that order (frl_cond_embed_kernel) ----
  f32x16 out;
  cfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; out[r] = b1[4 * (CE_ROWR(r) + 4 * lh) + j]; });
  const unsigned voff1 = (unsigned)(lh * (16 * ROWB) + li * 16 + j * 4);
  cfor<4>([&](auto BI) {
    constexpr int jb = decltype(BI)::value;
    float wv[16], hv[16];
    cfor<16>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      wv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, voff1, (jb + 4 * CE_ROWR(r)) * ROWB, 0));
      hv[r] = s_h[(4 * (CE_ROWR(r) + 4 * lh) + jb) * 32 + li];
    });
    cfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; out = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[r], hv[r], out, 0, 0, 0); });
  });
  if (valid) {
    float* crow = cfeat + state * E;
    cfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; crow[4 * (CE_ROWR(r) + 4 * lh) + j] = cact(act, out[r]); });
  }
}

// ---- cond_embedding backward (DACER's actor loss differentiates the embedding too: dacer.py:141-162, ddpm.py:89-90) -----
// activation and derivative in one go (the three activations the chain VJP supports)
__device__ __forceinline__ void cact_grad(int act, float z, float& h, float& d) {
  constexpr float LOG2E = 1.4426950408889634f;
  if (act == FRL_ACT_MISH) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(z, -88.0f, 20.0f) * LOG2E);
    const float n = e * (e + 2.0f), r = __builtin_amdgcn_rcpf(n + 2.0f);
    h = z * (n * r);
    d = n * r + 4.0f * z * e * (e + 1.0f) * r * r;
  } else if (act == FRL_ACT_SILU) {
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-z, -88.0f, 80.0f) * LOG2E);
    const float sg = __builtin_amdgcn_rcpf(1.0f + e);
    h = z * sg;
    d = h * (1.0f - sg) + sg;
  } else if (act == FRL_ACT_RELU) {
    h = fmaxf(z, 0.0f);
    d = z > 0.0f ? 1.0f : 0.0f;
  } else if (act == FRL_ACT_GELU_TANH) {
    constexpr float C2 = 2.0f * 0.7978845608028654f, K = 0.044715f;
    const float e = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(-C2 * (z + K * z * z * z), -88.0f, 80.0f) * LOG2E);
    const float sg = __builtin_amdgcn_rcpf(1.0f + e);
    h = z * sg;
    d = sg + h * (1.0f - sg) * C2 * (1.0f + 3.0f * K * z * z);
  } else if (act == FRL_ACT_TANH) {
    h = tanhf(z);
    d = 1.0f - h * h;
  } else {  // ELU
    const float e = __builtin_amdgcn_exp2f(fminf(z, 0.0f) * LOG2E);
    h = z > 0.0f ? z : e - 1.0f;
    d = z > 0.0f ? 1.0f : e;
  }
}

// One workgroup (128 threads = the 128 features) per STATE: recompute the two layers, sum dL/dcfeat over the state's
// particles, chain rule down to dL/dz1; leaves h1 | dL/dz1 | dL/dz2 in scratch[state][3][128] for the reduce kernel.
__global__ void __launch_bounds__(128)
frl_cond_embed_bwd_state_kernel(int act, const float* __restrict__ obs, int S, int particles, const float* __restrict__ w0,
                                const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                                const float* __restrict__ gcf, int gcf_stride, float* __restrict__ scratch) {
  constexpr int E = COND_EMBED_DIM;
  extern __shared__ float sm[];          // obs[S] | h1[E] | dz2[E]
  float* s_obs = sm;
  float* s_h1 = sm + S;
  float* s_dz2 = s_h1 + E;
  const long long state = blockIdx.x;
  const int t = threadIdx.x;
  for (int k = t; k < S; k += E) s_obs[k] = obs[state * S + k];
  __syncthreads();
  float z1 = b0[t];
#pragma unroll 8
  for (int k = 0; k < S; ++k) z1 = fmaf(s_obs[k], w0[(size_t)k * E + t], z1);
  float h1, a1;
  cact_grad(act, z1, h1, a1);
  s_h1[t] = h1;
  __syncthreads();
  float z2 = b1[t];
#pragma unroll 16
  for (int k = 0; k < E; ++k) z2 = fmaf(s_h1[k], w1[(size_t)k * E + t], z2);
  float h2, a2;
  cact_grad(act, z2, h2, a2);
  float g = 0.0f;
  for (int p = 0; p < particles; ++p) g += gcf[((size_t)state * particles + p) * gcf_stride + t];
  const float dz2 = g * a2;
  s_dz2[t] = dz2;
  __syncthreads();
  float dh1 = 0.0f;
#pragma unroll 16
  for (int o = 0; o < E; ++o) dh1 = fmaf(w1[(size_t)t * E + o], s_dz2[o], dh1);
  float* sc = scratch + (size_t)state * 3 * E;
  sc[t] = h1;
  sc[E + t] = dh1 * a1;
  sc[2 * E + t] = dz2;
}

// dW0 (S, 128), db0, dW1 (128, 128), db1: one thread per element, the states added in index order on four interleaved partial
// sums (a fixed order: bitwise reproducible).  O(n_states (S + 128) 128) — a training batch has a few hundred states.
__global__ void __launch_bounds__(256)
frl_cond_embed_bwd_reduce_kernel(const float* __restrict__ obs, long long n_states, int S, const float* __restrict__ scratch,
                                 float* __restrict__ g_w0, float* __restrict__ g_b0, float* __restrict__ g_w1, float* __restrict__ g_b1) {
  constexpr int E = COND_EMBED_DIM;
  const long long n0 = (long long)S * E, n1 = (long long)E * E, total = n0 + E + n1 + E;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int kind, i = 0, o;          // 0: dW0[i][o]  1: db0[o]  2: dW1[i][o]  3: db1[o]
  long long r = e;
  if (r < n0) { kind = 0; i = (int)(r / E); o = (int)(r % E); }
  else if ((r -= n0) < E) { kind = 1; o = (int)r; }
  else if ((r -= E) < n1) { kind = 2; i = (int)(r / E); o = (int)(r % E); }
  else { kind = 3; o = (int)(r - n1); }
  // state s contributes fa[s] * sc[s][ob] (fa = the observation column or h1 for the kernels, 1 for the biases), added in index
  // order into four interleaved partial sums — 16 states' loads in flight at a time (one load per iteration left this loop at
  // an L2 round trip per state: 90 us for 256 states)
  const int ob = (kind <= 1) ? E + o : 2 * E + o;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  long long s = 0;
  for (; s + 16 <= n_states; s += 16) {
    float va[16], vb[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float* sc = scratch + (size_t)(s + u) * 3 * E;
      vb[u] = sc[ob];
      va[u] = (kind == 0) ? obs[(s + u) * S + i] : (kind == 2) ? sc[i] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] += va[u] * vb[u];
  }
  for (; s < n_states; ++s) {
    const float* sc = scratch + (size_t)s * 3 * E;
    acc[s & 3] += ((kind == 0) ? obs[s * S + i] : (kind == 2) ? sc[i] : 1.0f) * sc[ob];
  }
  const float sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (kind == 0) g_w0[e] = sum;
  else if (kind == 1) g_b0[o] = sum;
  else if (kind == 2) g_w1[(size_t)i * E + o] = sum;
  else g_b1[o] = sum;
}

int launch_cond_embed_bwd(int act, const float* obs, long long n_states, int particles, int in_dim, const float* w0, const float* b0,
                          const float* w1, const float* b1, const float* gcf, int gcf_stride, float* scratch, float* g_w0, float* g_b0,
                          float* g_w1, float* g_b1, hipStream_t stream) {
  if (n_states <= 0) return FRL_OK;
  if (act < FRL_ACT_MISH || act > FRL_ACT_ELU) { set_error("cond_embedding backward: activation %d invalid", act); return FRL_EINVAL; }
  const size_t lds = sizeof(float) * ((size_t)in_dim + 2 * COND_EMBED_DIM);
  if (lds > 60000) { set_error("cond_embedding backward: observation width %d too large", in_dim); return FRL_EUNSUPPORTED; }
  if (n_states > 0x7fffffffLL) { set_error("cond_embedding backward: too many states"); return FRL_EUNSUPPORTED; }
  hipLaunchKernelGGL(frl_cond_embed_bwd_state_kernel, dim3((unsigned)n_states), dim3(COND_EMBED_DIM), lds, stream, act, obs, in_dim, particles,
                     w0, b0, w1, b1, gcf, gcf_stride, scratch);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("cond_embed bwd launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  const long long total = (long long)in_dim * COND_EMBED_DIM + COND_EMBED_DIM + (long long)COND_EMBED_DIM * COND_EMBED_DIM + COND_EMBED_DIM;
  hipLaunchKernelGGL(frl_cond_embed_bwd_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, obs, n_states, in_dim,
                     scratch, g_w0, g_b0, g_w1, g_b1);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("cond_embed bwd reduce launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_state_bias(const float* obs, long long n_states, int in_dim, int H, const float* w, const float* b, float* zs,
                      hipStream_t stream) {
  if (n_states <= 0) return FRL_OK;
  if (H % 128 != 0) { set_error("state bias: width %d is not a multiple of 128", H); return FRL_EUNSUPPORTED; }
  const long long n_wt = (n_states + 31) / 32;
  const int grid = (int)((n_wt + 3) / 4);
  if (n_states <= CE_WIDE_MAX_STATES) hipLaunchKernelGGL(frl_state_bias_wide_kernel, dim3(H / 32, (unsigned)n_wt), dim3(64), 0, stream, obs, n_states, in_dim, H, w, b, zs);
  else hipLaunchKernelGGL(frl_state_bias_kernel, dim3(grid), dim3(256), 0, stream, obs, n_states, in_dim, H, w, b, zs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("state bias launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

int launch_cond_embed(int act, const float* obs, long long n_states, int in_dim, const float* w0, const float* b0,
                      const float* w1, const float* b1, float* cfeat, hipStream_t stream) {
  if (n_states <= 0) return FRL_OK;
  const long long n_wt = (n_states + 31) / 32;
  const int grid = (int)((n_wt + 3) / 4);
  if (n_states <= CE_WIDE_MAX_STATES) hipLaunchKernelGGL(frl_cond_embed_wide_kernel, dim3((unsigned)n_wt), dim3(256), 0, stream, act, obs, n_states, in_dim, w0, b0, w1, b1, cfeat);
  else hipLaunchKernelGGL(frl_cond_embed_kernel, dim3(grid), dim3(256), 0, stream, act, obs, n_states, in_dim, w0, b0, w1, b1, cfeat);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("cond_embed launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

}  // namespace frl
