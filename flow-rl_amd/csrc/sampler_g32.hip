// Stage 2: the persistent fused T-step sampler, geometry "G32" (hidden width 256, exact-f32 MFMA).
//
// What one launch does (reference: DDPM.sample flowrl/flow/ddpm.py:184-221, ContinuousDDPM.sample
// continuous_ddpm.py:149-222, CNF.sample cnf.py:122-141; network ddpm.py:83-95 + mlp.py:25-33):
//
//   for each tile of 32*nw chains (persistent workgroups, nw waves, one wave per SIMD):
//       x <- xT ; cond operands <- cond rows (broadcast over particles)
//       for j in 0..T-1:   (all on chip)
//           P  = tb[j] + W0x^T x + W0c^T cond                    (Dense_0; time slice pre-folded into tb)
//           P  = b_l + W_l^T act(P)            l = 1..L-1        (hidden Dense layers)
//           e  = b_L + W_L^T act(P)                               (Dense_L, zero-padded to 32 rows)
//           x <- update(x, e, noise[j])                           (DDPM / DDIM / Euler epilogue)
//       x0 <- x
//
// Layout.  Everything is computed TRANSPOSED: an MFMA tile is [32 output features] x [32 chains], so
// a chain lives on a lane (column = lane & 31) and v_mfma_f32_32x32x2_f32 leaves feature
// row(r, h) = (r & 3) + 8 (r >> 2) + 4 h of the tile in accumulator register r of lane half
// h = lane >> 5.  The B operand of the same instruction wants k = h on lane half h, so accumulator
// register r of a finished layer IS the B operand of the next layer's k-step that contracts features
// {row(r,0), row(r,1)}: activations never leave the register file, there is no LDS bounce and no lane
// permute between layers.  Only weights move: each layer's kernel streams through LDS as 32-row
// K-slabs (32 KB) by LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per slab, and is
// read back as the A operand with ds_read_b128 (4 output blocks per read).  Output feature
// f = 128 g + 4 i + j sits in block (g, j), tile row i; the same map indexes the next layer's K rows,
// so hidden-layer slab (g, j) is the row gather {128 g + 4 q + j}, done by the DMA's per-lane source
// address.  Dense kernels are consumed in place in their Flax (in, out) row-major layout.
#include <utility>

#include "frl_internal.h"
#include "frl_rng.h"

namespace frl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FRL_ROWR(r) (((r) & 3) + 8 * ((r) >> 2))

// compile-time loop: register-array indices stay constants whatever the unroller decides
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// ---- activations (flowrl/functional/activation.py:8-18; flax gelu tanh form) -------------------
// exp via v_exp_f32 and reciprocal via v_rcp_f32 (1 ulp each); every form below is free of
// cancellation so the relative error stays at a few ulp over the whole range.
template <int ACT>
__device__ __forceinline__ float act_fn(float x) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (ACT == FRL_ACT_MISH) {
    // x * tanh(softplus(x)) == x * n / (n + 2),  n = e^x (e^x + 2)
    const float e = __builtin_amdgcn_exp2f(fminf(x, 20.0f) * LOG2E);
    const float n = e * (e + 2.0f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
  } else if constexpr (ACT == FRL_ACT_SILU) {
    const float e = __builtin_amdgcn_exp2f(fminf(-x, 80.0f) * LOG2E);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
  } else if constexpr (ACT == FRL_ACT_RELU) {
    return fmaxf(x, 0.0f);
  } else if constexpr (ACT == FRL_ACT_GELU_TANH) {
    // 0.5 x (1 + tanh(u)) == x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)
    const float u = 0.7978845608028654f * (x + 0.044715f * (x * x * x));
    const float e = __builtin_amdgcn_exp2f(fminf(-2.0f * u, 80.0f) * LOG2E);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
  } else if constexpr (ACT == FRL_ACT_TANH) {
    const float t = __builtin_amdgcn_exp2f(-2.0f * fabsf(x) * LOG2E);
    const float y = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
    return copysignf(y, x);
  } else {  // ELU
    const float e = __builtin_amdgcn_exp2f(fminf(x, 0.0f) * LOG2E);
    return x > 0.0f ? x : e - 1.0f;
  }
}

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  // 64 lanes x 16 B: per-lane global source, LDS destination = wave-uniform base + lane * 16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ const float* pick_ptr(const float* const (&arr)[FRL_MAX_HIDDEN - 1], int idx) {
  // runtime index into a kernel-argument array without spilling it to scratch
  const float* p = arr[0];
#pragma unroll
  for (int k = 1; k < FRL_MAX_HIDDEN - 1; ++k) p = (idx == k) ? arr[k] : p;
  return p;
}

// One hidden Dense layer: 8 K-slabs x 16 k-steps x 8 output blocks, software-pipelined by hand: the
// A operands (2 x ds_read_b128) and the lazily activated B operand of k-step n+1 are fetched before
// the 8 MFMAs of k-step n are issued, also across slab boundaries (the barrier + next DMA sit between
// the last operand fetch of a slab and its last 8 MFMAs, whose operands are already in registers).
template <int ACT, typename NextSlab>
__device__ __forceinline__ void hidden_layer(f32x16 (&Q)[8], const f32x16 (&P)[8], NextSlab& next_slab, unsigned lane_h) {
  const char* sl = next_slab() + lane_h;
  f32x4 a0 = *(const f32x4*)(sl);
  f32x4 a1 = *(const f32x4*)(sl + 512);
  float bop = act_fn<ACT>(P[0][0]);
  static_for<128>([&](auto IDX) {
    constexpr int idx = decltype(IDX)::value;
    f32x4 n0 = a0, n1 = a1;
    float bn = bop;
    if constexpr (idx + 1 < 128) {
      constexpr int nb = (idx + 1) >> 4, nr = (idx + 1) & 15;
      if constexpr (nr == 0) sl = next_slab() + lane_h;
      n0 = *(const f32x4*)(sl + FRL_ROWR(nr) * 1024);
      n1 = *(const f32x4*)(sl + FRL_ROWR(nr) * 1024 + 512);
      bn = act_fn<ACT>(P[nb][nr]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Q[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bop, Q[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) Q[4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bop, Q[4 + j], 0, 0, 0);
    // interleave: operand fetch + activation of k-step n+1 ride in the shadow of k-step n's MFMAs
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 x ds_read (A operands of k-step n+1)
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // VALU (activation of k-step n+1)
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    a0 = n0; a1 = n1; bop = bn;
  });
}

// accumulator init = bias of feature f = 128 g + 4 row(r,h) + j  (block (g,j), register r)
__device__ __forceinline__ void init_bias(f32x16 (&Q)[8], const float* s_bias, int lh) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const f32x4 bv = *(const f32x4*)(s_bias + g * 128 + 4 * (FRL_ROWR(r) + 4 * lh));
#pragma unroll
      for (int j = 0; j < 4; ++j) Q[g * 4 + j][r] = bv[j];
    }
  }
}

template <int ACT, int ARMAX, int CKMAX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_g32_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = G32_H;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int li = lane & 31, lh = lane >> 5;

  float* s_tb = (float*)(smem + NBUF * SLAB_BYTES);   // [T][H]  layer-0 bias table (time slice folded in)
  float* s_bh = s_tb + a.T * H;                       // [L-1][H] hidden biases
  float* s_bl = s_bh + (a.L - 1) * H;                 // [32]    padded last-layer bias
  float* s_coef = s_bl + LAST_PAD;                    // [T][8]

  for (int idx = tid; idx < a.T * H; idx += blockDim.x) s_tb[idx] = a.ws_tb[idx];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += blockDim.x) s_bh[(l - 1) * H + idx] = bp[idx];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.T * COEF_STRIDE; idx += blockDim.x) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  // ---- weight-slab stream: the same NS slabs every step of every tile -----------------------------
  const int NS = a.NS;
  auto issue = [&](int seq, unsigned bufoff) {
    const char* base; int first, stride, nrows;
    if (seq < a.n_l0) {                       // Dense_0 packed image
      base = (const char*)a.ws_w0p;
      stride = 1;
      if (a.merged) { first = 0; nrows = a.XR + 2 * a.CK; }
      else if (seq == 0) { first = 0; nrows = a.XR; }
      else { first = a.XR + (seq - 1) * SLAB_ROWS; nrows = SLAB_ROWS; }
    } else if (seq < NS - 1) {                // hidden layer l, input block b = (g, j): rows 128g + 4q + j
      const int hs = seq - a.n_l0;
      const int l = hs >> 3, b = hs & 7;
      base = (const char*)pick_ptr(a.wh, l);
      first = 128 * (b >> 2) + (b & 3); stride = 4; nrows = SLAB_ROWS;
    } else {                                  // Dense_L padded image, 256 x 32 f32 = 32 KB contiguous
      base = (const char*)a.ws_wlp; first = 0; stride = 1; nrows = SLAB_ROWS;
    }
    for (int q = wave; q < nrows; q += nw)
      dma16(base + (size_t)(first + q * stride) * 1024 + lane * 16, smem + bufoff + q * 1024);
  };

  int seq_issue = 0;
  unsigned boff_issue = 0, boff_cur = 0;
  issue(0, 0);
  seq_issue = (NS > 1) ? 1 : 0;
  boff_issue = SLAB_BYTES;

  // start of a slab's compute: wait for its DMA (and, through the barrier, for every wave to be done
  // with the previous slab, whose buffer the next DMA overwrites), then prefetch the next slab.
  auto next_slab = [&]() -> const char* {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(seq_issue, boff_issue);
    seq_issue = (seq_issue + 1 == NS) ? 0 : seq_issue + 1;
    boff_issue = (boff_issue + SLAB_BYTES == NBUF * SLAB_BYTES) ? 0u : boff_issue + SLAB_BYTES;
    const char* p = smem + boff_cur;
    boff_cur = (boff_cur + SLAB_BYTES == NBUF * SLAB_BYTES) ? 0u : boff_cur + SLAB_BYTES;
    return p;
  };

  const unsigned lane_h = lh * 4096 + li * 16;   // A-operand read base for rows row(r,h): + ROWR(r)*1024
  const unsigned lane_c = lh * 1024 + li * 16;   // ... for consecutive row pairs (2s, 2s+1)
  const int tile_chains = G32_CHAINS * nw;

  f32x16 P[8], Q[8];

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const long long chain = (long long)tile * tile_chains + wave * G32_CHAINS + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    float xs[ARMAX];      // chain state x^T: register r holds action row a = row(r, h)
#pragma unroll
    for (int r = 0; r < ARMAX; ++r) {
      const int ai = FRL_ROWR(r) + 4 * lh;
      xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f;
    }
    float cb[CKMAX];      // cond operands: k-step s contracts cond features (2s, 2s+1)
#pragma unroll
    for (int s = 0; s < CKMAX; ++s) {
      const int k = 2 * s + lh;
      cb[s] = (valid && k < a.C) ? a.cond[state * a.C + k] : 0.0f;
    }

    for (int j = 0; j < a.T; ++j) {
      // noise for this step: issued now, consumed after the last layer
      float zr[ARMAX];
#pragma unroll
      for (int r = 0; r < ARMAX; ++r) {
        const int ai = FRL_ROWR(r) + 4 * lh;
        zr[r] = (a.noise != nullptr && valid && ai < a.A) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
      }

      // ---------------- Dense_0 : P = tb[j] + W0x^T x + W0c^T cond ----------------------------------
      init_bias(P, s_tb + j * H, lh);
      {
        const char* sl = next_slab();
#pragma unroll
        for (int r = 0; r < ARMAX; ++r) {
          if (r < a.AR) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const f32x4 aw = *(const f32x4*)(sl + lane_h + FRL_ROWR(r) * 1024 + g * 512);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
                P[g * 4 + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[jj], xs[r], P[g * 4 + jj], 0, 0, 0);
            }
          }
        }
        int srow = a.merged ? a.XR : 0;   // first cond row inside the current slab
        if (!a.merged) { sl = next_slab(); }
#pragma unroll
        for (int s = 0; s < CKMAX; ++s) {
          if (s < a.CK) {
            if (!a.merged && s > 0 && (s & 15) == 0) { sl = next_slab(); srow = 0; }
            const char* rp = sl + lane_c + (unsigned)(srow + 2 * (s & 15)) * 1024;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const f32x4 aw = *(const f32x4*)(rp + g * 512);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj)
                P[g * 4 + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[jj], cb[s], P[g * 4 + jj], 0, 0, 0);
            }
          }
        }
      }

      // ---------------- hidden Dense_l : Q = b_l + W_l^T act(P) ; P <- Q ----------------------------
      for (int l = 1; l < a.L; ++l) {
        init_bias(Q, s_bh + (l - 1) * H, lh);
        hidden_layer<ACT>(Q, P, next_slab, lane_h);
#pragma unroll
        for (int b = 0; b < 8; ++b) P[b] = Q[b];
      }

      // ---------------- Dense_L : O = b_L + W_L^T act(P), padded to 32 rows -------------------------
      f32x16 O;
#pragma unroll
      for (int r = 0; r < 16; ++r) O[r] = s_bl[FRL_ROWR(r) + 4 * lh];
      {
        const char* sl = next_slab();
        // image [k][32] f32; k-step (b, r) contracts k = 128 g + 4 row(r,h) + j
        const char* wl = sl + lh * 2048 + li * 4;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float aw = *(const float*)(wl + (128 * (b >> 2) + (b & 3)) * 128 + FRL_ROWR(r) * 512);
            O = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, act_fn<ACT>(P[b][r]), O, 0, 0, 0);
          }
        }
      }

      // ---------------- epilogue: one reverse-diffusion / Euler step ---------------------------------
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
#pragma unroll
      for (int r = 0; r < ARMAX; ++r) {
        const int ai = FRL_ROWR(r) + 4 * lh;
        const bool in = valid && ai < a.A;
        const float eps = O[r], xt = xs[r];
        if (a.xt_hist != nullptr && in) {
          const long long o = ((long long)j * a.B + chain) * a.A + ai;
          a.xt_hist[o] = xt;
          a.eps_hist[o] = eps;
        }
        float z = zr[r];
        if (a.rng_impl != FRL_RNG_NONE && in && c5 != 0.0f) {   // in-kernel jax.random.normal(key_j, xt.shape)[chain, ai]; the
                                                                   // last iteration multiplies its draw by 0 (ddpm.py:206): skipped
          z = rng_step_normal(a.rng_keys, a.rng_impl, j, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai);
          __builtin_amdgcn_sched_barrier(0);       // one draw at a time: interleaved draws cost registers the hot loop needs
        }
        float xn;
        if (a.mode == MODE_DDPM_DDPM) {
          float x0h = c0 * (xt - c1 * eps);                       // ddpm.py:199
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);   // ddpm.py:200-201
          xn = c2 * (c3 * x0h + c4 * xt);                           // ddpm.py:202-203
          xn += c5 * z;                                         // ddpm.py:206
        } else if (a.mode == MODE_DDPM_DDIM) {
          xn = c0 * (xt - c1 * eps) + c2 * eps;                     // ddpm.py:211
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        } else if (a.mode == MODE_CDDPM_DDPM) {
          float x0h = (xt - c1 * eps) / c0;                         // continuous_ddpm.py:165
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * xt;                                  // :173
          xn += c5 * z;                                         // :174
        } else if (a.mode == MODE_CDDPM_DDIM) {
          float x0h = (xt - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;                                 // :177
        } else {
          xn = xt + eps / c0;                                       // cnf.py:108
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        }
        xs[r] = in ? xn : 0.0f;
      }
    }

#pragma unroll
    for (int r = 0; r < ARMAX; ++r) {
      const int ai = FRL_ROWR(r) + 4 * lh;
      if (valid && ai < a.A) a.x0[chain * a.A + ai] = xs[r];
    }
  }
  // the stream runs one slab ahead: drain the last prefetch before the wave ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- MFMA lane-map self test (tests only) ---------------------------------------------------------
__global__ void __launch_bounds__(64) frl_selftest_mfma32_kernel(const float* A, const float* B, float* D, int K) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k + lh], B[(k + lh) * 32 + li], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) D[(FRL_ROWR(r) + 4 * lh) * 32 + li] = acc[r];
}

int launch_selftest_mfma32(const float* A, const float* B, float* D, int K, hipStream_t stream) {
  hipLaunchKernelGGL(frl_selftest_mfma32_kernel, dim3(1), dim3(64), 0, stream, A, B, D, K);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("selftest launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

// ---- launcher --------------------------------------------------------------------------------------
template <int ACT, int ARMAX, int CKMAX>
static int launch_inst(const KArgs& a, int grid, int block, size_t lds, hipStream_t stream) {
  auto kern = frl_sampler_g32_kernel<ACT, ARMAX, CKMAX>;
  static thread_local size_t configured[16] = {0};
  int dev = 0;
  hipGetDevice(&dev);
  if (dev < 16 && configured[dev] < lds) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
    configured[dev] = lds;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

template <int ACT>
static int launch_act(const KArgs& a, int grid, int block, size_t lds, hipStream_t stream) {
  if (a.AR <= 4 && a.CK <= 12) return launch_inst<ACT, 4, 12>(a, grid, block, lds, stream);
  set_error("G32 kernel: x_dim %d / cond_dim %d outside the compiled operand budgets (A <= 8, C <= 24)", a.A, a.C);
  return FRL_EUNSUPPORTED;
}

int launch_sampler_g32(const FrlDesc& d, const KArgs& a0, hipStream_t stream) {
  KArgs a = a0;
  const int ncu = device_cus();
  // waves per workgroup: 4 (one per SIMD) when there is enough work to fill the chip, fewer for small
  // batches so that more CUs get a tile
  long long per_cu = (a.B + (long long)ncu * G32_CHAINS - 1) / ((long long)ncu * G32_CHAINS);
  int nw = (int)(per_cu < 1 ? 1 : (per_cu > MAX_WAVES ? MAX_WAVES : per_cu));
  if (nw == 3) nw = 4;
  const int tile_chains = G32_CHAINS * nw;
  a.ntiles = (int)((a.B + tile_chains - 1) / tile_chains);
  const int grid = a.ntiles < ncu ? a.ntiles : ncu;
  const size_t lds = (size_t)NBUF * SLAB_BYTES +
                     sizeof(float) * ((size_t)a.T * G32_H + (size_t)(a.L - 1) * G32_H + LAST_PAD + (size_t)a.T * COEF_STRIDE);
  if (lds > 160 * 1024) { set_error("G32 kernel: steps=%d needs %zu B of LDS (> 160 KiB)", a.T, lds); return FRL_EUNSUPPORTED; }
  switch (d.act) {
    case FRL_ACT_MISH: return launch_act<FRL_ACT_MISH>(a, grid, 64 * nw, lds, stream);
    case FRL_ACT_SILU: return launch_act<FRL_ACT_SILU>(a, grid, 64 * nw, lds, stream);
    case FRL_ACT_RELU: return launch_act<FRL_ACT_RELU>(a, grid, 64 * nw, lds, stream);
    default: break;
  }
  set_error("G32 kernel: activation %d not compiled in", d.act);
  return FRL_EUNSUPPORTED;
}

}  // namespace frl
