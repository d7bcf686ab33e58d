// Chain VJP of the fused sampler, bf16-MFMA variant (FrlDesc.dtype = FRL_DTYPE_BF16) — BASELINE config 5:
// "DQL backprop-through-chain VJP (T=5, batch 256k) ... bf16 MFMA variant".  Reference semantics: jax.grad through
// the T unrolled steps of DDPM.sample (flowrl/agent/offline/dql.py:93-123, flowrl/flow/ddpm.py:186-219,
// flowrl/module/model.py:78-86).  Arithmetic: every Dense contraction (forward recompute, dL/dh = W dL/dz, weight
// gradients) takes bf16 operands and accumulates in f32 (v_mfma_f32_32x32x16_bf16); biases, the hoisted time branch,
// activations and their derivatives, the chain state and the VJP of the update rule stay f32.
//
//   frl_prepare_bwd_bf16_kernel   backward half of the weight stream image (same 1 KB A-operand pieces / 32 KB chunks
//                                 as the forward image of sampler_bf16.hip): W_L^T, the transposed hidden kernels
//                                 top-down, Dense_0's x rows
//   frl_bwd_sweep_bf16_kernel     8 waves x 32 chains per workgroup, steps j = T-1 .. 0: recompute the forward of step
//                                 j from the stashed x_j with the forward kernel's output-block-major layers (h_l and
//                                 act'(z_l) packed to bf16 and stashed), VJP of the update -> dL/deps, then the same
//                                 layers downwards over the transposed stream: dL/dz_l = dL/dh_l * act'(z_l), packed
//                                 = the next B operand, stashed for the weight gradients.  One LDS-DMA ring feeds both
//                                 halves: 2 * bf_ns chunks per step.
//   frl_bwd_dw_bf16_kernel        dW_l = sum over chains and steps of X_l (x) dL/dz_l.  The contraction index (chains)
//                                 sits on the LANES of the stash pieces, an MFMA wants it inside a lane: pieces are
//                                 transposed through LDS ([feature][32 chains] bf16), each workgroup owns the whole
//                                 256 x 256 tile of a layer in accumulators over its range of tile-steps, one
//                                 float-atomic flush per layer.  HBM-bound (it streams the stash once).
//
// Three builds of the sweep (FRL_BWD16_PHASE, one translation unit each, as csrc/sampler_bwd.hip):
//   0  recompute + backward from xt_hist alone (the forward was a plain frl_sampler_fwd with history)
//   1  TRAINING FORWARD (frl_sampler_fwd_train, dtype bf16): the sampler itself, steps 0 .. T-1 over the forward image only,
//      writing x0 / history and the stash of h = act(z) and act'(z) the backward reads
//   2  backward from that stash: steps T-1 .. 0 over the backward image only, eps from eps_hist — no forward MFMAs, no
//      activation VALU
#include "frl_bf16.h"
#include "frl_rng.h"

#ifndef FRL_BWD16_PHASE
#error "compile with -DFRL_BWD16_PHASE=0|1|2"
#endif

namespace frl {

#if FRL_BWD16_PHASE == 0
// ---- backward weight stream image ---------------------------------------------------------------------------------
struct PBbArgs {
  const float* kernel[FRL_MAX_HIDDEN + 1];
  unsigned short* img;
  int A, L, ns;
};

__global__ void __launch_bounds__(256) frl_prepare_bwd_bf16_kernel(const PBbArgs p) {
  constexpr int H = 256;
  const long long total = (long long)p.ns * (BF_CHUNK_BYTES / 2);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / (BF_CHUNK_BYTES / 2));
    const int within = (int)(idx - (long long)c * (BF_CHUNK_BYTES / 2));
    const int piece = within >> 9, lane = (within >> 3) & 63, e = within & 7;
    const int i = lane & 31, s = lane >> 5;
    float v = 0.0f;
    if (c == 0) {                                       // dL/dh_{L-1} = W_L dL/deps: piece = output block, one k-step
      const int act_row = BF_ROWR(e) + 4 * s;           //   (k = action rows in the register order of the chain state)
      if (piece < 8 && act_row < p.A) v = p.kernel[p.L][(size_t)(32 * piece + i) * p.A + act_row];
    } else if (c < 1 + 4 * (p.L - 1)) {                 // dL/dh_{l-1} = W_l dL/dz_l, l = L-1 .. 1
      const int hc = c - 1, l = (p.L - 1) - (hc >> 2);
      const int ob = 2 * (hc & 3) + (piece >> 4), ks = piece & 15;
      const int featk = 32 * (ks >> 1) + BF_ROWR(8 * (ks & 1) + e) + 4 * s;
      v = p.kernel[l][(size_t)(32 * ob + i) * H + featk];
    } else if (piece < 16) {                            // dL/dx = W_0[x rows] dL/dz_0: one output block
      const int featk = 32 * (piece >> 1) + BF_ROWR(8 * (piece & 1) + e) + 4 * s;
      if (i < p.A) v = p.kernel[0][(size_t)i * H + featk];
    }
    p.img[idx] = f2bf(v);
  }
}

int launch_prepare_bwd_bf16(const FrlParams& prm, const BwdPlan16& bp, float* ws, hipStream_t stream) {
  PBbArgs p;
  for (int l = 0; l <= bp.f.L; ++l) p.kernel[l] = prm.kernel[l];
  p.img = (unsigned short*)(ws + bp.off_img_b);
  p.A = bp.f.A; p.L = bp.f.L; p.ns = bp.f.bf_ns;
  hipLaunchKernelGGL(frl_prepare_bwd_bf16_kernel, dim3(64), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 bwd prepare launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

#endif  // FRL_BWD16_PHASE == 0

// ---- activation and derivative (h bit-identical to the forward kernel's act_bf for the same z) ---------------------
template <int ACT>
__device__ __forceinline__ void act_grad_bf(float x, float& h, float& d) {
  constexpr float LOG2E = 1.4426950408889634f;
  if constexpr (ACT == FRL_ACT_MISH) {
    // h = x n / (n + 2) = x - 2 x r,  n = e (e + 2),  r = 1 / (n + 2);   h' = n r + 4 x e (e + 1) r^2
    // (the exponent is capped at 20: beyond it h = x, h' = 1 to f32 precision, and e^4 stays finite)
    const float e = __builtin_amdgcn_exp2f(__builtin_fminf(x, 20.0f) * LOG2E);
    const float dd = __builtin_fmaf(e, e + 2.0f, 2.0f);
    const float r = __builtin_amdgcn_rcpf(dd);
    const float m = x * r;
    h = __builtin_fmaf(-2.0f, m, x);
    const float nr = __builtin_fmaf(-2.0f, r, 1.0f);
    const float q = __builtin_fmaf(e, e, e);
    d = __builtin_fmaf(4.0f * q, m * r, nr);
  } else if constexpr (ACT == FRL_ACT_SILU) {
    const float e = __builtin_amdgcn_exp2f(x * -LOG2E);
    const float s = __builtin_amdgcn_rcpf(e + 1.0f);
    h = x * s;
    d = __builtin_fmaf(h, 1.0f - s, s);
  } else {  // RELU
    asm("v_max_f32 %0, 0, %1" : "=v"(h) : "v"(x));
    d = x > 0.0f ? 1.0f : 0.0f;
  }
}

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// Stash stores.  HAZARD (measured on gfx950): a buffer_store_dwordx4 with an SGPR soffset may still be reading its
// data registers when the next VALU instructions overwrite them — the last four lanes of each row of 16 then store
// the NEW register contents.  The compiler's hazard recognizer adds the wait states for > 64-bit stores to every
// global_store, but to a buffer_store only when soffset is NOT a register (its model exempts that form; this chip
// does not).  The stash stores are therefore global stores: uniform base + lane * 16 + constant.
__device__ __forceinline__ void st16(u32x4 v, char* base, unsigned lane16, int off) {
  *(u32x4*)(base + off + lane16) = v;
}
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}

// One Dense-shaped contraction, output-block-major: NOB output blocks of NK k-steps (pieces streamed from the LDS ring
// through a 4-deep register ring, a chunk boundary every 32 pieces).  init(ob, acc) sets the accumulator (and may
// issue loads the epilogue needs), epi(ob, acc) consumes a finished block; it is issued alongside the next block's
// MFMAs.
template <int NK, int NOB, typename BFun, typename Init, typename Epi, typename NextChunk>
__device__ __forceinline__ void layer_g(BFun&& bfun, Init&& init, Epi&& epi, NextChunk& next_chunk, unsigned lane16) {
  constexpr int NP = NOB * NK, R = 4;
  static_assert(NP >= R, "register ring deeper than the layer");
  f32x16 acc[2];
  u32x4 ring[R];
  const char* sl = next_chunk() + lane16;
  const char* sl_next = sl;
  sfor<R>([&](auto PI) { constexpr int p = decltype(PI)::value; ring[p] = *(const u32x4*)(sl + p * 1024); });
  sfor<NOB>([&](auto OI) {
    constexpr int ob = decltype(OI)::value;
    init(OI, acc[ob & 1]);
    if constexpr (ob > 0) epi(std::integral_constant<int, ob - 1>{}, acc[(ob - 1) & 1]);
    sfor<NK>([&](auto KI) {
      constexpr int ks = decltype(KI)::value;
      constexpr int p = ob * NK + ks;
      acc[ob & 1] = mfma_bf(ring[p % R], bfun(KI), acc[ob & 1]);
      if constexpr (p + R < NP) {
        constexpr int q = p + R;
        if constexpr ((q & 31) == 0) sl_next = next_chunk() + lane16;   // the ring runs R pieces ahead of the MFMAs
        ring[p % R] = *(const u32x4*)(sl_next + (q & 31) * 1024);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
  epi(std::integral_constant<int, NOB - 1>{}, acc[(NOB - 1) & 1]);
}

template <int ACT, int PHASE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
frl_bwd_sweep_bf16_kernel(const BArgs b) {
  constexpr int H = 256, NRX = 8, CKS = 2, K0 = 1 + CKS, NT = 512;
  const KArgs& a = b.k;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, ls = lane >> 5;

  // tables exactly as in frl_sampler_bf16_kernel: biases permuted to the accumulator order of a lane half
  float* s_tb = (float*)(smem + BF_NBUF * BF_CHUNK_BYTES);
  float* s_bh = s_tb + a.T * H;
  float* s_bl = s_bh + (a.L - 1) * H;
  float* s_coef = s_bl + LAST_PAD;
  auto perm_src = [](int p) { const int s = p >> 7, ob = (p >> 4) & 7, r = p & 15; return 32 * ob + BF_ROWR(r) + 4 * s; };
  for (int idx = tid; idx < a.T * H; idx += NT) s_tb[idx] = a.ws_tb[(idx & ~(H - 1)) + perm_src(idx & (H - 1))];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr_rs(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += NT) s_bh[(l - 1) * H + idx] = bp[perm_src(idx)];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.T * COEF_STRIDE; idx += NT) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  // ---- weight stream: forward image then backward image (phase 0: 2 * bf_ns chunks per step; phase 1 the forward half,
  //      phase 2 the backward half only), three chunks in flight -----
  const int NS = (PHASE == 0 ? 2 : 1) * a.bf_ns;
  const char* img = (const char*)a.ws_bf16 + (PHASE == 2 ? (size_t)a.bf_ns * BF_CHUNK_BYTES : 0) + wave * 4096 + lane * 16;
  auto issue = [&](int seq, unsigned boff) {
    const char* src = img + (size_t)seq * BF_CHUNK_BYTES;
    char* dst = smem + boff + wave * 4096;
    sfor<4>([&](auto II) { constexpr int i = decltype(II)::value; dma16_bf<i * 1024>(src, dst); });
  };
  issue(0, 0);
  issue(1 % NS, BF_CHUNK_BYTES);
  issue(2 % NS, 2 * BF_CHUNK_BYTES);
  int seq_issue = 3 % NS;                                 // NS >= 2 (L = 1, one phase: 1 + 0 + 1 chunks)
  unsigned boff_issue = 3 * BF_CHUNK_BYTES, boff_cur = 0;
  // vmcnt(8): at most the newest 8 vector-memory operations are still in flight.  Loads complete in issue order among
  // themselves, and the two newer chunks are 8 DMA loads of this wave, so the chunk about to be read has landed whatever the
  // stash stores do (stores are NOT ordered against loads on this chip: counting them into a larger vmcnt — tried in round 2 —
  // lets a wave pass with its own DMA still in flight; the relu VJP case caught it).  Stash traffic issued since only makes
  // the wait stricter than needed.
  auto next_chunk = [&]() -> const char* {
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(seq_issue, boff_issue);
    seq_issue = (seq_issue + 1 == NS) ? 0 : seq_issue + 1;
    boff_issue = (boff_issue + BF_CHUNK_BYTES) & (BF_NBUF * BF_CHUNK_BYTES - 1);
    const char* p = smem + boff_cur;
    boff_cur = (boff_cur + BF_CHUNK_BYTES) & (BF_NBUF * BF_CHUNK_BYTES - 1);
    return p;
  };

  const unsigned lane16 = lane * 16;
  const int ntiles = (int)((a.B + 255) / 256);
  u32x4 bin[16], bout[16];

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long wt = (long long)tile * 8 + wave;
    const long long chain = wt * 32 + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    u32x4 cbp[CKS];     // cond operands, packed once per tile (the stash backward never contracts them)
    if constexpr (PHASE != 2) sfor<CKS>([&](auto CI) {
      constexpr int kc = decltype(CI)::value;
      float cv[8];
      sfor<8>([&](auto EI) {
        constexpr int e = decltype(EI)::value;
        const int c = 16 * kc + 8 * ls + e;
        cv[e] = (valid && c < a.C) ? a.cond[state * a.C + c] : 0.0f;
      });
      sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; cbp[kc][w] = pack2(cv[2 * w], cv[2 * w + 1]); });
    });
    float gx[NRX];      // dL/dx_{j+1}: register r <-> action row ROWR(r) + 4 s, like the chain state
    float xs[NRX];      // phase 1: the chain state; phases 0 / 2: x_j from the history
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = BF_ROWR(r) + 4 * ls;
      if constexpr (PHASE == 1) { xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f; gx[r] = 0.0f; }
      else gx[r] = (valid && ai < a.A) ? b.g_x0[chain * a.A + ai] : 0.0f;
    });

    for (int jj = 0; jj < a.T; ++jj) {
      const int j = (PHASE == 1) ? jj : a.T - 1 - jj;
      const long long ts = wt * a.T + j;
      const size_t srow = (size_t)ts * a.L * 4096;          // floats: 16 KB per layer
      const __amdgpu_buffer_rsrc_t rs_ad = __builtin_amdgcn_make_buffer_rsrc((void*)(b.hd + srow), 0, a.L * 16384, 0x00020000);   // the stashed z

      if constexpr (PHASE != 1) {
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = BF_ROWR(r) + 4 * ls;
          xs[r] = (valid && ai < a.A) ? b.xt_hist[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
      }

      // epilogue of a forward block: z packed to bf16 and stashed ONCE (the consumers recompute act(z) / act'(z) from it:
      // half the stash bytes of keeping h and act'); h = act(z) from the f32 z is the next B operand
      auto epi_fwd = [&](u32x4 (&out)[16], int lsoff) {
        char* pz = (char*)(b.hd + srow) + lsoff;
        return [&out, pz, lane16](auto OI, const f32x16& acc) {
          constexpr int ob = decltype(OI)::value;
          sfor<2>([&](auto QI) {
            constexpr int q = decltype(QI)::value;
            u32x4 hp, zp;
            sfor<4>([&](auto WI) {
              constexpr int w = decltype(WI)::value;
              float h0, d0, h1, d1;
              act_grad_bf<ACT>(acc[8 * q + 2 * w], h0, d0);
              act_grad_bf<ACT>(acc[8 * q + 2 * w + 1], h1, d1);
              hp[w] = pack2(h0, h1);
              zp[w] = pack2(acc[8 * q + 2 * w], acc[8 * q + 2 * w + 1]);
            });
            out[2 * ob + q] = hp;
            st16(zp, pz, lane16, (2 * ob + q) * 1024);
          });
        };
      };
      auto init_bias = [&](const float* sb) {
        return [sb, ls](auto OI, f32x16& acc) {
          constexpr int ob = decltype(OI)::value;
          sfor<4>([&](auto VI) {
            constexpr int v = decltype(VI)::value;
            const f32x4 bv = *(const f32x4*)(sb + (ls * 8 + ob) * 16 + 4 * v);
            sfor<4>([&](auto JI) { constexpr int jx = decltype(JI)::value; acc[4 * v + jx] = bv[jx]; });
          });
        };
      };

      // ================= forward of step j (phase 0: recomputed; phase 1: THE forward; phase 2: read back) ============
      f32x16 O;
      if constexpr (PHASE == 2) {
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = BF_ROWR(r) + 4 * ls;
          O[r] = (valid && ai < a.A) ? b.eps_hist[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
      } else {
      {
        u32x4 bx;
        sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; bx[w] = pack2(xs[2 * w], xs[2 * w + 1]); });
        auto b0 = [&](auto KI) -> u32x4 {
          constexpr int ks = decltype(KI)::value;
          if constexpr (ks == 0) return bx; else return cbp[ks - 1];
        };
        layer_g<K0, 8>(b0, init_bias(s_tb + j * H), epi_fwd(bin, 0), next_chunk, lane16);
      }
      for (int l = 1; l < a.L; ++l) {
        auto bh = [&](auto KI) -> u32x4 { return bin[decltype(KI)::value]; };
        layer_g<16, 8>(bh, init_bias(s_bh + (l - 1) * H), epi_fwd(bout, l * 16384), next_chunk, lane16);
        sfor<16>([&](auto KI) { constexpr int k = decltype(KI)::value; bin[k] = bout[k]; });
      }
      {
        auto bh = [&](auto KI) -> u32x4 { return bin[decltype(KI)::value]; };
        auto init_o = [&](auto, f32x16& acc) {
          sfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; acc[r] = s_bl[BF_ROWR(r) + 4 * ls]; });
        };
        auto epi_o = [&](auto, const f32x16& acc) { O = acc; };
        layer_g<16, 1>(bh, init_o, epi_o, next_chunk, lane16);
      }
      }   // PHASE != 2

      if constexpr (PHASE == 1) {
        // ================= the sampler's update (the epilogue of csrc/sampler_bf16.hip, f32) ==========================
        const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
        const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
        const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = BF_ROWR(r) + 4 * ls;
          const bool in = valid && ai < a.A;
          const float eps = O[r], xt = xs[r];
          float z = (a.noise != nullptr && in) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
          if (a.rng_impl != FRL_RNG_NONE && in && c5 != 0.0f) {
            z = rng_step_normal(a.rng_keys, a.rng_impl, j, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (in) {
            const long long o = ((long long)j * a.B + chain) * a.A + ai;
            a.xt_hist[o] = xt;
            a.eps_hist[o] = eps;
          }
          float xn;
          if (a.mode == MODE_DDPM_DDPM) {
            float x0h = c0 * (xt - c1 * eps);                       // ddpm.py:199
            if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);   // ddpm.py:200-201
            xn = c2 * (c3 * x0h + c4 * xt);                           // ddpm.py:202-203
            xn += c5 * z;                                             // ddpm.py:206
          } else if (a.mode == MODE_DDPM_DDIM) {
            xn = c0 * (xt - c1 * eps) + c2 * eps;                     // ddpm.py:211
            if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
          } else if (a.mode == MODE_CDDPM_DDPM) {
            float x0h = (xt - c1 * eps) / c0;                         // continuous_ddpm.py:165
            if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
            xn = c2 * x0h + c3 * xt;                                  // :173
            xn += c5 * z;                                             // :174
          } else if (a.mode == MODE_CDDPM_DDIM) {
            float x0h = (xt - c1 * eps) / c0;
            if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
            xn = c2 * x0h + c3 * eps;                                 // :177
          } else {
            xn = xt + eps / c0;                                       // cnf.py:108
            if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
          }
          xs[r] = in ? xn : 0.0f;
        });
      } else {

      // ================= VJP of the update: dL/deps and the direct part of dL/dx_j ===============================
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4];
      float ge[NRX], gxd[NRX];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const float eps = O[r], xt = xs[r], gn = gx[r];
        float g_e, g_x;
        if (a.mode == MODE_DDPM_DDPM) {
          const float x0h = c0 * (xt - c1 * eps);
          const float m = (!a.clip || (x0h >= a.x_min && x0h <= a.x_max)) ? 1.0f : 0.0f;
          const float g0 = c2 * c3 * gn * m;
          g_e = -c0 * c1 * g0; g_x = c2 * c4 * gn + c0 * g0;
        } else if (a.mode == MODE_DDPM_DDIM) {
          const float xn = c0 * (xt - c1 * eps) + c2 * eps;
          const float m = (!a.clip || (xn >= a.x_min && xn <= a.x_max)) ? 1.0f : 0.0f;
          const float gp = gn * m;
          g_e = gp * (c2 - c0 * c1); g_x = c0 * gp;
        } else if (a.mode == MODE_CDDPM_DDPM || a.mode == MODE_CDDPM_DDIM) {
          const float x0h = (xt - c1 * eps) / c0;
          const float m = (!a.clip || (x0h >= a.x_min && x0h <= a.x_max)) ? 1.0f : 0.0f;
          const float g0 = c2 * gn * m;
          if (a.mode == MODE_CDDPM_DDPM) { g_e = -(c1 / c0) * g0; g_x = c3 * gn + g0 / c0; }
          else { g_e = -(c1 / c0) * g0 + c3 * gn; g_x = g0 / c0; }
        } else {
          const float xn = xt + eps / c0;
          const float m = (!a.clip || (xn >= a.x_min && xn <= a.x_max)) ? 1.0f : 0.0f;
          g_e = gn * m / c0; g_x = gn * m;
        }
        const int ai = BF_ROWR(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        ge[r] = in ? g_e : 0.0f;
        gxd[r] = in ? g_x : 0.0f;
      });
      u32x4 gep;
      sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; gep[w] = pack2(ge[2 * w], ge[2 * w + 1]); });
      *(u32x4*)((char*)b.ge + (size_t)ts * 1024 + lane16) = gep;

      // ================= backward through the net ================================================================
      // epilogue of a backward block: dL/dz = dL/dh * act'(z) (act' recomputed in f32 from the stashed bf16 z, fetched by
      // init), packed = the next B operand, stashed as the G operand of the weight gradients
      u32x4 dpre[2][2];
      auto init_bwd = [&](int lsoff) {
        return [&dpre, lsoff, &rs_ad, lane16](auto OI, f32x16& acc) {
          constexpr int ob = decltype(OI)::value;
          sfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; acc[r] = 0.0f; });
          dpre[ob & 1][0] = ld16(rs_ad, lane16, lsoff + (2 * ob) * 1024);
          dpre[ob & 1][1] = ld16(rs_ad, lane16, lsoff + (2 * ob + 1) * 1024);
        };
      };
      auto epi_bwd = [&](u32x4 (&out)[16], int lsoff) {
        char* pg = (char*)(b.gd + srow) + lsoff;
        return [&out, &dpre, pg, lane16](auto OI, const f32x16& acc) {
          constexpr int ob = decltype(OI)::value;
          sfor<2>([&](auto QI) {
            constexpr int q = decltype(QI)::value;
            u32x4 gp;
            sfor<4>([&](auto WI) {
              constexpr int w = decltype(WI)::value;
              const unsigned zw = dpre[ob & 1][q][w];
              float h0, d0, h1, d1;
              act_grad_bf<ACT>(bf_lo(zw), h0, d0);
              act_grad_bf<ACT>(bf_hi(zw), h1, d1);
              gp[w] = pack2(acc[8 * q + 2 * w] * d0, acc[8 * q + 2 * w + 1] * d1);
            });
            out[2 * ob + q] = gp;
            st16(gp, pg, lane16, (2 * ob + q) * 1024);
          });
        };
      };
      {
        auto bg = [&](auto) -> u32x4 { return gep; };
        layer_g<1, 8>(bg, init_bwd((a.L - 1) * 16384), epi_bwd(bin, (a.L - 1) * 16384), next_chunk, lane16);
      }
      for (int l = a.L - 1; l >= 1; --l) {
        auto bh = [&](auto KI) -> u32x4 { return bin[decltype(KI)::value]; };
        layer_g<16, 8>(bh, init_bwd((l - 1) * 16384), epi_bwd(bout, (l - 1) * 16384), next_chunk, lane16);
        sfor<16>([&](auto KI) { constexpr int k = decltype(KI)::value; bin[k] = bout[k]; });
      }
      f32x16 OX;
      {
        auto bh = [&](auto KI) -> u32x4 { return bin[decltype(KI)::value]; };
        auto init_z = [&](auto, f32x16& acc) { sfor<16>([&](auto RI) { constexpr int r = decltype(RI)::value; acc[r] = 0.0f; }); };
        auto epi_x = [&](auto, const f32x16& acc) { OX = acc; };
        layer_g<16, 1>(bh, init_z, epi_x, next_chunk, lane16);
      }
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = BF_ROWR(r) + 4 * ls;
        gx[r] = (valid && ai < a.A) ? gxd[r] + OX[r] : 0.0f;
      });
      }   // PHASE != 1
    }
    if constexpr (PHASE == 1) {
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = BF_ROWR(r) + 4 * ls;
        if (valid && ai < a.A) a.x0[chain * a.A + ai] = xs[r];
      });
    } else if (b.g_xT != nullptr) {
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = BF_ROWR(r) + 4 * ls;
        if (valid && ai < a.A) b.g_xT[chain * a.A + ai] = gx[r];
      });
    }
  }
  // the stream runs three chunks ahead: drain the prefetches before the wave ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

#if FRL_BWD16_PHASE == 0
// ---- weight gradients ---------------------------------------------------------------------------------------------
constexpr int DW_ROWB = 80;                    // bytes per LDS feature row: 32 chains x bf16 + 16 pad (conflict-free b128 reads)
constexpr int DW_TILE = 256 * DW_ROWB;         // one operand tile
enum { DW_FIRST = 0, DW_HIDDEN = 1, DW_LAST = 2 };

// feature held by element e of lane half s in stash piece ks (the B-operand layout of the sweep)
__device__ __forceinline__ int piece_feature(int ks, int s, int e) { return 32 * (ks >> 1) + BF_ROWR(8 * (ks & 1) + e) + 4 * s; }

__device__ __forceinline__ void lds_put_piece(char* tile, int ks, int li, int ls, u32x4 v) {
  sfor<4>([&](auto WI) {
    constexpr int w = decltype(WI)::value;
    *(unsigned short*)(tile + piece_feature(ks, ls, 2 * w) * DW_ROWB + li * 2) = (unsigned short)(v[w] & 0xffffu);
    *(unsigned short*)(tile + piece_feature(ks, ls, 2 * w + 1) * DW_ROWB + li * 2) = (unsigned short)(v[w] >> 16);
  });
}

// h = act(z) of a stashed piece (8 bf16 z per lane), re-packed: the X operand of the next layer's weight gradient
template <int ACT>
__device__ __forceinline__ u32x4 act_piece(u32x4 z) {
  u32x4 o;
  sfor<4>([&](auto WI) {
    constexpr int w = decltype(WI)::value;
    float h0, d0, h1, d1;
    act_grad_bf<ACT>(bf_lo(z[w]), h0, d0);
    act_grad_bf<ACT>(bf_hi(z[w]), h1, d1);
    o[w] = pack2(h0, h1);
  });
  return o;
}

template <int KIND, int ACT>
__device__ __forceinline__ void dw_layer(const BArgs& b, int lam, char* smem, long long u0, long long u1, long long n_wtv) {
  const KArgs& a = b.k;
  constexpr int H = 256;
  constexpr int NM = (KIND == DW_HIDDEN) ? 2 : 1, NN = (KIND == DW_HIDDEN) ? 4 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int A = a.A, C = a.C, T = a.T, L = a.L;
  const int mb0 = (KIND == DW_HIDDEN) ? 2 * (wave >> 1) : (KIND == DW_LAST ? wave : 0);
  const int nb0 = (KIND == DW_HIDDEN) ? 4 * (wave & 1) : (KIND == DW_FIRST ? wave : 0);

  f32x16 acc[NM][NN];
  sfor<NM>([&](auto MI) { sfor<NN>([&](auto NI) { sfor<16>([&](auto RI) {
    acc[decltype(MI)::value][decltype(NI)::value][decltype(RI)::value] = 0.0f; }); }); });
  float bsum = 0.0f;
  int jcur = -1;
  const DwPart dp = dw_part_layout(L, T);
  float* slot = b.part + (size_t)blockIdx.x * dp.stride;   // this workgroup's partial sums (frl_bwd_reduce_kernel adds the slots in order)
  if constexpr (KIND == DW_FIRST) {
    if (tid < 256) for (int j = 0; j < T; ++j) slot[dp.tb + (size_t)j * H + tid] = 0.0f;
  }

  if (KIND == DW_LAST) {   // dL/deps occupies rows 0..15 of its block: rows 16..31 stay zero in both buffers
    for (int idx = tid; idx < 2 * 32 * DW_ROWB / 4; idx += 512) {
      const int bufi = idx / (32 * DW_ROWB / 4), off = idx - bufi * (32 * DW_ROWB / 4);
      *(unsigned*)(smem + bufi * 2 * DW_TILE + DW_TILE + off * 4) = 0u;
    }
  }

  // staging registers of TWO tile-steps in flight (sets 0 / 1 by the parity of u - u0): the loads of tile-step u + 3 are
  // issued when tile-step u + 1 is staged, two iterations (~ 2 x (16 MFMAs + transposes)) before they are needed — one
  // iteration ahead left most of the HBM latency exposed (round 2: 0.93 ms at 4.2 TB/s)
  u32x4 xr[2][2], gr[2][2];
  float xv[2][2];
  auto unit = [&](long long u, long long& ts, int& j, long long& wt) { j = (int)(u / n_wtv); wt = u - (long long)j * n_wtv; ts = wt * T + j; };
  auto fetch = [&](long long u, auto SI) {
    constexpr int st = decltype(SI)::value;
    long long ts, wt; int j;
    unit(u, ts, j, wt);
    if constexpr (KIND == DW_FIRST) {
      sfor<2>([&](auto EI) {
        constexpr int k = decltype(EI)::value;
        const int idx = tid + 512 * k, row = idx >> 5, c = idx & 31;
        const long long chain = wt * 32 + c;
        float v = 0.0f;
        if (chain < a.B) {
          if (row < A) v = b.xt_hist[((long long)j * a.B + chain) * A + row];
          else if (row < A + C) v = a.cond[((a.particles == 1) ? chain : chain / a.particles) * C + (row - A)];
        }
        xv[st][k] = v;
      });
    } else {
      const char* src = (const char*)b.hd + ((size_t)ts * L + (lam - 1)) * 16384 + lane * 16;
      xr[st][0] = *(const u32x4*)(src + wave * 1024);
      xr[st][1] = *(const u32x4*)(src + (wave + 8) * 1024);
    }
    if constexpr (KIND == DW_LAST) {
      if (wave == 0) gr[st][0] = *(const u32x4*)((const char*)b.ge + (size_t)ts * 1024 + lane * 16);
    } else {
      const char* src = (const char*)b.gd + ((size_t)ts * L + lam) * 16384 + lane * 16;
      gr[st][0] = *(const u32x4*)(src + wave * 1024);
      gr[st][1] = *(const u32x4*)(src + (wave + 8) * 1024);
    }
  };
  auto stage = [&](char* buf, auto SI) {
    constexpr int st = decltype(SI)::value;
    char* sX = buf;
    char* sG = buf + DW_TILE;
    if constexpr (KIND == DW_FIRST) {
      sfor<2>([&](auto EI) {
        constexpr int k = decltype(EI)::value;
        const int idx = tid + 512 * k, row = idx >> 5, c = idx & 31;
        *(unsigned short*)(sX + row * DW_ROWB + c * 2) = (unsigned short)(pack2(xv[st][k], 0.0f) & 0xffffu);
      });
    } else {
      lds_put_piece(sX, wave, li, lh, act_piece<ACT>(xr[st][0]));
      lds_put_piece(sX, wave + 8, li, lh, act_piece<ACT>(xr[st][1]));
    }
    if constexpr (KIND == DW_LAST) {
      if (wave == 0) lds_put_piece(sG, 0, li, lh, gr[st][0]);
    } else {
      lds_put_piece(sG, wave, li, lh, gr[st][0]);
      lds_put_piece(sG, wave + 8, li, lh, gr[st][1]);
    }
  };
  typedef std::integral_constant<int, 0> S0;
  typedef std::integral_constant<int, 1> S1;

  if (u0 < u1) {
    fetch(u0, S0{});
    stage(smem, S0{});
    if (u0 + 1 < u1) fetch(u0 + 1, S1{});
    if (u0 + 2 < u1) fetch(u0 + 2, S0{});
  }
  __syncthreads();
  auto body = [&](long long u, auto PI) {
    constexpr int par = decltype(PI)::value;
    char* buf = smem + par * 2 * DW_TILE;
    const char* sX = buf;
    const char* sG = buf + DW_TILE;
    // ---- bias gradient: column sums of G ----
    if (KIND != DW_LAST || tid < 32) {
      if (tid < 256) {
        if constexpr (KIND == DW_FIRST) {
          const int j = (int)(u / n_wtv);
          if (j != jcur) {
            if (jcur >= 0) slot[dp.tb + (size_t)jcur * H + tid] = bsum;
            bsum = 0.0f; jcur = j;
          }
        }
        float s = 0.0f;
        sfor<4>([&](auto VI) {
          constexpr int v = decltype(VI)::value;
          const u32x4 gv = *(const u32x4*)(sG + tid * DW_ROWB + v * 16);
          sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; s += bf_lo(gv[w]) + bf_hi(gv[w]); });
        });
        bsum += s;
      }
    }
    // ---- contract over the 32 chains: two k-steps of 16 ----
    sfor<2>([&](auto KI) {
      constexpr int kk = decltype(KI)::value;
      u32x4 av[NM], bv[NN];
      sfor<NM>([&](auto MI) { constexpr int m = decltype(MI)::value; av[m] = *(const u32x4*)(sX + (32 * (mb0 + m) + li) * DW_ROWB + kk * 32 + lh * 16); });
      sfor<NN>([&](auto NI) { constexpr int n = decltype(NI)::value; bv[n] = *(const u32x4*)(sG + (32 * (nb0 + n) + li) * DW_ROWB + kk * 32 + lh * 16); });
      sfor<NM>([&](auto MI) { constexpr int m = decltype(MI)::value; sfor<NN>([&](auto NI) { constexpr int n = decltype(NI)::value;
        acc[m][n] = mfma_bf(av[m], bv[n], acc[m][n]); }); });
    });
    // ---- next tile-step into the other buffer (its loads were issued two iterations ago); then tile-step u + 3's loads ----
    if (u + 1 < u1) {
      stage(smem + (par ^ 1) * 2 * DW_TILE, std::integral_constant<int, par ^ 1>{});
      if (u + 3 < u1) fetch(u + 3, std::integral_constant<int, par ^ 1>{});
    }
    __syncthreads();
  };
  for (long long u = u0; u < u1; u += 2) {
    body(u, S0{});
    if (u + 1 < u1) body(u + 1, S1{});
  }

  // ---- flush: partial tile into this workgroup's slot (plain stores; the second stage adds the slots in order) ----
  float* pk = slot + (KIND == DW_FIRST ? dp.k0 : KIND == DW_HIDDEN ? dp.kh + (size_t)(lam - 1) * 65536 : dp.kl);
  constexpr int PCOL = (KIND == DW_LAST) ? 32 : H;
  sfor<NM>([&](auto MI) {
    constexpr int m = decltype(MI)::value;
    sfor<NN>([&](auto NI) {
      constexpr int n = decltype(NI)::value;
      sfor<16>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int in = 32 * (mb0 + m) + BF_ROWR(r) + 4 * lh;
        const int out = 32 * (nb0 + n) + li;
        pk[(size_t)in * PCOL + out] = acc[m][n][r];
      });
    });
  });
  if constexpr (KIND == DW_FIRST) {
    if (tid < 256 && jcur >= 0) slot[dp.tb + (size_t)jcur * H + tid] = bsum;
  } else {
    if (tid < 256) slot[dp.bias + (size_t)(lam - 1) * 256 + tid] = (KIND == DW_LAST && tid >= 32) ? 0.0f : bsum;
  }
  __syncthreads();
}

template <int ACT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
frl_bwd_dw_bf16_kernel(const BArgs b) {
  extern __shared__ __attribute__((aligned(16))) char smem_dw16[];
  const KArgs& a = b.k;
  const long long n_wtv = (a.B + 31) / 32;               // wave tiles that hold chains
  const long long U = n_wtv * a.T;                        // tile-steps, step-major: u = j * n_wtv + wt
  const long long u0 = U * blockIdx.x / gridDim.x, u1 = U * (blockIdx.x + 1) / gridDim.x;
  dw_layer<DW_FIRST, ACT>(b, 0, smem_dw16, u0, u1, n_wtv);
  for (int lam = 1; lam < a.L; ++lam) dw_layer<DW_HIDDEN, ACT>(b, lam, smem_dw16, u0, u1, n_wtv);
  dw_layer<DW_LAST, ACT>(b, a.L, smem_dw16, u0, u1, n_wtv);
}

template <int ACT>
static int launch_dw16_inst(const BArgs& b, hipStream_t stream) {
  const int g2 = b.n_part;                                // one partial slot per workgroup (BwdPlan16.n_part)
  const size_t lds2 = 4 * (size_t)DW_TILE;
  auto kern = frl_bwd_dw_bf16_kernel<ACT>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(g2), dim3(512), lds2, stream, b);
  note_kernel("frl_bwd_dw_bf16_kernel<act %d> [grid %d] + frl_bwd_reduce_kernel", ACT, g2);
  return FRL_OK;
}

int launch_bwd_dw_bf16(const BArgs& b, int act, hipStream_t stream) {
  int rc = act == FRL_ACT_MISH ? launch_dw16_inst<FRL_ACT_MISH>(b, stream) : act == FRL_ACT_SILU ? launch_dw16_inst<FRL_ACT_SILU>(b, stream)
                                                                                                  : launch_dw16_inst<FRL_ACT_RELU>(b, stream);
  if (rc != FRL_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 bwd dW launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return launch_bwd_reduce(b, stream);
}
#endif  // FRL_BWD16_PHASE == 0

// ---- launcher -----------------------------------------------------------------------------------------------------
template <int ACT>
static int launch_sweep16(const BArgs& b, int grid, size_t lds, hipStream_t stream) {
  auto kern = frl_bwd_sweep_bf16_kernel<ACT, FRL_BWD16_PHASE>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, b);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("bf16 bwd sweep launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_bwd_sweep_bf16_kernel<act %d, PHASE %d> [grid %d]", ACT, FRL_BWD16_PHASE, grid);
  return FRL_OK;
}

template <>
int launch_bwd16_sweep_phase<FRL_BWD16_PHASE>(const FrlDesc& d, const BArgs& b, hipStream_t stream) {
  const KArgs& a = b.k;
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
  const long long ntiles = (a.B + 255) / 256;
  const int grid = (int)(ntiles < ncu ? ntiles : ncu);
  const size_t lds = bf16_lds_bytes(a.T, a.L);
  if (lds > 160 * 1024) { set_error("bf16 bwd: steps=%d needs %zu B of LDS (> 160 KiB)", a.T, lds); return FRL_EUNSUPPORTED; }
  int rc;
  switch (d.act) {
    case FRL_ACT_MISH: rc = launch_sweep16<FRL_ACT_MISH>(b, grid, lds, stream); break;
    case FRL_ACT_SILU: rc = launch_sweep16<FRL_ACT_SILU>(b, grid, lds, stream); break;
    case FRL_ACT_RELU: rc = launch_sweep16<FRL_ACT_RELU>(b, grid, lds, stream); break;
    default: set_error("bf16 bwd: activation %d has no VJP kernel (mish / silu / relu)", d.act); return FRL_EUNSUPPORTED;
  }
  return rc;
}

#if FRL_BWD16_PHASE == 0
// phase 0: recompute + backward; 1: training forward (no gradients yet); 2: backward from the stash
int launch_sampler_bwd_bf16(const FrlDesc& d, const BArgs& b, int phase, hipStream_t stream) {
  int rc = phase == 0 ? launch_bwd16_sweep_phase<0>(d, b, stream) : phase == 1 ? launch_bwd16_sweep_phase<1>(d, b, stream)
                                                                                 : launch_bwd16_sweep_phase<2>(d, b, stream);
  if (rc != FRL_OK || phase == 1) return rc;
  return launch_bwd_dw_bf16(b, d.act, stream);
}
#endif

}  // namespace frl
