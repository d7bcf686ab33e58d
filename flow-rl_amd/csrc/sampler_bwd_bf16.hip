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
        // the network input of this step as the dW kernel's X operand of Dense_0: the packed operands themselves
        {
          char* px0 = (char*)b.x0d + (size_t)ts * 4096;
          st16(bx, px0, lane16, 0);
          sfor<CKS>([&](auto CI) { constexpr int kc = decltype(CI)::value; st16(cbp[kc], px0, lane16, (1 + kc) * 1024); });
        }
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
          const float xn = bf_solver_update(a.mode, a.clip, a.x_min, a.x_max, c0, c1, c2, c3, c4, c5, xt, eps, z);
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
// dW_l = sum over chains and steps of X_l (x) dL/dz_l.  The contraction index (chains) sits on the LANES of the stash
// pieces while an MFMA wants it inside a lane, so the pieces pass through LDS — but nothing is scattered: a 16 KB stash
// block goes into LDS by LDS-DMA (global_load_lds_dwordx4, per-lane SOURCE addresses permuted so that the linear LDS image
// is the one the reads want) and the operands come back out with gfx950's transposing read ds_read_b64_tr_b16, which
// hands each lane four CHAINS of one feature.  Image: the 16-byte chunk of chain c (0..31) and feature quad m = f / 4
// holding features 4m..4m+3 and 4m+8..4m+11 (m & 2 == 0: exactly a stash chunk, piece ks = 2 (m >> 3) + ((m >> 2) & 1),
// lane half m & 1) sits at slot
//     (c & 3) | (m & 1) << 2 | ((m >> 2) & 1) << 3 | (c >> 2) << 4 | (m >> 3) << 7
// so that the 32 lanes of a transposed read (4 chains x 32 features) cover one 256-byte bank row exactly: conflict-free.
// X = act(z) is applied IN PLACE on the landed image (one ds_read_b128 / ds_write_b128 per chunk, lane-linear) one
// tile-step ahead of the MFMAs; dL/dz needs nothing.  Bias gradients (column sums of dL/dz over chains) ride on the MFMA
// pipe: an all-ones A operand against the same B operand.  All global traffic of the loop is LDS-DMA with counted vmcnt
// waits (four 33 KB buffers, three tile-steps in flight, one s_barrier per tile-step).
constexpr int DW_IMG = 16384;                  // one operand image: 32 chains x 256 features, bf16
constexpr int DW_BUF = 2 * DW_IMG + 1024;      // X image | G image | the dL/deps piece as it is stashed (last layer)
constexpr int DW_NBUF = 4;
enum { DW_FIRST = 0, DW_HIDDEN = 1, DW_LAST = 2 };

typedef short s16x4 __attribute__((ext_vector_type(4)));

// The operands of one 16-chain k-step, transposed out of the images: per 32-feature block two ds_read_b64_tr_b16 (chains
// 8 h + 0..3 and 8 h + 4..7 of the lane's feature).  Inline asm, with its own lgkmcnt(0) at the end: behind the builtin
// (__builtin_amdgcn_ds_read_tr16_b64_*) hipcc waits vmcnt(0) before the first read of every k-step — it cannot tell that
// the LDS-DMA still in flight targets OTHER buffers — which would drain the three-deep DMA pipeline every tile-step.
// pa / pb: this lane's address in the X / G image for block 0 of the wave and this k-step.
template <int NM, int NN>
__device__ __forceinline__ void tr_operands(unsigned pa, unsigned pb, u32x4 (&av)[NM], u32x4 (&bv)[NN]) {
  u32x2 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h, b2l, b2h, b3l, b3h;
  if constexpr (NM == 2 && NN == 4) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %12\n"
        "ds_read_b64_tr_b16 %1, %12 offset:256\n"
        "ds_read_b64_tr_b16 %2, %12 offset:2048\n"
        "ds_read_b64_tr_b16 %3, %12 offset:2304\n"
        "ds_read_b64_tr_b16 %4, %13\n"
        "ds_read_b64_tr_b16 %5, %13 offset:256\n"
        "ds_read_b64_tr_b16 %6, %13 offset:2048\n"
        "ds_read_b64_tr_b16 %7, %13 offset:2304\n"
        "ds_read_b64_tr_b16 %8, %13 offset:4096\n"
        "ds_read_b64_tr_b16 %9, %13 offset:4352\n"
        "ds_read_b64_tr_b16 %10, %13 offset:6144\n"
        "ds_read_b64_tr_b16 %11, %13 offset:6400\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h), "=&v"(b0l), "=&v"(b0h), "=&v"(b1l), "=&v"(b1h), "=&v"(b2l), "=&v"(b2h),
          "=&v"(b3l), "=&v"(b3h)
        : "v"(pa), "v"(pb)
        : "memory");
    av[0] = u32x4{a0l[0], a0l[1], a0h[0], a0h[1]}; av[1] = u32x4{a1l[0], a1l[1], a1h[0], a1h[1]};
    bv[0] = u32x4{b0l[0], b0l[1], b0h[0], b0h[1]}; bv[1] = u32x4{b1l[0], b1l[1], b1h[0], b1h[1]};
    bv[2] = u32x4{b2l[0], b2l[1], b2h[0], b2h[1]}; bv[3] = u32x4{b3l[0], b3l[1], b3h[0], b3h[1]};
  } else if constexpr (NM == 2 && NN == 1) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %6\n"
        "ds_read_b64_tr_b16 %1, %6 offset:256\n"
        "ds_read_b64_tr_b16 %2, %6 offset:2048\n"
        "ds_read_b64_tr_b16 %3, %6 offset:2304\n"
        "ds_read_b64_tr_b16 %4, %7\n"
        "ds_read_b64_tr_b16 %5, %7 offset:256\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a0l), "=&v"(a0h), "=&v"(a1l), "=&v"(a1h), "=&v"(b0l), "=&v"(b0h)
        : "v"(pa), "v"(pb)
        : "memory");
    av[0] = u32x4{a0l[0], a0l[1], a0h[0], a0h[1]}; av[1] = u32x4{a1l[0], a1l[1], a1h[0], a1h[1]};
    bv[0] = u32x4{b0l[0], b0l[1], b0h[0], b0h[1]};
  } else {
    static_assert(NM == 1 && NN == 1, "operand block shape");
    asm volatile(
        "ds_read_b64_tr_b16 %0, %4\n"
        "ds_read_b64_tr_b16 %1, %4 offset:256\n"
        "ds_read_b64_tr_b16 %2, %5\n"
        "ds_read_b64_tr_b16 %3, %5 offset:256\n"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a0l), "=&v"(a0h), "=&v"(b0l), "=&v"(b0h)
        : "v"(pa), "v"(pb)
        : "memory");
    av[0] = u32x4{a0l[0], a0l[1], a0h[0], a0h[1]};
    bv[0] = u32x4{b0l[0], b0l[1], b0h[0], b0h[1]};
  }
  __builtin_amdgcn_sched_barrier(0);      // no MFMA may move above the wait (guide 5.4 rule 18)
}

__device__ __forceinline__ void dma16_at(const char* gsrc, char* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// h = act(z) of a stashed chunk (8 bf16 z per lane), re-packed: the X operand of the next layer's weight gradient
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
  // output blocks of a wave: hidden 2 x 4 of the 8 x 8; first layer (64 staged rows: x | cond k-steps | unused) 2 x 1;
  // last layer 1 x 1 of the 8 x 1
  constexpr int NM = (KIND == DW_LAST) ? 1 : 2, NN = (KIND == DW_HIDDEN) ? 4 : 1;
  constexpr int NDMA = (KIND == DW_HIDDEN) ? 4 : 3;         // LDS-DMA instructions per wave and tile-step
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int A = a.A, C = a.C, T = a.T, L = a.L;
  const int mb0 = (KIND == DW_HIDDEN) ? 2 * (wave >> 1) : (KIND == DW_LAST ? wave : 0);
  const int nb0 = (KIND == DW_HIDDEN) ? 4 * (wave & 1) : (KIND == DW_FIRST ? wave : 0);
  // the block whose column sums this wave carries (hidden: one of its four, so that the eight are spread over the waves)
  constexpr bool kBiasAll = (KIND != DW_LAST);
  const int nbias = (KIND == DW_HIDDEN) ? (wave >> 1) : 0;
  const bool do_bias = kBiasAll || wave == 0;

  f32x16 acc[NM][NN], accb;
  sfor<NM>([&](auto MI) { sfor<NN>([&](auto NI) { sfor<16>([&](auto RI) {
    acc[decltype(MI)::value][decltype(NI)::value][decltype(RI)::value] = 0.0f; }); }); });
  sfor<16>([&](auto RI) { accb[decltype(RI)::value] = 0.0f; });
  int jcur = -1;
  const DwPart dp = dw_part_layout(L, T);
  float* slot = b.part + (size_t)blockIdx.x * dp.stride;   // this workgroup's partial sums (frl_bwd_reduce_kernel adds the slots in order)
  const int bcol = 32 * (nb0 + nbias) + li;                 // the column of this lane's bias / dL/dtb sums
  if constexpr (KIND == DW_FIRST) {
    // rows of steps this workgroup never visits stay zero; the same lane later overwrites the rows it does visit
    if (lh == 0) for (int j = 0; j < T; ++j) slot[dp.tb + (size_t)j * H + bcol] = 0.0f;
  }
  if constexpr (KIND == DW_LAST) {
    // dL/deps fills features 0..15 of block 0 (chunks with m < 4); features 16..31 (m = 4..7) stay zero in every buffer
    for (int idx = tid; idx < DW_NBUF * 64; idx += 512) {
      const int bufi = idx >> 6, k = idx & 63;
      *(u32x4*)(smem + bufi * DW_BUF + DW_IMG + ((k & 7) | 8 | (k >> 3) << 4) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }

  // per-lane parts of the addresses
  //   DMA source inside a 16 KB stash block for LDS slot 64 i + lane: piece 2 (i >> 1) + ((lane >> 3) & 1), lane half
  //   (lane >> 2) & 1, chain (lane & 3) | ((lane >> 4) & 3) << 2 | (i & 1) << 4
  const unsigned g_lane = ((lane >> 3) & 1) * 1024 + ((lane >> 2) & 1) * 512 + ((lane & 3) + 4 * ((lane >> 4) & 3)) * 16;
  //   transposed read: lane 4 q + p of the 16-lane group G supplies chain row q, feature quad 4 (G & 1) + p
  const int trG = lane >> 4, trq = (lane >> 2) & 3, trp = lane & 3;
  const unsigned tr_lane = 16 * trq + 64 * (trp & 1) + 8 * (trp >> 1) + 128 * (trG & 1) + 512 * (trG >> 1);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;      // LDS offset of the dynamic segment

  // tile-steps are visited in order: (step, wave tile) advance incrementally (one 64-bit division per layer, not per tile-step)
  int ij = (int)(u0 / n_wtv);
  long long iwt = u0 - (long long)ij * n_wtv;
  auto issue = [&](int bufi) {
    const long long ts = iwt * T + ij;
    if (++iwt == n_wtv) { iwt = 0; ++ij; }
    char* buf = smem + bufi * DW_BUF;
    if constexpr (KIND == DW_FIRST) {
      const int i = wave & 3;              // 4 KB block (x | cond | cond | unused): waves 4..7 repeat waves 0..3 (uniform DMA count)
      dma16_at((const char*)b.x0d + (size_t)ts * 4096 + (i >> 1) * 2048 + (i & 1) * 256 + g_lane, buf + i * 1024);
    } else {
      const char* src = (const char*)b.hd + ((size_t)ts * L + (lam - 1)) * 16384 + g_lane;
      sfor<2>([&](auto EI) { const int i = wave + 8 * decltype(EI)::value; dma16_at(src + (i >> 1) * 2048 + (i & 1) * 256, buf + i * 1024); });
    }
    if constexpr (KIND == DW_LAST) {
      dma16_at((const char*)b.ge + (size_t)ts * 1024 + lane * 16, buf + 2 * DW_IMG);      // every wave: the same 1 KB (uniform DMA count)
    } else {
      const char* src = (const char*)b.gd + ((size_t)ts * L + lam) * 16384 + g_lane;
      sfor<2>([&](auto EI) { const int i = wave + 8 * decltype(EI)::value; dma16_at(src + (i >> 1) * 2048 + (i & 1) * 256, buf + DW_IMG + i * 1024); });
    }
  };
  // the landed X image of a buffer: z -> act(z) in place; last layer: the dL/deps piece into its image slots
  auto finish = [&](int bufi) {
    char* buf = smem + bufi * DW_BUF;
    if constexpr (KIND != DW_FIRST) {
      sfor<2>([&](auto EI) {
        char* p = buf + (tid + 512 * decltype(EI)::value) * 16;
        *(u32x4*)p = act_piece<ACT>(*(const u32x4*)p);
      });
    }
    if constexpr (KIND == DW_LAST) {
      if (wave == 0) *(u32x4*)(buf + DW_IMG + ((li & 3) | lh << 2 | (li >> 2) << 4) * 16) = *(const u32x4*)(buf + 2 * DW_IMG + lane * 16);
    }
  };
  const u32x4 ones = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  auto flush_tb = [&]() {
    if (jcur >= 0 && lh == 0) slot[dp.tb + (size_t)jcur * H + bcol] = accb[0];
    sfor<16>([&](auto RI) { accb[decltype(RI)::value] = 0.0f; });
  };

  const long long n = u1 - u0;
  const bool mfma_first = (wave & 4) != 0;
  int mj = (int)(u0 / n_wtv);                  // the MFMA loop's own (step, wave tile) position
  long long mwt = u0 - (long long)mj * n_wtv;
  if (n > 0) {
    issue(0);
    if (n > 1) issue(1);
    if (n > 2) issue(2);
    if (n > 2) wait_vm_lgkm<2 * NDMA>(); else if (n > 1) wait_vm_lgkm<NDMA>(); else wait_vm_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    finish(0);
  }
  for (long long r = 0; r < n; ++r) {
    const int cur = (int)(r & 3);
    // tile-step r + 1 has landed (the newest NDMA instructions, tile-step r + 2's, may still be in flight); every wave
    // has left iteration r - 1: its reads of buffer (r - 1) & 3 and its pass over buffer r & 3 are done
    if (r + 2 < n) wait_vm_lgkm<NDMA>(); else wait_vm_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (r + 3 < n) issue((int)((r + 3) & 3));
    // The two halves of the workgroup run the iteration's two independent parts in opposite orders: waves w and w + 4
    // share a SIMD, so while one is in its activation pass (VALU) the other is in its MFMAs — behind ONE barrier per
    // tile-step both would otherwise do the same part at the same time (measured: 2.6 k cycles per tile-step, the sum).
    if (!mfma_first && r + 1 < n) finish((int)((r + 1) & 3));
    if constexpr (KIND == DW_FIRST) {
      if (mj != jcur) { flush_tb(); jcur = mj; }
      if (++mwt == n_wtv) { mwt = 0; ++mj; }
    }
    const unsigned pX = lds0 + (unsigned)(cur * DW_BUF) + tr_lane + 2048u * mb0;      // LDS byte addresses
    const unsigned pG = lds0 + (unsigned)(cur * DW_BUF + DW_IMG) + tr_lane + 2048u * nb0;
    sfor<2>([&](auto KI) {
      constexpr int kk = decltype(KI)::value;
      u32x4 av[NM], bv[NN];
      tr_operands<NM, NN>(pX + 1024u * kk, pG + 1024u * kk, av, bv);
      sfor<NM>([&](auto MI) { constexpr int m = decltype(MI)::value; sfor<NN>([&](auto NI) { constexpr int nn = decltype(NI)::value;
        acc[m][nn] = mfma_bf(av[m], bv[nn], acc[m][nn]); }); });
      if (do_bias) {
        if constexpr (KIND == DW_HIDDEN) {
          const u32x4 bb = nbias == 0 ? bv[0] : nbias == 1 ? bv[1] : nbias == 2 ? bv[2] : bv[3];
          accb = mfma_bf(ones, bb, accb);
        } else {
          accb = mfma_bf(ones, bv[0], accb);
        }
      }
    });
    if (mfma_first && r + 1 < n) finish((int)((r + 1) & 3));
  }

  // ---- flush: partial tile into this workgroup's slot (plain stores; the second stage adds the slots in order) ----
  if constexpr (KIND == DW_FIRST) {
    // staged rows of the slot: x rows 0..A-1, then the condition features; the operand rows are x (piece 0: row
    // ROWR(e) + 4 s = the feature index itself), then cond k-step kc at 16 (1 + kc) + [8 (e >> 2) + 4 s + (e & 3)] for
    // condition feature 16 kc + 8 s + e
    float* pk = slot + dp.k0;
    sfor<NM>([&](auto MI) {
      constexpr int m = decltype(MI)::value;
      sfor<16>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int in = 32 * m + BF_ROWR(r) + 4 * lh;
        int row = -1;
        if (in < 16) { if (in < A) row = in; }
        else if (in < 48) {
          const int kc = (in >> 4) - 1, w = in & 15;
          const int cf = 16 * kc + 8 * ((w >> 2) & 1) + 4 * ((w >> 3) & 1) + (w & 3);
          if (cf < C) row = A + cf;
        }
        if (row >= 0) pk[(size_t)row * H + 32 * nb0 + li] = acc[m][0][r];
      });
    });
    flush_tb();
  } else {
    float* pk = slot + (KIND == DW_HIDDEN ? dp.kh + (size_t)(lam - 1) * 65536 : dp.kl);
    constexpr int PCOL = (KIND == DW_LAST) ? 32 : H;
    sfor<NM>([&](auto MI) {
      constexpr int m = decltype(MI)::value;
      sfor<NN>([&](auto NI) {
        constexpr int nn = decltype(NI)::value;
        sfor<16>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int in = 32 * (mb0 + m) + BF_ROWR(r) + 4 * lh;
          const int out = 32 * (nb0 + nn) + li;
          pk[(size_t)in * PCOL + out] = acc[m][nn][r];
        });
      });
    });
    if constexpr (KIND == DW_HIDDEN) {
      if ((wave & 1) * 0 == 0 && lh == 0) slot[dp.bias + (size_t)(lam - 1) * 256 + bcol] = accb[0];
    } else {
      if (tid < 256) slot[dp.bias + (size_t)(lam - 1) * 256 + tid] = (tid < 32) ? accb[0] : 0.0f;   // tid < 32: wave 0, lane half 0
    }
  }
  wait_vm_lgkm<0>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
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
  const size_t lds2 = (size_t)DW_NBUF * DW_BUF;
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
