// Chain VJP of the fused sampler (DQL-style actor update: flowrl/agent/offline/dql.py:93-123 runs jax.grad
// through the T unrolled steps of DDPM.sample, flowrl/flow/ddpm.py:186-219; model.py:78-86).
//
// Three kernels, exact-f32 MFMA, 256-wide predictor, the transposed [feature x chain] register layout of
// sampler_rs.hip:
//
//   frl_bwd_prepare_kernel   transposed hidden kernels WT_l[out][in], row-padded W_L^T, packed Dense_0 x rows
//   frl_bwd_sweep_kernel     per wave-tile of 32 chains, steps j = T-1 .. 0:
//                              recompute the forward of step j from the stashed x_j (only xt_hist was kept),
//                              writing h_l = act(z_l) and a_l = act'(z_l) of every layer to the stash;
//                              VJP of the DDPM / DDIM / Euler update -> dL/deps, direct dL/dx_j;
//                              dL/dh_{l-1} = W_l dL/dz_l layer by layer — the forward's hidden-layer code with
//                              the transposed kernel, dL/dz_l = dL/dh_l * a_l applied lazily as the B operand
//                              and written to the stash; dL/dx_j += W0[x rows] dL/dz_0
//   frl_bwd_dw_kernel        dW_l = sum over chains and steps of h_{l-1} (x) dL/dz_l: a split-K GEMM whose K is
//                              the 1.3 M chain-steps — each CU owns a K range and the whole 256 x 256 tile in
//                              accumulators (256 per lane), operands transposed through LDS; every workgroup writes its
//                              partial tiles to its own slot of the workspace (plain stores), bias grads = column sums
//                              taken from the same LDS tiles
//   frl_bwd_reduce_kernel    adds the slots in slot order and writes every gradient leaf once: no float atomics, so
//                              two calls on the same inputs give the same bits ("VJP determinism").
//
// Weight gradients contract over CHAINS (the lane axis of the sweep), so they cannot be accumulated inside the
// sweep without 64 K accumulators per wave: the stash (3 x 32 KB per tile-step and layer) is what links the two.
#include "frl_device.h"
#include "frl_rng.h"
#include "frl_actgrad.h"

// One translation unit per sweep phase (-DFRL_BWD_PHASE=0|1|2, flow-rl_amd/build.py) so the three heavy instantiation sets
// compile in parallel; the phase-0 unit also carries the prepare, dW and reduce kernels.
#ifndef FRL_BWD_HRING
#define FRL_BWD_HRING 8
#endif
#ifndef FRL_BWD_PHASE
#error "compile with -DFRL_BWD_PHASE=<0|1|2>"
#endif

namespace frl {

__device__ __forceinline__ f32x2 ld2(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
// LDS-DMA: 64 lanes x 16 B, per-lane global source, LDS destination = wave-uniform base + lane * 16 (+ OFF on both sides)
template <int OFF>
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, OFF, 0);
}

#if FRL_BWD_PHASE == 0
// ---- preparation: transposed / packed weight images -------------------------------------------------------
struct PBArgs {
  const float* kernel[FRL_MAX_HIDDEN + 1];
  float* wt[FRL_MAX_HIDDEN - 1];
  float* wlt; float* w0xq;
  float* w0cq;               // [4 blocks][H * 64]: Dense_0's cond rows 32 m .. 32 m + 31 as A operands (dL/dcfeat), or NULL
  int H, L, A, XR, C, c_row0;
};

__global__ void __launch_bounds__(256) frl_bwd_prepare_kernel(const PBArgs p) {
  const int H = p.H;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long gsz = (long long)gridDim.x * blockDim.x;
  for (int l = 1; l < p.L; ++l) {               // WT_l[o][i] = W_l[i][o]
    const float* W = p.kernel[l];
    float* WT = p.wt[l - 1];
    for (long long idx = gid; idx < (long long)H * H; idx += gsz) {
      const int o = (int)(idx / H), i = (int)(idx - (long long)o * H);
      WT[idx] = W[(size_t)i * H + o];
    }
  }
  const float* WL = p.kernel[p.L];               // (H, A)  ->  wlt[a][f] = WL[f][a], zero rows beyond A
  for (long long idx = gid; idx < (long long)p.XR * H; idx += gsz) {
    const int a = (int)(idx / H), f = (int)(idx - (long long)a * H);
    p.wlt[idx] = (a < p.A) ? WL[(size_t)f * p.A + a] : 0.0f;
  }
  // Dense_0 x rows as the A operand of dx[a] = sum_f W0[a][f] G_0[f]: same per-lane packing as the forward's
  // Dense_L image (load q = k-steps 4q..4q+3; lane (i, s) of k-step (block, r) holds feature
  // 128 g + j + 4 (ROWR(r) + 4 s), output row i = a)
  const float* W0 = p.kernel[0];
  for (long long idx = gid; idx < (long long)H * 64; idx += gsz) {
    const int q = (int)(idx >> 8), lane = (int)(idx >> 2) & 63, e = (int)idx & 3;
    const int k = 4 * q + e, blk = k / 16, r = k - blk * 16;
    const int i = lane & 31, sl = lane >> 5;
    const int kf = 128 * (blk >> 2) + (blk & 3) + 4 * ((r & 3) + 8 * (r >> 2) + 4 * sl);
    p.w0xq[idx] = (i < p.A && kf < H) ? W0[(size_t)i * H + kf] : 0.0f;
    if (p.w0cq != nullptr)
      for (int m = 0; m < 4; ++m)
        p.w0cq[(size_t)m * H * 64 + idx] = (32 * m + i < p.C && kf < H) ? W0[(size_t)(p.c_row0 + 32 * m + i) * H + kf] : 0.0f;
  }
}

int launch_prepare_bwd(const FrlParams& prm, const BwdPlan& bp, float* ws, hipStream_t stream) {
  PBArgs p;
  for (int l = 0; l <= bp.f.L; ++l) p.kernel[l] = prm.kernel[l];
  for (int l = 1; l < bp.f.L; ++l) p.wt[l - 1] = ws + bp.off_wt[l - 1];
  p.wlt = ws + bp.off_wlt; p.w0xq = ws + bp.off_w0xq;
  p.w0cq = bp.cond_embed ? ws + bp.off_w0cq : nullptr;
  p.H = bp.f.H; p.L = bp.f.L; p.A = bp.f.A; p.XR = bp.f.XR; p.C = bp.f.C; p.c_row0 = bp.f.c_row0;
  hipLaunchKernelGGL(frl_bwd_prepare_kernel, dim3(256), dim3(256), 0, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("bwd prepare launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

#endif  // FRL_BWD_PHASE == 0

// ---- the sweep ---------------------------------------------------------------------------------------------
// PHASE 0: recompute + backward in one kernel (the caller kept only xt_hist: frl_sampler_bwd without a stash)
// PHASE 1: the TRAINING FORWARD (frl_sampler_fwd_train): the sampler itself, steps j = 0 .. T-1 with the update rule,
//          writing x0, xt_hist, eps_hist AND the stash (h_l, act'(z_l)) the backward needs — so that
// PHASE 2: the backward sweep alone (steps T-1 .. 0) reads eps_hist and act' instead of recomputing the forward:
//          a third of the sweep's MFMA work and its two stash writes leave the backward call.
template <int ACT, int NRX, int CKMAX, int PHASE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_bwd_sweep_kernel(const BArgs b) {
  typedef Geo<32> G;
  typedef f32x16 acc_t;
  constexpr int TR = 32, H = 256, KS = 2, NR = 16, RING = 4, HRING = (NRX > 4) ? 4 : FRL_BWD_HRING;   // the wide build trades ring depth for operand registers
  constexpr bool WIDE = NRX > 4;
  constexpr int NB = H / TR, NG = NB / 4, GW = 4 * TR, NK = NB * NR, ROWB = H * 4, RINGL = 8, NLD = NK / 4;
  constexpr int N0 = NRX + CKMAX;
  constexpr int ARD = 4;            // act'(z) register ring of the backward: pair rows in flight (HBM latency at one wave per SIMD)
  const KArgs& a = b.k;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int li = lane % TR, ls = lane / TR;

  float* s_tb = (float*)smem;
  float* s_bh = s_tb + a.T * H;
  float* s_bl = s_bh + (a.L - 1) * H;
  float* s_coef = s_bl + LAST_PAD;
  for (int idx = tid; idx < a.T * H; idx += blockDim.x) s_tb[idx] = a.ws_tb[idx];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr_rs(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += blockDim.x) s_bh[(l - 1) * H + idx] = bp[idx];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.T * COEF_STRIDE; idx += blockDim.x) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  // PHASE 2: act'(z) of the layer being swept arrives through a per-wave 32 KB LDS buffer (two 16 KB halves = 32 pair rows
  // each), filled by LDS-DMA half a layer ahead.  On this chip every vector load of a wave returns in issue order, so an
  // HBM-latency act' load sitting between two weight-ring loads stalls the ring (L2 latency; 7 k-steps ≈ 1.5 us of lookahead) for the
  // difference — with one wave per SIMD nothing hides it (round 2: 60 % MFMA busy).  Batched, the latency is paid once per
  // half layer instead of once per pair row, and the consumer reads LDS.
  char* s_adw = (char*)(s_coef + a.T * COEF_STRIDE) + wave * 32768;
  auto issue_half = [&](const float* blk, int half) {          // blk: the 8192-float stash block of one (tile-step, layer)
    const char* src = (const char*)blk + half * 16384 + lane * 16;
    char* dst = s_adw + half * 16384;
    sfor<4>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      sfor<4>([&](auto II) { constexpr int i = decltype(II)::value; dma16<i * 1024>(src + g * 4096, dst + g * 4096); });
    });
  };

  const unsigned voff_h = ls * (16 * ROWB) + li * 16;
  const unsigned voff_x = ls * (4 * ROWB) + li * 16;
  const unsigned voff_c = ls * ROWB + li * 16;
  const unsigned voff_l = lane * 16;
  const unsigned voff_d = lane * 8;                       // stash rows: [pair][lane][2]
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_w0p, 0, a.w0p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_wlq, 0, NK * 64 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rslt = __builtin_amdgcn_make_buffer_rsrc((void*)b.ws_wlt, 0, a.XR * ROWB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsxq = __builtin_amdgcn_make_buffer_rsrc((void*)b.ws_w0xq, 0, NK * 64 * 4, 0x00020000);

  auto bias_init = [&](acc_t (&Q)[NB], const float* sb) {
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      sfor<NR>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const f32x4 bv = *(const f32x4*)(sb + g * GW + 4 * (G::rowr(r) + 4 * ls));
        sfor<4>([&](auto JI) { constexpr int j = decltype(JI)::value; Q[g * 4 + j][r] = bv[j]; });
      });
    });
  };
  auto zero_init = [&](acc_t (&Q)[NB]) {
    sfor<NB>([&](auto BI) { constexpr int bb = decltype(BI)::value; sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; Q[bb][r] = 0.0f; }); });
  };

  f32x4 ring[HRING][NG];   // hidden-layer weight ring: HRING - 1 k-steps ahead (stores share vmcnt with loads on gfx9: the lookahead must cover a store's acknowledge)
  auto hload = [&](__amdgpu_buffer_rsrc_t rs, auto KI) {
    constexpr int k = decltype(KI)::value % NK;
    constexpr int bb = k / NR, r = k % NR;
    constexpr int soff = (GW * (bb >> 2) + (bb & 3) + 4 * G::rowr(r)) * ROWB;
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      ring[k % HRING][g] = ldw(rs, voff_h + g * (GW * 4), soff);
    });
  };
  f32x4 r0[RING][NG];
  auto live = [&](int n) { return n < NRX ? n < a.AR : (n - NRX) < a.CK; };
  auto l0load = [&](auto NI) {
    constexpr int n = decltype(NI)::value;
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      if constexpr (n < NRX) r0[n % RING][g] = ldw(rs0, voff_x + g * (GW * 4), G::rowr(n) * ROWB);
      else r0[n % RING][g] = ldw(rs0, voff_c + g * (GW * 4), (a.XR + KS * (n - NRX)) * ROWB);
    });
  };

  acc_t P[NB], Q[NB];
  const long long n_wt = (a.B + TR - 1) / TR;
  if constexpr (PHASE == 2) {       // first half of the first block this wave sweeps: (its first tile, step T - 1, layer L - 1)
    const long long wt0 = (long long)blockIdx.x * nw + wave;
    if (wt0 < n_wt) {
      issue_half(b.ad + ((size_t)(wt0 * a.T + a.T - 1) * a.L + (a.L - 1)) * 8192, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }

  for (long long wt = (long long)blockIdx.x * nw + wave; wt < n_wt; wt += (long long)gridDim.x * nw) {
    const long long chain = wt * TR + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    float cb[CKMAX];
    sfor<CKMAX>([&](auto CI) {
      constexpr int c = decltype(CI)::value;
      const int k = KS * c + ls;
      cb[c] = (valid && k < a.C) ? a.cond[state * a.C + k] : 0.0f;
    });
    // forward of one step from the chain state xs: Dense_0, hidden layers, Dense_L -> O (= eps / velocity); every
    // layer's h = act(z) and act'(z) go to the stash rows rs_hd / rs_ad of this tile-step
    auto forward_step = [&](int j, const float (&xs)[NRX], __amdgpu_buffer_rsrc_t rs_hd, __amdgpu_buffer_rsrc_t rs_ad, acc_t& O) {
      // ================= forward of step j, recomputed =====================================================
      sfor<RING - 1>([&](auto NI) { l0load(NI); });
      bias_init(P, s_tb + j * H);
      if (a.state_bias != nullptr) {
        // single-evaluation VJP (frl_net_eval_train): the time and condition rows of Dense_0 arrive as one accumulator-init
        // row per chain (u[b] @ kernel_0[A:], frl_state_bias_kernel); the kernel itself contracts the x rows only
        const float* zrow = a.state_bias + (valid ? state : 0) * H;
        sfor<NG>([&](auto GI) {
          constexpr int g = decltype(GI)::value;
          sfor<NR>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            const f32x4 bv = *(const f32x4*)(zrow + g * GW + 4 * (G::rowr(r) + 4 * ls));
            sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[g * 4 + jj][r] += bv[jj]; });
          });
        });
      }
      sfor<N0>([&](auto NI) {
        constexpr int n = decltype(NI)::value;
        if constexpr (n + RING - 1 < N0) l0load(std::integral_constant<int, n + RING - 1>{});
        if (live(n)) {
          float bop;
          if constexpr (n < NRX) bop = xs[n]; else bop = cb[n - NRX];
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const f32x4 aw = r0[n % RING][g];
            sfor<4>([&](auto JI) { constexpr int q = decltype(JI)::value; P[g * 4 + q] = G::mfma(aw[q], bop, P[g * 4 + q]); });
          });
        }
      });

      // hidden Dense_l: consumes h_{l-1} = act(z_{l-1}) pair by pair, stashing h and act'
      for (int l = 1; l < a.L; ++l) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(a.wh, l - 1), 0, H * ROWB, 0x00020000);
        sfor<HRING - 1>([&](auto KI) { hload(rs, KI); });
        bias_init(Q, s_bh + (l - 1) * H);
        const int drow = (l - 1) * 32768;
        float bop, bn1, bn2 = 0.0f;
        {
          f32x2 h, d;
          act_grad2<ACT>(f32x2{P[0][0], P[0][1]}, h, d);
          st2(h, rs_hd, voff_d, drow); st2(d, rs_ad, voff_d, drow);
          bop = h[0]; bn1 = h[1];
        }
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr (k + HRING - 1 < NK) hload(rs, std::integral_constant<int, k + HRING - 1>{});
          if constexpr ((k & 1) == 1 && k + 2 < NK) {
            f32x2 h, d;
            act_grad2<ACT>(f32x2{P[(k + 1) / NR][(k + 1) % NR], P[(k + 2) / NR][(k + 2) % NR]}, h, d);
            st2(h, rs_hd, voff_d, drow + ((k + 1) / 2) * 512); st2(d, rs_ad, voff_d, drow + ((k + 1) / 2) * 512);
            bn1 = h[0]; bn2 = h[1];
          }
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const f32x4 aw = ring[k % HRING][g];
            sfor<4>([&](auto JI) { constexpr int q = decltype(JI)::value; Q[g * 4 + q] = G::mfma(aw[q], bop, Q[g * 4 + q]); });
          });
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, NG, 0);
          __builtin_amdgcn_sched_barrier(0);
          bop = bn1; bn1 = bn2;
        });
        sfor<NB>([&](auto BI) { constexpr int bb = decltype(BI)::value; P[bb] = Q[bb]; });
      }

      // Dense_L -> eps
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; O[r] = s_bl[G::rowr(r) + 4 * ls]; });
      {
        f32x4 rl[RINGL];
        sfor<RINGL - 1>([&](auto QI) { constexpr int q = decltype(QI)::value; rl[q] = ldw(rsl, voff_l, q * 1024); });
        const int drow = (a.L - 1) * 32768;
        float bop, bn1, bn2 = 0.0f;
        {
          f32x2 h, d;
          act_grad2<ACT>(f32x2{P[0][0], P[0][1]}, h, d);
          st2(h, rs_hd, voff_d, drow); st2(d, rs_ad, voff_d, drow);
          bop = h[0]; bn1 = h[1];
        }
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr ((k & 3) == 0 && k / 4 + RINGL - 1 < NLD)
            rl[(k / 4 + RINGL - 1) % RINGL] = ldw(rsl, voff_l, (k / 4 + RINGL - 1) * 1024);
          if constexpr ((k & 1) == 1 && k + 2 < NK) {
            f32x2 h, d;
            act_grad2<ACT>(f32x2{P[(k + 1) / NR][(k + 1) % NR], P[(k + 2) / NR][(k + 2) % NR]}, h, d);
            st2(h, rs_hd, voff_d, drow + ((k + 1) / 2) * 512); st2(d, rs_ad, voff_d, drow + ((k + 1) / 2) * 512);
            bn1 = h[0]; bn2 = h[1];
          }
          O = G::mfma(rl[(k / 4) % RINGL][k & 3], bop, O);
          bop = bn1; bn1 = bn2;
        });
      }
    };

    if constexpr (PHASE == 1) {
      // ===================== training forward: the sampler's own loop, stash kept =====================================
      float xs[NRX];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f;
      });
      for (int j = 0; j < a.T; ++j) {
        const long long ts = wt * a.T + j;
        const size_t srow = (size_t)ts * a.L * 8192;
        const __amdgpu_buffer_rsrc_t rs_hd = __builtin_amdgcn_make_buffer_rsrc((void*)(b.hd + srow), 0, a.L * 32768, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_ad = __builtin_amdgcn_make_buffer_rsrc((void*)(b.ad + srow), 0, a.L * 32768, 0x00020000);
        float zr[NRX];
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = G::rowr(r) + 4 * ls;
          const bool in = valid && ai < a.A;
          if (a.rng_impl != FRL_RNG_NONE && a.noise == nullptr)
            zr[r] = (in && s_coef[j * COEF_STRIDE + 5] != 0.0f) ? rng_step_normal(a.rng_keys, a.rng_impl, j, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai) : 0.0f;
          else
            zr[r] = (a.noise != nullptr && in) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
        acc_t O;
        forward_step(j, xs, rs_hd, rs_ad, O);
        const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
        const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
        const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
        sfor<NRX>([&](auto RI) {           // the update rules of frl_sampler_rs_kernel's epilogue (sampler_rs.hip), verbatim
          constexpr int r = decltype(RI)::value;
          const int ai = G::rowr(r) + 4 * ls;
          const bool in = valid && ai < a.A;
          const float eps = O[r], xt = xs[r];
          if (in && a.xt_hist != nullptr) {
            const long long o = ((long long)j * a.B + chain) * a.A + ai;
            a.xt_hist[o] = xt;
            a.eps_hist[o] = eps;
          }
          float xn;
          if (a.mode == MODE_DDPM_DDPM) {
            float x0h = c0 * (xt - c1 * eps);
            if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
            xn = c2 * (c3 * x0h + c4 * xt);
            xn += c5 * zr[r];
          } else if (a.mode == MODE_DDPM_DDIM) {
            xn = c0 * (xt - c1 * eps) + c2 * eps;
            if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
          } else if (a.mode == MODE_CDDPM_DDPM) {
            float x0h = (xt - c1 * eps) / c0;
            if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
            xn = c2 * x0h + c3 * xt;
            xn += c5 * zr[r];
          } else if (a.mode == MODE_CDDPM_DDIM) {
            float x0h = (xt - c1 * eps) / c0;
            if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
            xn = c2 * x0h + c3 * eps;
          } else if (a.mode == MODE_EVAL) {
            xn = eps;                                                 // one plain evaluation: the output itself
          } else {
            xn = xt + eps / c0;
            if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
          }
          xs[r] = in ? xn : 0.0f;
        });
      }
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        if (valid && ai < a.A) a.x0[chain * a.A + ai] = xs[r];
      });
    } else {
    // wide build with a cond_embedding: dL/dcfeat[chain] = sum over steps of W0[cond rows] dL/dz_0, four 32-feature blocks
    // that stay in accumulators over the whole step loop
    acc_t OC[WIDE ? 4 : 1];
    if constexpr (WIDE) sfor<4>([&](auto MI) { sfor<NR>([&](auto RI) { OC[decltype(MI)::value][decltype(RI)::value] = 0.0f; }); });
    const bool dcond = WIDE && b.gcf != nullptr;
    float gx[NRX];      // dL/dx_{j+1}, same register layout as the chain state
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = G::rowr(r) + 4 * ls;
      gx[r] = (valid && ai < a.A) ? b.g_x0[chain * a.A + ai] : 0.0f;
    });

    for (int jj = 0; jj < a.T; ++jj) {
      const int j = a.T - 1 - jj;
      const long long ts = wt * a.T + j;
      const size_t srow = (size_t)ts * a.L * 8192;          // floats; each layer = 8192 floats = 64 pair rows x 512 B
      const __amdgpu_buffer_rsrc_t rs_hd = __builtin_amdgcn_make_buffer_rsrc((void*)(b.hd + srow), 0, a.L * 32768, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_ad = __builtin_amdgcn_make_buffer_rsrc((void*)(b.ad + srow), 0, a.L * 32768, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_gd = __builtin_amdgcn_make_buffer_rsrc((void*)(b.gd + srow), 0, a.L * 32768, 0x00020000);

      float xs[NRX];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        xs[r] = (valid && ai < a.A) ? b.xt_hist[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
      });

      acc_t O;
      if constexpr (PHASE == 0) {
        forward_step(j, xs, rs_hd, rs_ad, O);             // the caller kept only xt_hist: recompute the step
      } else {
        sfor<NRX>([&](auto RI) {                           // the training forward kept eps_hist and the stash
          constexpr int r = decltype(RI)::value;
          const int ai = G::rowr(r) + 4 * ls;
          O[r] = (valid && ai < a.A && b.eps_hist != nullptr) ? b.eps_hist[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
      }

      // ================= VJP of the update: dL/deps and the direct part of dL/dx_j ============================
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
        } else if (a.mode == MODE_EVAL) {
          g_e = gn; g_x = 0.0f;                                     // the cotangent of the evaluation's output, nothing direct
        } else {
          const float xn = xt + eps / c0;
          const float m = (!a.clip || (xn >= a.x_min && xn <= a.x_max)) ? 1.0f : 0.0f;
          g_e = gn * m / c0; g_x = gn * m;
        }
        const int ai = G::rowr(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        ge[r] = in ? g_e : 0.0f;
        gxd[r] = in ? g_x : 0.0f;
        b.ge[(size_t)ts * b.ge_stride + r * 64 + lane] = ge[r];
      });

      // ================= backward through the net ==================================================================
      // dL/dh_{L-1} = W_L dL/deps  (k-steps over the action rows, like Dense_0's x part, kernel W_L^T)
      zero_init(P);
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        if (r < a.AR) {
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const f32x4 aw = ldw(rslt, voff_x + g * (GW * 4), G::rowr(r) * ROWB);
            sfor<4>([&](auto JI) { constexpr int q = decltype(JI)::value; P[g * 4 + q] = G::mfma(aw[q], ge[r], P[g * 4 + q]); });
          });
        }
      });
      // hidden layers, downwards: dL/dh_{l-1} = W_l (dL/dh_l * act'(z_l))
      for (int l = a.L - 1; l >= 1; --l) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(b.wt, l - 1), 0, H * ROWB, 0x00020000);
        sfor<HRING - 1>([&](auto KI) { hload(rs, KI); });
        zero_init(Q);
        const int drow = l * 32768;
        const float* blk = b.ad + srow + (size_t)l * 8192;             // this layer's act' block; the next one is layer l - 1
        auto adload = [&](int rho) -> f32x2 {
          if constexpr (PHASE == 2) return *(const f32x2*)(s_adw + rho * 512 + lane * 8);
          else return ld2(rs_ad, voff_d, drow + rho * 512);
        };
        f32x2 ar[ARD];
        sfor<ARD>([&](auto QI) { constexpr int q = decltype(QI)::value; ar[q] = adload(q); });
        float bop, bn1, bn2 = 0.0f;
        {
          const f32x2 gz = f32x2{P[0][0], P[0][1]} * ar[0];
          st2(gz, rs_gd, voff_d, drow);
          bop = gz[0]; bn1 = gz[1];
        }
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr (PHASE == 2 && k == 0) issue_half(blk, 1);
          if constexpr (PHASE == 2 && k == NK / 2) issue_half(blk - 8192, 0);
          if constexpr (k + HRING - 1 < NK) hload(rs, std::integral_constant<int, k + HRING - 1>{});
          if constexpr ((k & 1) == 1 && k + 2 < NK) {
            constexpr int rho = (k + 1) / 2;
            if constexpr (rho + ARD - 1 < NK / 2) ar[(rho + ARD - 1) % ARD] = adload(rho + ARD - 1);
            const f32x2 gz = f32x2{P[(k + 1) / NR][(k + 1) % NR], P[(k + 2) / NR][(k + 2) % NR]} * ar[rho % ARD];
            st2(gz, rs_gd, voff_d, drow + rho * 512);
            bn1 = gz[0]; bn2 = gz[1];
          }
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const f32x4 aw = ring[k % HRING][g];
            sfor<4>([&](auto JI) { constexpr int q = decltype(JI)::value; Q[g * 4 + q] = G::mfma(aw[q], bop, Q[g * 4 + q]); });
          });
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, NG, 0);
          __builtin_amdgcn_sched_barrier(0);
          bop = bn1; bn1 = bn2;
        });
        sfor<NB>([&](auto BI) { constexpr int bb = decltype(BI)::value; P[bb] = Q[bb]; });
      }
      // dL/dx_j (net part) = W0[x rows] (dL/dh_0 * act'(z_0)); also stashes dL/dz_0
      acc_t OX;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; OX[r] = 0.0f; });
      {
        f32x4 rl[RINGL];
        sfor<RINGL - 1>([&](auto QI) { constexpr int q = decltype(QI)::value; rl[q] = ldw(rsxq, voff_l, q * 1024); });
        f32x4 rc[WIDE ? 4 : 1][2];       // cond-row operands: one 16-byte load = 4 k-steps, requested 4 k-steps ahead
        const __amdgpu_buffer_rsrc_t rscq = __builtin_amdgcn_make_buffer_rsrc((void*)(WIDE && b.ws_w0cq != nullptr ? b.ws_w0cq : b.ws_w0xq), 0, (WIDE ? 4 : 1) * H * 64 * 4, 0x00020000);   // image m at m * H * 64 floats (the first NK * 64 of each are read)
        if constexpr (WIDE) { if (dcond) sfor<4>([&](auto MI) { constexpr int m = decltype(MI)::value; rc[m][0] = ldw(rscq, voff_l + m * (H * 64 * 4), 0); }); }
        // layer 0's act' block; after it the sweep moves to (step j - 1, layer L - 1), or to this wave's next tile
        const float* blk0 = b.ad + srow;
        const long long wt_n = wt + (long long)gridDim.x * nw;
        const float* nxt = (jj + 1 < a.T) ? b.ad + (srow - (size_t)a.L * 8192) + (size_t)(a.L - 1) * 8192
                           : (wt_n < n_wt ? b.ad + ((size_t)(wt_n * a.T + a.T - 1) * a.L + (a.L - 1)) * 8192 : nullptr);
        auto adload = [&](int rho) -> f32x2 {
          if constexpr (PHASE == 2) return *(const f32x2*)(s_adw + rho * 512 + lane * 8);
          else return ld2(rs_ad, voff_d, rho * 512);
        };
        f32x2 ar[ARD];
        sfor<ARD>([&](auto QI) { constexpr int q = decltype(QI)::value; ar[q] = adload(q); });
        float bop, bn1, bn2 = 0.0f;
        {
          const f32x2 gz = f32x2{P[0][0], P[0][1]} * ar[0];
          st2(gz, rs_gd, voff_d, 0);
          bop = gz[0]; bn1 = gz[1];
        }
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr (PHASE == 2 && k == 0) issue_half(blk0, 1);
          if constexpr (PHASE == 2 && k == NK / 2) { if (nxt != nullptr) issue_half(nxt, 0); }
          if constexpr ((k & 3) == 0 && k / 4 + RINGL - 1 < NLD)
            rl[(k / 4 + RINGL - 1) % RINGL] = ldw(rsxq, voff_l, (k / 4 + RINGL - 1) * 1024);
          if constexpr ((k & 1) == 1 && k + 2 < NK) {
            constexpr int rho = (k + 1) / 2;
            if constexpr (rho + ARD - 1 < NK / 2) ar[(rho + ARD - 1) % ARD] = adload(rho + ARD - 1);
            const f32x2 gz = f32x2{P[(k + 1) / NR][(k + 1) % NR], P[(k + 2) / NR][(k + 2) % NR]} * ar[rho % ARD];
            st2(gz, rs_gd, voff_d, rho * 512);
            bn1 = gz[0]; bn2 = gz[1];
          }
          OX = G::mfma(rl[(k / 4) % RINGL][k & 3], bop, OX);
          if constexpr (WIDE) {
            if (dcond) {
              if constexpr ((k & 3) == 0 && k / 4 + 1 < NLD)
                sfor<4>([&](auto MI) { constexpr int m = decltype(MI)::value; rc[m][(k / 4 + 1) & 1] = ldw(rscq, voff_l + m * (H * 64 * 4), (k / 4 + 1) * 1024); });
              sfor<4>([&](auto MI) { constexpr int m = decltype(MI)::value; OC[m] = G::mfma(rc[m][(k / 4) & 1][k & 3], bop, OC[m]); });
            }
          }
          bop = bn1; bn1 = bn2;
        });
      }
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        gx[r] = (valid && ai < a.A) ? gxd[r] + OX[r] : 0.0f;
      });
    }
    if (b.g_xT != nullptr) {
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        if (valid && ai < a.A) b.g_xT[chain * a.A + ai] = gx[r];
      });
    }
    if constexpr (WIDE) {
      if (dcond && valid) {        // accumulator register r of block m, lane group s <-> cond feature 32 m + ROWR(r) + 4 s
        sfor<4>([&](auto MI) { constexpr int m = decltype(MI)::value; sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value;
          b.gcf[chain * COND_EMBED_DIM + 32 * m + G::rowr(r) + 4 * ls] = OC[m][r]; }); });
      }
    }
    }   // PHASE != 1
  }
}

#if FRL_BWD_PHASE == 0
// ---- weight gradients: split-K GEMM over the stash -------------------------------------------------------------
// One workgroup (4 waves, 2 x 2) owns tile-steps ts = blockIdx, blockIdx + grid, ... and, layer after layer, the whole
// dW_l tile in accumulators.  Per tile-step the two operands (X = inputs of the layer, G = dL/dz of the layer, both
// [features][32 chains]) are staged transposed into LDS ([feature][33]: conflict-free for the write and for the
// feature-on-lane read) and contracted over the 32 chains with 16 k-steps of v_mfma_f32_32x32x2_f32.
__device__ __forceinline__ int stash_feature(int row2, int e, int lane) {
  // stash element (pair row rho, lane, e): k = 2 rho + e = block*16 + r; feature = 128 g + j + 4 (ROWR(r) + 4 s)
  const int k = 2 * row2 + e, blk = k >> 4, r = k & 15;
  return 128 * (blk >> 2) + (blk & 3) + 4 * ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
}

// Pipeline: the stash blocks of tile-step u + 1 are fetched into registers while tile-step u is contracted, staged into the
// other half of a double-buffered LDS tile pair, one barrier per tile-step.  A workgroup owns a contiguous STEP-MAJOR range
// of tile-steps (u = j * n_wt + wt), so the per-step column sums that make dL/dtb stay in a register between flushes.
template <int KIND, int NM0 = 1>   // 0: Dense_0 (X = [x_j ; cond], NM0 32-row blocks)  1: hidden  2: Dense_L (G = dL/deps, one block)
__device__ __forceinline__ void dw_layer_f32(const BArgs& b, int lam, float* smem, long long u0, long long u1, long long n_wt) {
  constexpr int H = 256, LDP = 33, TILE = H * LDP;
  const KArgs& a = b.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;       // 8 waves: 4 (X blocks 2 wm, 2 wm + 1) x 2 (G blocks 4 wn .. 4 wn + 3)
  const int A = a.A, C = b.dw_C, T = a.T, L = a.L;
  constexpr int NM = (KIND == 1) ? 2 : (KIND == 0 ? NM0 : 1), NN = (KIND == 1) ? 4 : 1;
  const int mb0 = (KIND == 1) ? 2 * wm : (KIND == 2 ? wave : 0);
  const int nb0 = (KIND == 1) ? 4 * wn : (KIND == 0 ? wave : 0);

  f32x16 acc[NM][NN];
  sfor<NM>([&](auto MI) { sfor<NN>([&](auto NI) { sfor<16>([&](auto RI) {
    acc[decltype(MI)::value][decltype(NI)::value][decltype(RI)::value] = 0.0f; }); }); });
  float bsum = 0.0f;                                   // column sum of G for feature `tid` (bias gradient)
  int jcur = -1;
  const DwPart dp = dw_part_layout(L, T, b.k0_rows);
  float* slot = b.part + (size_t)blockIdx.x * dp.stride;   // this workgroup's partial sums (deterministic second stage adds the slots)
  if constexpr (KIND == 0) {
    // per-step dL/dtb rows: a workgroup's tile-steps cover a few consecutive steps; the other rows of its slot are zero
    if (tid < 256) for (int j = 0; j < T; ++j) slot[dp.tb + (size_t)j * H + tid] = 0.0f;
  }

  if (KIND == 2) {     // dL/deps fills rows 0..7 (+4 s) of its block: the other rows stay zero in both buffers
    for (int idx = tid; idx < 2 * 32 * LDP; idx += 512) smem[(idx / (32 * LDP)) * 2 * TILE + TILE + idx % (32 * LDP)] = 0.0f;
    __syncthreads();
  }

  f32x2 xr[8], gr[8];       // staging registers: 64 pair rows / 8 waves
  float xv[2 * NM0], gv[2] = {0.0f, 0.0f};
  const int nge = b.ge_stride >> 6;      // dL/deps registers per lane in the stash (4 | 12)
  auto fetch = [&](long long u) {
    const int j = (int)(u / n_wt);
    const long long wt = u - (long long)j * n_wt, ts = wt * T + j;
    const size_t srow = (size_t)ts * L * 8192;
    if constexpr (KIND == 0) {
      sfor<2 * NM0>([&](auto EI) {
        constexpr int k = decltype(EI)::value;
        const int idx = tid + 512 * k, row = idx >> 5, c = idx & 31;
        const long long chain = wt * 32 + c;
        float v = 0.0f;
        if (chain < a.B) {
          if (row < A) v = b.xt_hist[((long long)j * a.B + chain) * A + row];
          else if (row < A + C) v = b.dw_cond[((a.particles == 1) ? chain : chain / a.particles) * C + (row - A)];
        }
        xv[k] = v;
      });
    } else {
      const float* src = b.hd + srow + (size_t)(lam - 1) * 8192;
      sfor<8>([&](auto QI) { constexpr int q = decltype(QI)::value; xr[q] = *(const f32x2*)(src + ((size_t)(wave + 8 * q) * 64 + lane) * 2); });
    }
    if constexpr (KIND == 2) {
      if (wave < nge) gv[0] = b.ge[(size_t)ts * b.ge_stride + wave * 64 + lane];
      if (wave + 8 < nge) gv[1] = b.ge[(size_t)ts * b.ge_stride + (wave + 8) * 64 + lane];
    } else {
      const float* src = b.gd + srow + (size_t)lam * 8192;
      sfor<8>([&](auto QI) { constexpr int q = decltype(QI)::value; gr[q] = *(const f32x2*)(src + ((size_t)(wave + 8 * q) * 64 + lane) * 2); });
    }
  };
  auto stage = [&](float* buf) {
    float* sX = buf;
    float* sG = buf + TILE;
    if constexpr (KIND == 0) {
      sfor<2 * NM0>([&](auto EI) { constexpr int k = decltype(EI)::value; const int idx = tid + 512 * k; sX[(idx >> 5) * LDP + (idx & 31)] = xv[k]; });
    } else {
      sfor<8>([&](auto QI) {
        constexpr int q = decltype(QI)::value;
        const int row2 = wave + 8 * q;
        sX[stash_feature(row2, 0, lane) * LDP + li] = xr[q][0];
        sX[stash_feature(row2, 1, lane) * LDP + li] = xr[q][1];
      });
    }
    if constexpr (KIND == 2) {
      if (wave < nge) sG[((wave & 3) + 8 * (wave >> 2) + 4 * lh) * LDP + li] = gv[0];   // register r = wave: action row ROWR(r) + 4 s
      if (wave + 8 < nge) sG[((wave & 3) + 8 * ((wave + 8) >> 2) + 4 * lh) * LDP + li] = gv[1];
    } else {
      sfor<8>([&](auto QI) {
        constexpr int q = decltype(QI)::value;
        const int row2 = wave + 8 * q;
        sG[stash_feature(row2, 0, lane) * LDP + li] = gr[q][0];
        sG[stash_feature(row2, 1, lane) * LDP + li] = gr[q][1];
      });
    }
  };

  if (u0 < u1) {
    fetch(u0);
    stage(smem);
    if (u0 + 1 < u1) fetch(u0 + 1);
  }
  __syncthreads();
  for (long long u = u0; u < u1; ++u) {
    const float* sX = smem + (int)((u - u0) & 1) * 2 * TILE;
    const float* sG = sX + TILE;
    // ---- bias gradient: column sums of G (per-step table for Dense_0: the time branch needs dL/dtb[j]) ----
    if (tid < (KIND == 2 ? 32 : 256)) {
      if constexpr (KIND == 0) {
        const int j = (int)(u / n_wt);
        if (j != jcur) {
          if (jcur >= 0) slot[dp.tb + (size_t)jcur * H + tid] = bsum;    // same thread zero-filled this column above
          bsum = 0.0f; jcur = j;
        }
      }
      float s = 0.0f;
#pragma unroll 8
      for (int c = 0; c < 32; ++c) s += sG[tid * LDP + c];
      bsum += s;
    }
    // ---- contract over the 32 chains ----
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      float av[NM], bv[NN];
      sfor<NM>([&](auto MI) { constexpr int m = decltype(MI)::value; av[m] = sX[(32 * (mb0 + m) + li) * LDP + 2 * kk + lh]; });
      sfor<NN>([&](auto NI) { constexpr int n = decltype(NI)::value; bv[n] = sG[(32 * (nb0 + n) + li) * LDP + 2 * kk + lh]; });
      sfor<NM>([&](auto MI) { constexpr int m = decltype(MI)::value; sfor<NN>([&](auto NI) { constexpr int n = decltype(NI)::value;
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[n], acc[m][n], 0, 0, 0); }); });
    }
    // ---- next tile-step into the other buffer ----
    if (u + 1 < u1) {
      stage(smem + (int)(((u - u0) & 1) ^ 1) * 2 * TILE);
      if (u + 2 < u1) fetch(u + 2);
    }
    __syncthreads();
  }

  // ---- flush: this workgroup's partial tile into its slot (plain stores, 128-B row segments; no atomics) ----
  //   Dense_0: [32 staged rows x | cond][256]   hidden: [256][256]   Dense_L: [256][32]
  float* pk = slot + (KIND == 0 ? dp.k0 : KIND == 1 ? dp.kh + (size_t)(lam - 1) * 65536 : dp.kl);
  constexpr int PCOL = (KIND == 2) ? 32 : H;
  sfor<NM>([&](auto MI) {
    constexpr int m = decltype(MI)::value;
    sfor<NN>([&](auto NI) {
      constexpr int n = decltype(NI)::value;
      sfor<16>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int in = 32 * (mb0 + m) + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int out = 32 * (nb0 + n) + li;
        if (KIND != 0 || in < b.k0_rows) pk[(size_t)in * PCOL + out] = acc[m][n][r];    // NM0 blocks may exceed the slot's k0_rows
      });
    });
  });
  if constexpr (KIND == 0) {
    if (tid < 256 && jcur >= 0) slot[dp.tb + (size_t)jcur * H + tid] = bsum;
  } else {
    if (tid < 256) slot[dp.bias + (size_t)(lam - 1) * 256 + tid] = (KIND == 2 && tid >= 32) ? 0.0f : bsum;
  }
  __syncthreads();
}

// ---- second stage: add the slots in slot order, write every leaf once ----------------------------------------------
// One thread per output element; slot w of every thread's element is one coalesced row of loads.  Four interleaved
// partial sums combined as a tree: a fixed order, hence bitwise reproducible, and shorter rounding chains than one sum.
__global__ void __launch_bounds__(256) frl_bwd_reduce_kernel(const BArgs b) {
  const KArgs& a = b.k;
  const int A = a.A, C = b.dw_C, T = a.T, L = a.L, Dp = b.Dp, H = 256;
  const DwPart dp = dw_part_layout(L, T, b.k0_rows);
  const long long n0 = (long long)(A + Dp + C) * H, nh = (long long)(L - 1) * 65536, nl = (long long)H * A;
  const long long nb = (long long)(L - 1) * H + A, ntb = (long long)T * H;
  const long long total = n0 + nh + nl + nb + ntb;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    float* dst;
    long long src = -1;          // float offset inside a slot; -1: a structural zero (time rows of Dense_0)
    long long i = e;
    if (i < n0) {                // Dense_0 kernel rows x | time | cond  <-  staged rows x | cond
      const int row = (int)(i / H), col = (int)(i - (long long)row * H);
      dst = b.grad_kernel[0] + i;
      if (row < A) src = (long long)dp.k0 + (long long)row * H + col;
      else if (row >= A + Dp) src = (long long)dp.k0 + (long long)(row - Dp) * H + col;
    } else if ((i -= n0) < nh) {
      const int l = 1 + (int)(i >> 16);
      dst = b.grad_kernel[l] + (i & 65535);
      src = (long long)dp.kh + i;
    } else if ((i -= nh) < nl) {
      const int row = (int)(i / A), col = (int)(i - (long long)row * A);
      dst = b.grad_kernel[L] + i;
      src = (long long)dp.kl + (long long)row * 32 + col;
    } else if ((i -= nl) < nb) {
      const int l = 1 + (int)(i / H), col = (int)(i - (long long)(l - 1) * H);     // l = L: the A entries of Dense_L's bias
      dst = b.grad_bias[l] + col;
      src = (long long)dp.bias + (long long)(l - 1) * 256 + col;
    } else {
      i -= nb;
      dst = b.grad_tb + i;
      src = (long long)dp.tb + i;
    }
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (src >= 0) {
      const float* p = b.part + src;
      int w = 0;
      for (; w + 4 <= b.n_part; w += 4) {
        s0 += p[(size_t)w * dp.stride]; s1 += p[(size_t)(w + 1) * dp.stride];
        s2 += p[(size_t)(w + 2) * dp.stride]; s3 += p[(size_t)(w + 3) * dp.stride];
      }
      if (w < b.n_part) s0 += p[(size_t)w * dp.stride];
      if (w + 1 < b.n_part) s1 += p[(size_t)(w + 1) * dp.stride];
      if (w + 2 < b.n_part) s2 += p[(size_t)(w + 2) * dp.stride];
    }
    *dst = (s0 + s1) + (s2 + s3);
  }
}

int launch_bwd_reduce(const BArgs& b, hipStream_t stream) {
  const KArgs& a = b.k;
  const long long total = (long long)(a.A + b.Dp + b.dw_C) * 256 + (long long)(a.L - 1) * 65536 + 256LL * a.A + (long long)(a.L - 1) * 256 + a.A + (long long)a.T * 256;
  const int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(frl_bwd_reduce_kernel, dim3(grid), dim3(256), 0, stream, b);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("bwd reduce launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
frl_bwd_dw_kernel(const BArgs b) {
  extern __shared__ __attribute__((aligned(16))) float smem_dw[];
  const KArgs& a = b.k;
  const long long n_wt = (a.B + 31) / 32;
  const long long U = n_wt * a.T;                         // tile-steps, step-major: u = j * n_wt + wt
  const long long u0 = U * blockIdx.x / gridDim.x, u1 = U * (blockIdx.x + 1) / gridDim.x;
  if (b.k0_rows <= 32) dw_layer_f32<0, 1>(b, 0, smem_dw, u0, u1, n_wt);
  else if (b.k0_rows <= 160) dw_layer_f32<0, 5>(b, 0, smem_dw, u0, u1, n_wt);   // DACER / QVPO shapes: x (<= 24) | cond (<= 128) rows = up to 5 blocks
  else dw_layer_f32<0, 7>(b, 0, smem_dw, u0, u1, n_wt);      // single-evaluation VJP: x | tfeat (64) | cfeat (128) = up to 7 blocks
  for (int lam = 1; lam < a.L; ++lam) dw_layer_f32<1>(b, lam, smem_dw, u0, u1, n_wt);
  dw_layer_f32<2>(b, a.L, smem_dw, u0, u1, n_wt);
}

// dL/du of the single-evaluation VJP: du[b][k] = sum_f W0[A + k][f] dL/dz_0[b][f] for the n_rows rows of Dense_0 that the
// forward hoisted into the per-sample init row (u = [tfeat | cond]).  One workgroup per wave-tile (T = 1): the tile's
// dL/dz_0 stash block goes through LDS transposed ([feature][33]), then 256 threads share the 32 x n_rows dot products.
__global__ void __launch_bounds__(256) frl_eval_du_kernel(const BArgs b, const float* __restrict__ w0_rows, int n_rows, float* __restrict__ du) {
  __shared__ float sG[256 * 33];
  const KArgs& a = b.k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31;
  const long long wt = blockIdx.x;
  const float* src = b.gd + (size_t)wt * a.T * a.L * 8192;          // layer 0 of tile-step (wt, j = 0)
  for (int q = 0; q < 16; ++q) {
    const int row2 = wave + 4 * q;
    const f32x2 v = *(const f32x2*)(src + ((size_t)row2 * 64 + lane) * 2);
    sG[stash_feature(row2, 0, lane) * 33 + li] = v[0];
    sG[stash_feature(row2, 1, lane) * 33 + li] = v[1];
  }
  __syncthreads();
  for (int idx = tid; idx < 32 * n_rows; idx += 256) {
    const int c = idx & 31, k = idx >> 5;
    const long long chain = wt * 32 + c;
    if (chain >= a.B) continue;
    const float* wr = w0_rows + (size_t)k * 256;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int f = 0; f < 256; f += 4) {
      s0 = fmaf(wr[f], sG[f * 33 + c], s0); s1 = fmaf(wr[f + 1], sG[(f + 1) * 33 + c], s1);
      s2 = fmaf(wr[f + 2], sG[(f + 2) * 33 + c], s2); s3 = fmaf(wr[f + 3], sG[(f + 3) * 33 + c], s3);
    }
    du[chain * n_rows + k] = (s0 + s1) + (s2 + s3);
  }
}

int launch_eval_du(const BArgs& b, const float* w0_rows, int n_rows, float* du, hipStream_t stream) {
  const long long n_wt = (b.k.B + 31) / 32;
  if (n_wt <= 0 || n_rows <= 0) return FRL_OK;
  hipLaunchKernelGGL(frl_eval_du_kernel, dim3((unsigned)n_wt), dim3(256), 0, stream, b, w0_rows, n_rows, du);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("eval du launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  return FRL_OK;
}
#endif  // FRL_BWD_PHASE == 0

// ---- launcher ---------------------------------------------------------------------------------------------------
template <int ACT, int NRX, int CKMAX>
static int launch_sweep_inst(const BArgs& b, int grid, int block, size_t lds, hipStream_t stream) {
  auto kern = frl_bwd_sweep_kernel<ACT, NRX, CKMAX, FRL_BWD_PHASE>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, b);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("bwd sweep launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_bwd_sweep_kernel<act %d, NRX %d, CKMAX %d, PHASE %d> [grid %d x %d waves]", ACT, NRX, CKMAX, FRL_BWD_PHASE, grid, block / 64);
  return FRL_OK;
}

// narrow operands (x_dim <= 8, <= 24 condition rows: DQL / DAC / FQL) or the wide build (x_dim <= 24, <= 128 condition rows:
// DACER / QVPO with their cond_embedding) — the wide build exists for the training forward and the stash backward only
template <int ACT>
static int launch_sweep(const BArgs& b, int grid, int block, size_t lds, hipStream_t stream) {
  if (b.k.AR <= 4 && b.k.CK <= 12) return launch_sweep_inst<ACT, 4, 12>(b, grid, block, lds, stream);
#if FRL_BWD_PHASE != 0
  if (b.k.AR <= 12 && b.k.CK <= 64) return launch_sweep_inst<ACT, 12, 64>(b, grid, block, lds, stream);
#endif
  set_error("VJP: x_dim %d / cond_dim %d: this path needs the training forward (frl_sampler_fwd_train) and its stash", b.k.A, b.k.C);
  return FRL_EUNSUPPORTED;
}

template <>
int launch_bwd_sweep_phase<FRL_BWD_PHASE>(const FrlDesc& d, const BArgs& b, hipStream_t stream) {
  const KArgs& a = b.k;
  const int ncu = device_cus();
  const long long n_wt = (a.B + 31) / 32;
  long long per_cu = (n_wt + ncu - 1) / ncu;
  int nw = (int)(per_cu < 1 ? 1 : (per_cu > MAX_WAVES ? MAX_WAVES : per_cu));
  if (nw == 3) nw = 4;
  const size_t tables = sizeof(float) * ((size_t)a.T * 256 + (size_t)(a.L - 1) * 256 + LAST_PAD + (size_t)a.T * COEF_STRIDE);
  if (FRL_BWD_PHASE == 2)       // + the per-wave act' staging buffer (32 KB each): fewer waves per workgroup when T is large
    while (nw > 1 && tables + (size_t)nw * 32768 > 160 * 1024) nw >>= 1;
  const long long nwg = (n_wt + nw - 1) / nw;
  const int grid = (int)(nwg < ncu ? nwg : ncu);
  const size_t lds = tables + (FRL_BWD_PHASE == 2 ? (size_t)nw * 32768 : 0);
  if (lds > 160 * 1024) { set_error("bwd: steps=%d needs %zu B of LDS", a.T, lds); return FRL_EUNSUPPORTED; }
  switch (d.act) {
    case FRL_ACT_MISH: return launch_sweep<FRL_ACT_MISH>(b, grid, 64 * nw, lds, stream);
    case FRL_ACT_SILU: return launch_sweep<FRL_ACT_SILU>(b, grid, 64 * nw, lds, stream);
    case FRL_ACT_RELU: return launch_sweep<FRL_ACT_RELU>(b, grid, 64 * nw, lds, stream);
    case FRL_ACT_GELU_TANH: return launch_sweep<FRL_ACT_GELU_TANH>(b, grid, 64 * nw, lds, stream);
    case FRL_ACT_TANH: return launch_sweep<FRL_ACT_TANH>(b, grid, 64 * nw, lds, stream);
    case FRL_ACT_ELU: return launch_sweep<FRL_ACT_ELU>(b, grid, 64 * nw, lds, stream);
    default: set_error("bwd: activation %d invalid", d.act); return FRL_EINVAL;
  }
}

#if FRL_BWD_PHASE == 0
// phase 1: the training forward alone; phase 0 / 2: the sweep (with / without recompute), the dW kernel and the reduce
int launch_sampler_bwd(const FrlDesc& d, const BArgs& b, int phase, hipStream_t stream) {
  if (phase == 1) return launch_bwd_sweep_phase<1>(d, b, stream);
  int rc = (phase == 2) ? (bwd_sweep_lat_supports(b) ? launch_bwd_sweep_lat(b, stream) : launch_bwd_sweep_phase<2>(d, b, stream))
                        : launch_bwd_sweep_phase<0>(d, b, stream);
  if (rc != FRL_OK) return rc;
  const long long g2 = b.n_part;                        // one partial slot per workgroup (BwdPlan.n_part)
  const size_t lds2 = sizeof(float) * 4 * 256 * 33;     // two (X, G) tile pairs
  hipError_t e = hipFuncSetAttribute((const void*)frl_bwd_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(frl_bwd_dw_kernel, dim3((int)g2), dim3(512), lds2, stream, b);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("bwd dW launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_bwd_dw_kernel [grid %d] + frl_bwd_reduce_kernel", (int)g2);
  return launch_bwd_reduce(b, stream);
}
#endif

}  // namespace frl
