// The fused T-step sampler for RESIDUAL backbones, exact-f32 MFMA (v_mfma_f32_32x32x2_f32), weights streamed L2 -> registers.
//
// Reference networks (the noise / velocity predictor of a DDPMBackbone / FlowBackbone, ddpm.py:83-95):
//   ResidualMLP  flowrl/module/mlp.py:37-98     x = Dense_0(in); per block: x = x + Dense_1(act(Dense_0([LN](x)))); out = Dense_1(act(x))
//                (DAC / BDPO with diffusion.resnet: agent/offline/dac.py:200-207, bdpo/bdpo.py:325-332: mish, layer_norm=True)
//   Simba        flowrl/module/simba.py:10-63   x = Dense_0(in); per block: x = x + Dense_1(act(Dense_0(LN(x)))); out = Dense_1(LN(x))
//                (backbone_cls: simba in dacer / sdac / qvpo / dpmd / qsm / dppo / genpo)
// Block width H = 256, expansion 4 H = 1024 (multiplier 4, the only value any reference config uses).
//
// Same transposed, register-resident formulation as sampler_rs.hip (a chain lives on a lane, accumulator register r of a
// finished tile IS the B operand of the k-step that contracts its two feature rows).  A residual block never materialises
// its 1024-wide hidden tile: the expansion is produced 128 features (one 16-byte load group = 4 MFMA blocks) at a time,
//     Y[4]   = b1[group] + W1[:, group]^T Xn          128 k-steps x 4 MFMAs        (Xn = LayerNorm(x) or x)
//     x[8]  += W2[group, :]^T act(Y)                    64 k-steps x 8 MFMAs        (x pre-loaded with + b2: residual + bias)
// for the 8 groups in turn, so the live state is x (128 registers) + Xn (128) + Y (64) + operand rings: one wave per SIMD
// with the 512-register budget, as the plain kernel.  Both GEMMs address the Flax (in, out) kernels IN PLACE with the hidden
// layer's row map (row = GW g + j + 4 ROWR(r) + 16 s).  One operand ring serves both: the tail of each GEMM prefetches the
// head of the next (next group / next block / block 0 of the next evaluation), so no phase starts cold.
// Condition rows of Dense_0 are always hoisted per state (frl_state_bias_kernel): the register budget goes to the blocks.
//
// Latency build (SPLIT = 4, round 3): acting with a Simba / ResidualMLP policy at batch 1 .. a few thousand chains leaves most
// SIMDs idle in the one-wave-per-tile build, and a block is a serial chain of 8192 64-cycle MFMAs per wave (0.22 ms per
// block and step).  The eight expansion groups of a block are independent given LayerNorm(x) and their contributions to x
// add up, so the FOUR waves of a workgroup share one tile: every wave holds the whole x (Dense_0, the LayerNorms, the head
// and the update rule are computed redundantly, bit-identically, in all four), takes two of the eight groups, and the
// partial sums (wave 0's includes x + b2) are exchanged through LDS in two halves and added in wave order by everyone.
// Fixed order: deterministic; the rounding differs from the throughput build's (one wave adds the groups in sequence).
#include "frl_device.h"
#include "frl_rng.h"

namespace frl {

__device__ __forceinline__ const float* pick_ptr8(const float* const (&arr)[FRL_MAX_HIDDEN], int idx) {
  const float* p = arr[0];
#pragma unroll
  for (int k = 1; k < FRL_MAX_HIDDEN; ++k) p = (idx == k) ? arr[k] : p;
  return p;
}

template <int ACT, int NRX, int SPLIT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_res_kernel(const KArgs a, const ResArgs ra) {
  typedef Geo<32> G;
  typedef G::acc_t acc_t;
  constexpr int TR = 32, H = 256, M = 4 * H;
  constexpr int NR = G::NR, RING = G::RING;
  constexpr int NB = H / TR;        // 8 feature blocks of x
  constexpr int NG = NB / 4;        // 2 load groups across x's features
  constexpr int GW = 4 * TR;        // 128 features per load group
  constexpr int NGM = M / GW;       // 8 groups of the expansion
  constexpr int NK = NB * NR;       // 128 k-steps contracting x
  constexpr int NK2 = 4 * NR;       // 64 k-steps contracting one group of the expansion
  constexpr int ROWB = H * 4;       // bytes per row of an (., 256) kernel
  constexpr int ROWB1 = M * 4;      // bytes per row of the (256, 1024) expansion kernel
  constexpr int RINGL = 8;
  constexpr int BLK = M + H + 2 * H;   // LDS floats per block: b1 | b2 | LN scale | LN bias
  static_assert(NK % RING == 0 && NK2 % RING == 0, "geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int li = lane % TR, ls = lane / TR;
  const int nblk = ra.n_blocks;

  float* s_tb = (float*)smem;                         // [NE][H] Dense_0 bias table (time slice folded in)
  float* s_bl = s_tb + a.NE * H;                      // [32]    padded last-layer bias
  float* s_coef = s_bl + LAST_PAD;                    // [NE][8]
  float* s_blk = s_coef + a.NE * COEF_STRIDE;         // [nblk][BLK]
  float* s_fln = s_blk + nblk * BLK;                  // [2][H] final LayerNorm (Simba)
  float* s_red = s_fln + 2 * H;                       // SPLIT > 1: [4 waves][64 registers][64 lanes] partial sums of half of x
  constexpr int GPW = NGM / SPLIT;                    // expansion groups per wave
  const int g_lo = (SPLIT > 1) ? wave * GPW : 0, g_hi = g_lo + GPW;
  for (int idx = tid; idx < a.NE * H; idx += blockDim.x) s_tb[idx] = a.ws_tb[idx];
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.NE * COEF_STRIDE; idx += blockDim.x) s_coef[idx] = a.ws_coef[idx];
  for (int i = 0; i < nblk; ++i) {
    const float* b1 = pick_ptr8(ra.b1, i);
    const float* b2 = pick_ptr8(ra.b2, i);
    float* sb = s_blk + i * BLK;
    for (int idx = tid; idx < M; idx += blockDim.x) sb[idx] = b1[idx];
    for (int idx = tid; idx < H; idx += blockDim.x) sb[M + idx] = b2[idx];
    if (ra.ln) {
      const float* sc = pick_ptr8(ra.ln_scale, i);
      const float* bi = pick_ptr8(ra.ln_bias, i);
      for (int idx = tid; idx < H; idx += blockDim.x) { sb[M + H + idx] = sc[idx]; sb[M + 2 * H + idx] = bi[idx]; }
    }
  }
  if (ra.final_ln)
    for (int idx = tid; idx < H; idx += blockDim.x) { s_fln[idx] = ra.fln_scale[idx]; s_fln[H + idx] = ra.fln_bias[idx]; }
  __syncthreads();

  const unsigned voff_2 = ls * (16 * ROWB) + li * 16;    // (1024, 256) kernel: row = GW g + j + 4 ROWR(r) + 16 s
  const unsigned voff_1 = ls * (16 * ROWB1) + li * 16;   // (256, 1024) kernel: same row map, 4 KB rows
  const unsigned voff_x = ls * (4 * ROWB) + li * 16;     // Dense_0 x rows: row = ROWR(r) + 4 s
  const unsigned voff_l = lane * 16;                     // Dense_L: per-lane packed image
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_w0p, 0, a.w0p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_wlq, 0, NK * 64 * 4, 0x00020000);

  // LayerNorm over the 256 features of every chain, in place (flax: eps 1e-6, scale and bias; variance about the mean):
  // the same pass as sampler_rs.hip's
  auto ln_pass = [&](acc_t (&Q)[NB], const float* sl) {
    float p1[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    sfor<NB>([&](auto BI) {
      constexpr int b = decltype(BI)::value;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; p1[(b * NR + r) & 7] += Q[b][r]; });
    });
    float s1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + (p1[6] + p1[7]));
    s1 += __shfl_xor(s1, 32);
    const float mean = s1 * (1.0f / H);
    float p2[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    sfor<NB>([&](auto BI) {
      constexpr int b = decltype(BI)::value;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; const float v = Q[b][r] - mean; p2[(b * NR + r) & 7] = fmaf(v, v, p2[(b * NR + r) & 7]); });
    });
    float s2 = ((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p2[4] + p2[5]) + (p2[6] + p2[7]));
    s2 += __shfl_xor(s2, 32);
    const float rstd = 1.0f / sqrtf(s2 * (1.0f / H) + 1e-6f);
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      sfor<NR>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const f32x4 sc = *(const f32x4*)(sl + g * GW + 4 * (G::rowr(r) + 4 * ls));
        const f32x4 bi = *(const f32x4*)(sl + H + g * GW + 4 * (G::rowr(r) + 4 * ls));
        sfor<4>([&](auto JI) {
          constexpr int j = decltype(JI)::value;
          Q[g * 4 + j][r] = (Q[g * 4 + j][r] - mean) * (rstd * sc[j]) + bi[j];
        });
      });
    });
  };

  // one operand ring for both GEMMs of a block: slot k % RING holds k-step k's loads (GEMM 1 uses [.][0] only)
  f32x4 ring[RING][NG];
  auto load1 = [&](__amdgpu_buffer_rsrc_t rs, int g, auto KI) {        // expansion kernel, output group g
    constexpr int k = decltype(KI)::value;
    constexpr int b = k / NR, r = k % NR;
    constexpr int soff = (GW * (b >> 2) + (b & 3) + 4 * G::rowr(r)) * ROWB1;
    ring[k % RING][0] = ldw(rs, voff_1, soff + g * (GW * 4));
  };
  auto load2 = [&](__amdgpu_buffer_rsrc_t rs, int g, auto KI) {        // contraction kernel, input group g
    constexpr int k = decltype(KI)::value;
    constexpr int j = k / NR, r = k % NR;
    constexpr int soff = (j + 4 * G::rowr(r)) * ROWB;
    sfor<NG>([&](auto GI) {
      constexpr int gg = decltype(GI)::value;
      ring[k % RING][gg] = ldw(rs, voff_2 + gg * (GW * 4), soff + g * (GW * ROWB));
    });
  };
  f32x4 r0[RING][NG];
  auto l0load = [&](auto NI) {
    constexpr int n = decltype(NI)::value;
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      r0[n % RING][g] = ldw(rs0, voff_x + g * (GW * 4), G::rowr(n) * ROWB);
    });
  };

  acc_t P[NB], Xn[NB];
  const int n_wt = (int)((a.B + TR - 1) / TR);
  const __amdgpu_buffer_rsrc_t rs_first = __builtin_amdgcn_make_buffer_rsrc((void*)ra.w1[0], 0, H * ROWB1, 0x00020000);

  // weight operands never depend on the data: the first ring entries of Dense_0 and of block 0 stay one phase ahead
  sfor<(NRX < RING - 1 ? NRX : RING - 1)>([&](auto NI) { l0load(NI); });
  sfor<RING - 1>([&](auto KI) { load1(rs_first, g_lo, KI); });

  for (int wt = (SPLIT > 1) ? (int)blockIdx.x : (int)(blockIdx.x * nw + wave); wt < n_wt; wt += (SPLIT > 1) ? (int)gridDim.x : (int)(gridDim.x * nw)) {
    const long long chain = (long long)wt * TR + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    float xs[NRX];
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = G::rowr(r) + 4 * ls;
      xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f;
    });
    float aux[NRX];     // dpm++2m: previous x0_hat; dpm2: the step's original xt
    sfor<NRX>([&](auto RI) { aux[decltype(RI)::value] = 0.0f; });

    for (int j = 0; j < a.NE; ++j) {
      float zr[NRX];
      if (a.rng_impl != FRL_RNG_NONE) {
        const bool used = s_coef[j * COEF_STRIDE + 5] != 0.0f;
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = G::rowr(r) + 4 * ls;
          zr[r] = (used && valid && ai < a.A) ? rng_step_normal(a.rng_keys, a.rng_impl, j, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai) : 0.0f;
        });
      } else {
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = G::rowr(r) + 4 * ls;
          zr[r] = (a.noise != nullptr && valid && ai < a.A) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
      }
      // ---------------- Dense_0 : P = tb[j] + zs[state] + W0x^T x ----------------------------------------------------
      {
        const float* sb = s_tb + j * H;
        sfor<NG>([&](auto GI) {
          constexpr int g = decltype(GI)::value;
          sfor<NR>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            const f32x4 bv = *(const f32x4*)(sb + g * GW + 4 * (G::rowr(r) + 4 * ls));
            sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[g * 4 + jj][r] = bv[jj]; });
          });
        });
      }
      if (a.state_bias != nullptr) {
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
      sfor<NRX>([&](auto NI) {
        constexpr int n = decltype(NI)::value;
        if constexpr (n + RING - 1 < NRX) l0load(std::integral_constant<int, n + RING - 1>{});
        if (n < a.AR) {
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const f32x4 aw = r0[n % RING][g];
            sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[g * 4 + jj] = G::mfma(aw[jj], xs[n], P[g * 4 + jj]); });
          });
        }
      });

      // ---------------- residual blocks -------------------------------------------------------------------------------
      for (int i = 0; i < nblk; ++i) {
        const float* sb = s_blk + i * BLK;
        const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr8(ra.w1, i), 0, H * ROWB1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr8(ra.w2, i), 0, M * ROWB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs1n = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr8(ra.w1, (i + 1 < nblk) ? i + 1 : 0), 0, H * ROWB1, 0x00020000);
        sfor<NB>([&](auto BI) { constexpr int b = decltype(BI)::value; Xn[b] = P[b]; });
        if (ra.ln) ln_pass(Xn, sb + M + H);
        // x + b2: the residual and Dense_1's bias are the accumulator's starting value (SPLIT: wave 0's; the others start at 0)
        if (SPLIT == 1 || wave == 0) {
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            sfor<NR>([&](auto RI) {
              constexpr int r = decltype(RI)::value;
              const f32x4 bv = *(const f32x4*)(sb + M + g * GW + 4 * (G::rowr(r) + 4 * ls));
              sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[g * 4 + jj][r] += bv[jj]; });
            });
          });
        } else {
          sfor<NB>([&](auto BI) { constexpr int b = decltype(BI)::value; sfor<NR>([&](auto RI) { P[b][decltype(RI)::value] = 0.0f; }); });
        }
        for (int g = g_lo; g < g_hi; ++g) {
          acc_t Y[4];
          sfor<NR>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            const f32x4 bv = *(const f32x4*)(sb + g * GW + 4 * (G::rowr(r) + 4 * ls));
            sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; Y[jj][r] = bv[jj]; });
          });
          // ---- Y = b1[g] + W1[:, g]^T Xn ----
          sfor<NK>([&](auto KI) {
            constexpr int k = decltype(KI)::value;
            if constexpr (k + RING - 1 < NK) load1(rs1, g, std::integral_constant<int, k + RING - 1>{});
            else load2(rs2, g, std::integral_constant<int, k + RING - 1 - NK>{});
            const f32x4 aw = ring[k % RING][0];
            const float bop = Xn[k / NR][k % NR];
            sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; Y[jj] = G::mfma(aw[jj], bop, Y[jj]); });
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, (k + RING - 1 < NK) ? 1 : NG, 0);   // VMEM reads
            __builtin_amdgcn_sched_barrier(0);
          });
          // ---- x += W2[g, :]^T act(Y) ----
          const int gn = (g + 1 < g_hi) ? g + 1 : g_lo;
          const __amdgpu_buffer_rsrc_t rsn = (g + 1 < g_hi) ? rs1 : rs1n;
          float bop, bn1, bn2 = 0.0f;
          { const f32x2 y = act_fn2<ACT>(f32x2{Y[0][0], Y[0][1]}); bop = y[0]; bn1 = y[1]; }
          sfor<NK2>([&](auto KI) {
            constexpr int k = decltype(KI)::value;
            if constexpr (k + RING - 1 < NK2) load2(rs2, g, std::integral_constant<int, k + RING - 1>{});
            else load1(rsn, gn, std::integral_constant<int, k + RING - 1 - NK2>{});
            if constexpr ((k & 1) == 1 && k + 2 < NK2) {
              const f32x2 y = act_fn2<ACT>(f32x2{Y[(k + 1) / NR][(k + 1) % NR], Y[(k + 2) / NR][(k + 2) % NR]});
              bn1 = y[0]; bn2 = y[1];
            }
            sfor<NG>([&](auto GI) {
              constexpr int gg = decltype(GI)::value;
              const f32x4 aw = ring[k % RING][gg];
              sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[gg * 4 + jj] = G::mfma(aw[jj], bop, P[gg * 4 + jj]); });
            });
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, (k + RING - 1 < NK2) ? NG : 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            bop = bn1; bn1 = bn2;
          });
        }
        if constexpr (SPLIT > 1) {
          // x <- sum over the four waves' partial sums, in wave order, in every wave: two halves of 4 feature blocks each
          sfor<2>([&](auto HI) {
            constexpr int hb = 4 * decltype(HI)::value;
            sfor<4>([&](auto BI) {
              constexpr int b = decltype(BI)::value;
              sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; s_red[((wave * 64) + b * NR + r) * 64 + lane] = P[hb + b][r]; });
            });
            __syncthreads();
            sfor<4>([&](auto BI) {
              constexpr int b = decltype(BI)::value;
              sfor<NR>([&](auto RI) {
                constexpr int r = decltype(RI)::value;
                const int o = (b * NR + r) * 64 + lane;
                P[hb + b][r] = ((s_red[o] + s_red[64 * 64 + o]) + s_red[2 * 64 * 64 + o]) + s_red[3 * 64 * 64 + o];
              });
            });
            __syncthreads();
          });
        }
      }

      // next step's (or next tile's) first Dense_0 operands: in flight under Dense_L and the epilogue
      sfor<(NRX < RING - 1 ? NRX : RING - 1)>([&](auto NI) { l0load(NI); });

      // ---------------- head: act (ResidualMLP, mlp.py:96) or LayerNorm (Simba, simba.py:58), then Dense_L -------------
      if (ra.final_ln) ln_pass(P, s_fln);
      if (ra.final_act)
        sfor<NB>([&](auto BI) {
          constexpr int b = decltype(BI)::value;
          sfor<NR / 2>([&](auto RI) {
            constexpr int r = 2 * decltype(RI)::value;
            const f32x2 y = act_fn2<ACT>(f32x2{P[b][r], P[b][r + 1]});
            P[b][r] = y[0]; P[b][r + 1] = y[1];
          });
        });
      constexpr int NLD = NK / 4;
      f32x4 rl[RINGL];
      sfor<RINGL - 1>([&](auto QI) { constexpr int q = decltype(QI)::value; rl[q] = ldw(rsl, voff_l, q * 1024); });
      acc_t O;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; O[r] = s_bl[G::rowr(r) + 4 * ls]; });
      sfor<NK>([&](auto KI) {
        constexpr int k = decltype(KI)::value;
        if constexpr ((k & 3) == 0 && k / 4 + RINGL - 1 < NLD)
          rl[(k / 4 + RINGL - 1) % RINGL] = ldw(rsl, voff_l, (k / 4 + RINGL - 1) * 1024);
        O = G::mfma(rl[(k / 4) % RINGL][k & 3], P[k / NR][k % NR], O);
      });

      // ---------------- epilogue: one reverse-diffusion / Euler step (the update rules of sampler_rs.hip) --------------
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
      const float c6 = s_coef[j * COEF_STRIDE + 6];            // dpm2: 1 on the second evaluation of a step
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        const float eps = O[r], xt = xs[r];
        if (a.xt_hist != nullptr && in && c6 == 0.0f && (SPLIT == 1 || wave == 0)) {
          const int jh = (a.mode == MODE_CDDPM_DPM2) ? j >> 1 : j;
          const long long o = ((long long)jh * a.B + chain) * a.A + ai;
          a.xt_hist[o] = xt;
          a.eps_hist[o] = eps;
        }
        float xn;
        if (a.mode == MODE_DDPM_DDPM) {
          float x0h = c0 * (xt - c1 * eps);                       // ddpm.py:199
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);   // ddpm.py:200-201
          xn = c2 * (c3 * x0h + c4 * xt);                           // ddpm.py:202-203
          xn += c5 * zr[r];                                         // ddpm.py:206
        } else if (a.mode == MODE_DDPM_DDIM) {
          xn = c0 * (xt - c1 * eps) + c2 * eps;                     // ddpm.py:211
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        } else if (a.mode == MODE_CDDPM_DDPM) {
          float x0h = (xt - c1 * eps) / c0;                         // continuous_ddpm.py:165
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * xt;                                  // :173
          xn += c5 * zr[r];                                         // :174
        } else if (a.mode == MODE_CDDPM_DDIM) {
          float x0h = (xt - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;                                 // :177
        } else if (a.mode == MODE_CDDPM_DPMPP2M) {
          float x0h = (xt - c1 * eps) / c0;                         // :165
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          const float d = c4 * x0h - c5 * aux[r];                   // :207-211
          xn = c2 * xt - c3 * d;                                    // :212
          aux[r] = x0h;                                             // carry, :217
        } else if (a.mode == MODE_EVAL) {
          xn = eps;
        } else if (a.mode == MODE_CDDPM_DPM2) {
          const float xo = (c6 == 0.0f) ? xt : aux[r];              // :165,:194
          float x0h = (xo - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;                                 // :187 / :196
          aux[r] = xo;
        } else {
          xn = xt + eps / c0;                                       // cnf.py:108
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        }
        xs[r] = (a.mode == MODE_EVAL) ? ((valid && ai < a.AO) ? xn : 0.0f) : (in ? xn : 0.0f);
      });
    }

    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = G::rowr(r) + 4 * ls;
      if (valid && ai < a.AO && (SPLIT == 1 || wave == 0)) a.x0[chain * a.AO + ai] = xs[r];
    });
  }
}

// ---- launcher -----------------------------------------------------------------------------------------------------
template <int ACT, int NRX, int SPLIT>
static int launch_res_inst(const KArgs& a, const ResArgs& ra, hipStream_t stream) {
  auto kern = frl_sampler_res_kernel<ACT, NRX, SPLIT>;
  const int ncu = device_cus();
  const long long n_wt = (a.B + 31) / 32;
  long long per_cu = (n_wt + ncu - 1) / ncu;
  int nw = (int)(per_cu < 1 ? 1 : (per_cu > MAX_WAVES ? MAX_WAVES : per_cu));
  if (nw == 3) nw = 4;
  long long nwg = (n_wt + nw - 1) / nw;
  int grid = (int)(nwg < ncu ? nwg : ncu);
  if (SPLIT > 1) { nw = 4; grid = (int)(n_wt < ncu ? n_wt : ncu); }     // one tile at a time per workgroup of four waves (= per CU)
  const size_t lds = res_lds_bytes(a.NE, ra.n_blocks, SPLIT);
  if (lds > 160 * 1024) { set_error("residual sampler: %d evaluations x %d blocks need %zu B of LDS (> 160 KiB)", a.NE, ra.n_blocks, lds); return FRL_EUNSUPPORTED; }
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nw), lds, stream, a, ra);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("residual sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_sampler_res_kernel<act %d, NRX %d, split %d> [grid %d x %d waves]", ACT, NRX, SPLIT, grid, nw);
  return FRL_OK;
}

template <int ACT>
int launch_sampler_res_act(const KArgs& a, const ResArgs& ra, hipStream_t stream) {
  if (a.H != 256 || a.TR != 32 || a.C != 0 || a.CK != 0) { set_error("residual sampler: 256-wide blocks with hoisted condition rows only"); return FRL_EUNSUPPORTED; }
  if (ra.n_blocks < 1 || ra.n_blocks > FRL_MAX_HIDDEN) { set_error("residual sampler: %d blocks", ra.n_blocks); return FRL_EINVAL; }
  if (ra.split > 1) {
    if (a.AR <= 4) return launch_res_inst<ACT, 4, 4>(a, ra, stream);
    if (a.AR <= 12) return launch_res_inst<ACT, 12, 4>(a, ra, stream);
  } else {
    if (a.AR <= 4) return launch_res_inst<ACT, 4, 1>(a, ra, stream);
    if (a.AR <= 12) return launch_res_inst<ACT, 12, 1>(a, ra, stream);
  }
  set_error("residual sampler: x_dim %d exceeds the compiled operand budget (x_dim <= 24)", a.A);
  return FRL_EUNSUPPORTED;
}

#ifndef FRL_RES_ACT
#error "compile with -DFRL_RES_ACT=<FRL_ACT_*>: one translation unit per activation"
#endif
template int launch_sampler_res_act<FRL_RES_ACT>(const KArgs&, const ResArgs&, hipStream_t);

}  // namespace frl
