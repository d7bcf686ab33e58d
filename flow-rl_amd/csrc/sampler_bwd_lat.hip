// Latency build of the chain VJP's backward sweep (phase 2 of sampler_bwd.hip: backward from the stash), exact-f32 MFMA
// 32x32x2: ONE workgroup of four waves per tile of 32 chains, the features of every layer split over the four SIMDs —
// what sampler_lat.hip is to the forward sampler.
//
// The one-wave-per-tile sweep gives a wave a whole tile: no barrier, 78 % of the MFMA peak at 262 144 chains, but a
// tile's step is 2 272 serial MFMAs, and an actor update on a few hundred transitions (dql.py:93-123 at batch 256: eight
// tiles) leaves the GPU idle behind that latency (0.35 ms of a 0.67 ms step).  The backward of the net is itself an MLP
// with the transposed kernels, so the forward latency kernel's structure carries over, wave w owning features
// [64 w, 64 (w + 1)) of every layer:
//
//   VJP of the update rule     every wave, on its own copy of the chain state (xt_hist, eps_hist, dL/dx_{j+1}) -> dL/deps
//   dL/dh_{L-1} = W_L dL/deps  k-steps over the action rows with the W_L^T image (BArgs.ws_wlt), like Dense_0's x part
//   hidden layers, downwards   dL/dz_l = dL/dh_l * act'(z_l) (act' from the training forward's stash, dL/dz_l written to
//                              the stash for the dW kernel, both in the one-wave layout: see sampler_lat.hip STASH for the
//                              index map), exchanged through LDS in B-operand order, one barrier per layer; then
//                              dL/dh_{l-1} = W_l dL/dz_l with the transposed kernels (BArgs.wt) through a register ring
//   dL/dx_j                    split-K over the waves: this wave's dL/dz_0 features are B operands already, Dense_0's x
//                              rows (packed layer-0 image, resident in registers) the A operands; partial tiles meet in LDS
//
// No activation is evaluated here (act' was stashed), so there is ONE kernel per operand width, not one per activation.
// Results equal the one-wave sweep's up to the order of the split-K sums; the dW kernel and the fixed-order reduce are
// shared.
#include "frl_device.h"
#include "frl_actgrad.h"

namespace frl {

namespace {

__device__ __forceinline__ f32x2 ld2b(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}

// DCOND (a cond_embedding, dacer.py:141-162): dL/dcfeat[chain] = sum over steps of W0[cond rows] dL/dz_0.  The contraction
// is linear, so this wave's dL/dz_0 features are first ADDED over the steps (32 registers) and contracted once per tile:
// four 32-row blocks, split-K partials summed over the waves through LDS.  The A operands come from the packed image
// BArgs.ws_w0cq (frl_bwd_prepare_kernel: [4 blocks][k-step / 4][lane][4] in the one-wave k-step order), where this wave's
// k-steps (jb, r) and (jb, r | 2) are two adjacent floats — the index map of the stash.
// TR = 16 (MFMA 16x16x4): tiles of 16 chains — half the latency per tile while every CU still gets at most one — on the SAME
// 32-chain stash blocks: tile t is chains 16 (t & 1) .. + 15 of wave-tile t >> 1 (sampler_lat.hip STASH has the index map:
// register (jb, r) of lane (li, ls) at k-step 16 (4 (w >> 1) + jb) + r + 8 (w & 1) + 4 (ls >> 1), lane 16 (t & 1) + li + 32 (ls & 1)).
template <int TR, int NRX, bool DCOND>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_bwd_sweep_lat_kernel(const BArgs b) {
  typedef Geo<TR> G;
  typedef typename G::acc_t acc_t;
  static_assert(TR == 32 || (TR == 16 && !DCOND && NRX == 4), "16-chain tiles: x_dim <= 16, no dL/dcfeat");
  constexpr int H = 256, NR = G::NR, KS = G::KS;
  constexpr int NBW = H / (4 * TR);   // 2 (TR 32) or 4 (TR 16) output blocks per wave
  constexpr int FW = H / 4;           // 64 features per wave
  constexpr int NK = H / KS;          // 128 k-steps per hidden layer
  constexpr int ROWB = H * 4;
  constexpr int RW = (TR == 16) ? 32 : 16;   // hidden-layer weight ring (k-steps ahead), as in the forward latency kernel
  constexpr int PSTEP = (TR == 32) ? 2 : 1;  // registers rp and rp + PSTEP are the two floats of one stash element
  auto pair_reg = [](int m) { return TR == 32 ? 4 * (m >> 1) + (m & 1) : 2 * m; };   // first register of pair m
  static_assert(NBW == KS && NK % RW == 0, "geometry");
  const KArgs& a = b.k;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane % TR, ls = lane / TR;

  float* s_coef = (float*)smem;                       // [T][8]
  float* s_act = s_coef + a.T * COEF_STRIDE;          // 2 x [H/8][64][4]: dL/dz in B-operand order | Dense_0 partial tiles
  for (int idx = tid; idx < a.T * COEF_STRIDE; idx += 256) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  const unsigned voff_col = (unsigned)(FW * wave + NBW * li) * 4;
  const unsigned voff_h = ls * ROWB + voff_col;         // transposed hidden kernels: k-step's row 2 ks + ls, this wave's columns
  const unsigned voff_x = ls * (4 * ROWB) + voff_col;   // W_L^T rows ROWR(n) + 4 ls
  const unsigned voff_st0 = (TR == 32) ? (unsigned)(32 * (wave >> 1) + 4 * (wave & 1) + ls) * 512 + li * 8      // stash element address, per-lane part
                                       : (unsigned)(32 * (wave >> 1) + 4 * (wave & 1) + 2 * (ls >> 1)) * 512 + (li + 32 * (ls & 1)) * 8;
  auto stash_off = [](int jb, int rp) { return TR == 32 ? (16 * (rp & 1) + 8 * jb + 2 * (rp >> 3)) * 512 + 256 * ((rp >> 2) & 1) : (8 * jb + (rp >> 1)) * 512; };
  const __amdgpu_buffer_rsrc_t rs_wlt = __builtin_amdgcn_make_buffer_rsrc((void*)b.ws_wlt, 0, a.XR * ROWB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cq = __builtin_amdgcn_make_buffer_rsrc((void*)(DCOND ? b.ws_w0cq : b.ws_wlt), 0, DCOND ? 4 * H * 64 * 4 : 0, 0x00020000);
  const unsigned voff_cq = (unsigned)((16 * (wave >> 1) + 2 * (wave & 1)) * 256 + li * 4 + 2 * ls) * 4;
  auto cq_off = [](int m, int jb, int rp) { return (m * (H * 64) + (8 * (rp & 1) + 4 * jb + (rp >> 3)) * 256 + 128 * ((rp >> 2) & 1)) * 4; };   // bytes; rp: bit 1 clear

  // Dense_0's x rows stay resident: k-step (jb, r) contracts feature FW w + NBW (ROWR(r) + 4 ls) + jb, lane i = action row
  float w0x[NBW][NR];
  sfor<NBW>([&](auto JI) {
    constexpr int jb = decltype(JI)::value;
    sfor<NR>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int f = FW * wave + NBW * (G::rowr(r) + 4 * ls) + jb;
      w0x[jb][r] = (li < a.A) ? a.ws_w0p[(size_t)li * H + f] : 0.0f;
    });
  });

  typedef typename std::conditional<TR == 32, f32x2, f32x4>::type wv_t;      // NBW floats: this lane's columns of one weight row
  auto wld = [](__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) -> wv_t {
    if constexpr (TR == 32) return ld2b(rs, voff, soff);
    else return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  };
  wv_t ring[RW];
  if (a.L > 1) {
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(b.wt, a.L - 2), 0, H * ROWB, 0x00020000);
    sfor<RW - 1>([&](auto KI) { constexpr int k = decltype(KI)::value; ring[k] = wld(rs1, voff_h, KS * k * ROWB); });
  }

  const int ntiles = (TR == 16) ? 2 * (int)((a.B + 31) / 32) : (int)((a.B + TR - 1) / TR);   // (16: both halves of the last wave-tile, so that every dL/dz row is written)
  int phase = 0;   // LDS buffer parity, runs on across layers, steps and tiles
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long chain = (long long)tile * TR + li;
    const bool valid = chain < a.B;

    f32x2 gsum[DCOND ? NBW : 1][DCOND ? NR / 2 : 1];      // sum over the steps of this wave's dL/dz_0 features (register pairs)
    if constexpr (DCOND) sfor<NBW>([&](auto JI) { sfor<NR / 2>([&](auto MI) { gsum[decltype(JI)::value][decltype(MI)::value] = f32x2{0.0f, 0.0f}; }); });
    float gx[NRX];      // dL/dx_{j+1}, same register layout as the chain state
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = G::rowr(r) + 4 * ls;
      gx[r] = (valid && ai < a.A) ? b.g_x0[chain * a.A + ai] : 0.0f;
    });

    for (int jj = 0; jj < a.T; ++jj) {
      const int j = a.T - 1 - jj;
      const long long ts = (long long)(TR == 32 ? tile : tile >> 1) * a.T + j;
      const size_t srow = (size_t)ts * a.L * 8192;
      const unsigned voff_st = voff_st0 + ((TR == 16) ? (unsigned)(tile & 1) * 128 : 0u);
      const __amdgpu_buffer_rsrc_t rs_ad = __builtin_amdgcn_make_buffer_rsrc((void*)(b.ad + srow), 0, a.L * 32768, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_gd = __builtin_amdgcn_make_buffer_rsrc((void*)(b.gd + srow), 0, a.L * 32768, 0x00020000);

      // act'(z_{L-1}) of this wave's features and the W_L^T operands: requested before the update's arithmetic
      f32x2 adv[NBW][NR / 2];
      auto load_ad = [&](int l) {
        sfor<NBW>([&](auto JI) {
          constexpr int jb = decltype(JI)::value;
          sfor<NR / 2>([&](auto MI) {
            constexpr int m = decltype(MI)::value;
            adv[jb][m] = ld2b(rs_ad, voff_st, l * 32768 + stash_off(jb, pair_reg(m)));
          });
        });
      };
      load_ad(a.L - 1);
      wv_t r0[NRX];
      sfor<NRX>([&](auto NI) { constexpr int n = decltype(NI)::value; r0[n] = wld(rs_wlt, voff_x, G::rowr(n) * ROWB); });

      float xs[NRX], eh[NRX];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        const long long o = ((long long)j * a.B + chain) * a.A + ai;
        xs[r] = (in && b.xt_hist != nullptr) ? b.xt_hist[o] : 0.0f;
        eh[r] = (in && b.eps_hist != nullptr) ? b.eps_hist[o] : 0.0f;
      });

      // ================= VJP of the update (sampler_bwd.hip, verbatim): dL/deps and the direct part of dL/dx_j =====
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4];
      float ge[NRX], gxd[NRX];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const float eps = eh[r], xt = xs[r], gn = gx[r];
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
          g_e = gn; g_x = 0.0f;
        } else {
          const float xn = xt + eps / c0;
          const float m = (!a.clip || (xn >= a.x_min && xn <= a.x_max)) ? 1.0f : 0.0f;
          g_e = gn * m / c0; g_x = gn * m;
        }
        const int ai = G::rowr(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        ge[r] = in ? g_e : 0.0f;
        gxd[r] = in ? g_x : 0.0f;
        if (wave == 0) {
          if constexpr (TR == 32) b.ge[(size_t)ts * b.ge_stride + r * 64 + lane] = ge[r];
          else {      // action row r + 4 ls of chain 16 (t & 1) + li: register r + 4 (ls >> 1), lane half ls & 1 of the one-wave layout
            const int r32 = r + 4 * (ls >> 1);
            if (r32 < (b.ge_stride >> 6)) b.ge[(size_t)ts * b.ge_stride + r32 * 64 + 16 * (tile & 1) + li + 32 * (ls & 1)] = ge[r];
            // the wide stash also has registers 8 .. 11 (action rows 16 .. 23: none here), which the dW kernel reads: zeros
            if (r == 0 && (b.ge_stride >> 6) > 8) {
              b.ge[(size_t)ts * b.ge_stride + (8 + ls) * 64 + 16 * (tile & 1) + li] = 0.0f;
              b.ge[(size_t)ts * b.ge_stride + (8 + ls) * 64 + 16 * (tile & 1) + li + 32] = 0.0f;
            }
          }
        }
      });

      // ================= dL/dh_{L-1} = W_L dL/deps: this wave's 64 features ====================================
      acc_t acc[NBW];
      sfor<NBW>([&](auto JI) { sfor<NR>([&](auto RI) { acc[decltype(JI)::value][decltype(RI)::value] = 0.0f; }); });
      sfor<NRX>([&](auto NI) {
        constexpr int n = decltype(NI)::value;
        if (n < a.AR) {
          const wv_t aw = r0[n];
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb] = G::mfma(aw[jb], ge[n], acc[jb]); });
        }
      });

      // ================= hidden layers, downwards: dL/dh_{l-1} = W_l (dL/dh_l * act'(z_l)) =======================
      for (int l = a.L - 1; l >= 1; --l) {
        float* abuf = s_act + (phase++ & 1) * (H * TR);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(b.wt, l - 1), 0, H * ROWB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(b.wt, (l >= 2) ? l - 2 : a.L - 2), 0, H * ROWB, 0x00020000);
        // feature f = FW w + NBW row + jb is contracted by k-step ks = f / KS on lane group f % KS: registers 4 q .. 4 q + 3
        // are the four k-steps of one 16-byte group of this lane's column (as in the forward latency kernel)
        sfor<NBW>([&](auto JI) {
          constexpr int jb = decltype(JI)::value;
          sfor<NR / 4>([&](auto QI) {
            constexpr int q = decltype(QI)::value;
            const int ks = (FW / KS) * wave + G::rowr(4 * q) + 4 * ls;
            f32x4 v;
            sfor<2>([&](auto EI) {
              constexpr int e = decltype(EI)::value;
              constexpr int e0 = (TR == 32) ? e : 2 * e, e1 = e0 + PSTEP;      // the pair's places in this group of four
              const f32x2 gz = f32x2{acc[jb][4 * q + e0], acc[jb][4 * q + e1]} * adv[jb][2 * q + e];
              st2(gz, rs_gd, voff_st, l * 32768 + stash_off(jb, 4 * q + e0));
              v[e0] = gz[0]; v[e1] = gz[1];
            });
            *(f32x4*)(abuf + (((ks >> 2) * 64) + jb * TR + li) * 4) = v;
          });
        });
        load_ad(l - 1);      // the next layer's act', in flight under this layer's MFMAs
        sfor<NBW>([&](auto JI) { sfor<NR>([&](auto RI) { acc[decltype(JI)::value][decltype(RI)::value] = 0.0f; }); });
        __syncthreads();
        const f32x4* bsrc = (const f32x4*)abuf + lane;
        f32x4 bq = bsrc[0], bqn = bq;
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr (k + RW - 1 < NK) ring[(k + RW - 1) % RW] = wld(rs, voff_h, KS * (k + RW - 1) * ROWB);
          else ring[(k + RW - 1) % RW] = wld(rsn, voff_h, KS * (k + RW - 1 - NK) * ROWB);
          if constexpr ((k & 3) == 0 && k + 4 < NK) bqn = bsrc[(k / 4 + 1) * 64];
          const wv_t aw = ring[k % RW];
          const float bop = bq[k & 3];
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb] = G::mfma(aw[jb], bop, acc[jb]); });
          if constexpr ((k & 3) == 3) bq = bqn;
        });
      }

      // ================= dL/dz_0 -> stash; dL/dx_j (net part) = W0[x rows] dL/dz_0, split-K over the waves ==========
      acc_t OX;
      sfor<NR>([&](auto RI) { OX[decltype(RI)::value] = 0.0f; });
      sfor<NBW>([&](auto JI) {
        constexpr int jb = decltype(JI)::value;
        sfor<NR / 2>([&](auto MI) {
          constexpr int m = decltype(MI)::value;
          constexpr int rp = (TR == 32) ? 4 * (m >> 1) + (m & 1) : 2 * m;
          const f32x2 gz = f32x2{acc[jb][rp], acc[jb][rp + PSTEP]} * adv[jb][m];
          st2(gz, rs_gd, voff_st, stash_off(jb, rp));
          OX = G::mfma(w0x[jb][rp], gz[0], OX);
          OX = G::mfma(w0x[jb][rp + PSTEP], gz[1], OX);
          if constexpr (DCOND) gsum[jb][m] = gsum[jb][m] + gz;
        });
      });
      {
        float* part = s_act + (phase++ & 1) * (H * TR);
        sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; part[(wave * NR + r) * 64 + lane] = OX[r]; });
        __syncthreads();
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          float s = 0.0f;
          sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; s += part[(w * NR + r) * 64 + lane]; });
          const int ai = G::rowr(r) + 4 * ls;
          gx[r] = (valid && ai < a.A) ? gxd[r] + s : 0.0f;
        });
      }
    }
    if constexpr (DCOND) {
      // W0[cond rows] (sum over steps of dL/dz_0): operands one register pair ahead, then the four waves' partial tiles meet
      // in LDS, one 32-row block per phase; wave m writes block m
      constexpr int NPAIR = NBW * NR / 2;
      acc_t OC[4];
      sfor<4>([&](auto MI) { sfor<NR>([&](auto RI) { OC[decltype(MI)::value][decltype(RI)::value] = 0.0f; }); });
      f32x2 cq[2][4];
      auto cq_load = [&](auto PI) {
        constexpr int p = decltype(PI)::value, jb = p / (NR / 2), m = p % (NR / 2), rp = 4 * (m >> 1) + (m & 1);
        sfor<4>([&](auto CI) { constexpr int c = decltype(CI)::value; cq[p & 1][c] = ld2b(rs_cq, voff_cq, cq_off(c, jb, rp)); });
      };
      cq_load(std::integral_constant<int, 0>{});
      sfor<NPAIR>([&](auto PI) {
        constexpr int p = decltype(PI)::value, jb = p / (NR / 2), m = p % (NR / 2);
        if constexpr (p + 1 < NPAIR) cq_load(std::integral_constant<int, p + 1>{});
        sfor<4>([&](auto CI) {
          constexpr int c = decltype(CI)::value;
          OC[c] = G::mfma(cq[p & 1][c][0], gsum[jb][m][0], OC[c]);
          OC[c] = G::mfma(cq[p & 1][c][1], gsum[jb][m][1], OC[c]);
        });
      });
      sfor<4>([&](auto MI) {
        constexpr int m = decltype(MI)::value;
        float* part = s_act + (phase++ & 1) * (H * TR);
        sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; part[(wave * NR + r) * 64 + lane] = OC[m][r]; });
        __syncthreads();
        if (wave == m && valid) {
          sfor<NR>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            float sum = 0.0f;
            sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; sum += part[(w * NR + r) * 64 + lane]; });
            b.gcf[chain * COND_EMBED_DIM + 32 * m + G::rowr(r) + 4 * ls] = sum;
          });
        }
      });
    }
    if (b.g_xT != nullptr && wave == 0) {
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        if (valid && ai < a.A) b.g_xT[chain * a.A + ai] = gx[r];
      });
    }
  }
}

template <int TR, int NRX, bool DCOND>
int launch_inst(const BArgs& b, hipStream_t stream) {
  auto kern = frl_bwd_sweep_lat_kernel<TR, NRX, DCOND>;
  const KArgs& a = b.k;
  const long long ntiles = (TR == 16) ? 2 * ((a.B + 31) / 32) : (a.B + 31) / 32;
  const int ncu = device_cus();
  const int grid = (int)(ntiles < 4LL * ncu ? ntiles : 4LL * ncu);
  const size_t lds = sizeof(float) * ((size_t)a.T * COEF_STRIDE + 2 * 256 * 32);
  if (lds > 160 * 1024) { set_error("latency backward sweep: %d steps need %zu B of LDS (> 160 KiB)", a.T, lds); return FRL_EUNSUPPORTED; }
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, b);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("latency backward sweep launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_bwd_sweep_lat_kernel<TR %d, NRX %d%s> [grid %d]", TR, NRX, DCOND ? ", DCOND" : "", grid);
  return FRL_OK;
}

}  // namespace

// The rule of capi.hip's lat_train_ok for the forward half: small batches (latency rounds at most twice the throughput
// rounds), 256 wide, 32-chain tiles, and a stash to read.
bool bwd_sweep_lat_supports(const BArgs& b) {
  const KArgs& a = b.k;
  if (a.H != 256 || a.TR != 32 || a.L < 1 || a.NE != a.T || (b.gcf != nullptr && b.ws_w0cq == nullptr) || (b.eps_hist == nullptr && a.mode != MODE_EVAL)) return false;
  if (a.AR > 12 || a.A > 24 || b.ws_wlt == nullptr || a.AR > (b.ge_stride >> 6)) return false;
  const long long n_wt = (a.B + 31) / 32, cus = device_cus();
  return (n_wt + cus - 1) / cus <= 2 * ((n_wt + 4 * cus - 1) / (4 * cus));
}

int launch_bwd_sweep_lat(const BArgs& b, hipStream_t stream) {
  // the dL/deps stash holds ge_stride / 64 registers per lane (4 | 12: BwdPlan.wide), all of which the dW kernel reads
  if (b.gcf != nullptr) {      // a cond_embedding always takes the wide operand build (128 condition rows)
    if ((b.ge_stride >> 6) <= 4) { set_error("latency backward sweep: dL/dcfeat with the narrow dL/deps stash"); return FRL_EINVAL; }
    return launch_inst<32, 12, true>(b, stream);
  }
  // 16-chain tiles while every CU still gets at most one tile (x_dim <= 16; the narrow dL/deps stash holds 8 action rows)
  if (b.k.A <= ((b.ge_stride >> 6) <= 4 ? 8 : 16) && 2 * ((b.k.B + 31) / 32) <= device_cus()) return launch_inst<16, 4, false>(b, stream);
  return (b.ge_stride >> 6) <= 4 ? launch_inst<32, 4, false>(b, stream) : launch_inst<32, 12, false>(b, stream);
}

}  // namespace frl
