// Stage 2: the persistent fused T-step sampler, exact-f32 MFMA, weights streamed L2 -> registers.
//
// What one launch does (reference: DDPM.sample flowrl/flow/ddpm.py:184-221, ContinuousDDPM.sample
// continuous_ddpm.py:149-222, CNF.sample cnf.py:122-141; network ddpm.py:83-95 + mlp.py:25-33):
//
//   for each wave-tile of TR chains (persistent waves, one wave per SIMD, 512 registers per lane):
//       x <- xT ; cond operands <- cond rows (broadcast over particles)
//       for j in 0..T-1:   (all on chip)
//           P  = tb[j] + W0x^T x + W0c^T cond                    (Dense_0; time slice pre-folded into tb)
//           P  = b_l + W_l^T act(P)            l = 1..L-1        (hidden Dense layers)
//           e  = b_L + W_L^T act(P)                               (Dense_L, zero-padded to TR rows)
//           x <- update(x, e, noise[j])                           (DDPM / DDIM / Euler epilogue)
//       x0 <- x
//
// Layout.  Everything is computed TRANSPOSED: an MFMA tile is [TR output features] x [TR chains], a
// chain lives on a lane (column = lane % TR) and the f32 MFMA leaves feature row(r, s) = ROWR(r) + 4 s
// of the tile in accumulator register r of lane group s = lane / TR.  The B operand of the same
// instruction wants k-slot s on lane group s, so accumulator register r of a finished layer IS the B
// operand of the next layer's k-step that contracts the KS features {row(r, s)}: activations never
// leave the register file — no LDS bounce, no lane permute between layers.  Output feature
// f = GW g + 4 i + j sits in block (g, j), tile row i (one 16-byte A-operand load feeds the 4 blocks
// j = 0..3 of group g), and the same map indexes the next layer's K rows.
//
// Two geometries share this file:
//   TR = 32: v_mfma_f32_32x32x2_f32, KS = 2, 16 accumulator registers per block, H <= 256
//   TR = 16: v_mfma_f32_16x16x4_f32, KS = 4,  4 accumulator registers per block, H <= 512
// Both keep H/2 resp. H/4 registers of activations in and as many accumulators out per lane.
//
// Weights are read IN PLACE (Flax (in, out) row-major) straight into registers with
// buffer_load_dwordx4 (per-lane row = the k-slot's feature row, scalar offset = the k-step), a ring of
// RING k-steps ahead of the MFMAs; they are L2-resident (0.6-3.3 MB) and the four waves of a CU read
// the same lines at nearly the same time.  There is no LDS weight staging and no barrier in the main
// loop (the LDS-DMA K-slab variant, the round-1 sampler_g32.hip, since removed, lost ~1.2k cycles per 32 KB slab to DMA issue
// and barrier skew — A/B in DESIGN.md); LDS only holds the small bias / coefficient tables.
#include "frl_device.h"
#include "frl_rng.h"

namespace frl {

// The activation of a register pair.  FRL_RS_SCALAR_ACT (A/B switch, profiles/r03_ab_mish_*.json): two scalar evaluations
// instead of the packed-f32 form of frl_device.h's act_fn2.
template <int ACT>
__device__ __forceinline__ f32x2 rs_act2(f32x2 x) {
#ifdef FRL_RS_SCALAR_ACT
  return f32x2{act_fn<ACT>(x[0]), act_fn<ACT>(x[1])};
#else
  return act_fn2<ACT>(x);
#endif
}

// A pointer read from the critic's device-side table, as the wave-uniform value it is: without this the descriptors built
// from it live in vector registers and every buffer load is wrapped in a readfirstlane loop.
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}

// EXT: 0 plain, 1 rarer solvers / evaluation / hoisted condition rows, 2 = 1 + LayerNorm,
//      3 = 1 + critic scoring and best-of-N selection on the finished tile (CriticTab, frl_internal.h), 4 = 3 with a relu critic,
//      5 = 1 + in-kernel threefry draws as micro-steps behind the MFMAs of the first hidden layer (no LayerNorm code: its
//          presence alone costs the hot loops ~4 % in accumulator moves; LayerNorm + draws = EXT 2, draws up front)
//      6 = 0 + micro-step draws, ONE dense draw per lane and step (the two-wave build's keyed variant, x_dim <= 4: nothing of
//          EXT 1, so that it fits 256 registers)
template <int TR, int H, int ACT, int NRX, int CKMAX, int EXTV>
__device__ __forceinline__ void sampler_rs_body(const KArgs& a, const CriticTab& ct) {
  typedef Geo<TR> G;
  typedef typename G::acc_t acc_t;
  constexpr int EXT = (EXTV == 6) ? 0 : EXTV;
  constexpr bool HAS_LN = (EXT == 2), HAS_RNG = (EXTV == 5 || EXTV == 6), HAS_CRITIC = (EXT == 3 || EXT == 4);
  constexpr int KS = G::KS, NR = G::NR, RING = G::RING;
  constexpr int NB = H / TR;        // output (and input) feature blocks per hidden layer
  constexpr int NG = NB / 4;        // 16-byte A-operand loads per k-step
  constexpr int GW = 4 * TR;        // features per load group
  constexpr int NK = NB * NR;       // k-steps per hidden layer
  constexpr int ROWB = H * 4;       // bytes per weight row
  constexpr int RINGL = 8;          // last-layer operand ring (one 16-byte load = 4 k-steps)
  static_assert(NB % 4 == 0 && NK % RING == 0 && (NK / 4) % RINGL == 0, "geometry");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6;
  const int li = lane % TR, ls = lane / TR;

  float* s_tb = (float*)smem;                         // [NE][H] layer-0 bias table (time slice folded in), one row per network evaluation
  float* s_bh = s_tb + a.NE * H;                      // [L-1][H] hidden biases
  float* s_bl = s_bh + (a.L - 1) * H;                 // [32]    padded last-layer bias
  float* s_coef = s_bl + LAST_PAD;                    // [NE][8]
  for (int idx = tid; idx < a.NE * H; idx += blockDim.x) s_tb[idx] = a.ws_tb[idx];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr_rs(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += blockDim.x) s_bh[(l - 1) * H + idx] = bp[idx];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.NE * COEF_STRIDE; idx += blockDim.x) s_coef[idx] = a.ws_coef[idx];
  // EXT builds with the per-state hoist and whole tiles per state (particles a multiple of TR): the tile's ONE row of hoisted
  // condition features, copied per wave into LDS at the head of the tile — every step then reads it back at LDS latency
  // instead of paying an L2 round trip in front of Dense_0's MFMAs (cfg4: 32 x 16-byte loads per lane and step)
  const bool zrow_lds = EXT != 0 && a.state_bias != nullptr && (a.particles % TR) == 0;
  float* s_zrow = s_coef + a.NE * COEF_STRIDE;
  float* s_ln = s_zrow + (zrow_lds ? nw * H : 0);     // EXT 2, LayerNorm backbones only: [L][2][H] scale | bias rows
  // EXT 6 (keyed two-wave build): when a tile's TR * A draws fit one per lane, ONE dense draw per lane and step replaces the NRX
  // per-register draws (at x_dim 3 only 3 of a chain's 16 register rows exist: 48 of 256 draws were used); the values go to
  // their owner lanes through 64 wave-private floats of LDS (the EXT 6 build has no hoisted row / LayerNorm / critic tables here)
  // The plan sends a keyed call to this build only when TR * A <= 64 (plan_two_wave_shape), so it carries no per-register variant.
  constexpr bool DENSE = (EXTV == 6);
  float* s_zd = s_zrow + wave * 64;
  if constexpr (HAS_LN) {
    if (a.ws_ln != nullptr)
      for (int idx = tid; idx < a.L * 2 * H; idx += blockDim.x) s_ln[idx] = a.ws_ln[idx];
  }
  // EXT 3: the critic members' biases behind the sampler's tables: [E][H] bias_0 | [E][Lc-1][H] hidden | [E][32] padded head
  float* s_ctb = s_ln;
  float* s_cbh = s_ctb;
  float* s_cbl = s_ctb;
  if constexpr (HAS_CRITIC) {
    static_assert(TR == 32, "fused critic: 32-chain tiles");
    s_cbh = s_ctb + ct.E * H;
    s_cbl = s_cbh + ct.E * (ct.L - 1) * H;
    for (int e = 0; e < ct.E; ++e) {
      const float* tp = uniform_ptr(ct.ptrs[e * CT_STRIDE + CT_TB]);
      for (int idx = tid; idx < H; idx += blockDim.x) s_ctb[e * H + idx] = tp[idx];
      for (int l = 1; l < ct.L; ++l) {
        const float* bp = uniform_ptr(ct.ptrs[e * CT_STRIDE + CT_BH + l - 1]);
        for (int idx = tid; idx < H; idx += blockDim.x) s_cbh[(e * (ct.L - 1) + l - 1) * H + idx] = bp[idx];
      }
      if (tid < LAST_PAD) s_cbl[e * LAST_PAD + tid] = uniform_ptr(ct.ptrs[e * CT_STRIDE + CT_BLP])[tid];
    }
  }
  __syncthreads();

  // per-lane parts of the weight addresses (bytes)
  const unsigned voff_h = ls * (16 * ROWB) + li * 16;   // hidden: row = GWg + j + 4 ROWR(r) + 16 s
  const unsigned voff_x = ls * (4 * ROWB) + li * 16;    // Dense_0 x rows: row = ROWR(r) + 4 s
  const unsigned voff_c = ls * ROWB + li * 16;          // Dense_0 cond rows: row = XR + KS c + s
  const unsigned voff_l = lane * 16;                    // Dense_L: per-lane packed image
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_w0p, 0, a.w0p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsl = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_wlq, 0, NK * 64 * 4, 0x00020000);

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

  // flax.linen.LayerNorm over the H features of every chain, in place on a finished pre-activation tile (mlp.py:26-29:
  // Dense -> LayerNorm -> act): mean and E[x^2] over the lane's own registers, combined across the 64 / TR lanes that
  // hold the other feature rows of the same chain; eps = 1e-6.
  auto ln_pass = [&](acc_t (&Q)[NB], const float* sl) {
    if (a.ln_post_act != 0) {      // FQL's order (fql/network.py:26-30): the activation first, in place; consumers are identity
      const int act = a.ln_post_act - 1;
      sfor<NB>([&](auto BI) { constexpr int b = decltype(BI)::value; sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; Q[b][r] = act_rt(act, Q[b][r]); }); });
    }
    // eight interleaved partial sums combined as a tree (then the lane-group butterfly): the rounding error grows with
    // NB NR / 8 + 3 additions instead of NB NR, i.e. no worse than numpy's pairwise mean
    float p1[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    sfor<NB>([&](auto BI) {
      constexpr int b = decltype(BI)::value;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; p1[(b * NR + r) & 7] += Q[b][r]; });
    });
    float s1 = ((p1[0] + p1[1]) + (p1[2] + p1[3])) + ((p1[4] + p1[5]) + (p1[6] + p1[7]));
#pragma unroll
    for (int m = TR; m < 64; m <<= 1) s1 += __shfl_xor(s1, m);
    const float mean = s1 * (1.0f / H);
    // the variance about the mean (two passes over registers): the same quantity as flax's E[x^2] - E[x]^2 without
    // its float32 cancellation, so the result sits closer to the exact value than the reference's own
    float p2[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    sfor<NB>([&](auto BI) {
      constexpr int b = decltype(BI)::value;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; const float v = Q[b][r] - mean; p2[(b * NR + r) & 7] = fmaf(v, v, p2[(b * NR + r) & 7]); });
    });
    float s2 = ((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p2[4] + p2[5]) + (p2[6] + p2[7]));
#pragma unroll
    for (int m = TR; m < 64; m <<= 1) s2 += __shfl_xor(s2, m);
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

  // hidden-layer operand ring: slot (k % RING) holds the NG loads of k-step k
  f32x4 ring[RING][NG];
  auto hload = [&](__amdgpu_buffer_rsrc_t rs, auto KI) {
    constexpr int k = decltype(KI)::value % NK;
    constexpr int b = k / NR, r = k % NR;
    constexpr int soff = (GW * (b >> 2) + (b & 3) + 4 * G::rowr(r)) * ROWB;
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      ring[k % RING][g] = ldw(rs, voff_h + g * (GW * 4), soff);
    });
  };

  // Dense_0 operand ring (x rows then cond rows).  Loads are unconditional (dead slots read rows nobody uses;
  // the descriptor bounds-checks) so that the vmcnt bookkeeping stays exact across the uniform `live` branches.
  constexpr int N0 = NRX + CKMAX;       // compile-time k-step slots; slot n is live iff live(n)
  f32x4 r0[RING][NG];
  auto l0load = [&](__amdgpu_buffer_rsrc_t rs, auto NI) {
    constexpr int n = decltype(NI)::value;
    sfor<NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      if constexpr (n < NRX) r0[n % RING][g] = ldw(rs, voff_x + g * (GW * 4), G::rowr(n) * ROWB);
      else r0[n % RING][g] = ldw(rs, voff_c + g * (GW * 4), (a.XR + KS * (n - NRX)) * ROWB);
    });
  };
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.L > 1 ? a.wh[0] : a.ws_w0p), 0, H * ROWB, 0x00020000);

  acc_t P[NB], Q[NB];
  const int n_wt = (int)((a.B + TR - 1) / TR);

  // Weight operands never depend on the data, so the first ring entries of Dense_0 and of the first hidden layer
  // are fetched one step ahead (before the previous step's Dense_L) and stay valid across steps and tiles.
  sfor<RING - 1>([&](auto NI) { l0load(rs0, NI); });
  if (a.L > 1) sfor<RING - 1>([&](auto KI) { hload(rs1, KI); });

  // EXT 3: a state's particles fill `group` CONSECUTIVE tiles, which one wave takes in turn (the running maximum over
  // the state's candidates stays in its registers); every other build: one tile per turn
  const int n_tg = (HAS_CRITIC) ? ct.group : 1;
  const int n_grp = (n_wt + n_tg - 1) / n_tg;
  float gkey = 0.0f;          // EXT 3: best selection key of the state so far, its particle index and its action
  int gbi = 0;
  float gx[HAS_CRITIC ? NRX : 1];
  sfor<(HAS_CRITIC ? NRX : 1)>([&](auto RI) { gx[decltype(RI)::value] = 0.0f; });

  for (int grp = blockIdx.x * nw + wave; grp < n_grp; grp += gridDim.x * nw)
  for (int tg = 0; tg < n_tg; ++tg) {
    const int wt = grp * n_tg + tg;
    if (wt >= n_wt) break;
    const long long chain = (long long)wt * TR + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    float xs[NRX];      // chain state x^T: register r holds action row a = row(r, s)
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = G::rowr(r) + 4 * ls;
      xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f;
    });
    float cb[CKMAX];    // cond operands: k-step c contracts cond features KS c + s
    sfor<CKMAX>([&](auto CI) {
      constexpr int c = decltype(CI)::value;
      const int k = KS * c + ls;
      cb[c] = (valid && k < a.C) ? a.cond[state * a.C + k] : 0.0f;
    });

    if constexpr (EXT != 0) {
      if (zrow_lds) {      // wave-private: DS operations of one wave execute in order, no barrier
        for (int o = lane * 4; o < H; o += 256)
          *(f32x4*)(s_zrow + wave * H + o) = *(const f32x4*)(a.state_bias + state * H + o);
      }
    }

    // EXT only — dpm++2m: previous x0_hat (scan carry, continuous_ddpm.py:219); dpm2: the step's original xt
    float aux[EXT ? NRX : 1];
    sfor<(EXT ? NRX : 1)>([&](auto RI) { aux[decltype(RI)::value] = 0.0f; });

    // one iteration per network evaluation (= per sampler step, except dpm2: two); EXT 3: then one per critic member on
    // the finished x0 (same Dense_0 / hidden / head code, the member's tables; the operand prefetches chain through)
    // One network evaluation: a sampler step (CRIT false; dpm2: half a step), or — EXT 3, on the finished x0 — critic
    // member ce (CRIT true: the member's tables, activation AC, no update rule).  Each use below is its own copy of the
    // code, so the sampler's loops are exactly those of the EXT 1 build; the operand prefetches chain through from one
    // evaluation to the next (last step -> member 0 -> ... -> the next tile's first step).
    float qmin = 0.0f;
    auto eval_body = [&](auto CRIT_T, auto AK, const int j, const int ce) __attribute__((always_inline)) {
      constexpr bool crit = decltype(CRIT_T)::value;
      constexpr int AC = decltype(AK)::value;
      const bool critn = crit ? (ce + 1 < ct.E) : ((HAS_CRITIC) && j + 1 == a.NE);   // the NEXT evaluation is a critic member
      const int cen = crit ? ce + 1 : 0;
      const int Lj = crit ? ct.L : a.L;
      const int jc = crit ? 0 : j;                          // row of the coefficient / time-bias tables
      __amdgpu_buffer_rsrc_t rs0c = rs0, rs0n = rs0, rslc = rsl;
      if constexpr (HAS_CRITIC) {
        if constexpr (crit) {
          rs0c = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(ct.ptrs[ce * CT_STRIDE + CT_W0P]), 0, ct.w0p_bytes, 0x00020000);
          rslc = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(ct.ptrs[ce * CT_STRIDE + CT_WLQ]), 0, NK * 64 * 4, 0x00020000);
        }
        if (critn) rs0n = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(ct.ptrs[cen * CT_STRIDE + CT_W0P]), 0, ct.w0p_bytes, 0x00020000);
      }
      auto live = [&](int n) { return n < NRX ? n < a.AR : (!crit && (n - NRX) < a.CK); };
      float zr[NRX];
      // EXT 5: in-kernel jax.random.normal(key_j, xt.shape)[chain, ai] (frl_rng.h).  A draw is twelve branch-free micro-steps
      // (frl_rng.h rng_micro); in the even k-steps of the FIRST hidden layer one micro-step follows each MFMA in program
      // order, fenced by a scheduling barrier, so that the matrix pipe covers its ~14 VALU instructions.  (Letting the
      // scheduler place a whole stage inside a k-step put the block in front of the MFMAs: measured, no gain.)  A network
      // without a hidden layer draws up front.
      uint32_t rk0 = 0, rk1 = 0;
      RngDraw rd = {0u, 0u, 0u, 0u, 0.0f, 0.0f, 0.0f, 0.0f};
      bool rused = false;
      if constexpr (HAS_RNG || HAS_LN) {
        if constexpr (HAS_RNG) { rk0 = a.rng_keys[2 * j]; rk1 = a.rng_keys[2 * j + 1]; }
        rused = (HAS_RNG || a.rng_impl != FRL_RNG_NONE) && s_coef[j * COEF_STRIDE + 5] != 0.0f;   // c5 = 0: the last iteration multiplies its draw by 0 (ddpm.py:206)
        if ((HAS_RNG && Lj < 2) || (HAS_LN && a.rng_impl != FRL_RNG_NONE)) {
          sfor<NRX>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            const int ai = G::rowr(r) + 4 * ls;
            zr[r] = (rused && valid && ai < a.A) ? rng_step_normal(a.rng_keys, a.rng_impl, j, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai) : 0.0f;
          });
        }
      }
      auto rng_slot_dense = [&](auto MI) __attribute__((always_inline)) {     // micro-step mm of the lane's ONE dense draw: element
        constexpr int mm = decltype(MI)::value;                               // lane of the tile's TR x A block of normal(key_j, (B, A))
        rng_micro<mm>(rd, rk0, rk1, a.rng_impl, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain - li) * a.A + lane);
        if constexpr (mm == 11) s_zd[lane] = rd.p;
      };
      auto rng_slot = [&](auto MI) __attribute__((always_inline)) {     // micro-step m of the step's NRX draws
        constexpr int m = decltype(MI)::value, r = m / 12, mm = m % 12;
        const int ai = G::rowr(r) + 4 * ls;
        rng_micro<mm>(rd, rk0, rk1, a.rng_impl, (unsigned long long)a.rng_total * a.A, (unsigned long long)(a.rng_off + chain) * a.A + ai);
        if constexpr (mm == 11) zr[r] = (rused && valid && ai < a.A) ? rd.p : 0.0f;
      };
      // ---------------- Dense_0 : P = tb[j] + W0x^T x + W0c^T cond ----------------------------------
      bias_init(P, crit ? s_ctb + ce * H : s_tb + j * H);
      if constexpr (EXT) {
        const float* sbias = a.state_bias;
        if constexpr (HAS_CRITIC && crit) sbias = uniform_ptr(ct.ptrs[ce * CT_STRIDE + CT_ZS]);
        if (sbias != nullptr) {
          // hoisted per-state slice of Dense_0 (the condition / observation rows, computed once per state by
          // frl_state_bias_kernel): every lane adds the row of ITS state; chains of one state share the cache lines
          auto add_row = [&](const float* zrow) __attribute__((always_inline)) {
            sfor<NG>([&](auto GI) {
              constexpr int g = decltype(GI)::value;
              sfor<NR>([&](auto RI) {
                constexpr int r = decltype(RI)::value;
                const f32x4 bv = *(const f32x4*)(zrow + g * GW + 4 * (G::rowr(r) + 4 * ls));
                sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[g * 4 + jj][r] += bv[jj]; });
              });
            });
          };
          if (zrow_lds && !crit) add_row(s_zrow + wave * H);
          else add_row(sbias + (valid ? state : 0) * H);
        }
      }
      sfor<N0>([&](auto NI) {
        constexpr int n = decltype(NI)::value;
        if constexpr (n + RING - 1 < N0) l0load(rs0c, std::integral_constant<int, n + RING - 1>{});
        if (live(n)) {
          float bop;
          if constexpr (n < NRX) bop = xs[n]; else bop = cb[n - NRX];
          sfor<NG>([&](auto GI) {
            constexpr int g = decltype(GI)::value;
            const f32x4 aw = r0[n % RING][g];
            sfor<4>([&](auto JI) { constexpr int jj = decltype(JI)::value; P[g * 4 + jj] = G::mfma(aw[jj], bop, P[g * 4 + jj]); });
          });
        }
      });

      // noise for this step and the first Dense_L operands: issued now, consumed after the hidden layers
      if (!HAS_RNG && !(HAS_LN && a.rng_impl != FRL_RNG_NONE)) {
        sfor<NRX>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int ai = G::rowr(r) + 4 * ls;
          zr[r] = (a.noise != nullptr && !crit && valid && ai < a.A) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
        });
      }
      if constexpr (HAS_LN) { if (a.ws_ln != nullptr) ln_pass(P, s_ln); }
      constexpr int NLD = NK / 4;    // Dense_L: 16-byte loads, load q covers k-steps 4q .. 4q+3
      f32x4 rl[RINGL];
      sfor<RINGL - 1>([&](auto QI) { constexpr int q = decltype(QI)::value; rl[q] = ldw(rslc, voff_l, q * 1024); });

      // ---------------- hidden Dense_l : Q = b_l + W_l^T act(P) ; P <- Q ----------------------------
      auto hidden_layer = [&](const int l, auto RNG_T) __attribute__((always_inline)) {
        constexpr int rng_mode = (int)decltype(RNG_T)::value;       // 0 no draws, 1 one draw per x register, 2 one dense draw per lane
        constexpr bool with_rng = rng_mode != 0;
        // operands of this layer's first RING-1 k-steps are already in flight; the tail prefetches the next
        // layer's — after the last hidden layer that is the FIRST hidden layer of the next evaluation (next step /
        // critic member / tile)
        const float* wc = pick_ptr_rs(a.wh, l - 1);
        const float* wn = pick_ptr_rs(a.wh, (l + 1 < Lj) ? l : 0);
        const float* sb = s_bh + (l - 1) * H;
        if constexpr (HAS_CRITIC) {
          if constexpr (crit) {
            wc = uniform_ptr(ct.ptrs[ce * CT_STRIDE + CT_WH + l - 1]);
            sb = s_cbh + (ce * (ct.L - 1) + l - 1) * H;
            if (l + 1 < Lj) wn = uniform_ptr(ct.ptrs[ce * CT_STRIDE + CT_WH + l]);
          }
          if (l + 1 >= Lj) wn = critn ? uniform_ptr(ct.ptrs[cen * CT_STRIDE + CT_WH]) : a.wh[0];
        }
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)wc, 0, H * ROWB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)wn, 0, H * ROWB, 0x00020000);
        bias_init(Q, sb);
        {
          // activations are computed two at a time on the even/odd register pair (2m, 2m+1) (an aligned pair, so
          // the packed ops need no moves), one k-step before the first of the two is consumed
          float bop, bn1, bn2 = 0.0f;
          { const f32x2 y = rs_act2<AC>(f32x2{P[0][0], P[0][1]}); bop = y[0]; bn1 = y[1]; }
          sfor<NK>([&](auto KI) {
            constexpr int k = decltype(KI)::value;
            if constexpr (k + RING - 1 < NK) hload(rs, std::integral_constant<int, k + RING - 1>{});
            else hload(rsn, std::integral_constant<int, k + RING - 1>{});
            if constexpr ((k & 1) == 1 && k + 2 < NK) {
              const f32x2 y = rs_act2<AC>(f32x2{P[(k + 1) / NR][(k + 1) % NR], P[(k + 2) / NR][(k + 2) % NR]});
              bn1 = y[0]; bn2 = y[1];
            }
            constexpr int NSLOT = 4 * NG;       // MFMAs per k-step
            constexpr int NMICRO = (rng_mode == 2) ? 12 : 12 * NRX;
            constexpr bool rng_here = with_rng && (k & 1) == 0 && k >= 2 && ((k - 2) / 2) * NSLOT < NMICRO;
            sfor<NG>([&](auto GI) {
              constexpr int g = decltype(GI)::value;
              const f32x4 aw = ring[k % RING][g];
              sfor<4>([&](auto JI) {
                constexpr int jj = decltype(JI)::value;
                Q[g * 4 + jj] = G::mfma(aw[jj], bop, Q[g * 4 + jj]);
                if constexpr (rng_here) {
                  constexpr int m = ((k - 2) / 2) * NSLOT + g * 4 + jj;
                  if constexpr (m < NMICRO) {
                    if constexpr (rng_mode == 2) rng_slot_dense(std::integral_constant<int, m>{});
                    else rng_slot(std::integral_constant<int, m>{});
                  }
                  if constexpr (g * 4 + jj > 0) __builtin_amdgcn_sched_barrier(0);     // MFMA i, then micro-step i: as written
                }
              });
            });
            // issue order inside the k-step: first MFMA, then the operand loads (RING-1 k-steps ahead)
            if constexpr (!rng_here) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
              __builtin_amdgcn_sched_group_barrier(0x020, NG, 0);   // VMEM reads
            }
            __builtin_amdgcn_sched_barrier(0);   // one scheduling region per k-step (the group solver is superlinear)
            bop = bn1; bn1 = bn2;
          });
        }
        sfor<NB>([&](auto BI) { constexpr int b = decltype(BI)::value; P[b] = Q[b]; });
        if constexpr (HAS_LN) { if (a.ws_ln != nullptr) ln_pass(P, s_ln + l * 2 * H); }
      };
      if constexpr (HAS_RNG) {
        if (Lj > 1) hidden_layer(1, std::integral_constant<int, DENSE ? 2 : 1>{});
        for (int l = 2; l < Lj; ++l) hidden_layer(l, std::false_type{});
        if constexpr (DENSE) {
          if (Lj > 1) {       // the dense draws back to their owners (same wave wrote them: LDS operations of a wave stay in order)
            sfor<NRX>([&](auto RI) {
              constexpr int r = decltype(RI)::value;
              const int ai = G::rowr(r) + 4 * ls;
              const bool own = rused && valid && ai < a.A;
              const float zv = s_zd[own ? li * a.A + ai : 0];
              zr[r] = own ? zv : 0.0f;
            });
          }
        }
      } else {
        for (int l = 1; l < Lj; ++l) hidden_layer(l, std::false_type{});
      }

      // next step's (or next tile's) first Dense_0 operands: in flight under Dense_L and the epilogue
      sfor<RING - 1>([&](auto NI) { l0load(rs0n, NI); });

      // ---------------- Dense_L : O = b_L + W_L^T act(P), padded to TR rows --------------------------
      acc_t O;
      sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; O[r] = (crit ? s_cbl + ce * LAST_PAD : s_bl)[G::rowr(r) + 4 * ls]; });
      {
        float bop, bn1, bn2 = 0.0f;
        { const f32x2 y = rs_act2<AC>(f32x2{P[0][0], P[0][1]}); bop = y[0]; bn1 = y[1]; }
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr ((k & 3) == 0 && k / 4 + RINGL - 1 < NLD)
            rl[(k / 4 + RINGL - 1) % RINGL] = ldw(rslc, voff_l, (k / 4 + RINGL - 1) * 1024);
          if constexpr ((k & 1) == 1 && k + 2 < NK) {
            const f32x2 y = rs_act2<AC>(f32x2{P[(k + 1) / NR][(k + 1) % NR], P[(k + 2) / NR][(k + 2) % NR]});
            bn1 = y[0]; bn2 = y[1];
          }
          O = G::mfma(rl[(k / 4) % RINGL][k & 3], bop, O);
          bop = bn1; bn1 = bn2;
        });
      }

      // ---------------- epilogue: one reverse-diffusion / Euler step ---------------------------------
      if constexpr (HAS_CRITIC && crit) {
        // critic member ce: its scalar output is row 0 of the head = register 0 of lanes 0 .. 31; qs.min(axis=0), dql.py:155
        qmin = (ce == 0) ? O[0] : fminf(qmin, O[0]);
        return;
      }
      const float c0 = s_coef[jc * COEF_STRIDE + 0], c1 = s_coef[jc * COEF_STRIDE + 1];
      const float c2 = s_coef[jc * COEF_STRIDE + 2], c3 = s_coef[jc * COEF_STRIDE + 3];
      const float c4 = s_coef[jc * COEF_STRIDE + 4], c5 = s_coef[jc * COEF_STRIDE + 5];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        const float eps = O[r], xt = xs[r];
        const float c6 = EXT ? s_coef[jc * COEF_STRIDE + 6] : 0.0f;   // dpm2: 1 on the second evaluation of a step
        if (a.xt_hist != nullptr && in && c6 == 0.0f) {           // history = (xt, eps_theta) of the step, :217
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
        } else if (EXT && a.mode == MODE_CDDPM_DPMPP2M) {
          float x0h = (xt - c1 * eps) / c0;                         // :165
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          const float d = c4 * x0h - c5 * aux[EXT ? r : 0];         // :207-211 (c4 = 1, c5 = 0 on the first-order step)
          xn = c2 * xt - c3 * d;                                    // :212
          aux[EXT ? r : 0] = x0h;                                   // carry, :217
        } else if (EXT && a.mode == MODE_EVAL) {
          xn = eps;                                                 // plain evaluation: the outputs themselves
        } else if (EXT && a.mode == MODE_CDDPM_DPM2) {
          const float xo = (c6 == 0.0f) ? xt : aux[EXT ? r : 0];    // both x0_hats come from the ORIGINAL xt, :165,:194
          float x0h = (xo - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;                                 // :187 (to the midpoint) / :196 (full step)
          aux[EXT ? r : 0] = xo;
        } else {
          xn = xt + eps / c0;                                       // cnf.py:108
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        }
        xs[r] = (EXT && a.mode == MODE_EVAL) ? ((valid && ai < a.AO) ? xn : 0.0f) : (in ? xn : 0.0f);
      });
    };
    for (int j = 0; j < a.NE; ++j) eval_body(std::false_type{}, std::integral_constant<int, ACT>{}, j, 0);
    if constexpr (HAS_CRITIC) {
      // EXT 4: the critic's activation is relu beside another sampler activation (QVPO / DACER: mish actor, relu critic)
      constexpr int CACT = (EXT == 4) ? (int)FRL_ACT_RELU : ACT;
      for (int e = 0; e < ct.E; ++e) eval_body(std::true_type{}, std::integral_constant<int, CACT>{}, 0, e);
    }

    if (!HAS_CRITIC || a.x0 != nullptr) {
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        if (valid && ai < a.AO) a.x0[chain * a.AO + ai] = xs[r];
      });
    }

    if constexpr (HAS_CRITIC) {
      // ---------------- best-of-N: argmax (or Gumbel-max categorical) over the state's particles, dql.py:156-161 ----------
      // A tile holds 32 / seg whole states of seg particles each (seg a power of two), or one 32-particle slice of a state.
      const int seg = ct.seg;
      float key = qmin;
      if (ct.gumbel != nullptr) key = fmaf(qmin, ct.inv_temperature, (valid && ls == 0) ? ct.gumbel[chain] : 0.0f);
      if (ct.q != nullptr && valid && ls == 0) ct.q[chain] = qmin;
      key = __shfl(key, li);                                  // lanes 32 .. 63 take the key lanes 0 .. 31 hold
      if (!valid) key = -__builtin_huge_valf();
      int bi = valid ? tg * TR + (li & (seg - 1)) : 0x7fffffff;
      for (int m = 1; m < seg; m <<= 1) {                     // larger key wins, equal keys -> the lower index (jnp.argmax)
        const float ob = __shfl_xor(key, m);
        const int oi = __shfl_xor(bi, m);
        if (ob > key || (ob == key && oi < bi)) { key = ob; bi = oi; }
      }
      const int seg0 = (li & ~(seg - 1)) + TR * ls;           // first lane of the segment in this half: its verdict stands
      key = __shfl(key, seg0);
      bi = __shfl(bi, seg0);
      const int wl = valid ? (li & ~(seg - 1)) + ((bi - tg * TR) & (TR - 1)) : li;
      const bool better = (tg == 0) || key > gkey;            // strict >: the earlier slice keeps a tie
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const float xw = __shfl(xs[r], wl + TR * ls);         // the winner's action rows, each half its own
        if (better) gx[r] = xw;
      });
      if (better) { gkey = key; gbi = bi; }
      if ((tg == n_tg - 1 || wt == n_wt - 1) && valid && (li & (seg - 1)) == 0) {
        if (ls == 0 && ct.best_idx != nullptr) ct.best_idx[state] = gbi;
        if (ct.best_action != nullptr) {
          sfor<NRX>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            const int ai = G::rowr(r) + 4 * ls;
            if (ai < a.A) ct.best_action[state * a.A + ai] = gx[r];
          });
        }
      }
    }
  }
}

// One wave per SIMD, 512 registers per lane: every geometry.
template <int TR, int H, int ACT, int NRX, int CKMAX, int EXT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_rs_kernel(const KArgs a) {
  const CriticTab none{};
  sampler_rs_body<TR, H, ACT, NRX, CKMAX, EXT>(a, none);
}

// The same with the critic tail (EXT 3): sampler + scoring + selection in one launch.
template <int TR, int H, int ACT, int NRX, int CKMAX, int CRELU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_rs_fused_kernel(const KArgs a, const CriticTab ct) {
  sampler_rs_body<TR, H, ACT, NRX, CKMAX, 3 + CRELU>(a, ct);
}

// Two waves per SIMD, 256 registers per lane: the 16-chain geometry of the plain sampler (EXT 0: 249 registers, no
// spills; the EXT 1 / 2 paths do not fit) and its keyed variant (EXT 6: 256 registers, 47 dwords of spills, all but a dozen
// reloads per step outside the MFMA loops — here the OTHER wave's MFMAs cover a wave's draws, which one wave per SIMD cannot).  Same body, same results per tile; the second wave fills the issue slots
// the first leaves between its MFMAs and ring loads (+2 .. 2.6 % at cfg2, same-box A/B in DESIGN.md 5.2).  Picked by
// plan_two_wave (frl_internal.h) for batches of >= 65536 chains.
template <int ACT, int RNG>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
frl_sampler_rs_kernel_w2(const KArgs a) {
  const CriticTab none{};
  sampler_rs_body<16, 256, ACT, 4, 12, RNG ? 6 : 0>(a, none);
}

// ---- launcher --------------------------------------------------------------------------------------
static int device_cus_rs() {
  static thread_local int cus[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16) dev = 0;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cus[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[dev];
}

// `configured`: the launching instantiation's per-device record of the dynamic-LDS size its kernel was last set up for
static int launch_kern(const void* kern, const char* tag, size_t* configured, int TR, int H, int max_waves, const KArgs& a, hipStream_t stream,
                       const CriticTab* ct = nullptr) {
  const int ncu = device_cus_rs();
  const long long n_tiles = (a.B + TR - 1) / TR;
  const long long n_wt = ct != nullptr ? (n_tiles + ct->group - 1) / ct->group : n_tiles;     // units a wave takes in turn
  long long per_cu = (n_wt + ncu - 1) / ncu;
  int nw = (int)(per_cu < 1 ? 1 : (per_cu > max_waves ? max_waves : per_cu));
  if (nw == 3) nw = 4;
  long long nwg = (n_wt + nw - 1) / nw;
  const int grid = (int)(nwg < ncu ? nwg : ncu);
  const size_t lds = sizeof(float) * ((size_t)a.NE * H + (size_t)(a.L - 1) * H + LAST_PAD + (size_t)a.NE * COEF_STRIDE +
                                      (a.ws_ln != nullptr ? (size_t)a.L * 2 * H : 0) +
                                      ((a.state_bias != nullptr && a.particles % TR == 0) ? (size_t)nw * H : 0) +
                                      ((a.rng_impl != FRL_RNG_NONE && TR == 16 && H == 256) ? (size_t)nw * 64 : 0) +     // keyed two-wave build: dense draws
                                      (ct != nullptr ? (size_t)ct->E * ((size_t)ct->L * H + LAST_PAD) : 0));
  if (lds > 160 * 1024) { set_error("sampler: %d network evaluations per chain need %zu B of LDS (> 160 KiB)", a.NE, lds); return FRL_EUNSUPPORTED; }
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && configured[dev] < lds) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
    configured[dev] = lds;
  }
  KArgs args = a;
  CriticTab cargs;
  if (ct != nullptr) cargs = *ct;
  void* argv[] = {(void*)&args, (void*)&cargs};
  hipError_t e = hipLaunchKernel(kern, dim3(grid), dim3(64 * nw), argv, lds, stream);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) { set_error("sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("%s [grid %d x %d waves]", tag, grid, nw);
  return FRL_OK;
}

template <int TR, int H, int ACT, int NRX, int CKMAX, int EXT>
static int launch_inst_v(const KArgs& a, hipStream_t stream) {
  static thread_local size_t configured[16] = {0};
  char tag[96];
  snprintf(tag, sizeof(tag), "frl_sampler_rs_kernel<TR %d, H %d, act %d, NRX %d, CKMAX %d, EXT %d>", TR, H, ACT, NRX, CKMAX, EXT);
  return launch_kern((const void*)frl_sampler_rs_kernel<TR, H, ACT, NRX, CKMAX, EXT>, tag, configured, TR, H, MAX_WAVES, a, stream);
}

// the two-wave build: up to eight waves per workgroup (two per SIMD), plain sampler only (plan_two_wave guarantees it)
template <int ACT>
static int launch_w2(const KArgs& a, hipStream_t stream) {
  if constexpr (ACT == FRL_ACT_IDENTITY) { set_error("two-wave build: no identity-activation unit"); return FRL_EINVAL; }
  else {
    if (a.ws_ln != nullptr || a.mode > MODE_CNF_EULER || a.state_bias != nullptr || a.AR > 4 || a.CK > 12 ||
        (a.rng_impl != FRL_RNG_NONE && 16 * a.A > 64)) {
      set_error("two-wave build asked for a path it does not carry (plan / launch mismatch)"); return FRL_EINVAL; }
    static thread_local size_t configured[2][16] = {{0}, {0}};
    char tag[64];
    const int keyed = a.rng_impl != FRL_RNG_NONE;
    snprintf(tag, sizeof(tag), "frl_sampler_rs_kernel_w2<act %d, draws %d>", ACT, keyed);
    return launch_kern(keyed ? (const void*)frl_sampler_rs_kernel_w2<ACT, 1> : (const void*)frl_sampler_rs_kernel_w2<ACT, 0>, tag, configured[keyed], 16, 256,
                       2 * MAX_WAVES, a, stream);
  }
}

// Three builds of every geometry: the plain one (ddpm / ddim / Euler, condition contracted in the kernel) keeps the
// register budget of the tuned hot loop; EXT 1 adds what the rarer paths need (dpm2 / dpm++2m carry registers, plain
// evaluation for critics, per-state hoisted condition rows); EXT 2 adds the in-kernel threefry draws and the LayerNorm
// pass, so that the hoisted cfg4 path does not carry their code either.
template <int TR, int H, int ACT, int NRX, int CKMAX>
static int launch_inst(const KArgs& a, hipStream_t stream) {
  if constexpr (ACT == FRL_ACT_IDENTITY) {     // the identity-activation unit only serves post-activation LayerNorm (EXT 2)
    if (a.ws_ln == nullptr || a.ln_post_act == 0) { set_error("identity-activation build without a post-activation LayerNorm"); return FRL_EINVAL; }
    return launch_inst_v<TR, H, ACT, NRX, CKMAX, 2>(a, stream);
  }
  if (a.ws_ln != nullptr) return launch_inst_v<TR, H, ACT, NRX, CKMAX, 2>(a, stream);
  if (a.rng_impl != FRL_RNG_NONE) return launch_inst_v<TR, H, ACT, NRX, CKMAX, 5>(a, stream);
  if constexpr (ACT != FRL_ACT_IDENTITY) {
    if (a.mode > MODE_CNF_EULER || a.state_bias != nullptr) return launch_inst_v<TR, H, ACT, NRX, CKMAX, 1>(a, stream);
    return launch_inst_v<TR, H, ACT, NRX, CKMAX, 0>(a, stream);
  }
  return FRL_EINVAL;
}

template <int ACT>
int launch_sampler_rs_act(const KArgs& a, hipStream_t stream) {
  if (a.H == 256 && a.TR == 16) return launch_w2<ACT>(a, stream);
#ifndef FRL_RS_SKIP_A
  if (a.H == 256 && a.AR <= 4 && a.CK <= 12) return launch_inst<32, 256, ACT, 4, 12>(a, stream);
#endif
#ifndef FRL_RS_SKIP_B
  // wide x with the condition rows hoisted per state (DACER / QVPO acting: CK = 0): no 64-register cond operand array
  if (a.H == 256 && a.AR <= 12 && a.CK == 0) return launch_inst<32, 256, ACT, 12, 1>(a, stream);
  if (a.H == 256 && a.AR <= 12 && a.CK <= 64) return launch_inst<32, 256, ACT, 12, 64>(a, stream);
#endif
#ifndef FRL_RS_SKIP_C
  if (a.H == 512 && a.AR <= 4 && a.CK <= 12) return launch_inst<16, 512, ACT, 4, 12>(a, stream);
#endif
  set_error("no compiled kernel for hidden width %d with x_dim %d / cond_dim %d", a.H, a.A, a.C);
  return FRL_EUNSUPPORTED;
}

// sampler + critic scoring + selection in one launch (frl_sampler_select_fwd): the two 256-wide 32-chain geometries
template <int TR, int H, int ACT, int NRX, int CKMAX, int CRELU>
static int launch_fused_vr(const KArgs& a, const CriticTab& ct, hipStream_t stream) {
  static thread_local size_t configured[16] = {0};
  char tag[128];
  snprintf(tag, sizeof(tag), "frl_sampler_rs_fused_kernel<TR %d, H %d, act %d, NRX %d, CKMAX %d, critic relu %d> [%d critics x %d layers]", TR, H, ACT, NRX, CKMAX, CRELU, ct.E, ct.L);
  return launch_kern((const void*)frl_sampler_rs_fused_kernel<TR, H, ACT, NRX, CKMAX, CRELU>, tag, configured, TR, H, MAX_WAVES, a, stream, &ct);
}
template <int TR, int H, int ACT, int NRX, int CKMAX>
static int launch_fused_v(const KArgs& a, const CriticTab& ct, hipStream_t stream) {
  if constexpr (ACT != FRL_ACT_RELU) { if (ct.relu) return launch_fused_vr<TR, H, ACT, NRX, CKMAX, 1>(a, ct, stream); }
  return launch_fused_vr<TR, H, ACT, NRX, CKMAX, 0>(a, ct, stream);
}

template <int ACT>
int launch_sampler_rs_fused_act(const KArgs& a, const CriticTab& ct, hipStream_t stream) {
  if constexpr (ACT == FRL_ACT_IDENTITY) { set_error("fused critic: no identity-activation unit"); return FRL_EINVAL; }
  else {
    if (a.H != 256 || a.TR != 32 || a.rng_impl != FRL_RNG_NONE || a.ws_ln != nullptr || a.L < 2 || ct.L < 2 || ct.E < 1 || ct.E > FUSE_MAX_E ||
        ct.L > FUSE_MAX_L || ct.seg < 1 || ct.seg > 32 || (ct.seg & (ct.seg - 1)) != 0 || ct.group < 1 || (ct.group > 1 && ct.seg != 32)) {
      set_error("fused critic: launch arguments outside the build (plan / launch mismatch)"); return FRL_EINVAL; }
    if (a.AR <= 4 && a.CK <= 12) return launch_fused_v<32, 256, ACT, 4, 12>(a, ct, stream);
    if (a.AR <= 12 && a.CK == 0) return launch_fused_v<32, 256, ACT, 12, 1>(a, ct, stream);
    set_error("fused critic: x_dim %d with %d in-kernel condition rows has no build (wide actions need the per-state hoist)", a.A, a.C);
    return FRL_EUNSUPPORTED;
  }
}

#ifndef FRL_RS_ACT
#error "compile with -DFRL_RS_ACT=<FRL_ACT_*>: one translation unit per activation"
#endif
template int launch_sampler_rs_act<FRL_RS_ACT>(const KArgs&, hipStream_t);
template int launch_sampler_rs_fused_act<FRL_RS_ACT>(const KArgs&, const CriticTab&, hipStream_t);

}  // namespace frl
