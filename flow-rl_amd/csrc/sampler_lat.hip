// Latency build of the fused T-step sampler (exact-f32 MFMA 32x32x2): ONE workgroup of four waves per tile of 32 chains,
// the OUTPUT FEATURES of every layer split over the four SIMDs of a CU.
//
// The throughput kernel (sampler_rs.hip) gives a wave a whole tile: no barrier, no LDS traffic, 84 % of the MFMA peak
// once there are >= 1024 wave-tiles — but a tile's step is then 2256 serial MFMAs (145k cycles), and a small batch
// (acting: agent.sample_actions on 1 .. a few thousand observations, examples/toy2d: 4096 chains x T = 20) leaves most
// SIMDs idle behind that latency.  Here wave w owns features [H/4 w, H/4 (w + 1)) of every layer:
//
//   Dense_0 / hidden layers   each wave contracts ALL inputs for ITS H/4 output features (weights: 8- or 16-byte loads of
//                             its own columns, streamed L2 -> registers through a ring); the activated outputs go to an
//                             LDS tile in the B-operand order of the next layer ([k-step/4][lane][4]: one ds_read_b128
//                             per four k-steps), one barrier per layer, two tiles ping-pong
//   Dense_L                   split-K: a wave's own activated features ARE B operands already (register r of lane half s
//                             <-> one feature, as in sampler_rs.hip); the four partial 32 x 32 tiles meet in LDS
//   epilogue                  every wave applies the update to its own copy of the chain state (same code as sampler_rs)
//
// A step is then ~1/4 of the serial MFMA chain plus L + 1 barriers.  Same arithmetic per output as the throughput kernel
// up to the order of the Dense_L partial sums.  Chosen by the launcher when the batch has no more tiles than the GPU
// has CUs (FrlDesc.tile_policy overrides).
#include "frl_device.h"
#include "frl_rng.h"
#include "frl_actgrad.h"

namespace frl {

template <int NBW> struct WVec;
template <> struct WVec<2> {
  typedef f32x2 type;
  static __device__ __forceinline__ type ld(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
  }
};
template <> struct WVec<4> {
  typedef f32x4 type;
  static __device__ __forceinline__ type ld(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
  }
};

// STASH: the kernel is the TRAINING forward of a small batch (frl_sampler_fwd_train / frl_net_eval_train below the
// throughput regime): it also writes h_l = act(z_l) and act'(z_l) of every layer and step where the chain VJP's backward
// sweep and dW kernel read them (sampler_bwd.hip: block (tile-step, layer) of 64 pair rows x 64 lanes x 2 floats, k-step
// k = 16 block + r of the one-wave-per-tile layout on pair row k / 2, element k & 1).  This wave's register (jb, r) of lane
// (li, ls) is feature 64 w + 2 (ROWR(r) + 4 ls) + jb, which that layout keeps at
//     k = 64 (w >> 1) + 32 (r & 1) + 16 jb + ((r >> 1) & 1) + 2 ls + 8 (w & 1) + 4 (r >> 3),   lane li + 32 ((r >> 2) & 1)
// so registers r and r | 2 are the two floats of ONE 8-byte stash element: a per-lane base plus compile-time offsets.
template <int TR, int H, int ACT, int NRX, int CKMAX, bool STASH>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
frl_sampler_lat_kernel(const KArgs a) {
  static_assert(!STASH || H == 256, "the stash layout is the 256-wide one");
  // 16-chain tiles (TR = 16) write the SAME 32-chain stash blocks: tile t is chains 16 (t & 1) .. + 15 of wave-tile t >> 1, and
  // register (jb, r) of lane (li, ls) — feature 64 w + 4 (r + 4 ls) + jb — sits at k-step 16 (4 (w >> 1) + jb) + r + 8 (w & 1)
  // + 4 (ls >> 1) of lane 16 (t & 1) + li + 32 (ls & 1): registers r and r + 1 (r even) are one 8-byte element.
  typedef Geo<TR> G;                // 32: MFMA 32x32x2, 16: MFMA 16x16x4 (tiles of 16 chains: half the latency again)
  typedef typename G::acc_t acc_t;
  constexpr int NR = G::NR, KS = G::KS;
  constexpr int NBW = H / (4 * TR); // TR-row output blocks per wave
  constexpr int FW = H / 4;         // features per wave
  constexpr int NK = H / KS;        // k-steps per hidden layer
  constexpr int NOT = (NRX + NR - 1) / NR;   // TR-row output tiles of Dense_L: two only on 16-chain tiles with x_dim > 16 (DACER / QVPO humanoid: 17)
  static_assert(NBW % KS == 0 && NRX <= 2 * NR, "geometry");
  constexpr int ROWB = H * 4;
  constexpr int RW = (TR == 16) ? 32 : 16;   // hidden-layer weight ring (k-steps ahead): lookahead of ~2000 MFMA cycles either way
  constexpr int N0 = NRX + CKMAX;            // Dense_0 k-step slots
  constexpr int R0 = N0 < 16 ? N0 : 16;      // Dense_0 weight ring: requested a whole step ahead (below), so its L2 latency
                                             // never sits between the epilogue of one step and the first MFMA of the next
  typedef WVec<NBW> WV;
  typedef typename WV::type wv_t;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane % TR, ls = lane / TR;

  float* s_tb = (float*)smem;                         // [NE][H] (absent when a.lat_tb_global)
  float* s_bh = s_tb + (a.lat_tb_global ? 0 : a.NE * H);   // [L-1][H]
  float* s_bl = s_bh + (a.L - 1) * H;                 // [32]
  float* s_coef = s_bl + LAST_PAD;                    // [NE][8]
  float* s_act = s_coef + a.NE * COEF_STRIDE;         // 2 x [H/8][64][4]: activations in B-operand order
  // The two s_act buffers alternate per PHASE (a hidden layer's activation exchange, or the Dense_L partial tiles
  // [4 waves][16][64], which fit one buffer): phase n writes buffer n & 1, barrier, reads it.  The other buffer was
  // last read in phase n - 1, which every wave left before it passed barrier n: one barrier per phase is enough.
  float* s_stat = s_act + 2 * H * TR;                 // LayerNorm only: [2][4 waves][TR] per-wave partial sums
  float* s_ln = s_stat + 256;                         // LayerNorm only: [L][2][H] scale | bias rows
  if (a.ws_ln != nullptr)
    for (int idx = tid; idx < a.L * 2 * H; idx += 256) s_ln[idx] = a.ws_ln[idx];
  if (!a.lat_tb_global)
    for (int idx = tid; idx < a.NE * H; idx += 256) s_tb[idx] = a.ws_tb[idx];
  for (int l = 1; l < a.L; ++l) {
    const float* bp = pick_ptr_rs(a.bh, l - 1);
    for (int idx = tid; idx < H; idx += 256) s_bh[(l - 1) * H + idx] = bp[idx];
  }
  if (tid < LAST_PAD) s_bl[tid] = a.ws_blp[tid];
  for (int idx = tid; idx < a.NE * COEF_STRIDE; idx += 256) s_coef[idx] = a.ws_coef[idx];
  __syncthreads();

  // column part of every weight address: this wave's features FW w + NBW i + jb, i = li
  const unsigned voff_col = (unsigned)(FW * wave + NBW * li) * 4;
  const unsigned voff_h = ls * ROWB + voff_col;         // hidden / cond rows: k-step's row 2 ks + ls
  const unsigned voff_x = ls * (4 * ROWB) + voff_col;   // Dense_0 x rows: ROWR(n) + 4 ls
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.ws_w0p, 0, a.w0p_bytes, 0x00020000);
  // STASH: this lane's part of the element address (16-chain tiles add 128 bytes for the odd tile of a pair, below)
  const unsigned voff_st0 = (TR == 32) ? (unsigned)(32 * (wave >> 1) + 4 * (wave & 1) + ls) * 512 + li * 8
                                       : (unsigned)(32 * (wave >> 1) + 4 * (wave & 1) + 2 * (ls >> 1)) * 512 + (li + 32 * (ls & 1)) * 8;
  // the first register rp of a pair -> compile-time part of the address; its partner is rp + 2 (TR 32) or rp + 1 (TR 16)
  auto stash_off = [](int jb, int rp) { return TR == 32 ? (16 * (rp & 1) + 8 * jb + 2 * (rp >> 3)) * 512 + 256 * ((rp >> 2) & 1) : (8 * jb + (rp >> 1)) * 512; };
  constexpr int PSTEP = (TR == 32) ? 2 : 1;

  // Dense_L operands stay resident: k-step (jb, r) contracts feature FW w + NBW (ROWR(r) + 4 ls) + jb, lane i = output row
  float wl[NOT][NBW][NR];
  {
    const float* WL = a.wl_raw;
    sfor<NOT>([&](auto TI) {
      constexpr int t = decltype(TI)::value;
      sfor<NBW>([&](auto JI) {
        constexpr int jb = decltype(JI)::value;
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const int f = FW * wave + NBW * (G::rowr(r) + 4 * ls) + jb;
          wl[t][jb][r] = (t * TR + li < a.AO) ? WL[(size_t)f * a.AO + t * TR + li] : 0.0f;
        });
      });
    });
  }

  // (STASH on 16-chain tiles: both halves of the last wave-tile are run, so that every stash row holds finite values)
  const int ntiles = (STASH && TR == 16) ? 2 * (int)((a.B + 31) / 32) : (int)((a.B + TR - 1) / TR);
  // Dense_0 operands never depend on the data: the first R0 - 1 k-steps of the NEXT evaluation are requested before
  // Dense_L and the epilogue of the current one (and before the first tile here)
  wv_t r0[R0];
  auto l0load = [&](auto NI) {
    constexpr int n = decltype(NI)::value;
    if constexpr (n < NRX) r0[n % R0] = WV::ld(rs0, voff_x, G::rowr(n) * ROWB);
    else r0[n % R0] = WV::ld(rs0, voff_h, (a.XR + KS * (n - NRX)) * ROWB);
  };
  sfor<R0 - 1>([&](auto NI) { l0load(NI); });
  static_assert(NK % RW == 0, "a layer's tail prefetch must land in the ring slots the next layer reads first");
  wv_t ring[RW];
  if (a.L > 1) {
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.wh[0], 0, H * ROWB, 0x00020000);
    sfor<RW - 1>([&](auto KI) { constexpr int k = decltype(KI)::value; ring[k] = WV::ld(rs1, voff_h, KS * k * ROWB); });
  }
  int phase = 0;   // runs on across steps and tiles
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long chain = (long long)tile * TR + li;
    const bool valid = chain < a.B;
    const long long state = (a.particles == 1) ? chain : chain / a.particles;

    float xs[NRX];
    sfor<NRX>([&](auto RI) {
      constexpr int r = decltype(RI)::value;
      const int ai = G::rowr(r) + 4 * ls;
      xs[r] = (valid && ai < a.A) ? a.xT[chain * a.A + ai] : 0.0f;
    });
    float cb[CKMAX];
    sfor<CKMAX>([&](auto CI) {
      constexpr int c = decltype(CI)::value;
      const int k = KS * c + ls;
      cb[c] = (valid && k < a.C) ? a.cond[state * a.C + k] : 0.0f;
    });
    float aux[NRX];
    sfor<NRX>([&](auto RI) { aux[decltype(RI)::value] = 0.0f; });

    for (int j = 0; j < a.NE; ++j) {
      acc_t acc[NBW];
      const size_t srow = STASH ? ((size_t)(TR == 32 ? tile : tile >> 1) * a.NE + j) * a.L * 8192 : 0;      // floats: this tile-step's L stash blocks
      const unsigned voff_st = voff_st0 + ((TR == 16) ? (unsigned)(tile & 1) * 128 : 0u);
      const __amdgpu_buffer_rsrc_t rs_hd = __builtin_amdgcn_make_buffer_rsrc((void*)(STASH ? a.stash_h + srow : nullptr), 0, STASH ? a.L * 32768 : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_ad = __builtin_amdgcn_make_buffer_rsrc((void*)(STASH ? a.stash_a + srow : nullptr), 0, STASH ? a.L * 32768 : 0, 0x00020000);
      // ---------------- Dense_0 ------------------------------------------------------------------------
      if (a.lat_tb_global) {
        const float* trow = a.ws_tb + (size_t)j * H + FW * wave;
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const wv_t bv = *(const wv_t*)(trow + NBW * (G::rowr(r) + 4 * ls));
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb][r] = bv[jb]; });
        });
      } else {
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const wv_t bv = *(const wv_t*)(s_tb + j * H + FW * wave + NBW * (G::rowr(r) + 4 * ls));
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb][r] = bv[jb]; });
        });
      }
      if (a.state_bias != nullptr) {
        const float* zrow = a.state_bias + (valid ? state : 0) * H + FW * wave;
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const wv_t bv = *(const wv_t*)(zrow + NBW * (G::rowr(r) + 4 * ls));
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb][r] += bv[jb]; });
        });
      }
      float zr[NRX];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        zr[r] = (a.noise != nullptr && valid && ai < a.A) ? a.noise[((long long)j * a.B + chain) * a.A + ai] : 0.0f;
      });
      {
        sfor<N0>([&](auto NI) {
          constexpr int n = decltype(NI)::value;
          if constexpr (n + R0 - 1 < N0) l0load(std::integral_constant<int, n + R0 - 1>{});
          const bool live = (n < NRX) ? (n < a.AR) : ((n - NRX) < a.CK);
          if (live) {
            float bop;
            if constexpr (n < NRX) bop = xs[n]; else bop = cb[n - NRX];
            const wv_t aw = r0[n % R0];
            sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb] = G::mfma(aw[jb], bop, acc[jb]); });
          }
        });
      }

      // flax.linen.LayerNorm of a finished pre-activation tile (mlp.py:26-29), features spread over the four waves: mean,
      // then the variance about it, each a register sum + lane-group exchange + four per-wave partials through LDS
      // (two extra barriers per layer; the partial buffers alternate, so no third one is needed)
      auto ln_pass = [&](int l) {
        if (a.ln_post_act != 0) {      // FQL's order (fql/network.py:26-30): the activation first, in place; consumers are identity
          const int act = a.ln_post_act - 1;
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; acc[jb][r] = act_rt(act, acc[jb][r]); }); });
        }
        // every level of these sums is a fixed pairwise tree (four interleaved register partials, the lane-group
        // butterfly, then the four waves' partials as (w0 + w1) + (w2 + w3)): error growth like numpy's pairwise mean
        float p1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; p1[(jb * NR + r) & 3] += acc[jb][r]; }); });
        float s1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
#pragma unroll
        for (int m = TR; m < 64; m <<= 1) s1 += __shfl_xor(s1, m);
        s_stat[wave * TR + li] = s1;
        __syncthreads();
        const float mean = ((s_stat[li] + s_stat[TR + li]) + (s_stat[2 * TR + li] + s_stat[3 * TR + li])) * (1.0f / H);
        float p2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        sfor<NBW>([&](auto JI) {
          constexpr int jb = decltype(JI)::value;
          sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; const float v = acc[jb][r] - mean; p2[(jb * NR + r) & 3] = fmaf(v, v, p2[(jb * NR + r) & 3]); });
        });
        float s2 = (p2[0] + p2[1]) + (p2[2] + p2[3]);
#pragma unroll
        for (int m = TR; m < 64; m <<= 1) s2 += __shfl_xor(s2, m);
        s_stat[128 + wave * TR + li] = s2;
        __syncthreads();
        const float var = ((s_stat[128 + li] + s_stat[128 + TR + li]) + (s_stat[128 + 2 * TR + li] + s_stat[128 + 3 * TR + li])) * (1.0f / H);
        const float rstd = 1.0f / sqrtf(var + 1e-6f);
        const float* sl = s_ln + l * 2 * H + FW * wave;
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const wv_t sc = *(const wv_t*)(sl + NBW * (G::rowr(r) + 4 * ls));
          const wv_t bi = *(const wv_t*)(sl + H + NBW * (G::rowr(r) + 4 * ls));
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb][r] = (acc[jb][r] - mean) * (rstd * sc[jb]) + bi[jb]; });
        });
      };
      if (a.ws_ln != nullptr) ln_pass(0);

      // ---------------- hidden layers: activations through LDS, one barrier each ------------------------
      for (int l = 1; l < a.L; ++l) {
        float* abuf = s_act + (phase++ & 1) * (H * TR);
        // this layer's first RW - 1 weight k-steps are already in flight (requested by the tail of the previous layer's
        // loop); its own tail requests the next layer's — after the last hidden layer the FIRST one's again (next
        // evaluation / tile) — so no L2 latency sits between a barrier and the first MFMA behind it
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(a.wh, l - 1), 0, H * ROWB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)pick_ptr_rs(a.wh, (l + 1 < a.L) ? l : 0), 0, H * ROWB, 0x00020000);
        // feature f = FW w + NBW row + jb is contracted by k-step ks = f / KS on lane group f % KS (KS divides NBW).
        if constexpr (NBW == KS) {
          // ks = (FW / KS) w + ROWR(r) + 4 s: registers r = 4 q .. 4 q + 3 are the four k-steps of ONE 16-byte group of
          // this lane's column — one conflict-free ds_write_b128 instead of four strided ds_write_b32
          sfor<NBW>([&](auto JI) {
            constexpr int jb = decltype(JI)::value;
            sfor<NR / 4>([&](auto QI) {
              constexpr int q = decltype(QI)::value;
              const int ks = (FW / KS) * wave + G::rowr(4 * q) + 4 * ls;
              f32x4 v;
              if constexpr (STASH) {
                sfor<2>([&](auto EI) {
                  constexpr int e = decltype(EI)::value;
                  constexpr int e0 = (TR == 32) ? e : 2 * e, e1 = e0 + PSTEP;      // the pair's places in this group of four
                  f32x2 h, d;
                  act_grad2<ACT>(f32x2{acc[jb][4 * q + e0], acc[jb][4 * q + e1]}, h, d);
                  v[e0] = h[0]; v[e1] = h[1];
                  st2(h, rs_hd, voff_st, (l - 1) * 32768 + stash_off(jb, 4 * q + e0));
                  st2(d, rs_ad, voff_st, (l - 1) * 32768 + stash_off(jb, 4 * q + e0));
                });
              } else {
                sfor<4>([&](auto EI) { constexpr int e = decltype(EI)::value; v[e] = act_fn<ACT>(acc[jb][4 * q + e]); });
              }
              *(f32x4*)(abuf + (((ks >> 2) * 64) + jb * TR + li) * 4) = v;
            });
          });
        } else {
          sfor<NBW>([&](auto JI) {
            constexpr int jb = decltype(JI)::value;
            sfor<NR>([&](auto RI) {
              constexpr int r = decltype(RI)::value;
              const int ks = (FW / KS) * wave + (NBW / KS) * (G::rowr(r) + 4 * ls) + (jb / KS);
              abuf[(((ks >> 2) * 64) + (jb % KS) * TR + li) * 4 + (ks & 3)] = act_fn<ACT>(acc[jb][r]);
            });
          });
        }
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const wv_t bv = *(const wv_t*)(s_bh + (l - 1) * H + FW * wave + NBW * (G::rowr(r) + 4 * ls));
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb][r] = bv[jb]; });
        });
        __syncthreads();
        const f32x4* bsrc = (const f32x4*)abuf + lane;
        f32x4 bq = bsrc[0], bqn = bq;
        sfor<NK>([&](auto KI) {
          constexpr int k = decltype(KI)::value;
          if constexpr (k + RW - 1 < NK) ring[(k + RW - 1) % RW] = WV::ld(rs, voff_h, KS * (k + RW - 1) * ROWB);
          else ring[(k + RW - 1) % RW] = WV::ld(rsn, voff_h, KS * (k + RW - 1 - NK) * ROWB);
          if constexpr ((k & 3) == 0 && k + 4 < NK) bqn = bsrc[(k / 4 + 1) * 64];
          const wv_t aw = ring[k % RW];
          const float bop = bq[k & 3];
          sfor<NBW>([&](auto JI) { constexpr int jb = decltype(JI)::value; acc[jb] = G::mfma(aw[jb], bop, acc[jb]); });
          if constexpr ((k & 3) == 3) bq = bqn;
        });
        if (a.ws_ln != nullptr) ln_pass(l);
      }

      sfor<R0 - 1>([&](auto NI) { l0load(NI); });   // next evaluation's (or tile's) Dense_0 operands

      // ---------------- Dense_L, split-K over the waves ------------------------------------------------------
      acc_t O[NOT];
      sfor<NOT>([&](auto TI) { sfor<NR>([&](auto RI) { O[decltype(TI)::value][decltype(RI)::value] = 0.0f; }); });
      if constexpr (STASH) {
        sfor<NBW>([&](auto JI) {
          constexpr int jb = decltype(JI)::value;
          sfor<NR / 2>([&](auto MI) {
            constexpr int m = decltype(MI)::value;
            constexpr int rp = (TR == 32) ? 4 * (m >> 1) + (m & 1) : 2 * m;      // registers rp and rp + PSTEP share a stash element
            f32x2 h, d;
            act_grad2<ACT>(f32x2{acc[jb][rp], acc[jb][rp + PSTEP]}, h, d);
            st2(h, rs_hd, voff_st, (a.L - 1) * 32768 + stash_off(jb, rp));
            st2(d, rs_ad, voff_st, (a.L - 1) * 32768 + stash_off(jb, rp));
            sfor<NOT>([&](auto TI) {
              constexpr int t = decltype(TI)::value;
              O[t] = G::mfma(wl[t][jb][rp], h[0], O[t]);
              O[t] = G::mfma(wl[t][jb][rp + PSTEP], h[1], O[t]);
            });
          });
        });
      } else {
      sfor<NBW>([&](auto JI) {
        constexpr int jb = decltype(JI)::value;
        sfor<NR>([&](auto RI) {
          constexpr int r = decltype(RI)::value;
          const float hv = act_fn<ACT>(acc[jb][r]);
          sfor<NOT>([&](auto TI) { constexpr int t = decltype(TI)::value; O[t] = G::mfma(wl[t][jb][r], hv, O[t]); });
        });
      });
      }
      {
        float* part = s_act + (phase++ & 1) * (H * TR);
        sfor<NOT>([&](auto TI) {
          constexpr int t = decltype(TI)::value;
          sfor<NR>([&](auto RI) { constexpr int r = decltype(RI)::value; part[((t * 4 + wave) * NR + r) * 64 + lane] = O[t][r]; });
        });
        __syncthreads();
        sfor<NOT>([&](auto TI) {
          constexpr int t = decltype(TI)::value;
          sfor<NR>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            float s = s_bl[t * TR + G::rowr(r) + 4 * ls];
            sfor<4>([&](auto WI) { constexpr int w = decltype(WI)::value; s += part[((t * 4 + w) * NR + r) * 64 + lane]; });
            O[t][r] = s;
          });
        });
      }

      // ---------------- epilogue (as in sampler_rs.hip) -----------------------------------------------------
      const float c0 = s_coef[j * COEF_STRIDE + 0], c1 = s_coef[j * COEF_STRIDE + 1];
      const float c2 = s_coef[j * COEF_STRIDE + 2], c3 = s_coef[j * COEF_STRIDE + 3];
      const float c4 = s_coef[j * COEF_STRIDE + 4], c5 = s_coef[j * COEF_STRIDE + 5];
      const float c6 = s_coef[j * COEF_STRIDE + 6];
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        const bool in = valid && ai < a.A;
        const float eps = O[r / NR][r % NR], xt = xs[r];
        if (a.xt_hist != nullptr && in && c6 == 0.0f && wave == 0) {
          const int jh = (a.mode == MODE_CDDPM_DPM2) ? j >> 1 : j;
          const long long o = ((long long)jh * a.B + chain) * a.A + ai;
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
          float x0h = c0 * (xt - c1 * eps);
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * (c3 * x0h + c4 * xt);
          xn += c5 * z;
        } else if (a.mode == MODE_DDPM_DDIM) {
          xn = c0 * (xt - c1 * eps) + c2 * eps;
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        } else if (a.mode == MODE_CDDPM_DDPM) {
          float x0h = (xt - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * xt;
          xn += c5 * z;
        } else if (a.mode == MODE_CDDPM_DDIM) {
          float x0h = (xt - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;
        } else if (a.mode == MODE_CDDPM_DPMPP2M) {
          float x0h = (xt - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          const float d = c4 * x0h - c5 * aux[r];
          xn = c2 * xt - c3 * d;
          aux[r] = x0h;
        } else if (a.mode == MODE_EVAL) {
          xn = eps;
        } else if (a.mode == MODE_CDDPM_DPM2) {
          const float xo = (c6 == 0.0f) ? xt : aux[r];
          float x0h = (xo - c1 * eps) / c0;
          if (a.clip) x0h = fminf(fmaxf(x0h, a.x_min), a.x_max);
          xn = c2 * x0h + c3 * eps;
          aux[r] = xo;
        } else {
          xn = xt + eps / c0;
          if (a.clip) xn = fminf(fmaxf(xn, a.x_min), a.x_max);
        }
        xs[r] = (a.mode == MODE_EVAL) ? ((valid && ai < a.AO) ? xn : 0.0f) : (in ? xn : 0.0f);
      });
    }

    if (wave == 0) {
      sfor<NRX>([&](auto RI) {
        constexpr int r = decltype(RI)::value;
        const int ai = G::rowr(r) + 4 * ls;
        if (valid && ai < a.AO) a.x0[chain * a.AO + ai] = xs[r];
      });
    }
  }
}

template <int TR, int H, int ACT, int NRX, int CKMAX, bool STASH = false>
static int launch_lat_inst(KArgs& a, int ncu, hipStream_t stream) {
  auto kern = frl_sampler_lat_kernel<TR, H, ACT, NRX, CKMAX, STASH>;
  a.AR = 4 * ((a.A + 4 * Geo<TR>::KS - 1) / (4 * Geo<TR>::KS));   // x k-steps: four of them contract 4 KS rows (ROWR(n) + 4 s)
  a.CK = (a.C + Geo<TR>::KS - 1) / Geo<TR>::KS;
  const long long ntiles = (STASH && TR == 16) ? 2 * ((a.B + 31) / 32) : (a.B + TR - 1) / TR;
  const int grid = (int)(ntiles < 4LL * ncu ? ntiles : 4LL * ncu);
  const bool ln = a.ws_ln != nullptr;
  a.lat_tb_global = lat_lds_bytes(a.NE, a.L, H, true, ln) > 160 * 1024 ? 1 : 0;
  const size_t lds = lat_lds_bytes(a.NE, a.L, H, !a.lat_tb_global, ln);
  if (lds > 160 * 1024) { set_error("latency sampler: %d network evaluations per chain need %zu B of LDS (> 160 KiB)", a.NE, lds); return FRL_EUNSUPPORTED; }
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e)); return FRL_EHIP; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) { set_error("latency sampler launch: %s", hipGetErrorString(e)); return FRL_EHIP; }
  note_kernel("frl_sampler_lat_kernel<TR %d, H %d, act %d, NRX %d, CKMAX %d%s> [grid %d]", TR, H, ACT, NRX, CKMAX, STASH ? ", STASH" : "", grid);
  return FRL_OK;
}

// tile_rows: 0 = choose (16-chain tiles while they still give every CU at most one tile), 16 / 32 = forced
template <int ACT>
int launch_sampler_lat_act(const KArgs& a0, int tile_rows, hipStream_t stream) {
  KArgs a = a0;
  const int ncu = device_cus();
  const bool can16 = sampler_lat16_supports(a), can32 = sampler_lat_supports(a);
  bool use16 = can16 && (a.B + 15) / 16 <= ncu;
  if (tile_rows == 16) use16 = true; else if (tile_rows == 32) use16 = false;
  if (a.stash_h != nullptr) {      // the training forward of a small batch: 32-chain tiles, 256 wide, stash kept (capi.hip: lat_train_ok)
    if constexpr (ACT == FRL_ACT_IDENTITY) { set_error("latency training forward: no identity-activation unit"); return FRL_EINVAL; }
    else {
      if (a.H != 256 || !can32 || a.stash_a == nullptr || a.ws_ln != nullptr || a.NE != a.T) { set_error("latency training forward: unsupported configuration (H %d, evaluations %d / steps %d)", a.H, a.NE, a.T); return FRL_EUNSUPPORTED; }
      // 16-chain tiles while they still give every CU at most one tile (the forward's own rule): half the latency per tile
      if (tile_rows != 32 && can16 && 2 * ((a.B + 31) / 32) <= ncu) {
        if (a.A > 16)      // x_dim 17 .. 32 (DACER / QVPO humanoid): eight x k-steps, two Dense_L output tiles; cond rows hoisted (C = 0) or in the operand
          return a.C == 0 ? launch_lat_inst<16, 256, ACT, 8, 1, true>(a, ncu, stream) : launch_lat_inst<16, 256, ACT, 8, 32, true>(a, ncu, stream);
        return (a.C + 3) / 4 <= 12 ? launch_lat_inst<16, 256, ACT, 4, 12, true>(a, ncu, stream) : launch_lat_inst<16, 256, ACT, 4, 32, true>(a, ncu, stream);
      }
      const bool narrow = 4 * ((a.A + 7) / 8) <= 4 && (a.C + 1) / 2 <= 12;
      if (narrow) return launch_lat_inst<32, 256, ACT, 4, 12, true>(a, ncu, stream);
      // wide x with the condition rows hoisted per state (capi.hip launch_lat_train): no 64-register cond operand array
      if (a.C == 0) return launch_lat_inst<32, 256, ACT, 12, 1, true>(a, ncu, stream);
      return launch_lat_inst<32, 256, ACT, 12, 64, true>(a, ncu, stream);
    }
  }
  if (use16 ? !can16 : !can32) { set_error("latency sampler: no %d-chain-tile build for this configuration (H %d, x_dim %d, operand cond rows %d)", use16 ? 16 : 32, a.H, a.A, a.C); return FRL_EUNSUPPORTED; }
  if (use16) {
    const bool narrow = (a.C + 3) / 4 <= 12;
    if (a.A > 16) return launch_lat_inst<16, 256, ACT, 8, 32>(a, ncu, stream);     // x_dim 17 .. 32: two Dense_L output tiles
    return narrow ? launch_lat_inst<16, 256, ACT, 4, 12>(a, ncu, stream) : launch_lat_inst<16, 256, ACT, 4, 32>(a, ncu, stream);
  }
  const bool narrow = 4 * ((a.A + 7) / 8) <= 4 && (a.C + 1) / 2 <= 12;
  if (a.H == 256) return narrow ? launch_lat_inst<32, 256, ACT, 4, 12>(a, ncu, stream) : launch_lat_inst<32, 256, ACT, 12, 64>(a, ncu, stream);
  return narrow ? launch_lat_inst<32, 512, ACT, 4, 12>(a, ncu, stream) : launch_lat_inst<32, 512, ACT, 12, 64>(a, ncu, stream);
}

#ifndef FRL_RS_ACT
#error "compile with -DFRL_RS_ACT=<FRL_ACT_*>: one translation unit per activation"
#endif
template int launch_sampler_lat_act<FRL_RS_ACT>(const KArgs&, int, hipStream_t);

}  // namespace frl
