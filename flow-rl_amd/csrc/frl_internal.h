// Internal definitions shared by the HIP translation units of libflowrl_hip.so.
// Not part of the public ABI (that is include/flowrl_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "flowrl_hip.h"

namespace frl {

// ---- geometry "G32": exact-f32 MFMA 32x32x2, one wave owns 32 chains, hidden width 256 ------------
constexpr int G32_H = 256;            // hidden width this geometry is built for
constexpr int G32_CHAINS = 32;        // chains (MFMA columns) per wave
constexpr int SLAB_BYTES = 32768;     // one weight K-slab in LDS: 32 rows x 256 f32
constexpr int SLAB_ROWS = 32;
constexpr int NBUF = 2;               // LDS slab ring depth
constexpr int COEF_STRIDE = 8;        // floats per step in the coefficient table
constexpr int MAX_WAVES = 4;          // waves per workgroup (one per SIMD)
constexpr int COND_EMBED_DIM = 128;   // width of both cond_embedding layers in every reference agent
constexpr int LAST_PAD = 32;          // last layer is zero-padded to 32 output rows (one MFMA block)
constexpr int BF_CHUNK_BYTES = 32768; // bf16 variant: one LDS ring slot = 32 A-operand pieces of 1 KB (4 k-steps x 8 output blocks)
constexpr int BF_NBUF = 4;            // ring depth (3 chunks in flight)

// per-step update modes (ws coef table is built for exactly one of them)
enum Mode : int {
  MODE_DDPM_DDPM = 0,   // ddpm.py:198-206
  MODE_DDPM_DDIM = 1,   // ddpm.py:207-213
  MODE_CDDPM_DDPM = 2,  // continuous_ddpm.py:165-174
  MODE_CDDPM_DDIM = 3,  // continuous_ddpm.py:165-166,176-177
  MODE_CNF_EULER = 4,   // cnf.py:108-110
  MODE_CDDPM_DPMPP2M = 5,  // continuous_ddpm.py:198-212 (carries the previous x0_hat in registers)
  MODE_CDDPM_DPM2 = 6,     // continuous_ddpm.py:179-196 (two network evaluations per step: NE = 2 T)
  MODE_EVAL = 7,           // one plain network evaluation, output = the net's AO outputs (critic scoring, critic.py:178-194)
};

// Derived sizes of one call (host + device agree through this one function).
struct Plan {
  int A, C, D, Dp;      // action dim, cond dim, time_dim, D' (= D or 1) rows of Dense_0 taken by time
  int AO;               // network outputs (= A for the samplers; 1 for a critic)
  int x_row0, c_row0;   // first Dense_0 kernel row of the x / cond inputs (samplers: x | time | cond; critic: obs | action)
  int H, L, T;
  int NE;               // network evaluations per chain = rows of the coefficient / bias tables (T; 2 T for dpm2)
  int TR, KS, NR;       // geometry: MFMA tile rows (32: 32x32x2, 16: 16x16x4), k per MFMA, accumulators per block
  int XR;               // layer-0 x rows in the packed image: 4 KS ceil(A / 4KS) (row a = ROWR(r) + 4 s)
  int AR;               // x registers per lane = XR / KS
  int CK;               // layer-0 cond k-steps = ceil(C / KS); 0 when the cond rows are hoisted
  int hoist;            // 1: the cond rows of Dense_0 are contracted once per STATE (frl_state_bias_kernel) and enter the
                        //    sampler as a per-state accumulator init: many chains per state, or a cond wider than the budget
  int K0p;              // rows of the packed layer-0 image = XR + KS*CK
  int merged;           // (layout of the packed layer-0 image; was the LDS-slab kernel's) 1: x rows and cond rows share ONE slab (K0p <= 32)
  int n_l0;             // (layout of the packed layer-0 image; was the LDS-slab kernel's) layer-0 slabs
  int NS;               // (layout of the packed layer-0 image; was the LDS-slab kernel's) slabs per step = n_l0 + 8*(L-1) + 1
  int mode;
  long long B;
  // workspace layout (float offsets)
  size_t off_coef, off_tb, off_w0p, off_wlp, off_wlq, off_blp, off_cfeat, off_zs, off_keys, off_ln, total_floats;
  size_t w0p_rows;
  // bf16 MFMA variant (32x32x16, hidden width 256): weight stream image = bf_ns chunks of 32 KB per step
  int bf16;             // 1: desc.dtype == FRL_DTYPE_BF16
  int bf_nxk;           // layer-0 x k-steps (16 action rows each): 1 (A <= 16 and C <= 32) or 2
  int bf_cks;           // layer-0 cond k-steps the instantiation runs (2 or 8; k-steps beyond ceil(C / 16) are zero)
  int bf_l0c;           // layer-0 chunks = ceil((H / 32) (nxk + cks) / 32)
  int bf_ns;            // chunks per step = l0c + (H / 32)(H / 16) / 32 (L - 1) + 1   (4 per hidden layer at 256, 16 at 512)
  size_t off_bf16;      // float offset of the image (bf_ns * 8192 floats)
  int w2_pair;          // 1: the workspace holds BOTH geometries' tables and packed images (32-chain region, then the
                        //    16-chain region), so that it does not depend on the batch size; this Plan is one of the two
};

// Plan flags.  PLAN_HOIST off: the condition rows of Dense_0 stay in the kernel's operand (the chain VJP contracts them
// per chain).  PLAN_W2 (the forward sampler's entry points only: prepare, run and the sizes they are called with must
// agree): large plain 256-wide batches may take the 16-chain geometry at two waves per SIMD (plan_two_wave below);
// every other consumer of a Plan (VJP, single evaluation, critic) has 32-chain kernels only.
enum { PLAN_HOIST = 1, PLAN_W2 = 2 };
constexpr long long W2_MIN_CHAINS = 65536;   // 16 chains x 8 waves x 256 CUs x 2 tiles per wave
enum { W2_PICK_BY_BATCH = -1, W2_PICK_32 = 0, W2_PICK_16 = 1 };

inline int plan_layout(const FrlDesc& d, Plan& p, int flags = PLAN_HOIST, int force_tr = 0, size_t base = 0);

// The 16x16x4 geometry at two waves per SIMD (csrc/sampler_rs.hip, frl_sampler_rs_kernel_w2) runs the plain sampler
// only: f32 MLP backbone, 256 wide, no LayerNorm, ddpm / ddim / Euler updates (tensor-fed or keyed), condition rows in
// the operand (x_dim <= 16, cond_dim <= 48) — plan_two_wave_shape, everything but the batch size — and a batch large
// enough that every wave still gets two tiles.  A descriptor of that shape gets a workspace with BOTH geometries'
// regions (Plan.w2_pair), so that a workspace prepared once serves frl_sampler_run at any batch size.
inline bool plan_two_wave_shape(const FrlDesc& d, const Plan& p) {
  if (d.dtype != FRL_DTYPE_F32 || d.backbone != FRL_BACKBONE_MLP || p.H != 256 || d.layer_norm != 0) return false;     // in-kernel draws: the keyed variant of the build (EXT 6)
  if (d.kind != FRL_KIND_CNF && d.solver != FRL_SOLVER_DDPM && d.solver != FRL_SOLVER_DDIM) return false;
  if (d.rng_impl != FRL_RNG_NONE && 16 * p.A > 64) return false;   // keyed variant: a tile's 16 x A draws, one per lane
  if (d.tile_policy != FRL_TILE_AUTO && d.tile_policy != FRL_TILE_THROUGHPUT) return false;
  if (d.cond_embed) return false;
  return !(p.A > 16 || p.C > 48 || (d.particles >= 4 && p.C >= 32));
}
inline bool plan_two_wave(const FrlDesc& d, const Plan& p) {
  return plan_two_wave_shape(d, p) && (long long)d.n_states * d.particles >= W2_MIN_CHAINS;
}

// pick (PLAN_W2 only): which of the two regions of a paired workspace the returned Plan describes
inline int plan_from_desc(const FrlDesc& d, Plan& p, int flags = PLAN_HOIST, int pick = W2_PICK_BY_BATCH) {
  p.A = d.x_dim; p.C = d.cond_dim; p.T = d.steps;
  p.L = (d.backbone != FRL_BACKBONE_MLP) ? 1 : d.n_hidden;   // residual classes: Dense_0 -> [blocks] -> Dense_1 around ONE 256-wide trunk
  p.D = (d.time_embed == FRL_TEMB_NONE) ? 0 : d.time_dim;
  p.Dp = (d.time_embed == FRL_TEMB_NONE) ? 1 : d.time_dim;
  p.AO = p.A; p.x_row0 = 0; p.c_row0 = p.A + p.Dp;
  p.H = d.hidden[0];
  plan_layout(d, p, flags);
  if ((flags & PLAN_W2) != 0 && plan_two_wave_shape(d, p)) {
    Plan q = p;
    plan_layout(d, q, flags, 16, p.total_floats);     // the 16-chain region sits behind the 32-chain one
    const bool take16 = pick == W2_PICK_16 || (pick == W2_PICK_BY_BATCH && plan_two_wave(d, p));
    const size_t total = q.total_floats;
    if (take16) p = q;
    p.total_floats = total;
    p.w2_pair = 1;
  }
  return 0;
}

// Second half of the plan: geometry and workspace layout from the (possibly overridden) dims above.
inline int plan_layout(const FrlDesc& d, Plan& p, int flags, int force_tr, size_t base) {
  const bool allow_hoist = (flags & PLAN_HOIST) != 0;
  p.TR = (p.H <= 256) ? 32 : 16;
  if (force_tr != 0) p.TR = force_tr;
  p.w2_pair = 0;
  p.KS = (p.TR == 32) ? 2 : 4;
  p.NR = (p.TR == 32) ? 16 : 4;
  p.XR = 4 * p.KS * ((p.A + 4 * p.KS - 1) / (4 * p.KS));
  p.AR = p.XR / p.KS;
  p.CK = (p.C + p.KS - 1) / p.KS;
  p.hoist = (allow_hoist && d.dtype == FRL_DTYPE_F32 && p.C > 0 && ((d.particles >= 4 && p.C >= 32) || p.CK > (p.TR == 32 ? 64 : 12))) ? 1 : 0;
  if (d.backbone != FRL_BACKBONE_MLP && p.C > 0) p.hoist = 1;   // residual kernel: condition rows always enter per state
  if (p.hoist) p.CK = 0;
  p.K0p = p.XR + p.KS * p.CK;
  p.merged = (p.K0p <= SLAB_ROWS) ? 1 : 0;
  p.n_l0 = p.merged ? 1 : 1 + (p.KS * p.CK + SLAB_ROWS - 1) / SLAB_ROWS;
  p.NS = p.n_l0 + 8 * (p.L - 1) + 1;
  p.B = (long long)d.n_states * d.particles;
  if (d.kind == FRL_KIND_DDPM) p.mode = (d.solver == FRL_SOLVER_DDIM) ? MODE_DDPM_DDIM : MODE_DDPM_DDPM;
  else if (d.kind == FRL_KIND_CDDPM)
    p.mode = (d.solver == FRL_SOLVER_DDIM) ? MODE_CDDPM_DDIM : (d.solver == FRL_SOLVER_DPM2) ? MODE_CDDPM_DPM2
           : (d.solver == FRL_SOLVER_DPMPP2M) ? MODE_CDDPM_DPMPP2M : MODE_CDDPM_DDPM;
  else p.mode = MODE_CNF_EULER;
  p.NE = (p.mode == MODE_CDDPM_DPM2) ? 2 * p.T : p.T;
  size_t o = base;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };  // 256-B aligned pieces
  p.off_coef = take((size_t)p.NE * COEF_STRIDE);
  p.off_tb = take((size_t)p.NE * p.H);
  p.w0p_rows = (size_t)(p.merged ? SLAB_ROWS : p.XR + (p.n_l0 - 1) * SLAB_ROWS);
  if (p.w0p_rows < (size_t)p.K0p) p.w0p_rows = (size_t)p.K0p;
  p.off_w0p = take(p.w0p_rows * p.H);
  p.off_wlp = take((size_t)p.H * LAST_PAD);   // [H][32] image (LDS-slab kernel)
  p.off_wlq = take((size_t)p.H * 64);         // per-lane packed image: H/TR * NR/4 loads x 64 lanes x 4 floats
  p.off_blp = take(LAST_PAD);
  p.off_keys = take(d.rng_impl != FRL_RNG_NONE ? (size_t)p.NE * 2 : 0);        // per-iteration threefry keys (uint32 pairs)
  p.off_ln = take((d.layer_norm && d.backbone == FRL_BACKBONE_MLP) ? (size_t)p.L * 2 * p.H : 0);                    // LayerNorm scale | bias rows, [L][2][H]
  p.bf16 = (d.dtype == FRL_DTYPE_BF16) ? 1 : 0;
  p.bf_nxk = (p.A <= 16 && p.C <= 32) ? 1 : 2;   // narrow instantiation <8, 2> or wide <16, 8>
  p.bf_cks = (p.bf_nxk == 1) ? 2 : 8;
  if (p.H == 512) { p.bf_nxk = 1; p.bf_cks = 4; } // the 512-wide build: one instantiation <8, 4> (x_dim <= 16, cond_dim <= 64)
  const int bf_nob = p.H / 32;                    // output blocks per layer; a hidden layer is nob * (H / 16) pieces of 1 KB
  p.bf_l0c = (bf_nob * (p.bf_nxk + p.bf_cks) + 31) / 32;
  p.bf_ns = p.bf_l0c + (bf_nob * (p.H / 16) / 32) * (p.L - 1) + 1;
  p.off_bf16 = take(p.bf16 ? (size_t)p.bf_ns * (BF_CHUNK_BYTES / 4) : 0);
  // Everything above is written by frl_sampler_prepare and sits at offsets that do not depend on n_states; the per-state
  // rows below are written by frl_sampler_run itself.  So a workspace prepared once serves runs at any batch size it is
  // large enough for (frl_workspace_bytes of the run's descriptor).
  p.off_cfeat = take(d.cond_embed ? (size_t)d.n_states * COND_EMBED_DIM : 0);   // cond_embedding output, one row per state
  p.off_zs = take(p.hoist ? (size_t)d.n_states * p.H : 0);                     // hoisted cond slice of Dense_0, one row per state
  p.total_floats = o;
  return 0;
}

// Kernel arguments of the fused sampler (passed by value).
struct KArgs {
  const float* xT; const float* cond; const float* noise;
  float* x0; float* xt_hist; float* eps_hist;
  const float* wh[FRL_MAX_HIDDEN - 1];   // Dense_1 .. Dense_{L-1} kernels (H x H), read in place
  const float* bh[FRL_MAX_HIDDEN - 1];
  const float* ws_coef; const float* ws_tb; const float* ws_w0p; const float* ws_wlp; const float* ws_wlq; const float* ws_blp;
  long long B;
  int particles, A, C, T, L, H, TR, mode, clip;
  int NE;                                 // network evaluations per chain (table rows); = T except dpm2
  int AO;                                 // network outputs per chain (= A except MODE_EVAL)
  const float* wl_raw;                    // Dense_L kernel (H, AO) in place (latency kernel)
  const uint32_t* rng_keys;               // [NE][2] iteration keys (rng_impl != FRL_RNG_NONE), else NULL
  int rng_impl;
  long long rng_off, rng_total;           // first chain of this call in the unsharded batch / its chain count (draw indices)
  const float* ws_ln;                     // [L][2][H] LayerNorm scale | bias of every hidden layer, or NULL (no LayerNorm)
  int ln_post_act;                        // 0: Dense -> LayerNorm -> act; 1 + FRL_ACT_*: Dense -> act -> LayerNorm (identity-activation build)
  int lat_tb_global;                      // latency kernel: time-bias rows read from the workspace, not staged in LDS
  const float* state_bias;                // MODE_EVAL: [n_states][H] Dense_0 accumulator init per STATE (hoisted obs rows) or NULL
  int w0p_bytes;
  float x_min, x_max;
  int XR, AR, CK, merged, n_l0, NS, ntiles;
  const void* ws_bf16;                    // bf16 weight stream image (bf16 variant only)
  int bf_nxk, bf_cks, bf_l0c, bf_ns;
  float* stash_h; float* stash_a;         // latency build as the TRAINING forward (small batches): h_l = act(z_l) and act'(z_l) of
                                          // every layer and step in the chain VJP's stash layout (BArgs.hd / ad), else NULL
};

// Critic scoring + best-of-N selection FUSED into the sampler launch (csrc/sampler_rs.hip, EXT 3 builds): after the last
// step a tile's x0 sits in the chain-state registers in exactly the B-operand layout Dense_0 wants, so the E critic members
// (Ensemblize(ScalarCritic(MLP)), critic.py:154-194) are evaluated on the tile in place, the minimum over the ensemble is
// kept in a register, and the argmax / Gumbel-max over a state's particles runs across lanes (and, for >= 32 particles,
// across the consecutive tiles of the state, which one wave owns).  Only (n_states, A) actions leave the chip.
constexpr int FUSE_MAX_E = 4, FUSE_MAX_L = 4;
// Per-member pointer table in device memory (written by frl_critic_tab_kernel, csrc/critic_select.hip): the kernel indexes
// it with the running member / layer number, which a kernel-argument array would pay for with a scratch copy.
//   [CT_W0P] packed action rows of Dense_0 (Plan.off_w0p of the critic plan)   [CT_WLQ] packed head   [CT_TB] [H] bias_0
//   [CT_BLP] [32] padded head bias   [CT_ZS] [n_states][H] observation rows of Dense_0, once per state, or NULL (obs_dim = 0)
//   [CT_WH + l - 1] / [CT_BH + l - 1] hidden kernel / bias of layer l
enum { CT_W0P = 0, CT_WLQ = 1, CT_TB = 2, CT_BLP = 3, CT_ZS = 4, CT_WH = 5, CT_BH = CT_WH + FUSE_MAX_L - 1, CT_STRIDE = 12 };
static_assert(CT_BH + FUSE_MAX_L - 1 <= CT_STRIDE, "critic pointer table");
struct CriticPtrs { const float* p[FUSE_MAX_E * CT_STRIDE]; };
struct CriticTab {
  int E, L;                 // ensemble members, hidden layers of the critic backbone (256 wide)
  int relu;                 // 1: the critic's activation is relu; 0: the sampler's own (compiled) activation
  int seg, group;           // particles per state inside a tile (power of two <= 32) / consecutive tiles per state
  float inv_temperature;    // with gumbel: argmax(q * inv_temperature + gumbel)
  int w0p_bytes;
  const float* const* ptrs; // device copy of CriticPtrs: member e at [e CT_STRIDE + CT_*]
  const float* gumbel;      // (n_states, particles) or NULL
  float* q;                 // (n_states, particles) min over the ensemble, or NULL
  int32_t* best_idx;        // (n_states,) or NULL
  float* best_action;       // (n_states, A)
};

// Residual-backbone part of the sampler's arguments (csrc/sampler_res.hip): block i = Dense (H, 4H) -> act -> Dense (4H, H)
// around a residual, optionally behind a LayerNorm; kernels read in place.
struct ResArgs {
  int n_blocks, ln, final_ln, final_act;      // final_act: ResidualMLP's act before Dense_1; final_ln: Simba's LayerNorm there
  int split;                                  // 1: one wave per 32-chain tile (throughput); 4: the latency build, four waves per tile
  const float* w1[FRL_MAX_HIDDEN]; const float* b1[FRL_MAX_HIDDEN];
  const float* w2[FRL_MAX_HIDDEN]; const float* b2[FRL_MAX_HIDDEN];
  const float* ln_scale[FRL_MAX_HIDDEN]; const float* ln_bias[FRL_MAX_HIDDEN];
  const float* fln_scale; const float* fln_bias;
};
constexpr int RES_RED_FLOATS = 4 * 64 * 64;   // latency build: four waves' partial sums of HALF of x (64 registers x 64 lanes each)
inline size_t res_lds_bytes(int NE, int n_blocks, int split = 1) {
  return sizeof(float) * ((size_t)NE * 256 + LAST_PAD + (size_t)NE * COEF_STRIDE + (size_t)n_blocks * (1024 + 3 * 256) + 2 * 256 +
                          (split > 1 ? (size_t)RES_RED_FLOATS : 0));
}

// Arguments of the prepare kernel.
struct PArgs {
  FrlParams p;
  FrlSchedule s;
  float* ws;
  Plan plan;
  int kind, time_embed;
};

// double scratch (counted in floats) of the time-branch VJP kernel (csrc/time_bwd.hip): (1 + 7 D) T doubles
inline size_t time_bwd_scratch_floats(int T, int D) { return 2 * ((size_t)T * (1 + 7 * (size_t)(D > 0 ? D : 0)) + 8); }

// ---- backward (chain VJP) -------------------------------------------------------------------------------
// Workspace of frl_sampler_bwd = the forward tables (Plan) followed by transposed / packed weight images and the
// per-(wave-tile, step, layer) stash [64 register pairs][64 lanes][2] f32 = 32 KB each:
//   Hd = act(z_l)   (input of Dense_{l+1}: the X operand of dW_{l+1})
//   Ad = act'(z_l)  (re-read by the backward sweep)
//   Gd = dL/dz_l    (the G operand of dW_l)
// Deterministic weight-gradient reduction (no float atomics): workgroup w of the dW kernels owns slot w of `part`, a
// [n_part][dw_part_stride] image of its partial sums — Dense_0 tile [k0_rows staged rows x | cond][256] (k0_rows = 32 or 160), hidden tiles [256][256],
// Dense_L tile [256][32], bias column sums [L][256] (layers 1..L), per-step dL/dtb [T][256] — and a second kernel
// (frl_bwd_reduce_kernel) adds the slots in slot order, writing every gradient leaf exactly once.
struct DwPart {
  size_t k0, kh, kl, bias, tb, stride;   // float offsets inside one slot; kh is hidden layer 1 (layer l at kh + (l-1) 65536)
};
__host__ __device__ inline DwPart dw_part_layout(int L, int T, int k0_rows = 32) {
  DwPart d;
  d.k0 = 0; d.kh = (size_t)k0_rows * 256; d.kl = d.kh + (size_t)(L - 1) * 65536; d.bias = d.kl + 8192; d.tb = d.bias + (size_t)L * 256;
  d.stride = d.tb + (size_t)T * 256;
  return d;
}
inline int device_cus();

struct BwdPlan {
  Plan f;
  int n_part; size_t off_part;   // dW partial slots (= workgroups of the dW kernel)
  int cond_embed;                // desc.cond_embed
  int wide;                      // 1: the <12, 64> operand build (x_dim <= 24, <= 128 condition rows): DACER / QVPO shapes
  int k0_rows;                   // staged rows of Dense_0's dW tile: 32 ceil((A + C) / 32)
  int ge_stride;                 // floats of dL/deps per tile-step: 64 x (4 | 12) registers
  size_t off_w0cq;               // per-lane packed images of Dense_0's cond rows, 4 blocks of 32 features (cond_embed only)
  size_t off_gcf;                // dL/dcfeat per CHAIN, (B, 128), summed over steps (cond_embed only)
  size_t off_ceb;                // cond_embedding backward scratch: (n_states, 3, 128) = h1 | dL/dz1 | dL/dz2
  long long n_wt, n_ts;          // wave tiles (32 chains), tile-steps
  size_t off_wt[FRL_MAX_HIDDEN - 1];   // transposed hidden kernels WT_l[out][in]
  size_t off_wlt;                // transposed, row-padded last kernel  [XR][H]
  size_t off_w0xq;               // per-lane packed image of Dense_0's x rows (A operand of dx = W0x G_0)
  size_t off_hd, off_ad, off_gd; // stash, n_ts * L * 8192 floats each
  size_t off_ge;                 // dL/deps per tile-step: [NRX=4 regs][64 lanes]
  size_t off_tbs;                // double scratch of the time-branch VJP
  size_t total_floats;
};

// Second half of the backward plan: stash / image / slot layout behind the forward tables b.f (set by the caller).
// dw_rows = rows of Dense_0's dW tile source (x rows + whatever is staged as "condition" rows for the dW kernel).
inline int bwd_plan_finish(BwdPlan& b, int cond_embed, long long n_states, int dw_rows) {
  const Plan& p = b.f;
  b.cond_embed = cond_embed ? 1 : 0;
  b.wide = (p.AR > 4 || p.CK > 12) ? 1 : 0;
  b.k0_rows = 32 * ((dw_rows + 31) / 32);
  if (b.k0_rows < 32) b.k0_rows = 32;
  b.ge_stride = 64 * (b.wide ? 12 : 4);
  b.n_wt = (p.B + 31) / 32;
  b.n_ts = b.n_wt * p.T;
  size_t o = p.total_floats;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
  for (int l = 1; l < p.L; ++l) b.off_wt[l - 1] = take((size_t)p.H * p.H);
  b.off_wlt = take((size_t)p.XR * p.H);
  b.off_w0xq = take((size_t)p.H * 64);
  b.off_hd = take((size_t)b.n_ts * p.L * 8192);
  b.off_ad = take((size_t)b.n_ts * p.L * 8192);
  b.off_gd = take((size_t)b.n_ts * p.L * 8192);
  b.off_ge = take((size_t)b.n_ts * b.ge_stride);
  b.off_tbs = take(time_bwd_scratch_floats(p.T, p.D));
  { const long long cus = device_cus(); b.n_part = (int)(b.n_ts < cus ? b.n_ts : cus); if (b.n_part < 1) b.n_part = 1; }
  b.off_part = take((size_t)b.n_part * dw_part_layout(p.L, p.T, b.k0_rows).stride);
  b.off_w0cq = take(cond_embed ? (size_t)4 * p.H * 64 : 0);
  b.off_gcf = take(cond_embed ? (size_t)p.B * COND_EMBED_DIM : 0);
  b.off_ceb = take(cond_embed ? (size_t)n_states * 3 * COND_EMBED_DIM : 0);
  b.total_floats = o;
  return 0;
}

inline int bwd_plan_from_desc(const FrlDesc& d, BwdPlan& b) {
  plan_from_desc(d, b.f, 0);
  return bwd_plan_finish(b, d.cond_embed, d.n_states, b.f.A + b.f.C);
}

struct BArgs {
  KArgs k;                        // forward arguments (xT unused; k.noise unused)
  const float* xt_hist; const float* g_x0; float* g_xT;
  const float* eps_hist;          // history[1] of the training forward (phase 2 only)
  const float* wt[FRL_MAX_HIDDEN - 1];  // transposed hidden kernels
  const float* ws_wlt; const float* ws_w0xq;
  float* hd; float* ad; float* gd; float* ge;
  float* x0d;                     // bf16 variant: packed Dense_0 operands of every tile-step (BwdPlan16.off_x0d)
  float* grad_kernel[FRL_MAX_HIDDEN + 1]; float* grad_bias[FRL_MAX_HIDDEN + 1]; float* grad_tb;
  int Dp;                         // time rows of Dense_0 (skipped by the dW_0 scatter)
  long long n_ts;
  float* part; int n_part;        // dW partial slots (DwPart layout), one per workgroup of the dW kernel
  int k0_rows, ge_stride;         // BwdPlan.k0_rows / ge_stride
  const float* dw_cond; int dw_C; // what the dW kernel stages behind the x rows of Dense_0's tile (the chain VJP: k.cond / k.C; the
                                  // single-evaluation VJP: u = [tfeat | cond] per sample while the sweep's own k.C is 0)
  const float* ws_w0cq;           // packed Dense_0 cond rows (4 blocks), or NULL
  float* gcf;                     // (B, 128) dL/dcfeat per chain, or NULL: written by the backward sweep of the wide build
};
// cond_embedding backward (csrc/cond_embed.hip): per-state chain rule, then a fixed-order sum over the states
int launch_cond_embed_bwd(int act, const float* obs, long long n_states, int particles, int in_dim, const float* w0, const float* b0,
                          const float* w1, const float* b1, const float* gcf, int gcf_stride, float* scratch, float* g_w0, float* g_b0,
                          float* g_w1, float* g_b1, hipStream_t stream);
// single-evaluation VJP pieces (csrc/sampler_bwd.hip, csrc/net_eval.hip)
int launch_eval_du(const BArgs& b, const float* w0_rows, int n_rows, float* du, hipStream_t stream);   // du[b][k] = W0[A + k] . dL/dz_0[b]
struct TimeBwdSampleArgs {
  const float* t; const float* du; int du_stride;        // (B,) time inputs; dL/dtfeat = du[b][0 .. D)
  const float* time_kernel; const float* tk0; const float* tb0; const float* tk1; const float* tb1;
  float* scratch;                                        // (B, 4 D + D / 2): temb | h1 | dz1 | dz2 | 2 pi t gf
  float* g_tk0; float* g_tb0; float* g_tk1; float* g_tb1; float* g_time_kernel;
  long long B; int D, time_embed;
};
int launch_time_bwd_samples(const TimeBwdSampleArgs& a, hipStream_t stream);
// The per-sample contributions are added in two fixed-order stages: chunks of TBS chunk-size samples (at most 256 chunks),
// then the chunks in index order.  Scratch = the per-sample rows followed by [chunks][leaf elements] doubles.
inline long long time_bwd_samples_chunk(long long B) { const long long c = (B + 255) / 256; return c < 64 ? 64 : c; }
inline size_t time_bwd_samples_scratch_floats(long long B, int D, bool fourier) {
  const long long S = 4 * D + D / 2, total = 2LL * D * D + 2 * D + (fourier ? D / 2 : 0);
  const long long ch = time_bwd_samples_chunk(B), nch = (B + ch - 1) / ch;
  return (size_t)(((B * S + 1) & ~1LL) + 2 * nch * total);
}
int launch_bwd_reduce(const BArgs& b, hipStream_t stream);   // fixed-order second stage of the weight-gradient reduction

// ---- layer-by-layer single-evaluation VJP (csrc/net_vjp_layered.hip): 512-wide, LayerNorm and residual predictors ----
struct LayeredPlan {
  long long B;
  int A, Cp, K0, H, L, Wmax, res, splits;
  size_t off_inp, off_dinp;
  size_t off_z[FRL_MAX_HIDDEN + 1], off_h[FRL_MAX_HIDDEN], off_e[FRL_MAX_HIDDEN], off_a[FRL_MAX_HIDDEN], off_hf;
  size_t off_g0, off_g1, off_g2, off_pdy, off_pxh, off_cs, off_part;
  size_t total_floats;
};
int layered_plan(const FrlDesc& d, int Cp, LayeredPlan& q);
int launch_layered_fwd(const FrlDesc& d, const FrlParams& p, const LayeredPlan& q, float* ws, const float* x, const float* u, float* out,
                       hipStream_t stream);
int launch_layered_bwd(const FrlDesc& d, const FrlParams& p, const LayeredPlan& q, float* ws, const float* g_out, const FrlEvalBwdBufs& gb,
                       float* g_x, hipStream_t stream);

// ---- backward, bf16-MFMA variant (csrc/sampler_bwd_bf16.hip) -------------------------------------------------
// Workspace = the forward plan with its bf16 weight stream image, DIRECTLY followed by the backward stream image
// (same chunk format: W_L^T, transposed hidden kernels top-down, Dense_0's x rows), so that the sweep cycles through
// 2 * bf_ns chunks per step; then the stash, packed bf16 in the B-operand layout of the sweep
// ([16 k-step pieces][64 lanes][8 bf16] = 16 KB per (wave-tile, step, layer)):
//   Zd = bf16(z_l) (one copy: the sweep recomputes act'(z), the dW kernel act(z) from it), Gd = bf16(dL/dz_l),
//   Ge = bf16(dL/deps) (one 1 KB piece), and X0d = the network INPUT of every step as the sweep's own packed Dense_0
//   operands (x rows | cond k-steps: three 1 KB pieces in a 4 KB block): the X operand of dW_0.
// Wave tiles are counted per 256-chain workgroup tile (8 each), so that every wave of the sweep owns a stash slot.
struct BwdPlan16 {
  Plan f;
  long long n_wt, n_ts;
  size_t off_img_b, off_hd, off_ad, off_gd, off_ge, off_x0d, off_tbs, total_floats;
  int n_part; size_t off_part;
};

inline int bwd16_plan_from_desc(const FrlDesc& d, BwdPlan16& b) {
  plan_from_desc(d, b.f);
  const Plan& p = b.f;
  b.n_wt = 8 * ((p.B + 255) / 256);
  b.n_ts = b.n_wt * p.T;
  size_t o = p.total_floats;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
  b.off_img_b = take((size_t)p.bf_ns * (BF_CHUNK_BYTES / 4));
  b.off_hd = take((size_t)b.n_ts * p.L * 4096);   // Zd = bf16(z_l): act(z) / act'(z) are recomputed by the consumers
  b.off_ad = b.off_hd;                            // (no separate act' stash)
  b.off_gd = take((size_t)b.n_ts * p.L * 4096);
  b.off_ge = take((size_t)b.n_ts * 256);
  b.off_x0d = take((size_t)b.n_ts * 1024);
  b.off_tbs = take(time_bwd_scratch_floats(p.T, p.D));
  { const long long U = ((p.B + 31) / 32) * p.T, cus = device_cus(); b.n_part = (int)(U < cus ? U : cus); if (b.n_part < 1) b.n_part = 1; }
  b.off_part = take((size_t)b.n_part * dw_part_layout(p.L, p.T).stride);
  b.total_floats = o;
  return 0;
}

int launch_prepare_bwd_bf16(const FrlParams& p, const BwdPlan16& bp, float* ws, hipStream_t stream);
template <int PHASE> int launch_bwd16_sweep_phase(const FrlDesc& d, const BArgs& b, hipStream_t stream);
template <> int launch_bwd16_sweep_phase<0>(const FrlDesc& d, const BArgs& b, hipStream_t stream);
template <> int launch_bwd16_sweep_phase<1>(const FrlDesc& d, const BArgs& b, hipStream_t stream);
template <> int launch_bwd16_sweep_phase<2>(const FrlDesc& d, const BArgs& b, hipStream_t stream);
int launch_sampler_bwd_bf16(const FrlDesc& d, const BArgs& b, int phase, hipStream_t stream);   // 0 recompute + backward, 1 training forward, 2 backward from the stash

// time-branch VJP (csrc/time_bwd.hip): one workgroup, double-precision scratch of (1 + 7 D) T doubles
struct TBArgs {
  FrlParams p;
  FrlSchedule s;
  const float* grad_tb;
  double* scratch;
  float* g_tmlp_kernel[2]; float* g_tmlp_bias[2]; float* g_time_kernel;
  float* g_bias0; float* g_kernel0;      // Dense_0 bias; Dense_0 kernel (rows [A, A + D') are written)
  int kind, time_embed, T, D, H, A;
};
int launch_time_bwd(const TBArgs& a, hipStream_t stream);
int launch_rng_chain(const uint32_t* key_in, int impl, int mode, int T, uint32_t* keys, uint32_t* rng_out, hipStream_t stream);
int launch_random_split(const uint32_t* key, int impl, int num, uint32_t* out, hipStream_t stream);
int launch_random_fill(const uint32_t* key, int impl, long long n, void* out, int kind, hipStream_t stream);   // 0 bits, 1 normal, 2 gumbel

int launch_prepare_bwd(const FrlParams& p, const BwdPlan& bp, float* ws, hipStream_t stream);
int launch_sampler_bwd(const FrlDesc& d, const BArgs& a, int phase, hipStream_t stream);   // phase 0 recompute + backward, 1 training forward, 2 backward from the stash
// latency build of the backward sweep from the stash (csrc/sampler_bwd_lat.hip): four waves per 32-chain tile, small batches
bool bwd_sweep_lat_supports(const BArgs& b);
int launch_bwd_sweep_lat(const BArgs& b, hipStream_t stream);
template <int PHASE> int launch_bwd_sweep_phase(const FrlDesc& d, const BArgs& b, hipStream_t stream);   // one translation unit each
template <> int launch_bwd_sweep_phase<0>(const FrlDesc& d, const BArgs& b, hipStream_t stream);
template <> int launch_bwd_sweep_phase<1>(const FrlDesc& d, const BArgs& b, hipStream_t stream);
template <> int launch_bwd_sweep_phase<2>(const FrlDesc& d, const BArgs& b, hipStream_t stream);

void set_error(const char* fmt, ...);
// Names of the compute kernels launched by the current C-ABI call on this thread (frl_last_kernels): every launcher appends
// the instantiation it actually enqueued, so a report of "which kernel ran" can never drift from the dispatch rules.
void note_kernel(const char* fmt, ...);

// launchers implemented in the .hip files
int launch_prepare(const PArgs& a, hipStream_t stream);
template <int ACT> int launch_sampler_rs_act(const KArgs& a, hipStream_t stream);  // register-streamed variant, one TU per activation
template <int ACT> int launch_sampler_rs_fused_act(const KArgs& a, const CriticTab& ct, hipStream_t stream);  // + critic scoring / selection (EXT 3)
template <int ACT> int launch_sampler_res_act(const KArgs& a, const ResArgs& ra, hipStream_t stream);  // residual backbones (csrc/sampler_res.hip)
template <int ACT> int launch_sampler_lat_act(const KArgs& a, int tile_rows, hipStream_t stream); // latency variant (4 waves per tile), one TU per activation
// CU count of the current device (cached per thread and device)
inline int device_cus() {
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

// LDS of the latency kernel; the [NE][H] time-bias rows are staged only when they fit (else read through L2 per evaluation)
inline size_t lat_lds_bytes(int NE, int L, int H, bool tb_rows, bool ln = false) {
  return sizeof(float) * ((tb_rows ? (size_t)NE * H : 0) + (size_t)(L - 1) * H + LAST_PAD + (size_t)NE * COEF_STRIDE + 2 * (size_t)H * 32 +
                          (ln ? 256 + (size_t)L * 2 * H : 0));   // LayerNorm: per-wave partial sums + the scale | bias rows
}
// AR / CK in the units of the latency kernel (32x32x2: two input rows per k-step), whatever geometry the plan chose
inline bool sampler_lat_supports(const KArgs& a) {
  const int ar = 4 * ((a.A + 7) / 8), ck = (a.C + 1) / 2;
  return (a.H == 256 || a.H == 512) && ((ar <= 4 && ck <= 12) || (ar <= 12 && ck <= 64)) && a.XR >= 2 * ar &&
         lat_lds_bytes(a.NE, a.L, a.H, false, a.ws_ln != nullptr) <= 160 * 1024;
}
// the 16-chain-tile build (MFMA 16x16x4): x rows ROWR(n) + 4 s of four k-steps cover 16 action rows (eight: 32, with two
// Dense_L output tiles); cond k-steps of 4 rows
inline bool sampler_lat16_supports(const KArgs& a) {
  return a.H == 256 && a.A <= 32 && (a.C + 3) / 4 <= 32 && lat_lds_bytes(a.NE, a.L, a.H, false, a.ws_ln != nullptr) <= 160 * 1024;
}
int launch_prepare_bf16(const FrlParams& p, const Plan& pl, float* ws, hipStream_t stream);   // bf16 weight stream image
int launch_sampler_bf16(const FrlDesc& d, const KArgs& a, hipStream_t stream);                 // bf16 MFMA variant (H = 256 or 512)
size_t bf16_lds_bytes(int T, int L, int H = 256);
int launch_cond_embed(int act, const float* obs, long long n_states, int in_dim, const float* w0, const float* b0,
                       const float* w1, const float* b1, float* cfeat, hipStream_t stream);
int launch_state_bias(const float* obs, long long n_states, int in_dim, int H, const float* w, const float* b, float* zs,
                      hipStream_t stream);
int launch_critic_tab(const CriticPtrs& v, const float** dst, hipStream_t stream);
int launch_critic_select(const float* q_members, int E, long long n_states, int particles, int A, const float* actions,
                         const float* gumbel, float inv_temperature, float* q, int32_t* best_idx, float* best_action,
                         hipStream_t stream);
int launch_random_uniform(const uint32_t* key, int impl, long long n, float lo, float hi, float* out, hipStream_t stream);
int launch_random_randint(const uint32_t* key, int impl, long long n, int32_t minval, int32_t maxval, int32_t* out, hipStream_t stream);
// csrc/net_eval.hip: per-sample-time evaluation rows and the forward-process pieces
int launch_time_rows(const FrlDesc& d, const FrlParams& p, const float* t, const float* cond, int C, float* u, long long B, hipStream_t stream);
int launch_add_noise(int kind, long long B, int A, int steps, const float* x0, const float* e, const void* t, const float* alpha_hats, float p0, float p1,
                     float* xt, float* vel, hipStream_t stream);
int launch_onestep_update(int solver, long long B, int A, int N, int steps, const float* xt, const float* eps, const int32_t* t, const float* alphas,
                          const float* alpha_hats, int clip, float x_min, float x_max, const float* noise, float* out, hipStream_t stream);
int launch_mse(long long n, const float* pred, const float* target, float* loss, float* grad, void* ws, hipStream_t stream);
int launch_selftest_mfma32(const float* A, const float* B, float* D, int K, hipStream_t stream);

}  // namespace frl
