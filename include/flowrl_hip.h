/*
 * flowrl_hip.h — C ABI of libflowrl_hip.so: the MI355X (gfx950) fused action sampler.
 *
 * Drop-in boundary for the ONE hot path of typoverflow/flow-rl: batched reverse-diffusion /
 * flow-ODE action sampling.  The reference has no plugin / FFI interface (it is pure JAX); the
 * boundary is created at its three sampler entry points, whose semantics each entry below
 * replaces (paths relative to the reference checkout):
 *
 *   DDPM.sample                        flowrl/flow/ddpm.py:173-221
 *   ContinuousDDPM.sample              flowrl/flow/continuous_ddpm.py:117-222   (solvers ddpm, ddim, dpm2, dpm++2m)
 *   ContinuousNormalizingFlow.sample   flowrl/flow/cnf.py:113-141
 *   DDPMBackbone / FlowBackbone        flowrl/flow/ddpm.py:70-95, flowrl/flow/cnf.py:17-42
 *   MLP (Dense -> act)*L -> Dense      flowrl/module/mlp.py:10-34, flowrl/agent/offline/fql/network.py:10-35
 *   Positional / LearnableFourier      flowrl/module/time_embedding.py:6-27
 *
 * Plain C, plain pointers and sizes, no torch / JAX types.  Every pointer in FrlParams,
 * FrlSchedule and FrlFwdBufs is DEVICE memory (float32, C-contiguous) owned by the caller; the
 * library never allocates, frees or retains them.  All entry points only ENQUEUE work on
 * `stream` (no host synchronisation) and return 0 on success or a negative FRL_E* code; the
 * message is available from frl_last_error() (thread-local).  Shape validation happens before
 * any launch.  The XLA-FFI handler (csrc/xla_ffi_shim.cc, compiled only where jaxlib's headers
 * exist) is a thin translation onto frl_sampler_fwd / frl_sampler_bwd; INTEGRATION.md shows the binding.
 */
#ifndef FLOWRL_HIP_H_
#define FLOWRL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRL_ABI_VERSION 6
#define FRL_MAX_HIDDEN 8 /* hidden Dense layers of the predictor MLP (mlp.py:25) */

/* error codes */
#define FRL_OK 0
#define FRL_EINVAL (-1)       /* bad descriptor / null pointer / size mismatch          */
#define FRL_EUNSUPPORTED (-2) /* valid reference config this build has no kernel for    */
#define FRL_EWORKSPACE (-3)   /* workspace too small / misaligned                        */
#define FRL_EHIP (-4)         /* a HIP runtime call failed (message has hipGetErrorString) */

/* sampler kind == which reference class' sample() is reproduced */
#define FRL_KIND_DDPM 0  /* flowrl/flow/ddpm.py:173            time input = int step t (as f32) */
#define FRL_KIND_CDDPM 1 /* flowrl/flow/continuous_ddpm.py:117 time input = ts[i]               */
#define FRL_KIND_CNF 2   /* flowrl/flow/cnf.py:113             time input = k / steps           */

#define FRL_SOLVER_DDPM 0  /* ddpm.py:198-206 / continuous_ddpm.py:170-174 */
#define FRL_SOLVER_DDIM 1  /* ddpm.py:207-213 / continuous_ddpm.py:176-177 */
#define FRL_SOLVER_EULER 2 /* cnf.py:92-111 */
#define FRL_SOLVER_DPM2 3    /* continuous_ddpm.py:179-196  DPM-Solver-2: midpoint, TWO network evaluations per step */
#define FRL_SOLVER_DPMPP2M 4 /* continuous_ddpm.py:198-212  DPM-Solver++(2M): multistep, carries the previous x0_hat  */

#define FRL_TEMB_NONE 0       /* raw scalar t concatenated (FQL: cnf.py:30-41 with time_embedding=None) */
#define FRL_TEMB_POSITIONAL 1 /* time_embedding.py:6-16  */
#define FRL_TEMB_FOURIER 2    /* time_embedding.py:19-27 */

#define FRL_ACT_MISH 0      /* functional/activation.py:17-18 */
#define FRL_ACT_SILU 1      /* jax.nn.silu  (activation.py:13) */
#define FRL_ACT_RELU 2      /* jax.nn.relu  (activation.py:10) */
#define FRL_ACT_GELU_TANH 3 /* flax.linen.gelu approximate=True (fql/fql.py:199) */
#define FRL_ACT_TANH 4      /* jax.nn.tanh  (activation.py:14) */
#define FRL_ACT_ELU 5       /* jax.nn.elu   (activation.py:11) */

#define FRL_TILE_AUTO 0       /* latency build when its rounds (one 32-chain tile per CU) cost less than the throughput build's (four per CU) */
#define FRL_TILE_THROUGHPUT 1 /* one wave per tile, no barriers (csrc/sampler_rs.hip): large batches */
#define FRL_TILE_LATENCY 2    /* four waves per tile, features split over the SIMDs (csrc/sampler_lat.hip): acting */
#define FRL_TILE_LATENCY_32 3 /* the latency build on 32-chain tiles (MFMA 32x32x2)                                  */
#define FRL_TILE_LATENCY_16 4 /* the latency build on 16-chain tiles (MFMA 16x16x4): x_dim <= 32, H = 256; LATENCY picks it
                                 while ceil(B / 16) <= CUs                                                           */

/* Where the per-step N(0, I) draws of the ddpm solvers come from (FrlDesc.rng_impl).  The reference draws them with
 * jax.random inside the scan (ddpm.py:188,204-206; continuous_ddpm.py:161,174,190; cnf.py:126 for the key sequence). */
#define FRL_RNG_NONE 0                   /* the caller supplies bufs.noise (T, B, A)                                   */
#define FRL_RNG_THREEFRY_PARTITIONABLE 1 /* in-kernel threefry2x32, jax_threefry_partitionable=True (jax >= 0.5.0 default) */
#define FRL_RNG_THREEFRY_LEGACY 2        /* in-kernel threefry2x32, the pre-0.5.0 default key / counter layout           */

#define FRL_LN_NONE 0
#define FRL_LN_PRE_ACT 1  /* flowrl/module/mlp.py:26-29 */
#define FRL_LN_POST_ACT 2 /* flowrl/agent/offline/fql/network.py:26-30 */

#define FRL_DTYPE_F32 0  /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32): the bit-faithful path */
#define FRL_DTYPE_BF16 1 /* bf16 MFMA (v_mfma_f32_32x32x16_bf16) operands, f32 accumulate / activations / chain state: stated tolerance;
                          * forward: hidden width 256 (x_dim <= 32, cond_dim <= 128) or 512 (x_dim <= 16, cond_dim <= 64) */

/* Static description of one sampler call — the analogue of the reference's jit static_argnames
 * plus the pytree_node=False fields of DDPM / ContinuousDDPM / ContinuousNormalizingFlow
 * (ddpm.py:103-107, continuous_ddpm.py:51-59, cnf.py:50-54). */
/* predictor classes (the noise_predictor / vel_predictor of a DDPMBackbone / FlowBackbone) */
#define FRL_BACKBONE_MLP 0       /* flowrl/module/mlp.py:10-34 (and FQL's variant) */
#define FRL_BACKBONE_RESIDUAL 1  /* ResidualMLP, flowrl/module/mlp.py:37-98: Dense_0, ResidualLinear_i, act, Dense_1 */
#define FRL_BACKBONE_SIMBA 2     /* Simba, flowrl/module/simba.py:10-63: Dense_0, SimbaBlock_i, LayerNorm_0, Dense_1 */

typedef struct FrlDesc {
  int32_t abi_version; /* FRL_ABI_VERSION */
  int32_t kind;        /* FRL_KIND_*   */
  int32_t solver;      /* FRL_SOLVER_* */
  int32_t steps;       /* T */
  int64_t n_states;    /* number of distinct condition rows */
  int32_t particles;   /* chains per state; B = n_states * particles chains (dql.py:145 obs_repeat is
                          NOT materialised: the kernel broadcasts each condition row) */
  int32_t x_dim;       /* A: action dim */
  int32_t cond_dim;    /* C: condition features reaching the predictor; 0 = no condition */
  int32_t time_embed;  /* FRL_TEMB_* */
  int32_t time_dim;    /* D (output_dim of the embedding); ignored for FRL_TEMB_NONE */
  int32_t n_hidden;    /* L: number of hidden Dense layers of the predictor */
  int32_t hidden[FRL_MAX_HIDDEN]; /* their widths */
  int32_t act;         /* FRL_ACT_* of the predictor (the time MLP is always mish, ddpm.py:85-88) */
  int32_t clip;        /* clip_sampler */
  float x_min, x_max;
  int32_t dtype;        /* FRL_DTYPE_* */
  int32_t want_history; /* 1: write xt_hist / eps_hist (ddpm.py:217) */
  int32_t cond_embed;   /* 1: a cond_embedding MLP(hidden_dims=(128,128), act) (dacer.py:263-266, qvpo.py:204-207)
                           maps each raw condition row (cond_in_dim) to the cond_dim = 128 features, once per STATE */
  int32_t cond_in_dim;  /* raw observation width when cond_embed = 1 */
  int32_t tile_policy;  /* FRL_TILE_*: which f32 kernel build runs the chains (results agree to rounding) */
  int32_t rng_impl;     /* FRL_RNG_*: with a threefry layout, bufs.rng_key replaces bufs.noise and the call also produces
                           the rng the reference's sample() returns (bufs.rng_out), for every solver */
  int32_t layer_norm;   /* FRL_LN_PRE_ACT (1): the predictor MLP is Dense -> LayerNorm -> act (MLP(layer_norm=True), mlp.py:26-29);
                           FRL_LN_POST_ACT (2): Dense -> act -> LayerNorm (FQL's MLP, agent/offline/fql/network.py:26-30).
                           flax LayerNorm: eps 1e-6, scale and bias; the variance is taken about the mean — f32 builds, forward only */
  int32_t backbone;     /* FRL_BACKBONE_*: the predictor's class.  For the residual classes n_hidden / hidden[] are the BLOCKS
                           (len(hidden_dims), each 256 wide) and layer_norm says whether ResidualLinear normalises its input
                           (ResidualMLP(layer_norm=True), dac.py:201-207); Simba always does.  f32, forward only */
  int32_t res_multiplier; /* expansion of a residual block (mlp.py:39, simba.py:13); 4 in every reference config, the only value built */
} FrlDesc;

/* Leaves of the Flax parameter tree produced by Model.create (module/model.py:31-60) for a
 * DDPMBackbone / FlowBackbone; nn.Dense kernels are (in, out) row-major and are read IN PLACE. */
typedef struct FrlParams {
  const float* time_kernel;    /* params/time_embedding/kernel (D/2, 1); FRL_TEMB_FOURIER only   */
  const float* tmlp_kernel[2]; /* params/MLP_0/Dense_{0,1}/kernel (D, D); NULL for FRL_TEMB_NONE  */
  const float* tmlp_bias[2];   /* params/MLP_0/Dense_{0,1}/bias   (D,)                            */
  const float* kernel[FRL_MAX_HIDDEN + 1]; /* params/{noise,vel}_predictor/Dense_i/kernel; i = 0..L     */
  const float* bias[FRL_MAX_HIDDEN + 1];   /* .../Dense_i/bias. Dense_0 kernel is (A + D' + C, H0) with
                                              row order x | time | cond (ddpm.py:92), D' = D or 1      */
  const float* cond_kernel[2]; /* params/cond_embedding/Dense_{0,1}/kernel (cond_in_dim,128), (128,128); cond_embed only */
  const float* cond_bias[2];   /* params/cond_embedding/Dense_{0,1}/bias (128,) */
  /* params/<predictor>/LayerNorm_i/{scale,bias}, (hidden[i],) each, i = 0..L-1; read when desc.layer_norm = 1 */
  const float* ln_scale[FRL_MAX_HIDDEN];
  const float* ln_bias[FRL_MAX_HIDDEN];
  /* Residual backbones (desc.backbone != FRL_BACKBONE_MLP): kernel[0] / bias[0] = params/<predictor>/Dense_0 (the input
   * projection, rows x | time | cond), kernel[1] / bias[1] = .../Dense_1 (the output layer), and per block i =
   * .../ResidualLinear_i or .../SimbaBlock_i: Dense_0 (H, m H), Dense_1 (m H, H), LayerNorm_0 (H,) */
  const float* res_kernel[FRL_MAX_HIDDEN][2];
  const float* res_bias[FRL_MAX_HIDDEN][2];
  const float* res_ln_scale[FRL_MAX_HIDDEN];
  const float* res_ln_bias[FRL_MAX_HIDDEN];
  const float* final_ln_scale; /* Simba: params/<predictor>/LayerNorm_0/{scale,bias} (simba.py:58) */
  const float* final_ln_bias;
} FrlParams;

/* Noise-schedule tables, each (T+1,) float32 on the device.
 * DDPM:  the model's own pytree leaves betas/alphas/alpha_hats/postvars (ddpm.py:108-111), verbatim.
 * CDDPM: the per-call tables of continuous_ddpm.py:134-138 (postvars unused, ts required); the log-SNR table of
 *        dpm++2m (continuous_ddpm.py:141-142) is derived from alpha_hats by frl_sampler_prepare.
 * CNF:   all NULL. */
typedef struct FrlSchedule {
  const float* betas;
  const float* alphas;
  const float* alpha_hats;
  const float* postvars;
  const float* ts;
  /* FRL_SOLVER_DPM2 only, each (T,) with entry i-1 belonging to loop index i (continuous_ddpm.py:145-147):
   * the interval midpoints (ts[:-1] + ts[1:]) / 2 and noise_schedule_func evaluated there */
  const float* mid_ts;
  const float* mid_alphas;
  const float* mid_sigmas;
} FrlSchedule;

typedef struct FrlFwdBufs {
  const float* xT;    /* (B, A)  start of the chain (x0 for CNF)                          */
  const float* cond;  /* (n_states, C) un-repeated condition rows, NULL iff cond_dim == 0;
                         (n_states, cond_in_dim) raw observations when desc.cond_embed = 1      */
  const float* noise; /* (T, B, A): noise[j] = the N(0,I) draw of loop iteration j (ddpm.py:206,
                         continuous_ddpm.py:174), drawn by the caller with the reference's key
                         sequence.  NULL for solvers without noise (ddim, euler)              */
  float* x0;          /* (B, A)  result                                                    */
  float* xt_hist;     /* (T, B, A) or NULL: input of iteration j  (history[0], ddpm.py:217)  */
  float* eps_hist;    /* (T, B, A) or NULL: net output of iteration j (history[1])           */
  const uint32_t* rng_key; /* device (2,) uint32: the `rng` argument of sample() (a raw threefry key, as
                              jax.random.key_data gives it); used when desc.rng_impl != FRL_RNG_NONE, else ignored */
  uint32_t* rng_out;       /* device (2,) uint32 or NULL: the rng sample() returns (after the scan's splits)     */
  int64_t rng_chain_offset; /* keyed calls on a SHARD of the batch (one rank of a state-sharded job): index of this call's
                               first chain in the unsharded batch, and ...                                          */
  int64_t rng_total_chains; /* ... the unsharded batch's chain count (0 = this call is the whole batch).  The draw of
                               chain c, row a is element (offset + c) * A + a of normal(key_j, (total, A)), so the
                               shards together reproduce the reference's single call bit for bit                    */
} FrlFwdBufs;

/* FRL_OK if this build has a kernel for `desc`, else FRL_EINVAL / FRL_EUNSUPPORTED (message set). */
int frl_validate(const FrlDesc* desc);

/* Bytes of device scratch a call with this descriptor needs (tables, hoisted time branch,
 * packed edge-layer weights).  256-byte aligned pointer required.  0 on invalid descriptor. */
size_t frl_workspace_bytes(const FrlDesc* desc);

/* Stage 1 (tiny kernel): per-step coefficients in the reference's op order, the batch-invariant
 * time branch (embedding -> 2-layer mish MLP -> layer-0 time slice => a [T, H0] bias table),
 * zero-padded copies of the first/last layer weights.  Must be re-run whenever params change. */
int frl_sampler_prepare(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                        void* workspace, size_t workspace_bytes, void* hip_stream);

/* Stage 2: (cond_embedding kernel, once per state, when desc.cond_embed) + the persistent fused T-step kernel.  Needs a workspace filled by frl_sampler_prepare
 * for the same desc/params on the same stream (or an earlier, completed one).  "Same desc" EXCEPT n_states: a prepared
 * workspace serves runs at any n_states whose frl_workspace_bytes fits it.  (The kernel geometry is chosen from the batch
 * size — 16-chain tiles at two waves per SIMD from 65536 chains of a plain 256-wide sampler — so for descriptors of that
 * shape the workspace holds, and frl_sampler_prepare writes, the tables and packed edge-layer images of BOTH geometries;
 * everything prepare writes sits at offsets that do not depend on n_states.)  particles, like every other field, must
 * match the prepared descriptor. */
int frl_sampler_run(const FrlDesc* desc, const FrlParams* params, const FrlFwdBufs* bufs,
                    void* workspace, size_t workspace_bytes, void* hip_stream);

/* prepare + run: the whole of DDPM.sample / ContinuousDDPM.sample / CNF.sample minus RNG. */
int frl_sampler_fwd(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                    const FrlFwdBufs* bufs, void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- frl_sampler_fwd for callers that cannot allocate ------------------------------------------------------------
 * No counterpart in the reference: this entry point exists for callers that cannot call hipMalloc — code under HIP
 * graph capture, tight serving / acting loops, harnesses that forbid allocation.  It is frl_sampler_fwd with the
 * workspace taken from a file-scope __device__ array of FRL_STATIC_WS_BYTES inside the library (one copy per GPU, held
 * until the process ends); it calls no hipMalloc / hipFree or any allocating routine.  The caller still passes inputs
 * and outputs.  A call whose frl_workspace_bytes(desc) exceeds FRL_STATIC_WS_BYTES is refused with FRL_EWORKSPACE (the
 * per-state rows of a cond_embedding or hoisted condition grow with n_states: ~8000 states at 128 + 256 floats each).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) RACE — serialise
 * them, or use frl_sampler_fwd with one workspace per stream.  Results are bit-identical to frl_sampler_fwd's. */
#define FRL_STATIC_WS_BYTES (12u << 20)
size_t frl_static_workspace_limit(void);
int frl_sampler_fwd_static(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                           const FrlFwdBufs* bufs, void* hip_stream);

/* ---- chain VJP (DQL-style backprop through the sampler: dql.py:93-123 -> jax.grad through ddpm.py:186-219) ----
 * Given the forward's history xt_hist (want_history = 1) and the cotangent g_x0 = dL/dx0, computes dL/d(params) of the
 * predictor MLP and dL/dxT.  The batch-invariant time branch (O(T * D^2) work) hangs off grad_tb
 * (dL/d(tb), tb[j] = bias_0 + tfeat[j] @ kernel_0[A:A+D']): either the caller differentiates it from grad_tb (a JAX
 * binding gets that from jax.vjp of its own tfeat function) or the library does, see the optional outputs below.
 * This build: DDPM / ContinuousDDPM / CNF update rules (ddpm, ddim, Euler), 256-wide predictor; every activation in f32,
 * mish / silu / relu in bf16.
 *   FRL_DTYPE_F32: x_dim <= 8 and <= 24 condition rows from xt_hist alone (the backward recomputes each step) or from
 *     the stash of frl_sampler_fwd_train; x_dim <= 24, <= 128 condition rows and a cond_embedding MLP (DACER / QVPO
 *     humanoid: act 17, obs 376 -> 128) through frl_sampler_fwd_train + have_stash only.
 *   FRL_DTYPE_BF16 (bf16 operands in every contraction, f32 accumulation): x_dim <= 16, x_dim + cond_dim <= 32, no
 *     cond_embedding; from xt_hist alone (recompute) or from the stash of a bf16 frl_sampler_fwd_train.
 * Everything else returns FRL_EUNSUPPORTED. */
typedef struct FrlBwdBufs {
  const float* xt_hist; /* (T, B, A) history[0] of the forward call                                     */
  const float* cond;    /* (n_states, C) as in the forward call                                         */
  const float* g_x0;    /* (B, A) cotangent of x0                                                        */
  float* g_xT;          /* (B, A) or NULL: dL/dxT                                                        */
  float* grad_kernel[FRL_MAX_HIDDEN + 1]; /* dL/d Dense_i kernel, shapes of FrlParams.kernel; OVERWRITTEN.
                                             Rows [A, A+D') of Dense_0 (time rows) are left zero: see grad_tb */
  float* grad_bias[FRL_MAX_HIDDEN + 1];   /* dL/d Dense_i bias for i >= 1; bias_0 comes from grad_tb (sum over steps) */
  float* grad_tb;       /* (T, H) dL/d(tb)                                                                */
  /* Optional, all-or-none: when grad_tmlp_kernel[0] is non-NULL the library also finishes the time branch natively
   * (one small double-precision kernel): dL/d MLP_0/Dense_{0,1} kernel (D, D) and bias (D,), dL/d
   * time_embedding/kernel (D/2, 1) (FRL_TEMB_FOURIER; NULL otherwise), and then ALSO writes grad_bias[0] (H,) and
   * the time rows [A, A + D') of grad_kernel[0].  With FRL_TEMB_NONE set grad_bias[0] to get the raw-time row. */
  float* grad_tmlp_kernel[2];
  float* grad_tmlp_bias[2];
  float* grad_time_kernel;
  /* Set both when the forward call was frl_sampler_fwd_train on THIS workspace: the backward then reads the forward's
   * stash (act(z), act'(z) of every layer) and eps_hist instead of recomputing every step from xt_hist. */
  const float* eps_hist; /* (T, B, A) history[1] of that forward call, or NULL */
  int32_t have_stash;    /* 1: the workspace holds the stash of frl_sampler_fwd_train for the same desc / params / inputs */
  /* desc.cond_embed = 1 (DACER's actor loss, dacer.py:141-162, differentiates the cond_embedding too): dL/d
   * cond_embedding/Dense_{0,1} kernel (cond_in_dim, 128), (128, 128) and bias (128,); required then, OVERWRITTEN */
  float* grad_cond_kernel[2];
  float* grad_cond_bias[2];
} FrlBwdBufs;

size_t frl_bwd_workspace_bytes(const FrlDesc* desc);

/* The TRAINING forward of the sampler (sample(training=True, params=p), ddpm.py:191-194, as jit_update_actor calls it,
 * dql.py:105-112): frl_sampler_fwd with desc.want_history = 1 that ALSO keeps, in the backward workspace
 * (frl_bwd_workspace_bytes), what jax.grad keeps of the unrolled scan — act(z) and act'(z) of every layer and step.
 * A following frl_sampler_bwd on the same workspace with bufs.have_stash = 1 and bufs.eps_hist then runs the backward
 * sweep only.  Results (x0, xt_hist, eps_hist) agree with frl_sampler_fwd's to float32 rounding (dtype bf16: bit for bit;
 * its stash is one bf16 copy of z per layer, from which the backward recomputes act(z) / act'(z)).  The VJP's shape family
 * (see above) in either dtype; FRL_EUNSUPPORTED otherwise. */
int frl_sampler_fwd_train(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched, const FrlFwdBufs* bufs,
                          void* bwd_workspace, size_t workspace_bytes, void* hip_stream);

int frl_sampler_bwd(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched, const FrlBwdBufs* bufs,
                    void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- critic scoring + best-of-N selection: the step right after every multi-particle sample ----------------------
 * Replaces, in jit_sample_and_select (flowrl/agent/offline/dql.py:151-161, dac.py:172-182) and jit_sample_actions
 * (flowrl/agent/online/qvpo.py:39-43, sdac.py:41-45):
 *     qs = critic(obs_repeat, actions)                    Ensemblize(ScalarCritic(MLP)), flowrl/module/critic.py:154-194
 *     qs = qs.min(axis=0).reshape(B, num_samples)
 *     idx = qs.argmax(-1)   |   jax.random.categorical(key, logits = qs / (1e-6 + temperature))
 *     actions = actions.reshape(B, num_samples, -1)[arange(B), idx]
 * obs is passed UN-repeated (n_states, obs_dim); the kernel broadcasts each row over its particles.  The categorical
 * draw is argmax(logits + Gumbel noise): the caller supplies the noise (drawn with the reference's key), like the
 * sampler's N(0, I) noise; gumbel = NULL selects the plain argmax (first maximum, as jnp.argmax). */
typedef struct FrlCriticDesc {
  int32_t abi_version; /* FRL_ABI_VERSION */
  int32_t particles;   /* num_samples: candidate actions per state */
  int64_t n_states;
  int32_t obs_dim;     /* S: rows [0, S) of the backbone's Dense_0 kernel (critic.py:187: concat([obs, action])) */
  int32_t act_dim;     /* A: rows [S, S + A) */
  int32_t n_hidden;    /* hidden Dense layers of the backbone MLP */
  int32_t hidden[FRL_MAX_HIDDEN];
  int32_t act;         /* FRL_ACT_* (mish in DQL / DAC, relu in QVPO: dql.py:227, qvpo.yaml:7) */
  int32_t ensemble;    /* E: ensemble_size */
} FrlCriticDesc;

/* Leaves of Model.create(Ensemblize(ScalarCritic(MLP))): every leaf carries the ensemble axis first (nn.vmap,
 * variable_axes = {"params": 0}). */
typedef struct FrlCriticParams {
  const float* kernel[FRL_MAX_HIDDEN + 1]; /* backbone/Dense_i/kernel (E, in, H), i < L; [L] = the head Dense_0/kernel (E, H, 1) */
  const float* bias[FRL_MAX_HIDDEN + 1];   /* backbone/Dense_i/bias (E, H); [L] = the head bias (E, 1) */
} FrlCriticParams;

typedef struct FrlSelectBufs {
  const float* obs;      /* (n_states, S), NULL iff obs_dim == 0 */
  const float* actions;  /* (n_states * particles, A): the sampler's x0 */
  const float* gumbel;   /* (n_states, particles) Gumbel(0, 1) noise or NULL */
  float inv_temperature; /* 1 / (1e-6 + temperature); used only with gumbel */
  float* q;              /* (n_states, particles) min over the ensemble, or NULL */
  int32_t* best_idx;     /* (n_states,) or NULL */
  float* best_action;    /* (n_states, A) or NULL */
} FrlSelectBufs;

size_t frl_critic_workspace_bytes(const FrlCriticDesc* desc);

/* E launches of the fused MFMA network kernel (one evaluation per chain, no update rule) + one selection kernel. */
int frl_critic_select(const FrlCriticDesc* desc, const FrlCriticParams* params, const FrlSelectBufs* bufs,
                      void* workspace, size_t workspace_bytes, void* hip_stream);

/* Sampler + critic scoring + selection in ONE launch (SURVEY 8(f) rank 1 as written: jit_sample_and_select dql.py:130-162,
 * jit_sample_actions qvpo.py:21-44).  frl_sampler_fwd followed by frl_critic_select, except that the candidate actions
 * never leave the chip: after the last step the E critic members are evaluated on each tile's x0 in registers, the minimum
 * over the ensemble and the argmax / Gumbel-max over a state's particles are taken in the kernel, and only best_action
 * (n_states, A) / best_idx (n_states,) / optionally q (n_states, particles) are written.  fwd.x0 may be NULL (it is written
 * when given); sel.actions is ignored.  This build: f32 MLP sampler, 256 wide, no LayerNorm, caller-supplied noise; critic
 * backbone 256 wide, <= 4 hidden layers, <= 4 members, activation = the sampler's or relu (QVPO: mish actor, relu critic);
 * particles a power of two <= 32 or a multiple of 32; cdesc.n_states / particles / act_dim = desc.n_states / particles /
 * x_dim.  Anything else: FRL_EUNSUPPORTED (use the two calls). */
size_t frl_sampler_select_workspace_bytes(const FrlDesc* desc, const FrlCriticDesc* cdesc);
int frl_sampler_select_fwd(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched, const FrlFwdBufs* fwd,
                           const FrlCriticDesc* cdesc, const FrlCriticParams* cparams, const FrlSelectBufs* sel,
                           void* workspace, size_t workspace_bytes, void* hip_stream);

/* Algorithmic FLOPs of one call, SURVEY.md section 8(d): 2*B*T*(K0*H + sum H_l*H_{l+1} + H*A) with
 * the reference's full K0 = A + D' + C; `executed` (may be NULL) receives the MFMA FLOPs the kernel
 * really issues (time slice hoisted, edge layers zero-padded). */
double frl_algorithmic_flops(const FrlDesc* desc, double* executed);

/* FLOPs of the algorithmic count above that this build does NOT execute per chain because the condition rows of
 * Dense_0 are contracted once per state instead (many chains per state, or a condition beyond the operand budget):
 * 2 * (B * evaluations - n_states) * C * H, else 0.  SURVEY.md 8(d): subtract it when quoting a fraction of peak. */
double frl_hoisted_flops(const FrlDesc* desc);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* frl_last_error(void);

/* Diagnostics (no reference counterpart): the compute kernels the LAST sampler / VJP / evaluation / critic entry point called
 * on this thread actually enqueued, as "name<instantiation> [grid] + ..." (thread-local; "" before any call).  Written by
 * the launchers themselves, so a benchmark that reports which build ran cannot drift from the dispatch rules. */
const char* frl_last_kernels(void);

/* ABI version this library was built with. */
/* jax.random pieces for the callers either side of the sampler (dql.py:142-146 draws xT with its own split before
 * calling sample): keys are device (2,) uint32, impl is FRL_RNG_THREEFRY_*.
 *   frl_random_split   out (num, 2) uint32 = jax.random.split(key, num)
 *   frl_random_normal  out (n,) f32        = jax.random.normal(key, shape) flattened, n = prod(shape)
 *   frl_random_bits    out (n,) uint32     = jax.random.bits(key, shape) flattened
 *   frl_random_gumbel  out (n,) f32        = jax.random.gumbel(key, shape) flattened: -log(-log(uniform(tiny, 1))), the
 *                                            noise of jax.random.categorical (dql.py:159-160) -> FrlSelectBufs.gumbel  */
int frl_random_split(const uint32_t* key, int32_t impl, int32_t num, uint32_t* out, void* hip_stream);
int frl_random_normal(const uint32_t* key, int32_t impl, int64_t n, float* out, void* hip_stream);
int frl_random_bits(const uint32_t* key, int32_t impl, int64_t n, uint32_t* out, void* hip_stream);
int frl_random_gumbel(const uint32_t* key, int32_t impl, int64_t n, float* out, void* hip_stream);

int frl_random_uniform(const uint32_t* key, int32_t impl, int64_t n, float minval, float maxval, float* out, void* hip_stream);
int frl_random_randint(const uint32_t* key, int32_t impl, int64_t n, int32_t minval, int32_t maxval, int32_t* out, void* hip_stream);

/* ---- the training-side half of the flow models --------------------------------------------------
 * frl_net_eval: ONE network evaluation per sample with a PER-SAMPLE time input — `self(xt, t, condition)` as
 * onestep_sample (ddpm.py:249-254) and the epsilon-MSE / flow-matching losses call it (ddpm.py:286-293,
 * continuous_ddpm.py:234-241, cnf.py:155-162).  desc describes the network (x_dim, cond_dim / cond_embed, time_embed,
 * hidden, act, layer_norm, n_states x particles samples; kind / solver / steps / clip are ignored); f32. */
typedef struct FrlEvalBufs {
  const float* x;    /* (B, A) network input x (xt)                                                         */
  const float* cond; /* (n_states, C) un-repeated condition rows ((n_states, cond_in_dim) with cond_embed)  */
  const float* t;    /* (B,) time input per sample AS THE NETWORK SEES IT: float(t) for DDPM's integer steps */
  float* out;        /* (B, A) eps_theta / velocity                                                          */
} FrlEvalBufs;
size_t frl_net_eval_workspace_bytes(const FrlDesc* desc);
int frl_net_eval(const FrlDesc* desc, const FrlParams* params, const FrlEvalBufs* bufs, void* workspace, size_t workspace_bytes,
                 void* hip_stream);

/* The same evaluation under jax.grad — `model.apply({"params": p}, xt, t, condition, training=True)` inside the loss
 * functions of jit_update_ddpm / jit_update_continuous_ddpm / jit_update_flow_matching (ddpm.py:286-296,
 * continuous_ddpm.py:234-243, cnf.py:155-165) and of every agent that adds its own epsilon / velocity loss on top of
 * add_noise (dql.py:88-92, dac.py:113, qvpo.py:140, sdac.py:91, ...): frl_net_eval_train is frl_net_eval that also keeps
 * act(z) / act'(z) of every layer in the workspace; frl_net_eval_bwd then takes dL/d out and writes dL/d of EVERY
 * parameter leaf (predictor, MLP_0, time_embedding, cond_embedding) and optionally dL/dx.  f32, deterministic.  Plain 256-wide
 * MLP predictors (x_dim <= 24, x_dim + time_dim + cond_dim <= 224) run the fused sweep of the chain VJP with T = 1; every
 * other predictor the forward evaluates — 512-wide (FQL's actor_bc, fql/fql.py:83-123,189-211), LayerNorm in both orders,
 * ResidualMLP / Simba (DAC with diffusion.resnet, dac.py:200-207) — runs layer by layer (csrc/net_vjp_layered.hip) and
 * also needs the LayerNorm / residual-block gradient pointers below. */
typedef struct FrlEvalBwdBufs {
  const float* x;     /* (B, A)   as the forward call */
  const float* cond;  /* (n_states, C | cond_in_dim) */
  const float* t;     /* (B,) */
  const float* g_out; /* (B, A) cotangent of the evaluation's output */
  float* g_x;         /* (B, A) or NULL: dL/dx */
  float* grad_kernel[FRL_MAX_HIDDEN + 1]; /* dL/d Dense_i kernel, ALL rows of Dense_0 (x | time | cond); OVERWRITTEN */
  float* grad_bias[FRL_MAX_HIDDEN + 1];   /* dL/d Dense_i bias, i = 0..L */
  float* grad_tmlp_kernel[2];             /* dL/d MLP_0/Dense_{0,1} (time_embed != FRL_TEMB_NONE) */
  float* grad_tmlp_bias[2];
  float* grad_time_kernel;                /* dL/d time_embedding/kernel (FRL_TEMB_FOURIER) */
  float* grad_cond_kernel[2];             /* dL/d cond_embedding/Dense_{0,1} (desc.cond_embed) */
  float* grad_cond_bias[2];
  /* desc.layer_norm != 0, MLP backbone: dL/d <predictor>/LayerNorm_i/{scale,bias}, i = 0..L-1 */
  float* grad_ln_scale[FRL_MAX_HIDDEN];
  float* grad_ln_bias[FRL_MAX_HIDDEN];
  /* residual backbones: grad_kernel[0..1] / grad_bias[0..1] = Dense_0 / Dense_1 of the predictor; per block i its Dense_0,
   * Dense_1 and (when it normalises) LayerNorm_0; Simba's final LayerNorm_0 */
  float* grad_res_kernel[FRL_MAX_HIDDEN][2];
  float* grad_res_bias[FRL_MAX_HIDDEN][2];
  float* grad_res_ln_scale[FRL_MAX_HIDDEN];
  float* grad_res_ln_bias[FRL_MAX_HIDDEN];
  float* grad_final_ln_scale;
  float* grad_final_ln_bias;
} FrlEvalBwdBufs;
size_t frl_net_eval_bwd_workspace_bytes(const FrlDesc* desc);
int frl_net_eval_train(const FrlDesc* desc, const FrlParams* params, const FrlEvalBufs* bufs, void* bwd_workspace, size_t workspace_bytes,
                       void* hip_stream);
int frl_net_eval_bwd(const FrlDesc* desc, const FrlParams* params, const FrlEvalBwdBufs* bufs, void* bwd_workspace, size_t workspace_bytes,
                     void* hip_stream);

/* Forward process, elementwise over (B, A):
 *   FRL_NOISE_DDPM          xt = sqrt(alpha_hats[t]) x0 + sqrt(1 - alpha_hats[t]) eps, t (B,) int32   ddpm.py:164-171
 *   FRL_NOISE_CDDPM_LINEAR  alpha, sigma = linear_noise_schedule(t, beta0 = p0, beta1 = p1)            continuous_ddpm.py:19-24,107-115
 *   FRL_NOISE_CDDPM_COSINE  alpha, sigma = cosine_noise_schedule(t, s = p0); xt = alpha x0 + sigma eps  continuous_ddpm.py:26-31
 *   FRL_NOISE_FLOW          xt = (1 - t) x0 + t x1, vel = x1 - x0  (e = x1; vel may be NULL)            cnf.py:85-90
 * t is (B,) int32 for FRL_NOISE_DDPM, (B,) float32 otherwise. */
#define FRL_NOISE_DDPM 0
#define FRL_NOISE_CDDPM_LINEAR 1
#define FRL_NOISE_CDDPM_COSINE 2
#define FRL_NOISE_FLOW 3
int frl_add_noise(int32_t kind, int64_t B, int32_t A, const float* x0, const float* e, const void* t, const float* alpha_hats,
                  int32_t steps, float p0, float p1, float* xt, float* vel, void* hip_stream);

/* DDPM.onestep_sample's update (ddpm.py:256-274) for per-sample integer t in [1, steps]: xt_1 from (xt, eps_theta), each
 * row repeated num_samples times; solver ddpm adds (t > 1) sqrt(1 - alphas[t]) noise (noise (B, num_samples, A)), ddim clips.
 * out is (B, num_samples, A). */
int frl_ddpm_onestep_update(int32_t solver, int64_t B, int32_t A, int32_t num_samples, const float* xt, const float* eps_theta,
                            const int32_t* t, const float* alphas, const float* alpha_hats, int32_t steps, int32_t clip, float x_min,
                            float x_max, const float* noise, float* out, void* hip_stream);

/* loss = ((pred - target) ** 2).mean() over n elements (ddpm.py:294, continuous_ddpm.py:242, cnf.py:163) into the device
 * scalar `loss`; grad (n,) or NULL receives dloss/dpred = 2 (pred - target) / n.  Deterministic (fixed-order sums);
 * workspace: FRL_MSE_WORKSPACE_BYTES. */
#define FRL_MSE_WORKSPACE_BYTES 2048
int frl_mse_loss(int64_t n, const float* pred, const float* target, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                 void* hip_stream);

int frl_abi_version(void);

/* Hardware self-test of the MFMA operand / accumulator lane maps the kernel relies on:
 * D(32x32) = A(32xK) * B(Kx32) with v_mfma_f32_32x32x2_f32, row-major f32 device buffers,
 * K even.  Used by tests/ only. */
int frl_selftest_mfma32(const float* A, const float* B, float* D, int K, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWRL_HIP_H_ */
