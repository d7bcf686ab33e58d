// C ABI of libflowrl_hip.so (include/flowrl_hip.h): validation, planning, kernel dispatch.
// Enqueue-only: no allocation, no host synchronisation, no retained pointers.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "frl_internal.h"

// Workspace of frl_sampler_fwd_static: device memory that exists from module load to process end, one copy per GPU.
// In host code this name is only a handle for hipGetSymbolAddress; the device address is static_scratch()'s result.
static __device__ __align__(256) unsigned char scratch[FRL_STATIC_WS_BYTES];

namespace frl {

constexpr int FRL_ACT_IDENTITY_BUILD = 6;     // = FRL_ACT_IDENTITY of frl_device.h: the kernels' post-activation-LayerNorm units

static thread_local char g_err[512] = "";

// Device address of `scratch` on the CURRENT device, resolved once per device.
static int static_scratch(void** out) {
  constexpr int MAXDEV = 64;
  static std::mutex mu;
  static void* ptrs[MAXDEV] = {nullptr};
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) { set_error("hipGetDevice: %s", hipGetErrorString(e)); return FRL_EHIP; }
  if (dev < 0 || dev >= MAXDEV) { set_error("device index %d outside [0, %d)", dev, MAXDEV); return FRL_EINVAL; }
  std::lock_guard<std::mutex> lock(mu);
  if (ptrs[dev] == nullptr) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) { set_error("hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e)); return FRL_EHIP; }
    ptrs[dev] = p;
  }
  *out = ptrs[dev];
  return FRL_OK;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_kern[1024] = "";
static void clear_kernels() { g_kern[0] = 0; }
void note_kernel(const char* fmt, ...) {
  const size_t n = strlen(g_kern);
  if (n + 4 >= sizeof(g_kern)) return;
  size_t o = n;
  if (n > 0) { memcpy(g_kern + n, " + ", 3); o = n + 3; }
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kern + o, sizeof(g_kern) - o, fmt, ap);
  va_end(ap);
}

// f32 network kernel dispatch: latency build for small batches, throughput build otherwise
static int launch_f32_network(const KArgs& a, int act, int tile_policy, hipStream_t stream) {
  if (a.ln_post_act != 0) act = FRL_ACT_IDENTITY_BUILD;     // post-activation LayerNorm: the pass applies the activation itself
  const long long ntiles = (a.B + 31) / 32;
  const long long cus = device_cus();
  bool lat = false;
  int tile_rows = 0;   // latency build: 0 = its own choice between 16- and 32-chain tiles
  if (tile_policy == FRL_TILE_LATENCY || tile_policy == FRL_TILE_LATENCY_32 || tile_policy == FRL_TILE_LATENCY_16) {
    tile_rows = tile_policy == FRL_TILE_LATENCY_32 ? 32 : tile_policy == FRL_TILE_LATENCY_16 ? 16 : 0;
    const bool ok = tile_rows == 32 ? sampler_lat_supports(a) : tile_rows == 16 ? sampler_lat16_supports(a)
                                    : (sampler_lat_supports(a) || sampler_lat16_supports(a));
    if (!ok) { set_error("tile_policy %d: this configuration has no such latency build (H %d, x_dim %d, operand cond rows %d, x rows %d, evaluations %d, LDS %zu bytes)", tile_policy, a.H, a.A, a.C, a.XR, a.NE, lat_lds_bytes(a.NE, a.L, a.H, false)); return FRL_EUNSUPPORTED; }
    lat = true;
  } else if (tile_policy == FRL_TILE_AUTO) {
    // Measured on MI355X (cfg2 geometry, scripts/sweep_tile.sh): one round of latency-build tiles (one per CU) takes
    // 0.12 ms, one round of throughput-build tiles (four per CU) 0.38 ms.  The latency build wins whenever its
    // rounds cost less, i.e. for small batches and for batches whose last throughput round would be mostly empty;
    // below 16 chains per CU its 16-chain tiles halve the round again.
    const long long r_lat = (ntiles + cus - 1) / cus, r_thr = (ntiles + 4 * cus - 1) / (4 * cus);
    lat = (sampler_lat16_supports(a) && (a.B + 15) / 16 <= cus) || (r_lat <= 3 * r_thr && sampler_lat_supports(a));
  }
  if (a.TR == 16 && a.H == 256) lat = false;   // Plan picked the two-wave throughput build (plan_two_wave): the packed images are its

  if (lat) {
    switch (act) {
      case FRL_ACT_MISH: return launch_sampler_lat_act<FRL_ACT_MISH>(a, tile_rows, stream);
      case FRL_ACT_SILU: return launch_sampler_lat_act<FRL_ACT_SILU>(a, tile_rows, stream);
      case FRL_ACT_RELU: return launch_sampler_lat_act<FRL_ACT_RELU>(a, tile_rows, stream);
      case FRL_ACT_GELU_TANH: return launch_sampler_lat_act<FRL_ACT_GELU_TANH>(a, tile_rows, stream);
      case FRL_ACT_TANH: return launch_sampler_lat_act<FRL_ACT_TANH>(a, tile_rows, stream);
      case FRL_ACT_IDENTITY_BUILD: return launch_sampler_lat_act<FRL_ACT_IDENTITY_BUILD>(a, tile_rows, stream);
      default: return launch_sampler_lat_act<FRL_ACT_ELU>(a, tile_rows, stream);
    }
  }
  switch (act) {
    case FRL_ACT_MISH: return launch_sampler_rs_act<FRL_ACT_MISH>(a, stream);
    case FRL_ACT_SILU: return launch_sampler_rs_act<FRL_ACT_SILU>(a, stream);
    case FRL_ACT_RELU: return launch_sampler_rs_act<FRL_ACT_RELU>(a, stream);
    case FRL_ACT_GELU_TANH: return launch_sampler_rs_act<FRL_ACT_GELU_TANH>(a, stream);
    case FRL_ACT_TANH: return launch_sampler_rs_act<FRL_ACT_TANH>(a, stream);
    case FRL_ACT_IDENTITY_BUILD: return launch_sampler_rs_act<FRL_ACT_IDENTITY_BUILD>(a, stream);
    default: return launch_sampler_rs_act<FRL_ACT_ELU>(a, stream);
  }
}

// residual backbones (ResidualMLP / Simba), csrc/sampler_res.hip: four waves per tile (latency build) while its rounds cost
// less than the throughput build's — one tile per CU against four, at about a quarter of the time per round
static int launch_res_network(const KArgs& a, const ResArgs& ra_in, int act, int tile_policy, hipStream_t stream) {
  ResArgs ra = ra_in;
  const long long ntiles = (a.B + 31) / 32, cus = device_cus();
  const long long r_lat = (ntiles + cus - 1) / cus, r_thr = (ntiles + 4 * cus - 1) / (4 * cus);
  ra.split = (tile_policy == FRL_TILE_LATENCY || tile_policy == FRL_TILE_LATENCY_32 ||
              (tile_policy == FRL_TILE_AUTO && r_lat <= 3 * r_thr && res_lds_bytes(a.NE, ra.n_blocks, 4) <= 160 * 1024)) ? 4 : 1;
  switch (act) {
    case FRL_ACT_MISH: return launch_sampler_res_act<FRL_ACT_MISH>(a, ra, stream);
    case FRL_ACT_SILU: return launch_sampler_res_act<FRL_ACT_SILU>(a, ra, stream);
    case FRL_ACT_RELU: return launch_sampler_res_act<FRL_ACT_RELU>(a, ra, stream);
    case FRL_ACT_GELU_TANH: return launch_sampler_res_act<FRL_ACT_GELU_TANH>(a, ra, stream);
    case FRL_ACT_TANH: return launch_sampler_res_act<FRL_ACT_TANH>(a, ra, stream);
    default: return launch_sampler_res_act<FRL_ACT_ELU>(a, ra, stream);
  }
}

static int fill_res_args(const FrlDesc* d, const FrlParams* p, ResArgs& ra) {
  memset(&ra, 0, sizeof(ra));
  ra.n_blocks = d->n_hidden;
  const bool simba = d->backbone == FRL_BACKBONE_SIMBA;
  ra.ln = (simba || d->layer_norm) ? 1 : 0;
  ra.final_ln = simba ? 1 : 0;
  ra.final_act = simba ? 0 : 1;
  for (int i = 0; i < d->n_hidden; ++i) {
    for (int k = 0; k < 2; ++k)
      if (p->res_kernel[i][k] == nullptr || p->res_bias[i][k] == nullptr) { set_error("params.res_kernel/res_bias[%d][%d] is NULL", i, k); return FRL_EINVAL; }
    ra.w1[i] = p->res_kernel[i][0]; ra.b1[i] = p->res_bias[i][0];
    ra.w2[i] = p->res_kernel[i][1]; ra.b2[i] = p->res_bias[i][1];
    if (ra.ln) {
      if (p->res_ln_scale[i] == nullptr || p->res_ln_bias[i] == nullptr) { set_error("params.res_ln_scale/res_ln_bias[%d] is NULL", i); return FRL_EINVAL; }
      ra.ln_scale[i] = p->res_ln_scale[i]; ra.ln_bias[i] = p->res_ln_bias[i];
    }
  }
  if (simba) {
    if (p->final_ln_scale == nullptr || p->final_ln_bias == nullptr) { set_error("params.final_ln_scale/final_ln_bias is NULL (Simba)"); return FRL_EINVAL; }
    ra.fln_scale = p->final_ln_scale; ra.fln_bias = p->final_ln_bias;
  }
  return FRL_OK;
}

static int validate_desc(const FrlDesc* d) {
  if (d == nullptr) { set_error("desc is NULL"); return FRL_EINVAL; }
  if (d->abi_version != FRL_ABI_VERSION) { set_error("desc.abi_version %d != library %d", d->abi_version, FRL_ABI_VERSION); return FRL_EINVAL; }
  if (d->kind < FRL_KIND_DDPM || d->kind > FRL_KIND_CNF) { set_error("desc.kind %d invalid", d->kind); return FRL_EINVAL; }
  if (d->kind == FRL_KIND_CNF) {
    if (d->solver != FRL_SOLVER_EULER) { set_error("CNF sampler supports solver EULER only (cnf.py:92-111)"); return FRL_EINVAL; }
  } else if (d->solver != FRL_SOLVER_DDPM && d->solver != FRL_SOLVER_DDIM) {
    const bool dpm = d->solver == FRL_SOLVER_DPM2 || d->solver == FRL_SOLVER_DPMPP2M;
    if (!dpm || d->kind != FRL_KIND_CDDPM) {   // ddpm.py:16-18: the discrete sampler knows ddim / ddpm only
      set_error("solver %d invalid for kind %d (DDPM: ddpm / ddim; ContinuousDDPM: ddpm / ddim / dpm2 / dpm++2m)", d->solver, d->kind);
      return FRL_EINVAL;
    }
    if (d->dtype != FRL_DTYPE_F32) { set_error("solvers dpm2 / dpm++2m run in the f32 kernel only"); return FRL_EUNSUPPORTED; }
    if (d->solver == FRL_SOLVER_DPM2 && d->steps > 128) { set_error("dpm2: steps %d > 128 (two table rows per step)", d->steps); return FRL_EUNSUPPORTED; }
  }
  if (d->steps < 1 || d->steps > 256) { set_error("desc.steps %d outside [1, 256]", d->steps); return FRL_EINVAL; }
  if (d->n_states < 0 || d->particles < 1) { set_error("n_states %lld / particles %d invalid", (long long)d->n_states, d->particles); return FRL_EINVAL; }
  if ((long long)d->n_states * d->particles > (1LL << 31) - 1 - 256) { set_error("more than 2^31 chains in one call"); return FRL_EINVAL; }
  if (d->x_dim < 1) { set_error("x_dim %d invalid", d->x_dim); return FRL_EINVAL; }
  if (d->cond_dim < 0) { set_error("cond_dim %d invalid", d->cond_dim); return FRL_EINVAL; }
  if (d->time_embed < FRL_TEMB_NONE || d->time_embed > FRL_TEMB_FOURIER) { set_error("time_embed %d invalid", d->time_embed); return FRL_EINVAL; }
  if (d->time_embed != FRL_TEMB_NONE && (d->time_dim < 4 || (d->time_dim & 1))) { set_error("time_dim %d must be even and >= 4", d->time_dim); return FRL_EINVAL; }
  if (d->time_embed != FRL_TEMB_NONE && d->time_dim > 4096) { set_error("time_dim %d > 4096 (prepare kernel's LDS rows)", d->time_dim); return FRL_EUNSUPPORTED; }
  if (d->n_hidden < 1 || d->n_hidden > FRL_MAX_HIDDEN) { set_error("n_hidden %d outside [1, %d]", d->n_hidden, FRL_MAX_HIDDEN); return FRL_EINVAL; }
  for (int l = 0; l < d->n_hidden; ++l)
    if (d->hidden[l] < 1) { set_error("hidden[%d] = %d invalid", l, d->hidden[l]); return FRL_EINVAL; }
  if (d->act < FRL_ACT_MISH || d->act > FRL_ACT_ELU) { set_error("act %d invalid", d->act); return FRL_EINVAL; }
  if (d->clip && !(d->x_min <= d->x_max)) { set_error("clip range [%g, %g] invalid", d->x_min, d->x_max); return FRL_EINVAL; }
  if (d->dtype != FRL_DTYPE_F32 && d->dtype != FRL_DTYPE_BF16) { set_error("dtype %d invalid", d->dtype); return FRL_EINVAL; }
  if (d->tile_policy < FRL_TILE_AUTO || d->tile_policy > FRL_TILE_LATENCY_16) { set_error("tile_policy %d invalid", d->tile_policy); return FRL_EINVAL; }
  if (d->layer_norm < 0 || d->layer_norm > FRL_LN_POST_ACT) { set_error("layer_norm %d invalid (0, FRL_LN_PRE_ACT, FRL_LN_POST_ACT)", d->layer_norm); return FRL_EINVAL; }
  if (d->layer_norm) {
    if (d->dtype != FRL_DTYPE_F32) { set_error("layer_norm: f32 kernel only"); return FRL_EUNSUPPORTED; }
  }
  if (d->rng_impl < FRL_RNG_NONE || d->rng_impl > FRL_RNG_THREEFRY_LEGACY) { set_error("rng_impl %d invalid", d->rng_impl); return FRL_EINVAL; }
  if (d->rng_impl == FRL_RNG_THREEFRY_LEGACY && (long long)d->n_states * d->particles * d->x_dim > 0xFFFFFFFFll) {
    set_error("legacy threefry layout: at most 2^32 - 1 elements per draw"); return FRL_EUNSUPPORTED; }
  if (d->cond_embed) {
    if (d->cond_dim != COND_EMBED_DIM) { set_error("cond_embed: cond_dim must be %d (MLP(128,128)), got %d", COND_EMBED_DIM, d->cond_dim); return FRL_EUNSUPPORTED; }
    if (d->cond_in_dim < 1) { set_error("cond_embed: cond_in_dim %d invalid", d->cond_in_dim); return FRL_EINVAL; }
  }
  if (d->backbone < FRL_BACKBONE_MLP || d->backbone > FRL_BACKBONE_SIMBA) { set_error("backbone %d invalid", d->backbone); return FRL_EINVAL; }
  // --- what this build has kernels for ---
  if (d->backbone != FRL_BACKBONE_MLP) {
    // csrc/sampler_res.hip: 256-wide blocks, expansion 4, f32 throughput build, forward
    if (d->dtype != FRL_DTYPE_F32) { set_error("residual backbones: f32 kernel only"); return FRL_EUNSUPPORTED; }
    if (d->res_multiplier != 4) { set_error("residual backbones: multiplier %d (4 is the only value built)", d->res_multiplier); return FRL_EUNSUPPORTED; }
    for (int l = 0; l < d->n_hidden; ++l)
      if (d->hidden[l] != 256) { set_error("residual backbones: hidden widths must all be 256 (hidden[%d] = %d)", l, d->hidden[l]); return FRL_EUNSUPPORTED; }
    if (d->layer_norm == FRL_LN_POST_ACT) { set_error("residual backbones: layer_norm is 0 or FRL_LN_PRE_ACT (ResidualLinear normalises its input)"); return FRL_EINVAL; }
    if (d->x_dim > 24) { set_error("residual backbones: x_dim %d > 24", d->x_dim); return FRL_EUNSUPPORTED; }
    if (d->tile_policy == FRL_TILE_LATENCY_16) { set_error("residual backbones: the latency build runs 32-chain tiles (tile_policy LATENCY / LATENCY_32)"); return FRL_EUNSUPPORTED; }
    if (res_lds_bytes((d->solver == FRL_SOLVER_DPM2 ? 2 : 1) * d->steps, d->n_hidden, (d->tile_policy == FRL_TILE_LATENCY || d->tile_policy == FRL_TILE_LATENCY_32) ? 4 : 1) > 160 * 1024) { set_error("residual backbones: %d steps x %d blocks exceed the LDS tables", d->steps, d->n_hidden); return FRL_EUNSUPPORTED; }
    return FRL_OK;
  }
  if (d->dtype == FRL_DTYPE_BF16) {
    // bf16 MFMA variant (csrc/sampler_bf16.hip): 256- or 512-wide predictor, bf16 operands / f32 accumulate / f32 chain state
    const int hw = d->hidden[0];
    if (hw != 256 && hw != 512) { set_error("bf16 variant: hidden width 256 or 512 required (hidden[0] = %d)", hw); return FRL_EUNSUPPORTED; }
    for (int l = 0; l < d->n_hidden; ++l)
      if (d->hidden[l] != hw) { set_error("bf16 variant: hidden widths must all be equal (hidden[%d] = %d, hidden[0] = %d)", l, d->hidden[l], hw); return FRL_EUNSUPPORTED; }
    if (hw == 256 && (d->x_dim > 32 || d->cond_dim > 128)) { set_error("bf16 variant: x_dim <= 32 and cond_dim <= 128 required (got %d, %d)", d->x_dim, d->cond_dim); return FRL_EUNSUPPORTED; }
    if (hw == 512 && (d->x_dim > 16 || d->cond_dim > 64)) { set_error("bf16 variant, 512 wide: x_dim <= 16 and cond_dim <= 64 required (got %d, %d)", d->x_dim, d->cond_dim); return FRL_EUNSUPPORTED; }
    if (d->act != FRL_ACT_MISH && d->act != FRL_ACT_SILU && d->act != FRL_ACT_RELU && d->act != FRL_ACT_GELU_TANH) {
      set_error("bf16 variant: activation %d not compiled in (mish / silu / relu / gelu_tanh)", d->act); return FRL_EUNSUPPORTED; }
    if (bf16_lds_bytes(d->steps, d->n_hidden, hw) > 160 * 1024) { set_error("bf16 variant: steps = %d does not fit the LDS tables", d->steps); return FRL_EUNSUPPORTED; }
    return FRL_OK;
  }
  for (int l = 0; l < d->n_hidden; ++l)
    if (d->hidden[l] != d->hidden[0] || (d->hidden[l] != 256 && d->hidden[l] != 512)) {
      set_error("hidden widths must be uniform and 256 or 512 in this build (hidden[%d] = %d)", l, d->hidden[l]);
      return FRL_EUNSUPPORTED;
    }
  {
    Plan pl;
    plan_from_desc(*d, pl);
    // (a condition wider than the budget is hoisted per state, Plan.hoist: CK = 0 then)
    const bool ok = (pl.TR == 32) ? ((pl.AR <= 4 && pl.CK <= 12) || (pl.AR <= 12 && pl.CK <= 64)) : (pl.AR <= 4 && pl.CK <= 12);
    if (!ok) {
      set_error("x_dim %d exceeds the compiled operand budget for hidden width %d (256: x_dim <= 24; 512: x_dim <= 16)", d->x_dim, pl.H);
      return FRL_EUNSUPPORTED;
    }
    if (pl.hoist && pl.H % 128 != 0) { set_error("hidden width %d: per-state hoist needs a multiple of 128", pl.H); return FRL_EUNSUPPORTED; }
  }
  return FRL_OK;
}

static int check_ws(const Plan& pl, const void* ws, size_t bytes) {
  if (ws == nullptr) { set_error("workspace is NULL"); return FRL_EWORKSPACE; }
  if (((uintptr_t)ws & 255) != 0) { set_error("workspace must be 256-byte aligned"); return FRL_EWORKSPACE; }
  if (bytes < pl.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B", bytes, pl.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  return FRL_OK;
}

}  // namespace frl

using namespace frl;

extern "C" {

int frl_abi_version(void) { return FRL_ABI_VERSION; }

const char* frl_last_error(void) { return g_err; }

const char* frl_last_kernels(void) { return g_kern; }

int frl_validate(const FrlDesc* desc) { return validate_desc(desc); }

size_t frl_workspace_bytes(const FrlDesc* desc) {
  if (validate_desc(desc) != FRL_OK) return 0;
  Plan pl;
  plan_from_desc(*desc, pl, PLAN_HOIST | PLAN_W2);
  return pl.total_floats * sizeof(float);
}

double frl_algorithmic_flops(const FrlDesc* desc, double* executed) {
  if (desc == nullptr || desc->n_hidden < 1 || desc->n_hidden > FRL_MAX_HIDDEN) return 0.0;
  const double B = (double)desc->n_states * desc->particles;
  const double T = (desc->solver == FRL_SOLVER_DPM2 ? 2.0 : 1.0) * desc->steps;   // network evaluations per chain
  const int A = desc->x_dim, C = desc->cond_dim;
  const int Dp = desc->time_embed == FRL_TEMB_NONE ? 1 : desc->time_dim;
  double macs = (double)(A + Dp + C) * desc->hidden[0];
  if (desc->backbone != FRL_BACKBONE_MLP) {     // Dense_0, per block H x 4H + 4H x H, Dense_1
    const double H = desc->hidden[0], blk = 2.0 * H * desc->res_multiplier * H;
    const double cf = desc->cond_embed ? 2.0 * (double)desc->n_states * ((double)desc->cond_in_dim * COND_EMBED_DIM + (double)COND_EMBED_DIM * COND_EMBED_DIM) : 0.0;
    if (executed != nullptr) {
      Plan pl;
      plan_from_desc(*desc, pl, PLAN_HOIST | PLAN_W2);
      *executed = 2.0 * B * T * ((double)pl.XR * H + desc->n_hidden * blk + H * pl.TR) + cf + 2.0 * (double)desc->n_states * C * H;
    }
    return 2.0 * B * T * (macs + desc->n_hidden * blk + H * A) + cf;
  }
  const double cond_flops = desc->cond_embed ? 2.0 * (double)desc->n_states * ((double)desc->cond_in_dim * COND_EMBED_DIM + (double)COND_EMBED_DIM * COND_EMBED_DIM) : 0.0;
  for (int l = 1; l < desc->n_hidden; ++l) macs += (double)desc->hidden[l - 1] * desc->hidden[l];
  macs += (double)desc->hidden[desc->n_hidden - 1] * A;
  if (executed != nullptr) {
    Plan pl;
    plan_from_desc(*desc, pl, PLAN_HOIST | PLAN_W2);
    double em = (double)(pl.XR + pl.KS * pl.CK) * pl.H;         // Dense_0 without the hoisted time slice, padded
    for (int l = 1; l < desc->n_hidden; ++l) em += (double)desc->hidden[l - 1] * desc->hidden[l];
    em += (double)pl.H * pl.TR;                                   // Dense_L padded to one MFMA block (32 or 16 rows: the geometry the plan picks)
    if (pl.bf16) em += (double)(16 * (pl.bf_nxk + pl.bf_cks) - (pl.XR + pl.KS * pl.CK)) * pl.H;   // bf16: 16-wide k-steps
    *executed = 2.0 * B * T * em + cond_flops + (pl.hoist ? 2.0 * (double)desc->n_states * pl.C * pl.H : 0.0);   // hoisted cond rows: once per state
  }
  return 2.0 * B * T * macs + cond_flops;
}

double frl_hoisted_flops(const FrlDesc* desc) {
  if (validate_desc(desc) != FRL_OK) return 0.0;
  Plan pl;
  plan_from_desc(*desc, pl, PLAN_HOIST | PLAN_W2);
  if (!pl.hoist) return 0.0;
  const double evals = (desc->solver == FRL_SOLVER_DPM2 ? 2.0 : 1.0) * desc->steps * (double)pl.B;
  return 2.0 * (evals - (double)desc->n_states) * pl.C * pl.H;
}

// both_regions: a descriptor whose workspace holds the 32-chain AND the 16-chain geometry's tables (Plan.w2_pair) gets
// both written, so that frl_sampler_run may be called on it at ANY batch size; frl_sampler_fwd, which knows the batch
// it prepares for, writes only the region its own run reads.
static int sampler_prepare_impl(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                                void* workspace, size_t workspace_bytes, void* hip_stream, int plan_flags, bool both_regions = false,
                                int pick = W2_PICK_BY_BATCH) {
  int rc = validate_desc(desc);
  if (rc != FRL_OK) return rc;
  if (params == nullptr) { set_error("params is NULL"); return FRL_EINVAL; }
  Plan pl;
  plan_from_desc(*desc, pl, plan_flags, pick);
  if (both_regions && pl.w2_pair) {
    rc = sampler_prepare_impl(desc, params, sched, workspace, workspace_bytes, hip_stream, plan_flags, false, W2_PICK_32);
    if (rc != FRL_OK) return rc;
    return sampler_prepare_impl(desc, params, sched, workspace, workspace_bytes, hip_stream, plan_flags, false, W2_PICK_16);
  }
  rc = check_ws(pl, workspace, workspace_bytes);
  if (rc != FRL_OK) return rc;
  for (int l = 0; l <= pl.L; ++l)
    if (params->kernel[l] == nullptr || params->bias[l] == nullptr) { set_error("params.kernel/bias[%d] is NULL", l); return FRL_EINVAL; }
  if (desc->time_embed != FRL_TEMB_NONE) {
    for (int k = 0; k < 2; ++k)
      if (params->tmlp_kernel[k] == nullptr || params->tmlp_bias[k] == nullptr) { set_error("params.tmlp_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
    if (desc->time_embed == FRL_TEMB_FOURIER && params->time_kernel == nullptr) { set_error("params.time_kernel is NULL"); return FRL_EINVAL; }
  }
  FrlSchedule s;
  memset(&s, 0, sizeof(s));
  if (desc->kind != FRL_KIND_CNF) {
    if (sched == nullptr) { set_error("sched is NULL"); return FRL_EINVAL; }
    s = *sched;
    if (s.alphas == nullptr || s.alpha_hats == nullptr) { set_error("sched.alphas / alpha_hats is NULL"); return FRL_EINVAL; }
    if (desc->kind == FRL_KIND_DDPM && desc->solver == FRL_SOLVER_DDPM && s.postvars == nullptr) { set_error("sched.postvars is NULL"); return FRL_EINVAL; }
    if (desc->kind == FRL_KIND_CDDPM && (s.ts == nullptr || s.betas == nullptr)) { set_error("sched.ts / betas is NULL"); return FRL_EINVAL; }
    if (desc->solver == FRL_SOLVER_DPM2 && (s.mid_ts == nullptr || s.mid_alphas == nullptr || s.mid_sigmas == nullptr)) {
      set_error("sched.mid_ts / mid_alphas / mid_sigmas is NULL (dpm2 evaluates the schedule at the interval midpoints)"); return FRL_EINVAL; }
  }
  PArgs pa;
  pa.p = *params;
  pa.s = s;
  pa.ws = (float*)workspace;
  pa.plan = pl;
  pa.kind = desc->kind;
  pa.time_embed = desc->time_embed;
  rc = launch_prepare(pa, (hipStream_t)hip_stream);
  if (rc == FRL_OK && desc->layer_norm && desc->backbone == FRL_BACKBONE_MLP) {
    // LayerNorm scale | bias rows of every hidden layer, gathered into one [L][2][H] image the kernel stages in LDS
    float* ln = (float*)workspace + pl.off_ln;
    for (int l = 0; l < pl.L && rc == FRL_OK; ++l) {
      if (params->ln_scale[l] == nullptr || params->ln_bias[l] == nullptr) { set_error("params.ln_scale/ln_bias[%d] is NULL (desc.layer_norm = 1)", l); return FRL_EINVAL; }
      hipError_t e = hipMemcpyAsync(ln + (size_t)(2 * l) * pl.H, params->ln_scale[l], pl.H * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream);
      if (e == hipSuccess) e = hipMemcpyAsync(ln + (size_t)(2 * l + 1) * pl.H, params->ln_bias[l], pl.H * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream);
      if (e != hipSuccess) { set_error("LayerNorm rows: %s", hipGetErrorString(e)); rc = FRL_EHIP; }
    }
  }
  if (rc == FRL_OK && pl.bf16) rc = launch_prepare_bf16(*params, pl, (float*)workspace, (hipStream_t)hip_stream);
  return rc;
}

int frl_sampler_prepare(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                        void* workspace, size_t workspace_bytes, void* hip_stream) {
  return sampler_prepare_impl(desc, params, sched, workspace, workspace_bytes, hip_stream, PLAN_HOIST | PLAN_W2, true);
}

static int launch_fused_network(const KArgs& a, const CriticTab& ct, int act, hipStream_t stream) {
  switch (act) {
    case FRL_ACT_MISH: return launch_sampler_rs_fused_act<FRL_ACT_MISH>(a, ct, stream);
    case FRL_ACT_SILU: return launch_sampler_rs_fused_act<FRL_ACT_SILU>(a, ct, stream);
    case FRL_ACT_RELU: return launch_sampler_rs_fused_act<FRL_ACT_RELU>(a, ct, stream);
    case FRL_ACT_GELU_TANH: return launch_sampler_rs_fused_act<FRL_ACT_GELU_TANH>(a, ct, stream);
    case FRL_ACT_TANH: return launch_sampler_rs_fused_act<FRL_ACT_TANH>(a, ct, stream);
    default: return launch_sampler_rs_fused_act<FRL_ACT_ELU>(a, ct, stream);
  }
}

// fused: the critic table of frl_sampler_select_fwd (the sampler's launch then scores and selects, x0 optional), or NULL
static int sampler_run_impl(const FrlDesc* desc, const FrlParams* params, const FrlFwdBufs* bufs,
                            void* workspace, size_t workspace_bytes, void* hip_stream, const CriticTab* fused) {
  int rc = validate_desc(desc);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  Plan pl;
  plan_from_desc(*desc, pl, fused != nullptr ? PLAN_HOIST : (PLAN_HOIST | PLAN_W2));
  rc = check_ws(pl, workspace, workspace_bytes);
  if (rc != FRL_OK) return rc;
  if (pl.B == 0) {   // empty batch: no chains (the reference returns empty arrays), but the rng still advances
    if (desc->rng_impl != FRL_RNG_NONE && bufs->rng_key != nullptr)
      return launch_rng_chain(bufs->rng_key, desc->rng_impl, pl.mode, pl.T, (uint32_t*)((float*)workspace + pl.off_keys), bufs->rng_out, (hipStream_t)hip_stream);
    return FRL_OK;
  }
  if (bufs->xT == nullptr || (bufs->x0 == nullptr && fused == nullptr)) { set_error("bufs.xT / x0 is NULL"); return FRL_EINVAL; }
  if ((desc->cond_dim > 0) != (bufs->cond != nullptr)) { set_error("bufs.cond must be non-NULL exactly when cond_dim > 0"); return FRL_EINVAL; }
  const bool needs_noise = (desc->kind != FRL_KIND_CNF) && desc->solver == FRL_SOLVER_DDPM;
  const bool rng = desc->rng_impl != FRL_RNG_NONE;
  if (rng && bufs->rng_total_chains != 0 &&
      (bufs->rng_chain_offset < 0 || bufs->rng_chain_offset + pl.B > bufs->rng_total_chains)) {
    set_error("bufs.rng_chain_offset %lld + %lld chains exceeds rng_total_chains %lld", (long long)bufs->rng_chain_offset, pl.B, (long long)bufs->rng_total_chains); return FRL_EINVAL; }
  if (rng && desc->rng_impl == FRL_RNG_THREEFRY_LEGACY && bufs->rng_total_chains * desc->x_dim > 0xFFFFFFFFll) {
    set_error("legacy threefry layout: at most 2^32 - 1 elements per draw"); return FRL_EUNSUPPORTED; }
  if (rng && bufs->rng_key == nullptr) { set_error("desc.rng_impl = %d but bufs.rng_key is NULL", desc->rng_impl); return FRL_EINVAL; }
  if (needs_noise && !rng && bufs->noise == nullptr) { set_error("bufs.noise is NULL but the ddpm solver draws noise every step (or set desc.rng_impl and bufs.rng_key)"); return FRL_EINVAL; }
  if (desc->want_history && (bufs->xt_hist == nullptr || bufs->eps_hist == nullptr)) { set_error("want_history set but xt_hist / eps_hist is NULL"); return FRL_EINVAL; }
  for (int l = 1; l < pl.L; ++l)
    if (params->kernel[l] == nullptr || params->bias[l] == nullptr) { set_error("params.kernel/bias[%d] is NULL", l); return FRL_EINVAL; }
  if (desc->cond_embed)
    for (int k = 0; k < 2; ++k)
      if (params->cond_kernel[k] == nullptr || params->cond_bias[k] == nullptr) { set_error("params.cond_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
  ResArgs ra;
  if (desc->backbone != FRL_BACKBONE_MLP) {
    rc = fill_res_args(desc, params, ra);
    if (rc != FRL_OK) return rc;
  }

  const float* ws = (const float*)workspace;
  KArgs a;
  memset(&a, 0, sizeof(a));
  // an explicit noise tensor wins over in-kernel draws (the rng is still advanced as the reference's scan does)
  const bool draw = rng && needs_noise && bufs->noise == nullptr;
  a.xT = bufs->xT; a.cond = bufs->cond; a.noise = (needs_noise && !draw) ? bufs->noise : nullptr;
  if (rng) {
    // the scan's split sequence: one key per iteration that draws noise, and the rng sample() returns
    uint32_t* keys = (uint32_t*)((float*)workspace + pl.off_keys);
    rc = launch_rng_chain(bufs->rng_key, desc->rng_impl, pl.mode, pl.T, keys, bufs->rng_out, (hipStream_t)hip_stream);
    if (rc != FRL_OK) return rc;
    if (draw) {
      a.rng_keys = keys; a.rng_impl = desc->rng_impl;
      a.rng_off = bufs->rng_total_chains > 0 ? bufs->rng_chain_offset : 0;
      a.rng_total = bufs->rng_total_chains > 0 ? bufs->rng_total_chains : pl.B;
    }
  }
  if (desc->cond_embed) {
    // cond_embedding MLP once per state (ddpm.py:89-90), into the workspace; the sampler then reads those rows
    float* cfeat = (float*)workspace + pl.off_cfeat;
    rc = launch_cond_embed(desc->act, bufs->cond, desc->n_states, desc->cond_in_dim, params->cond_kernel[0], params->cond_bias[0],
                           params->cond_kernel[1], params->cond_bias[1], cfeat, (hipStream_t)hip_stream);
    if (rc != FRL_OK) return rc;
    a.cond = cfeat;
  }
  if (pl.hoist) {
    // cond rows of Dense_0, once per state: zs[state] = cond[state] @ kernel_0[A + D' :]  (the bias lives in tb)
    float* zs = (float*)workspace + pl.off_zs;
    rc = launch_state_bias(a.cond, desc->n_states, pl.C, pl.H, params->kernel[0] + (size_t)pl.c_row0 * pl.H, nullptr, zs,
                           (hipStream_t)hip_stream);
    if (rc != FRL_OK) return rc;
    a.state_bias = zs;
  }
  a.x0 = bufs->x0;
  a.xt_hist = desc->want_history ? bufs->xt_hist : nullptr;
  a.eps_hist = desc->want_history ? bufs->eps_hist : nullptr;
  for (int l = 1; l < pl.L; ++l) { a.wh[l - 1] = params->kernel[l]; a.bh[l - 1] = params->bias[l]; }
  a.ws_coef = ws + pl.off_coef; a.ws_tb = ws + pl.off_tb; a.ws_w0p = ws + pl.off_w0p;
  a.ws_wlp = ws + pl.off_wlp; a.ws_wlq = ws + pl.off_wlq; a.ws_blp = ws + pl.off_blp;
  a.ws_ln = (desc->layer_norm && desc->backbone == FRL_BACKBONE_MLP) ? ws + pl.off_ln : nullptr;
  a.ln_post_act = desc->layer_norm == FRL_LN_POST_ACT ? 1 + desc->act : 0;
  a.B = pl.B; a.particles = desc->particles; a.A = pl.A; a.C = pl.C; a.T = pl.T; a.L = pl.L; a.H = pl.H; a.TR = pl.TR;
  a.NE = pl.NE; a.AO = pl.AO;
  if (pl.hoist) a.C = 0;
  a.w0p_bytes = (int)(pl.w0p_rows * pl.H * sizeof(float));
  a.mode = pl.mode; a.clip = desc->clip ? 1 : 0; a.x_min = desc->x_min; a.x_max = desc->x_max;
  a.XR = pl.XR; a.AR = pl.AR; a.CK = pl.CK; a.merged = pl.merged; a.n_l0 = pl.n_l0; a.NS = pl.NS;
  if (pl.bf16) {
    a.ws_bf16 = ws + pl.off_bf16;
    a.bf_nxk = pl.bf_nxk; a.bf_cks = pl.bf_cks; a.bf_l0c = pl.bf_l0c; a.bf_ns = pl.bf_ns;
    return launch_sampler_bf16(*desc, a, (hipStream_t)hip_stream);
  }
  a.wl_raw = params->kernel[pl.L];
  if (fused != nullptr) return launch_fused_network(a, *fused, desc->act, (hipStream_t)hip_stream);
  if (desc->backbone != FRL_BACKBONE_MLP) return launch_res_network(a, ra, desc->act, desc->tile_policy, (hipStream_t)hip_stream);
  return launch_f32_network(a, desc->act, desc->tile_policy, (hipStream_t)hip_stream);
}

int frl_sampler_run(const FrlDesc* desc, const FrlParams* params, const FrlFwdBufs* bufs,
                    void* workspace, size_t workspace_bytes, void* hip_stream) {
  clear_kernels();
  return sampler_run_impl(desc, params, bufs, workspace, workspace_bytes, hip_stream, nullptr);
}

int frl_sampler_fwd(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                    const FrlFwdBufs* bufs, void* workspace, size_t workspace_bytes, void* hip_stream) {
  int rc = sampler_prepare_impl(desc, params, sched, workspace, workspace_bytes, hip_stream, PLAN_HOIST | PLAN_W2);   // this batch's region only
  if (rc != FRL_OK) return rc;
  return frl_sampler_run(desc, params, bufs, workspace, workspace_bytes, hip_stream);
}

size_t frl_static_workspace_limit(void) { return (size_t)FRL_STATIC_WS_BYTES; }

int frl_sampler_fwd_static(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched,
                           const FrlFwdBufs* bufs, void* hip_stream) {
  int rc = validate_desc(desc);
  if (rc != FRL_OK) return rc;
  Plan pl;
  plan_from_desc(*desc, pl, PLAN_HOIST | PLAN_W2);
  const size_t need = pl.total_floats * sizeof(float);
  if (need > (size_t)FRL_STATIC_WS_BYTES) {
    set_error("frl_sampler_fwd_static: this call needs %zu B of scratch, the static buffer holds %zu B (per-state rows of a cond_embedding / "
              "hoisted condition grow with n_states = %lld): use frl_sampler_fwd with a caller workspace", need, (size_t)FRL_STATIC_WS_BYTES,
              (long long)desc->n_states);
    return FRL_EWORKSPACE;
  }
  void* ws = nullptr;
  rc = static_scratch(&ws);
  if (rc != FRL_OK) return rc;
  // frl_sampler_prepare rewrites every table and packed image this descriptor reads (padding included), so nothing of the
  // previous call's contents is consumed; the per-state rows (cfeat, zs) and iteration keys are written by frl_sampler_run.
  return frl_sampler_fwd(desc, params, sched, bufs, ws, (size_t)FRL_STATIC_WS_BYTES, hip_stream);
}

static int validate_bwd(const FrlDesc* d) {
  int rc = validate_desc(d);
  if (rc != FRL_OK) return rc;
  Plan pl;
  plan_from_desc(*d, pl, 0);           // the VJP never hoists the condition rows: they are contracted per chain
  if (pl.mode > MODE_CNF_EULER) { set_error("VJP: solvers dpm2 / dpm++2m have no backward kernel"); return FRL_EUNSUPPORTED; }
  if (d->layer_norm) { set_error("VJP: LayerNorm backbones have no backward kernel"); return FRL_EUNSUPPORTED; }
  if (d->backbone != FRL_BACKBONE_MLP) { set_error("VJP: residual backbones have no backward kernel"); return FRL_EUNSUPPORTED; }
  if (d->dtype == FRL_DTYPE_BF16) {
    // bf16-MFMA VJP (csrc/sampler_bwd_bf16.hip): the narrow instantiation of the bf16 forward kernel
    if (pl.H != 256) { set_error("bf16 VJP: only the 256-wide build has a backward kernel"); return FRL_EUNSUPPORTED; }
    if (pl.A > 16 || pl.C > 32 || pl.A + pl.C > 32) { set_error("bf16 VJP: x_dim <= 16 and x_dim + cond_dim <= 32 required (got %d, %d)", pl.A, pl.C); return FRL_EUNSUPPORTED; }
    if (d->cond_embed) { set_error("bf16 VJP: cond_embedding backward not built"); return FRL_EUNSUPPORTED; }
    if (d->act != FRL_ACT_MISH && d->act != FRL_ACT_SILU && d->act != FRL_ACT_RELU) { set_error("bf16 VJP: activation %d has no backward kernel (mish / silu / relu)", d->act); return FRL_EUNSUPPORTED; }
    return FRL_OK;
  }
  if (pl.H != 256 || pl.TR != 32) { set_error("VJP: only the 256-wide (32x32x2) geometry has a backward kernel"); return FRL_EUNSUPPORTED; }
  if (pl.AR > 12 || pl.CK > 64) { set_error("VJP: x_dim <= 24 and <= 128 condition rows required (got %d, %d)", pl.A, pl.C); return FRL_EUNSUPPORTED; }
  return FRL_OK;      // every activation of the forward has its derivative in the f32 sweep
}

// x_dim > 8, more than 24 condition rows or a cond_embedding: the wide sweep build, which exists for the stash path only
static bool bwd_needs_stash(const FrlDesc* d, const Plan& pl) { return d->dtype == FRL_DTYPE_F32 && (pl.AR > 4 || pl.CK > 12 || d->cond_embed); }

size_t frl_bwd_workspace_bytes(const FrlDesc* desc) {
  if (validate_bwd(desc) != FRL_OK) return 0;
  if (desc->dtype == FRL_DTYPE_BF16) {
    BwdPlan16 bp;
    bwd16_plan_from_desc(*desc, bp);
    return bp.total_floats * sizeof(float);
  }
  BwdPlan bp;
  bwd_plan_from_desc(*desc, bp);
  return bp.total_floats * sizeof(float);
}

// The training forward of a SMALL batch (an actor update on a few hundred transitions) runs as the latency build with the
// stash kept (csrc/sampler_lat.hip, STASH): four waves per 32-chain tile instead of one, a quarter of the serial MFMA chain
// per step (B = 256, T = 5: 0.415 ms -> see DESIGN.md 5.4).  Same rule of thumb as the forward's tile policy, one notch more
// conservative because of the scattered stash stores: latency rounds (one tile per CU) at most twice the throughput rounds.
static bool lat_train_ok(const KArgs& a) {
  if (a.H != 256 || a.TR != 32 || a.ws_ln != nullptr || a.NE != a.T || !sampler_lat_supports(a)) return false;
  const long long n_wt = (a.B + 31) / 32, cus = device_cus();
  return (n_wt + cus - 1) / cus <= 2 * ((n_wt + 4 * cus - 1) / (4 * cus));
}
static int launch_lat_train(const BArgs& b, int act, const float* w0_cond_rows, long long n_states, const float* wl_raw, hipStream_t stream) {
  KArgs a = b.k;
  a.stash_h = b.hd; a.stash_a = b.ad; a.wl_raw = wl_raw; a.ln_post_act = 0;
  if (a.state_bias == nullptr && a.cond != nullptr && w0_cond_rows != nullptr && a.C > 24 && a.T >= 2) {
    // A wide condition (DACER's 128 cond_embedding features, dacer.py:141-162) does not depend on the step: its rows of
    // Dense_0 are contracted once per STATE (frl_state_bias_kernel) instead of 64 k-steps in every one of the T steps, and
    // enter the kernel as a per-state accumulator init.  The rows wait in the head of the dL/dz stash region, which nothing
    // reads or writes before the backward sweep (n_states * H <= its n_ts * L * 8192 floats).  The backward is unchanged:
    // the dW kernel contracts the condition per chain from BArgs.dw_cond, the sweep's dL/dcfeat from its own packed image.
    int rc = launch_state_bias(a.cond, n_states, a.C, a.H, w0_cond_rows, nullptr, b.gd, stream);
    if (rc != FRL_OK) return rc;
    a.state_bias = b.gd; a.C = 0; a.CK = 0;
  }
  switch (act) {
    case FRL_ACT_MISH: return launch_sampler_lat_act<FRL_ACT_MISH>(a, 0, stream);
    case FRL_ACT_SILU: return launch_sampler_lat_act<FRL_ACT_SILU>(a, 0, stream);
    case FRL_ACT_RELU: return launch_sampler_lat_act<FRL_ACT_RELU>(a, 0, stream);
    case FRL_ACT_GELU_TANH: return launch_sampler_lat_act<FRL_ACT_GELU_TANH>(a, 0, stream);
    case FRL_ACT_TANH: return launch_sampler_lat_act<FRL_ACT_TANH>(a, 0, stream);
    default: return launch_sampler_lat_act<FRL_ACT_ELU>(a, 0, stream);
  }
}

// Kernel arguments shared by the training forward and the backward: forward tables + stash / image pointers of the workspace.
static void fill_bwd_args(const FrlDesc* desc, const FrlParams* params, const Plan& pl, const BwdPlan& bp, const BwdPlan16& bp16,
                          bool bf16, float* ws, const float* cond, BArgs& b) {
  memset(&b, 0, sizeof(b));
  KArgs& a = b.k;
  a.cond = cond;
  for (int l = 1; l < desc->n_hidden; ++l) { a.wh[l - 1] = params->kernel[l]; a.bh[l - 1] = params->bias[l]; b.wt[l - 1] = bf16 ? nullptr : ws + bp.off_wt[l - 1]; }
  a.ws_coef = ws + pl.off_coef; a.ws_tb = ws + pl.off_tb; a.ws_w0p = ws + pl.off_w0p;
  a.ws_wlp = ws + pl.off_wlp; a.ws_wlq = ws + pl.off_wlq; a.ws_blp = ws + pl.off_blp;
  a.B = pl.B; a.particles = desc->particles; a.A = pl.A; a.C = pl.C; a.T = pl.T; a.L = pl.L; a.H = pl.H; a.TR = pl.TR;
  a.NE = pl.NE; a.AO = pl.AO;
  a.w0p_bytes = (int)(pl.w0p_rows * pl.H * sizeof(float));
  a.mode = pl.mode; a.clip = desc->clip ? 1 : 0; a.x_min = desc->x_min; a.x_max = desc->x_max;
  a.XR = pl.XR; a.AR = pl.AR; a.CK = pl.CK; a.merged = pl.merged; a.n_l0 = pl.n_l0; a.NS = pl.NS;
  if (bf16) {
    a.ws_bf16 = ws + pl.off_bf16;
    a.bf_nxk = pl.bf_nxk; a.bf_cks = pl.bf_cks; a.bf_l0c = pl.bf_l0c; a.bf_ns = pl.bf_ns;
    b.hd = ws + bp16.off_hd; b.ad = ws + bp16.off_ad; b.gd = ws + bp16.off_gd; b.ge = ws + bp16.off_ge;
    b.x0d = ws + bp16.off_x0d;
  } else {
    b.ws_wlt = ws + bp.off_wlt; b.ws_w0xq = ws + bp.off_w0xq;
    b.hd = ws + bp.off_hd; b.ad = ws + bp.off_ad; b.gd = ws + bp.off_gd; b.ge = ws + bp.off_ge;
  }
  b.Dp = pl.Dp;
  b.n_ts = bf16 ? bp16.n_ts : bp.n_ts;
  b.part = ws + (bf16 ? bp16.off_part : bp.off_part);
  b.n_part = bf16 ? bp16.n_part : bp.n_part;
  b.k0_rows = bf16 ? 32 : bp.k0_rows;
  b.ge_stride = bf16 ? 256 : bp.ge_stride;
  b.dw_cond = a.cond; b.dw_C = pl.C;
  if (!bf16 && desc->cond_embed) {
    // the predictor sees cfeat = cond_embedding(obs) (n_states, 128), computed once per state into the workspace
    a.cond = ws + pl.off_cfeat;
    b.dw_cond = a.cond;
    b.ws_w0cq = ws + bp.off_w0cq;
    b.gcf = ws + bp.off_gcf;
  }
}

int frl_sampler_fwd_train(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched, const FrlFwdBufs* bufs,
                          void* workspace, size_t workspace_bytes, void* hip_stream) {
  clear_kernels();
  int rc = validate_bwd(desc);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  if (!desc->want_history) { set_error("training forward: desc.want_history must be 1 (the backward reads xt_hist / eps_hist)"); return FRL_EINVAL; }
  BwdPlan bp;
  BwdPlan16 bp16;
  memset(&bp16, 0, sizeof(bp16));
  const bool bf16 = desc->dtype == FRL_DTYPE_BF16;
  if (bf16) { bwd16_plan_from_desc(*desc, bp16); bp.f = bp16.f; bp.total_floats = bp16.total_floats; }
  else bwd_plan_from_desc(*desc, bp);
  const Plan& pl = bp.f;
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < bp.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B (frl_bwd_workspace_bytes)", workspace_bytes, bp.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  hipStream_t stream = (hipStream_t)hip_stream;
  float* ws = (float*)workspace;
  const bool rng = desc->rng_impl != FRL_RNG_NONE;
  const bool needs_noise = (desc->kind != FRL_KIND_CNF) && desc->solver == FRL_SOLVER_DDPM;
  if (pl.B == 0) {
    if (rng && bufs->rng_key != nullptr)
      return launch_rng_chain(bufs->rng_key, desc->rng_impl, pl.mode, pl.T, (uint32_t*)(ws + pl.off_keys), bufs->rng_out, stream);
    return FRL_OK;
  }
  if (bufs->xT == nullptr || bufs->x0 == nullptr || bufs->xt_hist == nullptr || bufs->eps_hist == nullptr) { set_error("bufs.xT / x0 / xt_hist / eps_hist is NULL"); return FRL_EINVAL; }
  if ((desc->cond_dim > 0) != (bufs->cond != nullptr)) { set_error("bufs.cond must be non-NULL exactly when cond_dim > 0"); return FRL_EINVAL; }
  if (rng && bufs->rng_key == nullptr) { set_error("desc.rng_impl = %d but bufs.rng_key is NULL", desc->rng_impl); return FRL_EINVAL; }
  if (rng && bufs->rng_total_chains != 0 && (bufs->rng_chain_offset < 0 || bufs->rng_chain_offset + pl.B > bufs->rng_total_chains)) {
    set_error("bufs.rng_chain_offset / rng_total_chains do not cover this call's chains"); return FRL_EINVAL; }
  if (needs_noise && !rng && bufs->noise == nullptr) { set_error("bufs.noise is NULL but the ddpm solver draws noise every step (or set desc.rng_impl and bufs.rng_key)"); return FRL_EINVAL; }
  rc = sampler_prepare_impl(desc, params, sched, workspace, pl.total_floats * sizeof(float), hip_stream, 0);
  if (rc != FRL_OK) return rc;
  if (desc->cond_embed) {
    for (int k = 0; k < 2; ++k)
      if (params->cond_kernel[k] == nullptr || params->cond_bias[k] == nullptr) { set_error("params.cond_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
    rc = launch_cond_embed(desc->act, bufs->cond, desc->n_states, desc->cond_in_dim, params->cond_kernel[0], params->cond_bias[0],
                           params->cond_kernel[1], params->cond_bias[1], ws + pl.off_cfeat, stream);
    if (rc != FRL_OK) return rc;
  }
  BArgs b;
  fill_bwd_args(desc, params, pl, bp, bp16, bf16, ws, bufs->cond, b);
  KArgs& a = b.k;
  a.xT = bufs->xT; a.x0 = bufs->x0; a.xt_hist = bufs->xt_hist; a.eps_hist = bufs->eps_hist;
  const bool draw = rng && needs_noise && bufs->noise == nullptr;
  a.noise = (needs_noise && !draw) ? bufs->noise : nullptr;
  if (rng) {
    uint32_t* keys = (uint32_t*)(ws + pl.off_keys);
    rc = launch_rng_chain(bufs->rng_key, desc->rng_impl, pl.mode, pl.T, keys, bufs->rng_out, stream);
    if (rc != FRL_OK) return rc;
    if (draw) {
      a.rng_keys = keys; a.rng_impl = desc->rng_impl;
      a.rng_off = bufs->rng_total_chains > 0 ? bufs->rng_chain_offset : 0;
      a.rng_total = bufs->rng_total_chains > 0 ? bufs->rng_total_chains : pl.B;
    }
  }
  if (!bf16 && lat_train_ok(a)) return launch_lat_train(b, desc->act, pl.C > 0 ? params->kernel[0] + (size_t)pl.c_row0 * pl.H : nullptr, desc->n_states, params->kernel[pl.L], stream);
  return bf16 ? launch_sampler_bwd_bf16(*desc, b, 1, stream) : launch_sampler_bwd(*desc, b, 1, stream);
}

int frl_sampler_bwd(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched, const FrlBwdBufs* bufs,
                    void* workspace, size_t workspace_bytes, void* hip_stream) {
  clear_kernels();
  int rc = validate_bwd(desc);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  BwdPlan bp;
  BwdPlan16 bp16;
  memset(&bp16, 0, sizeof(bp16));
  const bool bf16 = desc->dtype == FRL_DTYPE_BF16;
  if (bf16) { bwd16_plan_from_desc(*desc, bp16); bp.f = bp16.f; bp.total_floats = bp16.total_floats; }
  else bwd_plan_from_desc(*desc, bp);
  const Plan& pl = bp.f;
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < bp.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B", workspace_bytes, bp.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  if (bf16 && bp16.off_img_b != pl.off_bf16 + (size_t)pl.bf_ns * (BF_CHUNK_BYTES / 4)) { set_error("bf16 VJP: stream images not contiguous"); return FRL_EINVAL; }
  if (pl.B == 0) return FRL_OK;
  if (bufs->xt_hist == nullptr || bufs->g_x0 == nullptr || bufs->grad_tb == nullptr) { set_error("bufs.xt_hist / g_x0 / grad_tb is NULL"); return FRL_EINVAL; }
  if ((desc->cond_dim > 0) != (bufs->cond != nullptr)) { set_error("bufs.cond must be non-NULL exactly when cond_dim > 0"); return FRL_EINVAL; }
  const bool stash = bufs->have_stash != 0;
  if (stash && bufs->eps_hist == nullptr) { set_error("have_stash = 1 needs bufs.eps_hist (history[1] of frl_sampler_fwd_train)"); return FRL_EINVAL; }
  if (!stash && bwd_needs_stash(desc, pl)) {
    set_error("VJP: x_dim %d / %d condition rows%s: run the forward with frl_sampler_fwd_train and pass have_stash = 1 (the recompute path covers x_dim <= 8, <= 24 condition rows, no cond_embedding)",
              pl.A, pl.C, desc->cond_embed ? " / cond_embedding" : "");
    return FRL_EUNSUPPORTED;
  }
  if (desc->cond_embed)
    for (int k = 0; k < 2; ++k) {
      if (params->cond_kernel[k] == nullptr || params->cond_bias[k] == nullptr) { set_error("params.cond_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
      if (bufs->grad_cond_kernel[k] == nullptr || bufs->grad_cond_bias[k] == nullptr) { set_error("bufs.grad_cond_kernel/bias[%d] is NULL (desc.cond_embed = 1)", k); return FRL_EINVAL; }
    }
  for (int l = 0; l <= desc->n_hidden; ++l) {
    if (bufs->grad_kernel[l] == nullptr) { set_error("bufs.grad_kernel[%d] is NULL", l); return FRL_EINVAL; }
    if (l > 0 && bufs->grad_bias[l] == nullptr) { set_error("bufs.grad_bias[%d] is NULL", l); return FRL_EINVAL; }
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  float* ws = (float*)workspace;
  // forward tables (coefficients, tb, packed edge layers) — already there after frl_sampler_fwd_train — then the transposed images
  if (!stash) {
    rc = sampler_prepare_impl(desc, params, sched, workspace, pl.total_floats * sizeof(float), hip_stream, 0);
    if (rc != FRL_OK) return rc;
  }
  rc = bf16 ? launch_prepare_bwd_bf16(*params, bp16, ws, stream) : launch_prepare_bwd(*params, bp, ws, stream);
  if (rc != FRL_OK) return rc;
  // every gradient leaf is WRITTEN once by frl_bwd_reduce_kernel (fixed-order sum of per-workgroup partial tiles in the
  // workspace): no float atomics, no zero-fill, bitwise reproducible
  BArgs b;
  fill_bwd_args(desc, params, pl, bp, bp16, bf16, ws, bufs->cond, b);
  b.xt_hist = bufs->xt_hist; b.g_x0 = bufs->g_x0; b.g_xT = bufs->g_xT; b.eps_hist = bufs->eps_hist;
  for (int l = 0; l <= pl.L; ++l) { b.grad_kernel[l] = bufs->grad_kernel[l]; b.grad_bias[l] = bufs->grad_bias[l]; }
  b.grad_tb = bufs->grad_tb;
  rc = bf16 ? launch_sampler_bwd_bf16(*desc, b, stash ? 2 : 0, stream) : launch_sampler_bwd(*desc, b, stash ? 2 : 0, stream);
  if (rc != FRL_OK) return rc;
  if (desc->cond_embed) {
    // dL/dcfeat (per chain, summed over steps by the sweep) -> the two embedding layers (bufs.cond = the raw observations)
    rc = launch_cond_embed_bwd(desc->act, bufs->cond, desc->n_states, desc->particles, desc->cond_in_dim, params->cond_kernel[0],
                               params->cond_bias[0], params->cond_kernel[1], params->cond_bias[1], b.gcf, COND_EMBED_DIM, ws + bp.off_ceb,
                               bufs->grad_cond_kernel[0], bufs->grad_cond_bias[0], bufs->grad_cond_kernel[1], bufs->grad_cond_bias[1], stream);
    if (rc != FRL_OK) return rc;
  }
  const bool native_time = (desc->time_embed == FRL_TEMB_NONE) ? (bufs->grad_bias[0] != nullptr) : (bufs->grad_tmlp_kernel[0] != nullptr);
  if (!native_time) return FRL_OK;
  if (desc->time_embed != FRL_TEMB_NONE) {
    if (bufs->grad_tmlp_kernel[1] == nullptr || bufs->grad_tmlp_bias[0] == nullptr || bufs->grad_tmlp_bias[1] == nullptr || bufs->grad_bias[0] == nullptr) {
      set_error("bufs.grad_tmlp_kernel / grad_tmlp_bias / grad_bias[0]: all or none"); return FRL_EINVAL; }
    if (desc->time_embed == FRL_TEMB_FOURIER && bufs->grad_time_kernel == nullptr) { set_error("bufs.grad_time_kernel is NULL"); return FRL_EINVAL; }
  }
  TBArgs tb;
  memset(&tb, 0, sizeof(tb));
  tb.p = *params;
  if (sched != nullptr) tb.s = *sched;
  tb.grad_tb = bufs->grad_tb;
  tb.scratch = (double*)(ws + (bf16 ? bp16.off_tbs : bp.off_tbs));
  for (int k = 0; k < 2; ++k) { tb.g_tmlp_kernel[k] = bufs->grad_tmlp_kernel[k]; tb.g_tmlp_bias[k] = bufs->grad_tmlp_bias[k]; }
  tb.g_time_kernel = bufs->grad_time_kernel;
  tb.g_bias0 = bufs->grad_bias[0]; tb.g_kernel0 = bufs->grad_kernel[0];
  tb.kind = desc->kind; tb.time_embed = desc->time_embed; tb.T = pl.T; tb.D = pl.D; tb.H = pl.H; tb.A = pl.A;
  return launch_time_bwd(tb, stream);
}

// ---- critic scoring + best-of-N selection ------------------------------------------------------
static int critic_plan(const FrlCriticDesc* c, FrlDesc& d, Plan& pl, bool& hoist, bool force_hoist = false) {
  if (c == nullptr) { set_error("critic desc is NULL"); return FRL_EINVAL; }
  if (c->abi_version != FRL_ABI_VERSION) { set_error("critic desc.abi_version %d != library %d", c->abi_version, FRL_ABI_VERSION); return FRL_EINVAL; }
  if (c->ensemble < 1 || c->ensemble > 16) { set_error("ensemble %d outside [1, 16]", c->ensemble); return FRL_EINVAL; }
  if (c->act_dim < 1 || c->obs_dim < 0) { set_error("act_dim %d / obs_dim %d invalid", c->act_dim, c->obs_dim); return FRL_EINVAL; }
  // The observation rows of Dense_0 are the same for every candidate of a state: with several candidates per state (or
  // an observation wider than the kernel's condition operand budget) they are hoisted into a per-state accumulator init.
  hoist = c->obs_dim > 0 && (c->particles > 1 || c->obs_dim > 32 || force_hoist);
  memset(&d, 0, sizeof(d));
  d.abi_version = FRL_ABI_VERSION;
  d.kind = FRL_KIND_CNF; d.solver = FRL_SOLVER_EULER; d.steps = 1;      // carrier only: the mode is overridden below
  d.n_states = c->n_states; d.particles = c->particles;
  d.x_dim = c->act_dim; d.cond_dim = hoist ? 0 : c->obs_dim;
  d.time_embed = FRL_TEMB_NONE; d.time_dim = 0;
  d.n_hidden = c->n_hidden;
  for (int l = 0; l < FRL_MAX_HIDDEN; ++l) d.hidden[l] = (l < c->n_hidden) ? c->hidden[l] : 0;
  d.act = c->act; d.dtype = FRL_DTYPE_F32;
  int rc = validate_desc(&d);
  if (rc != FRL_OK) return rc;
  // ScalarCritic: x = concat([obs, action]) (critic.py:185-188) -> backbone MLP -> Dense(1): no time rows, the action
  // rows FOLLOW the observation rows, one output
  pl.A = d.x_dim; pl.C = d.cond_dim; pl.T = 1; pl.L = d.n_hidden; pl.D = 0; pl.Dp = 0;
  pl.AO = 1; pl.x_row0 = c->obs_dim; pl.c_row0 = 0;
  pl.H = d.hidden[0];
  plan_layout(d, pl);
  pl.mode = MODE_EVAL;
  return FRL_OK;
}

static size_t critic_ws_floats(const FrlCriticDesc* desc, const Plan& pl, bool hoist) {
  const size_t E = (size_t)desc->ensemble, B = (size_t)pl.B;
  size_t n = E * pl.total_floats + ((E * B + 63) & ~(size_t)63);
  if (hoist) n += (E * (size_t)desc->n_states * pl.H + 63) & ~(size_t)63;
  return n;
}

size_t frl_critic_workspace_bytes(const FrlCriticDesc* desc) {
  FrlDesc d; Plan pl; bool hoist;
  if (critic_plan(desc, d, pl, hoist) != FRL_OK) return 0;
  return critic_ws_floats(desc, pl, hoist) * sizeof(float);
}

int frl_critic_select(const FrlCriticDesc* desc, const FrlCriticParams* params, const FrlSelectBufs* bufs,
                      void* workspace, size_t workspace_bytes, void* hip_stream) {
  clear_kernels();
  FrlDesc d; Plan pl; bool hoist;
  int rc = critic_plan(desc, d, pl, hoist);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  const size_t need = critic_ws_floats(desc, pl, hoist) * sizeof(float);
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < need) { set_error("workspace %zu B < required %zu B", workspace_bytes, need); return FRL_EWORKSPACE; }
  if (pl.B == 0) return FRL_OK;
  if (bufs->actions == nullptr) { set_error("bufs.actions is NULL"); return FRL_EINVAL; }
  if ((desc->obs_dim > 0) != (bufs->obs != nullptr)) { set_error("bufs.obs must be non-NULL exactly when obs_dim > 0"); return FRL_EINVAL; }
  if (bufs->q == nullptr && bufs->best_idx == nullptr && bufs->best_action == nullptr) { set_error("no output requested (q / best_idx / best_action all NULL)"); return FRL_EINVAL; }
  for (int l = 0; l <= desc->n_hidden; ++l)
    if (params->kernel[l] == nullptr || params->bias[l] == nullptr) { set_error("params.kernel/bias[%d] is NULL", l); return FRL_EINVAL; }
  hipStream_t stream = (hipStream_t)hip_stream;
  const int E = desc->ensemble, L = pl.L, H = pl.H, K0 = desc->obs_dim + pl.A;
  float* ws = (float*)workspace;
  float* qm = ws + (size_t)E * pl.total_floats;                      // (E, B) member scores
  float* zs = qm + (((size_t)E * pl.B + 63) & ~(size_t)63);          // (E, n_states, H) hoisted Dense_0 slices
  for (int e = 0; e < E; ++e) {
    // member e of every stacked leaf (Ensemblize: nn.vmap with variable_axes params = 0, critic.py:154-172)
    PArgs pa;
    memset(&pa, 0, sizeof(pa));
    for (int l = 0; l <= L; ++l) {
      const size_t rows = (l == 0) ? (size_t)K0 : (size_t)H, cols = (l == L) ? 1 : (size_t)H;
      pa.p.kernel[l] = params->kernel[l] + (size_t)e * rows * cols;
      pa.p.bias[l] = params->bias[l] + (size_t)e * cols;
    }
    float* wse = ws + (size_t)e * pl.total_floats;
    pa.ws = wse; pa.plan = pl; pa.kind = FRL_KIND_CNF; pa.time_embed = FRL_TEMB_NONE;
    rc = launch_prepare(pa, stream);
    if (rc != FRL_OK) return rc;
    float* zse = zs + (size_t)e * desc->n_states * H;
    if (hoist) {
      rc = launch_state_bias(bufs->obs, desc->n_states, desc->obs_dim, H, pa.p.kernel[0], nullptr, zse, stream);   // bias_0 is in tb
      if (rc != FRL_OK) return rc;
    }
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.xT = bufs->actions; a.cond = hoist ? nullptr : bufs->obs; a.x0 = qm + (size_t)e * pl.B;
    a.state_bias = hoist ? zse : nullptr;
    for (int l = 1; l < L; ++l) { a.wh[l - 1] = pa.p.kernel[l]; a.bh[l - 1] = pa.p.bias[l]; }
    a.ws_coef = wse + pl.off_coef; a.ws_tb = wse + pl.off_tb; a.ws_w0p = wse + pl.off_w0p;
    a.ws_wlp = wse + pl.off_wlp; a.ws_wlq = wse + pl.off_wlq; a.ws_blp = wse + pl.off_blp;
    a.B = pl.B; a.particles = desc->particles; a.A = pl.A; a.C = pl.C; a.T = 1; a.L = L; a.H = H; a.TR = pl.TR;
    a.NE = 1; a.AO = 1;
    a.w0p_bytes = (int)(pl.w0p_rows * pl.H * sizeof(float));
    a.mode = MODE_EVAL;
    a.XR = pl.XR; a.AR = pl.AR; a.CK = pl.CK; a.merged = pl.merged; a.n_l0 = pl.n_l0; a.NS = pl.NS;
    a.wl_raw = pa.p.kernel[L];
    rc = launch_f32_network(a, desc->act, FRL_TILE_AUTO, stream);
    if (rc != FRL_OK) return rc;
  }
  return launch_critic_select(qm, E, desc->n_states, desc->particles, pl.A, bufs->actions, bufs->gumbel, bufs->inv_temperature,
                              bufs->q, bufs->best_idx, bufs->best_action, stream);
}

// ---- sampler + critic scoring + selection in one launch ----------------------------------------------------------------
struct SelectPlan {
  Plan pl, cpl;             // sampler (32-chain geometry, condition rows hoisted where the plan says) / one critic member
  FrlDesc cd;
  bool obs;
  size_t off_members, off_zs, off_tab, total_floats;
};

static int select_plan(const FrlDesc* desc, const FrlCriticDesc* cdesc, SelectPlan& sp) {
  int rc = validate_desc(desc);
  if (rc != FRL_OK) return rc;
  if (cdesc == nullptr) { set_error("critic desc is NULL"); return FRL_EINVAL; }
  if (desc->dtype != FRL_DTYPE_F32 || desc->backbone != FRL_BACKBONE_MLP || desc->layer_norm || desc->rng_impl != FRL_RNG_NONE) {
    set_error("fused sample+select: f32 MLP sampler without LayerNorm and with caller-supplied noise only"); return FRL_EUNSUPPORTED; }
  if (desc->n_hidden < 2) { set_error("fused sample+select: the sampler needs >= 2 hidden layers"); return FRL_EUNSUPPORTED; }
  plan_from_desc(*desc, sp.pl, PLAN_HOIST);
  const Plan& pl = sp.pl;
  if (pl.H != 256 || pl.TR != 32) { set_error("fused sample+select: 256-wide sampler only (got %d)", pl.H); return FRL_EUNSUPPORTED; }
  if (!((pl.AR <= 4 && pl.CK <= 12) || (pl.AR <= 12 && pl.CK == 0))) {
    set_error("fused sample+select: x_dim %d with %d in-kernel condition rows has no build (x_dim <= 8 with cond_dim <= 24, or a hoisted condition: particles >= 4 and cond_dim >= 32)", pl.A, pl.C);
    return FRL_EUNSUPPORTED; }
  if (cdesc->n_states != desc->n_states || cdesc->particles != desc->particles || cdesc->act_dim != desc->x_dim) {
    set_error("fused sample+select: critic n_states / particles / act_dim (%lld, %d, %d) differ from the sampler's (%lld, %d, %d)",
              (long long)cdesc->n_states, cdesc->particles, cdesc->act_dim, (long long)desc->n_states, desc->particles, desc->x_dim); return FRL_EINVAL; }
  const int N = desc->particles;
  if (!((N <= 32 && (N & (N - 1)) == 0) || N % 32 == 0)) { set_error("fused sample+select: particles %d (a power of two <= 32 or a multiple of 32)", N); return FRL_EUNSUPPORTED; }
  if (cdesc->ensemble < 1 || cdesc->ensemble > FUSE_MAX_E) { set_error("fused sample+select: ensemble %d outside [1, %d]", cdesc->ensemble, FUSE_MAX_E); return FRL_EUNSUPPORTED; }
  if (cdesc->n_hidden < 2 || cdesc->n_hidden > FUSE_MAX_L) { set_error("fused sample+select: critic hidden layers %d outside [2, %d]", cdesc->n_hidden, FUSE_MAX_L); return FRL_EUNSUPPORTED; }
  for (int l = 0; l < cdesc->n_hidden; ++l)
    if (cdesc->hidden[l] != 256) { set_error("fused sample+select: critic hidden widths must all be 256 (hidden[%d] = %d)", l, cdesc->hidden[l]); return FRL_EUNSUPPORTED; }
  if (cdesc->act != desc->act && cdesc->act != FRL_ACT_RELU) { set_error("fused sample+select: critic activation %d is neither the sampler's (%d) nor relu", cdesc->act, desc->act); return FRL_EUNSUPPORTED; }
  bool hoist;
  rc = critic_plan(cdesc, sp.cd, sp.cpl, hoist, true);
  if (rc != FRL_OK) return rc;
  if (sp.cpl.TR != 32 || sp.cpl.XR != pl.XR || sp.cpl.AR != pl.AR || sp.cpl.C != 0) { set_error("fused sample+select: critic plan does not match the sampler's geometry"); return FRL_EINVAL; }
  sp.obs = hoist;
  const size_t E = (size_t)cdesc->ensemble;
  sp.off_members = (pl.total_floats + 63) & ~(size_t)63;
  sp.off_zs = sp.off_members + E * ((sp.cpl.total_floats + 63) & ~(size_t)63);
  sp.off_tab = sp.off_zs + (sp.obs ? ((E * (size_t)desc->n_states * 256 + 63) & ~(size_t)63) : 0);
  sp.total_floats = sp.off_tab + ((sizeof(CriticPtrs) / sizeof(float) + 63) & ~(size_t)63);
  return FRL_OK;
}

size_t frl_sampler_select_workspace_bytes(const FrlDesc* desc, const FrlCriticDesc* cdesc) {
  SelectPlan sp;
  if (select_plan(desc, cdesc, sp) != FRL_OK) return 0;
  return sp.total_floats * sizeof(float);
}

int frl_sampler_select_fwd(const FrlDesc* desc, const FrlParams* params, const FrlSchedule* sched, const FrlFwdBufs* fwd,
                           const FrlCriticDesc* cdesc, const FrlCriticParams* cparams, const FrlSelectBufs* sel,
                           void* workspace, size_t workspace_bytes, void* hip_stream) {
  clear_kernels();
  SelectPlan sp;
  int rc = select_plan(desc, cdesc, sp);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || fwd == nullptr || cparams == nullptr || sel == nullptr) { set_error("params / fwd / cparams / sel is NULL"); return FRL_EINVAL; }
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < sp.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B (frl_sampler_select_workspace_bytes)", workspace_bytes, sp.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  if (sp.pl.B == 0) return FRL_OK;
  if (desc->want_history) { set_error("fused sample+select: want_history is not carried (use frl_sampler_fwd + frl_critic_select)"); return FRL_EUNSUPPORTED; }
  if ((cdesc->obs_dim > 0) != (sel->obs != nullptr)) { set_error("sel.obs must be non-NULL exactly when obs_dim > 0"); return FRL_EINVAL; }
  if (sel->best_action == nullptr && sel->best_idx == nullptr && sel->q == nullptr) { set_error("no output requested (q / best_idx / best_action all NULL)"); return FRL_EINVAL; }
  for (int l = 0; l <= cdesc->n_hidden; ++l)
    if (cparams->kernel[l] == nullptr || cparams->bias[l] == nullptr) { set_error("cparams.kernel/bias[%d] is NULL", l); return FRL_EINVAL; }
  hipStream_t stream = (hipStream_t)hip_stream;
  float* ws = (float*)workspace;
  const Plan& cpl = sp.cpl;
  const int E = cdesc->ensemble, L = cpl.L, H = 256, K0 = cdesc->obs_dim + cpl.A;
  const size_t mstride = (cpl.total_floats + 63) & ~(size_t)63;
  CriticPtrs cp;
  memset(&cp, 0, sizeof(cp));
  for (int e = 0; e < E; ++e) {
    // member e of every stacked leaf (Ensemblize: nn.vmap with variable_axes params = 0, critic.py:154-172)
    PArgs pa;
    memset(&pa, 0, sizeof(pa));
    for (int l = 0; l <= L; ++l) {
      const size_t rows = (l == 0) ? (size_t)K0 : (size_t)H, cols = (l == L) ? 1 : (size_t)H;
      pa.p.kernel[l] = cparams->kernel[l] + (size_t)e * rows * cols;
      pa.p.bias[l] = cparams->bias[l] + (size_t)e * cols;
    }
    float* wse = ws + sp.off_members + (size_t)e * mstride;
    pa.ws = wse; pa.plan = cpl; pa.kind = FRL_KIND_CNF; pa.time_embed = FRL_TEMB_NONE;
    rc = launch_prepare(pa, stream);
    if (rc != FRL_OK) return rc;
    const float** m = cp.p + e * CT_STRIDE;
    m[CT_W0P] = wse + cpl.off_w0p; m[CT_WLQ] = wse + cpl.off_wlq; m[CT_TB] = wse + cpl.off_tb; m[CT_BLP] = wse + cpl.off_blp;
    if (sp.obs) {
      float* zse = ws + sp.off_zs + (size_t)e * desc->n_states * H;
      rc = launch_state_bias(sel->obs, desc->n_states, cdesc->obs_dim, H, pa.p.kernel[0], nullptr, zse, stream);   // bias_0 is in tb
      if (rc != FRL_OK) return rc;
      m[CT_ZS] = zse;
    }
    for (int l = 1; l < L; ++l) { m[CT_WH + l - 1] = pa.p.kernel[l]; m[CT_BH + l - 1] = pa.p.bias[l]; }
  }
  const float** tab = (const float**)(ws + sp.off_tab);
  rc = launch_critic_tab(cp, tab, stream);
  if (rc != FRL_OK) return rc;
  CriticTab ct;
  memset(&ct, 0, sizeof(ct));
  ct.E = E; ct.L = L; ct.relu = (cdesc->act == FRL_ACT_RELU && desc->act != FRL_ACT_RELU) ? 1 : 0;
  const int N = desc->particles;
  ct.seg = N <= 32 ? N : 32; ct.group = N <= 32 ? 1 : N / 32;
  ct.w0p_bytes = (int)(cpl.w0p_rows * cpl.H * sizeof(float));
  ct.ptrs = tab;
  if (sel->gumbel != nullptr) { ct.gumbel = sel->gumbel; ct.inv_temperature = sel->inv_temperature; }
  ct.q = sel->q; ct.best_idx = sel->best_idx; ct.best_action = sel->best_action;
  rc = sampler_prepare_impl(desc, params, sched, workspace, sp.pl.total_floats * sizeof(float), hip_stream, PLAN_HOIST);
  if (rc != FRL_OK) return rc;
  return sampler_run_impl(desc, params, fwd, workspace, sp.pl.total_floats * sizeof(float), hip_stream, &ct);
}

int frl_random_split(const uint32_t* key, int32_t impl, int32_t num, uint32_t* out, void* hip_stream) {
  return launch_random_split(key, impl, num, out, (hipStream_t)hip_stream);
}
int frl_random_normal(const uint32_t* key, int32_t impl, int64_t n, float* out, void* hip_stream) {
  return launch_random_fill(key, impl, n, out, 1, (hipStream_t)hip_stream);
}
int frl_random_bits(const uint32_t* key, int32_t impl, int64_t n, uint32_t* out, void* hip_stream) {
  return launch_random_fill(key, impl, n, out, 0, (hipStream_t)hip_stream);
}

int frl_random_gumbel(const uint32_t* key, int32_t impl, int64_t n, float* out, void* hip_stream) {
  return launch_random_fill(key, impl, n, out, 2, (hipStream_t)hip_stream);
}

int frl_random_uniform(const uint32_t* key, int32_t impl, int64_t n, float minval, float maxval, float* out, void* hip_stream) {
  return launch_random_uniform(key, impl, n, minval, maxval, out, (hipStream_t)hip_stream);
}
int frl_random_randint(const uint32_t* key, int32_t impl, int64_t n, int32_t minval, int32_t maxval, int32_t* out, void* hip_stream) {
  return launch_random_randint(key, impl, n, minval, maxval, out, (hipStream_t)hip_stream);
}

// ---- per-sample-time network evaluation ----------------------------------------------------------
// The network as the MODE_EVAL kernel sees it: inputs = the x rows only; the time and condition rows of Dense_0 arrive as a
// per-sample accumulator init row (frl_time_rows_kernel + frl_state_bias_kernel).
struct EvalPlan {
  FrlDesc d;        // carrier descriptor of the MFMA kernel
  Plan pl;
  int Dp, C;        // time rows / condition rows of Dense_0
  size_t off_cfeat, off_u, off_zt, total_floats;
};

static int eval_plan(const FrlDesc* desc, EvalPlan& e) {
  int rc = validate_desc(desc);
  if (rc != FRL_OK) return rc;
  if (desc->dtype != FRL_DTYPE_F32) { set_error("frl_net_eval: f32 only"); return FRL_EUNSUPPORTED; }
  memset(&e.d, 0, sizeof(e.d));
  FrlDesc& d = e.d;
  d.abi_version = FRL_ABI_VERSION;
  d.kind = FRL_KIND_CNF; d.solver = FRL_SOLVER_EULER; d.steps = 1;     // carrier only: the mode is overridden below
  d.n_states = (int64_t)desc->n_states * desc->particles; d.particles = 1;
  d.x_dim = desc->x_dim; d.cond_dim = 0;
  d.time_embed = FRL_TEMB_NONE;
  d.n_hidden = desc->n_hidden;
  for (int l = 0; l < FRL_MAX_HIDDEN; ++l) d.hidden[l] = desc->hidden[l];
  d.act = desc->act; d.dtype = FRL_DTYPE_F32; d.layer_norm = desc->layer_norm; d.tile_policy = desc->tile_policy;
  d.backbone = desc->backbone; d.res_multiplier = desc->res_multiplier;
  rc = validate_desc(&d);
  if (rc != FRL_OK) return rc;
  Plan& pl = e.pl;
  pl.A = d.x_dim; pl.C = 0; pl.T = 1; pl.L = (d.backbone != FRL_BACKBONE_MLP) ? 1 : d.n_hidden; pl.D = 0; pl.Dp = 0;
  pl.AO = d.x_dim; pl.x_row0 = 0; pl.c_row0 = 0;
  pl.H = d.hidden[0];
  plan_layout(d, pl);
  pl.mode = MODE_EVAL;
  if (pl.H % 128 != 0) { set_error("frl_net_eval: hidden width %d is not a multiple of 128", pl.H); return FRL_EUNSUPPORTED; }
  e.Dp = desc->time_embed == FRL_TEMB_NONE ? 1 : desc->time_dim;
  e.C = desc->cond_dim;
  size_t o = pl.total_floats;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
  e.off_cfeat = take(desc->cond_embed ? (size_t)desc->n_states * COND_EMBED_DIM : 0);
  e.off_u = take((size_t)pl.B * (e.Dp + e.C));
  e.off_zt = take((size_t)pl.B * pl.H);
  e.total_floats = o;
  return FRL_OK;
}

size_t frl_net_eval_workspace_bytes(const FrlDesc* desc) {
  EvalPlan e;
  if (eval_plan(desc, e) != FRL_OK) return 0;
  return e.total_floats * sizeof(float);
}

int frl_net_eval(const FrlDesc* desc, const FrlParams* params, const FrlEvalBufs* bufs, void* workspace, size_t workspace_bytes,
                 void* hip_stream) {
  clear_kernels();
  EvalPlan e;
  int rc = eval_plan(desc, e);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  const Plan& pl = e.pl;
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < e.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B", workspace_bytes, e.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  if (pl.B == 0) return FRL_OK;
  if (bufs->x == nullptr || bufs->t == nullptr || bufs->out == nullptr) { set_error("bufs.x / t / out is NULL"); return FRL_EINVAL; }
  if ((desc->cond_dim > 0) != (bufs->cond != nullptr)) { set_error("bufs.cond must be non-NULL exactly when cond_dim > 0"); return FRL_EINVAL; }
  for (int l = 0; l <= pl.L; ++l)
    if (params->kernel[l] == nullptr || params->bias[l] == nullptr) { set_error("params.kernel/bias[%d] is NULL", l); return FRL_EINVAL; }
  ResArgs ra;
  if (desc->backbone != FRL_BACKBONE_MLP) {
    rc = fill_res_args(desc, params, ra);
    if (rc != FRL_OK) return rc;
  }
  if (desc->time_embed != FRL_TEMB_NONE) {
    for (int k = 0; k < 2; ++k)
      if (params->tmlp_kernel[k] == nullptr || params->tmlp_bias[k] == nullptr) { set_error("params.tmlp_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
    if (desc->time_embed == FRL_TEMB_FOURIER && params->time_kernel == nullptr) { set_error("params.time_kernel is NULL"); return FRL_EINVAL; }
    if ((size_t)desc->time_dim * 12 * sizeof(float) > 60000) { set_error("time_dim %d too large", desc->time_dim); return FRL_EUNSUPPORTED; }
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  float* ws = (float*)workspace;
  // packed x rows of Dense_0, Dense_L, tb = bias_0 (no time rows in the carrier plan)
  PArgs pa;
  memset(&pa, 0, sizeof(pa));
  for (int l = 0; l <= pl.L; ++l) { pa.p.kernel[l] = params->kernel[l]; pa.p.bias[l] = params->bias[l]; }
  pa.ws = ws; pa.plan = pl; pa.kind = FRL_KIND_CNF; pa.time_embed = FRL_TEMB_NONE;
  rc = launch_prepare(pa, stream);
  if (rc != FRL_OK) return rc;
  if (desc->layer_norm && desc->backbone == FRL_BACKBONE_MLP) {
    float* ln = ws + pl.off_ln;
    for (int l = 0; l < pl.L; ++l) {
      if (params->ln_scale[l] == nullptr || params->ln_bias[l] == nullptr) { set_error("params.ln_scale/ln_bias[%d] is NULL (desc.layer_norm = 1)", l); return FRL_EINVAL; }
      hipError_t er = hipMemcpyAsync(ln + (size_t)(2 * l) * pl.H, params->ln_scale[l], pl.H * sizeof(float), hipMemcpyDeviceToDevice, stream);
      if (er == hipSuccess) er = hipMemcpyAsync(ln + (size_t)(2 * l + 1) * pl.H, params->ln_bias[l], pl.H * sizeof(float), hipMemcpyDeviceToDevice, stream);
      if (er != hipSuccess) { set_error("LayerNorm rows: %s", hipGetErrorString(er)); return FRL_EHIP; }
    }
  }
  const float* cond = bufs->cond;
  if (desc->cond_embed) {
    for (int k = 0; k < 2; ++k)
      if (params->cond_kernel[k] == nullptr || params->cond_bias[k] == nullptr) { set_error("params.cond_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
    rc = launch_cond_embed(desc->act, bufs->cond, desc->n_states, desc->cond_in_dim, params->cond_kernel[0], params->cond_bias[0],
                           params->cond_kernel[1], params->cond_bias[1], ws + e.off_cfeat, stream);
    if (rc != FRL_OK) return rc;
    cond = ws + e.off_cfeat;
  }
  // u[b] = [tfeat(t_b) | cond(state(b))]; zt[b] = u[b] @ kernel_0[A:]  (bias_0 is in tb)
  rc = launch_time_rows(*desc, *params, bufs->t, cond, e.C, ws + e.off_u, pl.B, stream);
  if (rc != FRL_OK) return rc;
  rc = launch_state_bias(ws + e.off_u, pl.B, e.Dp + e.C, pl.H, params->kernel[0] + (size_t)desc->x_dim * pl.H, nullptr, ws + e.off_zt, stream);
  if (rc != FRL_OK) return rc;
  KArgs a;
  memset(&a, 0, sizeof(a));
  a.xT = bufs->x; a.x0 = bufs->out; a.state_bias = ws + e.off_zt;
  for (int l = 1; l < pl.L; ++l) { a.wh[l - 1] = params->kernel[l]; a.bh[l - 1] = params->bias[l]; }
  a.ws_coef = ws + pl.off_coef; a.ws_tb = ws + pl.off_tb; a.ws_w0p = ws + pl.off_w0p;
  a.ws_wlp = ws + pl.off_wlp; a.ws_wlq = ws + pl.off_wlq; a.ws_blp = ws + pl.off_blp;
  a.ws_ln = (desc->layer_norm && desc->backbone == FRL_BACKBONE_MLP) ? ws + pl.off_ln : nullptr;
  a.ln_post_act = desc->layer_norm == FRL_LN_POST_ACT ? 1 + desc->act : 0;
  a.B = pl.B; a.particles = 1; a.A = pl.A; a.C = 0; a.T = 1; a.L = pl.L; a.H = pl.H; a.TR = pl.TR;
  a.NE = 1; a.AO = pl.AO;
  a.w0p_bytes = (int)(pl.w0p_rows * pl.H * sizeof(float));
  a.mode = MODE_EVAL;
  a.XR = pl.XR; a.AR = pl.AR; a.CK = pl.CK; a.merged = pl.merged; a.n_l0 = pl.n_l0; a.NS = pl.NS;
  a.wl_raw = params->kernel[pl.L];
  if (desc->backbone != FRL_BACKBONE_MLP) return launch_res_network(a, ra, desc->act, desc->tile_policy, stream);
  return launch_f32_network(a, desc->act, desc->tile_policy, stream);
}

// ---- single-evaluation VJP: the evaluation under jax.grad --------------------------------------------------------------
struct EvalBwdPlan {
  EvalPlan e;           // carrier descriptor + forward tables (C = 0: time / condition rows hoisted into per-sample rows)
  BwdPlan bp;           // stash, transposed images, dW slots behind them
  int Cp;               // rows of u = [tfeat | cond] = Dp + C
  size_t off_cfeat, off_u, off_zt, off_du, off_tsc, off_ceb, total_floats;
  int layered;          // 1: layer-by-layer path (csrc/net_vjp_layered.hip): 512-wide, LayerNorm, residual predictors, wide inputs
  LayeredPlan lp;
  size_t off_lp;        // its workspace region
};

static int eval_bwd_plan(const FrlDesc* desc, EvalBwdPlan& q) {
  int rc = eval_plan(desc, q.e);
  if (rc != FRL_OK) return rc;
  const Plan& pl = q.e.pl;
  q.Cp = q.e.Dp + q.e.C;
  q.layered = (pl.H != 256 || pl.TR != 32 || desc->layer_norm != 0 || desc->backbone != FRL_BACKBONE_MLP || pl.AR > 12 || pl.A + q.Cp > 224) ? 1 : 0;
  if (desc->time_embed != FRL_TEMB_NONE && desc->time_dim > 128) { set_error("single-evaluation VJP: time_dim <= 128"); return FRL_EUNSUPPORTED; }
  if (q.layered) {
    // everything the fused sweep does not carry: forward tables are not needed, only the per-sample rows and the layer buffers
    rc = layered_plan(*desc, q.Cp, q.lp);
    if (rc != FRL_OK) return rc;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    const int D = desc->time_embed == FRL_TEMB_NONE ? 0 : desc->time_dim;
    q.off_cfeat = take(desc->cond_embed ? (size_t)desc->n_states * COND_EMBED_DIM : 0);
    q.off_u = take((size_t)pl.B * q.Cp);
    q.off_zt = 0; q.off_du = 0;
    q.off_tsc = take(time_bwd_samples_scratch_floats(pl.B, D, desc->time_embed == FRL_TEMB_FOURIER));
    q.off_ceb = take(desc->cond_embed ? (size_t)desc->n_states * 3 * COND_EMBED_DIM : 0);
    q.off_lp = take(q.lp.total_floats);
    q.total_floats = o;
    return FRL_OK;
  }
  q.bp.f = pl;
  bwd_plan_finish(q.bp, 0, desc->n_states, pl.A + q.Cp);
  q.off_lp = 0;
  size_t o = q.bp.total_floats;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
  const int D = desc->time_embed == FRL_TEMB_NONE ? 0 : desc->time_dim;
  q.off_cfeat = take(desc->cond_embed ? (size_t)desc->n_states * COND_EMBED_DIM : 0);
  q.off_u = take((size_t)pl.B * q.Cp);
  q.off_zt = take((size_t)pl.B * pl.H);
  q.off_du = take((size_t)pl.B * q.Cp);
  q.off_tsc = take(time_bwd_samples_scratch_floats(pl.B, D, desc->time_embed == FRL_TEMB_FOURIER));
  q.off_ceb = take(desc->cond_embed ? (size_t)desc->n_states * 3 * COND_EMBED_DIM : 0);
  q.total_floats = o;
  return FRL_OK;
}

static void fill_eval_bargs(const FrlDesc* desc, const FrlParams* params, const EvalBwdPlan& q, float* ws, BArgs& b) {
  const Plan& pl = q.e.pl;
  const BwdPlan& bp = q.bp;
  memset(&b, 0, sizeof(b));
  KArgs& a = b.k;
  for (int l = 1; l < pl.L; ++l) { a.wh[l - 1] = params->kernel[l]; a.bh[l - 1] = params->bias[l]; b.wt[l - 1] = ws + bp.off_wt[l - 1]; }
  a.ws_coef = ws + pl.off_coef; a.ws_tb = ws + pl.off_tb; a.ws_w0p = ws + pl.off_w0p;
  a.ws_wlp = ws + pl.off_wlp; a.ws_wlq = ws + pl.off_wlq; a.ws_blp = ws + pl.off_blp;
  a.B = pl.B; a.particles = 1; a.A = pl.A; a.C = 0; a.T = 1; a.L = pl.L; a.H = pl.H; a.TR = pl.TR;
  a.NE = 1; a.AO = pl.AO;
  a.w0p_bytes = (int)(pl.w0p_rows * pl.H * sizeof(float));
  a.mode = MODE_EVAL;
  a.XR = pl.XR; a.AR = pl.AR; a.CK = pl.CK; a.merged = pl.merged; a.n_l0 = pl.n_l0; a.NS = pl.NS;
  a.state_bias = ws + q.off_zt;
  b.ws_wlt = ws + bp.off_wlt; b.ws_w0xq = ws + bp.off_w0xq;
  b.hd = ws + bp.off_hd; b.ad = ws + bp.off_ad; b.gd = ws + bp.off_gd; b.ge = ws + bp.off_ge;
  b.Dp = 0;
  b.n_ts = bp.n_ts;
  b.part = ws + bp.off_part; b.n_part = bp.n_part;
  b.k0_rows = bp.k0_rows; b.ge_stride = bp.ge_stride;
  b.dw_cond = ws + q.off_u; b.dw_C = q.Cp;
}

size_t frl_net_eval_bwd_workspace_bytes(const FrlDesc* desc) {
  EvalBwdPlan q;
  if (eval_bwd_plan(desc, q) != FRL_OK) return 0;
  return q.total_floats * sizeof(float);
}

static int eval_check_params(const FrlDesc* desc, const FrlParams* params) {
  const int n_dense = desc->backbone != FRL_BACKBONE_MLP ? 1 : desc->n_hidden;
  for (int l = 0; l <= n_dense; ++l)
    if (params->kernel[l] == nullptr || params->bias[l] == nullptr) { set_error("params.kernel/bias[%d] is NULL", l); return FRL_EINVAL; }
  if (desc->backbone != FRL_BACKBONE_MLP) {
    ResArgs ra;
    int rc = fill_res_args(desc, params, ra);
    if (rc != FRL_OK) return rc;
  } else if (desc->layer_norm) {
    for (int l = 0; l < desc->n_hidden; ++l)
      if (params->ln_scale[l] == nullptr || params->ln_bias[l] == nullptr) { set_error("params.ln_scale/ln_bias[%d] is NULL (desc.layer_norm)", l); return FRL_EINVAL; }
  }
  if (desc->time_embed != FRL_TEMB_NONE) {
    for (int k = 0; k < 2; ++k)
      if (params->tmlp_kernel[k] == nullptr || params->tmlp_bias[k] == nullptr) { set_error("params.tmlp_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
    if (desc->time_embed == FRL_TEMB_FOURIER && params->time_kernel == nullptr) { set_error("params.time_kernel is NULL"); return FRL_EINVAL; }
  }
  if (desc->cond_embed)
    for (int k = 0; k < 2; ++k)
      if (params->cond_kernel[k] == nullptr || params->cond_bias[k] == nullptr) { set_error("params.cond_kernel/bias[%d] is NULL", k); return FRL_EINVAL; }
  return FRL_OK;
}

int frl_net_eval_train(const FrlDesc* desc, const FrlParams* params, const FrlEvalBufs* bufs, void* workspace, size_t workspace_bytes,
                       void* hip_stream) {
  clear_kernels();
  EvalBwdPlan q;
  int rc = eval_bwd_plan(desc, q);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  const Plan& pl = q.e.pl;
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < q.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B (frl_net_eval_bwd_workspace_bytes)", workspace_bytes, q.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  if (pl.B == 0) return FRL_OK;
  if (bufs->x == nullptr || bufs->t == nullptr || bufs->out == nullptr) { set_error("bufs.x / t / out is NULL"); return FRL_EINVAL; }
  if ((desc->cond_dim > 0) != (bufs->cond != nullptr)) { set_error("bufs.cond must be non-NULL exactly when cond_dim > 0"); return FRL_EINVAL; }
  rc = eval_check_params(desc, params);
  if (rc != FRL_OK) return rc;
  hipStream_t stream = (hipStream_t)hip_stream;
  float* ws = (float*)workspace;
  if (q.layered) {
    const float* cond = bufs->cond;
    if (desc->cond_embed) {
      rc = launch_cond_embed(desc->act, bufs->cond, desc->n_states, desc->cond_in_dim, params->cond_kernel[0], params->cond_bias[0],
                             params->cond_kernel[1], params->cond_bias[1], ws + q.off_cfeat, stream);
      if (rc != FRL_OK) return rc;
      cond = ws + q.off_cfeat;
    }
    rc = launch_time_rows(*desc, *params, bufs->t, cond, q.e.C, ws + q.off_u, pl.B, stream);
    if (rc != FRL_OK) return rc;
    return launch_layered_fwd(*desc, *params, q.lp, ws + q.off_lp, bufs->x, ws + q.off_u, bufs->out, stream);
  }
  PArgs pa;
  memset(&pa, 0, sizeof(pa));
  for (int l = 0; l <= pl.L; ++l) { pa.p.kernel[l] = params->kernel[l]; pa.p.bias[l] = params->bias[l]; }
  pa.ws = ws; pa.plan = pl; pa.kind = FRL_KIND_CNF; pa.time_embed = FRL_TEMB_NONE;
  rc = launch_prepare(pa, stream);
  if (rc != FRL_OK) return rc;
  const float* cond = bufs->cond;
  if (desc->cond_embed) {
    rc = launch_cond_embed(desc->act, bufs->cond, desc->n_states, desc->cond_in_dim, params->cond_kernel[0], params->cond_bias[0],
                           params->cond_kernel[1], params->cond_bias[1], ws + q.off_cfeat, stream);
    if (rc != FRL_OK) return rc;
    cond = ws + q.off_cfeat;
  }
  rc = launch_time_rows(*desc, *params, bufs->t, cond, q.e.C, ws + q.off_u, pl.B, stream);
  if (rc != FRL_OK) return rc;
  rc = launch_state_bias(ws + q.off_u, pl.B, q.Cp, pl.H, params->kernel[0] + (size_t)desc->x_dim * pl.H, nullptr, ws + q.off_zt, stream);
  if (rc != FRL_OK) return rc;
  BArgs b;
  fill_eval_bargs(desc, params, q, ws, b);
  b.k.xT = bufs->x; b.k.x0 = bufs->out;
  if (lat_train_ok(b.k)) return launch_lat_train(b, q.e.d.act, nullptr, 0, params->kernel[pl.L], stream);
  return launch_sampler_bwd(q.e.d, b, 1, stream);
}

int frl_net_eval_bwd(const FrlDesc* desc, const FrlParams* params, const FrlEvalBwdBufs* bufs, void* workspace, size_t workspace_bytes,
                     void* hip_stream) {
  clear_kernels();
  EvalBwdPlan q;
  int rc = eval_bwd_plan(desc, q);
  if (rc != FRL_OK) return rc;
  if (params == nullptr || bufs == nullptr) { set_error("params / bufs is NULL"); return FRL_EINVAL; }
  const Plan& pl = q.e.pl;
  if (workspace == nullptr || ((uintptr_t)workspace & 255) != 0) { set_error("workspace must be non-NULL and 256-byte aligned"); return FRL_EWORKSPACE; }
  if (workspace_bytes < q.total_floats * sizeof(float)) { set_error("workspace %zu B < required %zu B", workspace_bytes, q.total_floats * sizeof(float)); return FRL_EWORKSPACE; }
  if (pl.B == 0) return FRL_OK;
  if (bufs->x == nullptr || bufs->t == nullptr || bufs->g_out == nullptr) { set_error("bufs.x / t / g_out is NULL"); return FRL_EINVAL; }
  if ((desc->cond_dim > 0) != (bufs->cond != nullptr)) { set_error("bufs.cond must be non-NULL exactly when cond_dim > 0"); return FRL_EINVAL; }
  rc = eval_check_params(desc, params);
  if (rc != FRL_OK) return rc;
  const int n_dense = desc->backbone != FRL_BACKBONE_MLP ? 1 : desc->n_hidden;
  for (int l = 0; l <= n_dense; ++l)
    if (bufs->grad_kernel[l] == nullptr || bufs->grad_bias[l] == nullptr) { set_error("bufs.grad_kernel / grad_bias[%d] is NULL", l); return FRL_EINVAL; }
  if (desc->backbone != FRL_BACKBONE_MLP) {
    const bool simba = desc->backbone == FRL_BACKBONE_SIMBA;
    for (int i = 0; i < desc->n_hidden; ++i) {
      for (int k = 0; k < 2; ++k)
        if (bufs->grad_res_kernel[i][k] == nullptr || bufs->grad_res_bias[i][k] == nullptr) { set_error("bufs.grad_res_kernel / grad_res_bias[%d][%d] is NULL", i, k); return FRL_EINVAL; }
      if ((simba || desc->layer_norm) && (bufs->grad_res_ln_scale[i] == nullptr || bufs->grad_res_ln_bias[i] == nullptr)) { set_error("bufs.grad_res_ln_scale / grad_res_ln_bias[%d] is NULL", i); return FRL_EINVAL; }
    }
    if (simba && (bufs->grad_final_ln_scale == nullptr || bufs->grad_final_ln_bias == nullptr)) { set_error("bufs.grad_final_ln_scale / grad_final_ln_bias is NULL (Simba)"); return FRL_EINVAL; }
  } else if (desc->layer_norm) {
    for (int l = 0; l < desc->n_hidden; ++l)
      if (bufs->grad_ln_scale[l] == nullptr || bufs->grad_ln_bias[l] == nullptr) { set_error("bufs.grad_ln_scale / grad_ln_bias[%d] is NULL (desc.layer_norm)", l); return FRL_EINVAL; }
  }
  if (desc->time_embed != FRL_TEMB_NONE) {
    for (int k = 0; k < 2; ++k)
      if (bufs->grad_tmlp_kernel[k] == nullptr || bufs->grad_tmlp_bias[k] == nullptr) { set_error("bufs.grad_tmlp_kernel / grad_tmlp_bias[%d] is NULL", k); return FRL_EINVAL; }
    if (desc->time_embed == FRL_TEMB_FOURIER && bufs->grad_time_kernel == nullptr) { set_error("bufs.grad_time_kernel is NULL"); return FRL_EINVAL; }
  }
  if (desc->cond_embed)
    for (int k = 0; k < 2; ++k)
      if (bufs->grad_cond_kernel[k] == nullptr || bufs->grad_cond_bias[k] == nullptr) { set_error("bufs.grad_cond_kernel / grad_cond_bias[%d] is NULL", k); return FRL_EINVAL; }
  hipStream_t stream = (hipStream_t)hip_stream;
  float* ws = (float*)workspace;
  float* du = ws + q.off_du;
  int du_stride = q.Cp;
  if (q.layered) {
    rc = launch_layered_bwd(*desc, *params, q.lp, ws + q.off_lp, bufs->g_out, *bufs, bufs->g_x, stream);
    if (rc != FRL_OK) return rc;
    du = ws + q.off_lp + q.lp.off_dinp + desc->x_dim;          // dL/dinp = [dL/dx | dL/du], row stride K0
    du_stride = q.lp.K0;
  } else {
  rc = launch_prepare_bwd(*params, q.bp, ws, stream);          // transposed hidden kernels, W_L^T, packed x rows of Dense_0
  if (rc != FRL_OK) return rc;
  BArgs b;
  fill_eval_bargs(desc, params, q, ws, b);
  b.xt_hist = bufs->x;                                        // T = 1: the evaluation's input is "history[0]"
  b.g_x0 = bufs->g_out; b.g_xT = bufs->g_x; b.eps_hist = nullptr;
  for (int l = 0; l <= pl.L; ++l) { b.grad_kernel[l] = bufs->grad_kernel[l]; b.grad_bias[l] = bufs->grad_bias[l]; }
  b.grad_tb = bufs->grad_bias[0];                             // one "step": dL/dtb[0] IS dL/dbias_0 (tb = bias_0 in this plan)
  rc = launch_sampler_bwd(q.e.d, b, 2, stream);               // backward sweep from the stash + dW + fixed-order reduce
  if (rc != FRL_OK) return rc;
  // dL/du = dL/dz_0 . kernel_0[A:]^T per sample, then the two branches that produced u = [tfeat | cond]
  rc = launch_eval_du(b, params->kernel[0] + (size_t)desc->x_dim * pl.H, q.Cp, du, stream);
  if (rc != FRL_OK) return rc;
  }
  if (desc->time_embed != FRL_TEMB_NONE) {
    TimeBwdSampleArgs ta;
    memset(&ta, 0, sizeof(ta));
    ta.t = bufs->t; ta.du = du; ta.du_stride = du_stride;
    ta.time_kernel = params->time_kernel; ta.tk0 = params->tmlp_kernel[0]; ta.tb0 = params->tmlp_bias[0];
    ta.tk1 = params->tmlp_kernel[1]; ta.tb1 = params->tmlp_bias[1];
    ta.scratch = ws + q.off_tsc;
    ta.g_tk0 = bufs->grad_tmlp_kernel[0]; ta.g_tb0 = bufs->grad_tmlp_bias[0]; ta.g_tk1 = bufs->grad_tmlp_kernel[1]; ta.g_tb1 = bufs->grad_tmlp_bias[1];
    ta.g_time_kernel = bufs->grad_time_kernel;
    ta.B = pl.B; ta.D = desc->time_dim; ta.time_embed = desc->time_embed;
    rc = launch_time_bwd_samples(ta, stream);
    if (rc != FRL_OK) return rc;
  }
  if (desc->cond_embed) {
    rc = launch_cond_embed_bwd(desc->act, bufs->cond, desc->n_states, desc->particles, desc->cond_in_dim, params->cond_kernel[0],
                               params->cond_bias[0], params->cond_kernel[1], params->cond_bias[1], du + q.e.Dp, du_stride, ws + q.off_ceb,
                               bufs->grad_cond_kernel[0], bufs->grad_cond_bias[0], bufs->grad_cond_kernel[1], bufs->grad_cond_bias[1], stream);
    if (rc != FRL_OK) return rc;
  }
  return FRL_OK;
}

int frl_add_noise(int32_t kind, int64_t B, int32_t A, const float* x0, const float* e, const void* t, const float* alpha_hats,
                  int32_t steps, float p0, float p1, float* xt, float* vel, void* hip_stream) {
  if (kind < FRL_NOISE_DDPM || kind > FRL_NOISE_FLOW) { set_error("add_noise: kind %d invalid", kind); return FRL_EINVAL; }
  if (B < 0 || A < 1) { set_error("add_noise: B %lld / A %d invalid", (long long)B, A); return FRL_EINVAL; }
  if (B == 0) return FRL_OK;
  if (x0 == nullptr || e == nullptr || t == nullptr || xt == nullptr) { set_error("add_noise: x0 / e / t / xt is NULL"); return FRL_EINVAL; }
  if (kind == FRL_NOISE_DDPM && (alpha_hats == nullptr || steps < 1)) { set_error("add_noise: alpha_hats (steps + 1,) required"); return FRL_EINVAL; }
  return launch_add_noise(kind, B, A, steps, x0, e, t, alpha_hats, p0, p1, xt, vel, (hipStream_t)hip_stream);
}

int frl_ddpm_onestep_update(int32_t solver, int64_t B, int32_t A, int32_t num_samples, const float* xt, const float* eps_theta,
                            const int32_t* t, const float* alphas, const float* alpha_hats, int32_t steps, int32_t clip, float x_min,
                            float x_max, const float* noise, float* out, void* hip_stream) {
  if (solver != FRL_SOLVER_DDPM && solver != FRL_SOLVER_DDIM) { set_error("onestep update: solver %d (ddpm / ddim, ddpm.py:16-18)", solver); return FRL_EINVAL; }
  if (B < 0 || A < 1 || num_samples < 1 || steps < 1) { set_error("onestep update: bad sizes"); return FRL_EINVAL; }
  if (B == 0) return FRL_OK;
  if (xt == nullptr || eps_theta == nullptr || t == nullptr || alphas == nullptr || alpha_hats == nullptr || out == nullptr) { set_error("onestep update: NULL argument"); return FRL_EINVAL; }
  if (solver == FRL_SOLVER_DDPM && noise == nullptr) { set_error("onestep update: the ddpm solver needs noise (B, num_samples, A)"); return FRL_EINVAL; }
  return launch_onestep_update(solver, B, A, num_samples, steps, xt, eps_theta, t, alphas, alpha_hats, clip, x_min, x_max, noise, out, (hipStream_t)hip_stream);
}

int frl_mse_loss(int64_t n, const float* pred, const float* target, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                 void* hip_stream) {
  if (n < 1 || pred == nullptr || target == nullptr || loss == nullptr) { set_error("mse: bad arguments"); return FRL_EINVAL; }
  if (workspace == nullptr || workspace_bytes < FRL_MSE_WORKSPACE_BYTES || ((uintptr_t)workspace & 7) != 0) { set_error("mse: workspace of FRL_MSE_WORKSPACE_BYTES required"); return FRL_EWORKSPACE; }
  return launch_mse(n, pred, target, loss, grad, workspace, (hipStream_t)hip_stream);
}

int frl_selftest_mfma32(const float* A, const float* B, float* D, int K, void* hip_stream) {
  if (A == nullptr || B == nullptr || D == nullptr || K < 2 || (K & 1)) { set_error("selftest: bad arguments"); return FRL_EINVAL; }
  return launch_selftest_mfma32(A, B, D, K, (hipStream_t)hip_stream);
}

}  // extern "C"
