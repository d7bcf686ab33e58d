// XLA-FFI (jax.ffi) registration shim: a thin translation of XLA custom calls onto the plain C ABI
// (include/flowrl_hip.h).  Handlers: FrlSamplerFwd (noise-tensor or keyed rng-in / rng-out form), FrlSamplerFwdTrain (the
// same plus the stash workspace as a result: the residual of a jax.custom_vjp) and FrlSamplerBwd (from xt_hist alone, or
// from that stash).
//
// NOT compiled by build.py in this container: jaxlib's headers (xla/ffi/api/ffi.h, found via jax.ffi.include_dir())
// are absent here and cannot be fetched.  What IS checked here: tests/test_abi.py::test_ffi_shim_compile_smoke runs
// `g++ -fsyntax-only` on this file against tests/ffi_stub/xla/ffi/api/ffi.h — a minimal in-tree declaration of the
// handful of xla::ffi names used below, a COMPILE SMOKE ONLY (names, arity, types), not jaxlib's header.  A maintainer
// builds the real thing as INTEGRATION.md says:
//
//   hipcc -std=c++17 -fPIC -shared -I$(python -c "import jax; print(jax.ffi.include_dir())") \
//         -Iinclude flow-rl_amd/csrc/xla_ffi_shim.cc -Lflow-rl_amd/flowrl_hip -lflowrl_hip -o libflowrl_hip_ffi.so
//
// The xla::ffi surface is kept small on purpose: every operand / result arrives through RemainingArgs / RemainingRets
// as AnyBuffer in a FIXED order, every static through one attribute Dictionary (missing optional attributes take the
// defaults below), so the binding is the same whatever the layer count or the optional pieces.
//
// FrlSamplerFwd operands (a 0-element buffer = "absent"):
//    0 xT (B, A) f32          1 cond (n_states, C | cond_in_dim) f32
//    2 noise (T, B, A) f32   -- or, with attr rng_impl != 0, the raw threefry key (2,) u32 (jax.random.key_data)
//    3..10 betas, alphas, alpha_hats, postvars, ts, mid_ts, mid_alphas, mid_sigmas   (FrlSchedule)
//   11 time_embedding/kernel     12..15 MLP_0/Dense_{0,1} kernel, bias
//   16..19 cond_embedding/Dense_{0,1} kernel, bias (attr cond_embed = 1)
//   20.. (kernel_i, bias_i) for i = 0..L, then with attr layer_norm = 1 (scale_i, bias_i) for i = 0..L-1
// results: x0 (B, A); then rng_out (2,) u32 when rng_impl != 0; then xt_hist, eps_hist (T, B, A) when want_history = 1.
//
// Residual predictors (attr backbone = 1 ResidualMLP / 2 Simba, forward handlers only): operands 20.. are Dense_0 (kernel, bias),
// Dense_1 (kernel, bias), then per block Dense_0 (kernel, bias), Dense_1 (kernel, bias), LayerNorm_0 (scale, bias) — the
// LayerNorm pair present when attr layer_norm = 1 or Simba — and for Simba the final LayerNorm_0 (scale, bias).
//
// FrlSamplerFwdTrain: FrlSamplerFwd with want_history forced on and ONE more result after eps_hist: the stash workspace, a
// u8 buffer of frl_bwd_workspace_bytes(desc) bytes (256-byte aligned, as XLA allocates) that the backward consumes.
//
// FrlSamplerBwd operands: 0 xt_hist, 1 cond, 2 g_x0, 3..19 as above, 20.. (kernel_i, bias_i) for i = 0..L; with attr
// have_stash = 1 two more at the end: eps_hist (T, B, A) and the stash workspace of FrlSamplerFwdTrain (u8) — the backward
// writes its own scratch into that buffer, so the binding DONATES it (input_output_aliases onto the last result).
// results: g_xT (B, A); (grad_kernel_i, grad_bias_i) for i = 0..L; grad_tb (T, H); then, with attr native_time = 1,
//          grad MLP_0/Dense_{0,1} kernel, bias and grad time_embedding/kernel (FrlBwdBufs' optional outputs); then, with attr
//          cond_embed = 1, grad cond_embedding/Dense_{0,1} kernel, bias; then, with have_stash = 1, the donated workspace.
#include <cstdint>
#include <cstring>

#include <hip/hip_runtime_api.h>

#include "flowrl_hip.h"
#include "xla/ffi/api/ffi.h"

namespace ffi = xla::ffi;

namespace {

constexpr int kFixedOperands = 20;

ffi::Error Invalid(const char* msg) { return ffi::Error(ffi::ErrorCode::kInvalidArgument, msg); }

ffi::Error FromRc(int rc) {
  if (rc == FRL_OK) return ffi::Error::Success();
  return ffi::Error(rc == FRL_EUNSUPPORTED ? ffi::ErrorCode::kUnimplemented
                    : rc == FRL_EWORKSPACE ? ffi::ErrorCode::kResourceExhausted
                                           : ffi::ErrorCode::kInvalidArgument,
                    frl_last_error());
}

template <typename T>
T AttrOr(const ffi::Dictionary& attrs, const char* name, T fallback) {
  auto v = attrs.get<T>(name);
  return v.has_value() ? *v : fallback;
}

// operand i as a device pointer of the expected element type; nullptr when it has no elements
template <typename T>
bool Operand(const ffi::RemainingArgs& args, size_t i, ffi::DataType want, const T** out, int64_t* last_dim = nullptr) {
  auto b = args.get<ffi::AnyBuffer>(i);
  if (!b.has_value() || (b->element_count() != 0 && b->element_type() != want)) return false;
  *out = b->element_count() ? static_cast<const T*>(b->untyped_data()) : nullptr;
  if (last_dim != nullptr) *last_dim = b->dimensions().size() ? b->dimensions().back() : 0;
  return true;
}

template <typename T>
bool Output(const ffi::RemainingRets& rets, size_t i, ffi::DataType want, T** out) {
  auto b = rets.get<ffi::AnyBuffer>(i);
  if (!b.has_value() || (*b)->element_type() != want) return false;
  *out = static_cast<T*>((*b)->untyped_data());
  return true;
}

// FrlDesc from the attribute dictionary + operand shapes; FrlParams / FrlSchedule from operands 3.. (shared by both handlers)
ffi::Error Marshal(const ffi::RemainingArgs& args, const ffi::Dictionary& attrs, int64_t x_dim, int64_t cond_last,
                   FrlDesc* d, FrlParams* p, FrlSchedule* s, size_t trailing = 0) {
  std::memset(d, 0, sizeof(*d));
  std::memset(p, 0, sizeof(*p));
  std::memset(s, 0, sizeof(*s));
  d->abi_version = FRL_ABI_VERSION;
  d->kind = AttrOr<int32_t>(attrs, "kind", FRL_KIND_DDPM);
  d->solver = AttrOr<int32_t>(attrs, "solver", FRL_SOLVER_DDPM);
  d->steps = AttrOr<int32_t>(attrs, "steps", 0);
  d->n_states = AttrOr<int64_t>(attrs, "n_states", 0);
  d->particles = AttrOr<int32_t>(attrs, "particles", 1);
  d->x_dim = (int32_t)x_dim;
  d->time_embed = AttrOr<int32_t>(attrs, "time_embed", FRL_TEMB_POSITIONAL);
  d->time_dim = AttrOr<int32_t>(attrs, "time_dim", 64);
  d->act = AttrOr<int32_t>(attrs, "act", FRL_ACT_MISH);
  d->clip = AttrOr<int32_t>(attrs, "clip", 0);
  // an absent side is unbounded, as jnp.clip(x, None, None) leaves it (ddpm.py:200-201, cnf.py:109-110)
  d->x_min = AttrOr<float>(attrs, "x_min", -__builtin_huge_valf());
  d->x_max = AttrOr<float>(attrs, "x_max", __builtin_huge_valf());
  d->dtype = AttrOr<int32_t>(attrs, "dtype", FRL_DTYPE_F32);
  d->want_history = AttrOr<int32_t>(attrs, "want_history", 0);
  d->cond_embed = AttrOr<int32_t>(attrs, "cond_embed", 0);
  d->tile_policy = AttrOr<int32_t>(attrs, "tile_policy", FRL_TILE_AUTO);
  d->rng_impl = AttrOr<int32_t>(attrs, "rng_impl", FRL_RNG_NONE);
  d->layer_norm = AttrOr<int32_t>(attrs, "layer_norm", 0);
  if (d->cond_embed) { d->cond_in_dim = (int32_t)cond_last; d->cond_dim = 128; }
  else d->cond_dim = (int32_t)cond_last;

  const float** sched[8] = {&s->betas, &s->alphas, &s->alpha_hats, &s->postvars, &s->ts, &s->mid_ts, &s->mid_alphas, &s->mid_sigmas};
  for (int i = 0; i < 8; ++i)
    if (!Operand<float>(args, 3 + i, ffi::DataType::F32, sched[i])) return Invalid("schedule operands (3..10) must be f32");
  bool ok = Operand<float>(args, 11, ffi::DataType::F32, &p->time_kernel);
  for (int k = 0; k < 2 && ok; ++k) {
    ok = Operand<float>(args, 12 + 2 * k, ffi::DataType::F32, &p->tmlp_kernel[k]) &&
         Operand<float>(args, 13 + 2 * k, ffi::DataType::F32, &p->tmlp_bias[k]) &&
         Operand<float>(args, 16 + 2 * k, ffi::DataType::F32, &p->cond_kernel[k]) &&
         Operand<float>(args, 17 + 2 * k, ffi::DataType::F32, &p->cond_bias[k]);
  }
  if (!ok) return Invalid("time / cond-embedding operands (11..19) must be f32");

  d->backbone = AttrOr<int32_t>(attrs, "backbone", FRL_BACKBONE_MLP);
  if (d->backbone != FRL_BACKBONE_MLP) {
    d->res_multiplier = AttrOr<int32_t>(attrs, "res_multiplier", 4);
    const bool simba = d->backbone == FRL_BACKBONE_SIMBA, ln = simba || d->layer_norm;
    const size_t per_block = ln ? 6 : 4, head = 4 + (simba ? 2 : 0);
    const size_t restb = args.size() - kFixedOperands - trailing;
    if (args.size() < kFixedOperands + trailing + head + per_block || (restb - head) % per_block != 0) return Invalid("residual predictor operands: Dense_0, Dense_1, blocks, [final LayerNorm]");
    d->n_hidden = (int32_t)((restb - head) / per_block);
    if (d->n_hidden > FRL_MAX_HIDDEN) return Invalid("block count outside [1, FRL_MAX_HIDDEN]");
    size_t o = kFixedOperands;
    int64_t width = 0;
    bool okr = true;
    for (int i = 0; i < 2 && okr; ++i, o += 2)
      okr = Operand<float>(args, o, ffi::DataType::F32, &p->kernel[i], i == 0 ? &width : nullptr) && Operand<float>(args, o + 1, ffi::DataType::F32, &p->bias[i]);
    for (int b = 0; b < d->n_hidden && okr; ++b) {
      d->hidden[b] = (int32_t)width;
      for (int k = 0; k < 2 && okr; ++k, o += 2)
        okr = Operand<float>(args, o, ffi::DataType::F32, &p->res_kernel[b][k]) && Operand<float>(args, o + 1, ffi::DataType::F32, &p->res_bias[b][k]);
      if (ln && okr) { okr = Operand<float>(args, o, ffi::DataType::F32, &p->res_ln_scale[b]) && Operand<float>(args, o + 1, ffi::DataType::F32, &p->res_ln_bias[b]); o += 2; }
    }
    if (simba && okr) okr = Operand<float>(args, o, ffi::DataType::F32, &p->final_ln_scale) && Operand<float>(args, o + 1, ffi::DataType::F32, &p->final_ln_bias);
    return okr ? ffi::Error::Success() : Invalid("residual predictor operands must be f32");
  }
  const size_t rest = args.size() - kFixedOperands - trailing;
  const size_t per_hidden = d->layer_norm ? 4 : 2;            // (kernel, bias) [+ (ln scale, ln bias)] per hidden layer
  if (args.size() < kFixedOperands + trailing + 4 || (rest - 2) % per_hidden != 0) return Invalid("layer operands: (kernel_i, bias_i) for i = 0..L [+ LayerNorm rows]");
  d->n_hidden = (int32_t)((rest - 2) / per_hidden);
  if (d->n_hidden < 1 || d->n_hidden > FRL_MAX_HIDDEN) return Invalid("hidden layer count outside [1, FRL_MAX_HIDDEN]");
  for (int l = 0; l <= d->n_hidden; ++l) {
    int64_t width = 0;
    if (!Operand<float>(args, kFixedOperands + 2 * l, ffi::DataType::F32, &p->kernel[l], &width) ||
        !Operand<float>(args, kFixedOperands + 2 * l + 1, ffi::DataType::F32, &p->bias[l]))
      return Invalid("layer operands must be f32");
    if (l < d->n_hidden) d->hidden[l] = (int32_t)width;
  }
  if (d->layer_norm) {
    const size_t base = kFixedOperands + 2 * (size_t)(d->n_hidden + 1);
    for (int l = 0; l < d->n_hidden; ++l)
      if (!Operand<float>(args, base + 2 * l, ffi::DataType::F32, &p->ln_scale[l]) ||
          !Operand<float>(args, base + 2 * l + 1, ffi::DataType::F32, &p->ln_bias[l]))
        return Invalid("LayerNorm operands must be f32");
  }
  return ffi::Error::Success();
}

ffi::Error FwdCommon(hipStream_t stream, ffi::ScratchAllocator scratch, ffi::RemainingArgs args, ffi::RemainingRets rets,
                     ffi::Dictionary attrs, bool train) {
  if (args.size() < kFixedOperands + 4) return Invalid("FrlSamplerFwd: too few operands");
  FrlDesc d; FrlParams p; FrlSchedule s;
  FrlFwdBufs b;
  std::memset(&b, 0, sizeof(b));
  int64_t x_dim = 0, cond_last = 0;
  if (!Operand<float>(args, 0, ffi::DataType::F32, &b.xT, &x_dim) || !Operand<float>(args, 1, ffi::DataType::F32, &b.cond, &cond_last))
    return Invalid("xT / cond must be f32");
  if (b.cond == nullptr) cond_last = 0;
  if (ffi::Error e = Marshal(args, attrs, x_dim, cond_last, &d, &p, &s); e.failure()) return e;
  if (train) d.want_history = 1;
  size_t out = 0;
  if (!Output<float>(rets, out++, ffi::DataType::F32, &b.x0)) return Invalid("result 0 (x0) must be f32");
  if (d.rng_impl != FRL_RNG_NONE) {          // keyed form: operand 2 is the threefry key, result 1 the advanced key
    if (!Operand<uint32_t>(args, 2, ffi::DataType::U32, &b.rng_key) || !Output<uint32_t>(rets, out++, ffi::DataType::U32, &b.rng_out))
      return Invalid("rng_impl != 0: operand 2 = key (2,) u32, result 1 = rng_out (2,) u32");
  } else if (!Operand<float>(args, 2, ffi::DataType::F32, &b.noise)) {
    return Invalid("noise must be f32");
  }
  if (d.want_history && (!Output<float>(rets, out, ffi::DataType::F32, &b.xt_hist) || !Output<float>(rets, out + 1, ffi::DataType::F32, &b.eps_hist)))
    return Invalid("want_history = 1: results xt_hist, eps_hist (T, B, A) f32");
  if (train) {      // the stash workspace is a RESULT: it outlives this call and feeds FrlSamplerBwd (have_stash = 1)
    uint8_t* stash = nullptr;
    auto wb = rets.get<ffi::AnyBuffer>(out + 2);
    if (!Output<uint8_t>(rets, out + 2, ffi::DataType::U8, &stash) || !wb.has_value()) return Invalid("FrlSamplerFwdTrain: last result = stash workspace (u8)");
    const size_t need = frl_bwd_workspace_bytes(&d);
    if (need == 0) return FromRc(frl_sampler_fwd_train(&d, &p, &s, &b, nullptr, 0, (void*)stream));   // says why
    if ((size_t)(*wb)->element_count() < need) return ffi::Error(ffi::ErrorCode::kResourceExhausted, "stash workspace result smaller than frl_bwd_workspace_bytes(desc)");
    return FromRc(frl_sampler_fwd_train(&d, &p, &s, &b, stash, (size_t)(*wb)->element_count(), (void*)stream));
  }
  if (frl_validate(&d) != FRL_OK) return FromRc(frl_validate(&d));
  const size_t need = frl_workspace_bytes(&d);
  auto ws = scratch.Allocate(need > 0 ? need : 256, 256);
  if (!ws.has_value()) return ffi::Error(ffi::ErrorCode::kResourceExhausted, "scratch allocation failed");
  return FromRc(frl_sampler_fwd(&d, &p, &s, &b, *ws, need, (void*)stream));
}

ffi::Error FrlSamplerFwdImpl(hipStream_t stream, ffi::ScratchAllocator scratch, ffi::RemainingArgs args,
                             ffi::RemainingRets rets, ffi::Dictionary attrs) {
  return FwdCommon(stream, scratch, args, rets, attrs, false);
}

ffi::Error FrlSamplerFwdTrainImpl(hipStream_t stream, ffi::ScratchAllocator scratch, ffi::RemainingArgs args,
                                  ffi::RemainingRets rets, ffi::Dictionary attrs) {
  return FwdCommon(stream, scratch, args, rets, attrs, true);
}

ffi::Error FrlSamplerBwdImpl(hipStream_t stream, ffi::ScratchAllocator scratch, ffi::RemainingArgs args,
                             ffi::RemainingRets rets, ffi::Dictionary attrs) {
  if (args.size() < kFixedOperands + 4) return Invalid("FrlSamplerBwd: too few operands");
  const bool have_stash = AttrOr<int32_t>(attrs, "have_stash", 0) != 0;
  FrlDesc d; FrlParams p; FrlSchedule s;
  FrlBwdBufs b;
  std::memset(&b, 0, sizeof(b));
  int64_t x_dim = 0, cond_last = 0, unused = 0;
  if (!Operand<float>(args, 0, ffi::DataType::F32, &b.xt_hist, &unused) || !Operand<float>(args, 1, ffi::DataType::F32, &b.cond, &cond_last) ||
      !Operand<float>(args, 2, ffi::DataType::F32, &b.g_x0, &x_dim))
    return Invalid("xt_hist / cond / g_x0 must be f32");
  if (b.cond == nullptr) cond_last = 0;
  if (ffi::Error e = Marshal(args, attrs, x_dim, cond_last, &d, &p, &s, have_stash ? 2 : 0); e.failure()) return e;
  if (d.backbone != FRL_BACKBONE_MLP) return ffi::Error(ffi::ErrorCode::kUnimplemented, "FrlSamplerBwd: residual predictors have no backward kernel");
  d.want_history = 1;
  size_t out = 0;
  if (!Output<float>(rets, out++, ffi::DataType::F32, &b.g_xT)) return Invalid("result 0 (g_xT) must be f32");
  for (int l = 0; l <= d.n_hidden; ++l)
    if (!Output<float>(rets, out++, ffi::DataType::F32, &b.grad_kernel[l]) || !Output<float>(rets, out++, ffi::DataType::F32, &b.grad_bias[l]))
      return Invalid("results: (grad_kernel_i, grad_bias_i) for i = 0..L, f32");
  if (!Output<float>(rets, out++, ffi::DataType::F32, &b.grad_tb)) return Invalid("result grad_tb (T, H) must be f32");
  if (AttrOr<int32_t>(attrs, "native_time", 0)) {
    for (int k = 0; k < 2; ++k)
      if (!Output<float>(rets, out++, ffi::DataType::F32, &b.grad_tmlp_kernel[k]) || !Output<float>(rets, out++, ffi::DataType::F32, &b.grad_tmlp_bias[k]))
        return Invalid("native_time = 1: results grad MLP_0/Dense_{0,1} kernel, bias");
    if (d.time_embed == FRL_TEMB_FOURIER && !Output<float>(rets, out++, ffi::DataType::F32, &b.grad_time_kernel))
      return Invalid("native_time = 1 with a Fourier embedding: result grad time_embedding/kernel");
  } else {
    // the binding differentiates the time branch itself from grad_tb (jax.vjp): the library leaves grad_bias_0 alone, so
    // the result buffer is zero-filled here rather than handed back uninitialised (the binding ADDS its own term)
    auto gb0 = rets.get<ffi::AnyBuffer>(2);
    if (gb0.has_value() && b.grad_bias[0] != nullptr &&
        hipMemsetAsync(b.grad_bias[0], 0, (size_t)(*gb0)->element_count() * sizeof(float), stream) != hipSuccess)
      return ffi::Error(ffi::ErrorCode::kInternal, "zero-fill of grad_bias_0 failed");
    b.grad_bias[0] = nullptr;
  }
  if (d.cond_embed)
    for (int k = 0; k < 2; ++k)
      if (!Output<float>(rets, out++, ffi::DataType::F32, &b.grad_cond_kernel[k]) || !Output<float>(rets, out++, ffi::DataType::F32, &b.grad_cond_bias[k]))
        return Invalid("cond_embed = 1: results grad cond_embedding/Dense_{0,1} kernel, bias");
  const size_t need = frl_bwd_workspace_bytes(&d);
  if (need == 0) return FromRc(frl_sampler_bwd(&d, &p, &s, &b, nullptr, 0, (void*)stream));   // says why
  if (have_stash) {
    // operands ... eps_hist, stash workspace; the workspace is donated onto the last result (the backward writes into it)
    const uint8_t* ws_in = nullptr;
    uint8_t* ws_out = nullptr;
    auto wb = rets.get<ffi::AnyBuffer>(out);
    if (!Operand<float>(args, args.size() - 2, ffi::DataType::F32, &b.eps_hist) || !Operand<uint8_t>(args, args.size() - 1, ffi::DataType::U8, &ws_in) ||
        !Output<uint8_t>(rets, out, ffi::DataType::U8, &ws_out) || !wb.has_value())
      return Invalid("have_stash = 1: last operands eps_hist (f32), stash workspace (u8); last result the donated workspace (u8)");
    if ((size_t)(*wb)->element_count() < need) return ffi::Error(ffi::ErrorCode::kResourceExhausted, "stash workspace smaller than frl_bwd_workspace_bytes(desc)");
    if (ws_in != ws_out && hipMemcpyAsync(ws_out, ws_in, need, hipMemcpyDeviceToDevice, stream) != hipSuccess)   // not aliased: keep the stash
      return ffi::Error(ffi::ErrorCode::kInternal, "stash workspace copy failed");
    b.have_stash = 1;
    return FromRc(frl_sampler_bwd(&d, &p, &s, &b, ws_out, (size_t)(*wb)->element_count(), (void*)stream));
  }
  auto ws = scratch.Allocate(need, 256);
  if (!ws.has_value()) return ffi::Error(ffi::ErrorCode::kResourceExhausted, "scratch allocation failed (the VJP stash grows with B * T * L)");
  return FromRc(frl_sampler_bwd(&d, &p, &s, &b, *ws, need, (void*)stream));
}

// FrlSamplerSelectFwd: jit_sample_and_select (dql.py:130-162) / jit_sample_actions (qvpo.py:21-44) after the caller's draws, ONE
// launch (frl_sampler_select_fwd).  Operands: FrlSamplerFwd's (0 xT (n_states * particles, A), 1 obs (n_states, S) — the
// sampler's condition AND the critic's observation, 2 noise, 3..19, the sampler's layers), then at the END: gumbel
// (n_states, particles) f32 or empty, and the critic's stacked leaves (kernel_i (E, in, out), bias_i (E, out)) for
// i = 0..Lc (backbone Dense_0..Dense_{Lc-1}, then the head).  Attributes: FrlSamplerFwd's + critic_layers (Lc), ensemble,
// critic_act, inv_temperature (1 / (1e-6 + T); used with gumbel), want_q.  Results: best_action (n_states, A) f32,
// best_idx (n_states,) s32, then q (n_states, particles) f32 when want_q = 1.
ffi::Error FrlSamplerSelectFwdImpl(hipStream_t stream, ffi::ScratchAllocator scratch, ffi::RemainingArgs args,
                                   ffi::RemainingRets rets, ffi::Dictionary attrs) {
  const int32_t lc = AttrOr<int32_t>(attrs, "critic_layers", 0);
  if (lc < 1 || lc > FRL_MAX_HIDDEN) return Invalid("critic_layers outside [1, FRL_MAX_HIDDEN]");
  const size_t trailing = 1 + 2 * (size_t)(lc + 1);
  if (args.size() < kFixedOperands + 4 + trailing) return Invalid("FrlSamplerSelectFwd: too few operands");
  FrlDesc d; FrlParams p; FrlSchedule s;
  FrlFwdBufs b;
  std::memset(&b, 0, sizeof(b));
  int64_t x_dim = 0, cond_last = 0;
  if (!Operand<float>(args, 0, ffi::DataType::F32, &b.xT, &x_dim) || !Operand<float>(args, 1, ffi::DataType::F32, &b.cond, &cond_last) ||
      !Operand<float>(args, 2, ffi::DataType::F32, &b.noise))
    return Invalid("xT / obs / noise must be f32");
  if (b.cond == nullptr) cond_last = 0;
  if (ffi::Error e = Marshal(args, attrs, x_dim, cond_last, &d, &p, &s, trailing); e.failure()) return e;
  d.want_history = 0;
  FrlCriticDesc cd;
  FrlCriticParams cp;
  FrlSelectBufs sb;
  std::memset(&cd, 0, sizeof(cd)); std::memset(&cp, 0, sizeof(cp)); std::memset(&sb, 0, sizeof(sb));
  cd.abi_version = FRL_ABI_VERSION;
  cd.n_states = d.n_states; cd.particles = d.particles;
  cd.obs_dim = (int32_t)cond_last; cd.act_dim = d.x_dim; cd.n_hidden = lc;
  cd.act = AttrOr<int32_t>(attrs, "critic_act", d.act);
  cd.ensemble = AttrOr<int32_t>(attrs, "ensemble", 2);
  const size_t base = args.size() - trailing;
  if (!Operand<float>(args, base, ffi::DataType::F32, &sb.gumbel)) return Invalid("gumbel must be f32 (or empty)");
  sb.inv_temperature = AttrOr<float>(attrs, "inv_temperature", 1.0f);
  for (int l = 0; l <= lc; ++l) {
    int64_t width = 0;
    if (!Operand<float>(args, base + 1 + 2 * l, ffi::DataType::F32, &cp.kernel[l], &width) ||
        !Operand<float>(args, base + 2 + 2 * l, ffi::DataType::F32, &cp.bias[l]))
      return Invalid("critic operands must be f32");
    if (l < lc) cd.hidden[l] = (int32_t)width;
  }
  sb.obs = b.cond;
  size_t out = 0;
  if (!Output<float>(rets, out++, ffi::DataType::F32, &sb.best_action) || !Output<int32_t>(rets, out++, ffi::DataType::S32, &sb.best_idx))
    return Invalid("results: best_action (n_states, A) f32, best_idx (n_states,) s32");
  if (AttrOr<int32_t>(attrs, "want_q", 0) && !Output<float>(rets, out++, ffi::DataType::F32, &sb.q)) return Invalid("want_q = 1: result q (n_states, particles) f32");
  const size_t need = frl_sampler_select_workspace_bytes(&d, &cd);
  if (need == 0) return FromRc(frl_sampler_select_fwd(&d, &p, &s, &b, &cd, &cp, &sb, nullptr, 0, (void*)stream));   // says why
  auto ws = scratch.Allocate(need, 256);
  if (!ws.has_value()) return ffi::Error(ffi::ErrorCode::kResourceExhausted, "scratch allocation failed");
  return FromRc(frl_sampler_select_fwd(&d, &p, &s, &b, &cd, &cp, &sb, *ws, need, (void*)stream));
}

}  // namespace

XLA_FFI_DEFINE_HANDLER_SYMBOL(FrlSamplerSelectFwd, FrlSamplerSelectFwdImpl,
                              ffi::Ffi::Bind()
                                  .Ctx<ffi::PlatformStream<hipStream_t>>()
                                  .Ctx<ffi::ScratchAllocator>()
                                  .RemainingArgs()
                                  .RemainingRets()
                                  .Attrs());

XLA_FFI_DEFINE_HANDLER_SYMBOL(FrlSamplerFwd, FrlSamplerFwdImpl,
                              ffi::Ffi::Bind()
                                  .Ctx<ffi::PlatformStream<hipStream_t>>()
                                  .Ctx<ffi::ScratchAllocator>()
                                  .RemainingArgs()
                                  .RemainingRets()
                                  .Attrs());

XLA_FFI_DEFINE_HANDLER_SYMBOL(FrlSamplerFwdTrain, FrlSamplerFwdTrainImpl,
                              ffi::Ffi::Bind()
                                  .Ctx<ffi::PlatformStream<hipStream_t>>()
                                  .Ctx<ffi::ScratchAllocator>()
                                  .RemainingArgs()
                                  .RemainingRets()
                                  .Attrs());

XLA_FFI_DEFINE_HANDLER_SYMBOL(FrlSamplerBwd, FrlSamplerBwdImpl,
                              ffi::Ffi::Bind()
                                  .Ctx<ffi::PlatformStream<hipStream_t>>()
                                  .Ctx<ffi::ScratchAllocator>()
                                  .RemainingArgs()
                                  .RemainingRets()
                                  .Attrs());
