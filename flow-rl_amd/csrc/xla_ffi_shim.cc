// XLA-FFI (jax.ffi) registration shim: a thin translation of an XLA custom call onto the plain C ABI
// (include/flowrl_hip.h).  NOT compiled by build.py in this container: jaxlib's headers
// (xla/ffi/api/ffi.h, found via jax.ffi.include_dir()) are absent here and cannot be fetched, so this
// file has never been compile-checked — INTEGRATION.md says how a maintainer builds it:
//
//   hipcc -std=c++17 -fPIC -shared -I$(python -c "import jax; print(jax.ffi.include_dir())") \
//         -Iinclude flow-rl_amd/csrc/xla_ffi_shim.cc -Lflow-rl_amd/flowrl_hip -lflowrl_hip -o libflowrl_hip_ffi.so
//
// Operands (all F32, device): xT, cond, noise, betas, alphas, alpha_hats, postvars, ts, time_kernel,
// tmlp_k0, tmlp_b0, tmlp_k1, tmlp_b1, then (kernel_i, bias_i) for i = 0..L as remaining args.
// Results: x0 [, xt_hist, eps_hist].  Attrs mirror FrlDesc (the jit static_argnames).
#include <cstdint>
#include <cstring>

#include "flowrl_hip.h"
#include "xla/ffi/api/ffi.h"

namespace ffi = xla::ffi;

static ffi::Error FrlSamplerFwdImpl(hipStream_t stream, ffi::ScratchAllocator scratch,
                                    ffi::Buffer<ffi::F32> xT, ffi::Buffer<ffi::F32> cond, ffi::Buffer<ffi::F32> noise,
                                    ffi::Buffer<ffi::F32> betas, ffi::Buffer<ffi::F32> alphas,
                                    ffi::Buffer<ffi::F32> alpha_hats, ffi::Buffer<ffi::F32> postvars,
                                    ffi::Buffer<ffi::F32> ts, ffi::Buffer<ffi::F32> time_kernel,
                                    ffi::Buffer<ffi::F32> tk0, ffi::Buffer<ffi::F32> tb0,
                                    ffi::Buffer<ffi::F32> tk1, ffi::Buffer<ffi::F32> tb1,
                                    ffi::RemainingArgs layers, ffi::Result<ffi::Buffer<ffi::F32>> x0,
                                    ffi::RemainingRets hist, int32_t kind, int32_t solver, int32_t steps,
                                    int64_t n_states, int32_t particles, int32_t time_embed, int32_t time_dim,
                                    int32_t act, int32_t clip, float x_min, float x_max) {
  FrlDesc d;
  std::memset(&d, 0, sizeof(d));
  d.abi_version = FRL_ABI_VERSION;
  d.kind = kind; d.solver = solver; d.steps = steps; d.n_states = n_states; d.particles = particles;
  d.x_dim = (int32_t)xT.dimensions().back();
  d.cond_dim = cond.element_count() ? (int32_t)cond.dimensions().back() : 0;
  d.time_embed = time_embed; d.time_dim = time_dim;
  d.n_hidden = (int32_t)layers.size() / 2 - 1;
  if (d.n_hidden < 1 || d.n_hidden > FRL_MAX_HIDDEN) return ffi::Error(ffi::ErrorCode::kInvalidArgument, "bad layer count");
  FrlParams p;
  std::memset(&p, 0, sizeof(p));
  for (int l = 0; l <= d.n_hidden; ++l) {
    auto k = layers.get<ffi::Buffer<ffi::F32>>(2 * l);
    auto b = layers.get<ffi::Buffer<ffi::F32>>(2 * l + 1);
    if (!k.has_value() || !b.has_value()) return ffi::Error(ffi::ErrorCode::kInvalidArgument, "layer operands must be f32");
    p.kernel[l] = k->typed_data(); p.bias[l] = b->typed_data();
    if (l < d.n_hidden) d.hidden[l] = (int32_t)k->dimensions().back();
  }
  p.time_kernel = time_kernel.element_count() ? time_kernel.typed_data() : nullptr;
  p.tmlp_kernel[0] = tk0.typed_data(); p.tmlp_bias[0] = tb0.typed_data();
  p.tmlp_kernel[1] = tk1.typed_data(); p.tmlp_bias[1] = tb1.typed_data();
  d.act = act; d.clip = clip; d.x_min = x_min; d.x_max = x_max; d.dtype = FRL_DTYPE_F32;
  d.want_history = hist.size() == 2 ? 1 : 0;
  FrlSchedule s = {betas.typed_data(), alphas.typed_data(), alpha_hats.typed_data(), postvars.typed_data(),
                   ts.element_count() ? ts.typed_data() : nullptr};
  FrlFwdBufs b;
  std::memset(&b, 0, sizeof(b));
  b.xT = xT.typed_data();
  b.cond = d.cond_dim ? cond.typed_data() : nullptr;
  b.noise = noise.element_count() ? noise.typed_data() : nullptr;
  b.x0 = x0->typed_data();
  if (d.want_history) {
    b.xt_hist = hist.get<ffi::Buffer<ffi::F32>>(0).value()->typed_data();
    b.eps_hist = hist.get<ffi::Buffer<ffi::F32>>(1).value()->typed_data();
  }
  const size_t need = frl_workspace_bytes(&d);
  if (need == 0) return ffi::Error(ffi::ErrorCode::kInvalidArgument, frl_last_error());
  auto ws = scratch.Allocate(need, 256);
  if (!ws.has_value()) return ffi::Error(ffi::ErrorCode::kResourceExhausted, "scratch allocation failed");
  const int rc = frl_sampler_fwd(&d, &p, &s, &b, *ws, need, (void*)stream);
  if (rc != FRL_OK) return ffi::Error(rc == FRL_EUNSUPPORTED ? ffi::ErrorCode::kUnimplemented : ffi::ErrorCode::kInvalidArgument, frl_last_error());
  return ffi::Error::Success();
}

XLA_FFI_DEFINE_HANDLER_SYMBOL(
    FrlSamplerFwd, FrlSamplerFwdImpl,
    ffi::Ffi::Bind()
        .Ctx<ffi::PlatformStream<hipStream_t>>()
        .Ctx<ffi::ScratchAllocator>()
        .Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>()   // xT cond noise
        .Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>()   // betas alphas alpha_hats
        .Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>()   // postvars ts time_kernel
        .Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>().Arg<ffi::Buffer<ffi::F32>>()
        .RemainingArgs()
        .Ret<ffi::Buffer<ffi::F32>>()
        .RemainingRets()
        .Attr<int32_t>("kind").Attr<int32_t>("solver").Attr<int32_t>("steps").Attr<int64_t>("n_states")
        .Attr<int32_t>("particles").Attr<int32_t>("time_embed").Attr<int32_t>("time_dim").Attr<int32_t>("act")
        .Attr<int32_t>("clip").Attr<float>("x_min").Attr<float>("x_max"));
