"""Marshal one sampler call onto the C ABI (include/flowrl_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every byte of arithmetic happens in
libflowrl_hip.so.  There is no eager / CPU fallback — a missing library or an unsupported
configuration raises.
"""
from __future__ import annotations

import collections
import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from . import random as _random
from .module import LearnableFourierEmbedding, PositionalEmbedding, Simba, is_residual

_ws_cache: Dict[Tuple[int, int], torch.Tensor] = {}


def _workspace(nbytes: int, device: torch.device, stream: int = 0) -> torch.Tensor:
    """Scratch for one call, cached per (device, stream): calls enqueued on different streams never share it."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(stream))
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# Prepared workspaces (frl_sampler_prepare's output: coefficient / time-bias tables, packed edge layers) keyed by the
# descriptor and by (data_ptr, _version) of every parameter and schedule tensor: acting calls between two optimizer
# steps skip the prepare kernel (55..110 us, a fifth of an acting-sized call).  An entry owns its workspace and keeps
# its tensors alive, so a data_ptr cannot be recycled while cached; in-place torch updates bump _version and miss.
# Writes that bypass torch's version counter (p.data.copy_/lerp_ EMA updates, a foreign kernel through data_ptr) give a
# silent stale hit, so the cache is OPT-IN (cache_prepare=True) and clear_prepare_cache() is the explicit invalidation.
_prep_cache: "collections.OrderedDict" = collections.OrderedDict()
PREP_CACHE_ENTRIES = 8


def clear_prepare_cache() -> None:
    _prep_cache.clear()


def _f32c(t: torch.Tensor, device: torch.device, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: float32 required (got {t.dtype}); the fused path computes in f32")
    if t.device != device:
        raise ValueError(f"{what}: on {t.device}, expected {device}")
    return t if t.is_contiguous() else t.contiguous()


def _split_condition(xT: torch.Tensor, condition: Optional[torch.Tensor], particles: Optional[int]):
    """Flatten leading dims; recover (n_states, particles) so obs_repeat (dql.py:145, qvpo.py:33) is
    never materialised: a condition expanded with stride 0 over the particle axis, or an explicit
    ``particles=N`` with an un-repeated condition, is passed as (n_states, C)."""
    A = xT.shape[-1]
    B = xT.numel() // A if A > 0 else 0
    if condition is None:
        return B, 1, None
    Cdim = condition.shape[-1]
    if particles is not None and particles > 1:
        if condition.numel() // Cdim * particles != B:
            raise ValueError("particles does not match xT / condition leading dims")
        return B // particles, particles, condition.reshape(-1, Cdim)
    if condition.shape[:-1] != xT.shape[:-1]:
        raise ValueError(f"condition leading dims {tuple(condition.shape[:-1])} != xT {tuple(xT.shape[:-1])}")
    if condition.dim() >= 3 and condition.stride(-2) == 0 and condition.shape[-2] > 1:
        n = condition.shape[-2]
        base = condition.select(-2, 0)
        return B // n, n, base.reshape(-1, Cdim)
    return B, 1, condition.reshape(-1, Cdim)


def build_desc(network, *, kind: int, solver: int, steps: int, n_states: int, particles: int, x_dim: int,
               cond_dim: int, clip_sampler: bool, x_min, x_max, want_history: bool,
               compute_dtype: str = "f32", tile_policy: str = "auto", rng_impl: int = 0) -> _lib.FrlDesc:
    pred = network.predictor
    ce = network.cond_embedding
    if ce is not None:
        if (list(ce.hidden_dims) != [128, 128] or ce.output_dim or ce.layer_norm or (ce.dropout and ce.dropout > 0)
                or ce.act_name != pred.act_name):
            raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED,
                                            "cond_embedding must be MLP(hidden_dims=(128, 128), activation=<predictor's>) as in the reference agents")
    if pred.dropout and pred.dropout > 0:
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "dropout>0 backbones are outside the fused path")
    if len(pred.hidden_dims) < 1 or len(pred.hidden_dims) > _lib.FRL_MAX_HIDDEN:
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, f"{len(pred.hidden_dims)} hidden layers")
    d = _lib.FrlDesc()
    d.abi_version = _lib.FRL_ABI_VERSION
    d.kind, d.solver, d.steps = kind, solver, steps
    d.n_states, d.particles = n_states, particles
    d.x_dim = x_dim
    if ce is not None:
        d.cond_embed, d.cond_in_dim, d.cond_dim = 1, cond_dim, 128
    else:
        d.cond_embed, d.cond_in_dim, d.cond_dim = 0, 0, cond_dim
    te = network.time_embedding
    if te is None:
        d.time_embed, d.time_dim = _lib.FRL_TEMB_NONE, 0
    elif isinstance(te, PositionalEmbedding):
        if te.max_positions != 10000:
            raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "PositionalEmbedding.max_positions != 10000")
        d.time_embed, d.time_dim = _lib.FRL_TEMB_POSITIONAL, te.output_dim
    elif isinstance(te, LearnableFourierEmbedding):
        d.time_embed, d.time_dim = _lib.FRL_TEMB_FOURIER, te.output_dim
    else:
        raise TypeError(f"unknown time embedding {type(te)}")
    d.n_hidden = len(pred.hidden_dims)
    for i, h in enumerate(pred.hidden_dims):
        d.hidden[i] = int(h)
    d.act = _lib.FRL_ACT[pred.act_name]
    d.clip = 1 if clip_sampler else 0
    # jnp.clip(x, None, None) leaves a None side unbounded (ddpm.py:200-201, cnf.py:109-110)
    d.x_min = float(x_min) if x_min is not None else float("-inf")
    d.x_max = float(x_max) if x_max is not None else float("inf")
    if compute_dtype not in ("f32", "bf16"):
        raise ValueError(f"compute_dtype {compute_dtype!r}: 'f32' (bit-faithful) or 'bf16' (stated-tolerance MFMA variant)")
    d.dtype = _lib.FRL_DTYPE_BF16 if compute_dtype == "bf16" else _lib.FRL_DTYPE_F32
    d.want_history = 1 if want_history else 0
    if tile_policy not in _lib.FRL_TILE:
        raise ValueError(f"tile_policy {tile_policy!r}: 'auto', 'throughput' (one wave per tile), 'latency' (four waves per tile) or 'latency32' / 'latency16' (that build on 32- / 16-chain tiles)")
    d.tile_policy = _lib.FRL_TILE[tile_policy]
    d.rng_impl = int(rng_impl)
    # Dense -> LayerNorm -> act (mlp.py:26-29), or FQL's Dense -> act -> LayerNorm (fql/network.py:26-30)
    d.layer_norm = (2 if getattr(pred, "ln_after_act", False) else 1) if pred.layer_norm else 0
    if is_residual(pred):
        if len(set(int(h) for h in pred.hidden_dims)) != 1:
            raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "residual backbones: equal block widths only (no projection Dense)")
        d.backbone = _lib.FRL_BACKBONE_SIMBA if isinstance(pred, Simba) else _lib.FRL_BACKBONE_RESIDUAL
        d.res_multiplier = int(pred.multiplier)
    return d


def build_params(network, params: Dict, device: torch.device, keep: list) -> _lib.FrlParams:
    """Point FrlParams at the leaves of the Flax-layout tree (no copies for contiguous f32 leaves)."""
    fp = _lib.FrlParams()
    te = network.time_embedding

    def leaf(path):
        node = params
        for k in path:
            if k not in node:
                raise KeyError("params tree is missing " + "/".join(path))
            node = node[k]
        t = _f32c(node, device, "/".join(path))
        keep.append(t)
        return t.data_ptr()

    if te is not None:
        for i in range(2):
            fp.tmlp_kernel[i] = leaf(("MLP_0", f"Dense_{i}", "kernel"))
            fp.tmlp_bias[i] = leaf(("MLP_0", f"Dense_{i}", "bias"))
        if isinstance(te, LearnableFourierEmbedding):
            fp.time_kernel = leaf(("time_embedding", "kernel"))
    n = len(network.predictor.hidden_dims)
    pred, pn = network.predictor, network.predictor_name
    if is_residual(pred):
        for i in range(2):                # Dense_0: input projection, Dense_1: output layer (mlp.py:84,97; simba.py:49,62)
            fp.kernel[i] = leaf((pn, f"Dense_{i}", "kernel"))
            fp.bias[i] = leaf((pn, f"Dense_{i}", "bias"))
        for i in range(n):
            blk = f"{pred.block_name}_{i}"
            for k in range(2):
                fp.res_kernel[i][k] = leaf((pn, blk, f"Dense_{k}", "kernel"))
                fp.res_bias[i][k] = leaf((pn, blk, f"Dense_{k}", "bias"))
            if pred.layer_norm:
                fp.res_ln_scale[i] = leaf((pn, blk, "LayerNorm_0", "scale"))
                fp.res_ln_bias[i] = leaf((pn, blk, "LayerNorm_0", "bias"))
        if isinstance(pred, Simba):
            fp.final_ln_scale = leaf((pn, "LayerNorm_0", "scale"))
            fp.final_ln_bias = leaf((pn, "LayerNorm_0", "bias"))
        if network.cond_embedding is not None:
            for i in range(2):
                fp.cond_kernel[i] = leaf(("cond_embedding", f"Dense_{i}", "kernel"))
                fp.cond_bias[i] = leaf(("cond_embedding", f"Dense_{i}", "bias"))
        return fp
    for i in range(n + 1):
        fp.kernel[i] = leaf((network.predictor_name, f"Dense_{i}", "kernel"))
        fp.bias[i] = leaf((network.predictor_name, f"Dense_{i}", "bias"))
    if network.predictor.layer_norm:      # flax auto-names the LayerNorm of hidden layer i "LayerNorm_i"
        for i in range(n):
            fp.ln_scale[i] = leaf((network.predictor_name, f"LayerNorm_{i}", "scale"))
            fp.ln_bias[i] = leaf((network.predictor_name, f"LayerNorm_{i}", "bias"))
    if network.cond_embedding is not None:
        for i in range(2):
            fp.cond_kernel[i] = leaf(("cond_embedding", f"Dense_{i}", "kernel"))
            fp.cond_bias[i] = leaf(("cond_embedding", f"Dense_{i}", "bias"))
    return fp


def check_param_shapes(network, params: Dict, x_dim: int, cond_dim: int) -> None:
    te = network.time_embedding
    d_time = 1 if te is None else te.output_dim
    if network.cond_embedding is not None:
        cdims = [cond_dim, *network.cond_embedding.hidden_dims]
        for i in range(len(cdims) - 1):
            k = params["cond_embedding"][f"Dense_{i}"]["kernel"]
            if tuple(k.shape) != (cdims[i], cdims[i + 1]):
                raise ValueError(f"cond_embedding/Dense_{i}/kernel {tuple(k.shape)}, expected {(cdims[i], cdims[i + 1])}")
        cond_dim = cdims[-1]
    pp = params[network.predictor_name]
    pred = network.predictor
    if is_residual(pred):
        h, m = int(pred.hidden_dims[0]), int(pred.multiplier)

        def want(node, name, ks, bs):
            k, b = node[name]["kernel"], node[name]["bias"]
            if tuple(k.shape) != ks or tuple(b.shape) != bs:
                raise ValueError(f"{network.predictor_name}/.../{name}: kernel {tuple(k.shape)} bias {tuple(b.shape)}, expected {ks} / {bs}")

        want(pp, "Dense_0", (x_dim + d_time + cond_dim, h), (h,))
        want(pp, "Dense_1", (h, x_dim), (x_dim,))
        for i in range(len(pred.hidden_dims)):
            blk = pp.get(f"{pred.block_name}_{i}")
            if blk is None:
                raise ValueError(f"{network.predictor_name}/{pred.block_name}_{i} missing from the params tree")
            want(blk, "Dense_0", (h, m * h), (m * h,))
            want(blk, "Dense_1", (m * h, h), (h,))
        dims = []
    else:
        dims = [x_dim + d_time + cond_dim, *network.predictor.hidden_dims, x_dim]
    for i in range(len(dims) - 1):
        k, b = pp[f"Dense_{i}"]["kernel"], pp[f"Dense_{i}"]["bias"]
        if tuple(k.shape) != (dims[i], dims[i + 1]) or tuple(b.shape) != (dims[i + 1],):
            raise ValueError(f"{network.predictor_name}/Dense_{i}: kernel {tuple(k.shape)} bias {tuple(b.shape)}, "
                             f"expected {(dims[i], dims[i + 1])} / {(dims[i + 1],)} (input order x | time | cond)")
    if network.predictor.layer_norm and not is_residual(pred):
        for i, h in enumerate(network.predictor.hidden_dims):
            ln = pp.get(f"LayerNorm_{i}")
            if ln is None or tuple(ln["scale"].shape) != (h,) or tuple(ln["bias"].shape) != (h,):
                raise ValueError(f"{network.predictor_name}/LayerNorm_{i}: scale / bias of shape ({h},) expected (MLP(layer_norm=True))")
    if te is not None:
        for i in range(2):
            k = params["MLP_0"][f"Dense_{i}"]["kernel"]
            if tuple(k.shape) != (d_time, d_time):
                raise ValueError(f"MLP_0/Dense_{i}/kernel {tuple(k.shape)}, expected {(d_time, d_time)}")
        if isinstance(te, LearnableFourierEmbedding):
            k = params["time_embedding"]["kernel"]
            if tuple(k.shape) != (d_time // 2, 1):
                raise ValueError(f"time_embedding/kernel {tuple(k.shape)}, expected {(d_time // 2, 1)}")


def _marshal(network, params, *, kind, solver, steps, xT, condition, sched, clip_sampler, x_min, x_max, history,
             particles, compute_dtype="f32", tile_policy="auto", rng_impl=0):
    """Descriptor, parameter pointers and schedule pointers of one call (shared by forward and VJP)."""
    device = xT.device
    A = xT.shape[-1]
    n_states, parts, cond2d = _split_condition(xT, condition, particles)
    cond_dim = 0 if cond2d is None else cond2d.shape[-1]
    check_param_shapes(network, params, A, cond_dim)
    keep: list = []
    desc = build_desc(network, kind=kind, solver=solver, steps=steps, n_states=n_states, particles=parts, x_dim=A,
                      cond_dim=cond_dim, clip_sampler=clip_sampler, x_min=x_min, x_max=x_max, want_history=history,
                      compute_dtype=compute_dtype, tile_policy=tile_policy, rng_impl=rng_impl)
    fparams = build_params(network, params, device, keep)
    fs = _lib.FrlSchedule()
    for name in ("betas", "alphas", "alpha_hats", "postvars", "ts", "mid_ts", "mid_alphas", "mid_sigmas"):
        t = sched.get(name)
        if t is not None:
            t = _f32c(t, device, f"sched.{name}")
            want = steps if name.startswith("mid_") else steps + 1
            if t.numel() != want:
                raise ValueError(f"sched.{name} has {t.numel()} entries, expected {want}")
            keep.append(t)
            setattr(fs, name, t.data_ptr())
    if cond2d is not None:
        cond2d = _f32c(cond2d, device, "condition")
    return desc, fparams, fs, keep, n_states * parts, cond2d


def run_sampler(network, params: Dict, *, kind: int, solver: int, steps: int, xT: torch.Tensor,
                condition: Optional[torch.Tensor], noise: Optional[torch.Tensor], sched: Dict[str, torch.Tensor],
                clip_sampler: bool, x_min, x_max, history: bool, particles: Optional[int] = None,
                compute_dtype: str = "f32", tile_policy: str = "auto", rng_key: Optional[torch.Tensor] = None,
                rng_impl: Optional[str] = None, cache_prepare: bool = False, rng_shard: Optional[Tuple[int, int]] = None,
                static_scratch: bool = False, select=None):
    """``select`` (a :class:`flowrl_hip.critic.FusedSelect`): critic scoring and best-of-N selection run inside the
    sampler's launch (frl_sampler_select_fwd); its ``best`` / ``idx`` / ``q`` fields receive the outputs and x0 is only
    materialised when it asks for the candidates.
    ``rng_shard`` = (first chain, total chains) when this call is one state-shard of a larger batch: the in-kernel
    draws are then the elements of the UNSHARDED normal(key_j, (total, A)), so N ranks reproduce one big call.
    Returns (x0, (xt_hist, eps_hist) or (None, None)) with xT's leading dims restored — and, when ``rng_key`` (a (2,)
    uint32 threefry key, flowrl_hip.random) is given, a third element: the rng the reference's sample() returns; the
    per-step noise is then drawn in-kernel unless ``noise`` is passed.  ``compute_dtype`` = "bf16"
    selects the bf16-MFMA variant (bf16 operands, f32 accumulate / chain state; stated tolerance, not bit-faithful).
    ``cache_prepare`` (opt-in) keeps the prepared tables of a call keyed by (data_ptr, _version) of every leaf: only safe
    when every parameter update goes through torch's version counter (``p.data.copy_()``-style EMA updates and foreign
    kernels do NOT) — call :func:`clear_prepare_cache` after such an update, or leave it off.
    ``static_scratch`` routes the call through ``frl_sampler_fwd_static``: the workspace is the library's file-scope
    device array (no allocation inside the entry point; calls overlapping on one GPU race; small n_states only)."""
    lib = _lib.load()
    if static_scratch and cache_prepare:
        raise ValueError("static_scratch and cache_prepare are exclusive (the static buffer is rewritten by every call)")
    if not xT.is_cuda:
        raise RuntimeError("flowrl_hip samplers run on the GPU only (xT must be a CUDA/HIP tensor); there is no CPU fallback")
    device = xT.device
    A = xT.shape[-1]
    lead = tuple(xT.shape[:-1])
    desc, fparams, fs, keep, B, cond2d = _marshal(
        network, params, kind=kind, solver=solver, steps=steps, xT=xT, condition=condition, sched=sched,
        clip_sampler=clip_sampler, x_min=x_min, x_max=x_max, history=history, particles=particles,
        compute_dtype=compute_dtype, tile_policy=tile_policy,
        rng_impl=0 if rng_key is None else _random.impl_id(rng_impl))

    x_in = _f32c(xT.reshape(B, A), device, "xT")
    bufs = _lib.FrlFwdBufs()
    if select is not None:
        if static_scratch or cache_prepare or history or rng_key is not None:
            raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "fused sample+select: no static_scratch / cache_prepare / history / in-kernel rng")
        x0 = torch.empty((B, A), dtype=torch.float32, device=device) if select.want_actions else None
    else:
        x0 = torch.empty((B, A), dtype=torch.float32, device=device)
    bufs.xT = x_in.data_ptr()
    if x0 is not None:
        bufs.x0 = x0.data_ptr()
    if cond2d is not None:
        bufs.cond = cond2d.data_ptr()
    if noise is not None:
        if noise.numel() != steps * B * A:
            raise ValueError(f"noise has {noise.numel()} elements, expected (T, ..., A) = {steps * B * A}")
        noise = _f32c(noise.reshape(steps, B, A), device, "noise")
        bufs.noise = noise.data_ptr()
    rng_out = None
    if select is not None:
        cdesc, cparams, sbufs, ckeep = select.marshal(int(desc.n_states), int(desc.particles), A, device)
        need = lib.frl_sampler_select_workspace_bytes(C.byref(desc), C.byref(cdesc))
        stream = torch.cuda.current_stream(device).cuda_stream
        ws = _workspace(max(need, 256), device, stream)
        with torch.cuda.device(device):      # need == 0: the entry point says why (raises through check)
            rc = lib.frl_sampler_select_fwd(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs), C.byref(cdesc), C.byref(cparams),
                                            C.byref(sbufs), ws.data_ptr() if need else None, ws.numel() if need else 0, stream)
        _lib.check(rc)
        if need == 0:
            raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "fused sample+select: configuration has no kernel")
        del ckeep
        return (x0.reshape(*lead, A) if x0 is not None else None), (None, None)
    if rng_key is not None:
        rng_key = _random.as_key(rng_key, device)
        rng_out = torch.empty((2,), dtype=torch.uint32, device=device)
        bufs.rng_key, bufs.rng_out = rng_key.data_ptr(), rng_out.data_ptr()
        if rng_shard is not None:
            bufs.rng_chain_offset, bufs.rng_total_chains = int(rng_shard[0]), int(rng_shard[1])
    xt_hist = eps_hist = None
    if history:
        xt_hist = torch.empty((steps, B, A), dtype=torch.float32, device=device)
        eps_hist = torch.empty((steps, B, A), dtype=torch.float32, device=device)
        bufs.xt_hist, bufs.eps_hist = xt_hist.data_ptr(), eps_hist.data_ptr()

    _lib.check(lib.frl_validate(C.byref(desc)))
    need = lib.frl_workspace_bytes(C.byref(desc))
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        if static_scratch:
            rc = lib.frl_sampler_fwd_static(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs), stream)
        elif cache_prepare:
            key = (bytes(desc), device.index, int(stream), tuple((t.data_ptr(), t._version) for t in keep))
            ent = _prep_cache.get(key)
            if ent is None:
                ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
                rc = lib.frl_sampler_fwd(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs),
                                         ws.data_ptr(), ws.numel(), stream)
                if rc == _lib.FRL_OK:
                    _prep_cache[key] = (ws, keep)
                    while len(_prep_cache) > PREP_CACHE_ENTRIES:
                        _prep_cache.popitem(last=False)
            else:
                _prep_cache.move_to_end(key)
                ws = ent[0]
                rc = lib.frl_sampler_run(C.byref(desc), C.byref(fparams), C.byref(bufs), ws.data_ptr(), ws.numel(), stream)
        else:
            ws = _workspace(need, device, stream)
            rc = lib.frl_sampler_fwd(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs),
                                     ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc)
    x0 = x0.reshape(*lead, A)
    hist = (xt_hist.reshape(steps, *lead, A), eps_hist.reshape(steps, *lead, A)) if history else (None, None)
    if rng_key is not None:
        return x0, hist, rng_out
    return x0, hist


# ---- training forward (keeps what the chain VJP needs) --------------------------------------------------------
class VjpContext:
    """What a training forward (frl_sampler_fwd_train) leaves for the backward: the descriptor / parameter pointers of
    the call, its own workspace holding the stash (act(z), act'(z) of every layer and step), and the history."""
    __slots__ = ("desc", "fparams", "fs", "keep", "ws", "xt_hist", "eps_hist", "B", "cond2d", "lead")


def run_sampler_train(network, params: Dict, *, kind: int, solver: int, steps: int, xT: torch.Tensor,
                      condition: Optional[torch.Tensor], noise: Optional[torch.Tensor], sched: Dict[str, torch.Tensor],
                      clip_sampler: bool, x_min, x_max, particles: Optional[int] = None, rng_key: Optional[torch.Tensor] = None,
                      rng_impl: Optional[str] = None, rng_shard: Optional[Tuple[int, int]] = None, compute_dtype: str = "f32"):
    """sample(training=True, params=p) for a caller that will differentiate through the chain (dql.py:105-112): one
    launch of the training-forward kernel, which is the sampler plus the stash the backward sweep reads instead of
    recomputing.  Returns (x0, (xt_hist, eps_hist), ctx[, rng_out]); pass ``ctx`` to :func:`run_sampler_vjp`."""
    lib = _lib.load()
    if not xT.is_cuda:
        raise RuntimeError("flowrl_hip samplers run on the GPU only; there is no CPU fallback")
    device = xT.device
    A = xT.shape[-1]
    lead = tuple(xT.shape[:-1])
    desc, fparams, fs, keep, B, cond2d = _marshal(
        network, params, kind=kind, solver=solver, steps=steps, xT=xT, condition=condition, sched=sched,
        clip_sampler=clip_sampler, x_min=x_min, x_max=x_max, history=True, particles=particles, compute_dtype=compute_dtype,
        rng_impl=0 if rng_key is None else _random.impl_id(rng_impl))
    need = lib.frl_bwd_workspace_bytes(C.byref(desc))
    if need == 0:
        _lib.check(lib.frl_sampler_fwd_train(C.byref(desc), C.byref(fparams), C.byref(fs), None, None, 0, None))
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "this configuration has no VJP kernel")
    x_in = _f32c(xT.reshape(B, A), device, "xT")
    x0 = torch.empty((B, A), dtype=torch.float32, device=device)
    xt_hist = torch.empty((steps, B, A), dtype=torch.float32, device=device)
    eps_hist = torch.empty((steps, B, A), dtype=torch.float32, device=device)
    bufs = _lib.FrlFwdBufs()
    bufs.xT, bufs.x0, bufs.xt_hist, bufs.eps_hist = x_in.data_ptr(), x0.data_ptr(), xt_hist.data_ptr(), eps_hist.data_ptr()
    if cond2d is not None:
        bufs.cond = cond2d.data_ptr()
    if noise is not None:
        if noise.numel() != steps * B * A:
            raise ValueError(f"noise has {noise.numel()} elements, expected (T, ..., A) = {steps * B * A}")
        noise = _f32c(noise.reshape(steps, B, A), device, "noise")
        bufs.noise = noise.data_ptr()
    rng_out = None
    if rng_key is not None:
        rng_key = _random.as_key(rng_key, device)
        rng_out = torch.empty((2,), dtype=torch.uint32, device=device)
        bufs.rng_key, bufs.rng_out = rng_key.data_ptr(), rng_out.data_ptr()
        if rng_shard is not None:
            bufs.rng_chain_offset, bufs.rng_total_chains = int(rng_shard[0]), int(rng_shard[1])
    # the context owns its workspace: the stash must survive whatever runs between forward and backward (the critic)
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        rc = lib.frl_sampler_fwd_train(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs), ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc)
    ctx = VjpContext()
    ctx.desc, ctx.fparams, ctx.fs, ctx.keep, ctx.ws = desc, fparams, fs, keep, ws
    ctx.xt_hist, ctx.eps_hist, ctx.B, ctx.cond2d, ctx.lead = xt_hist, eps_hist, B, cond2d, lead
    hist = (xt_hist.reshape(steps, *lead, A), eps_hist.reshape(steps, *lead, A))
    if rng_key is not None:
        return x0.reshape(*lead, A), hist, ctx, rng_out
    return x0.reshape(*lead, A), hist, ctx


class GradTree(dict):
    """A gradient tree (Flax layout) whose leaves are views of ``flat``, one f32 buffer (leaves start at multiples of 64
    floats; the gaps are zero)."""
    flat: Optional[torch.Tensor] = None


def alloc_grad_tree(params: Dict, paths, device: torch.device) -> "GradTree":
    """Contiguous f32 leaves with the shapes of ``params`` at ``paths``, carved out of one zero-filled flat buffer."""
    shapes = []
    for path in paths:
        node = params
        for k in path:
            node = node[k]
        shapes.append(tuple(node.shape))
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    offs, o = [], 0
    for nel in sizes:
        offs.append(o)
        o += (nel + 63) & ~63
    flat = torch.zeros(max(o, 1), dtype=torch.float32, device=device)
    tree = GradTree()
    for path, sh, nel, off in zip(paths, shapes, sizes, offs):
        node = tree
        for k in path[:-1]:
            node = node.setdefault(k, {})
        node[path[-1]] = flat[off:off + nel].view(sh)
    tree.flat = flat
    return tree


# ---- chain VJP -----------------------------------------------------------------------------------------------
def run_sampler_vjp(network, params: Dict, *, kind: int, solver: int, steps: int, xt_hist: torch.Tensor,
                    condition: Optional[torch.Tensor], g_x0: torch.Tensor, sched: Dict[str, torch.Tensor],
                    tvals: Optional[torch.Tensor] = None, clip_sampler: bool = False, x_min=None, x_max=None, particles: Optional[int] = None,
                    want_gxT: bool = True, compute_dtype: str = "f32", ctx: Optional["VjpContext"] = None):
    """``ctx`` (from :func:`run_sampler_train`): the backward reads that call's stash and eps_hist instead of recomputing
    every step from xt_hist; descriptor, parameter pointers and workspace are the forward call's.
    dL/dparams (Flax-layout tree) and dL/dxT of one sampler call, given its history[0] and dL/dx0.  Everything is
    libflowrl_hip's frl_sampler_bwd: recompute + backward sweep + weight-gradient GEMMs (O(B)), and the batch-invariant
    time branch (MLP_0, time_embedding, Dense_0's bias and time rows) from the kernel's own dL/dtb."""
    lib = _lib.load()
    if not g_x0.is_cuda:
        raise RuntimeError("flowrl_hip VJP runs on the GPU only; there is no CPU fallback")
    device = g_x0.device
    A = g_x0.shape[-1]
    lead = tuple(g_x0.shape[:-1])
    if ctx is not None:
        desc, fparams, fs, keep, B, cond2d = ctx.desc, ctx.fparams, ctx.fs, ctx.keep, ctx.B, ctx.cond2d
        xt_hist = ctx.xt_hist
    else:
        desc, fparams, fs, keep, B, cond2d = _marshal(
            network, params, kind=kind, solver=solver, steps=steps, xT=g_x0, condition=condition, sched=sched,
            clip_sampler=clip_sampler, x_min=x_min, x_max=x_max, history=True, particles=particles,
            compute_dtype=compute_dtype)
    if g_x0.numel() != B * A:
        raise ValueError(f"g_x0 has {g_x0.numel()} elements, expected {B * A}")
    if xt_hist.numel() != steps * B * A:
        raise ValueError(f"xt_hist has {xt_hist.numel()} elements, expected (T, ..., A) = {steps * B * A}")
    xt_hist = _f32c(xt_hist.reshape(steps, B, A), device, "xt_hist")
    g_in = _f32c(g_x0.reshape(B, A), device, "g_x0")
    pname = network.predictor_name
    n = len(network.predictor.hidden_dims)
    H = int(network.predictor.hidden_dims[0])
    te = network.time_embedding
    # Every gradient leaf is a view of ONE flat f32 buffer (GradTree.flat): the multi-GPU all-reduce of the gradients is
    # then a single collective on that buffer, with no concatenation or copy-back kernels (parallel.all_reduce_grads).
    paths = [(pname, f"Dense_{i}", w) for i in range(n + 1) for w in ("kernel", "bias")]
    if te is not None:      # the library finishes the time branch: MLP_0 leaves (+ the Fourier kernel)
        paths += [("MLP_0", f"Dense_{i}", w) for i in range(2) for w in ("kernel", "bias")]
        if isinstance(te, LearnableFourierEmbedding):
            paths.append(("time_embedding", "kernel"))
    if network.cond_embedding is not None:      # DACER differentiates the embedding too (dacer.py:141-162)
        paths += [("cond_embedding", f"Dense_{i}", w) for i in range(2) for w in ("kernel", "bias")]
    grads = alloc_grad_tree(params, paths, device)
    bufs = _lib.FrlBwdBufs()
    bufs.xt_hist, bufs.g_x0 = xt_hist.data_ptr(), g_in.data_ptr()
    if cond2d is not None:
        bufs.cond = cond2d.data_ptr()
    g_xT = torch.empty((B, A), dtype=torch.float32, device=device) if want_gxT else None
    if g_xT is not None:
        bufs.g_xT = g_xT.data_ptr()
    for i in range(n + 1):
        g = grads[pname][f"Dense_{i}"]
        bufs.grad_kernel[i], bufs.grad_bias[i] = g["kernel"].data_ptr(), g["bias"].data_ptr()
    grad_tb = torch.empty((steps, H), dtype=torch.float32, device=device)
    bufs.grad_tb = grad_tb.data_ptr()
    if te is not None:
        for i in range(2):
            g = grads["MLP_0"][f"Dense_{i}"]
            bufs.grad_tmlp_kernel[i], bufs.grad_tmlp_bias[i] = g["kernel"].data_ptr(), g["bias"].data_ptr()
        if isinstance(te, LearnableFourierEmbedding):
            bufs.grad_time_kernel = grads["time_embedding"]["kernel"].data_ptr()
    if network.cond_embedding is not None:
        for i in range(2):
            g = grads["cond_embedding"][f"Dense_{i}"]
            bufs.grad_cond_kernel[i], bufs.grad_cond_bias[i] = g["kernel"].data_ptr(), g["bias"].data_ptr()

    need = lib.frl_bwd_workspace_bytes(C.byref(desc))
    if need == 0:   # invalid / unsupported descriptor: let the entry point say why
        _lib.check(lib.frl_sampler_bwd(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs), None, 0, None))
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "this configuration has no VJP kernel")
    stream = torch.cuda.current_stream(device).cuda_stream
    if ctx is not None:
        ws = ctx.ws
        bufs.eps_hist, bufs.have_stash = ctx.eps_hist.data_ptr(), 1
    else:
        ws = _workspace(need, device, stream)
    with torch.cuda.device(device):
        rc = lib.frl_sampler_bwd(C.byref(desc), C.byref(fparams), C.byref(fs), C.byref(bufs),
                                 ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc)
    return grads, (g_xT.reshape(*lead, A) if g_xT is not None else None)
