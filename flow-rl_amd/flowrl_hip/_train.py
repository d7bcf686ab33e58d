"""Host side of the training-side half of the flow models: the per-sample-time network
evaluation and the forward-process pieces, marshalled onto the C ABI (include/flowrl_hip.h: frl_net_eval, frl_add_noise,
frl_ddpm_onestep_update, frl_mse_loss, frl_random_uniform / randint).  As everywhere in this package PyTorch only owns
memory and streams; there is no eager / CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from . import random as _random
from .module import LearnableFourierEmbedding, Simba, is_residual
from ._sampler import _f32c, _split_condition, _workspace, alloc_grad_tree, build_desc, build_params, check_param_shapes


def net_eval(network, params: Dict, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, *,
             particles: Optional[int] = None, tile_policy: str = "auto") -> torch.Tensor:
    """``self(x, t, condition)`` of a DDPMBackbone / FlowBackbone model with a per-sample time input
    (ddpm.py:249-254, 286-293; continuous_ddpm.py:234-241; cnf.py:155-162).  ``t``: (..., 1) or (...) matching x's leading
    dims — integer steps (DDPM) are promoted to float32 exactly as ``t_proto * t`` is by the embedding."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("flowrl_hip runs on the GPU only; there is no CPU fallback")
    device = x.device
    A = x.shape[-1]
    lead = tuple(x.shape[:-1])
    n_states, parts, cond2d = _split_condition(x, condition, particles)
    cond_dim = 0 if cond2d is None else cond2d.shape[-1]
    check_param_shapes(network, params, A, cond_dim)
    keep: list = []
    desc = build_desc(network, kind=_lib.FRL_KIND_CNF, solver=_lib.FRL_SOLVER_EULER, steps=1, n_states=n_states, particles=parts,
                      x_dim=A, cond_dim=cond_dim, clip_sampler=False, x_min=None, x_max=None, want_history=False,
                      tile_policy=tile_policy)
    fparams = build_params(network, params, device, keep)
    B = n_states * parts
    if t.numel() != B:
        raise ValueError(f"t has {t.numel()} elements, expected one per sample ({B})")
    tf = t.reshape(B).to(torch.float32)
    tf = _f32c(tf, device, "t")
    xin = _f32c(x.reshape(B, A), device, "x")
    out = torch.empty((B, A), dtype=torch.float32, device=device)
    bufs = _lib.FrlEvalBufs()
    bufs.x, bufs.t, bufs.out = xin.data_ptr(), tf.data_ptr(), out.data_ptr()
    if cond2d is not None:
        cond2d = _f32c(cond2d, device, "condition")
        bufs.cond = cond2d.data_ptr()
    need = lib.frl_net_eval_workspace_bytes(C.byref(desc))
    stream = torch.cuda.current_stream(device).cuda_stream
    if need == 0:
        _lib.check(lib.frl_net_eval(C.byref(desc), C.byref(fparams), C.byref(bufs), None, 0, stream))
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "this configuration has no evaluation kernel")
    ws = _workspace(need, device, stream)
    with torch.cuda.device(device):
        rc = lib.frl_net_eval(C.byref(desc), C.byref(fparams), C.byref(bufs), ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc)
    return out.reshape(*lead, A)


class EvalVjpContext:
    """What :func:`net_eval_train` leaves for :func:`net_eval_vjp`: descriptor / parameter pointers, the inputs, and the
    workspace holding the stash of the one evaluation."""
    __slots__ = ("network", "params", "desc", "fparams", "keep", "ws", "x", "t", "cond2d", "B", "A", "lead")


def net_eval_train(network, params: Dict, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, *,
                   particles: Optional[int] = None):
    """``model.apply({"params": p}, xt, t, condition, training=True)`` inside a loss that jax.grad differentiates
    (ddpm.py:286-296, continuous_ddpm.py:234-243, cnf.py:155-165): frl_net_eval_train = the evaluation plus the stash
    its backward reads.  Returns (out, ctx); pass ``ctx`` and dL/dout to :func:`net_eval_vjp`."""
    lib = _lib.load()
    if not x.is_cuda:
        raise RuntimeError("flowrl_hip runs on the GPU only; there is no CPU fallback")
    device = x.device
    A = x.shape[-1]
    lead = tuple(x.shape[:-1])
    n_states, parts, cond2d = _split_condition(x, condition, particles)
    cond_dim = 0 if cond2d is None else cond2d.shape[-1]
    check_param_shapes(network, params, A, cond_dim)
    keep: list = []
    desc = build_desc(network, kind=_lib.FRL_KIND_CNF, solver=_lib.FRL_SOLVER_EULER, steps=1, n_states=n_states, particles=parts,
                      x_dim=A, cond_dim=cond_dim, clip_sampler=False, x_min=None, x_max=None, want_history=False,
                      tile_policy="auto")
    fparams = build_params(network, params, device, keep)
    B = n_states * parts
    if t.numel() != B:
        raise ValueError(f"t has {t.numel()} elements, expected one per sample ({B})")
    tf = _f32c(t.reshape(B).to(torch.float32), device, "t")
    xin = _f32c(x.reshape(B, A), device, "x")
    out = torch.empty((B, A), dtype=torch.float32, device=device)
    bufs = _lib.FrlEvalBufs()
    bufs.x, bufs.t, bufs.out = xin.data_ptr(), tf.data_ptr(), out.data_ptr()
    if cond2d is not None:
        cond2d = _f32c(cond2d, device, "condition")
        bufs.cond = cond2d.data_ptr()
    need = lib.frl_net_eval_bwd_workspace_bytes(C.byref(desc))
    stream = torch.cuda.current_stream(device).cuda_stream
    if need == 0:
        _lib.check(lib.frl_net_eval_train(C.byref(desc), C.byref(fparams), C.byref(bufs), None, 0, stream))
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "this configuration has no single-evaluation VJP kernel")
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=device)      # owned by the context: the stash must survive
    with torch.cuda.device(device):
        rc = lib.frl_net_eval_train(C.byref(desc), C.byref(fparams), C.byref(bufs), ws.data_ptr(), ws.numel(), stream)
    _lib.check(rc)
    ctx = EvalVjpContext()
    ctx.network, ctx.params, ctx.desc, ctx.fparams, ctx.keep, ctx.ws = network, params, desc, fparams, keep, ws
    ctx.x, ctx.t, ctx.cond2d, ctx.B, ctx.A, ctx.lead = xin, tf, cond2d, B, A, lead
    return out.reshape(*lead, A), ctx


def _grad_like(leaf: Dict):
    gk = torch.empty_like(leaf["kernel"], dtype=torch.float32, memory_format=torch.contiguous_format)
    gb = torch.empty_like(leaf["bias"], dtype=torch.float32, memory_format=torch.contiguous_format)
    return gk, gb


def net_eval_vjp(ctx: EvalVjpContext, g_out: torch.Tensor, want_gx: bool = False):
    """dL/dparams (Flax-layout tree, every leaf of the model) and optionally dL/dx of the evaluation ``ctx`` came from,
    given dL/dout: frl_net_eval_bwd (backward sweep from the stash, weight-gradient GEMMs with a fixed-order reduce,
    per-sample time branch and cond_embedding backward).  Deterministic."""
    lib = _lib.load()
    device = ctx.x.device
    network, params, B, A = ctx.network, ctx.params, ctx.B, ctx.A
    if g_out.numel() != B * A:
        raise ValueError(f"g_out has {g_out.numel()} elements, expected {B * A}")
    g_in = _f32c(g_out.reshape(B, A), device, "g_out")
    bufs = _lib.FrlEvalBwdBufs()
    bufs.x, bufs.t, bufs.g_out = ctx.x.data_ptr(), ctx.t.data_ptr(), g_in.data_ptr()
    if ctx.cond2d is not None:
        bufs.cond = ctx.cond2d.data_ptr()
    g_x = torch.empty((B, A), dtype=torch.float32, device=device) if want_gx else None
    if g_x is not None:
        bufs.g_x = g_x.data_ptr()
    pname = network.predictor_name
    pred = network.predictor
    n = len(pred.hidden_dims)
    res = is_residual(pred)
    te = network.time_embedding
    # every gradient leaf is a view of one flat f32 buffer (_sampler.GradTree), as in the chain VJP
    if res:
        paths = [(pname, f"Dense_{i}", w) for i in range(2) for w in ("kernel", "bias")]
        for i in range(n):
            blk = f"{pred.block_name}_{i}"
            paths += [(pname, blk, f"Dense_{k}", w) for k in range(2) for w in ("kernel", "bias")]
            if pred.layer_norm:
                paths += [(pname, blk, "LayerNorm_0", w) for w in ("scale", "bias")]
        if isinstance(pred, Simba):
            paths += [(pname, "LayerNorm_0", w) for w in ("scale", "bias")]
    else:
        paths = [(pname, f"Dense_{i}", w) for i in range(n + 1) for w in ("kernel", "bias")]
        if pred.layer_norm:
            paths += [(pname, f"LayerNorm_{i}", w) for i in range(n) for w in ("scale", "bias")]
    if te is not None:
        paths += [("MLP_0", f"Dense_{i}", w) for i in range(2) for w in ("kernel", "bias")]
        if isinstance(te, LearnableFourierEmbedding):
            paths.append(("time_embedding", "kernel"))
    if network.cond_embedding is not None:
        paths += [("cond_embedding", f"Dense_{i}", w) for i in range(2) for w in ("kernel", "bias")]
    grads = alloc_grad_tree(params, paths, device)
    gp = grads[pname]
    for i in range(2 if res else n + 1):
        bufs.grad_kernel[i], bufs.grad_bias[i] = gp[f"Dense_{i}"]["kernel"].data_ptr(), gp[f"Dense_{i}"]["bias"].data_ptr()
    if res:
        for i in range(n):
            blk = gp[f"{pred.block_name}_{i}"]
            for k in range(2):
                bufs.grad_res_kernel[i][k], bufs.grad_res_bias[i][k] = blk[f"Dense_{k}"]["kernel"].data_ptr(), blk[f"Dense_{k}"]["bias"].data_ptr()
            if pred.layer_norm:
                bufs.grad_res_ln_scale[i], bufs.grad_res_ln_bias[i] = blk["LayerNorm_0"]["scale"].data_ptr(), blk["LayerNorm_0"]["bias"].data_ptr()
        if isinstance(pred, Simba):
            bufs.grad_final_ln_scale, bufs.grad_final_ln_bias = gp["LayerNorm_0"]["scale"].data_ptr(), gp["LayerNorm_0"]["bias"].data_ptr()
    elif pred.layer_norm:
        for i in range(n):
            bufs.grad_ln_scale[i], bufs.grad_ln_bias[i] = gp[f"LayerNorm_{i}"]["scale"].data_ptr(), gp[f"LayerNorm_{i}"]["bias"].data_ptr()
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
    stream = torch.cuda.current_stream(device).cuda_stream
    with torch.cuda.device(device):
        rc = lib.frl_net_eval_bwd(C.byref(ctx.desc), C.byref(ctx.fparams), C.byref(bufs), ctx.ws.data_ptr(), ctx.ws.numel(), stream)
    _lib.check(rc)
    return grads, (g_x.reshape(*ctx.lead, A) if g_x is not None else None)


def add_noise(kind: int, x0: torch.Tensor, e: torch.Tensor, t: torch.Tensor, *, alpha_hats: Optional[torch.Tensor] = None,
              steps: int = 0, p0: float = 0.0, p1: float = 0.0, want_vel: bool = False):
    """frl_add_noise: xt (and vel for FRL_NOISE_FLOW) from (x0, e, t); t (..., 1) int32 (DDPM) or float32."""
    lib = _lib.load()
    device = x0.device
    A = x0.shape[-1]
    B = x0.numel() // A
    x0c, ec = _f32c(x0.reshape(B, A), device, "x0"), _f32c(e.reshape(B, A), device, "eps / x1")
    want_dtype = torch.int32 if kind == _lib.FRL_NOISE_DDPM else torch.float32
    if t.dtype != want_dtype or t.numel() != B:
        raise TypeError(f"t: {B} values of {want_dtype} expected")
    tc = t.reshape(B).contiguous()
    xt = torch.empty_like(x0c)
    vel = torch.empty_like(x0c) if want_vel else None
    ah = None if alpha_hats is None else _f32c(alpha_hats, device, "alpha_hats")
    with torch.cuda.device(device):
        _lib.check(lib.frl_add_noise(kind, B, A, x0c.data_ptr(), ec.data_ptr(), tc.data_ptr(), None if ah is None else ah.data_ptr(),
                                     int(steps), float(p0), float(p1), xt.data_ptr(), None if vel is None else vel.data_ptr(),
                                     torch.cuda.current_stream(device).cuda_stream))
    xt = xt.reshape(x0.shape)
    return (xt, vel.reshape(x0.shape)) if want_vel else xt


def onestep_update(solver: int, xt: torch.Tensor, eps_theta: torch.Tensor, t: torch.Tensor, alphas: torch.Tensor, alpha_hats: torch.Tensor,
                   steps: int, num_samples: int, clip: bool, x_min, x_max, noise: Optional[torch.Tensor]) -> torch.Tensor:
    lib = _lib.load()
    device = xt.device
    B, A = xt.shape
    out = torch.empty((B, num_samples, A), dtype=torch.float32, device=device)
    xc, ec = _f32c(xt, device, "xt"), _f32c(eps_theta, device, "eps_theta")
    tc = t.reshape(B).to(torch.int32).contiguous()
    nz = None if noise is None else _f32c(noise.reshape(B, num_samples, A), device, "noise")
    with torch.cuda.device(device):
        _lib.check(lib.frl_ddpm_onestep_update(solver, B, A, int(num_samples), xc.data_ptr(), ec.data_ptr(), tc.data_ptr(),
                                               alphas.data_ptr(), alpha_hats.data_ptr(), int(steps), 1 if clip else 0,
                                               float(x_min) if x_min is not None else float("-inf"),
                                               float(x_max) if x_max is not None else float("inf"),
                                               None if nz is None else nz.data_ptr(), out.data_ptr(),
                                               torch.cuda.current_stream(device).cuda_stream))
    return out


def mse_loss(pred: torch.Tensor, target: torch.Tensor, want_grad: bool = False):
    """((pred - target) ** 2).mean() (ddpm.py:294, cnf.py:163) as a device scalar; with ``want_grad`` also dloss/dpred."""
    lib = _lib.load()
    device = pred.device
    p, q = _f32c(pred.reshape(-1), device, "pred"), _f32c(target.reshape(-1), device, "target")
    if p.numel() != q.numel() or p.numel() == 0:
        raise ValueError("mse_loss: pred / target of the same non-zero size expected")
    loss = torch.empty((), dtype=torch.float32, device=device)
    grad = torch.empty_like(p) if want_grad else None
    ws = torch.empty(_lib.FRL_MSE_WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.frl_mse_loss(p.numel(), p.data_ptr(), q.data_ptr(), loss.data_ptr(), None if grad is None else grad.data_ptr(),
                                    ws.data_ptr(), ws.numel(), torch.cuda.current_stream(device).cuda_stream))
    return (loss, grad.reshape(pred.shape)) if want_grad else loss


def split3(rng, impl=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    k = _random.split(rng, 3, impl)
    return k[0].contiguous(), k[1].contiguous(), k[2].contiguous()
