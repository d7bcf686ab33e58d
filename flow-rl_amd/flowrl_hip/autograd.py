"""torch.autograd bridge — the host-side stand-in for ``jax.grad`` through the reference's model calls.

The reference differentiates its losses with ``jax.grad(loss_fn, has_aux=True)(params, rng)`` (module/model.py:78-86); inside
``loss_fn`` the actor is called as ``actor.sample(rng, xT, obs, training=True, params=p)`` (the DQL / DACER actor update,
agent/offline/dql.py:93-123, agent/online/dacer.py:141-162) or as ``model.apply({"params": p}, xt, t, cond, training=True)``
(the eps-MSE / flow-matching losses, flow/ddpm.py:286-296, continuous_ddpm.py:234-243, cnf.py:155-165).  This module
gives a torch caller the same shape of code:

    def loss_fn(p):
        rng_, x0, _ = fa.sample(actor, rng, xT, obs, params=p, solver="ddpm", noise=noise)
        return -(critic(obs, x0)).mean(), {"x0": x0}
    (loss, aux), grads = fa.value_and_grad(loss_fn, actor.params, has_aux=True)

``sample`` / ``apply`` are ``torch.autograd.Function``s whose forward is the library's training forward
(frl_sampler_fwd_train / frl_net_eval_train: the fused kernel plus the stash) and whose backward is the library's VJP
(frl_sampler_bwd / frl_net_eval_bwd) — nothing is recomputed or differentiated by torch itself, and there is no CPU path.
Gradients flow to every parameter leaf that requires grad and to ``xT`` / ``x``; the condition is not differentiated
(the reference's losses never ask for it: obs comes from the batch) and asking for it raises.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch

from .flow.cnf import ContinuousNormalizingFlow

Path = Tuple[str, ...]


def tree_leaves(tree: Dict, prefix: Path = ()) -> List[Tuple[Path, torch.Tensor]]:
    """(path, tensor) of every leaf, in sorted key order (the order jax.tree_util flattens a dict in)."""
    out: List[Tuple[Path, torch.Tensor]] = []
    for k in sorted(tree):
        v = tree[k]
        if isinstance(v, dict):
            out.extend(tree_leaves(v, prefix + (k,)))
        else:
            out.append((prefix + (k,), v))
    return out


def tree_build(paths: List[Path], leaves) -> Dict:
    tree: Dict = {}
    for path, leaf in zip(paths, leaves):
        node = tree
        for k in path[:-1]:
            node = node.setdefault(k, {})
        node[path[-1]] = leaf
    return tree


def _lookup(tree: Dict, path: Path):
    node = tree
    for k in path:
        if not isinstance(node, dict) or k not in node:
            return None
        node = node[k]
    return node


def _no_cond_grad(condition):
    if condition is not None and condition.requires_grad:
        raise NotImplementedError("flowrl_hip.autograd: the condition is not differentiated (the library's VJP returns dL/dparams "
                                  "and dL/dx only); detach it")


class _SampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rng, condition, kw, paths, xT, *leaves):
        params = tree_build(paths, leaves)
        rng_out, x0, (xh, eh, vctx) = model.sample(rng, xT, condition, True, params=params, keep_for_vjp=True, **kw)
        ctx.model, ctx.condition, ctx.kw, ctx.paths, ctx.vctx = model, condition, kw, paths, vctx
        # saved through autograd so that an in-place change of xT / a parameter leaf between forward and backward is caught
        # by the version counters (the stash was computed from the old values)
        ctx.save_for_backward(xT, *leaves)
        ctx.mark_non_differentiable(*[t for t in (xh, eh, rng_out) if torch.is_tensor(t)])
        return x0, xh, eh, rng_out      # rng_out: the advanced threefry key (a tensor), or the caller's generator / None as it came

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_x0, _g_xh, _g_eh, _g_rng):
        model, kw = ctx.model, ctx.kw
        want_gx = ctx.needs_input_grad[5]
        bkw = {k: kw[k] for k in ("solver", "steps", "t_schedule_n", "particles", "compute_dtype") if k in kw}
        bkw["want_gx0" if isinstance(model, ContinuousNormalizingFlow) else "want_gxT"] = want_gx
        params = tree_build(ctx.paths, ctx.saved_tensors[1:])
        # the stash (the context's workspace) is only read by the backward and lives as long as the autograd node: a second
        # backward through the same graph (retain_graph=True, two losses sharing one sample) runs from it again
        grads, g_x = model.sample_vjp(None, g_x0.contiguous(), ctx.condition, params=params, ctx=ctx.vctx, **bkw)
        g_leaves = [(_lookup(grads, p) if need else None) for p, need in zip(ctx.paths, ctx.needs_input_grad[6:])]
        return (None, None, None, None, None, g_x, *g_leaves)


def sample(model, rng, xT: torch.Tensor, condition: Optional[torch.Tensor] = None, params: Optional[Dict] = None, **kw):
    """``model.sample(rng, xT, condition, training=True, params=params, **kw)`` as a differentiable torch op: returns
    (rng, x0, (xt_hist, eps_hist)) like the reference's sample, with ``x0`` carrying a grad_fn whose backward is the
    library's chain VJP.  ``kw``: what the class's own ``sample`` takes by keyword (``solver``, ``noise``, ``particles``,
    ``compute_dtype``, ``rng_impl``; ContinuousDDPM also ``steps`` / ``t_schedule_n``)."""
    _no_cond_grad(condition)
    p = params if params is not None else model.params
    flat = tree_leaves(p)
    paths = [path for path, _ in flat]
    if not (torch.is_grad_enabled() and (xT.requires_grad or any(leaf.requires_grad for _, leaf in flat))):
        # nothing to differentiate (an evaluation under no_grad, a target network): the plain sampler, no stash
        rng_out, x0, hist = model.sample(rng, xT, condition, True, params=p, history=True, **kw)
        return rng_out, x0, (hist[0], hist[1])
    x0, xh, eh, rng_out = _SampleFn.apply(model, rng, condition, dict(kw), paths, xT, *[leaf for _, leaf in flat])
    return rng_out, x0, (xh, eh)


class _ApplyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, t, condition, particles, paths, x, *leaves):
        params = tree_build(paths, leaves)
        out, ectx = model(x, t, condition, True, params, particles=particles, keep_for_vjp=True)
        ctx.model, ctx.paths, ctx.ectx = model, paths, ectx
        ctx.save_for_backward(x, *leaves)      # version-counter checks, as in _SampleFn
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        _ = ctx.saved_tensors                   # raises if x / a leaf was modified in place since the forward
        grads, g_x = ctx.model.eval_vjp(ctx.ectx, g_out.contiguous(), want_gx=ctx.needs_input_grad[5])
        g_leaves = [(_lookup(grads, p) if need else None) for p, need in zip(ctx.paths, ctx.needs_input_grad[6:])]
        return (None, None, None, None, None, g_x, *g_leaves)


def apply(model, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, params: Optional[Dict] = None, *,
          particles: Optional[int] = None) -> torch.Tensor:
    """``model.apply({"params": params}, x, t, condition, training=True)`` as a differentiable torch op (one fused
    evaluation per sample at ITS time; backward = frl_net_eval_bwd: every parameter leaf and dL/dx)."""
    _no_cond_grad(condition)
    p = params if params is not None else model.params
    flat = tree_leaves(p)
    if not (torch.is_grad_enabled() and (x.requires_grad or any(leaf.requires_grad for _, leaf in flat))):
        return model(x, t, condition, True, p, particles=particles)      # nothing to differentiate: the plain evaluation
    return _ApplyFn.apply(model, t, condition, particles, [path for path, _ in flat], x, *[leaf for _, leaf in flat])


def value_and_grad(loss_fn: Callable, params: Dict, has_aux: bool = False):
    """``jax.value_and_grad(loss_fn, has_aux=has_aux)(params)`` for a loss built from :func:`sample` / :func:`apply` and
    torch ops: calls ``loss_fn(p)`` with a tree of grad-requiring aliases of ``params`` (same storage, so the kernels read
    the weights in place), backpropagates, and returns (value, grads) — ``grads`` in the layout of ``params``, zeros for
    leaves the loss does not reach (jax.grad's convention)."""
    flat = tree_leaves(params)
    paths = [path for path, _ in flat]
    aliases = [leaf.detach().requires_grad_(True) for _, leaf in flat]
    out = loss_fn(tree_build(paths, aliases))
    loss = out[0] if has_aux else out
    if not (torch.is_tensor(loss) and loss.numel() == 1):
        raise TypeError("value_and_grad: loss_fn must return a scalar tensor" + (" first (has_aux=True)" if has_aux else ""))
    gs = torch.autograd.grad(loss, aliases, allow_unused=True)
    grads = tree_build(paths, [g if g is not None else torch.zeros_like(a) for g, a in zip(gs, aliases)])
    return out, grads


def grad(loss_fn: Callable, params: Dict, has_aux: bool = False):
    """``jax.grad(loss_fn, has_aux=has_aux)(params)`` (module/model.py:83): grads, or (grads, aux)."""
    out, grads = value_and_grad(loss_fn, params, has_aux)
    return (grads, out[1]) if has_aux else grads
