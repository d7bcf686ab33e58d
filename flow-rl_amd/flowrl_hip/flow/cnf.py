"""Flow-matching / CNF Euler sampler — host mirror of flowrl/flow/cnf.py
(class ContinuousNormalizingFlow, :46-141)."""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Dict, Optional

import torch

from .. import _lib, _train
from .. import random as _random
from .._sampler import run_sampler, run_sampler_train, run_sampler_vjp
from ..module import FlowBackbone, init_params


@dataclass
class ContinuousNormalizingFlow:
    network: FlowBackbone
    params: Dict
    x_dim: int = None
    steps: int = None
    clip_sampler: bool = None
    x_min: float = None
    x_max: float = None

    @classmethod
    def create(cls, network: FlowBackbone, rng, inputs, x_dim: int, steps: int, clip_sampler: bool = False,
               x_min: Optional[float] = None, x_max: Optional[float] = None, optimizer=None,
               clip_grad_norm: float = None, *, params: Optional[Dict] = None, device="cuda"):
        """Mirror of ContinuousNormalizingFlow.create (cnf.py:56-78)."""
        cond_dim = inputs[2].shape[-1] if len(inputs) > 2 and inputs[2] is not None else 0
        if params is None:
            params = init_params(network, x_dim, cond_dim, seed=int(rng) if rng is not None else 0, device=device)
        return cls(network=network, params=params, x_dim=x_dim, steps=steps, clip_sampler=clip_sampler,
                   x_min=x_min, x_max=x_max)

    def replace(self, **kw):
        return replace(self, **kw)

    def sample(self, rng, x0: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
               params: Optional[Dict] = None, *, history: bool = False, particles: Optional[int] = None,
               compute_dtype: str = "f32", tile_policy: str = "auto", rng_impl: Optional[str] = None, cache_prepare: bool = False, rng_shard=None, static_scratch: bool = False, keep_for_vjp: bool = False, select=None):
        """ContinuousNormalizingFlow.sample (cnf.py:113-141): Euler steps t = k/steps, x += vel/steps.  A (2,) uint32
        threefry key as ``rng`` is advanced as the reference's scan does (cnf.py:126) and returned; anything else is
        returned unchanged.
        ``keep_for_vjp=True`` (training callers that differentiate through the chain, dql.py:105-112): runs the training
        forward, which also keeps act(z) / act'(z) of every layer and step; history is then (xt_hist, eps_hist, ctx) and
        ``sample_vjp(..., ctx=ctx)`` runs the backward sweep only instead of recomputing every step."""
        key = rng if _random.is_key(rng) else None
        p = params if (training and params is not None) else self.params
        if keep_for_vjp:
            tout = run_sampler_train(
            self.network, p, kind=_lib.FRL_KIND_CNF, solver=_lib.FRL_SOLVER_EULER, steps=self.steps, xT=x0,
            condition=condition, noise=None, sched={}, clip_sampler=bool(self.clip_sampler), x_min=self.x_min,
            x_max=self.x_max, particles=particles, 
            compute_dtype=compute_dtype, rng_key=key, rng_impl=rng_impl, rng_shard=rng_shard)
            return (tout[3] if key is not None else rng), tout[0], (tout[1][0], tout[1][1], tout[2])
        out = run_sampler(
            self.network, p, kind=_lib.FRL_KIND_CNF, solver=_lib.FRL_SOLVER_EULER, steps=self.steps, xT=x0,
            condition=condition, noise=None, sched={}, clip_sampler=bool(self.clip_sampler), x_min=self.x_min,
            x_max=self.x_max, history=history, particles=particles, compute_dtype=compute_dtype, tile_policy=tile_policy,
            rng_key=key, rng_impl=rng_impl, cache_prepare=cache_prepare, rng_shard=rng_shard, static_scratch=static_scratch, select=select)
        return (out[2] if key is not None else rng), out[0], out[1]

    # ---- training-side half ---------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
                 params: Optional[Dict] = None, *, particles: Optional[int] = None, keep_for_vjp: bool = False):
        """``self(xt, t, condition)`` with a per-sample time in [0, 1] (frl_net_eval)."""
        p = params if (training and params is not None) else self.params
        if keep_for_vjp:      # (out, ctx) for eval_vjp: the evaluation under jax.grad
            return _train.net_eval_train(self.network, p, x, t, condition, particles=particles)
        return _train.net_eval(self.network, p, x, t, condition, particles=particles)

    @staticmethod
    def eval_vjp(ctx, g_out: torch.Tensor, want_gx: bool = False):
        """dL/dparams (every leaf) [and dL/dx] of a ``self(x, t, condition, True, params, keep_for_vjp=True)`` call."""
        return _train.net_eval_vjp(ctx, g_out, want_gx)

    def step2t(self, step):
        """cnf.py:80-83."""
        return step / self.steps

    def linear_interpolation(self, rng, x0: torch.Tensor, x1: torch.Tensor, *, rng_impl: Optional[str] = None):
        """ContinuousNormalizingFlow.linear_interpolation (cnf.py:85-90): rng, t_rng = split(rng, 2);
        t = uniform(t_rng, (..., 1)); xt = (1 - t) x0 + t x1; vel = x1 - x0.  Returns (rng, xt, t, vel)."""
        keys = _random.split(rng, 2, rng_impl)
        t = _random.uniform(keys[1].contiguous(), (*x1.shape[:-1], 1), 0.0, 1.0, rng_impl)
        xt, vel = _train.add_noise(_lib.FRL_NOISE_FLOW, x0, x1, t, want_vel=True)
        return keys[0].contiguous(), xt, t, vel

    def sample_vjp(self, xt_hist: torch.Tensor, g_x1: torch.Tensor, condition: Optional[torch.Tensor] = None,
                   params: Optional[Dict] = None, *, particles: Optional[int] = None, want_gx0: bool = True,
                   compute_dtype: str = "f32", ctx=None):
        """VJP of :meth:`sample` w.r.t. the parameter tree and x0 (what ``jax.grad`` through the unrolled Euler scan
        cnf.py:122-141 computes), from history[0] of the forward call and dL/dx1."""
        p = params if params is not None else self.params
        # step2t(arange(steps)), cnf.py:80-83: float32 k / steps
        tvals = (torch.arange(self.steps, dtype=torch.float32) / torch.tensor(float(self.steps), dtype=torch.float32)).to(torch.float64)
        return run_sampler_vjp(
            self.network, p, kind=_lib.FRL_KIND_CNF, solver=_lib.FRL_SOLVER_EULER, steps=self.steps, xt_hist=xt_hist,
            condition=condition, g_x0=g_x1, sched={}, tvals=tvals, clip_sampler=bool(self.clip_sampler), x_min=self.x_min,
            x_max=self.x_max, particles=particles, want_gxT=want_gx0, compute_dtype=compute_dtype, ctx=ctx)


def flow_matching_loss(rng, model: ContinuousNormalizingFlow, x0: torch.Tensor, x1: torch.Tensor,
                       condition: Optional[torch.Tensor] = None, params: Optional[Dict] = None, *, rng_impl: Optional[str] = None,
                       want_grads: bool = False):
    """The loss of jit_update_flow_matching (cnf.py:145-170): rng, xt, t, vel = model.linear_interpolation(rng, x0, x1);
    vel_pred = model.apply(params, xt, t, condition); loss = ((vel_pred - vel) ** 2).mean().  Returns (rng, loss, aux);
    ``want_grads`` adds aux["grads"] = dloss / dparams."""
    rng, xt, t, vel = model.linear_interpolation(rng, x0, x1, rng_impl=rng_impl)
    p = params if params is not None else model.params
    if want_grads:
        vel_pred, ctx = model(xt, t, condition, True, p, keep_for_vjp=True)
    else:
        vel_pred = model(xt, t, condition, True, p)
    loss, g = _train.mse_loss(vel_pred, vel, want_grad=True)
    aux = {"xt": xt, "t": t, "vel": vel, "vel_pred": vel_pred, "g_vel_pred": g}
    if want_grads:
        aux["grads"] = model.eval_vjp(ctx, g)[0]
    return rng, loss, aux
