"""Continuous-time VP diffusion sampler — host mirror of flowrl/flow/continuous_ddpm.py
(class ContinuousDDPM, :47-222; all four solvers — ddpm, ddim, dpm2, dpm++2m — run in the fused f32 kernel)."""
from __future__ import annotations

from dataclasses import dataclass, replace
from functools import partial
from typing import Callable, Dict, Optional, Tuple

import torch

from .. import _lib, _train, schedules
from .. import random as _random
from .._sampler import run_sampler, run_sampler_train, run_sampler_vjp
from ..module import DDPMBackbone, init_params
from .ddpm import draw_noise


@dataclass
class ContinuousDDPM:
    network: DDPMBackbone
    params: Dict
    x_dim: int = None
    steps: int = None
    clip_sampler: bool = None
    x_min: float = None
    x_max: float = None
    solver: str = "ddpm"
    t_schedule_n: float = None
    t_diffusion: Tuple[float, float] = None
    noise_schedule_func: Callable = None

    @classmethod
    def create(cls, network: DDPMBackbone, rng, inputs, x_dim: int, steps: int, noise_schedule: str,
               noise_schedule_params: Optional[Dict] = None, clip_sampler: bool = False,
               x_min: Optional[float] = None, x_max: Optional[float] = None, solver: str = "ddpm",
               t_schedule_n: float = 1.0, epsilon: float = 0.001, optimizer=None, clip_grad_norm: float = None,
               *, params: Optional[Dict] = None, device="cuda") -> "ContinuousDDPM":
        """Mirror of ContinuousDDPM.create (continuous_ddpm.py:61-105)."""
        kw = dict(noise_schedule_params or {})
        if noise_schedule == "cosine":
            t_diffusion = [epsilon, 0.9946]
            fn = partial(schedules.cosine_noise_schedule, **kw)
        elif noise_schedule == "linear":
            t_diffusion = [epsilon, 1.0]
            fn = partial(schedules.linear_noise_schedule, **kw)
        else:
            raise NotImplementedError(f"Unsupported noise schedule: {noise_schedule}")
        cond_dim = inputs[2].shape[-1] if len(inputs) > 2 and inputs[2] is not None else 0
        if params is None:
            params = init_params(network, x_dim, cond_dim, seed=int(rng) if rng is not None else 0, device=device)
        return cls(network=network, params=params, x_dim=x_dim, steps=steps, clip_sampler=clip_sampler, x_min=x_min,
                   x_max=x_max, solver=solver, t_schedule_n=t_schedule_n, t_diffusion=t_diffusion, noise_schedule_func=fn)

    def replace(self, **kw) -> "ContinuousDDPM":
        """dataclasses.replace builds a NEW instance; the device-table cache travels with it (the entries are keyed by the
        schedule function and t_diffusion too, so a replaced schedule simply misses): `model.apply_gradient` replaces the
        model every optimizer step, and rebuilding the tables there costs four synchronous host-to-device copies per step."""
        new = replace(self, **kw)
        cache = self.__dict__.get("_tables")
        if cache is not None:
            new.__dict__["_tables"] = cache
        return new

    def _device_tables(self, steps: int, t_schedule_n: float, midpoints: bool, dev) -> Dict[str, torch.Tensor]:
        """The per-call tables of continuous_ddpm.py:134-138 (and the dpm2 midpoints, :181-184), built once per
        (steps, t_schedule_n, device) and kept on the GPU: the reference rebuilds them inside every jitted call, where
        XLA constant-folds them."""
        cache = self.__dict__.setdefault("_tables", {})
        key = (steps, float(t_schedule_n), bool(midpoints), str(dev), id(self.noise_schedule_func), tuple(float(v) for v in self.t_diffusion))
        ent = cache.get(key)
        sched = ent[1] if ent is not None and ent[0] is self.noise_schedule_func else None    # the entry keeps its function alive: ids are not recycled
        if sched is None:
            to = lambda v: torch.from_numpy(v).to(dev)
            ts, ah, al, be = schedules.continuous_tables(steps, self.noise_schedule_func, self.t_diffusion, t_schedule_n)
            sched = {"ts": to(ts), "alpha_hats": to(ah), "alphas": to(al), "betas": to(be)}
            if midpoints:
                mid, mal, msg = schedules.continuous_midpoints(ts, self.noise_schedule_func)
                sched.update(mid_ts=to(mid), mid_alphas=to(mal), mid_sigmas=to(msg))
            cache[key] = (self.noise_schedule_func, sched)
        return sched

    def sample(self, rng, xT: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
               solver: Optional[str] = None, steps: Optional[int] = None, t_schedule_n: Optional[float] = None,
               params: Optional[Dict] = None, *, noise: Optional[torch.Tensor] = None, history: bool = False,
               particles: Optional[int] = None, compute_dtype: str = "f32", tile_policy: str = "auto",
               rng_impl: Optional[str] = None, cache_prepare: bool = False, rng_shard=None, static_scratch: bool = False, keep_for_vjp: bool = False, select=None):
        """ContinuousDDPM.sample (continuous_ddpm.py:117-222).
        ``rng``: a (2,) uint32 threefry key (flowrl_hip.random.PRNGKey / jax.random.key_data) makes the call a
        true rng-in / rng-out drop-in: the scan's split sequence runs on the GPU, the per-step noise is drawn in-kernel
        (no (T, B, A) tensor) and the advanced rng is returned; ``rng_impl`` = "partitionable" (jax >= 0.5 default) or
        "legacy".  A torch.Generator / None keeps the torch-drawn noise path and is returned unchanged."""
        steps = steps or self.steps
        t_schedule_n = t_schedule_n or self.t_schedule_n
        solver = solver or self.solver
        solver_ids = {"ddpm": _lib.FRL_SOLVER_DDPM, "ddim": _lib.FRL_SOLVER_DDIM, "dpm2": _lib.FRL_SOLVER_DPM2,
                      "dpm++2m": _lib.FRL_SOLVER_DPMPP2M}
        if solver not in solver_ids:
            raise NotImplementedError(f"Unsupported solver: {solver}")       # continuous_ddpm.py:215
        dev = xT.device
        p = params if (training and params is not None) else self.params
        key = rng if _random.is_key(rng) else None
        if solver == "ddpm" and noise is None and key is None:
            noise = draw_noise(rng, (steps, *xT.shape), xT.device)
        sched = self._device_tables(steps, t_schedule_n, solver == "dpm2", dev)
        if keep_for_vjp:
            tout = run_sampler_train(
            self.network, p, kind=_lib.FRL_KIND_CDDPM, solver=solver_ids[solver],
            steps=steps, xT=xT, condition=condition, noise=noise if solver == "ddpm" else None, sched=sched,
            clip_sampler=bool(self.clip_sampler), x_min=self.x_min, x_max=self.x_max,
            particles=particles, compute_dtype=compute_dtype, rng_key=key, rng_impl=rng_impl, rng_shard=rng_shard)
            return (tout[3] if key is not None else rng), tout[0], (tout[1][0], tout[1][1], tout[2])
        out = run_sampler(
            self.network, p, kind=_lib.FRL_KIND_CDDPM, solver=solver_ids[solver],
            steps=steps, xT=xT, condition=condition, noise=noise if solver == "ddpm" else None, sched=sched,
            clip_sampler=bool(self.clip_sampler), x_min=self.x_min, x_max=self.x_max, history=history,
            particles=particles, compute_dtype=compute_dtype, tile_policy=tile_policy, rng_key=key, rng_impl=rng_impl, cache_prepare=cache_prepare, rng_shard=rng_shard, static_scratch=static_scratch, select=select)
        return (out[2] if key is not None else rng), out[0], out[1]

    # ---- training-side half ---------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
                 params: Optional[Dict] = None, *, particles: Optional[int] = None, keep_for_vjp: bool = False):
        """``self(xt, t, condition)`` with a per-sample continuous time (frl_net_eval)."""
        p = params if (training and params is not None) else self.params
        if keep_for_vjp:      # (out, ctx) for eval_vjp: the evaluation under jax.grad
            return _train.net_eval_train(self.network, p, x, t, condition, particles=particles)
        return _train.net_eval(self.network, p, x, t, condition, particles=particles)

    @staticmethod
    def eval_vjp(ctx, g_out: torch.Tensor, want_gx: bool = False):
        """dL/dparams (every leaf) [and dL/dx] of a ``self(x, t, condition, True, params, keep_for_vjp=True)`` call."""
        return _train.net_eval_vjp(ctx, g_out, want_gx)

    def _noise_kind(self):
        fn = self.noise_schedule_func
        name = getattr(getattr(fn, "func", fn), "__name__", "")
        kw = dict(getattr(fn, "keywords", {}) or {})
        if name == "linear_noise_schedule":
            return _lib.FRL_NOISE_CDDPM_LINEAR, float(kw.get("beta0", 0.1)), float(kw.get("beta1", 20.0))
        if name == "cosine_noise_schedule":
            return _lib.FRL_NOISE_CDDPM_COSINE, float(kw.get("s", 0.008)), 0.0
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, f"noise_schedule_func {name!r}: linear / cosine have kernels")

    def add_noise(self, rng, x0: torch.Tensor, *, rng_impl: Optional[str] = None):
        """ContinuousDDPM.add_noise (continuous_ddpm.py:107-115): t = uniform(t_rng, (..., 1), t_diffusion[0], t_diffusion[1]);
        eps = normal(noise_rng, x0.shape); alpha, sigma = noise_schedule_func(t); xt = alpha x0 + sigma eps.
        Returns (rng, xt, t, eps)."""
        keys = _random.split(rng, 3, rng_impl)
        t = _random.uniform(keys[1].contiguous(), (*x0.shape[:-1], 1), self.t_diffusion[0], self.t_diffusion[1], rng_impl)
        eps = _random.normal(keys[2].contiguous(), tuple(x0.shape), rng_impl)
        kind, p0, p1 = self._noise_kind()
        xt = _train.add_noise(kind, x0, eps, t, p0=p0, p1=p1)
        return keys[0].contiguous(), xt, t, eps

    def sample_vjp(self, xt_hist: torch.Tensor, g_x0: torch.Tensor, condition: Optional[torch.Tensor] = None,
                   solver: Optional[str] = None, steps: Optional[int] = None, t_schedule_n: Optional[float] = None,
                   params: Optional[Dict] = None, *, particles: Optional[int] = None, want_gxT: bool = True,
                   compute_dtype: str = "f32", ctx=None):
        """VJP of :meth:`sample` (solvers ddpm / ddim) w.r.t. the parameter tree and xT — what DACER's actor loss
        differentiates (agent/online/dacer.py:141-162: jax.grad through continuous_ddpm.py:159-220), from history[0] of the
        forward call and dL/dx0."""
        steps = steps or self.steps
        t_schedule_n = t_schedule_n or self.t_schedule_n
        solver = solver or self.solver
        if solver not in ("ddpm", "ddim"):
            raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, f"solver {solver} has no backward kernel (ddpm / ddim)")
        p = params if params is not None else self.params
        # the forward call's cached device tables: uploading them again would put four synchronous host-to-device copies
        # between the training forward and its backward (0.2 ms of idle GPU in a 2.5 ms DACER-size step)
        sched = self._device_tables(steps, t_schedule_n, False, g_x0.device)
        return run_sampler_vjp(
            self.network, p, kind=_lib.FRL_KIND_CDDPM,
            solver=_lib.FRL_SOLVER_DDPM if solver == "ddpm" else _lib.FRL_SOLVER_DDIM, steps=steps, xt_hist=xt_hist,
            condition=condition, g_x0=g_x0, sched=sched,
            tvals=None, clip_sampler=bool(self.clip_sampler), x_min=self.x_min, x_max=self.x_max, particles=particles,
            want_gxT=want_gxT, compute_dtype=compute_dtype, ctx=ctx)


def continuous_ddpm_loss(rng, model: ContinuousDDPM, x0: torch.Tensor, condition: Optional[torch.Tensor] = None,
                         params: Optional[Dict] = None, *, rng_impl: Optional[str] = None, want_grads: bool = False):
    """The loss of jit_update_continuous_ddpm (continuous_ddpm.py:226-245).  Returns (rng, loss, aux) as ddpm_loss does
    (``want_grads``: aux["grads"] = dloss / dparams)."""
    rng, xt, t, eps = model.add_noise(rng, x0, rng_impl=rng_impl)
    p = params if params is not None else model.params
    if want_grads:
        eps_pred, ctx = model(xt, t, condition, True, p, keep_for_vjp=True)
    else:
        eps_pred = model(xt, t, condition, True, p)
    loss, g = _train.mse_loss(eps_pred, eps, want_grad=True)
    aux = {"xt": xt, "t": t, "eps": eps, "eps_pred": eps_pred, "g_eps_pred": g}
    if want_grads:
        aux["grads"] = model.eval_vjp(ctx, g)[0]
    return rng, loss, aux
