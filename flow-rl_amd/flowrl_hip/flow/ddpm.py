"""Discrete-time DDPM sampler — host mirror of flowrl/flow/ddpm.py (class DDPM, :99-221).

Only what the sampling path needs is mirrored: the static fields, the noise tables of
``DDPM.create`` (:132-162) and ``sample`` (:173-221).  ``sample`` keeps the reference's positional
signature and return structure ``(rng, x0, (xt_hist, eps_hist))``; the arithmetic of all T steps runs
in one fused HIP kernel.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, Optional, Tuple

import torch

from .. import _lib, _train, schedules
from .. import random as _random
from .._sampler import run_sampler, run_sampler_train, run_sampler_vjp
from ..module import DDPMBackbone, init_params

SUPPORTED_SOLVERS = ["ddim", "ddpm"]   # ddpm.py:16-18


def draw_noise(rng, shape, device) -> torch.Tensor:
    """Noise for the whole chain when the caller did not supply it.  ``rng`` must be a torch.Generator
    (or None for the global one).  Bit-compatibility with jax.random's threefry key sequence
    (ddpm.py:188,204-206) requires the caller to pass ``noise=`` drawn with those keys."""
    gen = rng if isinstance(rng, torch.Generator) else None
    if gen is not None and gen.device.type != torch.device(device).type:
        return torch.randn(shape, generator=gen, dtype=torch.float32).to(device)
    return torch.randn(shape, generator=gen, dtype=torch.float32, device=device)


@dataclass
class DDPM:
    network: DDPMBackbone
    params: Dict
    x_dim: int = None
    steps: int = None
    clip_sampler: bool = None
    x_min: float = None
    x_max: float = None
    betas: torch.Tensor = None
    alphas: torch.Tensor = None
    alpha_hats: torch.Tensor = None
    postvars: torch.Tensor = None

    @classmethod
    def create(cls, network: DDPMBackbone, rng, inputs, x_dim: int, steps: int, noise_schedule: str,
               noise_schedule_params: Optional[Dict] = None, approx_postvar: bool = False,
               clip_sampler: bool = False, x_min: Optional[float] = None, x_max: Optional[float] = None,
               optimizer=None, clip_grad_norm: float = None, *, params: Optional[Dict] = None,
               device="cuda") -> "DDPM":
        """Mirror of DDPM.create (ddpm.py:113-162).  ``inputs`` = sample (x, t, cond) arrays, used only
        for their trailing dims; ``rng`` = int seed for the parameter init when ``params`` is None."""
        cond_dim = inputs[2].shape[-1] if len(inputs) > 2 and inputs[2] is not None else 0
        if params is None:
            params = init_params(network, x_dim, cond_dim, seed=int(rng) if rng is not None else 0, device=device)
        b, a, ah, pv = schedules.ddpm_tables(steps, noise_schedule, noise_schedule_params, approx_postvar)
        dev = torch.device(device)
        to = lambda v: torch.from_numpy(v).to(dev)
        return cls(network=network, params=params, x_dim=x_dim, steps=steps, clip_sampler=clip_sampler,
                   x_min=x_min, x_max=x_max, betas=to(b), alphas=to(a), alpha_hats=to(ah), postvars=to(pv))

    def replace(self, **kw) -> "DDPM":
        return replace(self, **kw)

    def sample(self, rng, xT: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
               solver: str = "ddpm", params: Optional[Dict] = None, *, noise: Optional[torch.Tensor] = None,
               history: bool = False, particles: Optional[int] = None, compute_dtype: str = "f32", tile_policy: str = "auto",
               rng_impl: Optional[str] = None, cache_prepare: bool = False, rng_shard=None, static_scratch: bool = False, keep_for_vjp: bool = False, select=None) -> Tuple[object, torch.Tensor, Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]]:
        """DDPM.sample (ddpm.py:173-221).  Extra keyword-only arguments:
        ``noise`` (T, *xT.shape): the per-iteration N(0, I) draws (iteration j <-> t = T - j);
        ``history``: also return (xt_hist, eps_hist) (ddpm.py:217); ``particles``: chains per condition
        row when ``condition`` is passed un-repeated; ``compute_dtype`` = "bf16" selects the bf16-MFMA variant (stated
        tolerance, see tests/test_gpu_bf16.py) instead of the bit-faithful f32 kernel.
        ``rng``: a (2,) uint32 threefry key (flowrl_hip.random.PRNGKey / jax.random.key_data) makes the call a
        true rng-in / rng-out drop-in: the scan's split sequence runs on the GPU, the per-step noise is drawn in-kernel
        (no (T, B, A) tensor) and the advanced rng is returned; ``rng_impl`` = "partitionable" (jax >= 0.5 default) or
        "legacy".  A torch.Generator / None keeps the torch-drawn noise path and is returned unchanged."""
        if solver not in SUPPORTED_SOLVERS:
            raise NotImplementedError(f"Unsupported solver {solver} for {type(self)}")
        p = params if (training and params is not None) else self.params   # ddpm.py:191-196
        key = rng if _random.is_key(rng) else None
        if solver == "ddpm" and noise is None and key is None:
            noise = draw_noise(rng, (self.steps, *xT.shape), xT.device)
        if keep_for_vjp:
            tout = run_sampler_train(
            self.network, p, kind=_lib.FRL_KIND_DDPM,
            solver=_lib.FRL_SOLVER_DDPM if solver == "ddpm" else _lib.FRL_SOLVER_DDIM,
            steps=self.steps, xT=xT, condition=condition, noise=noise if solver == "ddpm" else None,
            sched={"betas": self.betas, "alphas": self.alphas, "alpha_hats": self.alpha_hats, "postvars": self.postvars},
            clip_sampler=bool(self.clip_sampler), x_min=self.x_min, x_max=self.x_max,
            particles=particles, compute_dtype=compute_dtype, rng_key=key, rng_impl=rng_impl, rng_shard=rng_shard)
            return (tout[3] if key is not None else rng), tout[0], (tout[1][0], tout[1][1], tout[2])
        out = run_sampler(
            self.network, p, kind=_lib.FRL_KIND_DDPM,
            solver=_lib.FRL_SOLVER_DDPM if solver == "ddpm" else _lib.FRL_SOLVER_DDIM,
            steps=self.steps, xT=xT, condition=condition, noise=noise if solver == "ddpm" else None,
            sched={"betas": self.betas, "alphas": self.alphas, "alpha_hats": self.alpha_hats, "postvars": self.postvars},
            clip_sampler=bool(self.clip_sampler), x_min=self.x_min, x_max=self.x_max, history=history,
            particles=particles, compute_dtype=compute_dtype, tile_policy=tile_policy, rng_key=key, rng_impl=rng_impl, cache_prepare=cache_prepare, rng_shard=rng_shard, static_scratch=static_scratch, select=select)
        return (out[2] if key is not None else rng), out[0], out[1]

    # ---- training-side half ---------------------------------------------------------------------
    def __call__(self, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
                 params: Optional[Dict] = None, *, particles: Optional[int] = None, keep_for_vjp: bool = False):
        """``self(xt, t, condition)`` (module/model.py:62-73 -> DDPMBackbone, ddpm.py:83-95) with a per-sample time input:
        one fused MFMA evaluation per sample (frl_net_eval).  ``t``: integer steps, (..., 1) or (...)."""
        p = params if (training and params is not None) else self.params
        if keep_for_vjp:      # (out, ctx) for eval_vjp: the evaluation under jax.grad
            return _train.net_eval_train(self.network, p, x, t, condition, particles=particles)
        return _train.net_eval(self.network, p, x, t, condition, particles=particles)

    @staticmethod
    def eval_vjp(ctx, g_out: torch.Tensor, want_gx: bool = False):
        """dL/dparams (every leaf) [and dL/dx] of a ``self(x, t, condition, True, params, keep_for_vjp=True)`` call."""
        return _train.net_eval_vjp(ctx, g_out, want_gx)

    def add_noise(self, rng, x0: torch.Tensor, *, rng_impl: Optional[str] = None):
        """DDPM.add_noise (ddpm.py:164-171): rng, t_rng, noise_rng = split(rng, 3); t = randint(t_rng, (..., 1), 1, steps + 1);
        eps = normal(noise_rng, x0.shape); xt = sqrt(alpha_hats[t]) x0 + sqrt(1 - alpha_hats[t]) eps.  ``rng``: a threefry key
        (flowrl_hip.random).  Returns (rng, xt, t, eps) — t int32 (..., 1)."""
        keys = _random.split(rng, 3, rng_impl)
        t = _random.randint(keys[1].contiguous(), (*x0.shape[:-1], 1), 1, self.steps + 1, rng_impl)
        eps = _random.normal(keys[2].contiguous(), tuple(x0.shape), rng_impl)
        xt = _train.add_noise(_lib.FRL_NOISE_DDPM, x0, eps, t, alpha_hats=self.alpha_hats, steps=self.steps)
        return keys[0].contiguous(), xt, t, eps

    def onestep_sample(self, rng, xt_or_x0: torch.Tensor, condition: Optional[torch.Tensor] = None, training: bool = False,
                       num_samples: Optional[int] = None, solver: str = "ddpm", sample_xt: bool = True,
                       t: Optional[torch.Tensor] = None, params: Optional[Dict] = None, *, rng_impl: Optional[str] = None):
        """DDPM.onestep_sample (ddpm.py:223-276), same positional signature and return tuple
        (rng, repeated_xt_1, xt, repeated_t_1, t, eps_theta): optionally noise x0 to a random step, evaluate the network
        once per sample at ITS step, take one reverse step and repeat it num_samples times with fresh noise."""
        if solver not in SUPPORTED_SOLVERS:
            raise NotImplementedError(f"Unsupported solver {solver} for {type(self)}")
        if xt_or_x0.dim() != 2:
            raise ValueError("onestep_sample: (B, A) input expected (ddpm.py:235: B = xt_or_x0.shape[0])")
        B = xt_or_x0.shape[0]
        N = int(num_samples) if num_samples is not None else 1
        keys = _random.split(rng, 4, rng_impl)                     # rng, sample_rng, noise_rng, dropout_rng (ddpm.py:236)
        if sample_xt:
            _, xt, t, _eps_sample = self.add_noise(keys[1].contiguous(), xt_or_x0, rng_impl=rng_impl)
        else:
            if t is None:
                raise ValueError("onestep_sample(sample_xt=False) needs t")
            xt = xt_or_x0
            t = t.reshape(B, 1).to(torch.int32)
        repeated_t = t.repeat_interleave(N, dim=0).reshape(B, N, -1)        # t.repeat(num_samples, axis=0).reshape(...), :243
        repeated_t_1 = repeated_t - 1
        eps_theta = self(xt, t, condition, training, params)
        noise = _random.normal(keys[2].contiguous(), (B, N, xt.shape[-1]), rng_impl) if solver == "ddpm" else None
        rep = _train.onestep_update(_lib.FRL_SOLVER_DDPM if solver == "ddpm" else _lib.FRL_SOLVER_DDIM, xt, eps_theta, t,
                                    self.alphas, self.alpha_hats, self.steps, N, bool(self.clip_sampler), self.x_min, self.x_max, noise)
        return keys[0].contiguous(), rep, xt, repeated_t_1, t, eps_theta

    def sample_vjp(self, xt_hist: torch.Tensor, g_x0: torch.Tensor, condition: Optional[torch.Tensor] = None,
                   solver: str = "ddpm", params: Optional[Dict] = None, *, particles: Optional[int] = None,
                   want_gxT: bool = True, compute_dtype: str = "f32", ctx=None) -> Tuple[Dict, Optional[torch.Tensor]]:
        """VJP of :meth:`sample` w.r.t. the parameter tree and xT: what ``jax.grad`` through
        ``actor.sample(..., training=True, params=p)`` computes in the DQL actor update (dql.py:93-123 through
        ddpm.py:186-219).  ``xt_hist`` = history[0] of the forward call (``sample(..., history=True)``),
        ``g_x0`` = dL/dx0.  Returns (grads in the layout of ``params``, dL/dxT).  ``compute_dtype`` = "bf16": the
        bf16-MFMA VJP (bf16 operands, f32 accumulation; stated tolerance, tests/test_gpu_vjp_bf16.py) — pair it with a
        forward call of the same compute_dtype."""
        if solver not in SUPPORTED_SOLVERS:
            raise NotImplementedError(f"Unsupported solver {solver} for {type(self)}")
        p = params if params is not None else self.params
        tvals = torch.arange(self.steps, 0, -1, dtype=torch.float64)     # iteration j <-> t = T - j
        return run_sampler_vjp(
            self.network, p, kind=_lib.FRL_KIND_DDPM,
            solver=_lib.FRL_SOLVER_DDPM if solver == "ddpm" else _lib.FRL_SOLVER_DDIM, steps=self.steps,
            xt_hist=xt_hist, condition=condition, g_x0=g_x0,
            sched={"betas": self.betas, "alphas": self.alphas, "alpha_hats": self.alpha_hats, "postvars": self.postvars},
            tvals=tvals, clip_sampler=bool(self.clip_sampler), x_min=self.x_min, x_max=self.x_max,
            particles=particles, want_gxT=want_gxT, compute_dtype=compute_dtype, ctx=ctx)


def ddpm_loss(rng, model: DDPM, x0: torch.Tensor, condition: Optional[torch.Tensor] = None, params: Optional[Dict] = None, *,
              rng_impl: Optional[str] = None, want_grads: bool = False):
    """The loss of jit_update_ddpm (ddpm.py:280-304): rng, xt, t, eps = model.add_noise(rng, x0);
    eps_pred = model.apply(params, xt, t, condition); loss = ((eps_pred - eps) ** 2).mean().  Returns (rng, loss, aux) with
    aux = {xt, t, eps, eps_pred, g_eps_pred = dloss / deps_pred}; ``want_grads`` adds aux["grads"] = dloss / dparams (what
    model.apply_gradient's jax.grad, module/model.py:78-86, hands optax).  The optimizer step stays with the caller."""
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
