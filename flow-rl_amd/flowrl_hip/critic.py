"""Critic scoring + best-of-N selection — host mirror of what follows every multi-particle sample in the reference:

    flowrl/module/critic.py:154-194     Ensemblize(ScalarCritic(backbone=MLP)):  q_e = Dense(1)(MLP(concat[obs, action]))
    flowrl/agent/offline/dql.py:130-162 jit_sample_and_select (DAC dac.py:151-183, BDPO bdpo.py:60-92 alike)
    flowrl/agent/online/qvpo.py:21-44   jit_sample_actions (best_of_n)

Parameter tree of ``Model.create(Ensemblize(ScalarCritic(MLP)))`` (flax naming: the dataclass fields ``base`` /
``backbone``, auto-named Dense layers; every leaf carries the ensemble axis first because of nn.vmap with
variable_axes={"params": 0}):

    params/base/backbone/Dense_i/{kernel (E, in, H), bias (E, H)}     i = 0 .. L-1, input = concat([obs, action])
    params/base/Dense_0/{kernel (E, H, 1), bias (E, 1)}

No arithmetic lives here: the leaves go, in place, to ``frl_critic_select`` (include/flowrl_hip.h).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._sampler import _f32c, _workspace
from .module import MLP, _dense, tree_to


FUSED_MIN_CHAINS = 32768        # sample_and_select(fused="auto"): below this the separate launches' latency builds win


@dataclass
class ScalarCritic:
    backbone: MLP


@dataclass
class Ensemblize:
    base: ScalarCritic
    ensemble_size: int = 2


def init_critic_params(critic: Ensemblize, obs_dim: int, act_dim: int, seed: int = 0, bias_std: float = 0.0, device="cpu") -> Dict:
    """Default-initialised tree (orthogonal(sqrt 2) kernels, zero biases) with the ensemble axis stacked first."""
    gen = torch.Generator().manual_seed(seed)
    mlp = critic.base.backbone
    dims = [obs_dim + act_dim, *mlp.hidden_dims]
    E = critic.ensemble_size

    def stacked(n_in, n_out):
        ms = [_dense(gen, n_in, n_out, bias_std) for _ in range(E)]
        return {"kernel": torch.stack([m["kernel"] for m in ms]), "bias": torch.stack([m["bias"] for m in ms])}

    p = {"base": {"backbone": {f"Dense_{i}": stacked(dims[i], dims[i + 1]) for i in range(len(mlp.hidden_dims))},
                  "Dense_0": stacked(dims[-1], 1)}}
    return tree_to(p, device)


def _critic_marshal(critic: Ensemblize, params: Dict, obs: Optional[torch.Tensor], B: int, N: int, A: int, dev: torch.device,
                    temperature: Optional[float], gumbel: Optional[torch.Tensor], want_q: bool):
    """FrlCriticDesc / FrlCriticParams / FrlSelectBufs of one scoring call and its output tensors (best, idx, q)."""
    mlp = critic.base.backbone
    if mlp.layer_norm or (mlp.dropout and mlp.dropout > 0) or mlp.output_dim:
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "critic backbone must be a plain MLP (no layer_norm / dropout / output layer)")
    S = 0 if obs is None else obs.shape[-1]
    if obs is not None and obs.numel() != B * S:
        raise ValueError(f"obs has {obs.numel() // max(S, 1)} rows, expected {B} (un-repeated)")
    d = _lib.FrlCriticDesc()
    d.abi_version, d.particles, d.n_states = _lib.FRL_ABI_VERSION, N, B
    d.obs_dim, d.act_dim, d.n_hidden = S, A, len(mlp.hidden_dims)
    for i, h in enumerate(mlp.hidden_dims):
        d.hidden[i] = int(h)
    d.act, d.ensemble = _lib.FRL_ACT[mlp.act_name], critic.ensemble_size
    keep = []
    fp = _lib.FrlCriticParams()
    E, L = critic.ensemble_size, len(mlp.hidden_dims)
    dims = [S + A, *mlp.hidden_dims]
    for i in range(L + 1):
        node = params["base"]["backbone"][f"Dense_{i}"] if i < L else params["base"]["Dense_0"]
        want_k = (E, dims[i], dims[i + 1]) if i < L else (E, dims[-1], 1)
        if tuple(node["kernel"].shape) != want_k or tuple(node["bias"].shape) != (E, want_k[2]):
            raise ValueError(f"critic Dense_{i}: kernel {tuple(node['kernel'].shape)} bias {tuple(node['bias'].shape)}, expected {want_k}")
        k, b = _f32c(node["kernel"], dev, f"critic kernel {i}"), _f32c(node["bias"], dev, f"critic bias {i}")
        keep += [k, b]
        fp.kernel[i], fp.bias[i] = k.data_ptr(), b.data_ptr()
    bufs = _lib.FrlSelectBufs()
    if obs is not None:
        obs2 = _f32c(obs.reshape(B, S), dev, "obs")
        keep.append(obs2)
        bufs.obs = obs2.data_ptr()
    if temperature is not None and temperature > 0.0:
        if gumbel is None:
            raise ValueError("temperature > 0 draws from a categorical: pass gumbel=(B, N) Gumbel(0,1) noise (jax.random.categorical == argmax(logits + gumbel))")
        g = _f32c(gumbel.reshape(B, N), dev, "gumbel")
        keep.append(g)
        bufs.gumbel, bufs.inv_temperature = g.data_ptr(), 1.0 / (1e-6 + float(temperature))
    q = torch.empty((B, N), dtype=torch.float32, device=dev) if want_q else None
    idx = torch.empty((B,), dtype=torch.int32, device=dev)
    best = torch.empty((B, A), dtype=torch.float32, device=dev)
    if q is not None:
        bufs.q = q.data_ptr()
    bufs.best_idx, bufs.best_action = idx.data_ptr(), best.data_ptr()
    return d, fp, bufs, keep, (best, idx, q)


def critic_select(critic: Ensemblize, params: Dict, obs: Optional[torch.Tensor], actions: torch.Tensor, num_samples: int,
                  *, temperature: Optional[float] = 0.0, gumbel: Optional[torch.Tensor] = None,
                  want_q: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """``obs`` (B, S) UN-repeated, ``actions`` (B, N, A) or (B*N, A).  Returns (best actions (B, A), idx (B,) int32,
    qs (B, N) = min over the ensemble).  temperature <= 0: argmax (dql.py:156-157); temperature > 0: the categorical
    draw of dql.py:158-160 as argmax(qs / (1e-6 + T) + gumbel) with ``gumbel`` (B, N) supplied by the caller."""
    lib = _lib.load()
    if not actions.is_cuda:
        raise RuntimeError("flowrl_hip critic scoring runs on the GPU only; there is no CPU fallback")
    dev = actions.device
    A = actions.shape[-1]
    N = int(num_samples)
    B = actions.numel() // (A * N)
    if B * N * A != actions.numel():
        raise ValueError("actions do not hold num_samples candidates per state")
    d, fp, bufs, keep, (best, idx, q) = _critic_marshal(critic, params, obs, B, N, A, dev, temperature, gumbel, want_q)
    act2 = _f32c(actions.reshape(B * N, A), dev, "actions")
    bufs.actions = act2.data_ptr()
    need = lib.frl_critic_workspace_bytes(C.byref(d))
    if need == 0:
        _lib.check(lib.frl_critic_select(C.byref(d), C.byref(fp), C.byref(bufs), None, 0, None))
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "this critic configuration has no kernel")
    ws = _workspace(need, dev, torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.device(dev):
        rc = lib.frl_critic_select(C.byref(d), C.byref(fp), C.byref(bufs), ws.data_ptr(), ws.numel(),
                                   torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc)
    return best, idx, q


class FusedSelect:
    """Critic scoring + selection carried INSIDE the sampler's launch (frl_sampler_select_fwd): pass as ``select=`` to a
    model's ``sample``; afterwards ``best`` (B, A), ``idx`` (B,) int32 and ``q`` (B, N) (``want_q``) hold what
    :func:`critic_select` would have returned for the sampled candidates, which are only written out (the sample call's
    actions) with ``want_actions``."""

    def __init__(self, critic: Ensemblize, params: Dict, obs: Optional[torch.Tensor], *, temperature: Optional[float] = 0.0,
                 gumbel: Optional[torch.Tensor] = None, want_q: bool = False, want_actions: bool = False):
        self.critic, self.params, self.obs = critic, params, obs
        self.temperature, self.gumbel, self.want_q, self.want_actions = temperature, gumbel, want_q, want_actions
        self.best = self.idx = self.q = None

    def marshal(self, n_states: int, particles: int, A: int, dev: torch.device):
        d, fp, bufs, keep, (self.best, self.idx, self.q) = _critic_marshal(
            self.critic, self.params, self.obs, n_states, particles, A, dev, self.temperature, self.gumbel, self.want_q)
        return d, fp, bufs, keep


def sample_and_select(model, critic: Ensemblize, critic_params: Dict, rng, obs: torch.Tensor, training: bool, num_samples: int,
                      solver: str, temperature: Optional[float], *, xT: Optional[torch.Tensor] = None,
                      noise: Optional[torch.Tensor] = None, gumbel: Optional[torch.Tensor] = None, compute_dtype: str = "f32",
                      rng_impl: Optional[str] = None, fused="auto"):
    """jit_sample_and_select (dql.py:130-162): num_samples chains per observation, then the critic picks one.
    With a threefry key as ``rng`` (flowrl_hip.random.PRNGKey) the call is the reference's rng-in / rng-out protocol,
    all on the GPU: ``rng, xT_rng = split(rng)``; ``xT = normal(xT_rng, (B, N, A))``; the sampler draws its noise
    in-kernel and advances the rng; ``rng, select_rng = split(rng)`` and Gumbel-max over ``gumbel(select_rng, (B, N))``
    when temperature > 0.  Otherwise ``xT`` (B, N, A) / ``noise`` / ``gumbel`` are the reference's draws supplied by the
    caller (torch draws when absent) and ``rng`` is returned unchanged.
    ``fused``: True = one launch for sampler + critic + selection (frl_sampler_select_fwd; raises when the configuration
    has no such build), False = sampler launch, critic launches, selection launch; "auto" = fused when the build exists
    and the batch is throughput-sized (>= FUSED_MIN_CHAINS chains), else the separate launches (whose small-batch
    latency builds the fused kernel does not have).  The fused path needs caller-side draws (no threefry key)."""
    from . import random as fr
    B = obs.shape[0]
    keyed = fr.is_key(rng)
    if keyed:
        rng, xT_rng = fr.split(fr.as_key(rng, obs.device), 2, rng_impl)                                  # dql.py:142
        if xT is None:
            xT = fr.normal(xT_rng, (B, num_samples, model.x_dim), rng_impl)                              # dql.py:146
    elif xT is None:
        xT = torch.randn((B, num_samples, model.x_dim), dtype=torch.float32, device=obs.device)
    import inspect
    accepted = inspect.signature(model.sample).parameters          # CNF.sample has neither solver nor noise
    kw = dict(noise=noise, particles=num_samples, compute_dtype=compute_dtype)
    if keyed:
        kw["rng_impl"] = rng_impl
    if solver is not None:
        kw["solver"] = solver
    kw = {k: v for k, v in kw.items() if k in accepted}
    can_fuse = temperature is not None and not keyed and compute_dtype == "f32" and (temperature <= 0.0 or gumbel is not None)
    if fused is True and not can_fuse:
        raise _lib.FlowRLHipUnsupported(_lib.FRL_EUNSUPPORTED, "fused sample+select: needs a temperature, caller-side draws (no threefry key, gumbel given) and f32")
    if can_fuse and (fused is True or (fused == "auto" and B * num_samples >= FUSED_MIN_CHAINS)):
        sel = FusedSelect(critic, critic_params, obs, temperature=temperature, gumbel=gumbel)
        try:
            model.sample(rng, xT, obs, training, select=sel, **kw)
            return rng, sel.best
        except _lib.FlowRLHipUnsupported:
            if fused is True:
                raise
    rng, actions, _ = model.sample(rng, xT, obs, training, **kw)
    if temperature is None:
        return rng, actions[:, 0]                                           # dql.py:148-149
    if keyed and temperature > 0.0:
        rng, select_rng = fr.split(rng, 2, rng_impl)                                                     # dql.py:159
        if gumbel is None:
            gumbel = fr.gumbel(select_rng, (B, num_samples), rng_impl)                                   # categorical, dql.py:160
    best, _, _ = critic_select(critic, critic_params, obs, actions, num_samples, temperature=temperature, gumbel=gumbel, want_q=False)
    return rng, best
