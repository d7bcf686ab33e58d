"""Host mirror of ``Model.apply_gradient`` (flowrl/module/model.py:31-60,78-86): jax.grad of a loss over the parameter
tree, then ``TrainState.apply_gradients`` with the optimizer ``Model.create`` was given — ``optax.adam`` behind an optional
``optax.clip_by_global_norm`` (model.py:43-48).

The gradients come from the library (flowrl_hip.autograd: the fused training forward and VJP kernels); what is restated
here is only the per-leaf update rule, a few elementwise torch ops on the device the parameters live on:

    optax.clip_by_global_norm(c):  n = sqrt(sum_leaves sum(g^2));  g <- g            if n < c
                                                                   g <- g / n * c    otherwise
    optax.adam(lr, b1, b2, eps):   mu <- b1 mu + (1 - b1) g;  nu <- b2 nu + (1 - b2) g^2;  k <- k + 1
                                   p  <- p - lr * (mu / (1 - b1^k)) / (sqrt(nu / (1 - b2^k)) + eps)

optax itself is not importable in this image (the update rule above is its published algorithm, optax 0.2.x
``scale_by_adam`` / ``clip_by_global_norm``); tests/test_model_host.py checks it against a float64 numpy restatement —
PARITY UNPINNED by the reference, which holds no optimizer fixtures.  Updates are functional, as in jax: ``apply_gradient``
returns a NEW model and state whose leaves are new tensors; the old ones are left untouched (and a prepared-workspace cache
keyed on them cannot give a stale hit).
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Callable, Dict, Optional, Tuple

import torch

from . import autograd as _ag
from . import random as _random


@dataclass(frozen=True)
class Adam:
    """optax.adam(learning_rate, b1, b2, eps) (eps_root = 0, no weight decay)."""
    learning_rate: float
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8

    def init(self, params: Dict) -> Dict:
        flat = _ag.tree_leaves(params)
        paths = [p for p, _ in flat]
        zeros = lambda: _ag.tree_build(paths, [torch.zeros_like(v, dtype=torch.float32) for _, v in flat])
        return {"count": 0, "mu": zeros(), "nu": zeros()}

    def update(self, grads: Dict, state: Dict, params: Dict) -> Tuple[Dict, Dict]:
        """(new params, new optimizer state) — optax.apply_updates(params, tx.update(grads, state)[0])."""
        k = state["count"] + 1
        c1, c2 = 1.0 - self.b1 ** k, 1.0 - self.b2 ** k
        paths = [p for p, _ in _ag.tree_leaves(params)]
        new_p, new_mu, new_nu = [], [], []
        for path in paths:
            g = _ag._lookup(grads, path)
            p, mu, nu = _ag._lookup(params, path), _ag._lookup(state["mu"], path), _ag._lookup(state["nu"], path)
            if g is None:
                raise KeyError(f"apply_gradients: no gradient for leaf {'/'.join(path)}")
            mu = self.b1 * mu + (1.0 - self.b1) * g
            nu = self.b2 * nu + (1.0 - self.b2) * (g * g)
            new_p.append(p - self.learning_rate * (mu / c1) / (torch.sqrt(nu / c2) + self.eps))
            new_mu.append(mu)
            new_nu.append(nu)
        return _ag.tree_build(paths, new_p), {"count": k, "mu": _ag.tree_build(paths, new_mu), "nu": _ag.tree_build(paths, new_nu)}


def adam(learning_rate: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> Adam:
    return Adam(learning_rate, b1, b2, eps)


def global_norm(grads: Dict) -> torch.Tensor:
    """optax.global_norm: sqrt of the sum of squares over every leaf (a 0-dim tensor on the leaves' device)."""
    leaves = [v for _, v in _ag.tree_leaves(grads)]
    return torch.sqrt(sum((v.to(torch.float32) ** 2).sum() for v in leaves))


def clip_by_global_norm(grads: Dict, max_norm: float) -> Dict:
    """optax.clip_by_global_norm(max_norm): leaves unchanged while the norm is below max_norm, else scaled by max_norm / norm
    (computed on the device: no host synchronisation)."""
    flat = _ag.tree_leaves(grads)
    n = global_norm(grads)
    scale = torch.where(n < max_norm, torch.ones_like(n), max_norm / n)
    return _ag.tree_build([p for p, _ in flat], [v * scale for _, v in flat])


@dataclass
class TrainState:
    """flax.training.train_state.TrainState as Model.create builds it (model.py:43-56): the optimizer (None = the reference's
    empty_optimizer, which refuses to update), the optional global-norm clip in front of it, its state, the step count and the
    model's dropout key (a threefry key or None: the fused path has no dropout, the key is only threaded through)."""
    tx: Optional[Adam]
    opt_state: Optional[Dict]
    clip_grad_norm: Optional[float] = None
    step: int = 0
    dropout_rng: Optional[torch.Tensor] = field(default=None, repr=False)

    @classmethod
    def create(cls, params: Dict, tx: Optional[Adam] = None, clip_grad_norm: Optional[float] = None,
               dropout_rng: Optional[torch.Tensor] = None) -> "TrainState":
        return cls(tx=tx, opt_state=None if tx is None else tx.init(params), clip_grad_norm=clip_grad_norm or None,
                   dropout_rng=dropout_rng)

    def apply_gradients(self, params: Dict, grads: Dict) -> Tuple[Dict, "TrainState"]:
        if self.tx is None:
            raise ValueError("Empty optimizer does not update any parameters.")      # model.py:19
        if self.clip_grad_norm:
            grads = clip_by_global_norm(grads, float(self.clip_grad_norm))
        new_params, opt_state = self.tx.update(grads, self.opt_state, params)
        return new_params, replace(self, opt_state=opt_state, step=self.step + 1)


def apply_gradient(model, state: TrainState, loss_fn: Callable):
    """``model.apply_gradient(loss_fn)`` (model.py:78-86).  ``loss_fn(params, dropout_rng) -> (loss, info)`` is built from
    flowrl_hip.autograd.sample / apply and torch ops; returns (model with the updated params, new state, info).  ``model``
    is any of the flow classes (a dataclass with ``params`` and ``replace``)."""
    dropout_rng = next_rng = state.dropout_rng
    if _random.is_key(state.dropout_rng):
        ks = _random.split(state.dropout_rng, 2)
        dropout_rng, next_rng = ks[0].contiguous(), ks[1].contiguous()
    grads, info = _ag.grad(lambda p: loss_fn(p, dropout_rng), model.params, has_aux=True)
    new_params, new_state = state.apply_gradients(model.params, grads)
    return model.replace(params=new_params), replace(new_state, dropout_rng=next_rng), info
