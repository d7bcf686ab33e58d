"""Host-side mirror of the reference's network *descriptions* and Flax parameter layout.

No arithmetic lives here: these classes only describe the backbone so that the parameter tree of
``Model.create`` (flowrl/module/model.py:31-60) can be handed, leaf by leaf and in place, to the
fused HIP kernel.  Names follow the reference:

  MLP                          flowrl/module/mlp.py:10-34 (Dense -> [LayerNorm] -> act); with ln_after_act=True FQL's
                               variant agent/offline/fql/network.py:10-35 (Dense -> act -> [LayerNorm])
  PositionalEmbedding          flowrl/module/time_embedding.py:6-16
  LearnableFourierEmbedding    flowrl/module/time_embedding.py:19-27
  DDPMBackbone / FlowBackbone  flowrl/flow/ddpm.py:70-95 / flowrl/flow/cnf.py:17-42

Parameter tree (flax auto-naming inside @nn.compact):
  params/time_embedding/kernel            (D/2, 1)      learnable-Fourier only
  params/MLP_0/Dense_{0,1}/{kernel,bias}  (D, D), (D,)  time MLP, mish after both layers
  params/<predictor>/Dense_i/{kernel,bias}              predictor MLP, kernel (in, out) row-major
with <predictor> = noise_predictor (DDPMBackbone) or vel_predictor (FlowBackbone).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional, Sequence, Union

import torch

ACT_NAMES = ("mish", "silu", "relu", "gelu_tanh", "gelu", "tanh", "elu")


def _act_name(activation: Union[str, Callable]) -> str:
    if isinstance(activation, str):
        name = activation
    else:
        name = getattr(activation, "__name__", str(activation))
    if name == "gelu":
        name = "gelu_tanh"   # flax.linen.gelu defaults to approximate=True (fql/fql.py:199)
    if name not in ACT_NAMES:
        raise NotImplementedError(f"activation {name!r} has no fused kernel")
    return name


@dataclass
class MLP:
    hidden_dims: Sequence[int] = field(default_factory=list)
    output_dim: int = 0
    activation: Union[str, Callable] = "relu"
    layer_norm: bool = False
    dropout: Optional[float] = None
    ln_after_act: bool = False      # FQL's MLP (agent/offline/fql/network.py:26-30): Dense -> act -> LayerNorm instead of
                                    # flowrl/module/mlp.py:26-29's Dense -> LayerNorm -> act; only read when layer_norm is set

    @property
    def act_name(self) -> str:
        return _act_name(self.activation)


@dataclass
class ResidualMLP:
    """flowrl/module/mlp.py:67-98.  hidden_dims = one entry per residual block (all equal: a block whose width differs
    from its input's gets a projection Dense the fused kernel does not build)."""
    hidden_dims: Sequence[int] = field(default_factory=list)
    output_dim: int = 0
    multiplier: int = 4
    activation: Union[str, Callable] = "relu"
    layer_norm: bool = False
    dropout: Optional[float] = None
    block_name = "ResidualLinear"

    @property
    def act_name(self) -> str:
        return _act_name(self.activation)


@dataclass
class Simba:
    """flowrl/module/simba.py:28-63 (https://arxiv.org/abs/2410.09754): LayerNorm in every block and before the output layer."""
    hidden_dims: Sequence[int] = field(default_factory=list)
    output_dim: int = 0
    activation: Union[str, Callable] = "relu"
    multiplier: int = 4
    layer_norm = True       # always (simba.py:22)
    dropout = None
    block_name = "SimbaBlock"

    @property
    def act_name(self) -> str:
        return _act_name(self.activation)


def is_residual(pred) -> bool:
    return isinstance(pred, (ResidualMLP, Simba))


@dataclass
class PositionalEmbedding:
    output_dim: int
    max_positions: int = 10000


@dataclass
class LearnableFourierEmbedding:
    output_dim: int


@dataclass
class DDPMBackbone:
    noise_predictor: Union[MLP, ResidualMLP, Simba]
    time_embedding: Optional[Union[PositionalEmbedding, LearnableFourierEmbedding]] = None
    cond_embedding: Optional[MLP] = None
    predictor_name = "noise_predictor"

    @property
    def predictor(self) -> MLP:
        return self.noise_predictor


@dataclass
class FlowBackbone:
    vel_predictor: Union[MLP, ResidualMLP, Simba]
    time_embedding: Optional[Union[PositionalEmbedding, LearnableFourierEmbedding]] = None
    cond_embedding: Optional[MLP] = None
    predictor_name = "vel_predictor"

    @property
    def predictor(self) -> MLP:
        return self.vel_predictor


ContinuousDDPMBackbone = DDPMBackbone   # continuous_ddpm.py:44


def _orthogonal(gen: torch.Generator, n_in: int, n_out: int, gain: float) -> torch.Tensor:
    # default_kernel_init = orthogonal(sqrt(2)), flowrl/module/initialization.py:9-21
    a = torch.randn((max(n_in, n_out), min(n_in, n_out)), generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if n_in < n_out:
        q = q.T
    return (gain * q[:n_in, :n_out]).to(torch.float32).contiguous()


def _dense(gen, n_in, n_out, bias_std):
    b = torch.zeros(n_out) if bias_std == 0.0 else bias_std * torch.randn(n_out, generator=gen)
    return {"kernel": _orthogonal(gen, n_in, n_out, math.sqrt(2.0)), "bias": b.to(torch.float32)}


def init_params(network, x_dim: int, cond_dim: int, seed: int = 0, bias_std: float = 0.0,
                device: Union[str, torch.device] = "cpu") -> Dict:
    """Parameter tree with the reference's default init (orthogonal(sqrt 2) kernels, zero biases;
    ``bias_std`` > 0 draws N(0, bias_std^2) biases for tests).  Mirrors ``network.init`` in
    Model.create (module/model.py:42) for the inputs (x (1,A), t (1,1), cond (1,C))."""
    pred = network.predictor
    if pred.dropout and pred.dropout > 0:
        raise NotImplementedError("dropout backbones are outside the fused path")
    gen = torch.Generator().manual_seed(seed)
    p: Dict = {}
    te = network.time_embedding
    d_time = 1
    if te is not None:
        d_time = te.output_dim
        p["MLP_0"] = {"Dense_0": _dense(gen, d_time, d_time, bias_std), "Dense_1": _dense(gen, d_time, d_time, bias_std)}
        if isinstance(te, LearnableFourierEmbedding):
            p["time_embedding"] = {"kernel": (0.2 * torch.randn((d_time // 2, 1), generator=gen)).to(torch.float32)}
    ce = network.cond_embedding
    if ce is not None:   # MLP(hidden_dims=(128, 128), activation=act), no output layer (dacer.py:263-266)
        cdims = [cond_dim, *ce.hidden_dims]
        p["cond_embedding"] = {f"Dense_{i}": _dense(gen, cdims[i], cdims[i + 1], bias_std) for i in range(len(ce.hidden_dims))}
        cond_dim = cdims[-1]
    if is_residual(pred):
        # flax auto-naming inside the @nn.compact bodies: Dense_0 (input projection), <Block>_i/{LayerNorm_0, Dense_0, Dense_1},
        # [LayerNorm_0 (Simba)], Dense_1 (output).  Kernels orthogonal(sqrt 2) for ResidualMLP; Simba's he_normal /
        # orthogonal(1) (simba.py:23-25,49,62) only change the scale of the random test weights, not the layout.
        h, m = int(pred.hidden_dims[0]), int(pred.multiplier)
        q: Dict = {"Dense_0": _dense(gen, x_dim + d_time + cond_dim, h, bias_std)}

        def _ln():
            sc = torch.ones(h) if bias_std == 0.0 else 1.0 + bias_std * torch.randn(h, generator=gen)
            bi = torch.zeros(h) if bias_std == 0.0 else bias_std * torch.randn(h, generator=gen)
            return {"scale": sc.to(torch.float32), "bias": bi.to(torch.float32)}

        for i in range(len(pred.hidden_dims)):
            blk = {"Dense_0": _dense(gen, h, m * h, bias_std), "Dense_1": _dense(gen, m * h, h, bias_std)}
            if pred.layer_norm:
                blk["LayerNorm_0"] = _ln()
            q[f"{pred.block_name}_{i}"] = blk
        if isinstance(pred, Simba):
            q["LayerNorm_0"] = _ln()
        q["Dense_1"] = _dense(gen, h, x_dim, bias_std)
        p[network.predictor_name] = q
        return tree_to(p, device)
    dims = [x_dim + d_time + cond_dim, *pred.hidden_dims, x_dim]
    p[network.predictor_name] = {f"Dense_{i}": _dense(gen, dims[i], dims[i + 1], bias_std) for i in range(len(dims) - 1)}
    if pred.layer_norm:   # flax.linen.LayerNorm: scale = ones, bias = zeros (bias_std > 0 perturbs both, for tests)
        for i, h in enumerate(pred.hidden_dims):
            sc = torch.ones(h) if bias_std == 0.0 else 1.0 + bias_std * torch.randn(h, generator=gen)
            bi = torch.zeros(h) if bias_std == 0.0 else bias_std * torch.randn(h, generator=gen)
            p[network.predictor_name][f"LayerNorm_{i}"] = {"scale": sc.to(torch.float32), "bias": bi.to(torch.float32)}
    return tree_to(p, device)


def tree_to(tree, device):
    if isinstance(tree, dict):
        return {k: tree_to(v, device) for k, v in tree.items()}
    return tree.to(device)


def tree_from_numpy(tree, device="cpu"):
    if isinstance(tree, dict):
        return {k: tree_from_numpy(v, device) for k, v in tree.items()}
    return torch.as_tensor(tree, dtype=torch.float32).contiguous().to(device)
